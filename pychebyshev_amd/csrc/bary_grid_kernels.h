// bary_grid_kernels.h -- k_bary_mfma_grid: the MFMA form of the barycentric contraction for SHORT plans (round 4).
//
// Same GEMM as k_bary_mfma (bary_kernels.h): y(p) = sum_m wM[m,p] (sum_k T2[m,k] wK[k,p]), reference
// barycentric.py:1035-1046.  What differs is how the 16 rows of a row tile are chosen and how the head weights wM
// reach the accumulators.  k_bary_mfma takes 16 CONSECUTIVE rows of the (M x K) view and looks the weight product of
// every row up through a packed row code: 4 codes decoded into 8-16 table addresses, as many LDS reads, 4-12
// multiplies and 4 FMAs per lane, tile and column tile.  With 5-10 k-steps per tile (3-D tensors of 17-40 nodes, 64^4:
// spline pieces, auto-N builds) that epilogue costs as much as the tile's matrix instructions (0.39-0.61 of the peak,
// profiles/r03_bary_rate_probe.txt).  Here a tile is a RA x RB block (RA * RB = 16, RA = 1, 2 or 4) of the LAST TWO
// head dimensions A and B: in the D layout (lane l, register j: row (l >> 4) + 4 j) the lane group g = l >> 4 fixes
// the A index and the B index runs over the registers, so
//     q  += acc[j] * bB[iB(j)]          4 FMAs, the four bB rows at fixed strides from a per-lane base: no codes
//     cv  = (bO * bA) * q               once per CHUNK = all B tiles of one (outer index, A tile)
//     total += cv                       chunk sums in chunk order, then (s0 + s1) + (s2 + s3) over the lane groups
// -- the reference's own nesting (contract B, then A, then the outer dimensions).  A and B are padded to whole tiles
// in the fragment image (zero rows; a zero accumulator times whatever finite table row lies behind the dimension
// contributes +0).  Split launches (small batches) store the chunk sums and k_bary_reduce adds them in the same
// order: a point's value does not depend on the batch it is evaluated in.
#pragma once

#include "pcx_common.h"
#include "bary_weights.h"

// (BaryGridPlan: pcx_common.h)

// frag[t][s][l] for tile t = (chunk, tB): the A operand of v_mfma_f64_16x16x4_f64, tile row r = l & 15 =
// (lane group r & 3, register r >> 2), k = 4 s + (l >> 4)
__global__ void k_pack_fragments_grid(const double *__restrict__ T2, double *__restrict__ frag, BaryGridPlan gp, int K, int KS) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)gp.MT * KS * 64;
    if (idx >= total) return;
    const int l = (int)(idx & 63);
    const long ts = idx >> 6;
    const int s = (int)(ts % KS);
    const long t = ts / KS;
    const int tB = (int)(t % gp.TB);
    const long ch = t / gp.TB;
    const int tA = (int)(ch % gp.TA);
    const long o = ch / gp.TA;
    const int r = l & 15, gq = r & 3, j = r >> 2;
    const int GB = 1 << gp.gbs;
    const int iA = tA * gp.RA + (gq >> gp.gbs);
    const int iB = tB * (16 / gp.RA) + (gq & (GB - 1)) + GB * j;
    const int k = 4 * s + (l >> 4);
    double v = 0.0;
    if (iA < gp.nA && iB < gp.nB && k < K) v = T2[((o * gp.nA + iA) * gp.nB + iB) * (long)K + k];
    frag[idx] = v;
}

__device__ __forceinline__ double grid_code_weight(unsigned code, const double *bw_col, int PW) {
    const double w0 = bw_col[(code & 255u) * PW];
    const double w1 = bw_col[((code >> 8) & 255u) * PW];
    const double w2 = bw_col[((code >> 16) & 255u) * PW];
    const double w3 = bw_col[(code >> 24) * PW];
    return (w0 * w1) * (w2 * w3);
}

__device__ __forceinline__ void grid_weights_1d(double x, const double *__restrict__ nodes, const double *__restrict__ wts,
                                                int n, double *dst, int stride) {
    // (loops kept rolled: unrolled division sequences would set the kernel's register count while the B operands are live)
    int exact = -1;                               // barycentric.py:1039-1043: first node within 1e-14 -> that node's slice
#pragma unroll 1
    for (int j = 0; j < n; ++j) {
        const double diff = x - nodes[j];
        if (exact < 0 && __builtin_fabs(diff) < 1e-14) exact = j;
    }
    if (exact >= 0) {
#pragma unroll 1
        for (int j = 0; j < n; ++j) dst[j * stride] = (j == exact) ? 1.0 : 0.0;
    } else {
        double su = 0.0;
#pragma unroll 1
        for (int j = 0; j < n; ++j) {
            const double u = wts[j] / (x - nodes[j]);
            dst[j * stride] = u;
            su += u;
        }
        const double rcp = 1.0 / su;
#pragma unroll 1
        for (int j = 0; j < n; ++j) dst[j * stride] *= rcp;
    }
}

// The same weights without a division per node (k_bary_small's form, bary_kernels.h): with t_i = (x - x_i) 2^e,
// ONE wave per workgroup (64 threads), PW = 16 NT points per wave: the waves of the row-code kernel share nothing but a
// barrier, and its four-wave workgroups with a table of ALL dimensions' weights (sum_n + 2 rows of PW doubles per wave:
// 66 KB for 21^3, 122 KB for 40^3) left one or two workgroups per CU -- one or two waves per SIMD to hide an L2 round
// trip per row tile and a prologue of ~1,500 vector instructions (measured: the pipe 56 % busy, each wave idle 73 % of
// its life).  Here the table holds the TAIL dimensions only until the B operands are in registers and is then
// overwritten by the HEAD dimensions -- without dimension A, whose weight is formed once per chunk (prologue 3):
// gp.trows = max(outer rows + B rows + slack, tail rows + 1) + 2 rows, 7 KB for 21^3, 11 KB for 40^3, so that 14-22 waves fit a CU.  dynamic LDS = WPB * gp.trows * PW * 8 bytes.  grid = (point blocks, chunk splits, specs).
// WPB = waves per workgroup: 4 wherever two such workgroups fit a CU -- four waves started together walk the fragment image
// in step, so a fragment fetched by one is an L1 hit for the others (with one wave per workgroup every fragment load of every
// wave misses L1: 64^4 0.49 of the peak against 0.76, 30^3 0.50 against 0.54) --, 1 when the table is too large for that.
// AF ("A formed"): dimension A's weight is formed once per chunk instead of being read from the table (prologue 3) -- for
// chunks of >= 48 matrix-instruction pairs (32^3 ... 48^3, 64^4): half the table, twice the waves, one division per chunk;
// shorter chunks (20^3 ... 30^3) keep A in the table (measured: 24^3 0.53 against 0.50 formed, 40^3 0.64 against 0.68).
template <int KS, int NT, int WPB, bool AF>
__global__ void __launch_bounds__(64 * WPB, AF ? ((KS * NT <= 10) ? 4 : ((KS * NT <= 24) ? 3 : 2)) : ((KS * NT <= 20) ? 4 : ((KS * NT <= 32) ? 3 : 2)))
k_bary_mfma_grid(BaryDims dims, BaryMfmaPlan plan, BaryGridPlan gp, const double *__restrict__ nodes,
                 const double *__restrict__ wts, const double *__restrict__ snodes, const double *const *__restrict__ frag_tab,
                 const unsigned *__restrict__ kcode, const double *__restrict__ pts, double *__restrict__ out, long N,
                 long ostride, long ooff, int chunks_per_split, double *__restrict__ partial, const int *__restrict__ perm) {
    // snodes: NULL = weights by division (grid_weights_1d), else scaled nodes followed by the per-dimension scales
    constexpr int PW = 16 * NT;
    constexpr int PH = 64 / PW;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4;
    const int c = lane & 15;
    double *bw = lds + (size_t)(threadIdx.x >> 6) * gp.trows * PW;
    const long base = ((long)blockIdx.x * WPB + (threadIdx.x >> 6)) * PW;
    typedef const double __attribute__((address_space(1))) *gptr_t;
    const gptr_t frag = (gptr_t)frag_tab[blockIdx.z];
    const int ch0 = blockIdx.y * chunks_per_split;
    const int ch1 = (ch0 + chunks_per_split < gp.nchunks) ? ch0 + chunks_per_split : gp.nchunks;
    const int pp = lane % PW;
    const int ph = lane / PW;
    const long pidx0 = base + pp;
    const bool valid = pidx0 < N;
    const long prow = valid ? (perm ? (long)perm[pidx0] : pidx0) : 0;
    const int tail0 = dims.off[plan.split];                   // first node row of the tail dimensions

    // ---- prologue 1: weights of the TAIL dimensions, rows relative to the tail part (what the k codes index) ----
    for (int k = plan.split + ph; k < dims.d; k += PH) {
        const double *nd = nodes + dims.off[k];
        const double x = valid ? pts[prow * dims.d + k] : nd[0];
        double *dst = bw + (size_t)(dims.off[k] - tail0) * PW + pp;
        if (snodes) grid_weights_prod(x, snodes[dims.sum_n + k], snodes + dims.off[k], wts + dims.off[k], dims.n[k], dst, PW);
        else grid_weights_1d(x, nd, wts + dims.off[k], dims.n[k], dst, PW);
    }
    if (ph == 0) bw[(size_t)(dims.sum_n - tail0) * PW + pp] = 1.0;           // the ones row that pads unused code fields
    __syncthreads();

    // ---- prologue 2: B operands (tail weights folded into the K axis) in registers ----
    // (a dead code field names the ones row and a product with exactly 1.0 changes no bit: with one or two tail
    // dimensions only the live fields are read; the fence per k-step keeps the table reads of all k-steps from being
    // in flight at once -- four registers per operand, which would set the kernel's register count)
    double B[NT][KS];
    const int ntail = dims.d - plan.split;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const unsigned code = kcode[4 * s + g];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const double *col = bw + 16 * nt + c;
            if (ntail == 1) B[nt][s] = col[(code & 255u) * PW];
            else if (ntail == 2) B[nt][s] = col[(code & 255u) * PW] * col[((code >> 8) & 255u) * PW];
            else B[nt][s] = grid_code_weight(code, col, PW);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    // ---- prologue 3: weights of the HEAD dimensions over the same rows -- all but dimension A.  A's weight is needed once
    // per chunk and lane group, so its table (as many rows as B's) would halve the waves per CU for nothing: of A only the
    // reciprocal of S = sum_k w_k / (x - x_k) and the index of an exact node (-1: none) are kept per point, and
    // b_A[i] = (w_i / (x - x_i)) (1 / S) is formed where it is used (one division per chunk, lane group and column tile).
    // B sits where A's rows would be (gp.rowB); slack rows behind it are zero. ----
    const int dimA = plan.split - 2;
    double *srow = bw + (size_t)(gp.trows - 2) * PW, *erow = srow + PW;
    for (int k = ph; k < plan.split; k += PH) {
        const double *nd = nodes + dims.off[k];
        const double x = valid ? pts[prow * dims.d + k] : nd[0];
        if (AF && k == dimA) {
            const double *wk = wts + dims.off[k];
            double su = 0.0;
            int exact = -1;
#pragma unroll 1
            for (int j = 0; j < dims.n[k]; ++j) {
                const double diff = x - nd[j];
                if (exact < 0 && __builtin_fabs(diff) < 1e-14) exact = j;      // barycentric.py:1039-1043
                su += wk[j] / diff;
            }
            srow[pp] = 1.0 / su;
            erow[pp] = (double)exact;
            continue;
        }
        double *dst = bw + (size_t)(k == plan.split - 1 ? gp.rowB : (k == dimA ? gp.rowA : dims.off[k])) * PW + pp;
        if (snodes) grid_weights_prod(x, snodes[dims.sum_n + k], snodes + dims.off[k], wts + dims.off[k], dims.n[k], dst, PW);
        else grid_weights_1d(x, nd, wts + dims.off[k], dims.n[k], dst, PW);
    }
    if (ph == 0)
        for (int r = gp.rowB + gp.nB; r < gp.hrows; ++r) bw[(size_t)r * PW + pp] = 0.0;
    __syncthreads();
    double xA[NT], rS[NT];
    int eA[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        xA[nt] = 0.0; rS[nt] = 0.0; eA[nt] = -1;
        if constexpr (AF) {
            const long pidx = base + 16 * nt + c;
            const long rw = pidx < N ? (perm ? (long)perm[pidx] : pidx) : -1;
            xA[nt] = rw >= 0 ? pts[rw * dims.d + dimA] : nodes[dims.off[dimA]];
            rS[nt] = srow[16 * nt + c];
            eA[nt] = (int)erow[16 * nt + c];
        }
    }
    const double *nodeA = nodes + dims.off[dimA], *wtA = wts + dims.off[dimA];

    const int ga = g >> gp.gbs, gb = g & ((1 << gp.gbs) - 1);
    const int jstep = (PW << gp.gbs);                         // table rows GB apart, in doubles
    const int tbstep = (16 / gp.RA) * PW;                     // one B tile further
    const double *colB = bw + (size_t)(gp.rowB + gb) * PW + c;
    const double *colA = bw + (size_t)(gp.rowA + ga) * PW + c;
    const double *colO0 = bw + (size_t)gp.rowo0 * PW + c, *colO1 = bw + (size_t)gp.rowo1 * PW + c;
    const bool split = gridDim.y > 1;
    const gptr_t tf = frag + lane;
    double total[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) total[nt] = 0.0;

    // Fragment loads stay DEPTH k-steps ahead ACROSS tile boundaries, as in k_bary_mfma's long-plan loop.  Plans of up to 12
    // k-steps keep a WHOLE tile ahead: the register a k-step has just consumed is refilled with the same k-step's fragment
    // of the next tile (no copies).  Longer plans run a ring of 8: the first 8 fragments of tile t + 1 are fetched during the
    // tail of tile t into `head` and copied into the ring at the tile's start (the ring's phase would otherwise rotate by
    // KS mod 8 per tile).  Fences keep hipcc from sinking the loads to their uses.
    constexpr bool WHOLE = KS <= 12;
    constexpr int DEPTH = WHOLE ? KS : 8;
    double head[DEPTH];
    const long t_first = (long)ch0 * gp.TB, t_last = (long)ch1 * gp.TB;
    if (t_first < t_last) {
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) head[i] = tf[((size_t)t_first * KS + i) * 64];
    }
    long t = t_first;
    for (int ch = ch0; ch < ch1; ++ch) {
        const int tA = ch % gp.TA;
        const int o = ch / gp.TA;
        const int iA = tA * gp.RA + ga;                        // this lane group's A index; its node and weight are fetched now
        double xA_i = 0.0, wA_i = 0.0;
        if constexpr (AF) { xA_i = nodeA[iA < gp.nA ? iA : 0]; wA_i = wtA[iA < gp.nA ? iA : 0]; }
        double q[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) q[nt] = 0.0;
        for (int tB = 0; tB < gp.TB; ++tB, ++t) {
            pcx_d4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (pcx_d4){0.0, 0.0, 0.0, 0.0};
            double wb[NT][4];
            const double *cb = colB + (size_t)tB * tbstep;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) wb[nt][j] = cb[16 * nt + j * jstep];
            const gptr_t tt = tf + (size_t)t * KS * 64;
            const gptr_t tn = tf + (size_t)(t + 1 < t_last ? t + 1 : t) * KS * 64;
            if constexpr (WHOLE) {
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = head[s];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[nt][s], acc[nt], 0, 0, 0);
                    head[s] = tn[s * 64];                   // the same k-step of the next tile, into the register just read
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                double ring[DEPTH];
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) ring[i] = head[i];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = ring[s % DEPTH];
                    if (s + DEPTH < KS) ring[s % DEPTH] = tt[(s + DEPTH) * 64];
                    else head[s + DEPTH - KS] = tn[(s + DEPTH - KS) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[nt][s], acc[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) q[nt] = __builtin_fma(acc[nt][j], wb[nt][j], q[nt]);
        }
        // chunk end: the A weight of this lane group (formed here, see prologue 3), the outer weights of this chunk
        double wo[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if constexpr (AF) {
                double b = (wA_i / (xA[nt] - xA_i)) * rS[nt];
                if (eA[nt] >= 0) b = (iA == eA[nt]) ? 1.0 : 0.0;
                wo[nt] = iA < gp.nA ? b : 0.0;
            } else {
                wo[nt] = colA[(size_t)tA * gp.RA * PW + 16 * nt];
            }
        }
        if (gp.nouter == 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wo[nt] = colO0[(size_t)o * PW + 16 * nt] * wo[nt];
        } else if (gp.nouter == 2) {
            const int o0 = o / gp.no1, o1 = o - o0 * gp.no1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wo[nt] = (colO0[(size_t)o0 * PW + 16 * nt] * colO1[(size_t)o1 * PW + 16 * nt]) * wo[nt];
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const double cv = wo[nt] * q[nt];
            if (split) {
                const long pidx = base + 16 * nt + c;
                if (pidx < N) partial[(((size_t)blockIdx.z * gp.nchunks + ch) * 4 + g) * (size_t)N + pidx] = cv;
            }
            total[nt] += cv;
        }
    }
    if (!split) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double v = total[nt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const long pidx = base + 16 * nt + c;
            if (g == 0 && pidx < N) {
                const long row = perm ? (long)perm[pidx] : pidx;
                out[row * ostride + ooff + blockIdx.z] = v;
            }
        }
    }
}

// Finishes a split launch with exactly the additions of a non-split one (as k_bary_reduce of bary_kernels.h): per lane
// group the chunk values in chunk order, then (s0 + s1) + (s2 + s3).  partial layout: [spec][chunk][group][point].
__global__ void k_bary_grid_reduce(const double *__restrict__ partial, double *__restrict__ out, long N, int nchunks, int nspec,
                                   long ostride, long ooff, const int *__restrict__ perm) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * nspec) return;
    const long p = idx % N;
    const int z = (int)(idx / N);
    const double *src = partial + (size_t)z * nchunks * 4 * (size_t)N + p;
    double sg[4];
    for (int gq = 0; gq < 4; ++gq) {
        double t = 0.0;
        for (int ch = 0; ch < nchunks; ++ch) t += src[((size_t)ch * 4 + gq) * (size_t)N];
        sg[gq] = t;
    }
    const long row = perm ? (long)perm[p] : p;
    out[row * ostride + ooff + z] = (sg[0] + sg[1]) + (sg[2] + sg[3]);
}
