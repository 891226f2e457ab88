// bary_kernels.h -- HIP kernels for the full-tensor barycentric interpolant (gfx950).
//
// Replaces the per-point NumPy loop of ChebyshevApproximation.vectorized_eval_batch
// (reference barycentric.py:1035-1046) and _apply_derivative_passes (:951-990).
//
//   k_mode_product    K3: T' = T x_axis D  (one fma chain per output, j ascending --
//                         the order OpenBLAS dgemm uses for the reference's arr @ D.T)
//   k_pack_fragments  re-lays the (M x K) view of T' out in MFMA A-fragment order
//   k_bary_mfma       K1+K2 fused: barycentric weights per point, then the dense
//                         contraction as a (M x K) . (K x points) GEMM on
//                         v_mfma_f64_16x16x4_f64 plus a VALU epilogue over the head dims
//   k_bary_rows       K1+K2 for any shape: LPP lanes per point walk the tensor rows
#pragma once

#include "pcx_common.h"

// ---------------------------------------------------------------------------------
// K3: one pass of _apply_derivative_passes: out[o,i,q] = sum_j in[o,j,q] * D[i,j].
// ---------------------------------------------------------------------------------
__global__ void k_mode_product(const double *__restrict__ in, double *__restrict__ out,
                               const double *__restrict__ D, long outer, int na, long inner) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = outer * na * inner;
    if (idx >= total) return;
    long q = idx % inner;
    long t = idx / inner;
    int i = (int)(t % na);
    long o = t / na;
    const double *src = in + (o * na) * inner + q;
    const double *drow = D + (long)i * na;
    double s = 0.0;
    for (int j = 0; j < na; ++j) s = __builtin_fma(src[(long)j * inner], drow[j], s);
    out[idx] = s;
}

// ---------------------------------------------------------------------------------
// Contraction of one tensor axis with a vector: out[o,q] = sum_j in[o,j,q] * vec[j]
// (ChebyshevApproximation.slice: vec = normalised barycentric weights or a one-hot row,
// reference _extrude_slice.py:79-92; the same kernel serves quadrature weights).
// ---------------------------------------------------------------------------------
__global__ void k_contract_axis(const double *__restrict__ in, double *__restrict__ out,
                                const double *__restrict__ vec, long outer, int na, long inner) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= outer * inner) return;
    long q = idx % inner;
    long o = idx / inner;
    const double *src = in + (o * na) * inner + q;
    double s = 0.0;
    for (int j = 0; j < na; ++j) s = __builtin_fma(src[(long)j * inner], vec[j], s);
    out[idx] = s;
}

// ---------------------------------------------------------------------------------
// A-fragment packing: frag[t][s][l] = T2[16 t + (l & 15)][4 s + (l >> 4)], zero padded,
// where T2 is the C-order tensor viewed as (M x K).  One coalesced 512-byte read then
// feeds one v_mfma_f64_16x16x4_f64 (A operand: lane l holds A[l & 15][l >> 4]).
// ---------------------------------------------------------------------------------
__global__ void k_pack_fragments(const double *__restrict__ T2, double *__restrict__ frag, int M,
                                 int K, int MT, int KS) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)MT * KS * 64;
    if (idx >= total) return;
    int l = (int)(idx & 63);
    long ts = idx >> 6;
    int s = (int)(ts % KS);
    int t = (int)(ts / KS);
    int m = 16 * t + (l & 15);
    int k = 4 * s + (l >> 4);
    frag[idx] = (m < M && k < K) ? T2[(long)m * K + k] : 0.0;
}

// Slab packing for dim-0 groups (BaryG0): the tensor viewed as (n0 x M1 x K); slab i0 is packed like a
// tensor of its own into `tps` row tiles (M1 rows padded to 16 tps), so no row tile straddles two i0.
__global__ void k_pack_fragments_slabs(const double *__restrict__ T3, double *__restrict__ frag, int n0, int M1,
                                       int K, int tps, int KS) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per_slab = (long)tps * KS * 64;
    if (idx >= per_slab * n0) return;
    const int i0 = (int)(idx / per_slab);
    const long r = idx - (long)i0 * per_slab;
    int l = (int)(r & 63);
    long ts = r >> 6;
    int s = (int)(ts % KS);
    int t = (int)(ts / KS);
    int m = 16 * t + (l & 15);
    int k = 4 * s + (l >> 4);
    frag[idx] = (m < M1 && k < K) ? T3[((long)i0 * M1 + m) * K + k] : 0.0;
}

// Finish of a dim-0 group for ONE point: vec (n0 entries at vecA[i * stride]) holds the per-i0 partial sums P;
// for o = 0 .. maxorder: y_o = sum_i b0[i] (D_0^o P)[i], written to the columns of the members with order o.
// D_0 is wave-uniform (scalar loads); one j-ascending fma chain per entry, like k_mode_product.
__device__ __forceinline__ void bary_g0_finish(const BaryG0 &gs, const double *__restrict__ diff0, double *vecA,
                                               double *vecB, const double *b0, int stride, double *out_row) {
    typedef const double __attribute__((address_space(4))) *cptr_t;
    const cptr_t D0 = (cptr_t)(unsigned long long)diff0;
    const int n0 = gs.n0;
    for (int o = 0; o <= gs.maxorder; ++o) {
        if (o > 0) {
            for (int i = 0; i < n0; ++i) {
                double s = 0.0;
                for (int j = 0; j < n0; ++j) s = __builtin_fma(vecA[j * stride], D0[i * n0 + j], s);
                vecB[i * stride] = s;
            }
            double *t = vecA; vecA = vecB; vecB = t;
        }
        double y = 0.0;
        for (int i = 0; i < n0; ++i) y = __builtin_fma(b0[i * stride], vecA[i * stride], y);
        if (out_row)
            for (int s = 0; s < gs.nmem; ++s)
                if (gs.order[s] == o) out_row[gs.col[s]] = y;
    }
}

// ---------------------------------------------------------------------------------
// K1: normalised barycentric weights of one coordinate for one dimension
// (reference barycentric.py:1039-1045 / :1083-1094): first node with |x - node| < 1e-14
// gives a one-hot row, otherwise u_j = w_j / (x - node_j), b_j = u_j / sum(u).
// Written to dst[j * stride].  (The division-free form of bary_weights.h was tried here in round 4: 11^5 +1 %, 7^5 -4 %,
// 15^4 -2 %, the 4-D shapes unchanged, 64^2 -11 %, 40 x 64 -13 % -- the row-code kernel's prologue is latency (three dependent
// passes per dimension against divisions that pipeline), not instruction count; not kept.)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void bary_weights_1d(double x, const double *__restrict__ nodes,
                                                const double *__restrict__ wts, int n, double *dst,
                                                int stride) {
    int exact = -1;
    for (int j = 0; j < n; ++j) {
        double diff = x - nodes[j];
        if (exact < 0 && __builtin_fabs(diff) < 1e-14) exact = j;
    }
    if (exact >= 0) {
        for (int j = 0; j < n; ++j) dst[j * stride] = (j == exact) ? 1.0 : 0.0;
    } else {
        double su = 0.0;
        for (int j = 0; j < n; ++j) {
            double u = wts[j] / (x - nodes[j]);
            dst[j * stride] = u;
            su += u;
        }
        double r = 1.0 / su;
        for (int j = 0; j < n; ++j) dst[j * stride] *= r;
    }
}

__device__ __forceinline__ double code_weight(unsigned code, const double *bw_col, int PW) {
    // product of the four table rows named by the 8-bit fields of `code`
    double w0 = bw_col[(code & 255u) * PW];
    double w1 = bw_col[((code >> 8) & 255u) * PW];
    double w2 = bw_col[((code >> 16) & 255u) * PW];
    double w3 = bw_col[(code >> 24) * PW];
    return (w0 * w1) * (w2 * w3);
}

// NF live fields, known at compile time (fields are filled low first; a dead field names the all-ones row and
// a product with exactly 1.0 changes nothing: bit-identical to code_weight).  NF = 2 serves short plans whose
// head has at most two dimensions: their few matrix instructions per tile cost no more than these look-ups.
// The four row codes of lane group g for tile t (rows g, g + 4, g + 8, g + 12): stored side by side
// ([t][g][j], see pcx_bary_create), one 16-byte load.
struct pcx_u4 { unsigned v[4]; };
__device__ __forceinline__ pcx_u4 load_row_codes(const unsigned *__restrict__ codes, long t, int g) {
    const uint4 q = reinterpret_cast<const uint4 *>(codes)[4 * t + g];
    return pcx_u4{{q.x, q.y, q.z, q.w}};
}

template <int NF>
__device__ __forceinline__ double code_weight_t(unsigned code, const double *bw_col, int PW) {
    if constexpr (NF >= 4) return code_weight(code, bw_col, PW);
    else {
        double w0 = bw_col[(code & 255u) * PW];
        double w1 = bw_col[((code >> 8) & 255u) * PW];
        if constexpr (NF == 3) return (w0 * w1) * bw_col[((code >> 16) & 255u) * PW];
        else return w0 * w1;
    }
}

// ---------------------------------------------------------------------------------
// K1+K2 fused, MFMA form.  One wave owns PW = 16*NT query points.
//
//   prologue  weights of every dimension for the wave's points -> LDS table
//             bw[row][point], rows = concatenated dims + one row of ones;
//             B operands  B[nt][s] (lane l: k = 4s + (l>>4), point 16nt + (l&15))
//             = product of the tail-dim weights named by kcode[k], kept in VGPRs.
//   main      row tiles are walked in CHUNKS of PCX_CHUNK_TILES; for each tile t:
//             acc[nt] = sum_s mfma(A = frag[t][s], B[nt][s]);
//             D layout: lane l, reg j holds row (l>>4) + 4j, column (point) l & 15;
//             cs[nt] += acc[nt][j] * (head-dim weight product named by rowcode[...]).
//   every PCX_CHUNK_TILES tiles the per-lane chunk sum is added to the per-lane total.
//
// Summation order is FIXED by the chunking, not by the launch geometry: per lane group g
//   s_g = ((cs_0 + cs_1) + cs_2) + ...  over the chunks, then y = (s_0 + s_1) + (s_2 + s_3).
// grid.y = 1: one workgroup walks all chunks, adds the lane groups by two shuffles, stores y.
// grid.y > 1 ("split" launches for small batches): block y handles chunks
//   [y*cps, (y+1)*cps) and stores each per-lane chunk sum to partial[z][chunk][g][p];
//   k_bary_reduce then performs the same additions in the same order -> results are
//   bit-identical for every batch size.
// grid.z = number of derivative specs evaluated in this launch (frag_tab[z]).
//
// 256 threads = 4 waves; dynamic LDS = 4 * (sum_n + 2) * PW * 8 bytes.
// ---------------------------------------------------------------------------------
#define PCX_CHUNK_TILES 4

// WIDE: more than four head or tail dimensions (d up to 16): every code has a second word
// (rowcode_hi / kcode_hi, fields 4..7) and a weight is the product of both words' products.
// G0: dim-0 group launch (BaryG0, pcx_common.h): the fragment image is slab-packed (plan.MT = n0 * tps row tiles),
// the row codes name the head dimensions 1 .. split-1 only, at the end of every slab the wave's partial sum
// P[i0] is reduced over the four lane groups into the (by then dead) tail part of its LDS table, and the
// epilogue finishes all members of the group from P (bary_g0_finish).  grid.y = grid.z = 1.
template <int KS, int NT, bool WIDE, int NF = 4, bool G0 = false>
__global__ void __launch_bounds__(256, 2)
k_bary_mfma(BaryDims dims, BaryMfmaPlan plan, const double *__restrict__ nodes,
            const double *__restrict__ wts, const double *const *__restrict__ frag_tab,
            const unsigned *__restrict__ rowcode, const unsigned *__restrict__ kcode,
            const unsigned *__restrict__ rowcode_hi, const unsigned *__restrict__ kcode_hi,
            const double *__restrict__ pts, double *__restrict__ out, long N, long ostride,
            long ooff, int chunks_per_split, double *__restrict__ partial,
            const int *__restrict__ perm, BaryG0 gs, const double *__restrict__ diff0) {
    // perm (optional): the launch covers the N rows perm[0..N) of pts/out (a bucket of a
    // piecewise interpolant) instead of rows 0..N.
    static_assert(NT == 1 || NT == 2 || NT == 4, "PW must divide the wave");
    constexpr int PW = 16 * NT;
    constexpr int PH = 64 / PW;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int c = lane & 15;
    double *bw = lds + (size_t)wave * plan.rows * PW;
    const double *bwt = bw + (size_t)plan.tail_base * PW;      // tail part: what the k codes index
    const long base = ((long)blockIdx.x * 4 + wave) * PW;
    // a pointer loaded from memory is "generic" to the compiler (flat_load + combined
    // vmcnt/lgkmcnt waits); it is known to be global memory, say so
    typedef const double __attribute__((address_space(1))) *gptr_t;
    const gptr_t frag = (gptr_t)frag_tab[blockIdx.z];
    const int nchunks = (plan.MT + PCX_CHUNK_TILES - 1) / PCX_CHUNK_TILES;
    const int ch0 = blockIdx.y * chunks_per_split;
    const int ch1 = (ch0 + chunks_per_split < nchunks) ? ch0 + chunks_per_split : nchunks;

    // ---- prologue 1: barycentric weights (lane -> point lane % PW, dims strided by PH)
    {
        const int pp = lane % PW;
        const int ph = lane / PW;
        const long pidx = base + pp;
        const bool valid = pidx < N;
        const long row = valid ? (perm ? (long)perm[pidx] : pidx) : 0;
        for (int k = ph; k < dims.d; k += PH) {
            const double *nd = nodes + dims.off[k];
            double x = valid ? pts[row * dims.d + k] : nd[0];
            const int trow = dims.off[k] + (k >= plan.split ? 1 : 0);
            bary_weights_1d(x, nd, wts + dims.off[k], dims.n[k], bw + (size_t)trow * PW + pp, PW);
        }
        if (ph == 0) {
            bw[(size_t)(plan.tail_base - 1) * PW + pp] = 1.0;
            bw[(size_t)(plan.rows - 1) * PW + pp] = 1.0;
        }
    }
    __syncthreads();

    // ---- prologue 2: B operands in registers
    double B[NT][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        unsigned code = kcode[4 * s + g];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) B[nt][s] = code_weight(code, bwt + 16 * nt + c, PW);
        if (WIDE) {
            unsigned hi = kcode_hi[4 * s + g];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) B[nt][s] *= code_weight(hi, bwt + 16 * nt + c, PW);
        }
    }

    // ---- main loop over row tiles; every PCX_CHUNK_TILES tiles the per-lane chunk sum cs
    //      is folded into the per-lane total (or stored, in a split launch)
    double total[NT], cs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { total[nt] = 0.0; cs[nt] = 0.0; }
    const gptr_t tf = frag + lane;
    const int t_begin = ch0 * PCX_CHUNK_TILES;
    const int t_end = (ch1 * PCX_CHUNK_TILES < plan.MT) ? ch1 * PCX_CHUNK_TILES : plan.MT;
    const bool split = gridDim.y > 1;
    // Long narrow plans (the 11^5 headline: 31 k-steps) run a hand-pipelined tile loop: fragment
    // loads stay DEPTH k-steps ahead ACROSS tile boundaries (the first DEPTH fragments of tile t+1
    // are fetched during the tail of tile t), the row codes of tile t+1 are fetched at the top of
    // tile t (loads return in order: waiting for codes issued behind a tile's own fragment loads
    // drained them all, once per tile), and the head-weight look-ups of row j are issued at k-step
    // 2j and multiplied two k-steps later.  Fences keep hipcc from sinking the loads back to
    // their uses.  Same arithmetic in the same order as the plain loop below: identical results.
    constexpr bool PIPELINED = (KS >= 12);
    constexpr int DEPTH = 6;           // <= 12 <= KS: the ring never wraps a tile (A/B on one box: 4 -0.6 %, 8 -0.3 %)
    unsigned cn[4] = {0u, 0u, 0u, 0u}, cnh[4] = {0u, 0u, 0u, 0u};
    double head[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) head[i] = 0.0;
    if (PIPELINED && t_begin < t_end) {
        {
            const pcx_u4 q = load_row_codes(rowcode, t_begin, g);
            pcx_u4 qh = q;
            if (WIDE) qh = load_row_codes(rowcode_hi, t_begin, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cn[j] = q.v[j];
                if (WIDE) cnh[j] = qh.v[j];
            }
        }
#pragma unroll
        for (int i = 0; i < DEPTH; ++i) head[i] = tf[((size_t)t_begin * KS + i) * 64];
    }
    for (int t = t_begin; t < t_end; ++t) {
        double w[NT][4];
        pcx_d4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = (pcx_d4){0.0, 0.0, 0.0, 0.0};
        const gptr_t tt = tf + (size_t)t * KS * 64;
        if constexpr (PIPELINED) {
            const int t_next = (t + 1 < t_end) ? t + 1 : t;
            const gptr_t tn = tf + (size_t)t_next * KS * 64;
            double ring[DEPTH];
            unsigned cc[4], cch[4];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) ring[i] = head[i];
            {
                const pcx_u4 q = load_row_codes(rowcode, t_next, g);
                pcx_u4 qh = q;
                if (WIDE) qh = load_row_codes(rowcode_hi, t_next, g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    cc[j] = cn[j];
                    cn[j] = q.v[j];
                    if (WIDE) { cch[j] = cnh[j]; cnh[j] = qh.v[j]; }
                }
            }
            double wr[4][NT][4];          // raw table entries of row j, looked up at k-step 2j
            double wh[4][NT][4];          // ... and of its second code word (wide plans), at k-step 2j + 1
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double a = ring[s % DEPTH];
                if (s + DEPTH < KS) ring[s % DEPTH] = tt[(s + DEPTH) * 64];
                else head[s + DEPTH - KS] = tn[(s + DEPTH - KS) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (s == 2 * j) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int f = 0; f < (NF < 4 ? NF : 4); ++f)          // dead fields name the ones row: not read
                                wr[j][nt][f] = bw[(size_t)((cc[j] >> (8 * f)) & 255u) * PW + 16 * nt + c];
                    }
                    if (s == 2 * j + 2) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            w[nt][j] = NF >= 4 ? (wr[j][nt][0] * wr[j][nt][1]) * (wr[j][nt][2] * wr[j][nt][3])
                                     : (NF == 3 ? (wr[j][nt][0] * wr[j][nt][1]) * wr[j][nt][2]
                                                : wr[j][nt][0] * wr[j][nt][1]);
                    }
                    if (WIDE && s == 2 * j + 1) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int f = 0; f < 4; ++f)
                                wh[j][nt][f] = bw[(size_t)((cch[j] >> (8 * f)) & 255u) * PW + 16 * nt + c];
                    }
                    if (WIDE && s == 2 * j + 3) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            w[nt][j] *= (wh[j][nt][0] * wh[j][nt][1]) * (wh[j][nt][2] * wh[j][nt][3]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[nt][s], acc[nt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            const pcx_u4 q = load_row_codes(rowcode, t, g);
            pcx_u4 qh = q;
            if (WIDE) qh = load_row_codes(rowcode_hi, t, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned code = q.v[j];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) w[nt][j] = code_weight_t<NF>(code, bw + 16 * nt + c, PW);
                if (WIDE) {
                    unsigned hi = qh.v[j];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) w[nt][j] *= code_weight(hi, bw + 16 * nt + c, PW);
                }
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                double a = tt[s * 64];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[nt][s], acc[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[nt] = __builtin_fma(acc[nt][j], w[nt][j], cs[nt]);
        if constexpr (G0) {
            if ((t + 1) % gs.tps == 0) {            // end of slab i0: P[i0] = (s0 + s1) + (s2 + s3) -> LDS
                const int i0 = t / gs.tps;
                double *Pl = const_cast<double *>(bwt);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    double v = cs[nt];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (g == 0) Pl[(size_t)i0 * PW + 16 * nt + c] = v;
                    cs[nt] = 0.0;
                }
            }
            continue;
        }
        const bool chunk_end = ((t + 1) % PCX_CHUNK_TILES == 0) || (t + 1 == plan.MT);
        if (chunk_end) {
            if (split) {
                const int ch = t / PCX_CHUNK_TILES;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    long pidx = base + 16 * nt + c;
                    if (pidx < N)
                        partial[(((size_t)blockIdx.z * nchunks + ch) * 4 + g) * (size_t)N + pidx] = cs[nt];
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { total[nt] += cs[nt]; cs[nt] = 0.0; }
        }
    }

    if constexpr (G0) {
        // one lane per point finishes the group from P (wave-private LDS: a wave's LDS operations are in order)
        if (lane < PW) {
            double *Pl = const_cast<double *>(bwt) + lane;
            const long pidx = base + lane;
            const long row = (pidx < N) ? (perm ? (long)perm[pidx] : pidx) : 0;
            bary_g0_finish(gs, diff0, Pl, Pl + (size_t)gs.n0 * PW, bw + (size_t)dims.off[0] * PW + lane, PW,
                           pidx < N ? out + row * ostride + ooff : nullptr);
        }
        return;
    }
    // ---- add the four 16-lane groups: lane group 0 ends with (s0 + s1) + (s2 + s3)
    if (!split) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double v = total[nt];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            long pidx = base + 16 * nt + c;
            if (g == 0 && pidx < N) {
                long row = perm ? (long)perm[pidx] : pidx;
                out[row * ostride + ooff + blockIdx.z] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// K1+K2 fused, MFMA form built on v_mfma_f64_4x4x4_4b_f64 (large batches).
//
// Why: tools/fp64_peak.hip + tools/fp64_mfma4x4.hip measure 66-67 TFLOP/s for back-to-back
// v_mfma_f64_16x16x4_f64 (it issues every ~74 cycles, not 64) but 76 TFLOP/s (97 % of peak)
// for the four-block 4x4x4 form, which issues every 16 cycles.  Same arithmetic, same
// operands: a 16-row tile x 4 k-step is four instructions, one per group of 4 rows.
//   lane map (probed, tools/mfma4x4_probe.hip): A: lane = 16k + 4b + i, B: lane = 16k + 4b + j,
//   D: lane = 16i + 4b + j  (b = block).  Blocks = the four 4-point groups of a 16-point
//   column tile, so B[nt][s] is EXACTLY the B operand of the 16x16x4 kernel (k = l>>4,
//   point = l&15) and D = row 4rg + (l>>4), point l&15: a lane sees the same rows
//   {g, g+4, g+8, g+12} in the same order -> bit-identical results to k_bary_mfma.
//   A must be the same for the four blocks: the row tile is staged in LDS (a straight copy
//   of frag[t], double-buffered, one barrier per tile) and read with a broadcast
//   ds_read_b64 (16 distinct addresses per wave-instruction, conflict-free).
// 512 threads = 8 waves share each staged tile; every wave owns 32 points (NT = 2).
// dynamic LDS = 8 * (sum_n + 2) * 32 * 8 + 2 * KS * 64 * 8 bytes.
// ---------------------------------------------------------------------------------
template <int KS>
__global__ void __launch_bounds__(512, 2)
k_bary_mfma4(BaryDims dims, BaryMfmaPlan plan, const double *__restrict__ nodes,
             const double *__restrict__ wts, const double *const *__restrict__ frag_tab,
             const unsigned *__restrict__ rowcode, const unsigned *__restrict__ kcode,
             const double *__restrict__ pts, double *__restrict__ out, long N, long ostride,
             long ooff, const int *__restrict__ perm) {
    constexpr int NT = 2;
    constexpr int PW = 32;
    constexpr int SLAB = KS * 64;                 // doubles per staged row tile
    constexpr int CPT = (SLAB + 511) / 512;       // doubles each thread copies per tile
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int c = lane & 15;
    double *bw = lds + (size_t)wave * plan.rows * PW;
    const double *bwt = bw + (size_t)plan.tail_base * PW;
    double *slab = lds + (size_t)8 * plan.rows * PW;
    const long base = ((long)blockIdx.x * 8 + wave) * PW;
    typedef const double __attribute__((address_space(1))) *gptr_t;
    const gptr_t frag = (gptr_t)frag_tab[blockIdx.z];

    // ---- prologue 1: barycentric weights (lane -> point lane % 32, dims strided by 2)
    {
        const int pp = lane & 31;
        const int ph = lane >> 5;
        const long pidx = base + pp;
        const bool valid = pidx < N;
        const long row = valid ? (perm ? (long)perm[pidx] : pidx) : 0;
        for (int k = ph; k < dims.d; k += 2) {
            const double *nd = nodes + dims.off[k];
            double x = valid ? pts[row * dims.d + k] : nd[0];
            const int trow = dims.off[k] + (k >= plan.split ? 1 : 0);
            bary_weights_1d(x, nd, wts + dims.off[k], dims.n[k], bw + (size_t)trow * PW + pp, PW);
        }
        if (ph == 0) {
            bw[(size_t)(plan.tail_base - 1) * PW + pp] = 1.0;
            bw[(size_t)(plan.rows - 1) * PW + pp] = 1.0;
        }
    }
    // first tile -> LDS, second tile -> registers (in flight)
    double stage[CPT];
    auto img_index = [](int i) {       // frag index (s, l = 16k + 4rg + ii)  ->  LDS image index
        const int s_ = i >> 6, l_ = i & 63;
        const int k_ = l_ >> 4, rg_ = (l_ >> 2) & 3, ii_ = l_ & 3;
        return s_ * 64 + (rg_ >> 1) * 32 + (k_ * 4 + ii_) * 2 + (rg_ & 1);
    };
#pragma unroll
    for (int r = 0; r < CPT; ++r) {
        const int i = threadIdx.x + 512 * r;
        if (i < SLAB) slab[img_index(i)] = frag[i];
    }
    if (plan.MT > 1) {
#pragma unroll
        for (int r = 0; r < CPT; ++r) {
            const int i = threadIdx.x + 512 * r;
            stage[r] = (i < SLAB) ? frag[(size_t)SLAB + i] : 0.0;
        }
    }
    __syncthreads();

    // ---- prologue 2: B operands in registers (same layout as the 16x16x4 kernel)
    double B[NT][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        unsigned code = kcode[4 * s + g];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) B[nt][s] = code_weight(code, bwt + 16 * nt + c, PW);
    }

    double total[NT], cs[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { total[nt] = 0.0; cs[nt] = 0.0; }
    // LDS image of a tile: img[s][half][k][i][pair] = frag[s][16k + 4(2 half + pair) + i], so the A
    // operands of the two row groups a lane multiplies together are one aligned 16-byte read
    const int aoff = ((lane >> 4) * 4 + (lane & 3)) * 2;
    for (int t = 0; t < plan.MT; ++t) {
        const double *cur = slab + (size_t)(t & 1) * SLAB;
        double *nxt = slab + (size_t)((t + 1) & 1) * SLAB;
        if (t + 1 < plan.MT) {      // park tile t+1 (free since the barrier that ended tile t-1)
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
                const int i = threadIdx.x + 512 * r;
                if (i < SLAB) nxt[img_index(i)] = stage[r];
            }
        }
        if (t + 2 < plan.MT) {      // fetch tile t+2 while tile t is multiplied
            const gptr_t src = frag + (size_t)(t + 2) * SLAB;
#pragma unroll
            for (int r = 0; r < CPT; ++r) {
                const int i = threadIdx.x + 512 * r;
                stage[r] = (i < SLAB) ? src[i] : 0.0;
            }
        }
        // the tile's four row codes of this lane group, fetched before the MFMA chains
        const pcx_u4 q4 = load_row_codes(rowcode, t, g);
        const unsigned code0 = q4.v[0], code1 = q4.v[1], code2 = q4.v[2], code3 = q4.v[3];
        // two row groups at a time: four independent accumulator chains, and the broadcast
        // LDS reads of the next DEPTH k-steps are in flight while the current one multiplies
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            constexpr int DEPTH = (KS < 6) ? KS : 6;
            double acc[2][NT];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[r][nt] = 0.0;
            typedef double d2_t __attribute__((ext_vector_type(2)));
            const d2_t *ar = reinterpret_cast<const d2_t *>(cur + aoff + 32 * half);   // 32 pairs per k-step
            d2_t ring[DEPTH];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) ring[s] = ar[s * 32];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double a0 = ring[s % DEPTH][0], a1 = ring[s % DEPTH][1];
                if (s + DEPTH < KS) ring[s % DEPTH] = ar[(s + DEPTH) * 32];
                // keep the prefetch ahead of the MFMAs: without the fence hipcc sinks every LDS
                // read to just before its use (one register pair, lgkmcnt(0) per k-step)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, B[nt][s], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, B[nt][s], acc[1][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned codeA = half ? code2 : code0, codeB = half ? code3 : code1;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {      // rows g + 8*half, then g + 8*half + 4: ascending
                cs[nt] = __builtin_fma(acc[0][nt], code_weight(codeA, bw + 16 * nt + c, PW), cs[nt]);
                cs[nt] = __builtin_fma(acc[1][nt], code_weight(codeB, bw + 16 * nt + c, PW), cs[nt]);
            }
        }
        const bool chunk_end = ((t + 1) % PCX_CHUNK_TILES == 0) || (t + 1 == plan.MT);
        if (chunk_end) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { total[nt] += cs[nt]; cs[nt] = 0.0; }
        }
        __syncthreads();            // tile t+1 visible; tile t's buffer free
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        double v = total[nt];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        long pidx = base + 16 * nt + c;
        if (g == 0 && pidx < N) {
            long row = perm ? (long)perm[pidx] : pidx;
            out[row * ostride + ooff + blockIdx.z] = v;
        }
    }
}

// Finishes a split launch with exactly the additions of a non-split one: per lane group g
// the chunk sums in chunk order, s_g = ((cs_0 + cs_1) + cs_2) + ..., then (s0 + s1) + (s2 + s3).
// partial layout: [spec][chunk][group][point].
__global__ void k_bary_reduce(const double *__restrict__ partial, double *__restrict__ out, long N,
                              int nchunks, int nspec, long ostride, long ooff,
                              const int *__restrict__ perm) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * nspec) return;
    long p = idx % N;
    int z = (int)(idx / N);
    const double *src = partial + (size_t)z * nchunks * 4 * (size_t)N + p;
    double sg[4];
    for (int g = 0; g < 4; ++g) {
        double t = 0.0;
        for (int ch = 0; ch < nchunks; ++ch) t += src[((size_t)ch * 4 + g) * (size_t)N];
        sg[g] = t;
    }
    long row = perm ? (long)perm[p] : p;
    out[row * ostride + ooff + z] = (sg[0] + sg[1]) + (sg[2] + sg[3]);
}

// ---------------------------------------------------------------------------------
// K1+K2, row-parallel VALU form for any shape.  LPP (power of two <= 64) lanes share
// one point: weights -> LDS, then lane r walks rows r, r+LPP, ... of the (M x K) view
// with K = n[d-1], accumulating wM(row) * sum_k T[row,k] b_last[k]; shuffle-reduce.
// dynamic LDS = (256 / LPP) * sum_n * 8 bytes.
// ---------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_bary_rows(BaryDims dims, int LPP, const double *__restrict__ nodes,
            const double *__restrict__ wts, const double *__restrict__ T,
            const double *__restrict__ pts, double *__restrict__ out, long N, long ostride,
            long ooff, const int *__restrict__ perm) {
    extern __shared__ double lds[];
    const int ppw = 256 / LPP;                 // points per workgroup
    const int pl = threadIdx.x / LPP;          // local point
    const int sub = threadIdx.x % LPP;         // lane within the point's group
    const long pidx = (long)blockIdx.x * ppw + pl;
    const bool valid = pidx < N;
    const long row = valid ? (perm ? (long)perm[pidx] : pidx) : 0;
    double *bw = lds + (size_t)pl * dims.sum_n;
    for (int k = sub; k < dims.d; k += LPP) {
        const double *nd = nodes + dims.off[k];
        double x = valid ? pts[row * dims.d + k] : nd[0];
        bary_weights_1d(x, nd, wts + dims.off[k], dims.n[k], bw + dims.off[k], 1);
    }
    __syncthreads();
    const int d = dims.d;
    const int K = dims.n[d - 1];
    long M = 1;
    for (int k = 0; k < d - 1; ++k) M *= dims.n[k];
    const double *bl = bw + dims.off[d - 1];
    double acc = 0.0;
    for (long m = sub; m < M; m += LPP) {
        double w = 1.0;
        long rem = m;
        for (int k = d - 2; k >= 0; --k) {
            int nk = dims.n[k];
            int i = (int)(rem % nk);
            rem /= nk;
            w *= bw[dims.off[k] + i];
        }
        const double *row = T + m * K;
        double s = 0.0;
        for (int j = 0; j < K; ++j) s = __builtin_fma(row[j], bl[j], s);
        acc = __builtin_fma(s, w, acc);
    }
    for (int o = LPP >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (sub == 0 && valid) out[row * ostride + ooff] = acc;
}

// ---------------------------------------------------------------------------------
// K1+K2 for SMALL tensors (d <= 4, a few thousand elements): one lane = one point.
//
// For shapes like 12 x 12 (BASELINE config 1) or the 9 x 7 x 6 pieces of a spline the MFMA
// kernel is all prologue: 3 matrix instructions per 32 points behind ~500 instructions of
// weights, code look-ups and table traffic.  Here every lane owns a point and walks the whole
// tensor in the reference's own nesting (barycentric.py:1036-1046: contract the last
// dimension, then the one before, ...):
//     y = sum_{i0} b0[i0] ( sum_{i1} b1[i1] ( ... sum_{iL} T[i0, i1, ..., iL] bL[iL] ) )
// The tensor element is wave-uniform: it is read with SCALAR loads (s_load, scalar cache /
// L2) and enters v_fma_f64 as an SGPR operand -- one 8-byte load feeds 64 FMAs, and neither
// LDS (which could not feed four SIMDs one broadcast read per FMA) nor VGPR space is spent
// on it.  The last dimension's weights live in registers (NLP = n_last rounded up, zeros
// beyond n_last; the row read runs up to NLP - n_last doubles into the next row, which is
// multiplied by those zeros: tensors carry PCX_PLAIN_PAD zeroed doubles behind their end), the
// outer dimensions' weights in a per-wave LDS table [row][lane], read once per inner row.
// One FMA per tensor element and one per inner result: the algorithmic count.  (Tried and dropped: fetching
// row i + 1 into a second SGPR set while row i is multiplied -- scalar loads return out of order, every wait
// is lgkmcnt(0), and the ping-pong version was 0-15 % slower on 8^3 ... 16^3; the waves of other workgroups
// hide the load latency better.)
// 64 threads per workgroup; dynamic LDS = (sum of outer n) * 64 * 8 bytes.
// ---------------------------------------------------------------------------------
// (PCX_PLAIN_PAD, the zeroed doubles behind a plain tensor that these fixed-width scalar loads may reach, is in pcx_common.h)

// Normalised barycentric weights WITHOUT a division per node: with t_i = (x - x_i) 2^e,
//     b_j = w_j prod_{i != j} t_i / sum_k w_k prod_{i != k} t_i          (the common factor cancels)
// from running prefix and suffix products: 8 vector instructions per node and one division per
// dimension, against ~18 per node for u_j = w_j / (x - x_j) with its IEEE division sequence --
// for a 12 x 12 tensor the weights, not the 156 FMAs of the contraction, are most of the work.
// 2^e (a power of two near 2 / width, so that the products of up to 64 factors stay far from
// over/underflow) is applied exactly: snodes = nodes 2^e is precomputed and t_i = fma(x, 2^e,
// -snodes_i) is the correctly rounded (x - x_i) 2^e.  Within 1e-14 of a node the
// reference switches to the node's slice (barycentric.py:1039-1043) and so does this: a divergent fix-up
// behind one v_min per node, see bary_weights_reg (round 3; round 2 evaluated the interpolant at x there).
// Model data the evaluation kernels only read (tensor, scaled nodes, weights) goes through a
// CONSTANT-address-space pointer: with a wave-uniform address that is what lets hipcc use scalar loads
// (s_load into SGPRs) whatever it can or cannot prove about aliasing with `out` -- a plain `const double *`
// fell back to 64-lane vector loads of the same address, one exposed memory round trip per node.
typedef const double __attribute__((address_space(4))) *pcx_cptr;
__device__ __forceinline__ pcx_cptr pcx_as_constant(const double *p) { return (pcx_cptr)(unsigned long long)p; }

// EXACT: the dimension has exactly NLP nodes -- no per-node `j < n` tests (hipcc turns those into six
// v_cndmask per node in the suffix pass and a branch + dependent load per node in the prefix pass).
template <int NLP, bool EXACT>
__device__ __forceinline__ void bary_weights_reg(double x, double scale, pcx_cptr snodes, pcx_cptr wts, int n,
                                                 double (&b)[NLP]) {
    double t[NLP];
    double run = 1.0, amin = 1.0e300;
#pragma unroll
    for (int j = 0; j < NLP; ++j) {                 // b_j <- w_j * prefix_j
        b[j] = 0.0;
        t[j] = 1.0;
        if (EXACT || j < n) {
            t[j] = __builtin_fma(x, scale, -snodes[j]);
            amin = __builtin_fmin(amin, __builtin_fabs(t[j]));
            b[j] = wts[j] * run;
            run *= t[j];
        }
    }
    run = 1.0;
    double su = 0.0;
#pragma unroll
    for (int j = NLP - 1; j >= 0; --j) {            // b_j <- b_j * suffix_j
        if (EXACT || j < n) {
            b[j] *= run;
            su += b[j];
            run *= t[j];
        }
    }
    const double r = 1.0 / su;
#pragma unroll
    for (int j = 0; j < NLP; ++j) b[j] *= r;
    // The reference's node rule, literally (barycentric.py:1039-1043): within 1e-14 of a node -- |x - x_j| < 1e-14 is
    // |t_j| < 1e-14 2^e exactly, t_j being the correctly rounded (x - x_j) 2^e -- the FIRST such node's slice is taken.
    // One v_min per node on the common path; the lanes it concerns (grid points, mostly) rewrite their row as one-hot,
    // which also makes the weight of an exact node exactly 1.0 (c (1 / c) is not always 1).  Round 2 evaluated the
    // interpolant at x there: O(1e-14 |f'|) away, which is no longer inside 1e-12 for 30 noisy nodes.
    if (amin < 1e-14 * scale) {
        bool found = false;
#pragma unroll
        for (int j = 0; j < NLP; ++j) {
            const bool hit = !found && (EXACT || j < n) && __builtin_fabs(t[j]) < 1e-14 * scale;
            b[j] = hit ? 1.0 : 0.0;
            found = found || hit;
        }
    }
}

// The same through a lane's column of an LDS table (runtime node count): dst[j * stride].
__device__ __forceinline__ void bary_weights_prod(double x, double scale, pcx_cptr snodes, pcx_cptr wts, int n,
                                                  double *dst, int stride) {
    double run = 1.0, amin = 1.0e300;
    for (int j = 0; j < n; ++j) {
        dst[j * stride] = wts[j] * run;
        const double t = __builtin_fma(x, scale, -snodes[j]);
        amin = __builtin_fmin(amin, __builtin_fabs(t));
        run *= t;
    }
    run = 1.0;
    double su = 0.0;
    for (int j = n - 1; j >= 0; --j) {
        const double c = dst[j * stride] * run;
        dst[j * stride] = c;
        su += c;
        run *= __builtin_fma(x, scale, -snodes[j]);
    }
    const double r = 1.0 / su;
    for (int j = 0; j < n; ++j) dst[j * stride] *= r;
    if (amin < 1e-14 * scale) {                     // the reference's node rule: see bary_weights_reg
        bool found = false;
        for (int j = 0; j < n; ++j) {
            const bool hit = !found && __builtin_fabs(__builtin_fma(x, scale, -snodes[j])) < 1e-14 * scale;
            dst[j * stride] = hit ? 1.0 : 0.0;
            found = found || hit;
        }
    }
}

template <int LEVEL, int DOUT, int NLP>
__device__ __forceinline__ double bary_small_nest(const BaryDims &dims, pcx_cptr Tb,
                                                  const double *bw_lane, const double (&bl)[NLP]) {
    if constexpr (LEVEL == DOUT) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NLP; ++i) s = __builtin_fma(Tb[i], bl[i], s);
        return s;
    } else {
        long stride = 1;
        for (int q = LEVEL + 1; q <= DOUT; ++q) stride *= dims.n[q];
        const double *wk = bw_lane + (size_t)dims.off[LEVEL] * 64;
        double s = 0.0;
        for (int i = 0; i < dims.n[LEVEL]; ++i)
            s = __builtin_fma(wk[(size_t)i * 64], bary_small_nest<LEVEL + 1, DOUT, NLP>(dims, Tb + i * stride, bw_lane, bl), s);
        return s;
    }
}

// (BarySmallScale: pcx_common.h)

// The body shared by the single-tensor kernel and the all-pieces-of-a-spline kernel: weights of the lane's
// point (row `row` of pts; an invalid lane computes on a node and stores nothing), then every tensor.
template <int DOUT, int NLP>
__device__ __forceinline__ void bary_small_body(const BaryDims &dims, const BarySmallScale &sc, pcx_cptr csn, pcx_cptr cnd,
                                                pcx_cptr cw, const double *T, const double *const *T_tab, int m,
                                                const double *__restrict__ pts, double *__restrict__ out, bool valid,
                                                long row, long ostride, long ooff, double *bw_lane) {
    const int d = DOUT + 1;
#pragma unroll
    for (int k = 0; k < DOUT; ++k) {
        const double x = valid ? pts[row * d + k] : cnd[dims.off[k]];
        bary_weights_prod(x, sc.s[k], csn + dims.off[k], cw + dims.off[k], dims.n[k],
                          bw_lane + (size_t)dims.off[k] * 64, 64);
    }
    double bl[NLP];
    {
        const double x = valid ? pts[row * d + DOUT] : cnd[dims.off[DOUT]];
        if (dims.n[DOUT] == NLP)     // wave-uniform
            bary_weights_reg<NLP, true>(x, sc.s[DOUT], csn + dims.off[DOUT], cw + dims.off[DOUT], NLP, bl);
        else
            bary_weights_reg<NLP, false>(x, sc.s[DOUT], csn + dims.off[DOUT], cw + dims.off[DOUT], dims.n[DOUT], bl);
    }
    // the table is wave-private (one wave per workgroup): no barrier
    if (T_tab == nullptr) {
        const double y = bary_small_nest<0, DOUT, NLP>(dims, pcx_as_constant(T), bw_lane, bl);
        if (valid) out[row * ostride + ooff] = y;
    } else {
        for (int z = 0; z < m; ++z) {
            const double y = bary_small_nest<0, DOUT, NLP>(dims, pcx_as_constant(T_tab[z]), bw_lane, bl);
            if (valid) out[row * ostride + ooff + z] = y;
        }
    }
}

template <int DOUT, int NLP>
__global__ void __launch_bounds__(64)
k_bary_small(BaryDims dims, BarySmallScale sc, const double *__restrict__ snodes, const double *__restrict__ nodes,
             const double *__restrict__ wts, const double *__restrict__ T, const double *const *__restrict__ T_tab,
             int m, const double *__restrict__ pts, double *__restrict__ out, long N, long ostride, long ooff,
             const int *__restrict__ perm) {
    // m derivative specs in one launch (T_tab: device table of m tensors; NULL: the single tensor T):
    // the weights of a point are formed once and every tensor is contracted with them.
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long pidx = (long)blockIdx.x * 64 + lane;
    const bool valid = pidx < N;
    const long row = valid ? (perm ? (long)perm[pidx] : pidx) : 0;
    bary_small_body<DOUT, NLP>(dims, sc, pcx_as_constant(snodes), pcx_as_constant(nodes), pcx_as_constant(wts), T, T_tab, m,
                               pts, out, valid, row, ostride, ooff, lds + lane);
}

// ---------------------------------------------------------------------------------
// K1+K2, lane-per-point form for MID-SIZE tensors whose last two dimensions have the same node count NL
// (n^3, n^4, a x n x n ...; round 3).  k_bary_small keeps every outer dimension's weights in a per-wave LDS
// table (sum of outer n x 512 B: 21 KB for 21^3, i.e. < 2 waves per SIMD) and sums a row in ONE dependent
// FMA chain, so neither other waves nor independent instructions hide the scalar-load latency of the tensor
// (0.3 .. 0.4 of the FP64 peak on 16^3 .. 40^3; the MFMA kernel's short plans are no better there: 5 .. 10
// k-steps per row tile against the tile's head-weight look-ups).  Here, as in k_tt_eval_lpp:
//   * the weights of BOTH trailing dimensions live in registers (b1[NL], b2[NL]); only the leading dimensions'
//     weights go through LDS (one read per NL x NL block) -> 10 KB per wave for 21^3, 3 .. 4 waves per SIMD;
//   * an NL x NL block is straight-line code: rows four at a time (four independent chains), every tensor
//     element a scalar operand of one v_fma_f64 (NL^2 + NL FMAs per block: the reference's own nesting,
//     barycentric.py:1036-1046, and its FMA count);
//   * leading dimensions (at most two: d <= 4) are run-time loops around the block.
// 64 threads per workgroup; dynamic LDS = (sum of leading n) * 64 * 8 bytes.
// ---------------------------------------------------------------------------------
template <int NL>
__device__ __forceinline__ double bary_sq_block(pcx_cptr Tb, const double (&b1)[NL], const double (&b2)[NL]) {
    // rows per pass: four independent chains, or three / five when that leaves no ragged last pass (a single row is
    // one dependent chain of NL FMAs: 21^3 ran at 0.375 of the peak with 5 x 4 + 1 rows, against 0.46 for 20^3)
    constexpr int R4 = (NL % 4 == 0) ? 4 : (NL % 3 == 0) ? 3 : (NL % 5 == 0) ? 5 : (NL % 4 == 1 && NL > 5) ? 3 : 4;
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int i = 0; i < NL; i += R4) {
        double s[R4];
#pragma unroll
        for (int r = 0; r < R4; ++r)
            if (i + r < NL) s[r] = Tb[(i + r) * NL] * b2[0];
#pragma unroll
        for (int k = 1; k < NL; ++k)
#pragma unroll
            for (int r = 0; r < R4; ++r)
                if (i + r < NL) s[r] = __builtin_fma(Tb[(i + r) * NL + k], b2[k], s[r]);
#pragma unroll
        for (int r = 0; r < R4; ++r)
            if (i + r < NL) {
                if (r & 1) t1 = __builtin_fma(b1[i + r], s[r], t1);
                else t0 = __builtin_fma(b1[i + r], s[r], t0);
            }
        // (a scheduling barrier here -- one pass at a time -- removes the v_writelane / v_readlane pairs hipcc parks hoisted
        // scalar loads in, and is worth 5..8 % in tools/bary_sq_lab.hip's stripped-down kernel; in THIS kernel it costs up
        // to 38 % (32^3 0.48 -> 0.30, 24^3 0.54 -> 0.40; 13^3 +9 %): the hoisted loads are what hides the scalar-load
        // latency.  Measured, not kept.)
    }
    return t0 + t1;
}

template <int NL, int LEAD>
__device__ __forceinline__ double bary_sq_tensor(const BaryDims &dims, pcx_cptr T, const double *bw_lane,
                                                 const double (&b1)[NL], const double (&b2)[NL]) {
    if constexpr (LEAD == 0) return bary_sq_block<NL>(T, b1, b2);
    else if constexpr (LEAD == 1) {
        double y = 0.0;
        for (int i0 = 0; i0 < dims.n[0]; ++i0, T += NL * NL)
            y = __builtin_fma(bw_lane[(size_t)i0 * 64], bary_sq_block<NL>(T, b1, b2), y);
        return y;
    } else {
        double y = 0.0;
        const double *w1 = bw_lane + (size_t)dims.off[1] * 64;
        for (int i0 = 0; i0 < dims.n[0]; ++i0) {
            double a = 0.0;
            for (int i1 = 0; i1 < dims.n[1]; ++i1, T += NL * NL)
                a = __builtin_fma(w1[(size_t)i1 * 64], bary_sq_block<NL>(T, b1, b2), a);
            y = __builtin_fma(bw_lane[(size_t)i0 * 64], a, y);
        }
        return y;
    }
}

// The body shared by the single-model kernel and the all-pieces-of-a-spline kernel.
template <int NL, int LEAD>
__device__ __forceinline__ void bary_sq_body(const BaryDims &dims, const BarySmallScale &sc, pcx_cptr csn, pcx_cptr cnd, pcx_cptr cw,
                                             const double *T, const double *const *T_tab, int m, const double *__restrict__ pts,
                                             double *__restrict__ out, bool valid, long row, long ostride, long ooff,
                                             double *bw_lane) {
    constexpr int D = LEAD + 2;
#pragma unroll
    for (int k = 0; k < LEAD; ++k) {
        const double x = valid ? pts[row * D + k] : cnd[dims.off[k]];
        bary_weights_prod(x, sc.s[k], csn + dims.off[k], cw + dims.off[k], dims.n[k], bw_lane + (size_t)dims.off[k] * 64, 64);
    }
    double b1[NL], b2[NL];
    {
        const double x1 = valid ? pts[row * D + LEAD] : cnd[dims.off[LEAD]];
        const double x2 = valid ? pts[row * D + LEAD + 1] : cnd[dims.off[LEAD + 1]];
        bary_weights_reg<NL, true>(x1, sc.s[LEAD], csn + dims.off[LEAD], cw + dims.off[LEAD], NL, b1);
        bary_weights_reg<NL, true>(x2, sc.s[LEAD + 1], csn + dims.off[LEAD + 1], cw + dims.off[LEAD + 1], NL, b2);
    }
    // the table is wave-private (one wave per workgroup): no barrier
    for (int z = 0; z < m; ++z) {
        const double y = bary_sq_tensor<NL, LEAD>(dims, pcx_as_constant(T_tab ? T_tab[z] : T), bw_lane, b1, b2);
        if (valid) out[row * ostride + ooff + z] = y;
    }
}

template <int NL, int LEAD>
__global__ void __launch_bounds__(64)
k_bary_sq(BaryDims dims, BarySmallScale sc, const double *__restrict__ snodes, const double *__restrict__ nodes,
          const double *__restrict__ wts, const double *__restrict__ T, const double *const *__restrict__ T_tab, int m,
          const double *__restrict__ pts, double *__restrict__ out, long N, long ostride, long ooff,
          const int *__restrict__ perm) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long pidx = (long)blockIdx.x * 64 + lane;
    const bool valid = pidx < N;
    const long row = valid ? (perm ? (long)perm[pidx] : pidx) : 0;
    bary_sq_body<NL, LEAD>(dims, sc, pcx_as_constant(snodes), pcx_as_constant(nodes), pcx_as_constant(wts), T, T_tab, m, pts, out,
                           valid, row, ostride, ooff, lds + lane);
}

// All pieces of a piecewise interpolant in ONE launch (pieces of equal shape on the lane-per-point kernel):
// a launch per piece is launch-bound once a spline has tens of pieces (64 pieces x 15 k points: 0.46 ms even
// fanned over four streams).  Workgroup b serves 64 consecutive slots of piece blk_piece[b]'s bucket, starting
// at slot blk_first[b] of the bucket permutation; the piece's model (scaled nodes, weights, tensor or tensor
// table, scale) comes from a device table indexed by the piece -- wave-uniform, i.e. scalar loads.
// (SplinePieceModel: pcx_common.h)

template <int DOUT, int NLP>
__global__ void __launch_bounds__(64)
k_bary_small_pieces(BaryDims dims, const SplinePieceModel *__restrict__ models, const int *__restrict__ blk_piece,
                    const int *__restrict__ blk_first, const int *__restrict__ piece_end, int m,
                    const double *__restrict__ pts, double *__restrict__ out, long ostride, long ooff,
                    const int *__restrict__ perm) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int piece = blk_piece[blockIdx.x];
    const long pidx = (long)blk_first[blockIdx.x] + lane;
    const bool valid = pidx < (long)piece_end[piece];
    const long row = valid ? (long)perm[pidx] : 0;
    typedef const SplinePieceModel __attribute__((address_space(4))) *model_cptr;
    const model_cptr mp = (model_cptr)(unsigned long long)(models + piece);
    BarySmallScale sc;
#pragma unroll
    for (int k = 0; k < 4; ++k) sc.s[k] = mp->sc.s[k];
    bary_small_body<DOUT, NLP>(dims, sc, pcx_as_constant(mp->snodes), pcx_as_constant(mp->nodes), pcx_as_constant(mp->wts),
                               mp->T, mp->T_tab, m, pts, out, valid, row, ostride, ooff, lds + lane);
}

// The same for pieces whose last two dimensions share a node count (k_bary_sq's body).
template <int NL, int LEAD>
__global__ void __launch_bounds__(64)
k_bary_sq_pieces(BaryDims dims, const SplinePieceModel *__restrict__ models, const int *__restrict__ blk_piece,
                 const int *__restrict__ blk_first, const int *__restrict__ piece_end, int m,
                 const double *__restrict__ pts, double *__restrict__ out, long ostride, long ooff,
                 const int *__restrict__ perm) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const int piece = blk_piece[blockIdx.x];
    const long pidx = (long)blk_first[blockIdx.x] + lane;
    const bool valid = pidx < (long)piece_end[piece];
    const long row = valid ? (long)perm[pidx] : 0;
    typedef const SplinePieceModel __attribute__((address_space(4))) *model_cptr;
    const model_cptr mp = (model_cptr)(unsigned long long)(models + piece);
    BarySmallScale sc;
#pragma unroll
    for (int k = 0; k < 4; ++k) sc.s[k] = mp->sc.s[k];
    bary_sq_body<NL, LEAD>(dims, sc, pcx_as_constant(mp->snodes), pcx_as_constant(mp->nodes), pcx_as_constant(mp->wts), mp->T,
                           mp->T_tab, m, pts, out, valid, row, ostride, ooff, lds + lane);
}
