// bary_kfold_kernels.h -- k_bary_mfma_kfold: the MFMA form of the barycentric contraction for 3-D tensors ONE of whose
// dimensions fills whole row tiles (26 ... 32, 44 ... 48, 59 ... 64 nodes; 13 ... 16 beside a long last dimension) -- round 4.
//
// ("Dimension 0 / 1 / 2" below are ROLES: the planner gives the rows to whichever tensor dimension fills its row tiles best
// -- 20 x 16 x 64 runs with its last dimension as rows --, BaryKfoldPlan::dim / stride; cubes keep the natural order.)
// Same contraction as k_bary_mfma / k_bary_mfma_grid (reference barycentric.py:1035-1046), folded the other way round:
//     y(p) = sum_{i0} b0[i0,p] ( sum_{(i1,i2)} T[i0][i1][i2] b1[i1,p] b2[i2,p] )
// rows M = n0 (1 ... 4 row tiles that stay in the accumulators for the WHOLE contraction), K = n1 n2 (hundreds to
// thousands: far too many B operands for registers).  The B operand of k-step (i1, s2) is formed where it is used:
// b2's KS2 = ceil(n2 / 4) operands sit in registers, b1[i1] is one LDS read per i1, B = b1[i1] * b2[s2] is one multiply
// per k-step and column tile -- against MT matrix instructions it feeds.  There is NO per-tile epilogue at all: the row
// weights b0 meet the accumulators once, at the end.  A grid-plan tile of 5-8 k-steps (bary_grid_kernels.h) carries four
// FMAs, four table reads and 17 wait states per 16 matrix instructions; here a k-step carries half a multiply per
// matrix instruction.  Padding: n0 to 16 MT rows, n2 to 4 KS2 columns per i1 (30^3: 0.94 x 0.94, as the grid plan).
// The per-wave LDS table holds ONE dimension's weights at a time (b2 -> registers, then b1 for the loop, then b0 for the
// epilogue): max(16 MT, n1 + 1, 4 KS2) rows of PW doubles.  Forming the weights is the kernel's only vector work of any size
// (30^3, first version: 1,550 of a wave's 2,200 non-matrix vector instructions, all of them 12 % of the cycles of the pipe the
// matrix instructions share): two lanes share a point and take half the nodes each, and the weights stay unnormalised in the
// table -- the factors 1 / sum go into the register operands.
// Fragment image: frag[(i1 KS2 + s2) MT + t][lane] = T[16 t + (l & 15)][i1][4 s2 + (l >> 4)] (zero outside).
// STR ("straddle", n2 = 4 KS2 - 2: 26, 30): K runs over PAIRS of i1 without padding -- 2 n2 elements in P = 2 KS2 - 1
// k-steps instead of 2 KS2, the middle k-step holding the last two nodes of the first index (lane groups 0, 1) and the first
// two of the second (lane groups 2, 3): its b1 factor is a per-lane select, and a lane's P b2 registers hold node
// (4 s' + g) mod n2.  30^3: 900 matrix instructions per 32 points instead of 960.
// Small batches (SM, below): a workgroup of MT waves owns one column tile and every wave one of its row tiles -- the accumulators
// of different row tiles never meet before the epilogue, whose FMA chain the waves then run one after the other, so the
// arithmetic and its order are the unsplit kernel's and a point's value does not depend on its batch.  Dimension 1 is NOT split:
// a version that finished it in four chunks (b0 resident beside b1, chunk sums added in a fixed order, small batches split over
// blockIdx.y) was built and measured -- the larger table costs a workgroup per CU and 5-7 % of the throughput on every shape
// (30^3 0.654 -> 0.616, 64^3 0.90 -> 0.85) -- not kept.  Four column tiles per wave (a lane per point, each fragment feeding four
// matrix instructions; 204 VGPRs, two workgroups per CU) measured 2 % behind two on 26^3 ... 32^3 -- not kept either.
#pragma once

#include "pcx_common.h"

__global__ void k_pack_fragments_kfold(const double *__restrict__ T, double *__restrict__ frag, BaryKfoldPlan kp) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nfrag = (long)kp.nbody * kp.P * kp.MT;
    if (idx >= (nfrag + PCX_KFOLD_PAD) * 64) return;
    const int l = (int)(idx & 63);
    long f = idx >> 6;
    if (f >= nfrag) { frag[idx] = 0.0; return; }      // the pad the prefetch ring reads past the end
    const int t = (int)(f % kp.MT); f /= kp.MT;
    const int sp = (int)(f % kp.P);                    // k-step within the loop body
    const int body = (int)(f / kp.P);
    const int i0 = 16 * t + (l & 15);
    int i1 = body, i2 = 4 * sp + (l >> 4);
    if (kp.str) {                                      // a body = two indices of dimension 1, its k-steps run over 2 n2 elements
        i1 = 2 * body;
        if (i2 >= kp.n2) { i2 -= kp.n2; ++i1; }
    }
    frag[idx] = (i0 < kp.n0 && i1 < kp.n1 && i2 < kp.n2) ? T[i0 * kp.stride[0] + i1 * kp.stride[1] + i2 * kp.stride[2]] : 0.0;
}

// UNNORMALISED barycentric weights of one dimension into a point's LDS column, the work split between the two lanes that
// share the point (half 0: nodes [0, mid), half 1: nodes [mid, n); partner lane = lane ^ pw); returns the normalisation
// factor 1 / sum (1.0 when the point sits within the reference's 1e-14 of a node: one-hot weights are written).  The
// caller folds the factor into a register operand instead of running another pass over the column.
// Product form (scaled nodes given): c_j = w_j prod_{i != j} (x - x_i) from a prefix and a suffix product; each half runs
// its own prefix / suffix and starts the suffix pass from the OTHER half's full product.  Else by division.
__device__ __forceinline__ double kfold_weights(double x, const double *__restrict__ nodes, const double *__restrict__ wts,
                                                const double *__restrict__ snodes, double scale, int n, double *col, int stride,
                                                int half, int pw) {
    const int mid = (n + 1) >> 1;
    const int lo = half ? mid : 0, hi = half ? n : mid;
    if (snodes) {
        double run = 1.0, amin = 1.0e300;
#pragma unroll 2
        for (int j = lo; j < hi; ++j) {
            col[j * stride] = wts[j] * run;
            const double t = __builtin_fma(x, scale, -snodes[j]);
            amin = __builtin_fmin(amin, __builtin_fabs(t));
            run *= t;
        }
        run = __shfl_xor(run, pw, 64);
        amin = __builtin_fmin(amin, __shfl_xor(amin, pw, 64));
        double su = 0.0;
#pragma unroll 2
        for (int j = hi - 1; j >= lo; --j) {
            const double cj = col[j * stride] * run;
            col[j * stride] = cj;
            su += cj;
            run *= __builtin_fma(x, scale, -snodes[j]);
        }
        su += __shfl_xor(su, pw, 64);
        if (amin < 1e-14 * scale) {                       // both lanes of the point take this branch together
            int first = -1;
#pragma unroll 1
            for (int j = 0; j < n; ++j)
                if (first < 0 && __builtin_fabs(__builtin_fma(x, scale, -snodes[j])) < 1e-14 * scale) first = j;
#pragma unroll 1
            for (int j = lo; j < hi; ++j) col[j * stride] = (j == first) ? 1.0 : 0.0;
            return 1.0;
        }
        return 1.0 / su;
    }
    int exact = -1;
#pragma unroll 1
    for (int j = 0; j < n; ++j)
        if (exact < 0 && __builtin_fabs(x - nodes[j]) < 1e-14) exact = j;
    if (exact >= 0) {
#pragma unroll 1
        for (int j = lo; j < hi; ++j) col[j * stride] = (j == exact) ? 1.0 : 0.0;
        return 1.0;
    }
    double su = 0.0;
#pragma unroll 1
    for (int j = lo; j < hi; ++j) {
        const double u = wts[j] / (x - nodes[j]);
        col[j * stride] = u;
        su += u;
    }
    su += __shfl_xor(su, pw, 64);
    return 1.0 / su;
}

// prefetch ring: the largest divisor of the FR = KS2 MT fragments of one i1 that is at most 16, so that ring slots are
// compile-time registers inside the unrolled i1 body
__host__ __device__ constexpr int kfold_depth(int fr) {
    int best = 1;
    for (int dd = 1; dd <= 16 && dd <= fr; ++dd)
        if (fr % dd == 0) best = dd;
    return best;
}

// 256 threads = 4 waves walking the fragment image in step (L1 sharing, as k_bary_mfma_grid), PW = 16 NT points per wave;
// dynamic LDS = 4 * kp.trows * PW * 8 bytes.  grid = (point blocks, 1, specs).
// SM ("split M", small batches): a workgroup of MT waves owns ONE column tile and every wave ONE of its row tiles, so a
// single query waits for n1 KS2 matrix instructions instead of n1 KS2 MT; the waves then run the epilogue's FMA chain one after
// the other (wave t continues from wave t - 1's value through LDS): the arithmetic of the unsplit kernel in its order, so
// a point's value still does not depend on its batch.  Host call with one point, unsplit -> split (grid form): 64^3 171 -> 106
// (89) us, 48^3 95 -> 78 (72) us; 4,096 points: 64^3 167 -> 103 (119) us.
template <int MT, int KS2, int NT, bool STR, bool SM = false>
// STR holds 2 KS2 - 1 b2 registers per column tile: n2 = 22, 26 fit three waves per SIMD (168 VGPRs; same box: 26^3 0.565 -> 0.585),
// n2 = 30 would spill 18 registers there (0.667 -> 0.650) and stays at two.
__global__ void __launch_bounds__(SM ? 64 * MT : 256, (STR && MT <= 2 && KS2 <= 7) ? 3 : 2)
k_bary_mfma_kfold(BaryDims dims, BaryKfoldPlan kp, const double *__restrict__ nodes, const double *__restrict__ wts,
                  const double *__restrict__ snodes, const double *const *__restrict__ frag_tab,
                  const double *__restrict__ pts, double *__restrict__ out, long N, long ostride, long ooff,
                  const int *__restrict__ perm) {
    constexpr int PW = 16 * NT;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    static_assert(!SM || (NT == 1 && MT >= 2), "split-M launches: one column tile per workgroup, one row tile per wave");
    constexpr int MTL = SM ? 1 : MT;                    // row tiles this wave accumulates
    const int t0 = SM ? wave : 0;                       // ... starting at this one
    double *bw = lds + (size_t)wave * kp.trows * PW;     // (SM: every wave forms its own copy of the table)
    const long base = SM ? (long)blockIdx.x * PW : ((long)blockIdx.x * 4 + wave) * PW;
    typedef const double __attribute__((address_space(1))) *gptr_t;
    const gptr_t tf = (gptr_t)frag_tab[blockIdx.z] + lane;
    // two lanes form the weights of one point, half the nodes each (NT = 1: lanes 32 .. 63 idle through it); one dimension
    // at a time, unnormalised -- the factors 1 / sum go into the register operands (r2, r1 into B2; r0 into the result)
    const bool former = lane < 2 * PW;
    const int fpoint = lane & (PW - 1), fhalf = (lane / PW) & 1;
    const long pidx0 = base + fpoint;
    const bool valid = pidx0 < N;
    const long prow = valid ? (perm ? (long)perm[pidx0] : pidx0) : 0;
    // weights of the dimension in `role` into table rows [row, row + n), zeros up to row `upto` (rows a padded index reads)
    auto weights_of = [&](int role, int row, int upto) -> double {
        double r = 1.0;
        if (former) {
            const int k = kp.dim[role];                     // the tensor dimension that plays this role
            const double *nd = nodes + dims.off[k];
            const double x = valid ? pts[prow * 3 + k] : nd[0];
            double *col = bw + (size_t)row * PW + fpoint;
            r = kfold_weights(x, nd, wts + dims.off[k], snodes ? snodes + dims.off[k] : nullptr, snodes ? snodes[dims.sum_n + k] : 1.0,
                              dims.n[k], col, PW, fhalf, PW);
            for (int rr = dims.n[k] + fhalf; rr < upto - row; rr += 2) col[(size_t)rr * PW] = 0.0;
        }
        __syncthreads();
        return r;
    };

    // ---- b2 -> registers (the B layout: lane group g holds node 4 s2 + g of column c), normalised on the way ----
    constexpr int P = STR ? 2 * KS2 - 1 : KS2;          // k-steps per loop body (STR: a body is two indices of dimension 1)
    const double r2 = weights_of(2, 0, 4 * KS2);
    double B2[NT][P];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const double rp = __shfl(r2, 16 * nt + c, 64);
#pragma unroll
        for (int s = 0; s < P; ++s) {
            int row = 4 * s + g;
            if (STR && row >= 4 * KS2 - 2) row -= 4 * KS2 - 2;
            B2[nt][s] = bw[(size_t)row * PW + 16 * nt + c] * rp;
        }
    }
    __syncthreads();
    // ---- b1 -> table (unnormalised; its factor goes into B2), main loop ----
    const double r1 = weights_of(1, 0, kp.n1 + 1);       // + a zero row: the second index of an odd n1's last pair
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const double rp = __shfl(r1, 16 * nt + c, 64);
#pragma unroll
        for (int s = 0; s < P; ++s) B2[nt][s] *= rp;
    }
    pcx_d4 acc[MTL][NT];
#pragma unroll
    for (int t = 0; t < MTL; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (pcx_d4){0.0, 0.0, 0.0, 0.0};
    // a ring of DEPTH fragments runs ahead of the multiplication: the register a k-step has read is refilled at once with
    // the fragment DEPTH positions on (the image carries PCX_KFOLD_PAD fragments behind its end for the last refills)
    constexpr int FR = P * MT;                           // fragments per body in the image
    constexpr int FRL = P * MTL;                         // ... of them this wave's (SM: every MT-th)
    constexpr int DEPTH = kfold_depth(FRL);
    static_assert(DEPTH <= PCX_KFOLD_PAD, "ring deeper than the image's pad");
    // position i of this wave's fragment sequence -> index in the image (SM: the fragments of row tile t0 only)
    // (there the look-ahead is clamped to the last body: DEPTH positions of ONE tile's sequence are DEPTH MT fragments of the image)
    const size_t lastpos = (size_t)kp.nbody * P - 1;
    auto frag_at = [&](size_t i) -> size_t { return SM ? (i < lastpos ? i : lastpos) * MT + t0 : i; };
    double ring[DEPTH];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) ring[i] = tf[frag_at(i) * 64];
    constexpr int NW = STR ? 2 : 1;                      // b1 rows per body
    double w1c[NW][NT];
#pragma unroll
    for (int w = 0; w < NW; ++w)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w1c[w][nt] = bw[(size_t)w * PW + 16 * nt + c];
    for (int body = 0; body < kp.nbody; ++body) {
        const int bn = body + 1 < kp.nbody ? body + 1 : body;
        const size_t nxt = (size_t)body * FRL + DEPTH;     // this wave's sequence position DEPTH ahead of the body's first
        double w1n[NW][NT];
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w1n[w][nt] = bw[(size_t)(NW * bn + w) * PW + 16 * nt + c];
#pragma unroll
        for (int s = 0; s < P; ++s) {
            double b[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                double w1 = w1c[0][nt];
                if constexpr (STR) {
                    if (s == KS2 - 1) w1 = g < 2 ? w1c[0][nt] : w1c[1][nt];
                    else if (s >= KS2) w1 = w1c[1][nt];
                }
                b[nt] = B2[nt][s] * w1;
            }
#pragma unroll
            for (int t = 0; t < MTL; ++t) {
                const int q = s * MTL + t;
                const double a = ring[q % DEPTH];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[nt], acc[t][nt], 0, 0, 0);
                ring[q % DEPTH] = tf[frag_at(nxt + q) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int w = 0; w < NW; ++w)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w1c[w][nt] = w1n[w][nt];
    }
    __syncthreads();
    // ---- b0 -> table, epilogue: rows (g + 4 j) of every tile, then the four lane groups, times 1 / sum of dimension 0 ----
    const double r0 = weights_of(0, 0, 16 * MT);
    double v[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) v[nt] = 0.0;
    if constexpr (SM) {
        // the chain v = fma(acc[t][j], b0[16 t + g + 4 j], v) runs over t in order: wave t takes over from wave t - 1
        double *hand = lds + (size_t)MT * kp.trows * PW;           // 64 doubles behind the tables
        for (int t = 0; t < MT; ++t) {
            if (wave == t) {
                if (t > 0) v[0] = hand[lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[0] = __builtin_fma(acc[0][0][j], bw[(size_t)(16 * t + g + 4 * j) * PW + c], v[0]);
                hand[lane] = v[0];
            }
            __syncthreads();
        }
        if (wave != MT - 1) return;
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    v[nt] = __builtin_fma(acc[t][nt][j], bw[(size_t)(16 * t + g + 4 * j) * PW + 16 * nt + c], v[nt]);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const double rp = __shfl(r0, 16 * nt + c, 64);
        double y = v[nt];
        y += __shfl_xor(y, 16, 64);
        y += __shfl_xor(y, 32, 64);
        const long pidx = base + 16 * nt + c;
        if (g == 0 && pidx < N) {
            const long row = perm ? (long)perm[pidx] : pidx;
            out[row * ostride + ooff + blockIdx.z] = y * rp;
        }
    }
}
