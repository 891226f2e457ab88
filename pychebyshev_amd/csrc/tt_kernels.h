// tt_kernels.h -- HIP kernels for the tensor-train interpolant (gfx950).
//
// Replaces ChebyshevTT.eval_batch (reference tensor_train.py:2217-2265): for each
// storage dimension k, v <- v . (sum_j T_j(s_k) G_k[:, j, :]).
//
// GEMM form per dimension, points on the MFMA N axis:
//     v'[b, p] = sum_{(j,a)} G_k[a, j, b] * ( v[a, p] * T_j(s_k(p)) )
//   A operand  = core fragment, lane l holds G_k[a = 4c + (l>>4)][j][b = 16t + (l&15)]
//   B operand  = v[a = 4c + (l>>4), p = l&15] * q_j(p)              (one VALU multiply)
//   D (16x16)  = lane l, reg i holds row b = 16t + (l>>4) + 4i, column p = l & 15
// With the k index ordered (j, a) and ranks padded to multiples of 4, the row a a lane
// needs as B operand for chunk c is exactly the row it already holds in D register
// (c & 3) of tile (c >> 2): the chain runs dimension after dimension with NO cross-lane
// movement and no LDS.
#pragma once

#include "pcx_common.h"

// Core packing: frag[dim][j][c][t][l] = G[a = 4c + (l>>4)][j][b = 16t + (l&15)], zero padded;
// c < rc = ceil(r_left/4), t < rt = ceil(r_right/16).
__global__ void k_tt_pack_core(const double *__restrict__ G, double *__restrict__ frag, int rl,
                               int n, int rr, int rc, int rt) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)n * rc * rt * 64;
    if (idx >= total) return;
    int l = (int)(idx & 63);
    long rest = idx >> 6;
    int t = (int)(rest % rt);
    rest /= rt;
    int c = (int)(rest % rc);
    int j = (int)(rest / rc);
    int a = 4 * c + (l >> 4);
    int b = 16 * t + (l & 15);
    frag[idx] = (a < rl && b < rr) ? G[((long)a * n + j) * rr + b] : 0.0;
}

struct TTRanks {
    int rc[PCX_MAX_DIMS];  // left-rank chunks of 4 per storage dim
    int rt[PCX_MAX_DIMS];  // right-rank tiles of 16 per storage dim
};

// One wave owns 16*NT points.  RC/RT are compile-time upper bounds for the register
// arrays; the per-dimension chunk/tile counts are wave-uniform runtime values.
template <int RC, int RT, int NT>
__global__ void __launch_bounds__(256)
k_tt_eval_mfma(TTDims dims, TTRanks rk, const double *__restrict__ frag,
               const double *__restrict__ pts, double *__restrict__ out, long N) {
    static_assert(RC <= 4 * RT, "left chunks must fit the previous D tiles");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int c16 = lane & 15;
    const long base = ((long)blockIdx.x * 4 + wave) * (16 * NT);

    double v[NT][RC];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int c = 0; c < RC; ++c) v[nt][c] = 0.0;
        v[nt][0] = (g == 0) ? 1.0 : 0.0;   // v = e_0 (rank-1 left boundary)
    }

    for (int k = 0; k < dims.d; ++k) {
        const int n = dims.n[k];
        const int rc = rk.rc[k];
        const int rt = rk.rt[k];
        const double lo = dims.lo[k], hi = dims.hi[k];
        double s[NT], tprev[NT], tcur[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            long p = base + 16 * nt + c16;
            double x = (p < N) ? pts[p * dims.d + dims.col[k]] : lo;
            s[nt] = 2.0 * (x - lo) / (hi - lo) - 1.0;   // tensor_train.py:2254
            tprev[nt] = 1.0;
            tcur[nt] = s[nt];
        }
        pcx_d4 acc[NT][RT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[nt][t] = (pcx_d4){0.0, 0.0, 0.0, 0.0};

        const double *fk = frag + dims.frag_off[k] + lane;
        for (int j = 0; j < n; ++j) {
            double q[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (j == 0) q[nt] = 1.0;
                else if (j == 1) q[nt] = s[nt];
                else {
                    double tn = __builtin_fma(2.0 * s[nt], tcur[nt], -tprev[nt]);
                    tprev[nt] = tcur[nt];
                    tcur[nt] = tn;
                    q[nt] = tn;
                }
            }
            const double *fj = fk + (size_t)j * rc * rt * 64;
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                if (c < rc) {
                    double bop[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bop[nt] = v[nt][c] * q[nt];
#pragma unroll
                    for (int t = 0; t < RT; ++t) {
                        if (t < rt) {
                            double a = fj[(size_t)(c * rt + t) * 64];
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
                                acc[nt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                                    a, bop[nt], acc[nt][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int c = 0; c < RC; ++c) v[nt][c] = acc[nt][c >> 2][c & 3];
    }

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        long p = base + 16 * nt + c16;
        if (g == 0 && p < N) out[p] = v[nt][0];
    }
}

// _eval_tt (tensor_train.py:223-228) batched over integer grid index tuples: one thread
// per tuple walks the chain of VALUE cores (cores in their natural (r, n, r') layout).
__global__ void k_tt_grid_eval(int d, const int *__restrict__ n, const int *__restrict__ ranks,
                               const long *__restrict__ coff, const double *__restrict__ cores,
                               const int *__restrict__ idx, int count, double *__restrict__ out,
                               double *__restrict__ work, int rmax) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= count) return;
    double *v = work + (size_t)p * 2 * rmax;
    double *v2 = v + rmax;
    v[0] = 1.0;
    for (int k = 0; k < d; ++k) {
        int rl = ranks[k], rr = ranks[k + 1], nk = n[k];
        int i = idx[(long)p * d + k];
        const double *G = cores + coff[k];
        for (int b = 0; b < rr; ++b) {
            double s = 0.0;
            for (int a = 0; a < rl; ++a) s = __builtin_fma(v[a], G[((long)a * nk + i) * rr + b], s);
            v2[b] = s;
        }
        double *sw = v; v = v2; v2 = sw;
    }
    out[p] = v[0];
}
