// ttcross_kernels.h -- dense steps of the TT-Cross build on the device (gfx950).
//
// Replaces, inside _tt_cross (reference tensor_train.py:123-540), the NumPy/LAPACK
// calls on the small cross matrices:
//     np.linalg.svd(C)            -> one-sided (Hestenes) Jacobi SVD           (:336, :453)
//     _maxvol(U)                  -> pivoted orthogonalisation + greedy swaps  (:38-120)
//     U @ np.linalg.inv(U[piv])   -> Gauss-Jordan inverse + product            (:359, :471)
// and _value_core_to_coeff_core (:997-1016).
//
// The matrices are tiny (rows <= n*r, cols <= r): each step is ONE wavefront (64
// lanes) working out of global memory (L1/L2-resident) with wave shuffles for the
// reductions and a 64x64 LDS tile for the r x r inverse.  This is latency-bound by
// design; the build's wall time is the Python callback, exactly as in the reference.
//
// Only U's column SPACE matters downstream: pivots and C_hat are invariant under
// U -> U G for orthogonal G, so Jacobi vs LAPACK SVD sign/rotation
// differences do not change the result.
#pragma once

#include "pcx_common.h"

#define TTX_THREADS 64
#define TTX_MAX_R 64
#define TTX_MAX_M 8192

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// (value, index) arg-max with ties resolved to the smallest index (np.argmax semantics).
__device__ __forceinline__ void wave_argmax(double &v, long &i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_xor(v, o, 64);
        long oi = __shfl_xor((long long)i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// In-place inverse of the r x r matrix held in LDS (row stride TTX_MAX_R), Gauss-Jordan
// with partial pivoting; lane i owns row i.  Returns false when a pivot is exactly 0
// (the reference's LinAlgError branch).
__device__ bool lds_invert(double *a, int r, int *perm) {
    const int lane = threadIdx.x;
    for (int k = 0; k < r; ++k) {
        double v = (lane >= k && lane < r) ? __builtin_fabs(a[lane * TTX_MAX_R + k]) : -1.0;
        long p = lane;
        wave_argmax(v, p);
        if (!(v > 0.0)) return false;
        if (lane == 0) perm[k] = (int)p;
        __syncthreads();
        if (p != k && lane < r) {  // swap rows k and p: lane = column
            double t = a[k * TTX_MAX_R + lane];
            a[k * TTX_MAX_R + lane] = a[p * TTX_MAX_R + lane];
            a[p * TTX_MAX_R + lane] = t;
        }
        __syncthreads();
        double piv = a[k * TTX_MAX_R + k];
        __syncthreads();
        if (lane < r) {  // scale pivot row: lane = column
            double x = (lane == k) ? 1.0 : a[k * TTX_MAX_R + lane];
            a[k * TTX_MAX_R + lane] = x / piv;
        }
        __syncthreads();
        if (lane < r && lane != k) {  // eliminate: lane = row
            double f = a[lane * TTX_MAX_R + k];
            a[lane * TTX_MAX_R + k] = 0.0;
            for (int j = 0; j < r; ++j)
                a[lane * TTX_MAX_R + j] = __builtin_fma(-f, a[k * TTX_MAX_R + j], a[lane * TTX_MAX_R + j]);
        }
        __syncthreads();
    }
    for (int k = r - 1; k >= 0; --k) {  // undo the row swaps as column swaps
        int p = perm[k];
        if (p != k && lane < r) {
            double t = a[lane * TTX_MAX_R + k];
            a[lane * TTX_MAX_R + k] = a[lane * TTX_MAX_R + p];
            a[lane * TTX_MAX_R + p] = t;
        }
        __syncthreads();
    }
    return true;
}

// B (m x r, row-major, ldb) = A (m x r, lda) . Ainv (LDS r x r)
__device__ void mat_times_lds(const double *A, int lda, int m, int r, const double *ainv, double *B,
                              int ldb) {
    for (int i = threadIdx.x; i < m; i += TTX_THREADS)
        for (int j = 0; j < r; ++j) {
            double s = 0.0;
            for (int k = 0; k < r; ++k) s = __builtin_fma(A[(long)i * lda + k], ainv[k * TTX_MAX_R + j], s);
            B[(long)i * ldb + j] = s;
        }
}

// _maxvol (tensor_train.py:38-120) for m > r.  A: m x r (lda).  work: m x r scratch
// (residuals, then B).  idx: r outputs.  `inv` = 64x64 LDS tile, `q`/`perm` LDS.
__device__ void maxvol_device(const double *A, int lda, int m, int r, double tol, int max_iters,
                              double *work, long long *idx, double *inv, double *q, int *perm) {
    const int lane = threadIdx.x;
    // phase 1: the r pivots of a column-pivoted QR of A^T = greedy choice of the row
    // with the largest residual after projecting out the rows already chosen.
    for (int i = lane; i < m; i += TTX_THREADS)
        for (int j = 0; j < r; ++j) work[(long)i * r + j] = A[(long)i * lda + j];
    __syncthreads();
    for (int k = 0; k < r; ++k) {
        double best = -1.0;
        long bi = 0x7fffffff;
        for (int i = lane; i < m; i += TTX_THREADS) {
            double s = 0.0;
            for (int j = 0; j < r; ++j) { double x = work[(long)i * r + j]; s = __builtin_fma(x, x, s); }
            if (s > best) { best = s; bi = i; }
        }
        wave_argmax(best, bi);
        if (lane == 0) idx[k] = bi;
        double nrm = __builtin_sqrt(best);
        if (lane < r) q[lane] = (nrm > 0.0) ? work[bi * r + lane] / nrm : 0.0;
        __syncthreads();
        for (int i = lane; i < m; i += TTX_THREADS) {
            double dot = 0.0;
            for (int j = 0; j < r; ++j) dot = __builtin_fma(work[(long)i * r + j], q[j], dot);
            for (int j = 0; j < r; ++j) work[(long)i * r + j] = __builtin_fma(-dot, q[j], work[(long)i * r + j]);
            if (i == bi)
                for (int j = 0; j < r; ++j) work[(long)i * r + j] = 0.0;  // chosen: never again
        }
        __syncthreads();
    }
    // phase 2: B = A inv(A[idx]); greedy swaps on the largest |B| entry
    if (lane < r)
        for (int j = 0; j < r; ++j) inv[lane * TTX_MAX_R + j] = A[idx[lane] * lda + j];
    __syncthreads();
    if (!lds_invert(inv, r, perm)) return;  // LinAlgError branch: keep the QR pivots
    mat_times_lds(A, lda, m, r, inv, work, r);
    __syncthreads();
    for (int it = 0; it < max_iters; ++it) {
        double best = -1.0;
        long bf = 0x7fffffffffffL;
        for (long f = lane; f < (long)m * r; f += TTX_THREADS) {
            double a = __builtin_fabs(work[f]);
            if (a > best) { best = a; bf = f; }
        }
        wave_argmax(best, bf);
        if (!(best > tol)) break;
        int i = (int)(bf / r), j = (int)(bf % r);
        if (lane == 0) idx[j] = i;
        double bij = work[(long)i * r + j];
        if (lane < r) q[lane] = work[(long)i * r + lane];  // row_i copy
        __syncthreads();
        for (int row = lane; row < m; row += TTX_THREADS) {
            double cj = work[(long)row * r + j];
            for (int col = 0; col < r; ++col)
                work[(long)row * r + col] -= cj * q[col] / bij;
            work[(long)row * r + j] = cj / bij;
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(TTX_THREADS)
k_maxvol(const double *__restrict__ A, int m, int r, double tol, int max_iters, double *work,
         long long *idx) {
    __shared__ double inv[TTX_MAX_R * TTX_MAX_R];
    __shared__ double q[TTX_MAX_R];
    __shared__ int perm[TTX_MAX_R];
    maxvol_device(A, r, m, r, tol, max_iters, work, idx, inv, q, perm);
}

// One unfolding step of _tt_cross (tensor_train.py:332-362 / :449-474).
//   C: m x c row-major (left intact).  W: m x c scratch (Jacobi iterate, then U in its first
//   `rank` columns, leading dimension c).  work: m x c scratch.  chat: m x rank output.
__global__ void __launch_bounds__(TTX_THREADS)
k_cross_step(const double *__restrict__ C, int m, int c, int cap, double rel_thresh, double *W,
             double *work, double *chat, long long *pivots, int *rank_out) {
    __shared__ double inv[TTX_MAX_R * TTX_MAX_R];
    __shared__ double q[TTX_MAX_R];
    __shared__ int perm[TTX_MAX_R];
    __shared__ int order[TTX_MAX_R];
    const int lane = threadIdx.x;

    for (long f = lane; f < (long)m * c; f += TTX_THREADS) W[f] = C[f];
    __syncthreads();

    // ---- one-sided Jacobi: rotate column pairs until mutually orthogonal
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < c - 1; ++p)
            for (int qq = p + 1; qq < c; ++qq) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = lane; i < m; i += TTX_THREADS) {
                    double wp = W[(long)i * c + p], wq = W[(long)i * c + qq];
                    al = __builtin_fma(wp, wp, al);
                    be = __builtin_fma(wq, wq, be);
                    ga = __builtin_fma(wp, wq, ga);
                }
                al = wave_sum(al); be = wave_sum(be); ga = wave_sum(ga);
                if (al == 0.0 || be == 0.0) continue;
                if (__builtin_fabs(ga) <= 1e-15 * __builtin_sqrt(al * be)) continue;
                rotated = 1;
                double zeta = (be - al) / (2.0 * ga);
                double t = ((zeta >= 0.0) ? 1.0 : -1.0) / (__builtin_fabs(zeta) + __builtin_sqrt(1.0 + zeta * zeta));
                double cs = 1.0 / __builtin_sqrt(1.0 + t * t);
                double sn = cs * t;
                for (int i = lane; i < m; i += TTX_THREADS) {
                    double wp = W[(long)i * c + p], wq = W[(long)i * c + qq];
                    W[(long)i * c + p] = cs * wp - sn * wq;
                    W[(long)i * c + qq] = sn * wp + cs * wq;
                }
                __syncthreads();
            }
        if (!rotated) break;
    }

    // ---- singular values = column norms; order descending (stable)
    double sj = -1.0;
    if (lane < c) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) { double x = W[(long)i * c + lane]; s = __builtin_fma(x, x, s); }
        sj = __builtin_sqrt(s);
    }
    int pos = 0;
    double s0 = 0.0;
    for (int k = 0; k < c; ++k) {
        double sk = __shfl(sj, k, 64);
        if (sk > sj || (sk == sj && k < lane)) ++pos;
        s0 = (sk > s0) ? sk : s0;
    }
    if (lane < c) { order[pos] = lane; q[pos] = sj; }
    // tensor_train.py:337-341: effective = #{S > 1e-12 S0} (1 if S0 == 0)
    int keep = (lane < c && s0 > 0.0 && sj > rel_thresh * s0) ? 1 : 0;
    int effective = 0;
    for (int k = 0; k < c; ++k) effective += __shfl(keep, k, 64);
    if (!(s0 > 0.0)) effective = 1;
    int ucols = (m < c) ? m : c;
    int rank = effective;
    if (rank > cap) rank = cap;
    if (rank > ucols) rank = ucols;
    if (rank < 1) rank = 1;
    __syncthreads();

    // ---- U = first `rank` sorted columns, normalised; stored in `work` then copied to W
    for (int i = lane; i < m; i += TTX_THREADS)
        for (int k = 0; k < rank; ++k) {
            double nrm = q[k];
            work[(long)i * c + k] = (nrm > 0.0) ? W[(long)i * c + order[k]] / nrm : 0.0;
        }
    __syncthreads();
    for (int i = lane; i < m; i += TTX_THREADS)
        for (int k = 0; k < rank; ++k) W[(long)i * c + k] = work[(long)i * c + k];
    __syncthreads();

    // ---- pivots
    if (m > rank) {
        maxvol_device(W, c, m, rank, 1.05, 100, work, pivots, inv, q, perm);
    } else if (lane < rank) {
        pivots[lane] = lane;
    }
    __syncthreads();

    // ---- C_hat = U inv(U[pivots])  (fallback C_hat = U when singular)
    if (lane < rank)
        for (int j = 0; j < rank; ++j) inv[lane * TTX_MAX_R + j] = W[pivots[lane] * c + j];
    __syncthreads();
    if (lds_invert(inv, rank, perm)) {
        mat_times_lds(W, c, m, rank, inv, chat, rank);
    } else {
        for (int i = lane; i < m; i += TTX_THREADS)
            for (int k = 0; k < rank; ++k) chat[(long)i * rank + k] = W[(long)i * c + k];
    }
    if (lane == 0) *rank_out = rank;
}

// _value_core_to_coeff_core (tensor_train.py:997-1016): reverse the node axis, DCT-II
// (SciPy backward norm: y_k = 2 sum_j x_j cos(pi k (2j+1) / (2n))), divide by n, halve k=0.
__global__ void k_value_to_coeff_core(const double *__restrict__ v, double *__restrict__ out, int rl,
                                      int n, int rr) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)rl * n * rr;
    if (idx >= total) return;
    int cc = (int)(idx % rr);
    int k = (int)((idx / rr) % n);
    int i = (int)(idx / ((long)rr * n));
    const double pi = 3.14159265358979323846;
    double s = 0.0;
    for (int j = 0; j < n; ++j) {
        double x = v[((long)i * n + (n - 1 - j)) * rr + cc];
        s = __builtin_fma(x, cos(pi * (double)k * (double)(2 * j + 1) / (double)(2 * n)), s);
    }
    s = 2.0 * s / (double)n;
    if (k == 0) s *= 0.5;
    out[idx] = s;
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
