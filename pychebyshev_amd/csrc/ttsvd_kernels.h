// ttsvd_kernels.h -- TT-SVD compression of a dense value tensor on the device (gfx950).
//
// Replaces _tt_svd_from_tensor / the decomposition half of _tt_svd (reference
// tensor_train.py:543-690): for k = 0 .. d-2 the current unfolding C (r_{k-1} n_k rows,
// everything else as columns) is factored as C = U (S V^T); the leading `rank` columns of
// U become value core k and the rows S V^T = U^T C become the next unfolding.
//
// The unfoldings are short and very wide (11 x 14641, 121 x 1331, ...), so the SVD is a
// one-sided (Hestenes) Jacobi iteration on the ROWS of C: plane rotations J are applied to
// row pairs until all rows are mutually orthogonal.  Then C = U B with U = the product of
// the rotations and B's rows = sigma_i v_i^T, i.e. exactly the two factors TT-SVD needs --
// no separate V, no S^-1 scaling, and the small singular values keep their relative
// accuracy (a Gram-matrix eigensolver would lose everything below 1e-8 sigma_max).
// One workgroup per row pair; the m/2 disjoint pairs of a round-robin tournament step run
// in one launch, m-1 steps make a sweep.  Everything a sweep touches is L2-resident.
#pragma once

#include "pcx_common.h"

#define TTSVD_THREADS 256

__device__ __forceinline__ double ttsvd_block_sum(double v, double *red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();                       // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// Tournament step `step` (0 .. mp-2, mp = m rounded up to even) of a Jacobi sweep over the
// rows of B (m x N, row stride ldb).  U (m x m, row-major) accumulates the rotations as
// column operations, so that U B stays equal to the input.  rotated: number of pairs that
// were not yet orthogonal to working precision.
// floor2: rows whose squared norm is below it are rounding noise relative to the largest
// singular value ((8 eps ||C||_F)^2): they are left alone, otherwise the iteration spends
// dozens of sweeps orthogonalising noise against noise.
__global__ void __launch_bounds__(TTSVD_THREADS)
k_rowjacobi_step(double *__restrict__ B, long ldb, int m, long N, double *__restrict__ U,
                 int step, int *__restrict__ rotated, double floor2) {
    __shared__ double red[4];
    const int mp = (m + 1) & ~1;
    const int i = blockIdx.x;
    int p, q;
    if (i == 0) { p = mp - 1; q = step % (mp - 1); }
    else { p = (step + i) % (mp - 1); q = (step - i + 2 * (mp - 1)) % (mp - 1); }
    if (p >= m || q >= m) return;          // the bye of an odd field
    if (p > q) { int t = p; p = q; q = t; }
    double *a = B + (long)p * ldb, *b = B + (long)q * ldb;
    double aa = 0.0, bb = 0.0, ab = 0.0;
    for (long j = threadIdx.x; j < N; j += TTSVD_THREADS) {
        const double x = a[j], y = b[j];
        aa = __builtin_fma(x, x, aa);
        bb = __builtin_fma(y, y, bb);
        ab = __builtin_fma(x, y, ab);
    }
    aa = ttsvd_block_sum(aa, red);
    bb = ttsvd_block_sum(bb, red);
    ab = ttsvd_block_sum(ab, red);
    if (!(aa > floor2) || !(bb > floor2)) return;
    if (__builtin_fabs(ab) <= 1e-15 * __builtin_sqrt(aa) * __builtin_sqrt(bb)) return;
    const double zeta = (bb - aa) / (2.0 * ab);
    const double t = __builtin_copysign(1.0, zeta) / (__builtin_fabs(zeta) + __builtin_sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / __builtin_sqrt(1.0 + t * t);
    const double s = c * t;
    for (long j = threadIdx.x; j < N; j += TTSVD_THREADS) {
        const double x = a[j], y = b[j];
        a[j] = c * x - s * y;
        b[j] = s * x + c * y;
    }
    for (int r = threadIdx.x; r < m; r += TTSVD_THREADS) {
        const double x = U[(long)r * m + p], y = U[(long)r * m + q];
        U[(long)r * m + p] = c * x - s * y;
        U[(long)r * m + q] = s * x + c * y;
    }
    if (threadIdx.x == 0) atomicAdd(rotated, 1);
}

__global__ void k_set_identity(double *__restrict__ U, int m) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)m * m) U[idx] = (idx / m == idx % m) ? 1.0 : 0.0;
}

// squared Euclidean norm of every row: one workgroup per row
__global__ void __launch_bounds__(TTSVD_THREADS)
k_row_sqnorms(const double *__restrict__ B, long ldb, long N, double *__restrict__ out) {
    __shared__ double red[4];
    const double *a = B + (long)blockIdx.x * ldb;
    double s = 0.0;
    for (long j = threadIdx.x; j < N; j += TTSVD_THREADS) s = __builtin_fma(a[j], a[j], s);
    s = ttsvd_block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// out[r][:] = B[rows[r]][:]  (the kept rows sigma_i v_i^T, in descending order, packed)
__global__ void k_gather_rows(const double *__restrict__ B, long ldb, long N,
                              const int *__restrict__ rows, double *__restrict__ out) {
    const double *a = B + (long)rows[blockIdx.y] * ldb;
    double *o = out + (long)blockIdx.y * N;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < N; j += (long)gridDim.x * blockDim.x)
        o[j] = a[j];
}
