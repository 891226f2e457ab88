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
// rot_tol: a pair counts as orthogonal when |a.b| <= rot_tol |a||b|.  The computed a.b of two
// orthogonal rows of length N is itself ~eps sqrt(N) |a||b| of rounding: the host passes
// max(1e-15, 2 eps sqrt(N)) -- a fixed 1e-15 kept rotating that noise for sweep after sweep on
// the wide unfoldings (13 sweeps at N = 1331 instead of 7).
// sig2 = (tol * largest row norm)^2 / m (the largest row norm is a lower bound of sigma_max): a pair whose
// squared norms ADD UP to less than it is skipped.  Every row in a skipped pair is then below sig2, so all
// such rows together (at most m) carry less than (tol sigma_max)^2: whatever they would merge into stays
// below the reference's S > tol S[0] cut (tensor_train.py:673-678) and is dropped -- their mutual
// orthogonality is nobody's business.  (Round 2 skipped a pair as soon as BOTH rows were below
// (tol * largest row norm)^2: two nearly parallel rows just under the cut could then hide a singular value
// just above it.)  Every row that can be kept is still rotated against every other row, so the kept
// subspace is exact; a smooth tensor's unfolding has dozens of rows at 1e-9 .. 1e-15 sigma_max that
// otherwise cost ten sweeps among themselves.
__global__ void __launch_bounds__(TTSVD_THREADS)
k_rowjacobi_step(double *__restrict__ B, long ldb, int m, long N, double *__restrict__ U,
                 int step, int *__restrict__ rotated, double floor2, double rot_tol, double sig2) {
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
    if (aa + bb < sig2) return;              // far below the truncation threshold even merged: both will be dropped
    if (ab * ab <= (rot_tol * rot_tol) * aa * bb) return;          // |a.b| <= rot_tol |a||b| without two square roots
    // rotated[1] counts the pairs that were further than 1e-8 from orthogonal: a sweep without any leaves
    // every pair below ~1e-16 (the iteration converges quadratically), so it is the last one -- no extra
    // sweep just to see zero rotations
    const bool large = ab * ab > 1e-16 * aa * bb;
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
    if (threadIdx.x == 0) {
        atomicAdd(rotated, 1);
        if (large) atomicAdd(rotated + 1, 1);
    }
}

// The whole iteration in ONE workgroup for unfoldings that fit LDS (the later, small ones of a
// TT-SVD: 88 x 121, 88 x 11, ...): the rows live in LDS, a wave owns a row pair at a time, the
// m/2 pairs of a tournament step are dealt over the 16 waves, __syncthreads() separates steps,
// and the sweep loop with its convergence test runs inside the kernel -- one launch instead of
// (m - 1) x sweeps.  U stays in global memory (each step touches disjoint column pairs).
// Same rotations in the same order as k_rowjacobi_step; sums over a row are taken per 16-lane
// group (strided, shuffle tree) instead of per workgroup.
#define TTSVD_LDS_THREADS 1024
__global__ void __launch_bounds__(TTSVD_LDS_THREADS)
k_rowjacobi_lds(double *__restrict__ Bg, int m, int N, double *__restrict__ Ug, double floor2, double rot_tol,
                double sig2, int max_sweeps, int *__restrict__ sweeps_out, int u_in_lds) {
    extern __shared__ double rows[];               // m x N, row stride N (+ m x m for U when it fits)
    __shared__ int rotated, rotated_large;
    const int tid = threadIdx.x;
    for (long i = tid; i < (long)m * N; i += TTSVD_LDS_THREADS) rows[i] = Bg[i];
    double *U = Ug;
    if (u_in_lds) {
        U = rows + (size_t)m * N;
        for (int i = tid; i < m * m; i += TTSVD_LDS_THREADS) U[i] = Ug[i];
    }
    __syncthreads();
    const int mp = (m + 1) & ~1;
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        if (tid == 0) { rotated = 0; rotated_large = 0; }
        __syncthreads();
        for (int step = 0; step < mp - 1; ++step) {
            // a row pair per 16-lane group (64 groups): the 44 pairs of an 88-row step run in ONE round,
            // and the sums need four shuffle levels inside a group instead of six across a wave
            for (int i = tid >> 4; i < mp / 2; i += TTSVD_LDS_THREADS / 16) {
                const int l16 = tid & 15;
                int p, q;
                if (i == 0) { p = mp - 1; q = step % (mp - 1); }
                else { p = (step + i) % (mp - 1); q = (step - i + 2 * (mp - 1)) % (mp - 1); }
                if (p >= m || q >= m) continue;
                if (p > q) { int t = p; p = q; q = t; }
                double *a = rows + (long)p * N, *b = rows + (long)q * N;
                double aa = 0.0, bb = 0.0, ab = 0.0;
                for (int j = l16; j < N; j += 16) {
                    const double x = a[j], y = b[j];
                    aa = __builtin_fma(x, x, aa);
                    bb = __builtin_fma(y, y, bb);
                    ab = __builtin_fma(x, y, ab);
                }
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    aa += __shfl_xor(aa, o, 16);
                    bb += __shfl_xor(bb, o, 16);
                    ab += __shfl_xor(ab, o, 16);
                }
                if (!(aa > floor2) || !(bb > floor2)) continue;
                if (aa + bb < sig2) continue;
                if (ab * ab <= (rot_tol * rot_tol) * aa * bb) continue;
                const bool large = ab * ab > 1e-16 * aa * bb;
                const double zeta = (bb - aa) / (2.0 * ab);
                const double t = __builtin_copysign(1.0, zeta) / (__builtin_fabs(zeta) + __builtin_sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / __builtin_sqrt(1.0 + t * t);
                const double sn = c * t;
                for (int j = l16; j < N; j += 16) {
                    const double x = a[j], y = b[j];
                    a[j] = c * x - sn * y;
                    b[j] = sn * x + c * y;
                }
                for (int r = l16; r < m; r += 16) {
                    const double x = U[(long)r * m + p], y = U[(long)r * m + q];
                    U[(long)r * m + p] = c * x - sn * y;
                    U[(long)r * m + q] = sn * x + c * y;
                }
                if (l16 == 0) { atomicAdd(&rotated, 1); if (large) atomicAdd(&rotated_large, 1); }
            }
            __syncthreads();
        }
        const int done = (rotated_large == 0);     // see k_rowjacobi_step: the sweep just finished was the last
        __syncthreads();
        if (done) { ++sweep; break; }
    }
    for (long i = tid; i < (long)m * N; i += TTSVD_LDS_THREADS) Bg[i] = rows[i];
    if (u_in_lds)
        for (int i = tid; i < m * m; i += TTSVD_LDS_THREADS) Ug[i] = U[i];
    if (tid == 0 && sweeps_out) *sweeps_out = sweep;
}

// ---- Gram preconditioner for wide unfoldings (N >> m) ----------------------------------------
// The rotations of the row iteration depend on the rows only through their inner products
// G = C C^T.  Running the same tournament on the m x m matrix G (two-sided: G <- R G R^T) in LDS
// costs O(m) per rotation instead of O(N), but sees singular values only through their squares:
// everything below sqrt(eps) sigma_max is noise there.  So it is used as a PRECONDITIONER: its
// accumulated rotations V make the rows of B0 = V^T C orthogonal to ~1e-8, and the accurate
// one-sided iteration on B0 (k_rowjacobi_step, U started at V) then converges in one or two
// sweeps instead of a dozen -- with the relative accuracy of small singular values intact.
__global__ void __launch_bounds__(TTSVD_THREADS)
k_gram_rows(const double *__restrict__ B, long ldb, int m, long N, double *__restrict__ G) {
    __shared__ double red[4];
    const int p = blockIdx.x, q = blockIdx.y;
    if (p > q) return;
    const double *a = B + (long)p * ldb, *b = B + (long)q * ldb;
    double s = 0.0;
    for (long j = threadIdx.x; j < N; j += TTSVD_THREADS) s = __builtin_fma(a[j], b[j], s);
    s = ttsvd_block_sum(s, red);
    if (threadIdx.x == 0) { G[(long)p * m + q] = s; G[(long)q * m + p] = s; }
}

// Two-sided cyclic Jacobi on the symmetric m x m matrix G, one workgroup, G and V in LDS.
// Same tournament order and rotation formula as the row iteration; looser stopping rule (1e-9:
// a preconditioner) and a noise floor relative to the trace (squares below eps trace(G) are noise).
__global__ void __launch_bounds__(TTSVD_LDS_THREADS)
k_symjacobi_lds(const double *__restrict__ Gg, int m, double *__restrict__ Vg, double floor2, double rot_tol,
                int max_sweeps) {
    extern __shared__ double sm[];
    double *G = sm, *V = sm + (size_t)m * m;
    double *cs = V + (size_t)m * m;              // c[i], s[i] per pair
    int *pq = (int *)(cs + 2 * ((m + 1) / 2 + 1));
    __shared__ int rotated;
    const int tid = threadIdx.x;
    for (int i = tid; i < m * m; i += TTSVD_LDS_THREADS) { G[i] = Gg[i]; V[i] = (i / m == i % m) ? 1.0 : 0.0; }
    __syncthreads();
    const int mp = (m + 1) & ~1, np = mp / 2;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        if (tid == 0) rotated = 0;
        __syncthreads();
        for (int step = 0; step < mp - 1; ++step) {
            if (tid < np) {
                const int i = tid;
                int p, q;
                if (i == 0) { p = mp - 1; q = step % (mp - 1); }
                else { p = (step + i) % (mp - 1); q = (step - i + 2 * (mp - 1)) % (mp - 1); }
                if (p > q) { int t = p; p = q; q = t; }
                double c = 1.0, sn = 0.0;
                if (p < m && q < m) {
                    const double aa = G[p * m + p], bb = G[q * m + q], ab = G[p * m + q];
                    // a pair is rotated when at least one of its rows is above the Gram noise floor: the large
                    // rows must be cleared of their overlap with the small ones too, or the accurate iteration
                    // afterwards needs as many sweeps as without a preconditioner (11 instead of 5, measured)
                    if ((aa > floor2 || bb > floor2) && aa > 0.0 && bb > 0.0 &&
                        ab * ab > (rot_tol * rot_tol) * aa * bb) {
                        const double zeta = (bb - aa) / (2.0 * ab);
                        const double t = __builtin_copysign(1.0, zeta) / (__builtin_fabs(zeta) + __builtin_sqrt(1.0 + zeta * zeta));
                        c = 1.0 / __builtin_sqrt(1.0 + t * t);
                        sn = c * t;
                        // only rotations further than sqrt(rot_tol) from orthogonal call for another sweep
                        if (ab * ab > rot_tol * aa * bb) atomicAdd(&rotated, 1);
                    }
                } else { p = q = -1; }
                cs[2 * i] = c; cs[2 * i + 1] = sn; pq[2 * i] = p; pq[2 * i + 1] = q;
            }
            __syncthreads();
            for (int e = tid; e < np * m; e += TTSVD_LDS_THREADS) {        // rows: G <- R G
                const int i = e / m, j = e % m, p = pq[2 * i], q = pq[2 * i + 1];
                if (p < 0 || cs[2 * i + 1] == 0.0) continue;
                const double c = cs[2 * i], sn = cs[2 * i + 1];
                const double x = G[p * m + j], y = G[q * m + j];
                G[p * m + j] = c * x - sn * y;
                G[q * m + j] = sn * x + c * y;
            }
            __syncthreads();
            for (int e = tid; e < np * m; e += TTSVD_LDS_THREADS) {        // columns: G <- G R^T, V <- V R^T
                const int i = e / m, j = e % m, p = pq[2 * i], q = pq[2 * i + 1];
                if (p < 0 || cs[2 * i + 1] == 0.0) continue;
                const double c = cs[2 * i], sn = cs[2 * i + 1];
                double x = G[j * m + p], y = G[j * m + q];
                G[j * m + p] = c * x - sn * y;
                G[j * m + q] = sn * x + c * y;
                x = V[j * m + p]; y = V[j * m + q];
                V[j * m + p] = c * x - sn * y;
                V[j * m + q] = sn * x + c * y;
            }
            __syncthreads();
        }
        const int done = (rotated == 0);
        __syncthreads();
        if (done) break;
    }
    for (int i = tid; i < m * m; i += TTSVD_LDS_THREADS) Vg[i] = V[i];
}

// out = V^T C  (m x N): out[i][j] = sum_r V[r][i] C[r][j]
__global__ void k_apply_vt(const double *__restrict__ C, long ldc, int m, long N, const double *__restrict__ V,
                           double *__restrict__ out) {
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= N) return;
    double s = 0.0;
    for (int r = 0; r < m; ++r) s = __builtin_fma(V[(long)r * m + i], C[(long)r * ldc + j], s);
    out[(long)i * N + j] = s;
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
