// tt_wfirst_lab.hip -- bench of candidate forms of the small-rank TT evaluation kernel
// (5-D, ranks [1,8,8,8,6,1], n = 11: BASELINE config 3) against the shipped
// k_tt_eval_wfirst<8,3,1>.  Every variant is checked against a plain one-thread-per-point
// chain before it is timed.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pychebyshev_amd/csrc tools/tt_wfirst_lab.hip -o /tmp/tt_lab
//   /tmp/tt_lab [points]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "tt_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- reference: one thread per point, natural core layout --------------------------
struct RefModel { int d; int n[8]; int r[9]; long off[8]; double lo[8], hi[8]; };
__global__ void k_ref(RefModel m, const double *cores, const double *pts, double *out, long N) {
    long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    double v[16], w[16], q[32];
    v[0] = 1.0;
    for (int k = 0; k < m.d; ++k) {
        const double x = pts[p * m.d + k];
        const double s = 2.0 * (x - m.lo[k]) / (m.hi[k] - m.lo[k]) - 1.0;
        double tp = 1.0, tc = s;
        for (int j = 0; j < m.n[k]; ++j) { q[j] = tp; double tn = fma(2.0 * s, tc, -tp); tp = tc; tc = tn; }
        const double *G = cores + m.off[k];
        for (int b = 0; b < m.r[k + 1]; ++b) {
            double acc = 0.0;
            for (int a = 0; a < m.r[k]; ++a) {
                double ww = 0.0;
                for (int j = 0; j < m.n[k]; ++j) ww = fma(q[j], G[((long)a * m.n[k] + j) * m.r[k + 1] + b], ww);
                acc = fma(v[a], ww, acc);
            }
            w[b] = acc;
        }
        for (int b = 0; b < m.r[k + 1]; ++b) v[b] = w[b];
    }
    out[p] = v[0];
}

// ---- probe of the gfx950 lane-swap instructions -------------------------------------
__global__ void k_swap_probe(unsigned *o) {
    unsigned x = threadIdx.x, y = 100 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}

// Exchange-and-add across lane halves / 16-lane rows (the two steps of a reduce-scatter over
// the four lane groups g = lane >> 4):
//   xadd32(X, Y): lanes 0..31 get X(l) + X(l+32), lanes 32..63 get Y(l-32) + Y(l)
//   xadd16(X, Y): even rows get X(l) + X(l+16),   odd rows  get Y(l-16) + Y(l)
__device__ __forceinline__ double xadd32(double X, double Y) {
    unsigned xl = __double2loint(X), xh = __double2hiint(X), yl = __double2loint(Y), yh = __double2hiint(Y);
    auto lo = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
    auto hi = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ double xadd16(double X, double Y) {
    unsigned xl = __double2loint(X), xh = __double2hiint(X), yl = __double2loint(Y), yh = __double2hiint(Y);
    auto lo = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
    auto hi = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// ---- V1 / V2: every A fragment of the model lives in registers ------------------------
// image: dim 0: [s][64] rows rho = b;  dims 1..D-2: [s][t][64], row rho = 16 t + (l & 15);
//   ORDER 0 (V1): rho = a R + b (as the shipped kernel; fold needs all of v in every lane)
//   ORDER 1 (V2): rho = 4u + g  <->  a = 4 (u % RA) + g, b = u / RA: lane group g meets only
//                 the entries a = g, g+4, ... it owns; partial sums meet by lane swaps
//   last dim: table [R][4 KS] in LDS.
template <int R, int KS, int D, int ORDER>
__global__ void __launch_bounds__(256, 2)
k_tt_wreg(TTDims dims, const double *__restrict__ img, const double *__restrict__ pts, double *__restrict__ out, long N) {
    constexpr int TILES = R * R / 16;
    constexpr int RB = R / 4;
    constexpr int NP = 4 * KS;
    constexpr int MID = D - 2;
    constexpr int CNT = 16 * D;               // doubles in a wave's block of query rows
    constexpr int PF = (CNT + 63) / 64;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    // LDS: last-dim table, lo/width tables, per-wave query rows
    double *gl_t = lds;
    double *lo_t = gl_t + R * NP, *wd_t = lo_t + CNT;
    double *xs = wd_t + CNT + wave * CNT;
    const long off_last = (long)KS * 64 + (long)MID * KS * TILES * 64;
    for (int i = threadIdx.x; i < R * NP; i += 256) gl_t[i] = img[off_last + i];
    for (int i = threadIdx.x; i < CNT; i += 256) {
        const int c = i % D;
        double lo = 0.0, wd = 1.0;
        for (int kk = 0; kk < D; ++kk)
            if (dims.col[kk] == c) { lo = dims.lo[kk]; wd = dims.hi[kk] - dims.lo[kk]; }
        lo_t[i] = lo; wd_t[i] = (ORDER == 2) ? 2.0 / wd : wd;
    }
    // A fragments -> registers, once per workgroup lifetime
    double a0[KS], am[MID][KS][TILES];
#pragma unroll
    for (int s = 0; s < KS; ++s) a0[s] = img[s * 64 + lane];
#pragma unroll
    for (int k = 0; k < MID; ++k)
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int t = 0; t < TILES; ++t) am[k][s][t] = img[KS * 64 + ((k * KS + s) * TILES + t) * 64 + lane];

    const long nbatch = (N + 63) / 64;
    double pf[PF];
    auto fetch = [&](long batch) {
        const long base = (batch * 4 + wave) * 16;
        const long first = base * D, avail = (N - base) * (long)D;
#pragma unroll
        for (int r = 0; r < PF; ++r) { const int i = lane + 64 * r; pf[r] = (i < CNT && i < avail) ? pts[first + i] : 0.0; }
    };
    if ((long)blockIdx.x < nbatch) fetch(blockIdx.x);
    __syncthreads();

    for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const long base = (batch * 4 + wave) * 16;
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            const int i = lane + 64 * r;
            if (i < CNT) xs[i] = (ORDER == 2) ? fma(pf[r] - lo_t[i], wd_t[i], -1.0) : 2.0 * (pf[r] - lo_t[i]) / wd_t[i] - 1.0;
        }
        if (batch + gridDim.x < nbatch) fetch(batch + gridDim.x);

        // Chebyshev values for lane group g: T_g, T_{4+g}, ... (stride-4 recurrence)
        auto cheb = [&](double x, double (&u)[KS]) {
            const double x2 = 2.0 * x, t2 = fma(x2, x, -1.0), t3 = fma(x2, t2, -x), t4 = fma(x2, t3, -t2);
            const double tg = (g == 0) ? 1.0 : (g == 1) ? x : (g == 2) ? t2 : t3;
            const double tm = (g == 0) ? t4 : (g == 1) ? t3 : (g == 2) ? t2 : x;
            const double c4 = 2.0 * t4;
            u[0] = tg;
            if (KS > 1) u[1] = fma(c4, tg, -tm);
#pragma unroll
            for (int s = 2; s < KS; ++s) u[s] = fma(c4, u[s - 1], -u[s - 2]);
        };
        double vown[RB];
        {   // dimension 0: v'[b] = W[b], lane (g, reg i) holds b = 4 i + g
            double u[KS];
            cheb(xs[c16 * D + dims.col[0]], u);
            pcx_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s], u[s], acc, 0, 0, 0);
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) vown[bi] = acc[bi];
        }
#pragma unroll
        for (int k = 0; k < MID; ++k) {
            double u[KS];
            cheb(xs[c16 * D + dims.col[k + 1]], u);
            pcx_d4 acc[TILES];
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int t = 0; t < TILES; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(am[k][s][t], u[s], s == 0 ? (pcx_d4){0.0, 0.0, 0.0, 0.0} : acc[t], 0, 0, 0);
            if (ORDER == 0) {
                double v[R];
#pragma unroll
                for (int b = 0; b < R; ++b) v[b] = __shfl(vown[b >> 2], ((b & 3) << 4) | c16, 64);
                double pb[RB];
#pragma unroll
                for (int bi = 0; bi < RB; ++bi) pb[bi] = 0.0;
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const int uu = 4 * t + i; pb[uu % RB] = fma(v[uu / RB], acc[t][i], pb[uu % RB]); }
#pragma unroll
                for (int bi = 0; bi < RB; ++bi) vown[bi] = pb[bi];
            } else {
                // partial sums over the a this lane group owns, for every b
                double P[R];
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    double s = 0.0;
#pragma unroll
                    for (int ca = 0; ca < RB; ++ca) { const int uu = b * RB + ca; s = fma(vown[ca], acc[uu >> 2][uu & 3], s); }
                    P[b] = s;
                }
                // reduce-scatter over the four lane groups: group g ends with b = 4 cb + g
                double Q[R / 2];
#pragma unroll
                for (int h = 0; h < R / 2; ++h) { const int b = (h >> 1) * 4 + (h & 1); Q[h] = xadd32(P[b], P[b | 2]); }
#pragma unroll
                for (int cb = 0; cb < RB; ++cb) vown[cb] = xadd16(Q[2 * cb], Q[2 * cb + 1]);
            }
        }
        {   // last dimension: y = sum_a v[a] sum_j T_j G[a][j]; lane group g takes a = 4 bi + g
            const double *gl = gl_t + g * NP;
            const double sc = xs[c16 * D + dims.col[D - 1]];
            double w[RB];
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) w[bi] = 0.0;
            double tp = 1.0, tc = sc;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
#pragma unroll
                for (int bi = 0; bi < RB; ++bi) w[bi] = fma(tp, gl[bi * 4 * NP + j], w[bi]);
                const double tn = fma(2.0 * sc, tc, -tp);
                tp = tc; tc = tn;
            }
            double y = 0.0;
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) y = fma(vown[bi], w[bi], y);
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            const long p = base + c16;
            if (g == 0 && p < N) out[p] = y;
        }
    }
}

// ---- V3: the same W-first GEMM on v_mfma_f64_4x4x4_4b_f64 ----------------------------------
// Blocks = the four 4-point groups of the wave's 16 points (B / D lane = 16 k + 4 b + j: the
// same lane <-> (node phase, point) map as above), A = one 4-row chunk of the core matrix,
// identical in the four blocks: read from the LDS image with a broadcast ds_read_b128 that
// brings the two left chunks (a = 4 ca + i, ca = 0, 1) of one output b.  16 cycles per
// instruction with no issue gap (profiles/r01_fp64_mfma4x4_microbench.txt: 97 % of peak vs
// 85 % for 16x16x4), no padding of 8 rows to 16, and chunks beyond a dimension's true right
// rank are skipped (uniform branch).
struct TTW4Plan {
    int rr[PCX_MAX_DIMS];       // true right rank of dim k (outputs b < rr[k] are computed)
    int lds_off[PCX_MAX_DIMS];  // doubles
    int total;
};
typedef double pcx_d2 __attribute__((ext_vector_type(2)));

template <int KS, int STAGE>
__global__ void __launch_bounds__(256, STAGE)
k_tt_w4(TTDims dims, TTW4Plan plan, const double *__restrict__ img, const double *__restrict__ pts,
        double *__restrict__ out, long N) {
    constexpr int R = 8, RB = 2, NP = 4 * KS;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int d = dims.d;
    const int cnt = 16 * d;
    constexpr int PF = 4;                           // 16 d / 64 staged elements per lane, d <= 16
    for (int i = threadIdx.x; i < plan.total; i += 256) lds[i] = img[i];
    double *lo_t = lds + plan.total, *wd_t = lo_t + cnt;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int c = i % d;
        double lo = 0.0, wd = 1.0;
        for (int kk = 0; kk < d; ++kk)
            if (dims.col[kk] == c) { lo = dims.lo[kk]; wd = dims.hi[kk] - dims.lo[kk]; }
        lo_t[i] = lo; wd_t[i] = 2.0 / wd;
    }
    double *xs = wd_t + cnt + (size_t)wave * cnt;
    const int a_idx = ((lane >> 4) * 4 + (lane & 3)) * 2;      // (k, i) slot of this lane in a chunk-pair block
    const long nbatch = (N + 63) / 64;
    double pf[PF];
    auto fetch = [&](long batch) {
        const long base = (batch * 4 + wave) * 16;
        const long first = base * d, avail = (N - base) * (long)d;
#pragma unroll
        for (int r = 0; r < PF; ++r) { const int i = lane + 64 * r; pf[r] = (i < cnt && i < avail) ? pts[first + i] : 0.0; }
    };
    if ((long)blockIdx.x < nbatch) fetch(blockIdx.x);
    __syncthreads();

    for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const long base = (batch * 4 + wave) * 16;
#pragma unroll
        for (int r = 0; r < PF; ++r) { const int i = lane + 64 * r; if (i < cnt) xs[i] = fma(pf[r] - lo_t[i], wd_t[i], -1.0); }
        if (batch + gridDim.x < nbatch) fetch(batch + gridDim.x);
        auto cheb = [&](double x, double (&u)[KS]) {
            const double x2 = 2.0 * x, t2 = fma(x2, x, -1.0), t3 = fma(x2, t2, -x), t4 = fma(x2, t3, -t2);
            const double tg = (g == 0) ? 1.0 : (g == 1) ? x : (g == 2) ? t2 : t3;
            const double tm = (g == 0) ? t4 : (g == 1) ? t3 : (g == 2) ? t2 : x;
            const double c4 = 2.0 * t4;
            u[0] = tg;
            if (KS > 1) u[1] = fma(c4, tg, -tm);
#pragma unroll
            for (int s = 2; s < KS; ++s) u[s] = fma(c4, u[s - 1], -u[s - 2]);
        };
        double vown[RB];
        {   // dimension 0: rows = b, one chunk pair per k-step
            double u[KS];
            cheb(xs[c16 * d + dims.col[0]], u);
            const double *f0 = lds + plan.lds_off[0] + a_idx;
            double c0 = 0.0, c1 = 0.0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const pcx_d2 a = *(const pcx_d2 *)(f0 + s * 32);
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, u[s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, u[s], c1, 0, 0, 0);
            }
            vown[0] = c0; vown[1] = c1;
        }
        for (int k = 1; k < d - 1; ++k) {
            double u[KS];
            cheb(xs[c16 * d + dims.col[k]], u);
            const double *fk = lds + plan.lds_off[k] + a_idx;
            const int rr = plan.rr[k];
            double acc[R][2];
            pcx_d2 A[2][R];
#pragma unroll
            for (int b = 0; b < R; ++b) A[0][b] = *(const pcx_d2 *)(fk + b * 32);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                if (s + 1 < KS) {
#pragma unroll
                    for (int b = 0; b < R; ++b) A[(s + 1) & 1][b] = *(const pcx_d2 *)(fk + ((s + 1) * R + b) * 32);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    if (b < rr) {
                        acc[b][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[s & 1][b].x, u[s], s == 0 ? 0.0 : acc[b][0], 0, 0, 0);
                        acc[b][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[s & 1][b].y, u[s], s == 0 ? 0.0 : acc[b][1], 0, 0, 0);
                    }
                }
            }
            double P[R];
#pragma unroll
            for (int b = 0; b < R; ++b) P[b] = (b < rr) ? fma(vown[1], acc[b][1], vown[0] * acc[b][0]) : 0.0;
            double Q[R / 2];
#pragma unroll
            for (int h = 0; h < R / 2; ++h) { const int b = (h >> 1) * 4 + (h & 1); Q[h] = xadd32(P[b], P[b | 2]); }
#pragma unroll
            for (int cb = 0; cb < RB; ++cb) vown[cb] = xadd16(Q[2 * cb], Q[2 * cb + 1]);
        }
        {   // last dimension on the VALU, as before
            const double *gl = lds + plan.lds_off[d - 1] + g * NP;
            const double sc = xs[c16 * d + dims.col[d - 1]];
            double w[RB];
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) w[bi] = 0.0;
            double tp = 1.0, tc = sc;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
#pragma unroll
                for (int bi = 0; bi < RB; ++bi) w[bi] = fma(tp, gl[bi * 4 * NP + j], w[bi]);
                const double tn = fma(2.0 * sc, tc, -tp);
                tp = tc; tc = tn;
            }
            double y = 0.0;
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) y = fma(vown[bi], w[bi], y);
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            const long p = base + c16;
            if (g == 0 && p < N) out[p] = y;
        }
    }
}

// ---- V4: V3 as straight-line code (compile-time D, all 8 outputs per dim), two schedules ----
template <int KS, int D, int SCHED>
__global__ void __launch_bounds__(256, 2)
k_tt_w4s(TTDims dims, TTW4Plan plan, const double *__restrict__ img, const double *__restrict__ pts,
         double *__restrict__ out, long N) {
    constexpr int R = 8, RB = 2, NP = 4 * KS, MID = D - 2;
    constexpr int CNT = 16 * D, PF = (CNT + 63) / 64;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    for (int i = threadIdx.x; i < plan.total; i += 256) lds[i] = img[i];
    double *lo_t = lds + plan.total, *wd_t = lo_t + CNT;
    for (int i = threadIdx.x; i < CNT; i += 256) {
        const int c = i % D;
        double lo = 0.0, wd = 1.0;
        for (int kk = 0; kk < D; ++kk)
            if (dims.col[kk] == c) { lo = dims.lo[kk]; wd = dims.hi[kk] - dims.lo[kk]; }
        lo_t[i] = lo; wd_t[i] = 2.0 / wd;
    }
    double *xs = wd_t + CNT + (size_t)wave * CNT;
    const int a_idx = ((lane >> 4) * 4 + (lane & 3)) * 2;
    const long nbatch = (N + 63) / 64;
    double pf[PF];
    auto fetch = [&](long batch) {
        const long base = (batch * 4 + wave) * 16;
        const long first = base * D, avail = (N - base) * (long)D;
#pragma unroll
        for (int r = 0; r < PF; ++r) { const int i = lane + 64 * r; pf[r] = (i < CNT && i < avail) ? pts[first + i] : 0.0; }
    };
    if ((long)blockIdx.x < nbatch) fetch(blockIdx.x);
    __syncthreads();
    const double *f0 = lds + plan.lds_off[0] + a_idx;
    const double *fm = lds + plan.lds_off[1] + a_idx;        // mid dims are contiguous: KS * R * 32 doubles each

    for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const long base = (batch * 4 + wave) * 16;
#pragma unroll
        for (int r = 0; r < PF; ++r) { const int i = lane + 64 * r; if (i < CNT) xs[i] = fma(pf[r] - lo_t[i], wd_t[i], -1.0); }
        if (batch + gridDim.x < nbatch) fetch(batch + gridDim.x);
        auto cheb = [&](double x, double (&u)[KS]) {
            const double x2 = 2.0 * x, t2 = fma(x2, x, -1.0), t3 = fma(x2, t2, -x), t4 = fma(x2, t3, -t2);
            const double tg = (g == 0) ? 1.0 : (g == 1) ? x : (g == 2) ? t2 : t3;
            const double tm = (g == 0) ? t4 : (g == 1) ? t3 : (g == 2) ? t2 : x;
            const double c4 = 2.0 * t4;
            u[0] = tg;
            if (KS > 1) u[1] = fma(c4, tg, -tm);
#pragma unroll
            for (int s = 2; s < KS; ++s) u[s] = fma(c4, u[s - 1], -u[s - 2]);
        };
        // every Chebyshev value of the batch first: the GEMMs of all dimensions are independent
        // of the chain, only the folds are sequential
        double u[D - 1][KS];
#pragma unroll
        for (int k = 0; k < D - 1; ++k) cheb(xs[c16 * D + dims.col[k]], u[k]);
        double vown[RB];
        {
            double c0 = 0.0, c1 = 0.0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const pcx_d2 a = *(const pcx_d2 *)(f0 + s * 32);
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, u[0][s], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, u[0][s], c1, 0, 0, 0);
            }
            vown[0] = c0; vown[1] = c1;
        }
#pragma unroll
        for (int k = 0; k < MID; ++k) {
            const double *fk = fm + (size_t)k * KS * R * 32;
            double acc[R][2];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                pcx_d2 A[R];
#pragma unroll
                for (int b = 0; b < R; ++b) A[b] = *(const pcx_d2 *)(fk + (s * R + b) * 32);
                if (SCHED == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < R; ++b) {
                    acc[b][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[b].x, u[k + 1][s], s == 0 ? 0.0 : acc[b][0], 0, 0, 0);
                    acc[b][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[b].y, u[k + 1][s], s == 0 ? 0.0 : acc[b][1], 0, 0, 0);
                }
                if (SCHED == 2) {
                    // one LDS read per two MFMAs, reads running one step ahead
#pragma unroll
                    for (int b = 0; b < R; ++b) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    }
                }
            }
            double P[R];
#pragma unroll
            for (int b = 0; b < R; ++b) P[b] = fma(vown[1], acc[b][1], vown[0] * acc[b][0]);
            double Q[R / 2];
#pragma unroll
            for (int h = 0; h < R / 2; ++h) { const int b = (h >> 1) * 4 + (h & 1); Q[h] = xadd32(P[b], P[b | 2]); }
#pragma unroll
            for (int cb = 0; cb < RB; ++cb) vown[cb] = xadd16(Q[2 * cb], Q[2 * cb + 1]);
        }
        {
            const double *gl = lds + plan.lds_off[D - 1] + g * NP;
            const double sc = xs[c16 * D + dims.col[D - 1]];
            double w[RB];
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) w[bi] = 0.0;
            double tp = 1.0, tc = sc;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
#pragma unroll
                for (int bi = 0; bi < RB; ++bi) w[bi] = fma(tp, gl[bi * 4 * NP + j], w[bi]);
                const double tn = fma(2.0 * sc, tc, -tp);
                tp = tc; tc = tn;
            }
            double y = 0.0;
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) y = fma(vown[bi], w[bi], y);
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            const long p = base + c16;
            if (g == 0 && p < N) out[p] = y;
        }
    }
}

// ---- V5: 4x4x4_4b, runtime d, outputs trimmed to the true right rank, and the Chebyshev
// seeds computed ONCE per (dimension, point): lane group g takes dimension 4r + g of round r,
// writes {T_0 .. T_4, 2 T_4}(x) to a per-wave LDS table, and every lane then reads T_g,
// T_{4-g}, 2 T_4 of the dimension being multiplied with a lane-dependent ADDRESS instead of
// computing them redundantly in the four lane groups and selecting (branches / cndmasks).
template <int KS, int NB>
__device__ __forceinline__ void w4_mid(const double *fk, const double (&u)[KS], double (&vown)[2]) {
    double acc[NB][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        pcx_d2 A[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) A[b] = *(const pcx_d2 *)(fk + (s * 8 + b) * 32);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            acc[b][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[b].x, u[s], s == 0 ? 0.0 : acc[b][0], 0, 0, 0);
            acc[b][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(A[b].y, u[s], s == 0 ? 0.0 : acc[b][1], 0, 0, 0);
        }
    }
    double P[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) P[b] = (b < NB) ? fma(vown[1], acc[b < NB ? b : 0][1], vown[0] * acc[b < NB ? b : 0][0]) : 0.0;
    if (NB <= 4) {
        const double q0 = xadd32(P[0], P[2]), q1 = xadd32(P[1], P[3]);
        vown[0] = xadd16(q0, q1);
        vown[1] = 0.0;
    } else {
        double Q[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) { const int b = (h >> 1) * 4 + (h & 1); Q[h] = xadd32(P[b], P[b | 2]); }
        vown[0] = xadd16(Q[0], Q[1]);
        vown[1] = xadd16(Q[2], Q[3]);
    }
}

template <int KS, int WPS>
__global__ void __launch_bounds__(256, WPS)
k_tt_w5(TTDims dims, TTW4Plan plan, const double *__restrict__ img, const double *__restrict__ pts,
        double *__restrict__ out, long N) {
    constexpr int RB = 2, NP = 4 * KS;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int d = dims.d;
    const int cnt = 16 * d;
    constexpr int PF = 4;
    for (int i = threadIdx.x; i < plan.total; i += 256) lds[i] = img[i];
    double *lo_t = lds + plan.total, *wd_t = lo_t + cnt;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int c = i % d;
        double lo = 0.0, wd = 1.0;
        for (int kk = 0; kk < d; ++kk)
            if (dims.col[kk] == c) { lo = dims.lo[kk]; wd = dims.hi[kk] - dims.lo[kk]; }
        lo_t[i] = lo; wd_t[i] = 2.0 / wd;
    }
    double *xs = wd_t + cnt + (size_t)wave * (cnt + 4 * 16 * 6);
    double *seed = xs + cnt;                                   // [slot g][point][6]
    const int a_idx = ((lane >> 4) * 4 + (lane & 3)) * 2;
    double *my_seed = seed + (g * 16 + c16) * 6;               // what this lane writes in a round
    const long nbatch = (N + 63) / 64;
    double pf[PF];
    auto fetch = [&](long batch) {
        const long base = (batch * 4 + wave) * 16;
        const long first = base * d, avail = (N - base) * (long)d;
#pragma unroll
        for (int r = 0; r < PF; ++r) { const int i = lane + 64 * r; pf[r] = (i < cnt && i < avail) ? pts[first + i] : 0.0; }
    };
    if ((long)blockIdx.x < nbatch) fetch(blockIdx.x);
    __syncthreads();

    for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const long base = (batch * 4 + wave) * 16;
#pragma unroll
        for (int r = 0; r < PF; ++r) { const int i = lane + 64 * r; if (i < cnt) xs[i] = fma(pf[r] - lo_t[i], wd_t[i], -1.0); }
        if (batch + gridDim.x < nbatch) fetch(batch + gridDim.x);
        double vown[RB];
        for (int k = 0; k < d - 1; ++k) {
            if ((k & 3) == 0) {
                // seeds of dimensions k .. k+3, one per lane group
                const int kd = (k + g < d - 1) ? k + g : k;
                const double x = xs[c16 * d + dims.col[kd]];
                const double x2 = x + x, t2 = fma(x2, x, -1.0), t3 = fma(x2, t2, -x), t4 = fma(x2, t3, -t2);
                asm volatile("" ::: "memory");
                *(pcx_d2 *)(my_seed) = (pcx_d2){1.0, x};
                *(pcx_d2 *)(my_seed + 2) = (pcx_d2){t2, t3};
                *(pcx_d2 *)(my_seed + 4) = (pcx_d2){t4, t4 + t4};
                asm volatile("" ::: "memory");
            }
            const double *sk = seed + ((k & 3) * 16 + c16) * 6;
            double u[KS];
            u[0] = sk[g];
            if (KS > 1) { const double tm = sk[4 - g], c4 = sk[5]; u[1] = fma(c4, u[0], -tm);
#pragma unroll
                for (int s = 2; s < KS; ++s) u[s] = fma(c4, u[s - 1], -u[s - 2]); }
            const double *fk = lds + plan.lds_off[k] + a_idx;
            if (k == 0) {
                double c0 = 0.0, c1 = 0.0;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const pcx_d2 a = *(const pcx_d2 *)(fk + s * 32);
                    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, u[s], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, u[s], c1, 0, 0, 0);
                }
                vown[0] = c0; vown[1] = c1;
            } else {
                const int rr = plan.rr[k];
                if (rr > 6) w4_mid<KS, 8>(fk, u, vown);
                else if (rr > 4) w4_mid<KS, 6>(fk, u, vown);
                else if (rr > 2) w4_mid<KS, 4>(fk, u, vown);
                else w4_mid<KS, 2>(fk, u, vown);
            }
        }
        {
            const double *gl = lds + plan.lds_off[d - 1] + g * NP;
            const double sc = xs[c16 * d + dims.col[d - 1]];
            double w[RB];
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) w[bi] = 0.0;
            double tp = 1.0, tc = sc;
            const double sc2 = sc + sc;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
#pragma unroll
                for (int bi = 0; bi < RB; ++bi) w[bi] = fma(tp, gl[bi * 4 * NP + j], w[bi]);
                const double tn = fma(sc2, tc, -tp);
                tp = tc; tc = tn;
            }
            double y = 0.0;
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) y = fma(vown[bi], w[bi], y);
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            const long p = base + c16;
            if (g == 0 && p < N) out[p] = y;
        }
    }
}

// ---- V6: direct form on 4x4x4_4b: v'[b,p] = sum_{(j,a)} G[a,j,b] (v[a,p] T_j(x_p)) ----------
// K runs over (node j, left chunk c); B operand of lane (k, p) = z_j[c] = v[4c + k][p] T_j(x_p),
// generated by the Chebyshev recurrence ON THE PRODUCTS (z_{j+1} = 2x z_j - z_{j-1},
// z_0 = v, z_{-1} = x v): one FMA per (j, c), no table of T_j, no fold, no lane exchange -- the
// rows a lane needs as B for chunk c are the D values (rows 4m + i, i = lane >> 4) it already
// holds.  No padding of n to a multiple of 4 and none of 8 rows to 16.
// image: dim 0: [s][32] chunk-pair blocks (rows b = 4m + i, nodes 4s + k4);
//        mid dims: [j][c][32]: slot (k4*4 + i)*2 + m = G[a = 4c + k4][j][b = 4m + i]; last: [8][NP].
template <int PF, int NJ, int WPS, int SPLIT = 0>
__global__ void __launch_bounds__(256, WPS)
k_tt_d4(TTDims dims, TTW4Plan plan, const double *__restrict__ img, const double *__restrict__ pts,
        double *__restrict__ out, long N, int ks0, int np) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int d = dims.d;
    const int cnt = 16 * d;
    for (int i = threadIdx.x; i < plan.total; i += 256) lds[i] = img[i];
    double *lo_t = lds + plan.total, *wd_t = lo_t + cnt;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int c = i % d;
        double lo = 0.0, wd = 1.0;
        for (int kk = 0; kk < d; ++kk)
            if (dims.col[kk] == c) { lo = dims.lo[kk]; wd = dims.hi[kk] - dims.lo[kk]; }
        lo_t[i] = lo; wd_t[i] = 2.0 / wd;
    }
    double *xs = wd_t + cnt + (size_t)wave * (cnt + 16 * 6);
    double *seed = xs + cnt + c16 * 6;                          // this lane's point: {T_0..T_4, 2 T_4}(x_0)
    const int a_idx = ((lane >> 4) * 4 + (lane & 3)) * 2;
    const long nbatch = (N + 63) / 64;
    double pf[PF];
    auto fetch = [&](long batch) {
        const long base = (batch * 4 + wave) * 16;
        const long first = base * d, avail = (N - base) * (long)d;
#pragma unroll
        for (int r = 0; r < PF; ++r) { const int i = lane + 64 * r; pf[r] = (i < cnt && i < avail) ? pts[first + i] : 0.0; }
    };
    if ((long)blockIdx.x < nbatch) fetch(blockIdx.x);
    __syncthreads();

    for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const long base = (batch * 4 + wave) * 16;
#pragma unroll
        for (int r = 0; r < PF; ++r) { const int i = lane + 64 * r; if (i < cnt) xs[i] = fma(pf[r] - lo_t[i], wd_t[i], -1.0); }
        if (batch + gridDim.x < nbatch) fetch(batch + gridDim.x);
        double v0, v1;
        {   // dimension 0 (left rank 1): B = T_{4s + k}(x): stride-4 recurrence from a seed table
            const double x = xs[c16 * d + dims.col[0]];
            const double x2 = x + x, t2 = fma(x2, x, -1.0), t3 = fma(x2, t2, -x), t4 = fma(x2, t3, -t2);
            asm volatile("" ::: "memory");       // LDS ops of one wave execute in order; keep the compiler from reordering them
            if (g == 0) {
                *(pcx_d2 *)(seed) = (pcx_d2){1.0, x};
                *(pcx_d2 *)(seed + 2) = (pcx_d2){t2, t3};
                *(pcx_d2 *)(seed + 4) = (pcx_d2){t4, t4 + t4};
            }
            asm volatile("" ::: "memory");
            double up = seed[4 - g], uc = seed[g];
            const double c4 = seed[5];
            const double *f0 = lds + plan.lds_off[0] + a_idx;
            double c0 = 0.0, c1 = 0.0;
            for (int s = 0; s < ks0; ++s) {
                const pcx_d2 a = *(const pcx_d2 *)(f0 + s * 32);
                c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.x, uc, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a.y, uc, c1, 0, 0, 0);
                const double un = fma(c4, uc, -up);     // T_{4(s+1)+g} = 2 T_4 T_{4s+g} - T_{|4(s-1)+g|}
                up = uc; uc = un;
            }
            v0 = c0; v1 = c1;
        }
        for (int k = 1; k < d - 1; ++k) {
            const double x = xs[c16 * d + dims.col[k]];
            const double x2 = x + x;
            double zc0 = v0, zc1 = v1, zp0 = v0 * x, zp1 = v1 * x;
            const double *fk = lds + plan.lds_off[k] + a_idx;
            double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const pcx_d2 A0 = *(const pcx_d2 *)(fk + j * 64), A1 = *(const pcx_d2 *)(fk + j * 64 + 32);
                acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A0.x, zc0, j == 0 ? 0.0 : acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A0.y, zc0, j == 0 ? 0.0 : acc1, 0, 0, 0);
                if (SPLIT) {
                    acc2 = __builtin_amdgcn_mfma_f64_4x4x4f64(A1.x, zc1, j == 0 ? 0.0 : acc2, 0, 0, 0);
                    acc3 = __builtin_amdgcn_mfma_f64_4x4x4f64(A1.y, zc1, j == 0 ? 0.0 : acc3, 0, 0, 0);
                } else {
                acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(A1.x, zc1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(A1.y, zc1, acc1, 0, 0, 0);
                }
                if (j + 1 < NJ) {
                    const double zn0 = fma(x2, zc0, -zp0), zn1 = fma(x2, zc1, -zp1);
                    zp0 = zc0; zc0 = zn0; zp1 = zc1; zc1 = zn1;
                }
            }
            v0 = SPLIT ? acc0 + acc2 : acc0; v1 = SPLIT ? acc1 + acc3 : acc1;
        }
        {
            const double *gl = lds + plan.lds_off[d - 1] + g * np;
            const double sc = xs[c16 * d + dims.col[d - 1]];
            double w0 = 0.0, w1 = 0.0, tp = 1.0, tc = sc;
            const double sc2 = sc + sc;
            for (int j = 0; j < np; ++j) {
                w0 = fma(tp, gl[j], w0);
                w1 = fma(tp, gl[4 * np + j], w1);
                const double tn = fma(sc2, tc, -tp);
                tp = tc; tc = tn;
            }
            double y = fma(v1, w1, v0 * w0);
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            const long p = base + c16;
            if (g == 0 && p < N) out[p] = y;
        }
    }
}

// ---- V7: V6 in its product shape: left/right chunk count RA as template (ranks padded to 4 RA),
// node counts dispatched at run time to fully unrolled bodies, staging loop over runtime d.
// image: dim 0: [s][16 slots][NMP]; mid dims: [j][c < RA][16 slots][NMP]; last: [4 RA][n_last]
// (NMP = RA rounded up to 1, 2 or 4 so that a lane's values are one 8- or 16-byte aligned read).
template <int RA> struct D4Frag;
template <> struct D4Frag<1> { double v[1]; static constexpr int NMP = 1;
    __device__ __forceinline__ void load(const double *p) { v[0] = p[0]; } };
template <> struct D4Frag<2> { double v[2]; static constexpr int NMP = 2;
    __device__ __forceinline__ void load(const double *p) { const pcx_d2 t = *(const pcx_d2 *)p; v[0] = t.x; v[1] = t.y; } };
template <> struct D4Frag<3> { double v[3]; static constexpr int NMP = 4;
    __device__ __forceinline__ void load(const double *p) { const pcx_d2 t = *(const pcx_d2 *)p; v[0] = t.x; v[1] = t.y; v[2] = p[2]; } };

struct TTD4Plan {
    int lds_off[PCX_MAX_DIMS];
    int total;
};

template <int RA, int NJ>
__device__ __forceinline__ void d4_mid(const double *fk, double x, double (&v)[RA]) {
    constexpr int NMP = D4Frag<RA>::NMP;
    const double x2 = x + x;
    double zc[RA], zp[RA], acc[RA];
#pragma unroll
    for (int c = 0; c < RA; ++c) { zc[c] = v[c]; zp[c] = v[c] * x; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int c = 0; c < RA; ++c) {
            D4Frag<RA> a;
            a.load(fk + (j * RA + c) * 16 * NMP);
#pragma unroll
            for (int m = 0; m < RA; ++m)
                acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[m], zc[c], (j == 0 && c == 0) ? 0.0 : acc[m], 0, 0, 0);
        }
        if (j + 1 < NJ) {
#pragma unroll
            for (int c = 0; c < RA; ++c) { const double zn = fma(x2, zc[c], -zp[c]); zp[c] = zc[c]; zc[c] = zn; }
        }
    }
#pragma unroll
    for (int m = 0; m < RA; ++m) v[m] = acc[m];
}

template <int RA, int NJ>
__device__ __forceinline__ double d4_last(const double *gl, double x, const double (&v)[RA]) {
    // y_partial = sum_c v[c] * sum_j T_j(x) G[a = 4c + g][j];  gl -> row g, rows 4 apart are NJ*4 doubles apart
    const double x2 = x + x;
    double w[RA], tp = 1.0, tc = x;
#pragma unroll
    for (int c = 0; c < RA; ++c) w[c] = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int c = 0; c < RA; ++c) w[c] = fma(tp, gl[c * 4 * NJ + j], w[c]);
        if (j + 1 < NJ) { const double tn = fma(x2, tc, -tp); tp = tc; tc = tn; }
    }
    double y = v[0] * w[0];
#pragma unroll
    for (int c = 1; c < RA; ++c) y = fma(v[c], w[c], y);
    return y;
}

#define D4_NODE_SWITCH(n, CALL)                                                            \
    switch (n) {                                                                           \
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break;     \
    case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;     \
    case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; default: CALL(16); break; }

template <int RA, int WPS>
__global__ void __launch_bounds__(256, WPS)
k_tt_eval_d4(TTDims dims, TTD4Plan plan, const double *__restrict__ img, const double *__restrict__ pts,
             double *__restrict__ out, long N) {
    constexpr int NMP = D4Frag<RA>::NMP;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int d = dims.d;
    const int cnt = 16 * d;
    for (int i = threadIdx.x; i < plan.total; i += 256) lds[i] = img[i];
    double *lo_t = lds + plan.total, *wd_t = lo_t + cnt;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int c = i % d;
        double lo = 0.0, wd = 1.0;
        for (int kk = 0; kk < d; ++kk)
            if (dims.col[kk] == c) { lo = dims.lo[kk]; wd = dims.hi[kk] - dims.lo[kk]; }
        lo_t[i] = lo; wd_t[i] = 2.0 / wd;
    }
    double *xs = wd_t + cnt + (size_t)wave * (cnt + 16 * 6);
    double *seed = xs + cnt + c16 * 6;
    const int a_idx = ((lane >> 4) * 4 + (lane & 3)) * NMP;
    const long nbatch = (N + 63) / 64;
    constexpr int PFMAX = 4;                        // 16 d / 64, d <= 16
    const int npf = (cnt + 63) >> 6;
    double pf[PFMAX];
    auto fetch = [&](long batch) {
        const long base = (batch * 4 + wave) * 16;
        const long first = base * d, avail = (N - base) * (long)d;
#pragma unroll
        for (int r = 0; r < PFMAX; ++r)
            if (r < npf) { const int i = lane + 64 * r; pf[r] = (i < cnt && i < avail) ? pts[first + i] : 0.0; }
    };
    if ((long)blockIdx.x < nbatch) fetch(blockIdx.x);
    __syncthreads();
    const int n0 = dims.n[0], ks0 = (n0 + 3) >> 2;

    for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const long base = (batch * 4 + wave) * 16;
#pragma unroll
        for (int r = 0; r < PFMAX; ++r)
            if (r < npf) { const int i = lane + 64 * r; if (i < cnt) xs[i] = fma(pf[r] - lo_t[i], wd_t[i], -1.0); }
        if (batch + gridDim.x < nbatch) fetch(batch + gridDim.x);
        double v[RA];
        if (d > 1) {   // dimension 0 (left rank 1): B = T_{4s + k}(x): stride-4 recurrence from a seed table
            const double x = xs[c16 * d + dims.col[0]];
            const double x2 = x + x, t2 = fma(x2, x, -1.0), t3 = fma(x2, t2, -x), t4 = fma(x2, t3, -t2);
            asm volatile("" ::: "memory");       // LDS ops of one wave execute in order; keep the compiler from reordering them
            if (g == 0) {
                *(pcx_d2 *)(seed) = (pcx_d2){1.0, x};
                *(pcx_d2 *)(seed + 2) = (pcx_d2){t2, t3};
                *(pcx_d2 *)(seed + 4) = (pcx_d2){t4, t4 + t4};
            }
            asm volatile("" ::: "memory");
            double up = seed[4 - g], uc = seed[g];
            const double c4 = seed[5];
            const double *f0 = lds + plan.lds_off[0] + a_idx;
            double acc[RA];
#pragma unroll
            for (int m = 0; m < RA; ++m) acc[m] = 0.0;
            for (int s = 0; s < ks0; ++s) {
                D4Frag<RA> a;
                a.load(f0 + s * 16 * NMP);
#pragma unroll
                for (int m = 0; m < RA; ++m) acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[m], uc, acc[m], 0, 0, 0);
                const double un = fma(c4, uc, -up);
                up = uc; uc = un;
            }
#pragma unroll
            for (int m = 0; m < RA; ++m) v[m] = acc[m];
        } else {
#pragma unroll
            for (int m = 0; m < RA; ++m) v[m] = (m == 0 && g == 0) ? 1.0 : 0.0;
        }
        for (int k = 1; k < d - 1; ++k) {
            const double x = xs[c16 * d + dims.col[k]];
            const double *fk = lds + plan.lds_off[k] + a_idx;
#define D4_MID(NJ) d4_mid<RA, NJ>(fk, x, v)
            D4_NODE_SWITCH(dims.n[k], D4_MID)
#undef D4_MID
        }
        {
            const int nl = dims.n[d - 1];
            const double *gl = lds + plan.lds_off[d - 1] + g * nl;
            const double x = xs[c16 * d + dims.col[d - 1]];
            double y;
#define D4_LAST(NJ) y = d4_last<RA, NJ>(gl, x, v)
            D4_NODE_SWITCH(nl, D4_LAST)
#undef D4_LAST
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            const long p = base + c16;
            if (g == 0 && p < N) out[p] = y;
        }
    }
}

// ---- host ------------------------------------------------------------------------------
int main(int argc, char **argv) {
    const long N = argc > 1 ? atol(argv[1]) : 10000000L;
    const int D = 5, R = 8, KS = 3, n = 11;
    const int ranks[6] = {1, 8, 8, 8, 6, 1};
    std::mt19937_64 rng(7);
    std::normal_distribution<double> nd;
    RefModel rm; rm.d = D;
    std::vector<double> cores;
    for (int k = 0; k < D; ++k) {
        rm.n[k] = n; rm.r[k] = ranks[k]; rm.off[k] = (long)cores.size(); rm.lo[k] = -1.0 + 0.1 * k; rm.hi[k] = 1.0 + 0.3 * k;
        for (int i = 0; i < ranks[k] * n * ranks[k + 1]; ++i) cores.push_back(nd(rng) / std::sqrt((double)ranks[k] * n));
    }
    rm.r[D] = 1;
    TTDims dims{}; dims.d = D;
    for (int k = 0; k < D; ++k) { dims.n[k] = n; dims.col[k] = k; dims.lo[k] = rm.lo[k]; dims.hi[k] = rm.hi[k]; }
    std::vector<double> pts((size_t)N * D);
    std::uniform_real_distribution<double> ud(0.0, 1.0);
    for (long p = 0; p < N; ++p) for (int k = 0; k < D; ++k) pts[p * D + k] = rm.lo[k] + (rm.hi[k] - rm.lo[k]) * ud(rng);

    double *d_cores, *d_pts, *d_out, *d_ref;
    CK(hipMalloc(&d_cores, cores.size() * 8)); CK(hipMemcpy(d_cores, cores.data(), cores.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pts, pts.size() * 8)); CK(hipMemcpy(d_pts, pts.data(), pts.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_out, N * 8)); CK(hipMalloc(&d_ref, N * 8));
    hipLaunchKernelGGL(k_ref, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, rm, d_cores, d_pts, d_ref, N);
    CK(hipDeviceSynchronize());
    std::vector<double> ref(N), got(N);
    CK(hipMemcpy(ref.data(), d_ref, N * 8, hipMemcpyDeviceToHost));
    double scale = 0; for (double v : ref) scale = std::max(scale, std::fabs(v));

    {   // lane-swap semantics
        unsigned *d_o; CK(hipMalloc(&d_o, 256 * 4));
        hipLaunchKernelGGL(k_swap_probe, dim3(1), dim3(64), 0, 0, d_o);
        unsigned o[256]; CK(hipMemcpy(o, d_o, sizeof(o), hipMemcpyDeviceToHost));
        const char *names[4] = {"permlane32_swap[0]", "permlane32_swap[1]", "permlane16_swap[0]", "permlane16_swap[1]"};
        for (int r = 0; r < 4; ++r) { printf("%-20s", names[r]); for (int l = 0; l < 64; l += 8) printf(" l%-2d=%-3u", l, o[64 * r + l]); printf("\n"); }
    }

    auto check = [&](const char *name) {
        CK(hipMemcpy(got.data(), d_out, N * 8, hipMemcpyDeviceToHost));
        double e = 0; for (long p = 0; p < N; ++p) e = std::max(e, std::fabs(got[p] - ref[p]));
        printf("%-34s E_norm vs reference chain %.2e %s\n", name, e / scale, e / scale <= 1e-12 ? "ok" : "** MISMATCH **");
    };
    auto time_it = [&](const char *name, auto launch) {
        CK(hipMemset(d_out, 0, N * 8));
        launch(); CK(hipDeviceSynchronize()); check(name);
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(a));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
        printf("%-34s %.4f ms  %.3e pts/s  %.1f TFLOP/s algorithmic (%.3f of 78.6)\n", name, ms, N / (ms * 1e-3),
               4560.0 * N / (ms * 1e-3) / 1e12, 4560.0 * N / (ms * 1e-3) / 78.6e12);
    };
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;

    // ---- V0: shipped kernel --------------------------------------------------------------
    {
        TTWPlan plan{}; long total = 0;
        for (int k = 0; k < D; ++k) { plan.ntiles[k] = k == 0 ? 1 : R * R / 16; plan.lds_off[k] = (int)total; total += (k == D - 1) ? R * 4 * KS : (long)KS * plan.ntiles[k] * 64; }
        plan.ks = KS; plan.total = (int)total;
        double *d_img; CK(hipMalloc(&d_img, total * 8));
        for (int k = 0; k < D; ++k) {
            int last = k == D - 1; long cnt = last ? R * 4 * KS : (long)KS * plan.ntiles[k] * 64;
            hipLaunchKernelGGL(k_tt_pack_wfirst<8>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, d_cores + rm.off[k], d_img + plan.lds_off[k], ranks[k], n, ranks[k + 1], KS, plan.ntiles[k], last);
        }
        size_t ldsb = ((size_t)total + 6 * 16 * D) * 8;
        int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tt_eval_wfirst<8, 3, 1>, 256, ldsb));
        long blocks = std::min<long>((N + 63) / 64, (long)per_cu * cus * 4);
        printf("V0 occupancy %d WG/CU, %ld blocks\n", per_cu, blocks);
        time_it("V0 shipped k_tt_eval_wfirst<8,3,1>", [&] { hipLaunchKernelGGL((k_tt_eval_wfirst<8, 3, 1>), dim3((unsigned)blocks), dim3(256), ldsb, 0, dims, plan, d_img, d_pts, d_out, N); });
    }
    // ---- V1 / V2 images -----------------------------------------------------------------
    auto G = [&](int k, int a, int j, int b) -> double {
        if (a >= ranks[k] || b >= ranks[k + 1] || j >= n) return 0.0;
        return cores[rm.off[k] + ((long)a * n + j) * ranks[k + 1] + b];
    };
    for (int order = 0; order < 3; ++order) {
        const int TILES = R * R / 16, RA = R / 4, MID = D - 2, NP = 4 * KS;
        std::vector<double> img((size_t)KS * 64 + (size_t)MID * KS * TILES * 64 + R * NP, 0.0);
        for (int s = 0; s < KS; ++s) for (int l = 0; l < 64; ++l) img[s * 64 + l] = G(0, 0, 4 * s + (l >> 4), l & 15);
        for (int k = 0; k < MID; ++k) for (int s = 0; s < KS; ++s) for (int t = 0; t < TILES; ++t) for (int l = 0; l < 64; ++l) {
            const int rho = 16 * t + (l & 15), j = 4 * s + (l >> 4);
            int a, b;
            if (order == 0) { a = rho / R; b = rho % R; }
            else { const int u = rho >> 2, g = rho & 3; a = 4 * (u % RA) + g; b = u / RA; }   // orders 1, 2
            img[KS * 64 + ((size_t)(k * KS + s) * TILES + t) * 64 + l] = G(k + 1, a, j, b);
        }
        const size_t off_last = (size_t)KS * 64 + (size_t)MID * KS * TILES * 64;
        for (int a = 0; a < R; ++a) for (int j = 0; j < NP; ++j) img[off_last + a * NP + j] = G(D - 1, a, j, 0);
        double *d_img; CK(hipMalloc(&d_img, img.size() * 8)); CK(hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice));
        size_t ldsb = ((size_t)R * NP + 2 * 16 * D + 4 * 16 * D) * 8;
        for (int mult = 1; mult <= 4; mult *= 2) {
            char name[64];
            if (order == 0) {
                int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tt_wreg<8, 3, 5, 0>, 256, ldsb));
                long blocks = std::min<long>((N + 63) / 64, (long)per_cu * cus * mult);
                snprintf(name, sizeof name, "V1 regA+bpermute occ%d x%d", per_cu, mult);
                time_it(name, [&] { hipLaunchKernelGGL((k_tt_wreg<8, 3, 5, 0>), dim3((unsigned)blocks), dim3(256), ldsb, 0, dims, d_img, d_pts, d_out, N); });
            } else if (order == 2) {
                int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tt_wreg<8, 3, 5, 2>, 256, ldsb));
                long blocks = std::min<long>((N + 63) / 64, (long)per_cu * cus * mult);
                snprintf(name, sizeof name, "V2b laneswap+fma staging occ%d x%d", per_cu, mult);
                time_it(name, [&] { hipLaunchKernelGGL((k_tt_wreg<8, 3, 5, 2>), dim3((unsigned)blocks), dim3(256), ldsb, 0, dims, d_img, d_pts, d_out, N); });
            } else {
                int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tt_wreg<8, 3, 5, 1>, 256, ldsb));
                long blocks = std::min<long>((N + 63) / 64, (long)per_cu * cus * mult);
                snprintf(name, sizeof name, "V2 regA+laneswap occ%d x%d", per_cu, mult);
                time_it(name, [&] { hipLaunchKernelGGL((k_tt_wreg<8, 3, 5, 1>), dim3((unsigned)blocks), dim3(256), ldsb, 0, dims, d_img, d_pts, d_out, N); });
            }
        }
    }
    // ---- V3: 4x4x4_4b ------------------------------------------------------------------------
    for (int trim = 0; trim < 2; ++trim) {
        const int Rr = 8, NP = 4 * KS;
        TTW4Plan plan{}; long total = 0;
        for (int k = 0; k < D; ++k) {
            plan.rr[k] = trim ? ranks[k + 1] : Rr;
            plan.lds_off[k] = (int)total;
            total += (k == 0) ? (long)KS * 32 : (k == D - 1) ? (long)Rr * NP : (long)KS * Rr * 32;
        }
        plan.total = (int)total;
        std::vector<double> img(total, 0.0);
        // chunk-pair block: [slot = k4 * 4 + i][2] with k4 = node phase, i = row in chunk, pair = (ca 0, ca 1)
        for (int s = 0; s < KS; ++s) for (int k4 = 0; k4 < 4; ++k4) for (int i = 0; i < 4; ++i) for (int c = 0; c < 2; ++c)
            img[plan.lds_off[0] + s * 32 + (k4 * 4 + i) * 2 + c] = G(0, 0, 4 * s + k4, 4 * c + i);
        for (int k = 1; k < D - 1; ++k) for (int s = 0; s < KS; ++s) for (int b = 0; b < Rr; ++b)
            for (int k4 = 0; k4 < 4; ++k4) for (int i = 0; i < 4; ++i) for (int c = 0; c < 2; ++c)
                img[plan.lds_off[k] + (s * Rr + b) * 32 + (k4 * 4 + i) * 2 + c] = G(k, 4 * c + i, 4 * s + k4, b);
        for (int a = 0; a < Rr; ++a) for (int j = 0; j < NP; ++j) img[plan.lds_off[D - 1] + a * NP + j] = G(D - 1, a, j, 0);
        double *d_img; CK(hipMalloc(&d_img, img.size() * 8)); CK(hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice));
        size_t ldsb = ((size_t)total + 6 * 16 * D) * 8;
        auto run = [&](auto kern, const char *tag) {
            int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, ldsb));
            for (int mult = 1; mult <= 4; mult *= 2) {
                long blocks = std::min<long>((N + 63) / 64, (long)per_cu * cus * mult);
                char name[64]; snprintf(name, sizeof name, "V3 4x4x4 %s%s occ%d x%d", tag, trim ? " trim" : "", per_cu, mult);
                time_it(name, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), ldsb, 0, dims, plan, d_img, d_pts, d_out, N); });
            }
        };
        {
            size_t lds5 = ((size_t)total + 2 * 16 * D + 4 * (16 * D + 4 * 16 * 6)) * 8;
            auto run5 = [&](auto kern, const char *tag) {
                int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds5));
                for (int mult = 1; mult <= 4; mult *= 2) {
                    long blocks = std::min<long>((N + 63) / 64, (long)per_cu * cus * mult);
                    char name[64]; snprintf(name, sizeof name, "V5 seeds %s%s occ%d x%d", tag, trim ? " trim" : "", per_cu, mult);
                    time_it(name, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds5, 0, dims, plan, d_img, d_pts, d_out, N); });
                }
            };
            run5(k_tt_w5<3, 2>, "lb2");
            run5(k_tt_w5<3, 3>, "lb3");
            run5(k_tt_w5<3, 4>, "lb4");
        }
        if (!trim) {
            run(k_tt_w4s<3, 5, 0>, "straight s0");

        }
    }
    // ---- V6: direct form on 4x4x4_4b ------------------------------------------------------
    {
        const int Rr = 8, NP = 12;
        TTW4Plan plan{}; long total = 0;
        for (int k = 0; k < D; ++k) {
            plan.rr[k] = ranks[k + 1];
            plan.lds_off[k] = (int)total;
            total += (k == 0) ? (long)KS * 32 : (k == D - 1) ? (long)Rr * NP : (long)(n + 1) * 2 * 32;   // +1: the loop prefetches one j ahead
        }
        plan.total = (int)total;
        std::vector<double> img(total, 0.0);
        for (int s = 0; s < KS; ++s) for (int k4 = 0; k4 < 4; ++k4) for (int i = 0; i < 4; ++i) for (int m = 0; m < 2; ++m)
            img[plan.lds_off[0] + s * 32 + (k4 * 4 + i) * 2 + m] = G(0, 0, 4 * s + k4, 4 * m + i);
        for (int k = 1; k < D - 1; ++k) for (int j = 0; j < n; ++j) for (int c = 0; c < 2; ++c)
            for (int k4 = 0; k4 < 4; ++k4) for (int i = 0; i < 4; ++i) for (int m = 0; m < 2; ++m)
                img[plan.lds_off[k] + (j * 2 + c) * 32 + (k4 * 4 + i) * 2 + m] = G(k, 4 * c + k4, j, 4 * m + i);
        for (int a = 0; a < Rr; ++a) for (int j = 0; j < NP; ++j) img[plan.lds_off[D - 1] + a * NP + j] = G(D - 1, a, j, 0);
        double *d_img; CK(hipMalloc(&d_img, img.size() * 8)); CK(hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice));
        size_t ldsb = ((size_t)total + 2 * 16 * D + 4 * (16 * D + 16 * 6)) * 8;
        auto run6 = [&](auto kern, const char *tag) {
            int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, ldsb));
            for (int mult = 1; mult <= 4; mult *= 2) {
                long blocks = std::min<long>((N + 63) / 64, (long)per_cu * cus * mult);
                char name[64]; snprintf(name, sizeof name, "V6 direct 4x4x4 %s occ%d x%d", tag, per_cu, mult);
                time_it(name, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), ldsb, 0, dims, plan, d_img, d_pts, d_out, N, KS, NP); });
            }
        };
        run6(k_tt_d4<2, 11, 4>, "lb4");
        run6(k_tt_d4<2, 11, 6>, "lb6");
        run6(k_tt_d4<2, 11, 8>, "lb8");
        run6(k_tt_d4<2, 11, 4, 1>, "lb4 split");
        run6(k_tt_d4<2, 11, 6, 1>, "lb6 split");
    }
    // ---- V7: product-shaped direct 4x4x4 -------------------------------------------------
    {
        const int RA = 2, NMP = 2;
        TTD4Plan plan{}; long total = 0;
        const int ks0 = (n + 3) / 4;
        for (int k = 0; k < D; ++k) {
            plan.lds_off[k] = (int)total;
            total += (k == 0) ? (long)ks0 * 16 * NMP : (k == D - 1) ? (long)4 * RA * n : (long)n * RA * 16 * NMP;
        }
        plan.total = (int)total;
        std::vector<double> img(total, 0.0);
        for (int s = 0; s < ks0; ++s) for (int k4 = 0; k4 < 4; ++k4) for (int i = 0; i < 4; ++i) for (int m = 0; m < RA; ++m)
            img[plan.lds_off[0] + s * 16 * NMP + (k4 * 4 + i) * NMP + m] = G(0, 0, 4 * s + k4, 4 * m + i);
        for (int k = 1; k < D - 1; ++k) for (int j = 0; j < n; ++j) for (int c = 0; c < RA; ++c)
            for (int k4 = 0; k4 < 4; ++k4) for (int i = 0; i < 4; ++i) for (int m = 0; m < RA; ++m)
                img[plan.lds_off[k] + (j * RA + c) * 16 * NMP + (k4 * 4 + i) * NMP + m] = G(k, 4 * c + k4, j, 4 * m + i);
        for (int a = 0; a < 4 * RA; ++a) for (int j = 0; j < n; ++j) img[plan.lds_off[D - 1] + a * n + j] = G(D - 1, a, j, 0);
        double *d_img; CK(hipMalloc(&d_img, img.size() * 8)); CK(hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice));
        size_t ldsb = ((size_t)total + 2 * 16 * D + 4 * (16 * D + 16 * 6)) * 8;
        auto run7 = [&](auto kern, const char *tag) {
            int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, ldsb));
            for (int mult = 1; mult <= 4; mult *= 2) {
                long blocks = std::min<long>((N + 63) / 64, (long)per_cu * cus * mult);
                char name[64]; snprintf(name, sizeof name, "V7 product d4 %s occ%d x%d", tag, per_cu, mult);
                time_it(name, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), ldsb, 0, dims, plan, d_img, d_pts, d_out, N); });
            }
        };
        run7(k_tt_eval_d4<2, 4>, "lb4");
        run7(k_tt_eval_d4<2, 6>, "lb6");
    }
    return 0;
}
