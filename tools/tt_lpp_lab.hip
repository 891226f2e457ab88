// tt_lpp_lab.hip -- round-3 bench of a VALU-only, lane-per-point form of the small-rank TT
// evaluation kernel (5-D, ranks [1,8,8,8,6,1], n = 11: BASELINE config 3) against the shipped
// k_tt_eval_d4<2>.  Every variant is checked against a plain one-thread-per-point chain first.
//
// Why: on gfx950 the FP64 MFMA occupies the SIMD's vector pipe (profiles/r02_fp64_mix_microbench.txt),
// v_fma_f64 sustains the same flop rate, and the MFMA forms carry ~180 vector instructions per 16 points
// next to the matrix work.  With lane = point a vector instruction serves 64 points, and a core element is
// wave-uniform: it is read with scalar loads and enters v_fma_f64 as an SGPR operand.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pychebyshev_amd/csrc tools/tt_lpp_lab.hip -o /tmp/tt_lpp_lab
//   /tmp/tt_lpp_lab [points]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "bary_kernels.h"
#include "tt_kernels.h"
#include "tt_lpp_kernels.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

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

// ---- lane-per-point form ------------------------------------------------------------------
// image of storage dim k: img[off[k] + ((b * RLk + a) * NJ + j)] = G_k[a][j][b], RLk = 1 for k = 0 and R after
// (left ranks zero-padded to R), b < r_{k+1} exactly.
struct LppPlan {
    int d;
    int rr[PCX_MAX_DIMS];       // right rank of storage dim k
    int off[PCX_MAX_DIMS];      // offset (doubles) of dim k in the image
    int col[PCX_MAX_DIMS];
    double lo[PCX_MAX_DIMS], scale[PCX_MAX_DIMS];
};

// vn[b] = sum_a v[a] * (sum_j T_j G[a][j][b]) for b < rr; NP points per lane share every scalar operand
template <int RL, int R, int NJ, int NP, bool ONE = false, bool NOLOAD = false>
__device__ __forceinline__ void lpp_dim(pcx_cptr G, int rr, const double (&T)[NP][NJ], const double (&v)[NP][R],
                                        double (&vn)[NP][R]) {
    double gl[16];
    if constexpr (NOLOAD) {
#pragma unroll
        for (int i = 0; i < 16; ++i) gl[i] = G[i];
    }
    double one = 1.0;
    if constexpr (ONE) asm volatile("" : "+v"(one));      // T_0 as a register: one v_mul_f64 instead of two v_mov_b32
#pragma unroll
    for (int b = 0; b < R; ++b) {
#pragma unroll
        for (int q = 0; q < NP; ++q) vn[q][b] = 0.0;
        if (b < rr) {                      // wave-uniform
            double M[NP][RL];
#pragma unroll
            for (int a = 0; a < RL; ++a) {
                const double g0 = NOLOAD ? gl[(a * NJ) & 15] : G[(b * RL + a) * NJ];
#pragma unroll
                for (int q = 0; q < NP; ++q) M[q][a] = ONE ? g0 * one : g0;        // T_0 = 1
            }
#pragma unroll
            for (int j = 1; j < NJ; ++j)
#pragma unroll
                for (int a = 0; a < RL; ++a) {
                    const double g = NOLOAD ? gl[(a * NJ + j) & 15] : G[(b * RL + a) * NJ + j];
#pragma unroll
                    for (int q = 0; q < NP; ++q) M[q][a] = __builtin_fma(T[q][j], g, M[q][a]);
                }
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                if constexpr (RL == 1) vn[q][b] = M[q][0];        // left rank 1: v = [1]
                else {
                    double s0 = v[q][0] * M[q][0], s1 = v[q][1] * M[q][1];
#pragma unroll
                    for (int a = 2; a < RL; a += 2) {
                        s0 = __builtin_fma(v[q][a], M[q][a], s0);
                        if (a + 1 < RL) s1 = __builtin_fma(v[q][a + 1], M[q][a + 1], s1);
                    }
                    vn[q][b] = s0 + s1;
                }
            }
        }
    }
}

template <int R, int NJ, int NP, int WPS>
__global__ void __launch_bounds__(256, WPS)
k_tt_lpp(LppPlan plan, const double *__restrict__ img, const double *__restrict__ pts, double *__restrict__ out, long N) {
    const pcx_cptr cimg = pcx_as_constant(img);
    const int d = plan.d;
    const long p0 = ((long)blockIdx.x * 256 + threadIdx.x) * NP;      // NP consecutive points per lane
    double v[NP][R], vn[NP][R], T[NP][NJ];
    auto cheb = [&](int k) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const long p = (p0 + q < N) ? p0 + q : N - 1;
            const double x = pts[p * d + plan.col[k]];
            const double s = __builtin_fma(x - plan.lo[k], plan.scale[k], -1.0);
            const double s2 = s + s;
            T[q][0] = 1.0;
            if constexpr (NJ > 1) T[q][1] = s;
#pragma unroll
            for (int j = 2; j < NJ; ++j) T[q][j] = __builtin_fma(s2, T[q][j - 1], -T[q][j - 2]);
        }
    };
    cheb(0);
    lpp_dim<1, R, NJ, NP>(cimg + plan.off[0], plan.rr[0], T, v, vn);
#pragma unroll
    for (int q = 0; q < NP; ++q)
#pragma unroll
        for (int b = 0; b < R; ++b) v[q][b] = vn[q][b];
    for (int k = 1; k < d; ++k) {
        cheb(k);
        lpp_dim<R, R, NJ, NP>(cimg + plan.off[k], plan.rr[k], T, v, vn);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int b = 0; b < R; ++b) v[q][b] = vn[q][b];
    }
#pragma unroll
    for (int q = 0; q < NP; ++q)
        if (p0 + q < N) out[p0 + q] = v[q][0];
}

// V2: workgroups of WG threads, optionally persistent (grid-stride over batches of WG * NP points) and in lock
// step (a barrier per dimension keeps every wave of the workgroup on the same part of the core image, so a
// line of the scalar cache fetched by one wave serves the others); the coordinate of the NEXT dimension (or of
// the next batch's first) is loaded while the current dimension is contracted.
template <int R, int NJ, int NP, int WG, int MINB, bool SYNC, bool ONE, bool NOLOAD = false>
__global__ void __launch_bounds__(WG, MINB)
k_tt_lpp2(LppPlan plan, const double *__restrict__ img, const double *__restrict__ pts, double *__restrict__ out, long N,
          long nbatch) {
    const pcx_cptr cimg = pcx_as_constant(img);
    const int d = plan.d;
    double xn[NP];
    auto fetch = [&](long batch, int k) {
        const long p0 = (batch * WG + threadIdx.x) * NP;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const long p = (p0 + q < N) ? p0 + q : N - 1;
            xn[q] = pts[p * d + plan.col[k]];
        }
    };
    fetch(blockIdx.x, 0);
    for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const long p0 = (batch * WG + threadIdx.x) * NP;
        double v[NP][R], vn[NP][R], T[NP][NJ];
        auto cheb = [&](int k) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const double s = __builtin_fma(xn[q] - plan.lo[k], plan.scale[k], -1.0);
                const double s2 = s + s;
                T[q][0] = 1.0;
                if constexpr (NJ > 1) T[q][1] = s;
#pragma unroll
                for (int j = 2; j < NJ; ++j) T[q][j] = __builtin_fma(s2, T[q][j - 1], -T[q][j - 2]);
            }
        };
        if (SYNC) __syncthreads();
        cheb(0);
        if (d > 1) fetch(batch, 1);
        lpp_dim<1, R, NJ, NP, ONE, NOLOAD>(cimg + plan.off[0], plan.rr[0], T, v, vn);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int b = 0; b < R; ++b) v[q][b] = vn[q][b];
        for (int k = 1; k < d; ++k) {
            if (SYNC) __syncthreads();
            cheb(k);
            if (k + 1 < d) fetch(batch, k + 1);
            else {
                const long nb = batch + gridDim.x;
                fetch(nb < nbatch ? nb : batch, 0);
            }
            lpp_dim<R, R, NJ, NP, ONE, NOLOAD>(cimg + plan.off[k], plan.rr[k], T, v, vn);
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int b = 0; b < R; ++b) v[q][b] = vn[q][b];
        }
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (p0 + q < N) out[p0 + q] = v[q][0];
    }
}

// V3: the product's shape -- right-rank loop at run time (code size RL * NJ, no guards, exact left rank: the image
// is [b][a < rl][j < n] with no padding at all), v' through a lane-private LDS column (one ds_write per b, RL
// ds_reads per dimension), bodies picked per dimension by a switch over the left rank.
struct Lpp3Plan {
    int d;
    int rank[PCX_MAX_DIMS + 1];
    int off[PCX_MAX_DIMS];
    int col[PCX_MAX_DIMS];
    double lo[PCX_MAX_DIMS], scale[PCX_MAX_DIMS];
};

template <int RL, int NJ, int WG>
__device__ __forceinline__ void lpp3_body(pcx_cptr G, int rr, double x, double *vl) {
    double T[NJ], v[RL];
    const double x2 = x + x;
    T[0] = 1.0;
    asm volatile("" : "+v"(T[0]));          // T_0 as a register: M = g * T_0 is one v_mul_f64, not two v_mov_b32
    if constexpr (NJ > 1) T[1] = x;
#pragma unroll
    for (int j = 2; j < NJ; ++j) T[j] = __builtin_fma(x2, T[j - 1], -T[j - 2]);
    if constexpr (RL > 1) {
#pragma unroll
        for (int a = 0; a < RL; ++a) v[a] = vl[a * WG];
    }
    for (int b = 0; b < rr; ++b, G += RL * NJ) {
        double M[RL];
#pragma unroll
        for (int a = 0; a < RL; ++a) M[a] = G[a * NJ] * T[0];
#pragma unroll
        for (int j = 1; j < NJ; ++j)
#pragma unroll
            for (int a = 0; a < RL; ++a) M[a] = __builtin_fma(T[j], G[a * NJ + j], M[a]);
        double s;
        if constexpr (RL == 1) s = M[0];
        else if constexpr (RL == 2) s = __builtin_fma(v[1], M[1], v[0] * M[0]);
        else {
            double s0 = v[0] * M[0], s1 = v[1] * M[1];
#pragma unroll
            for (int a = 2; a < RL; a += 2) {
                s0 = __builtin_fma(v[a], M[a], s0);
                if (a + 1 < RL) s1 = __builtin_fma(v[a + 1], M[a + 1], s1);
            }
            s = s0 + s1;
        }
        vl[b * WG] = s;
    }
}

template <int NJ, int WG, int MINB>
__global__ void __launch_bounds__(WG, MINB)
k_tt_lpp3(Lpp3Plan plan, const double *__restrict__ img, const double *__restrict__ pts, double *__restrict__ out, long N) {
    extern __shared__ double lds[];
    double *vl = lds + threadIdx.x;
    const pcx_cptr cimg = pcx_as_constant(img);
    const int d = plan.d;
    const long p = (long)blockIdx.x * WG + threadIdx.x;
    const long pc = p < N ? p : N - 1;
    double xn = pts[pc * d + plan.col[0]];
    for (int k = 0; k < d; ++k) {
        const double x = __builtin_fma(xn - plan.lo[k], plan.scale[k], -1.0);
        if (k + 1 < d) xn = pts[pc * d + plan.col[k + 1]];
        const pcx_cptr G = cimg + plan.off[k];
        const int rr = plan.rank[k + 1];
        switch (plan.rank[k]) {
        case 1: lpp3_body<1, NJ, WG>(G, rr, x, vl); break;
        case 2: lpp3_body<2, NJ, WG>(G, rr, x, vl); break;
        case 3: lpp3_body<3, NJ, WG>(G, rr, x, vl); break;
        case 4: lpp3_body<4, NJ, WG>(G, rr, x, vl); break;
        case 5: lpp3_body<5, NJ, WG>(G, rr, x, vl); break;
        case 6: lpp3_body<6, NJ, WG>(G, rr, x, vl); break;
        case 7: lpp3_body<7, NJ, WG>(G, rr, x, vl); break;
        default: lpp3_body<8, NJ, WG>(G, rr, x, vl); break;
        }
    }
    if (p < N) out[p] = vl[0];
}

// V5: the product's shape with NP points per lane (each scalar operand feeds NP FMAs; per-dimension overhead per 64 NP points)
template <int RL, int NJ, int NP>
__device__ __forceinline__ void lpp5_body(pcx_lpp_cptr G, int rr, const double (&x)[NP], double *vl) {
    double T[NP][NJ], v[NP][RL];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const double x2 = x[q] + x[q];
        T[q][0] = 1.0;
        asm volatile("" : "+v"(T[q][0]));
        if constexpr (NJ > 1) T[q][1] = x[q];
#pragma unroll
        for (int j = 2; j < NJ; ++j) T[q][j] = __builtin_fma(x2, T[q][j - 1], -T[q][j - 2]);
#pragma unroll
        for (int a = 0; a < RL; ++a) v[q][a] = vl[(a * NP + q) * PCX_LPP_WG];
    }
    for (int b = 0; b < rr; ++b, G += RL * NJ) {
        double M[NP][RL];
#pragma unroll
        for (int a = 0; a < RL; ++a) {
            const double g = G[a * NJ];
#pragma unroll
            for (int q = 0; q < NP; ++q) M[q][a] = g * T[q][0];
        }
#pragma unroll
        for (int j = 1; j < NJ; ++j)
#pragma unroll
            for (int a = 0; a < RL; ++a) {
                const double g = G[a * NJ + j];
#pragma unroll
                for (int q = 0; q < NP; ++q) M[q][a] = __builtin_fma(T[q][j], g, M[q][a]);
            }
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            double s0 = v[q][0] * M[q][0];
#pragma unroll
            for (int a = 1; a < RL; ++a) s0 = __builtin_fma(v[q][a], M[q][a], s0);
            vl[(b * NP + q) * PCX_LPP_WG] = s0;
        }
    }
}

template <int NJ, int NP, int MINB>
__global__ void __launch_bounds__(PCX_LPP_WG, MINB)
k_tt_lpp5(const TTLppDim *__restrict__ tab, int d, const double *__restrict__ img, const double *__restrict__ pts,
          double *__restrict__ out, long N) {
    extern __shared__ double lds_lpp[];
    double *vl = lds_lpp + threadIdx.x;
    typedef const TTLppDim __attribute__((address_space(4))) *tab_cptr;
    const tab_cptr ct = (tab_cptr)(unsigned long long)tab;
    const pcx_lpp_cptr cimg = (pcx_lpp_cptr)(unsigned long long)img;
    long pc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const long p = ((long)blockIdx.x * NP + q) * PCX_LPP_WG + threadIdx.x;      // coalesced output: point q of lane l
        pc[q] = p < N ? p : N - 1;
        vl[q * PCX_LPP_WG] = 1.0;
    }
    double xn[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) xn[q] = pts[pc[q] * d + ct[0].col];
    for (int k = 0; k < d; ++k) {
        double x[NP];
#pragma unroll
        for (int q = 0; q < NP; ++q) x[q] = __builtin_fma(xn[q] - ct[k].lo, ct[k].scale, -1.0);
        if (k + 1 < d) {
#pragma unroll
            for (int q = 0; q < NP; ++q) xn[q] = pts[pc[q] * d + ct[k + 1].col];
        }
        const pcx_lpp_cptr G = cimg + ct[k].off;
        const int rl = ct[k].rl, rr = ct[k].rr;
        switch (rl) {
        case 1: lpp5_body<1, NJ, NP>(G, rr, x, vl); break;
        case 2: lpp5_body<2, NJ, NP>(G, rr, x, vl); break;
        case 3: lpp5_body<3, NJ, NP>(G, rr, x, vl); break;
        case 4: lpp5_body<4, NJ, NP>(G, rr, x, vl); break;
        case 5: lpp5_body<5, NJ, NP>(G, rr, x, vl); break;
        case 6: lpp5_body<6, NJ, NP>(G, rr, x, vl); break;
        case 7: lpp5_body<7, NJ, NP>(G, rr, x, vl); break;
        case 8: lpp5_body<8, NJ, NP>(G, rr, x, vl); break;
        default: break;
        }
    }
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const long p = ((long)blockIdx.x * NP + q) * PCX_LPP_WG + threadIdx.x;
        if (p < N) out[p] = vl[q * PCX_LPP_WG];
    }
}

int main(int argc, char **argv) {
    const long N = argc > 1 ? atol(argv[1]) : 10000000L;
    const int D = 5, n = 11;
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
    for (int k = 0; k < D; ++k) { dims.n[k] = n; dims.col[k] = k; dims.lo[k] = rm.lo[k]; dims.hi[k] = rm.hi[k]; dims.scale[k] = 2.0 / (rm.hi[k] - rm.lo[k]); }
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
        fflush(stdout);
    };
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;

    // ---- V0: shipped k_tt_eval_d4<2> -----------------------------------------------------------
    {
        const int RA = 2, NMP = 2;
        TTD4Plan plan{}; long total = 0;
        std::vector<double> img;
        for (int k = 0; k < D; ++k) {
            plan.lds_off[k] = (int)img.size();
            const double *G = cores.data() + rm.off[k];
            const int rl = ranks[k], rr = ranks[k + 1];
            if (k == 0) {
                const int ks0 = (n + 3) / 4;
                for (int s = 0; s < ks0; ++s) for (int slot = 0; slot < 16; ++slot) for (int m = 0; m < NMP; ++m) {
                    const int kk = slot >> 2, i = slot & 3, j = 4 * s + kk, b = 4 * m + i;
                    img.push_back((j < n && b < rr && m < RA) ? G[(long)j * rr + b] : 0.0);
                }
            } else if (k < D - 1) {
                for (int j = 0; j < n; ++j) for (int c = 0; c < RA; ++c) for (int slot = 0; slot < 16; ++slot) for (int m = 0; m < NMP; ++m) {
                    const int kk = slot >> 2, i = slot & 3, a = 4 * c + kk, b = 4 * m + i;
                    img.push_back((a < rl && b < rr && m < RA) ? G[((long)a * n + j) * rr + b] : 0.0);
                }
            } else {
                for (int a = 0; a < 4 * RA; ++a) for (int j = 0; j < n; ++j) img.push_back(a < rl ? G[(long)a * n + j] : 0.0);
            }
        }
        total = (long)img.size(); plan.total = (int)total;
        double *d_img; CK(hipMalloc(&d_img, total * 8)); CK(hipMemcpy(d_img, img.data(), total * 8, hipMemcpyHostToDevice));
        size_t ldsb = ((size_t)total + 2 * 16 * D + 4 * (16 * D + 16 * 6)) * 8;
        int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tt_eval_d4<2>, 256, ldsb));
        long blocks = std::min<long>((N + 63) / 64, (long)per_cu * cus * 4);
        printf("V0 occupancy %d WG/CU, %ld blocks\n", per_cu, blocks);
        time_it("V0 shipped k_tt_eval_d4<2>", [&] { hipLaunchKernelGGL((k_tt_eval_d4<2>), dim3((unsigned)blocks), dim3(256), ldsb, 0, dims, plan, d_img, d_pts, d_out, N); });
    }

    // ---- V1: lane per point, SGPR core operands ------------------------------------------------
    {
        const int R = 8;
        LppPlan plan{}; plan.d = D;
        std::vector<double> img;
        for (int k = 0; k < D; ++k) {
            plan.off[k] = (int)img.size(); plan.rr[k] = ranks[k + 1]; plan.col[k] = k; plan.lo[k] = rm.lo[k]; plan.scale[k] = 2.0 / (rm.hi[k] - rm.lo[k]);
            const double *G = cores.data() + rm.off[k];
            const int rl = ranks[k], rr = ranks[k + 1], RL = k == 0 ? 1 : R;
            for (int b = 0; b < rr; ++b) for (int a = 0; a < RL; ++a) for (int j = 0; j < n; ++j)
                img.push_back(a < rl ? G[((long)a * n + j) * rr + b] : 0.0);
        }
        for (int i = 0; i < 64; ++i) img.push_back(0.0);
        double *d_img; CK(hipMalloc(&d_img, img.size() * 8)); CK(hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice));
#define RUN_LPP(NP, WPS)                                                                                              \
        {                                                                                                             \
            int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_tt_lpp<8, 11, NP, WPS>, 256, 0)); \
            const long blocks = (N + 256 * NP - 1) / (256 * NP);                                                      \
            char name[64]; snprintf(name, sizeof name, "V1 lpp NP=%d WPS=%d occ%d", NP, WPS, per_cu);                 \
            time_it(name, [&] { hipLaunchKernelGGL((k_tt_lpp<8, 11, NP, WPS>), dim3((unsigned)blocks), dim3(256), 0, 0, plan, d_img, d_pts, d_out, N); }); \
        }
        RUN_LPP(1, 6)
        RUN_LPP(2, 3)
#define RUN_LPP2(NP, WG, MINB, SYNC, ONE, MULT)                                                                       \
        {                                                                                                             \
            auto kern = k_tt_lpp2<8, 11, NP, WG, MINB, SYNC, ONE>;                                                    \
            int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, WG, 0));                   \
            const long nbatch = (N + (long)WG * NP - 1) / ((long)WG * NP);                                            \
            const long blocks = MULT > 0 ? std::min<long>(nbatch, (long)per_cu * cus * MULT) : nbatch;                \
            char name[96]; snprintf(name, sizeof name, "V2 NP=%d WG=%d sync%d one%d occ%d x%d", NP, WG, (int)SYNC, (int)ONE, per_cu, MULT); \
            time_it(name, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WG), 0, 0, plan, d_img, d_pts, d_out, N, nbatch); }); \
        }
        RUN_LPP2(1, 256, 6, false, true, 0)
        RUN_LPP2(2, 256, 3, false, true, 0)
#define RUN_NOLOAD(NP, WG, MINB)                                                                                      \
        {                                                                                                             \
            auto kern = k_tt_lpp2<8, 11, NP, WG, MINB, false, true, true>;                                            \
            const long nbatch = (N + (long)WG * NP - 1) / ((long)WG * NP);                                            \
            char name[96]; snprintf(name, sizeof name, "V2 NOLOAD (wrong values) NP=%d MINB=%d", NP, MINB);           \
            time_it(name, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)nbatch), dim3(WG), 0, 0, plan, d_img, d_pts, d_out, N, nbatch); }); \
        }
        RUN_NOLOAD(1, 256, 6)
        RUN_NOLOAD(1, 256, 4)
        RUN_NOLOAD(2, 256, 3)
    }
    // ---- V3: runtime right-rank loop, exact ranks ----------------------------------------------
    {
        Lpp3Plan plan{}; plan.d = D;
        std::vector<double> img;
        for (int k = 0; k < D; ++k) {
            plan.off[k] = (int)img.size(); plan.rank[k] = ranks[k]; plan.col[k] = k; plan.lo[k] = rm.lo[k]; plan.scale[k] = 2.0 / (rm.hi[k] - rm.lo[k]);
            const double *G = cores.data() + rm.off[k];
            const int rl = ranks[k], rr = ranks[k + 1];
            for (int b = 0; b < rr; ++b) for (int a = 0; a < rl; ++a) for (int j = 0; j < n; ++j) img.push_back(G[((long)a * n + j) * rr + b]);
        }
        plan.rank[D] = 1;
        for (int i = 0; i < 64; ++i) img.push_back(0.0);
        double *d_img; CK(hipMalloc(&d_img, img.size() * 8)); CK(hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice));
#define RUN_LPP3(WG, MINB)                                                                                            \
        {                                                                                                             \
            auto kern = k_tt_lpp3<11, WG, MINB>;                                                                      \
            const size_t ldsb = (size_t)8 * WG * 8;                                                                   \
            int per_cu = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, WG, ldsb));                \
            const long blocks = (N + WG - 1) / WG;                                                                    \
            char name[96]; snprintf(name, sizeof name, "V3 runtime-b WG=%d MINB=%d occ%d", WG, MINB, per_cu);         \
            time_it(name, [&] { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(WG), ldsb, 0, plan, d_img, d_pts, d_out, N); }); \
        }
        RUN_LPP3(256, 4)
        RUN_LPP3(256, 5)
        RUN_LPP3(256, 6)
        RUN_LPP3(256, 8)
        RUN_LPP3(64, 8)
    }
    // ---- V4: the product kernel (tt_lpp_kernels.h) -------------------------------------------------
    {
        std::vector<TTLppDim> tab(D);
        std::vector<double> img;
        for (int k = 0; k < D; ++k) {
            tab[k] = TTLppDim{(int)img.size(), ranks[k], ranks[k + 1], n, k, 0, rm.lo[k], 2.0 / (rm.hi[k] - rm.lo[k])};
            const double *G = cores.data() + rm.off[k];
            const int rl = ranks[k], rr = ranks[k + 1];
            for (int b = 0; b < rr; ++b) for (int a = 0; a < rl; ++a) for (int j = 0; j < n; ++j) img.push_back(G[((long)a * n + j) * rr + b]);
        }
        for (int i = 0; i < 64; ++i) img.push_back(0.0);
        double *d_img; CK(hipMalloc(&d_img, img.size() * 8)); CK(hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice));
        TTLppDim *d_tab; CK(hipMalloc(&d_tab, sizeof(TTLppDim) * D)); CK(hipMemcpy(d_tab, tab.data(), sizeof(TTLppDim) * D, hipMemcpyHostToDevice));
        const size_t ldsb = (size_t)8 * PCX_LPP_WG * 8;
        const long blocks = (N + PCX_LPP_WG - 1) / PCX_LPP_WG;
        time_it("V4 product k_tt_eval_lpp<8,11>", [&] { hipLaunchKernelGGL((k_tt_eval_lpp<8, 11>), dim3((unsigned)blocks), dim3(PCX_LPP_WG), ldsb, 0, d_tab, D, d_img, d_pts, d_out, N); });
        time_it("V4 product k_tt_eval_lpp<16,11>", [&] { hipLaunchKernelGGL((k_tt_eval_lpp<16, 11>), dim3((unsigned)blocks), dim3(PCX_LPP_WG), ldsb, 0, d_tab, D, d_img, d_pts, d_out, N); });
#define RUN_LPP5(NP, MINB)                                                                                            \
        {                                                                                                             \
            const size_t l5 = (size_t)8 * NP * PCX_LPP_WG * 8;                                                        \
            const long b5 = (N + (long)PCX_LPP_WG * NP - 1) / ((long)PCX_LPP_WG * NP);                                \
            char name[96]; snprintf(name, sizeof name, "V5 product shape NP=%d MINB=%d", NP, MINB);                   \
            time_it(name, [&] { hipLaunchKernelGGL((k_tt_lpp5<11, NP, MINB>), dim3((unsigned)b5), dim3(PCX_LPP_WG), l5, 0, d_tab, D, d_img, d_pts, d_out, N); }); \
        }
        RUN_LPP5(1, 8)
        RUN_LPP5(2, 4)
        RUN_LPP5(2, 3)
        RUN_LPP5(3, 2)
        time_it("V4 product k_tt_eval_lpp<8,0>", [&] { hipLaunchKernelGGL((k_tt_eval_lpp<8, 0>), dim3((unsigned)blocks), dim3(PCX_LPP_WG), ldsb, 0, d_tab, D, d_img, d_pts, d_out, N); });
        time_it("V4 product k_tt_eval_lpp<16,0>", [&] { hipLaunchKernelGGL((k_tt_eval_lpp<16, 0>), dim3((unsigned)blocks), dim3(PCX_LPP_WG), ldsb, 0, d_tab, D, d_img, d_pts, d_out, N); });
    }
    // ---- V6: ranks 9..15 (the reference's max_rank = 15 Black-Scholes model has ranks [1,11,11,11,7,1]) ----------
    {
        const int ranks2[6] = {1, 11, 11, 11, 7, 1};
        std::vector<double> cores2; std::vector<long> off2(D);
        RefModel rm2 = rm;
        for (int k = 0; k < D; ++k) {
            rm2.r[k] = ranks2[k]; rm2.off[k] = (long)cores2.size(); off2[k] = rm2.off[k];
            for (int i = 0; i < ranks2[k] * n * ranks2[k + 1]; ++i) cores2.push_back(nd(rng) / std::sqrt((double)ranks2[k] * n));
        }
        rm2.r[D] = 1;
        double *d_cores2; CK(hipMalloc(&d_cores2, cores2.size() * 8)); CK(hipMemcpy(d_cores2, cores2.data(), cores2.size() * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_ref, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, rm2, d_cores2, d_pts, d_ref, N);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), d_ref, N * 8, hipMemcpyDeviceToHost));
        scale = 0; for (double v : ref) scale = std::max(scale, std::fabs(v));
        std::vector<TTLppDim> tab(D);
        std::vector<double> img;
        double fma = 0;
        for (int k = 0; k < D; ++k) {
            tab[k] = TTLppDim{(int)img.size(), ranks2[k], ranks2[k + 1], n, k, 0, rm.lo[k], 2.0 / (rm.hi[k] - rm.lo[k])};
            const double *G = cores2.data() + off2[k];
            const int rl = ranks2[k], rr = ranks2[k + 1];
            fma += (double)(n + 1) * rl * rr;
            for (int b = 0; b < rr; ++b) for (int a = 0; a < rl; ++a) for (int j2 = 0; j2 < n; ++j2) img.push_back(G[((long)a * n + j2) * rr + b]);
        }
        for (int i = 0; i < 64; ++i) img.push_back(0.0);
        double *d_img; CK(hipMalloc(&d_img, img.size() * 8)); CK(hipMemcpy(d_img, img.data(), img.size() * 8, hipMemcpyHostToDevice));
        TTLppDim *d_tab; CK(hipMalloc(&d_tab, sizeof(TTLppDim) * D)); CK(hipMemcpy(d_tab, tab.data(), sizeof(TTLppDim) * D, hipMemcpyHostToDevice));
        const size_t ldsb = (size_t)12 * PCX_LPP_WG * 8;
        const long blocks = (N + PCX_LPP_WG - 1) / PCX_LPP_WG;
        printf("ranks [1,11,11,11,7,1]: %.0f FMA per point; the fractions printed below assume 2,280 -- multiply by %.3f\n", fma, fma / 2280.0);
        time_it("V6 r15 k_tt_eval_lpp<16,11>", [&] { hipLaunchKernelGGL((k_tt_eval_lpp<16, 11>), dim3((unsigned)blocks), dim3(PCX_LPP_WG), ldsb, 0, d_tab, D, d_img, d_pts, d_out, N); });
        time_it("V6 r15 k_tt_eval_lpp<12,11>", [&] { hipLaunchKernelGGL((k_tt_eval_lpp<12, 11>), dim3((unsigned)blocks), dim3(PCX_LPP_WG), ldsb, 0, d_tab, D, d_img, d_pts, d_out, N); });
    }
    return 0;
}
