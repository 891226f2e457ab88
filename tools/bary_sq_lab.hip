// Lab for the NL x NL block of k_bary_sq: the same kernel body with the block rebuilt under different
// scheduling constraints, timed on random data.  Not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Ipychebyshev_amd/csrc -o build_exp/bary_sq_lab tools/bary_sq_lab.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bary_kernels.h"

template <int NL, int V>
__device__ __forceinline__ double lab_block(pcx_cptr Tb, const double (&b1)[NL], const double (&b2)[NL]) {
    if constexpr (V == 0) return bary_sq_block<NL>(Tb, b1, b2);
    else if constexpr (V == 1 || V == 4) {          // barrier after each pass of four rows (V4: pass loop not unrolled)
        constexpr int R4 = 4;
        double t0 = 0.0, t1 = 0.0;
        if constexpr (V == 1) {
#pragma unroll
            for (int i = 0; i < NL; i += R4) {
                double s[R4];
#pragma unroll
                for (int r = 0; r < R4; ++r) s[r] = Tb[(i + r) * NL] * b2[0];
#pragma unroll
                for (int k = 1; k < NL; ++k)
#pragma unroll
                    for (int r = 0; r < R4; ++r) s[r] = __builtin_fma(Tb[(i + r) * NL + k], b2[k], s[r]);
                t0 = __builtin_fma(b1[i], s[0], t0); t1 = __builtin_fma(b1[i + 1], s[1], t1);
                t0 = __builtin_fma(b1[i + 2], s[2], t0); t1 = __builtin_fma(b1[i + 3], s[3], t1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // b1 cannot be indexed at run time (registers): the pass loop stays unrolled for b1 but the rows' sums are
            // produced by a non-inlined helper shape: two passes per iteration of a rolled loop over halves
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int ii = 0; ii < NL / 2; ii += R4) {
                    const int i = ii;
                    double s[R4];
                    pcx_cptr Th = Tb + (size_t)h * (NL / 2) * NL;
#pragma unroll
                    for (int r = 0; r < R4; ++r) s[r] = Th[(i + r) * NL] * b2[0];
#pragma unroll
                    for (int k = 1; k < NL; ++k)
#pragma unroll
                        for (int r = 0; r < R4; ++r) s[r] = __builtin_fma(Th[(i + r) * NL + k], b2[k], s[r]);
#pragma unroll
                    for (int r = 0; r < R4; ++r) {
                        const double w = h == 0 ? b1[i + r] : b1[NL / 2 + i + r];
                        if (r & 1) t1 = __builtin_fma(w, s[r], t1); else t0 = __builtin_fma(w, s[r], t0);
                    }
                }
            }
        }
        return t0 + t1;
    } else if constexpr (V == 2 || V == 3) {        // k in chunks of KC columns, barrier after each chunk of a pass
        constexpr int R4 = (V == 2) ? 4 : 8;
        constexpr int KC = (V == 2) ? 8 : 4;
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int i = 0; i < NL; i += R4) {
            double s[R4];
#pragma unroll
            for (int r = 0; r < R4; ++r) s[r] = 0.0;
#pragma unroll
            for (int k0 = 0; k0 < NL; k0 += KC) {
#pragma unroll
                for (int k = k0; k < k0 + KC; ++k)
#pragma unroll
                    for (int r = 0; r < R4; ++r) s[r] = __builtin_fma(Tb[(i + r) * NL + k], b2[k], s[r]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < R4; ++r) {
                if (r & 1) t1 = __builtin_fma(b1[i + r], s[r], t1); else t0 = __builtin_fma(b1[i + r], s[r], t0);
            }
        }
        return t0 + t1;
    } else {                                        // V5: eight rows per pass, no barriers
        constexpr int R4 = 8;
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int i = 0; i < NL; i += R4) {
            double s[R4];
#pragma unroll
            for (int r = 0; r < R4; ++r) s[r] = Tb[(i + r) * NL] * b2[0];
#pragma unroll
            for (int k = 1; k < NL; ++k)
#pragma unroll
                for (int r = 0; r < R4; ++r) s[r] = __builtin_fma(Tb[(i + r) * NL + k], b2[k], s[r]);
#pragma unroll
            for (int r = 0; r < R4; ++r) {
                if (r & 1) t1 = __builtin_fma(b1[i + r], s[r], t1); else t0 = __builtin_fma(b1[i + r], s[r], t0);
            }
        }
        return t0 + t1;
    }
}

template <int NL, int V, int MINB>
__global__ void __launch_bounds__(64, (MINB >= 100 ? MINB - 100 : MINB))
k_lab(BaryDims dims, BarySmallScale sc, const double *__restrict__ snodes, const double *__restrict__ nodes,
      const double *__restrict__ wts, const double *__restrict__ T, const double *__restrict__ pts, double *__restrict__ out, long N) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x;
    const long pidx = (long)blockIdx.x * 64 + lane;
    const bool valid = pidx < N;
    const long row = valid ? pidx : 0;
    pcx_cptr csn = pcx_as_constant(snodes), cnd = pcx_as_constant(nodes), cw = pcx_as_constant(wts);
    double *bw_lane = lds + lane;
    {
        const double x = valid ? pts[row * 3] : cnd[0];
        bary_weights_prod(x, sc.s[0], csn, cw, dims.n[0], bw_lane, 64);
    }
    double b1[NL], b2[NL];
    if constexpr (MINB >= 100) {        // no weight formation: what the blocks alone run at
        const double x1 = pts[row * 3 + 1], x2 = pts[row * 3 + 2];
#pragma unroll
        for (int j = 0; j < NL; ++j) { b1[j] = x1 + j; b2[j] = x2 - j; }
    } else {
        bary_weights_reg<NL, true>(valid ? pts[row * 3 + 1] : cnd[dims.off[1]], sc.s[1], csn + dims.off[1], cw + dims.off[1], NL, b1);
        bary_weights_reg<NL, true>(valid ? pts[row * 3 + 2] : cnd[dims.off[2]], sc.s[2], csn + dims.off[2], cw + dims.off[2], NL, b2);
    }
    pcx_cptr Tc = pcx_as_constant(T);
    double y = 0.0;
    for (int i0 = 0; i0 < dims.n[0]; ++i0, Tc += NL * NL)
        y = __builtin_fma(bw_lane[(size_t)i0 * 64], lab_block<NL, V>(Tc, b1, b2), y);
    if (valid) out[row] = y;
}

// WG waves per workgroup, a barrier in front of every block: the waves of a workgroup read the same 2..5 KB of the tensor
// at the same time, so one of them misses the scalar cache and the others hit
template <int NL, int V, int WG>
__global__ void __launch_bounds__(64 * WG)
k_lab_sync(BaryDims dims, BarySmallScale sc, const double *__restrict__ snodes, const double *__restrict__ nodes,
           const double *__restrict__ wts, const double *__restrict__ T, const double *__restrict__ pts, double *__restrict__ out, long N) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long pidx = (long)blockIdx.x * 64 * WG + threadIdx.x;
    const bool valid = pidx < N;
    const long row = valid ? pidx : 0;
    pcx_cptr csn = pcx_as_constant(snodes), cnd = pcx_as_constant(nodes), cw = pcx_as_constant(wts);
    double *bw_lane = lds + (size_t)wave * dims.n[0] * 64 + lane;
    {
        const double x = valid ? pts[row * 3] : cnd[0];
        bary_weights_prod(x, sc.s[0], csn, cw, dims.n[0], bw_lane, 64);
    }
    double b1[NL], b2[NL];
    bary_weights_reg<NL, true>(valid ? pts[row * 3 + 1] : cnd[dims.off[1]], sc.s[1], csn + dims.off[1], cw + dims.off[1], NL, b1);
    bary_weights_reg<NL, true>(valid ? pts[row * 3 + 2] : cnd[dims.off[2]], sc.s[2], csn + dims.off[2], cw + dims.off[2], NL, b2);
    pcx_cptr Tc = pcx_as_constant(T);
    double y = 0.0;
    for (int i0 = 0; i0 < dims.n[0]; ++i0, Tc += NL * NL) {
        __syncthreads();
        y = __builtin_fma(bw_lane[(size_t)i0 * 64], lab_block<NL, V>(Tc, b1, b2), y);
    }
    if (valid) out[row] = y;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NL, int V, int MINB, int WG = 0>
static void run(long N, const char *what, std::vector<double> &ref) {
    const int n0 = NL;
    BaryDims dims{};
    dims.d = 3; dims.sum_n = 3 * NL;
    for (int k = 0; k < 3; ++k) { dims.n[k] = NL; dims.off[k] = k * NL; }
    for (int k = 3; k < PCX_MAX_DIMS; ++k) { dims.n[k] = 1; dims.off[k] = 0; }
    std::vector<double> nodes(3 * NL), w(3 * NL), sn(3 * NL), T((size_t)NL * NL * NL + 64, 0.0), P((size_t)N * 3);
    BarySmallScale sc{};
    for (int k = 0; k < 3; ++k) {
        for (int j = 0; j < NL; ++j) nodes[k * NL + j] = std::sin(0.5 * M_PI / NL * (-NL + 1 + 2 * j));
        for (int i = 0; i < NL; ++i) { double wi = 1.0; for (int j = 0; j < NL; ++j) if (j != i) wi /= (nodes[k * NL + i] - nodes[k * NL + j]); w[k * NL + i] = wi; }
        sc.s[k] = 1.0;
        for (int j = 0; j < NL; ++j) sn[k * NL + j] = nodes[k * NL + j];
    }
    srand(1);
    for (size_t i = 0; i < (size_t)NL * NL * NL; ++i) T[i] = rand() / (double)RAND_MAX - 0.5;
    for (size_t i = 0; i < P.size(); ++i) P[i] = 2.0 * rand() / (double)RAND_MAX - 1.0;
    double *dn, *dw, *ds, *dT, *dP, *dO;
    CK(hipMalloc(&dn, nodes.size() * 8)); CK(hipMalloc(&dw, w.size() * 8)); CK(hipMalloc(&ds, sn.size() * 8));
    CK(hipMalloc(&dT, T.size() * 8)); CK(hipMalloc(&dP, P.size() * 8)); CK(hipMalloc(&dO, N * 8));
    CK(hipMemcpy(dn, nodes.data(), nodes.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(ds, sn.data(), sn.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dT, T.data(), T.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice));
    const size_t lds = (size_t)n0 * 64 * 8 * (WG ? WG : 1);
    const unsigned blocks = (unsigned)((N + 64 * (WG ? WG : 1) - 1) / (64 * (WG ? WG : 1)));
    auto launch = [&]() {
        if constexpr (WG == 0) hipLaunchKernelGGL((k_lab<NL, V, MINB>), dim3(blocks), dim3(64), lds, 0, dims, sc, ds, dn, dw, dT, dP, dO, N);
        else hipLaunchKernelGGL((k_lab_sync<NL, V, WG>), dim3(blocks), dim3(64 * WG), lds, 0, dims, sc, ds, dn, dw, dT, dP, dO, N);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    std::vector<double> O(N);
    CK(hipMemcpy(O.data(), dO, N * 8, hipMemcpyDeviceToHost));
    double dmax = 0.0;
    if (ref.empty()) ref = O; else for (long i = 0; i < N; ++i) dmax = std::fmax(dmax, std::fabs(O[i] - ref[i]));
    const double flop = 2.0 * ((double)NL * NL * NL + (double)NL * NL + NL) * N;
    printf("NL=%2d %-34s wg %d minb %d  %8.3f ms  %6.2f TFLOP/s  frac %.3f  max|diff to V0| %.1e\n", NL, what, WG, MINB, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 78.6, dmax);
    for (double *p : {dn, dw, ds, dT, dP, dO}) CK(hipFree(p));
}

int main(int argc, char **argv) {
    const long N = argc > 1 ? atol(argv[1]) : 2000000;
    { std::vector<double> ref;
      run<16, 0, 1>(N, "V0 shipped block", ref); run<16, 1, 1>(N, "V1 barrier per pass", ref);
      { std::vector<double> r2; run<16, 1, 101>(N, "V1, no weight formation", r2); run<16, 5, 101>(N, "V5, no weight formation", r2); run<16, 1, 104>(N, "V1, no weight formation", r2); }
      { std::vector<double> r2; run<24, 1, 101>(N, "V1, no weight formation", r2); run<11, 0, 101>(N, "V0, no weight formation", r2); }
      run<16, 1, 1, 2>(N, "V1 + workgroup barrier per block", ref); run<16, 1, 1, 4>(N, "V1 + workgroup barrier per block", ref);
      run<16, 1, 1, 8>(N, "V1 + workgroup barrier per block", ref); run<16, 1, 1, 16>(N, "V1 + workgroup barrier per block", ref);
      run<16, 0, 1, 4>(N, "V0 + workgroup barrier per block", ref); run<16, 5, 1, 4>(N, "V5 + workgroup barrier per block", ref); }
    { std::vector<double> ref;
      run<24, 0, 1>(N, "V0 shipped block", ref); run<24, 1, 1>(N, "V1 barrier per pass", ref);
      run<24, 1, 1, 2>(N, "V1 + workgroup barrier per block", ref); run<24, 1, 1, 4>(N, "V1 + workgroup barrier per block", ref);
      run<24, 1, 1, 8>(N, "V1 + workgroup barrier per block", ref); run<24, 0, 1, 4>(N, "V0 + workgroup barrier per block", ref); }
    { std::vector<double> ref;
      run<11, 0, 1>(N, "V0 shipped block", ref); run<11, 0, 1, 4>(N, "V0 + workgroup barrier per block", ref); run<11, 0, 1, 8>(N, "V0 + workgroup barrier per block", ref); }
    { std::vector<double> ref;
      run<32, 0, 1>(N / 2, "V0 shipped block", ref); run<32, 1, 1, 4>(N / 2, "V1 + workgroup barrier per block", ref); run<32, 0, 1, 4>(N / 2, "V0 + workgroup barrier per block", ref); }
    return 0;
}
