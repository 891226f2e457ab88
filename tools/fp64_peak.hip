// fp64_peak.hip -- microbenchmark: what FP64 rate does one MI355X actually sustain?
//   mode 0: v_mfma_f64_16x16x4_f64 back to back (4 independent accumulators per wave)
//   mode 1: v_fma_f64 back to back (16 independent chains per lane)
//   mode 2: both in the same wave (do the matrix and vector FP64 pipes co-execute?)
//   mode 3: half the waves MFMA-only, half VALU-only (co-execution across waves)
// Build: hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip -o gpurun_out/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, double seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){seed, seed * 2, seed * 3, seed * 4};
    double a = seed + lane * 1e-3, b = seed * 0.5 + lane * 1e-4;
    double v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i + lane;
    bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    bool do_valu = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    if (MODE == 3) { do_mfma = __builtin_amdgcn_readfirstlane(do_mfma); do_valu = __builtin_amdgcn_readfirstlane(do_valu); }
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < (MODE == 2 ? 4 : 16); ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fma(v[i], a, b);
        }
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
static void run(const char *name, int blocks, int iters, double mfma_per_iter_per_wave, double fma_per_iter_per_wave) {
    double *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.0);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0 + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    double waves = (double)blocks * 4;
    double mf = mfma_per_iter_per_wave * iters * waves * 2048.0;   // flop per 16x16x4 MFMA
    double vf = fma_per_iter_per_wave * iters * waves * 128.0;     // 64 lanes x 2 flop
    printf("%-34s blocks %5d  %8.3f ms  MFMA %7.2f TF  VALU %7.2f TF  total %7.2f TF\n", name, blocks, best,
           mf / best / 1e9, vf / best / 1e9, (mf + vf) / best / 1e9);
    hipFree(out);
}

int main() {
    for (int wpc : {1, 2, 4}) {   // workgroups (4 waves) per CU
        int blocks = 256 * wpc;
        printf("--- %d workgroup(s) of 4 waves per CU\n", wpc);
        run<0>("mfma_f64_16x16x4 only", blocks, 20000, 16, 0);
        run<1>("v_fma_f64 only", blocks, 2000, 0, 256);
        run<2>("same wave: 16 mfma + 64 fma / iter", blocks, 20000, 16, 64);
        run<3>("alternate waves mfma | fma", blocks, 4000, 8, 128);   // per-wave averages over the pair
    }
    return 0;
}
