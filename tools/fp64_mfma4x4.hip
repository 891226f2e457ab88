// Rate of v_mfma_f64_4x4x4_4b_f64 (four 4x4x4 blocks per instruction) vs 16x16x4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int iters, double seed) {
    const int lane = threadIdx.x & 63;
    double a = seed + lane * 1e-3, b = seed * 0.5 + lane * 1e-4;
    double c[8];
    d4 acc[4];
    for (int i = 0; i < 8; ++i) c[i] = seed + i;
    for (int i = 0; i < 4; ++i) acc[i] = (d4){seed, seed, seed, seed};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += c[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, int blocks, int iters, double flop_per_iter_per_wave) {
    double *out; hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, 1.0); hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0 + rep); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-28s blocks %5d %8.3f ms %7.2f TF\n", name, blocks, best, flop_per_iter_per_wave * iters * blocks * 4 / best / 1e9);
    hipFree(out);
}
int main() {
    for (int wpc : {1, 2, 4}) {
        run<0>("mfma_f64_4x4x4_4b", 256 * wpc, 20000, 32 * 512.0);   // 32 instr x (4 blocks x 64 fma x 2)
        run<1>("mfma_f64_16x16x4", 256 * wpc, 20000, 8 * 2048.0);
    }
}
