// Does the real inner-loop shape sustain the 4x4x4_4b rate?  A operands by broadcast
// ds_read_b64 from LDS (16 distinct addresses per wave-instruction), B operands in registers,
// 4 row-groups x 2 point-tiles of accumulators, 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int KS = 31, NT = 2;
__global__ void __launch_bounds__(512) k(const double *src, double *out, int tiles) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4 * KS * 16; i += 512) lds[i] = src[i];
    __syncthreads();
    double B[NT][KS];
    for (int nt = 0; nt < NT; ++nt)
        for (int s = 0; s < KS; ++s) B[nt][s] = 1.0 + 1e-3 * (lane + s + nt);
    double sum[NT] = {0.0, 0.0};
    const double *ap = lds + (lane >> 4) * 4 + (lane & 3);
    for (int t = 0; t < tiles; ++t) {
#pragma unroll 1
        for (int rg = 0; rg < 4; ++rg) {
            double acc[NT] = {0.0, 0.0};
            const double *ar = ap + rg * KS * 16;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                double a = ar[s * 16];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, B[nt][s], acc[nt], 0, 0, 0);
            }
            for (int nt = 0; nt < NT; ++nt) sum[nt] += acc[nt] * (1.0 + 1e-9 * (t + rg));
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = sum[0] + sum[1];
}
int main() {
    double *src, *out;
    hipMalloc(&src, 4 * KS * 16 * 8); hipMemset(src, 0, 4 * KS * 16 * 8);
    for (int blocks : {256, 512}) {
        hipMalloc(&out, (size_t)blocks * 512 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int tiles = 84 * 4;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 4 * KS * 16 * 8, 0, src, out, 8); hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 4 * KS * 16 * 8, 0, src, out, tiles); hipEventRecord(e1);
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double flop = (double)blocks * 8 * tiles * 4 * KS * NT * 512.0;
        printf("blocks %d: %.3f ms  %.2f TF (MFMA-executed)\n", blocks, best, flop / best / 1e9);
        hipFree(out);
    }
}
