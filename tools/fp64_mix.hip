// fp64_mix.hip -- what shares a pipe with the FP64 MFMA on gfx950?
// One loop body per pattern: M matrix instructions (v_mfma_f64_4x4x4_4b or 16x16x4, independent
// accumulators) and V other instructions of one kind, issued by the SAME wave; 1, 2 and 4 waves per
// SIMD.  Reported: WALL nanoseconds per loop iteration and SIMD (HIP events around the launch; s_memtime
// ticks turned out not to track the shader clock under DVFS), next to the matrix-busy cycles of the body
// (16 or 64 per instruction; 128 cycles = 53 ns at 2.4 GHz).  If a kind of instruction co-executed with
// the matrix pipe, the time per iteration would stay at the matrix-only row.
//   hipcc --offload-arch=gfx950 -O3 tools/fp64_mix.hip -o build_exp/fp64_mix && build_exp/fp64_mix
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

enum Kind { NONE, FMA64, ADD64, MUL64, IADD, FMA32, MOV, LDS64, PERM };

template <int M4, int M16, int V, int KIND>
__global__ void __launch_bounds__(1024) k_mix(double *out, long long *cyc, int iters, double seed) {
    __shared__ double lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x & 1023] = seed * threadIdx.x;
    __syncthreads();
    double acc4[M4 > 0 ? M4 : 1];
    d4 acc16[M16 > 0 ? M16 : 1];
    for (int i = 0; i < (M4 > 0 ? M4 : 1); ++i) acc4[i] = 0.0;
    for (int i = 0; i < (M16 > 0 ? M16 : 1); ++i) acc16[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double a = seed + lane, b = 1.0 - seed * lane;
    double vd[V > 0 ? V : 1];
    float vf[V > 0 ? V : 1];
    int vi[V > 0 ? V : 1];
    for (int i = 0; i < (V > 0 ? V : 1); ++i) { vd[i] = seed * (i + 1); vf[i] = (float)seed * (i + 2); vi[i] = lane + i; }
    const double c1 = 1.0000001, c2 = 1e-9;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < (M4 > V ? (M4 > M16 ? M4 : M16) : (V > M16 ? V : M16)); ++i) {
            if (i < M4) acc4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc4[i], 0, 0, 0);
            if (M16 > 0 && i < M16) acc16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc16[i], 0, 0, 0);
            if (i < V) {
                if (KIND == FMA64) vd[i] = __builtin_fma(vd[i], c1, c2);
                if (KIND == ADD64) vd[i] = vd[i] + c2;
                if (KIND == MUL64) vd[i] = vd[i] * c1;
                if (KIND == IADD) vi[i] = vi[i] * 3 + it;
                if (KIND == FMA32) vf[i] = __builtin_fmaf(vf[i], 1.0000001f, 1e-9f);
                if (KIND == MOV) asm volatile("v_mov_b32 %0, %0" : "+v"(vi[i]));
                if (KIND == LDS64) vd[i] += lds[(lane + 64 * i + it) & 1023];
                if (KIND == PERM) vi[i] = __builtin_amdgcn_permlane32_swap((unsigned)vi[i], (unsigned)vi[(i + 1) % (V > 0 ? V : 1)], false, false)[0];
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
    for (int i = 0; i < (M4 > 0 ? M4 : 1); ++i) s += acc4[i];
    for (int i = 0; i < (M16 > 0 ? M16 : 1); ++i) s += acc16[i][0] + acc16[i][3];
    for (int i = 0; i < (V > 0 ? V : 1); ++i) s += vd[i] + vf[i] + vi[i];
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = s;
    if (lane == 0) cyc[(size_t)blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int M4, int M16, int V, int KIND>
static int run(const char *name, int cus, double *d_out, long long *d_cyc) {
    const int iters = 200000;
    printf("%-44s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        // ONE workgroup of 256 wps threads per CU: its 4 wps waves are dealt round-robin over the CU's four SIMDs
        const int blocks = cus;
        hipLaunchKernelGGL((k_mix<M4, M16, V, KIND>), dim3(blocks), dim3(256 * wps), 0, 0, d_out, d_cyc, iters, 1e-3);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 5; ++rep)
            hipLaunchKernelGGL((k_mix<M4, M16, V, KIND>), dim3(blocks), dim3(256 * wps), 0, 0, d_out, d_cyc, iters, 1e-3);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double ns = (double)ms * 1e6 / 5.0 / iters / wps;      // SIMD time per wave-iteration
        printf("  %dw/SIMD: %6.1f ns", wps, ns);
    }
    printf("   (matrix busy %d)\n", M4 * 16 + M16 * 64);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *d_out;
    long long *d_cyc;
    CK(hipMalloc(&d_out, (size_t)cus * 1024 * 8));
    CK(hipMalloc(&d_cyc, (size_t)cus * 16 * 8));
    printf("# wall ns per loop iteration and SIMD (launch time / iterations / waves per SIMD); 128 busy cycles = 53.3 ns at 2.4 GHz\n");
    run<8, 0, 0, NONE>("8 x mfma 4x4x4", cus, d_out, d_cyc);
    run<0, 2, 0, NONE>("2 x mfma 16x16x4", cus, d_out, d_cyc);
    run<0, 0, 8, FMA64>("8 x v_fma_f64", cus, d_out, d_cyc);
    run<0, 0, 8, IADD>("8 x (v_mul_lo + v_add) u32", cus, d_out, d_cyc);
    run<0, 0, 8, FMA32>("8 x v_fma_f32", cus, d_out, d_cyc);
    run<8, 0, 8, FMA64>("8 x mfma 4x4x4 + 8 x v_fma_f64", cus, d_out, d_cyc);
    run<8, 0, 8, ADD64>("8 x mfma 4x4x4 + 8 x v_add_f64", cus, d_out, d_cyc);
    run<8, 0, 8, IADD>("8 x mfma 4x4x4 + 8 x int mul-add", cus, d_out, d_cyc);
    run<8, 0, 8, FMA32>("8 x mfma 4x4x4 + 8 x v_fma_f32", cus, d_out, d_cyc);
    run<8, 0, 8, MOV>("8 x mfma 4x4x4 + 8 x v_mov_b32", cus, d_out, d_cyc);
    run<8, 0, 8, LDS64>("8 x mfma 4x4x4 + 8 x (ds_read_b64 + v_add_f64)", cus, d_out, d_cyc);
    run<8, 0, 8, PERM>("8 x mfma 4x4x4 + 8 x v_permlane32_swap", cus, d_out, d_cyc);
    run<8, 0, 2, FMA64>("8 x mfma 4x4x4 + 2 x v_fma_f64", cus, d_out, d_cyc);
    run<8, 0, 4, FMA64>("8 x mfma 4x4x4 + 4 x v_fma_f64", cus, d_out, d_cyc);
    run<0, 2, 8, FMA64>("2 x mfma 16x16x4 + 8 x v_fma_f64", cus, d_out, d_cyc);
    run<0, 2, 8, IADD>("2 x mfma 16x16x4 + 8 x int mul-add", cus, d_out, d_cyc);
    run<0, 2, 8, FMA32>("2 x mfma 16x16x4 + 8 x v_fma_f32", cus, d_out, d_cyc);
    return 0;
}
