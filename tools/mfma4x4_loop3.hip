// "Variant 4" sandbox: v_mfma_f64_4x4x4_4b_f64 with the four blocks = the four 4-row groups of a
// 16-row tile.  A is then NOT replicated: lane 16k + 4b + i holds A[row 4b + i][k], exactly the
// coalesced 512-byte fragment the 16x16x4 kernel loads from L2 -- no LDS staging, no barriers.
// The replicated operand is B (tail-weight products, computed in registers): lane 16k + 4b + j
// holds B[k][point j], a wave covers 4 points per MFMA and 16 points with four MFMAs per
// fragment.  D: lane 16i + 4b + j = row 4b + i, point j.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_loop3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int KS = 31, MT = 84, PW = 16;

__device__ __forceinline__ double code_weight(unsigned code, const double *bw_col) {
    double w0 = bw_col[(code & 255u) * PW];
    double w1 = bw_col[((code >> 8) & 255u) * PW];
    double w2 = bw_col[((code >> 16) & 255u) * PW];
    double w3 = bw_col[(code >> 24) * PW];
    return (w0 * w1) * (w2 * w3);
}

// F: 1 = epilogue with LDS weight look-ups, 2 = row codes from global memory, 4 = explicit ring of 8
// fragment loads in flight, fenced (hipcc otherwise sinks loads towards their use)
template <int F, int NQ>
__global__ void __launch_bounds__(256, 2) k(const double *frag, const unsigned *codes, double *out, int reps) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *hw = lds + wave * 34 * PW;
    for (int i = lane; i < 34 * PW; i += 64) hw[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    double B[NQ][KS];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int s = 0; s < KS; ++s) B[q][s] = 1.0 + 1e-3 * (lane + s + q);
    typedef const double __attribute__((address_space(1))) *gptr_t;
    const gptr_t tf = (gptr_t)frag + lane;
    const int r16 = ((lane >> 2) & 3) * 4 + (lane >> 4);      // this lane's row within a tile: 4b + i
    const int pj = lane & 3;                                  // point within a group of four
    double total[NQ] = {};
    for (int rep = 0; rep < reps; ++rep) {
        double cs[NQ] = {};
        for (int t = 0; t < MT; ++t) {
            double w[NQ];
            unsigned code = 0x21000305u + (t & 3);
            if (F & 2) code = codes[16 * t + r16];
#pragma unroll
            for (int q = 0; q < NQ; ++q) w[q] = (F & 1) ? code_weight(code, hw + 4 * q + pj) : 1.0 + 1e-9 * q;
            double acc[NQ] = {};
            const gptr_t tt = tf + (size_t)t * KS * 64;
            if (F & 4) {
                constexpr int DEPTH = 8;
                double ring[DEPTH];
#pragma unroll
                for (int s = 0; s < DEPTH; ++s) ring[s] = tt[s * 64];
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = ring[s % DEPTH];
                    if (s + DEPTH < KS) ring[s % DEPTH] = tt[(s + DEPTH) * 64];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, B[q][s], acc[q], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double a = tt[s * 64];
#pragma unroll
                for (int q = 0; q < NQ; ++q) acc[q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, B[q][s], acc[q], 0, 0, 0);
            }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) cs[q] = __builtin_fma(acc[q], w[q], cs[q]);
            if ((t & 3) == 3) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) { total[q] += cs[q]; cs[q] = 0.0; }
            }
        }
    }
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) v += total[q];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
}

template <int F, int NQ>
void run(const double *frag, const unsigned *codes) {
    for (int per_cu : {2, 8, 32}) {
        const int blocks = 256 * per_cu;
        double *out;
        hipMalloc(&out, (size_t)blocks * 256 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const size_t lds = (size_t)4 * 34 * PW * 8;
        hipLaunchKernelGGL((k<F, NQ>), dim3(blocks), dim3(256), lds, 0, frag, codes, out, 1); hipDeviceSynchronize();
        float best = 1e30f;
        const int reps = 2;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0); hipLaunchKernelGGL((k<F, NQ>), dim3(blocks), dim3(256), lds, 0, frag, codes, out, reps); hipEventRecord(e1);
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double flop = (double)blocks * 4 * reps * MT * KS * NQ * 512.0;
        printf("F=%d NQ=%d  %d WG: %.3f ms  %.2f TF (MFMA-executed)\n", F, NQ, blocks, best, flop / best / 1e9);
        hipFree(out);
    }
}
int main() {
    double *frag;
    hipMalloc(&frag, (size_t)MT * KS * 64 * 8); hipMemset(frag, 0, (size_t)MT * KS * 64 * 8);
    unsigned *codes, hc[MT * 16];
    for (int i = 0; i < MT * 16; ++i) hc[i] = (unsigned)(i % 11) | ((11 + (i / 11) % 11) << 8) | ((22 + (i / 121) % 11) << 16) | (33u << 24);
    hipMalloc(&codes, sizeof(hc)); hipMemcpy(codes, hc, sizeof(hc), hipMemcpyHostToDevice);
    run<3, 4>(frag, codes); run<7, 4>(frag, codes); run<4, 4>(frag, codes);
    return 0;
}
