// Variants of the k_bary_mfma4 inner loop (no staging, no epilogue weights): which loop shape
// sustains the v_mfma_f64_4x4x4_4b_f64 rate?   hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_loop2.hip
//   V0  kernel as shipped: one ds_read_b128 (two row groups) + 4 MFMAs per k-step, ring of 6, fenced
//   V1  the same without scheduling fences
//   V2  both halves at once: two ds_read_b128 + 8 MFMAs per k-step (8 accumulator chains), ring of 4
//   V3  no LDS reads (A operand constant in registers): the B-from-registers ceiling
//   V4  V0 with a ring of 3
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int KS = 31, NT = 2;
typedef double d2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double code_weight(unsigned code, const double *bw_col, int PW) {
    double w0 = bw_col[(code & 255u) * PW];
    double w1 = bw_col[((code >> 8) & 255u) * PW];
    double w2 = bw_col[((code >> 16) & 255u) * PW];
    double w3 = bw_col[(code >> 24) * PW];
    return (w0 * w1) * (w2 * w3);
}

// F: 1 = four global row-code loads per tile, 2 = one barrier per tile, 4 = tile staging
// (global -> registers -> LDS, double buffered), 8 = epilogue weights looked up in an LDS table,
// 16 = three-field codes (no read of the all-ones row), 32 = epilogue deferred into the next
// half's MFMA stream (weights read at k-step 1, used at k-step 8), 64 = row codes fetched one
// tile ahead, 128 = staging as a straight 16-byte copy (image pre-packed in global memory)
template <int F, int WAVES, int NTP, int MINB>
__global__ void __launch_bounds__(64 * WAVES, MINB) kf(const double *src, const unsigned *codes, double *out, int tiles) {
    constexpr int NT = NTP; constexpr int THREADS = 64 * WAVES, SLAB = KS * 64, CPT = (SLAB + THREADS - 1) / THREADS, PW = 32;
    constexpr int CPT2 = (SLAB / 2 + THREADS - 1) / THREADS;
    constexpr int NF = (F & 16) ? 3 : 4;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c = lane & 15;
    double *hw = lds + 2 * SLAB + wave * 34 * PW;
    for (int i = threadIdx.x; i < 2 * SLAB; i += THREADS) lds[i] = src[i % SLAB];
    for (int i = lane; i < 34 * PW; i += 64) hw[i] = 1.0 + 1e-6 * i;
    __syncthreads();
    double B[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KS; ++s) B[nt][s] = 1.0 + 1e-3 * (lane + s + nt);
    double cs[NT] = {};
    double stage[CPT];
    d2_t stage2[CPT2];
#pragma unroll
    for (int r = 0; r < CPT; ++r) stage[r] = 0.0;
#pragma unroll
    for (int r = 0; r < CPT2; ++r) stage2[r] = (d2_t){0.0, 0.0};
    const int aoff = ((lane >> 4) * 4 + (lane & 3)) * 2;
    auto load_codes = [&](int t, unsigned (&cd)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cd[j] = codes[16 * (t % 84) + 4 * j + g];
    };
    auto weights = [&](unsigned code, int nt, double (&w)[NF]) {
#pragma unroll
        for (int f = 0; f < NF; ++f) w[f] = hw[((code >> (8 * f)) & 255u) * PW + 16 * nt + c];
    };
    auto wprod = [&](const double (&w)[NF]) { return NF == 4 ? (w[0] * w[1]) * (w[2] * w[3]) : (w[0] * w[1]) * w[2]; };
    unsigned cnext[4] = {0x21000305u, 0x21010406u, 0x21020507u, 0x21030608u};
    if (F & 64) load_codes(0, cnext);
    double pacc[2][NT] = {};
    unsigned pcode[2] = {0x21212121u, 0x21212121u};
    for (int t = 0; t < tiles; ++t) {
        const double *cur = lds + (size_t)(t & 1) * SLAB;
        double *nxt = lds + (size_t)((t + 1) & 1) * SLAB;
        if (F & 128) {
            d2_t *nx2 = reinterpret_cast<d2_t *>(nxt);
            const d2_t *s2 = reinterpret_cast<const d2_t *>(src + (size_t)(t % 84) * SLAB);
#pragma unroll
            for (int r = 0; r < CPT2; ++r) { const int i = threadIdx.x + THREADS * r; if (i < SLAB / 2) nx2[i] = stage2[r]; }
#pragma unroll
            for (int r = 0; r < CPT2; ++r) { const int i = threadIdx.x + THREADS * r; stage2[r] = (i < SLAB / 2) ? s2[i] : (d2_t){0.0, 0.0}; }
        } else if (F & 4) {
#pragma unroll
            for (int r = 0; r < CPT; ++r) { const int i = threadIdx.x + THREADS * r; if (i < SLAB) nxt[i] = stage[r]; }
#pragma unroll
            for (int r = 0; r < CPT; ++r) { const int i = threadIdx.x + THREADS * r; stage[r] = (i < SLAB) ? src[(size_t)(t % 84) * SLAB + i] : 0.0; }
        }
        unsigned cd[4] = {0x21000305u, 0x21010406u, 0x21020507u, 0x21030608u};
        if (F & 64) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cd[j] = cnext[j];
            load_codes(t + 1, cnext);
        } else if (F & 1) {
            load_codes(t, cd);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            constexpr int DEPTH = 6;
            double acc[2][NT] = {};
            const d2_t *ar = reinterpret_cast<const d2_t *>(cur + aoff + 32 * half);
            d2_t ring[DEPTH];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) ring[s] = ar[s * 32];
            double wv[2][NT][NF];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double a0 = ring[s % DEPTH][0], a1 = ring[s % DEPTH][1];
                if (s + DEPTH < KS) ring[s % DEPTH] = ar[(s + DEPTH) * 32];
                if ((F & 32) && s == 1) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) weights(pcode[r], nt, wv[r][nt]);
                }
                if ((F & 32) && s == 8) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        cs[nt] = __builtin_fma(pacc[0][nt], wprod(wv[0][nt]), cs[nt]);
                        cs[nt] = __builtin_fma(pacc[1][nt], wprod(wv[1][nt]), cs[nt]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, B[nt][s], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, B[nt][s], acc[1][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const unsigned codeA = cd[2 * half], codeB = cd[2 * half + 1];
            if (F & 32) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { pacc[0][nt] = acc[0][nt]; pacc[1][nt] = acc[1][nt]; }
                pcode[0] = codeA; pcode[1] = codeB;
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (F & 8) {
                        double w0[NF], w1[NF];
                        weights(codeA, nt, w0); weights(codeB, nt, w1);
                        cs[nt] = __builtin_fma(acc[0][nt], wprod(w0), cs[nt]);
                        cs[nt] = __builtin_fma(acc[1][nt], wprod(w1), cs[nt]);
                    } else {
                        cs[nt] += acc[0][nt] + acc[1][nt] * (double)(codeA + codeB);
                    }
                }
            }
        }
        if (F & 2) __syncthreads();
    }
    if (F & 32) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double w0[NF], w1[NF];
            weights(pcode[0], nt, w0); weights(pcode[1], nt, w1);
            cs[nt] = __builtin_fma(pacc[0][nt], wprod(w0), cs[nt]);
            cs[nt] = __builtin_fma(pacc[1][nt], wprod(w1), cs[nt]);
        }
    }
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = cs[0] + cs[NT - 1];
}

template <int F, int WAVES, int NTP = 2, int MINB = 2>
void runf(const double *src, const unsigned *codes) {
    for (int per_cu : {1, 2}) {
        const int blocks = 256 * per_cu;
        double *out;
        hipMalloc(&out, (size_t)blocks * 64 * WAVES * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int tiles = 84 * 4;
        const size_t lds = ((size_t)2 * KS * 64 + WAVES * 34 * 32) * 8;
        hipFuncSetAttribute((const void *)kf<F, WAVES, NTP, MINB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((kf<F, WAVES, NTP, MINB>), dim3(blocks), dim3(64 * WAVES), lds, 0, src, codes, out, 8); hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0); hipLaunchKernelGGL((kf<F, WAVES, NTP, MINB>), dim3(blocks), dim3(64 * WAVES), lds, 0, src, codes, out, tiles); hipEventRecord(e1);
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double flop = (double)blocks * WAVES * tiles * 4 * KS * NTP * 512.0;
        printf("NT=%d F=%3d  %d waves/WG x %d WG/CU: %.3f ms  %.2f TF (MFMA-executed)\n", NTP, F, WAVES, per_cu, best, flop / best / 1e9);
        hipFree(out);
    }
}

template <int V, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) k(const double *src, double *out, int tiles) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < KS * 64; i += 64 * WAVES) lds[i] = src[i];
    __syncthreads();
    double B[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int s = 0; s < KS; ++s) B[nt][s] = 1.0 + 1e-3 * (lane + s + nt);
    double sum[NT] = {0.0, 0.0};
    const int aoff = ((lane >> 4) * 4 + (lane & 3)) * 2;
    for (int t = 0; t < tiles; ++t) {
        if (V == 2) {
            constexpr int DEPTH = 4;
            double acc[4][NT] = {};
            const d2_t *ar = reinterpret_cast<const d2_t *>(lds + aoff);
            d2_t r0[DEPTH], r1[DEPTH];
#pragma unroll
            for (int s = 0; s < DEPTH; ++s) { r0[s] = ar[s * 32]; r1[s] = ar[s * 32 + 16]; }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const d2_t a = r0[s % DEPTH], b = r1[s % DEPTH];
                if (s + DEPTH < KS) { r0[s % DEPTH] = ar[(s + DEPTH) * 32]; r1[s % DEPTH] = ar[(s + DEPTH) * 32 + 16]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[0], B[nt][s], acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[1], B[nt][s], acc[1][nt], 0, 0, 0);
                    acc[2][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(b[0], B[nt][s], acc[2][nt], 0, 0, 0);
                    acc[3][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(b[1], B[nt][s], acc[3][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sum[nt] += (acc[0][nt] + acc[1][nt]) + (acc[2][nt] + acc[3][nt]) * (1.0 + 1e-9 * t);
        } else {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                constexpr int DEPTH = (V == 4) ? 3 : 6;
                double acc[2][NT] = {};
                const d2_t *ar = reinterpret_cast<const d2_t *>(lds + aoff + 32 * half);
                d2_t ring[DEPTH];
                if (V != 3) {
#pragma unroll
                    for (int s = 0; s < DEPTH; ++s) ring[s] = ar[s * 32];
                } else {
#pragma unroll
                    for (int s = 0; s < DEPTH; ++s) ring[s] = (d2_t){1.0 + lane, 2.0 + t};
                }
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a0 = ring[s % DEPTH][0], a1 = ring[s % DEPTH][1];
                    if (V != 3 && s + DEPTH < KS) ring[s % DEPTH] = ar[(s + DEPTH) * 32];
                    if (V != 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[0][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, B[nt][s], acc[0][nt], 0, 0, 0);
                        acc[1][nt] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, B[nt][s], acc[1][nt], 0, 0, 0);
                    }
                    if (V != 1) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) sum[nt] += (acc[0][nt] + acc[1][nt]) * (1.0 + 1e-9 * (t + half));
            }
        }
    }
    out[(size_t)blockIdx.x * 64 * WAVES + threadIdx.x] = sum[0] + sum[1];
}

template <int V, int WAVES>
void run(const double *src) {
    for (int per_cu : {1, 2}) {
        const int blocks = 256 * per_cu;
        double *out;
        hipMalloc(&out, (size_t)blocks * 64 * WAVES * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int tiles = 84 * 4;
        const size_t lds = (size_t)KS * 64 * 8 + ((per_cu == 1) ? 0 : 0);
        hipLaunchKernelGGL((k<V, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, 0, src, out, 8); hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 5; ++r) {
            hipEventRecord(e0); hipLaunchKernelGGL((k<V, WAVES>), dim3(blocks), dim3(64 * WAVES), lds, 0, src, out, tiles); hipEventRecord(e1);
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        double flop = (double)blocks * WAVES * tiles * 4 * KS * NT * 512.0;
        printf("V%d  %d waves/WG x %d WG/CU: %.3f ms  %.2f TF (MFMA-executed)\n", V, WAVES, per_cu, best, flop / best / 1e9);
        hipFree(out);
    }
}
int main(int argc, char **) {
    double *src;
    hipMalloc(&src, 84 * KS * 64 * 8); hipMemset(src, 0, 84 * KS * 64 * 8);
    unsigned *codes, hc[84 * 16];
    for (int i = 0; i < 84 * 16; ++i) hc[i] = (unsigned)(i % 11) | ((11 + (i / 11) % 11) << 8) | ((22 + (i / 121) % 11) << 16) | (33u << 24);
    hipMalloc(&codes, sizeof(hc)); hipMemcpy(codes, hc, sizeof(hc), hipMemcpyHostToDevice);
    if (argc > 1) { run<0, 4>(src); run<0, 8>(src); run<1, 4>(src); run<2, 4>(src); run<2, 8>(src); run<4, 4>(src); }
    runf<255, 8>(src, codes);
    runf<255, 8, 1, 2>(src, codes);      // NT = 1: 8 waves/WG x {1,2} WG/CU
    runf<255, 16, 1, 1>(src, codes);     // 16 waves/WG
    runf<255, 12, 1, 1>(src, codes);
    runf<0, 16, 1, 1>(src, codes);       // bare loop, NT = 1
    return 0;
}
