// Sandbox: the 16x16x4 barycentric main loop with the B operands (tail-weight products) read
// from an LDS table instead of held in registers (124 VGPRs for KS = 31, NT = 2): frees the
// register file for 4 waves per SIMD and more column tiles per wave (A-fragment reuse).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma16_ldsB.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int KS = 31, MT = 84;
typedef double d4 __attribute__((ext_vector_type(4)));

// MODE 0: B in registers (the shipped kernel's shape), MODE 1: B from LDS, MODE 2: column tile 0 from
// registers and the others from LDS
__device__ __forceinline__ double code_weight(unsigned code, const double *bw_col, int PW) {
    double w0 = bw_col[(code & 255u) * PW];
    double w1 = bw_col[((code >> 8) & 255u) * PW];
    double w2 = bw_col[((code >> 16) & 255u) * PW];
    double w3 = bw_col[(code >> 24) * PW];
    return (w0 * w1) * (w2 * w3);
}

__device__ __forceinline__ double code_weight3(unsigned code, const double *bw_col, int PW) {
    double w0 = bw_col[(code & 255u) * PW];
    double w1 = bw_col[((code >> 8) & 255u) * PW];
    double w2 = bw_col[((code >> 16) & 255u) * PW];
    return (w0 * w1) * w2;
}

// EPI 3: EPI 1 with three-field codes (no read of the all-ones row)
// EPI 1: the shipped kernel's epilogue (row codes from global memory, head weights from an LDS table,
// looked up before the MFMA chain), EPI 2: the same look-ups placed AFTER the chain
template <int MODE, int NT, int MINW, int EPI = 0>
__global__ void __launch_bounds__(256, MINW) k(const double *frag, double *out, int reps, const unsigned *codes = nullptr) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NL = (MODE == 1) ? NT : (MODE == 2 ? NT - 1 : 0);   // column tiles kept in LDS
    double *bt = lds + (size_t)wave * NL * KS * 64;            // [nt][s][64 lanes]
    for (int i = lane; i < NL * KS * 64; i += 64) bt[i] = 1.0 + 1e-6 * i;
    constexpr int PW = 16 * NT;
    double *hw = lds + (size_t)4 * NL * KS * 64 + (size_t)wave * 34 * PW;
    if (EPI) for (int i = lane; i < 34 * PW; i += 64) hw[i] = 1.0 + 1e-6 * i;
    const int g = lane >> 4, c = lane & 15;
    __syncthreads();
    constexpr int NR = (MODE == 0) ? NT : (MODE == 2 ? 1 : 0);          // column tiles kept in registers
    double B[NR > 0 ? NR : 1][NR > 0 ? KS : 1];
    if (NR > 0) {
#pragma unroll
        for (int nt = 0; nt < NR; ++nt)
#pragma unroll
            for (int s = 0; s < KS; ++s) B[nt][s] = frag[(size_t)(nt * KS + s) * 64 + lane] + 1.0;   // from memory: cannot be rematerialised
    }
    typedef const double __attribute__((address_space(1))) *gptr_t;
    const gptr_t tf = (gptr_t)frag + lane;
    double total[NT] = {};
    if (EPI == 8) {
        // hand-pipelined: fragment loads run DEPTH k-steps ahead ACROSS tile boundaries (the first DEPTH
        // fragments of tile t+1 are fetched during the tail of tile t), row codes one tile ahead, the
        // weight look-ups of a tile are issued at its start and multiplied a few k-steps later; fences
        // keep hipcc from sinking the loads back to their uses
        constexpr int DEPTH = 6;
        for (int rep = 0; rep < reps; ++rep) {
            unsigned cn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cn[j] = codes[g + 4 * j];
            double head[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) head[i] = tf[i * 64];
            d4 pacc[NT];
            double pw[NT][4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                pacc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[nt][j] = 0.0;
            }
            for (int t = 0; t < MT; ++t) {
                const gptr_t tt = tf + (size_t)t * KS * 64;
                const gptr_t tn = tf + (size_t)((t + 1 < MT) ? t + 1 : t) * KS * 64;
                double ring[DEPTH];
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) ring[i] = head[i];
                unsigned cc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) cc[j] = cn[j];
                double wr[4][NT][4];          // raw table entries of the tile's weights, row j looked up at k-step 2j
#pragma unroll
                for (int j = 0; j < 4; ++j) cn[j] = codes[16 * ((t + 1 < MT) ? t + 1 : t) + g + 4 * j];
                double w[NT][4];
                d4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = ring[s % DEPTH];
                    if (s + DEPTH < KS) ring[s % DEPTH] = tt[(s + DEPTH) * 64];
                    else head[s + DEPTH - KS] = tn[(s + DEPTH - KS) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (s == 2 * j) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int f = 0; f < 4; ++f) wr[j][nt][f] = hw[((cc[j] >> (8 * f)) & 255u) * PW + 16 * nt + c];
                        }
                        if (s == 2 * j + 2) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) w[nt][j] = (wr[j][nt][0] * wr[j][nt][1]) * (wr[j][nt][2] * wr[j][nt][3]);
                        }
                    }
                    if (s == 10) {      // fold the previous tile while this one multiplies
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int j = 0; j < 4; ++j) total[nt] = __builtin_fma(pacc[nt][j], pw[nt][j], total[nt]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[nt < NR ? nt : 0][NR > 0 ? s : 0], acc[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    pacc[nt] = acc[nt];
#pragma unroll
                    for (int j = 0; j < 4; ++j) pw[nt][j] = w[nt][j];
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) total[nt] = __builtin_fma(pacc[nt][j], pw[nt][j], total[nt]);
        }
    } else
    if (EPI == 7) {
        // hand-pipelined: fragment loads run DEPTH k-steps ahead ACROSS tile boundaries (the first DEPTH
        // fragments of tile t+1 are fetched during the tail of tile t), row codes one tile ahead, the
        // weight look-ups of a tile are issued at its start and multiplied a few k-steps later; fences
        // keep hipcc from sinking the loads back to their uses
        constexpr int DEPTH = 6;
        for (int rep = 0; rep < reps; ++rep) {
            unsigned cn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cn[j] = codes[g + 4 * j];
            double head[DEPTH];
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) head[i] = tf[i * 64];
            for (int t = 0; t < MT; ++t) {
                const gptr_t tt = tf + (size_t)t * KS * 64;
                const gptr_t tn = tf + (size_t)((t + 1 < MT) ? t + 1 : t) * KS * 64;
                double ring[DEPTH];
#pragma unroll
                for (int i = 0; i < DEPTH; ++i) ring[i] = head[i];
                unsigned cc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) cc[j] = cn[j];
                double wr[4][NT][4];          // raw table entries of the tile's weights, row j looked up at k-step 2j
#pragma unroll
                for (int j = 0; j < 4; ++j) cn[j] = codes[16 * ((t + 1 < MT) ? t + 1 : t) + g + 4 * j];
                double w[NT][4];
                d4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = ring[s % DEPTH];
                    if (s + DEPTH < KS) ring[s % DEPTH] = tt[(s + DEPTH) * 64];
                    else head[s + DEPTH - KS] = tn[(s + DEPTH - KS) * 64];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (s == 2 * j) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                for (int f = 0; f < 4; ++f) wr[j][nt][f] = hw[((cc[j] >> (8 * f)) & 255u) * PW + 16 * nt + c];
                        }
                        if (s == 2 * j + 2) {
#pragma unroll
                            for (int nt = 0; nt < NT; ++nt) w[nt][j] = (wr[j][nt][0] * wr[j][nt][1]) * (wr[j][nt][2] * wr[j][nt][3]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[nt < NR ? nt : 0][NR > 0 ? s : 0], acc[nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) total[nt] = __builtin_fma(acc[nt][j], w[nt][j], total[nt]);
            }
        }
    } else
    if (EPI == 6) {
        // EPI 1 with the four row codes of tile t+1 fetched at the top of tile t: the wait for them no
        // longer drains the tile's own fragment loads (loads return in order: vmcnt(0) otherwise)
        for (int rep = 0; rep < reps; ++rep) {
            unsigned cn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cn[j] = codes[g + 4 * j];
            for (int t = 0; t < MT; ++t) {
                unsigned cc[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) cc[j] = cn[j];
                const int tn = (t + 1 < MT) ? t + 1 : t;
#pragma unroll
                for (int j = 0; j < 4; ++j) cn[j] = codes[16 * tn + g + 4 * j];
                double w[NT][4];
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) w[nt][j] = code_weight(cc[j], hw + 16 * nt + c, PW);
                d4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
                const gptr_t tt = tf + (size_t)t * KS * 64;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = tt[s * 64];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[nt < NR ? nt : 0][NR > 0 ? s : 0], acc[nt], 0, 0, 0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) total[nt] = __builtin_fma(acc[nt][j], w[nt][j], total[nt]);
            }
        }
    } else
    if (EPI == 4 || EPI == 5) {
        // deferred epilogue: the products of tile t-1 are folded in after tile t's MFMA chain has been
        // issued (its accumulators are a second set), so no wave waits for a chain to drain
        for (int rep = 0; rep < reps; ++rep) {
            d4 pacc[NT];
            double pw[NT][4];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                pacc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < 4; ++j) pw[nt][j] = 0.0;
            }
            for (int t = 0; t < MT; ++t) {
                double w[NT][4];
                if (EPI == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned code = codes[16 * t + g + 4 * j];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) w[nt][j] = code_weight(code, hw + 16 * nt + c, PW);
                    }
                }
                d4 acc[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
                const gptr_t tt = tf + (size_t)t * KS * 64;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const double a = tt[s * 64];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B[nt < NR ? nt : 0][NR > 0 ? s : 0], acc[nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) total[nt] = __builtin_fma(pacc[nt][j], pw[nt][j], total[nt]);
                if (EPI == 5) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        unsigned code = codes[16 * t + g + 4 * j];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) w[nt][j] = code_weight(code, hw + 16 * nt + c, PW);
                    }
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    pacc[nt] = acc[nt];
#pragma unroll
                    for (int j = 0; j < 4; ++j) pw[nt][j] = w[nt][j];
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) total[nt] = __builtin_fma(pacc[nt][j], pw[nt][j], total[nt]);
        }
    } else
    for (int rep = 0; rep < reps; ++rep) {
        for (int t = 0; t < MT; ++t) {
            double w[NT][4];
            if (EPI == 1 || EPI == 3) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned code = codes[16 * t + g + 4 * j];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        w[nt][j] = (EPI == 3) ? code_weight3(code, hw + 16 * nt + c, PW) : code_weight(code, hw + 16 * nt + c, PW);
                }
            }
            d4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (d4){0.0, 0.0, 0.0, 0.0};
            const gptr_t tt = tf + (size_t)t * KS * 64;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const double a = tt[s * 64];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const double b = (nt < NR) ? B[nt < NR ? nt : 0][NR > 0 ? s : 0] : bt[((nt - NR < 0 ? 0 : nt - NR) * KS + s) * 64 + lane];
                    acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[nt], 0, 0, 0);
                }
            }
            if (EPI == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    unsigned code = codes[16 * t + g + 4 * j];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) w[nt][j] = code_weight(code, hw + 16 * nt + c, PW);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j) total[nt] = __builtin_fma(acc[nt][j], EPI ? w[nt][j] : 1.0 + 1e-9 * (j + t), total[nt]);
        }
    }
    double v = 0.0;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) v += total[nt];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
}

template <int MODE, int NT, int MINW, int EPI = 0>
void run(const double *frag, const unsigned *codes = nullptr, size_t lds_pad = 0) {
    const int blocks = 256 * 16;
    double *out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = ((size_t)4 * ((MODE == 1) ? NT : (MODE == 2 ? NT - 1 : 0)) * KS * 64 + 4 * 34 * 16 * NT) * 8 + lds_pad;
    hipFuncSetAttribute((const void *)k<MODE, NT, MINW, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k<MODE, NT, MINW, EPI>), dim3(blocks), dim3(256), lds, 0, frag, out, 1, codes); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0); hipLaunchKernelGGL((k<MODE, NT, MINW, EPI>), dim3(blocks), dim3(256), lds, 0, frag, out, 1, codes); hipEventRecord(e1);
        hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flop = (double)blocks * 4 * MT * KS * NT * 2048.0;
    printf("%s NT=%d EPI=%d min waves/SIMD %d, LDS %zu KB/WG: %.3f ms  %.2f TF (MFMA-executed)\n", MODE == 1 ? "B from LDS " : (MODE == 2 ? "B mixed    " : "B in VGPRs "), NT, EPI, MINW, lds / 1024, best, flop / best / 1e9);
    hipFree(out);
}
int main() {
    double *frag;
    hipMalloc(&frag, (size_t)MT * KS * 64 * 8); hipMemset(frag, 0, (size_t)MT * KS * 64 * 8);
    unsigned *codes, hc[MT * 16];
    for (int i = 0; i < MT * 16; ++i) hc[i] = (unsigned)(i % 11) | ((11 + (i / 11) % 11) << 8) | ((22 + (i / 121) % 11) << 16) | (33u << 24);
    hipMalloc(&codes, sizeof(hc)); hipMemcpy(codes, hc, sizeof(hc), hipMemcpyHostToDevice);
    run<0, 2, 2>(frag); run<0, 2, 2, 1>(frag, codes); run<0, 2, 2, 2>(frag, codes);
    run<0, 2, 2, 3>(frag, codes); run<0, 2, 2, 7>(frag, codes); run<0, 2, 2, 8>(frag, codes);
    return 0;
}
