// tt_lpp_kernels.h -- lane-per-point tensor-train evaluation for small ranks (gfx950), round 3.
//
// Replaces ChebyshevTT.eval_batch (reference tensor_train.py:2217-2265) for ranks <= 16, n <= 16:
//     v <- v . (sum_j T_j(s_k) G_k[:, j, :])   for every storage dimension k.
//
// Why not MFMA here: on gfx950 the FP64 matrix instruction occupies the SIMD's vector pipe
// (profiles/r02_fp64_mix_microbench.txt: nothing co-executes with it) and v_fma_f64 has the same
// peak flop rate, so MFMA buys nothing for FP64 -- and at ranks <= 12 the MFMA forms carry ~180
// vector instructions per 16 points (recurrences, folds, the last dimension) next to their matrix
// work (profiles/r02_tt_kernel_lab.txt).  With lane = point every vector instruction serves 64
// points and the instruction count is the algorithmic FMA count plus ~4 %:
//     M[a]  = sum_j T_j(x) G[a][j][b]        (rl * n FMAs per b; T_0 = 1 held in a register)
//     v'[b] = sum_a v[a] M[a]                (rl FMAs per b)
// A core element is wave-uniform: it is read with SCALAR loads (s_load_dwordx16 through the scalar
// cache) and enters v_fma_f64 as an SGPR operand -- one 8-byte load feeds 64 FMAs; neither LDS
// bandwidth nor VGPRs are spent on the cores.  The right-rank loop runs at run time (v' goes
// through a lane-private LDS column, one ds_write per b and rl ds_reads per dimension), the left
// rank and the node count are compile-time: a body is rl * n FMAs of straight-line code, picked per
// dimension by a switch, so ranks and node counts are exact -- nothing is padded.
// image: img[off_k + (b * rl + a) * n + j] = G_k[a][j][b].
// Measured on config 3 (5-D, ranks [1,8,8,8,6,1], n = 11; tools/tt_lpp_lab.hip,
// profiles/r03_tt_lpp_lab.txt): 2,469 vector instructions per 64 points at 4.2 cycles each, 0.70 - 0.78 of the FP64
// peak depending on the clock the box settles at (round 2, k_tt_eval_d4: 0.555).
#pragma once

#include "pcx_common.h"

typedef const double __attribute__((address_space(4))) *pcx_lpp_cptr;

struct TTLppDim {
    int off;        // offset (doubles) of storage dim k in the image
    int rl, rr;     // left / right rank
    int n;          // nodes
    int col;        // user column read by storage position k (dim_order)
    int pad_;
    double lo;
    double scale;   // 2 / (hi - lo): s = fma(x - lo, scale, -1)
};

#define PCX_LPP_MAX_RANK 16
#define PCX_LPP_MAX_NODES 16
#define PCX_LPP_WG 64
#ifndef PCX_LPP_MINB8
#define PCX_LPP_MINB8 8       // ranks <= 8
#endif
#ifndef PCX_LPP_MINB12
#define PCX_LPP_MINB12 6      // ranks 9..12 at <= 80 VGPRs: at 8 (<= 64 VGPRs) the rank-12 body spills (10-D rank 12: 0.55 against 0.65)
#endif

template <int RL, int NJ>
__device__ __forceinline__ void tt_lpp_body(pcx_lpp_cptr G, int rr, double x, double *vl) {
    double T[NJ], v[RL];
    const double x2 = x + x;
    T[0] = 1.0;
    asm volatile("" : "+v"(T[0]));          // T_0 in a register: M = g * T_0 is one v_mul_f64, not two v_mov_b32
    if constexpr (NJ > 1) T[1] = x;
#pragma unroll
    for (int j = 2; j < NJ; ++j) T[j] = __builtin_fma(x2, T[j - 1], -T[j - 2]);
#pragma unroll
    for (int a = 0; a < RL; ++a) v[a] = vl[a * PCX_LPP_WG];      // dimension 0 reads the 1.0 the kernel put there
    for (int b = 0; b < rr; ++b, G += RL * NJ) {
        double M[RL];
#pragma unroll
        for (int a = 0; a < RL; ++a) M[a] = G[a * NJ] * T[0];
#pragma unroll
        for (int j = 1; j < NJ; ++j)
#pragma unroll
            for (int a = 0; a < RL; ++a) M[a] = __builtin_fma(T[j], G[a * NJ + j], M[a]);
        double s;
        if constexpr (RL < 4) {
            s = v[0] * M[0];
#pragma unroll
            for (int a = 1; a < RL; ++a) s = __builtin_fma(v[a], M[a], s);
        } else {                                 // two chains: half the dependent-FMA latency (one chain: +0.4 %, not worth a different sum order)
            double s0 = v[0] * M[0], s1 = v[1] * M[1];
#pragma unroll
            for (int a = 2; a < RL; a += 2) {
                s0 = __builtin_fma(v[a], M[a], s0);
                if (a + 1 < RL) s1 = __builtin_fma(v[a + 1], M[a + 1], s1);
            }
            s = s0 + s1;
        }
        vl[b * PCX_LPP_WG] = s;
    }
}

#define PCX_LPP_RANK_CASES_8(NJ)                                                                          \
    case 1: tt_lpp_body<1, NJ>(G, rr, x, vl); break; case 2: tt_lpp_body<2, NJ>(G, rr, x, vl); break;     \
    case 3: tt_lpp_body<3, NJ>(G, rr, x, vl); break; case 4: tt_lpp_body<4, NJ>(G, rr, x, vl); break;     \
    case 5: tt_lpp_body<5, NJ>(G, rr, x, vl); break; case 6: tt_lpp_body<6, NJ>(G, rr, x, vl); break;     \
    case 7: tt_lpp_body<7, NJ>(G, rr, x, vl); break; case 8: tt_lpp_body<8, NJ>(G, rr, x, vl); break;
#define PCX_LPP_RANK_CASES_12(NJ)                                                                         \
    case 9: tt_lpp_body<9, NJ>(G, rr, x, vl); break; case 10: tt_lpp_body<10, NJ>(G, rr, x, vl); break;   \
    case 11: tt_lpp_body<11, NJ>(G, rr, x, vl); break; case 12: tt_lpp_body<12, NJ>(G, rr, x, vl); break;
#define PCX_LPP_RANK_CASES_16(NJ)                                                                         \
    case 13: tt_lpp_body<13, NJ>(G, rr, x, vl); break; case 14: tt_lpp_body<14, NJ>(G, rr, x, vl); break; \
    case 15: tt_lpp_body<15, NJ>(G, rr, x, vl); break; case 16: tt_lpp_body<16, NJ>(G, rr, x, vl); break;

template <int RCAP, int NJ>
__device__ __forceinline__ void tt_lpp_dim(int rl, pcx_lpp_cptr G, int rr, double x, double *vl) {
    switch (rl) {
        PCX_LPP_RANK_CASES_8(NJ)
        default:
            if constexpr (RCAP > 12) {
                switch (rl) { PCX_LPP_RANK_CASES_12(NJ) PCX_LPP_RANK_CASES_16(NJ) default: break; }
            } else if constexpr (RCAP > 8) {
                switch (rl) { PCX_LPP_RANK_CASES_12(NJ) default: break; }
            }
            break;
    }
}

// One wave per workgroup, one point per lane; dynamic LDS = max rank * 64 * 8 bytes (the lane-private
// columns of v').  RCAP = 8 / 12 / 16: the left ranks the instantiation covers (its register budget follows
// the largest body; the reference's max_rank = 15 Black-Scholes model, ranks [1,11,11,11,7,1], runs on RCAP = 12:
// 0.72 of the FP64 peak against 0.68 on RCAP = 16).  NJ = the node count when every dimension has the same one (the usual model: the
// kernel then holds only the <= RCAP bodies of that node count, and hipcc allocates registers far better
// than across the 256 bodies of the two-level switch), 0 = node counts differ: dispatch on both.
// The coordinate of dimension k + 1 is fetched while dimension k is contracted.
template <int RCAP, int NJ>
__global__ void __launch_bounds__(PCX_LPP_WG, RCAP <= 8 ? PCX_LPP_MINB8 : (RCAP <= 12 ? PCX_LPP_MINB12 : 4))
k_tt_eval_lpp(const TTLppDim *__restrict__ tab, int d, const double *__restrict__ img,
              const double *__restrict__ pts, double *__restrict__ out, long N) {
    extern __shared__ double lds_lpp[];
    double *vl = lds_lpp + threadIdx.x;
    typedef const TTLppDim __attribute__((address_space(4))) *tab_cptr;
    const tab_cptr ct = (tab_cptr)(unsigned long long)tab;
    const pcx_lpp_cptr cimg = (pcx_lpp_cptr)(unsigned long long)img;
    const long p = (long)blockIdx.x * PCX_LPP_WG + threadIdx.x;
#ifdef PCX_LPP_LAB_WRAP      // tools/tt_w4_lab.hip only: every wave reads the same 1,024 rows and stores nothing beyond them (no HBM traffic)
    const long pc = p & 1023;
#else
    const long pc = p < N ? p : N - 1;
#endif
    double xn = pts[pc * d + ct[0].col];
    vl[0] = 1.0;                                   // v of the (rank-1) left boundary
    for (int k = 0; k < d; ++k) {
        const double x = __builtin_fma(xn - ct[k].lo, ct[k].scale, -1.0);    // tensor_train.py:2254
        if (k + 1 < d) xn = pts[pc * d + ct[k + 1].col];
        const pcx_lpp_cptr G = cimg + ct[k].off;
        const int rl = ct[k].rl, rr = ct[k].rr;
        if constexpr (NJ > 0) {
            tt_lpp_dim<RCAP, NJ>(rl, G, rr, x, vl);
        } else {
            switch (ct[k].n) {
            case 1: tt_lpp_dim<RCAP, 1>(rl, G, rr, x, vl); break;
            case 2: tt_lpp_dim<RCAP, 2>(rl, G, rr, x, vl); break;
            case 3: tt_lpp_dim<RCAP, 3>(rl, G, rr, x, vl); break;
            case 4: tt_lpp_dim<RCAP, 4>(rl, G, rr, x, vl); break;
            case 5: tt_lpp_dim<RCAP, 5>(rl, G, rr, x, vl); break;
            case 6: tt_lpp_dim<RCAP, 6>(rl, G, rr, x, vl); break;
            case 7: tt_lpp_dim<RCAP, 7>(rl, G, rr, x, vl); break;
            case 8: tt_lpp_dim<RCAP, 8>(rl, G, rr, x, vl); break;
            case 9: tt_lpp_dim<RCAP, 9>(rl, G, rr, x, vl); break;
            case 10: tt_lpp_dim<RCAP, 10>(rl, G, rr, x, vl); break;
            case 11: tt_lpp_dim<RCAP, 11>(rl, G, rr, x, vl); break;
            case 12: tt_lpp_dim<RCAP, 12>(rl, G, rr, x, vl); break;
            case 13: tt_lpp_dim<RCAP, 13>(rl, G, rr, x, vl); break;
            case 14: tt_lpp_dim<RCAP, 14>(rl, G, rr, x, vl); break;
            case 15: tt_lpp_dim<RCAP, 15>(rl, G, rr, x, vl); break;
            case 16: tt_lpp_dim<RCAP, 16>(rl, G, rr, x, vl); break;
            default: break;
            }
        }
    }
#ifdef PCX_LPP_LAB_WRAP
    if (p < 1024) out[p] = vl[0];
#else
    if (p < N) out[p] = vl[0];
#endif
}
