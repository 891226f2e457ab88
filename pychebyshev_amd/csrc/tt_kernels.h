// tt_kernels.h -- HIP kernels for the tensor-train interpolant (gfx950).
//
// Replaces ChebyshevTT.eval_batch (reference tensor_train.py:2217-2265): for each
// storage dimension k, v <- v . (sum_j T_j(s_k) G_k[:, j, :]).
//
// GEMM form per dimension, points on the MFMA N axis:
//     v'[b, p] = sum_{(j,a)} G_k[a, j, b] * ( v[a, p] * T_j(s_k(p)) )
//   A operand  = core fragment, lane l holds G_k[a = 4c + (l>>4)][j][b = 16t + (l&15)]
//   B operand  = z_j = v[a = 4c + (l>>4), p = l&15] * T_j(p)        (round 3: from the recurrence on the products,
//                                                                    one FMA; ranks > 16: one VALU multiply)
//   D (16x16)  = lane l, reg i holds row b = 16t + (l>>4) + 4i, column p = l & 15
// With the k index ordered (j, a) and ranks padded to multiples of 4, the row a a lane
// needs as B operand for chunk c is exactly the row it already holds in D register
// (c & 3) of tile (c >> 2): the chain runs dimension after dimension with NO cross-lane
// movement and no LDS.
#pragma once

#include <type_traits>

#include "pcx_common.h"

// Core packing: frag[dim][j][c][t][l] = G[a = 4c + (l>>4)][j][b = 16t + (l&15)], zero padded;
// c < rc = ceil(r_left/4), t < rt = ceil(r_right/16).
__global__ void k_tt_pack_core(const double *__restrict__ G, double *__restrict__ frag, int rl,
                               int n, int rr, int rc, int rt) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)n * rc * rt * 64;
    if (idx >= total) return;
    int l = (int)(idx & 63);
    long rest = idx >> 6;
    int t = (int)(rest % rt);
    rest /= rt;
    int c = (int)(rest % rc);
    int j = (int)(rest / rc);
    int a = 4 * c + (l >> 4);
    int b = 16 * t + (l & 15);
    frag[idx] = (a < rl && b < rr) ? G[((long)a * n + j) * rr + b] : 0.0;
}

// ---- one dimension of the direct form -------------------------------------------------
// acc[nt][t] += sum_j sum_{c < RCX} mfma(frag[j][c][t], v[nt][c] * T_j(s[nt])).
// RCX/RT are compile-time: every loop below is fully unrolled and branch-free, the
// fragments of node j+1 are in flight (registers, ping-pong) while node j is multiplied.
template <int RCX, int RT, int NT>
__device__ __forceinline__ void tt_mfma_nodes(const double *__restrict__ fk, int n,
                                              const double (&v)[NT][RCX > 0 ? RCX : 1],
                                              const double (&s)[NT], pcx_d4 (&acc)[NT][RT]) {
    constexpr int F = RCX * RT;          // fragments per node
    constexpr bool PINGPONG = (F <= 8);           // F = 16 (rank 32) measured: 184 VGPRs, 0.886 against 0.90 without the second fragment set
    // B operands from the Chebyshev recurrence applied to the PRODUCTS z_j = v T_j (as in k_tt_eval_d4):
    //     z_{j+1} = 2 s z_j - z_{j-1},   z_0 = v,  z_1 = s v
    // one FMA per (chunk, column tile) and node -- no table of T_j, no multiply per MFMA (round 2: one multiply per
    // MFMA plus the T recurrence, 20 vector instructions per 16 MFMAs; now 16).  The recurrence runs IN PLACE,
    // alternating two names: za <- 2s zb - za = z_{j+2} after node j, zb <- 2s za - zb = z_{j+3} after node j+1 -- two
    // nodes later every value sits in the register it started in, so the two-node loop body needs no register moves
    // (round 2 rotated three names through two: hipcc closed the loop with ~17 v_mov_b64 per 32 MFMAs).
    constexpr int RCV = RCX > 0 ? RCX : 1;
    double za[NT][RCV], zb[NT][RCV], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        s2[nt] = s[nt] + s[nt];
#pragma unroll
        for (int c = 0; c < RCX; ++c) { za[nt][c] = v[nt][c]; zb[nt][c] = v[nt][c] * s[nt]; }
    }

    auto load = [&](double (&a)[F], int j) {
        const double *fj = fk + (size_t)j * F * 64;
#pragma unroll
        for (int f = 0; f < F; ++f) a[f] = fj[f * 64];
    };
    // one node: the MFMAs on z_j (q), then q <- z_{j+2} in place (o: z_{j+1})
    auto node = [&](const double (&a)[F], double (&q)[NT][RCV], const double (&o)[NT][RCV]) {
#pragma unroll
        for (int c = 0; c < RCX; ++c) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c * RT + t], q[nt][c], acc[nt][t], 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < RCX; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) q[nt][c] = __builtin_fma(s2[nt], o[nt][c], -q[nt][c]);
    };

    if (PINGPONG) {
        // the fragments of node j+1 are requested BEFORE node j is multiplied and those of node j+2 before node j+1:
        // fenced, because hipcc otherwise sinks each load next to its first use (one exposed L2 round trip per
        // two nodes: s_waitcnt vmcnt right behind the load, profiles/r03_tt10d_*)
        double a0[F], a1[F];
        load(a0, 0);
        int j = 0;
        // full pairs only: both loads of the body are used unconditionally (a load whose only use sits under an
        // `if` is sunk into it by LLVM before scheduling, fences or not); an odd last node is peeled
        for (; j + 1 < n; j += 2) {
            load(a1, j + 1);
            __builtin_amdgcn_sched_barrier(0);
            node(a0, za, zb);
            __builtin_amdgcn_sched_barrier(0);
            load(a0, (j + 2 < n) ? j + 2 : j + 1);
            __builtin_amdgcn_sched_barrier(0);
            node(a1, zb, za);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (j < n) node(a0, za, zb);
    } else {
        // many fragments per node (ranks > 32): one multiply per MFMA on v T_j, as in round 2 (the product recurrence
        // keeps 2 RCX more doubles live and measured 2.5 % slower at rank 64)
        double tprev[NT], tcur[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) { tprev[nt] = 1.0; tcur[nt] = s[nt]; }
        for (int j = 0; j < n; ++j) {
            double a[F];
            load(a, j);
            double q[NT];                      // (tprev, tcur) = (T_j, T_{j+1}) on entry
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                q[nt] = tprev[nt];
                const double tn = __builtin_fma(2.0 * s[nt], tcur[nt], -tprev[nt]);
                tprev[nt] = tcur[nt];
                tcur[nt] = tn;
            }
#pragma unroll
            for (int c = 0; c < RCX; ++c) {
                double bop[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bop[nt] = v[nt][c] * q[nt];
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[c * RT + t], bop[nt], acc[nt][t], 0, 0, 0);
            }
        }
    }
}

// Packing for the direct form: dim 0 keeps its single left chunk, every later dim is
// padded to the kernel's compile-time RC chunks and RT tiles (zeros), so the node loop
// needs no rank predicates:  frag[dim][j][c][t][l] = G[a = 4c + (l>>4)][j][b = 16t + (l&15)].
struct TTRanks {
    int rc[PCX_MAX_DIMS];  // left-rank chunks of 4 stored per storage dim (1 for dim 0, RC after)
    int rt[PCX_MAX_DIMS];  // right-rank tiles of 16 stored per storage dim (RT)
};

// One wave owns 16*NT points; v (lane group g holds rows a = 4c + g) runs through the
// dimensions in registers.  The last dimension (right rank 1) is a VALU dot product against
// the plain core `glast` ([a][j]) split over the four lane groups.
template <int RC, int RT, int NT>
__global__ void __launch_bounds__(256, 2)
k_tt_eval_mfma(TTDims dims, TTRanks rk, const double *__restrict__ frag,
               const double *__restrict__ glast, int rl_last,
               const double *__restrict__ pts, double *__restrict__ out, long N) {
    static_assert(RC <= 4 * RT, "left chunks must fit the previous D tiles");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int c16 = lane & 15;
    const long base = ((long)blockIdx.x * 4 + wave) * (16 * NT);
    const int d = dims.d;

    // The wave's 16*NT query rows are one contiguous block of `pts`: copy it to LDS with
    // coalesced loads once, instead of a strided global load per dimension.
    extern __shared__ double lds_x[];
    double *xs = lds_x + (size_t)wave * (16 * NT) * d;
    {
        const long first = base * d;
        const long avail = (N - base) * (long)d;              // doubles that exist
        const int cnt = 16 * NT * d;
        for (int i = lane; i < cnt; i += 64) xs[i] = (i < avail) ? pts[first + i] : 0.0;
    }
    // small last cores (4 RC rows x n, zero padded) are copied behind the query rows
    double *gl_lds = lds_x + (size_t)4 * (16 * NT) * d;
    const bool last_in_lds = (rl_last != 0);       // the host passes 0 when the table does not fit
    if (last_in_lds)
        for (int i = threadIdx.x; i < 4 * RC * dims.n[d - 1]; i += 256) gl_lds[i] = glast[i];
    __syncthreads();

    double v[NT][RC];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int c = 0; c < RC; ++c) v[nt][c] = 0.0;

    if (d > 1) {
        // dimension 0: left rank 1 -> a single chunk whose only non-zero row is a = 0
        double v0[NT][1], s[NT];
        pcx_d4 acc[NT][RT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            v0[nt][0] = (g == 0) ? 1.0 : 0.0;
            double x = xs[(16 * nt + c16) * d + dims.col[0]];
            s[nt] = __builtin_fma(x - dims.lo[0], dims.scale[0], -1.0);   // tensor_train.py:2254, 2 / (hi - lo) from the host
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[nt][t] = (pcx_d4){0.0, 0.0, 0.0, 0.0};
        }
        tt_mfma_nodes<1, RT, NT>(frag + dims.frag_off[0] + lane, dims.n[0], v0, s, acc);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int c = 0; c < RC; ++c) v[nt][c] = acc[nt][c >> 2][c & 3];
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) v[nt][0] = (g == 0) ? 1.0 : 0.0;
    }

    for (int k = 1; k < d - 1; ++k) {
        const double lo = dims.lo[k], sck = dims.scale[k];
        double s[NT];
        pcx_d4 acc[NT][RT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double x = xs[(16 * nt + c16) * d + dims.col[k]];
            s[nt] = __builtin_fma(x - lo, sck, -1.0);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[nt][t] = (pcx_d4){0.0, 0.0, 0.0, 0.0};
        }
        tt_mfma_nodes<RC, RT, NT>(frag + dims.frag_off[k] + lane, dims.n[k], v, s, acc);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int c = 0; c < RC; ++c) v[nt][c] = acc[nt][c >> 2][c & 3];
    }

    // last dimension: y = sum_a v[a] * sum_j T_j(s) G[a][j]; lane group g owns a = 4c + g.
    // `glast` is the last core zero-padded to 4 RC rows ([a][j]): no rank predicate, one table read
    // per (j, c) shared by the NT column tiles; small tables are read from the LDS copy made above.
    {
        const int k = d - 1;
        const int n = dims.n[k];
        const double lo = dims.lo[k], sck = dims.scale[k];
        double sc[NT], tp[NT], tc[NT], w[NT][RC];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double x = xs[(16 * nt + c16) * d + dims.col[k]];
            sc[nt] = __builtin_fma(x - lo, sck, -1.0);
            tp[nt] = 1.0;
            tc[nt] = sc[nt];
#pragma unroll
            for (int c = 0; c < RC; ++c) w[nt][c] = 0.0;
        }
        // the table is read through a pointer of KNOWN address space (LDS copy or global): as one generic pointer the
        // loop was flat_load + s_waitcnt vmcnt(0) lgkmcnt(0) per node
        auto last_dim = [&](auto gl) {
            for (int j = 0; j < n; ++j) {
                double gv[RC];
#pragma unroll
                for (int c = 0; c < RC; ++c) gv[c] = gl[(4 * c + g) * n + j];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const double q = tp[nt];                   // T_j; (tp, tc) = (T_j, T_{j+1})
#pragma unroll
                    for (int c = 0; c < RC; ++c) w[nt][c] = __builtin_fma(q, gv[c], w[nt][c]);
                    const double tn = __builtin_fma(2.0 * sc[nt], tc[nt], -tp[nt]);
                    tp[nt] = tc[nt];
                    tc[nt] = tn;
                }
            }
        };
        typedef const double __attribute__((address_space(3))) *lds_cptr;
        typedef const double __attribute__((address_space(1))) *glb_cptr;
        if (last_in_lds) last_dim((lds_cptr)gl_lds);
        else last_dim((glb_cptr)glast);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            long p = base + 16 * nt + c16;
            double y = 0.0;
#pragma unroll
            for (int c = 0; c < RC; ++c) y = __builtin_fma(v[nt][c], w[nt][c], y);
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            if (g == 0 && p < N) out[p] = y;
        }
    }
}

// Ranks above 64 (outside the MFMA tilings): one wavefront walks one point at a time through
// the chain with the cores in their natural (r, n, r') layout -- lanes over the right rank b
// (coalesced core reads from L2), v and the polynomial values in a per-wave LDS slice.
// LDS per wave: 2 rmax + nmax doubles.
struct TTGeneric {
    int rank[PCX_MAX_DIMS + 1];
    long coff[PCX_MAX_DIMS];
    int rmax, nmax;
};

__global__ void __launch_bounds__(256)
k_tt_eval_generic(TTDims dims, TTGeneric gi, const double *__restrict__ cores,
                  const double *__restrict__ pts, double *__restrict__ out, long N) {
    extern __shared__ double lds_g[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double *va = lds_g + (size_t)wave * (2 * gi.rmax + gi.nmax);
    double *vb = va + gi.rmax;
    double *q = vb + gi.rmax;
    const int d = dims.d;
    for (long p = (long)blockIdx.x * 4 + wave; p < N; p += (long)gridDim.x * 4) {
        if (lane == 0) va[0] = 1.0;
        for (int k = 0; k < d; ++k) {
            const int rl = gi.rank[k], rr = gi.rank[k + 1], n = dims.n[k];
            const double x = pts[p * d + dims.col[k]];
            const double sc = 2.0 * (x - dims.lo[k]) / (dims.hi[k] - dims.lo[k]) - 1.0;   // tensor_train.py:2254
            if (lane == 0) {                  // T_0 .. T_{n-1} by the forward recurrence
                double tp = 1.0, tc = sc;
                for (int j = 0; j < n; ++j) {
                    q[j] = tp;
                    const double tn = __builtin_fma(2.0 * sc, tc, -tp);
                    tp = tc;
                    tc = tn;
                }
            }
            // wave-private LDS: operations of one wave execute in order, no barrier needed
            const double *G = cores + gi.coff[k];
            for (int b = lane; b < rr; b += 64) {
                double s = 0.0;
                for (int a = 0; a < rl; ++a) {
                    const double *ga = G + ((long)a * n) * rr + b;
                    double w = 0.0;
                    for (int j = 0; j < n; ++j) w = __builtin_fma(q[j], ga[(long)j * rr], w);
                    s = __builtin_fma(va[a], w, s);
                }
                vb[b] = s;
            }
            double *t = va; va = vb; vb = t;
        }
        if (lane == 0) out[p] = va[0];
    }
}

// =====================================================================================
// Small-rank form ("W first"): for ranks <= 12 the direct form above wastes half of every
// MFMA (r' rows padded to 16).  Here each dimension is ONE dense GEMM over the node index
//     W[(a,b), p] = sum_j G_k[a, j, b] * T_j(s_k(p))          rows = R*R, K = n_k -> KS*4
// (A = core viewed as an (R*R) x n matrix, B = the Chebyshev polynomial values alone),
// followed by the per-point contraction v'[b] = sum_a v[a] W[(a,b)] on the VALU:
//   D layout: lane (p = l&15, g = l>>4), tile t, reg i holds row m = 16t + g + 4i = 4u + g
//   with u = 4t + i, and with rows ordered m = a*R + b (R a multiple of 4):
//   a = u / (R/4), b = 4 (u % (R/4)) + g   ->  lane g owns outputs b = g, g+4, ..., the
//   products need only compile-time register indices, and one shuffle per b re-assembles
//   v' in every lane.  All packed cores live in LDS for the whole kernel.
// The last dimension (r' = 1) is a VALU dot product split over the four lane groups.
// Every dimension is zero-padded to the kernel's compile-time KS k-steps, so the whole
// batch body is straight-line code.
// =====================================================================================
struct TTWPlan {
    int ntiles[PCX_MAX_DIMS];   // row tiles stored: ceil(R/16) for dim 0 (left rank 1), R*R/16 after
    int lds_off[PCX_MAX_DIMS];  // offset (doubles) of dim k's block inside the LDS image
    int ks;                     // k-steps of 4 every dimension is padded to (= the kernel's KS)
    int total;                  // doubles in the LDS image
};

// image layout: for k < d-1: frag[k][s][t][64] = G_k[a][4s + (l>>4)][b], m = 16t + (l&15),
// a = m / R, b = m % R (zero outside the core); last dim: table [a < R][j < 4 ks], zero padded.
template <int R>
__global__ void k_tt_pack_wfirst(const double *__restrict__ G, double *__restrict__ img, int rl,
                                 int n, int rr, int ks, int ntiles, int is_last) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (is_last) {
        if (idx >= (long)R * 4 * ks) return;
        int a = (int)(idx / (4 * ks)), j = (int)(idx % (4 * ks));
        img[idx] = (a < rl && j < n) ? G[(long)a * n + j] : 0.0;   // (rl, n, 1) is [a][j]
        return;
    }
    long total = (long)ks * ntiles * 64;
    if (idx >= total) return;
    int l = (int)(idx & 63);
    int t = (int)((idx >> 6) % ntiles);
    int s = (int)((idx >> 6) / ntiles);
    int m = 16 * t + (l & 15);
    int a = m / R, b = m % R;
    int j = 4 * s + (l >> 4);
    img[idx] = (a < rl && b < rr && j < n) ? G[((long)a * n + j) * rr + b] : 0.0;
}

// One dimension of the W-first form with TL row tiles and KS k-steps (both compile-time):
//   acc[nt][t] = sum_s mfma(frag[s][t], B = T_{4s + g}(x))
// Lane group g needs only every fourth polynomial, T_g, T_{4+g}, T_{8+g}, ...: they obey
// the stride-4 Chebyshev recurrence T_{m+4} = 2 T_4 T_m - T_{|m-4|}, so after two selects
// per dimension a k-step costs ONE fma and no lane-dependent select.  Polynomials beyond
// the node count multiply zero-padded core rows, so they need no masking.
template <int TL, int KS, int NT>
__device__ __forceinline__ void tt_w_gemm(const double *fk, int g, const double (&sc)[NT],
                                          pcx_d4 (&acc)[NT][TL]) {
    double a[KS][TL];
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < TL; ++t) a[s][t] = fk[(size_t)(s * TL + t) * 64];
    double u[NT][KS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const double x = sc[nt];
        const double x2 = 2.0 * x;
        const double t2 = __builtin_fma(x2, x, -1.0);
        const double t3 = __builtin_fma(x2, t2, -x);
        const double t4 = __builtin_fma(x2, t3, -t2);
        const double tg = (g == 0) ? 1.0 : (g == 1) ? x : (g == 2) ? t2 : t3;      // T_g
        const double tm = (g == 0) ? t4 : (g == 1) ? t3 : (g == 2) ? t2 : x;       // T_{4-g}
        const double c4 = 2.0 * t4;
        u[nt][0] = tg;
        if constexpr (KS > 1) u[nt][1] = __builtin_fma(c4, tg, -tm);                // T_{4+g}
#pragma unroll
        for (int s = 2; s < KS; ++s) u[nt][s] = __builtin_fma(c4, u[nt][s - 1], -u[nt][s - 2]);
    }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int t = 0; t < TL; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                    a[s][t], u[nt][s], s == 0 ? (pcx_d4){0.0, 0.0, 0.0, 0.0} : acc[nt][t], 0, 0, 0);
}

// Persistent workgroups of 4 waves; a wave owns 16*NT points per batch.  The query rows of
// the NEXT batch are fetched (coalesced) while the current one is computed, and are
// affinely mapped to [-1, 1] -- the reference's 2 (x - a) / (b - a) - 1 with its IEEE
// division -- on their way into the wave's LDS slice, one element per lane instead of one
// division per lane and dimension.  __launch_bounds__(256, 2): at most 256 registers per
// lane, which lets the compiler keep the MFMA accumulators in ordinary VGPRs (no
// v_accvgpr_read traffic in front of the VALU fold).
template <int R, int KS, int NT>
__global__ void __launch_bounds__(256, 2)
k_tt_eval_wfirst(TTDims dims, TTWPlan plan, const double *__restrict__ img,
                 const double *__restrict__ pts, double *__restrict__ out, long N) {
    constexpr int TILES = R * R / 16 > 0 ? R * R / 16 : 1;
    constexpr int RB = R / 4;                       // outputs b owned per lane group
    constexpr int PF = 4 * NT;                      // staged elements per lane: 16 NT d / 64, d <= 16
    constexpr int NP = 4 * KS;                      // padded node count
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4;
    const int c16 = lane & 15;
    const int d = dims.d;
    const int cnt = 16 * NT * d;                    // doubles in a wave's block of query rows
    // the packed cores are loaded ONCE per workgroup; the workgroup then walks over many
    // batches of 4 x 16*NT points (grid-stride), so this prologue is amortised
    for (int i = threadIdx.x; i < plan.total; i += 256) lds[i] = img[i];
    // element i of a wave's block is column i % d: table of its storage dimension's bounds
    double *lo_t = lds + plan.total, *wd_t = lo_t + cnt;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int c = i % d;
        double lo = 0.0, wd = 1.0;
        for (int kk = 0; kk < d; ++kk)
            if (dims.col[kk] == c) { lo = dims.lo[kk]; wd = dims.hi[kk] - dims.lo[kk]; }
        lo_t[i] = lo;
        wd_t[i] = wd;
    }
    double *xs = wd_t + cnt + (size_t)wave * cnt;
    const long nbatch = (N + 4 * 16 * NT - 1) / (4 * 16 * NT);
    double pf[PF];
    auto fetch = [&](long batch) {
        const long base = (batch * 4 + wave) * (16 * NT);
        const long first = base * d;
        const long avail = (N - base) * (long)d;
#pragma unroll
        for (int r = 0; r < PF; ++r) {
            const int i = lane + 64 * r;
            pf[r] = (i < cnt && i < avail) ? pts[first + i] : 0.0;
        }
    };
    if ((long)blockIdx.x < nbatch) fetch(blockIdx.x);
    __syncthreads();

  for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
    const long base = (batch * 4 + wave) * (16 * NT);
    // wave-private LDS slice: LDS operations of one wave execute in order, no barrier needed
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        const int i = lane + 64 * r;
        if (i < cnt) xs[i] = 2.0 * (pf[r] - lo_t[i]) / wd_t[i] - 1.0;       // tensor_train.py:2254
    }
    if (batch + gridDim.x < nbatch) fetch(batch + gridDim.x);

    double v[NT][R];       // full left vector, replicated in the four lane groups
    double vown[NT][RB];   // the entries a = 4 bi + g this lane group owns
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
        for (int a = 0; a < R; ++a) v[nt][a] = 0.0;
#pragma unroll
        for (int bi = 0; bi < RB; ++bi) vown[nt][bi] = 0.0;
        v[nt][0] = 1.0;
        vown[nt][0] = (g == 0) ? 1.0 : 0.0;
    }
    // one shuffle per b re-assembles v' in every lane from the owners' partial vectors
    auto spread = [&](int nt) {
#pragma unroll
        for (int b = 0; b < R; ++b) v[nt][b] = __shfl(vown[nt][b >> 2], ((b & 3) << 4) | c16, 64);
    };

    if (d > 1) {   // dimension 0 (left rank 1): rows a = 0 only -> v'[b] = W[(0, b)]
        constexpr int TL0 = (R + 15) / 16;
        static_assert(TL0 == 1, "R <= 16");
        double sc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sc[nt] = xs[(16 * nt + c16) * d + dims.col[0]];
        pcx_d4 acc[NT][TL0];
        tt_w_gemm<TL0, KS, NT>(lds + plan.lds_off[0] + lane, g, sc, acc);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) vown[nt][bi] = acc[nt][0][bi];
            spread(nt);
        }
    }
    for (int k = 1; k < d - 1; ++k) {
        double sc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sc[nt] = xs[(16 * nt + c16) * d + dims.col[k]];
        pcx_d4 acc[NT][TILES];
        tt_w_gemm<TILES, KS, NT>(lds + plan.lds_off[k] + lane, g, sc, acc);
        // fold W into v: v'[b] = sum_a v[a] W[(a,b)] for the b this lane group owns
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double pb[RB];
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) pb[bi] = 0.0;
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int u = 4 * t + i;
                    pb[u % RB] = __builtin_fma(v[nt][u / RB], acc[nt][t][i], pb[u % RB]);
                }
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) vown[nt][bi] = pb[bi];
            if (k < d - 2) spread(nt);
        }
    }

    // last dimension: y = sum_a v[a] * sum_j T_j(s) G[a][j]; lane group g takes a = g, g+4, ...
    {
        const double *gl = lds + plan.lds_off[d - 1] + g * NP;     // rows a = 4 bi + g
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const long p = base + 16 * nt + c16;
            const double sc = xs[(16 * nt + c16) * d + dims.col[d - 1]];
            double w[RB];
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) w[bi] = 0.0;
            double tp = 1.0, tc = sc;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const double q = tp;                       // T_j; (tp, tc) = (T_j, T_{j+1})
#pragma unroll
                for (int bi = 0; bi < RB; ++bi) w[bi] = __builtin_fma(q, gl[bi * 4 * NP + j], w[bi]);
                const double tn = __builtin_fma(2.0 * sc, tc, -tp);
                tp = tc;
                tc = tn;
            }
            double y = 0.0;
#pragma unroll
            for (int bi = 0; bi < RB; ++bi) y = __builtin_fma(vown[nt][bi], w[bi], y);
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            if (g == 0 && p < N) out[p] = y;
        }
    }
  }   // batch loop
}

// =====================================================================================
// Small-rank direct form on v_mfma_f64_4x4x4_4b_f64 (ranks <= 12, n <= 16): the default for
// small models since round 2.
//     v'[b, p] = sum_{(j, c)} sum_k G[a = 4c + k, j, b] * z_j[c][k, p],   z_j[c] = v[4c + k, p] T_j(x_p)
//   four blocks = the four 4-point groups of a wave's 16 points
//   A lane = 16 k + 4 blk + i : G[a = 4c + k][j][b = 4m + i], identical in the four blocks -> a
//                               broadcast LDS read brings the RA output chunks m of one (j, c)
//   B lane = 16 k + 4 blk + jj: z_j[c] of point 4 blk + jj, left-rank row k
//   D lane = 16 i + 4 blk + jj: v'[4m + i] of that point -- exactly the row (k := i) the lane
//                               needs as B operand of left chunk c = m in the next dimension:
//                               the chain runs with NO cross-lane movement and no fold.
// The B operands come from the Chebyshev recurrence applied to the PRODUCTS,
//     z_{j+1} = 2 x z_j - z_{j-1},  z_0 = v,  z_{-1} = x v     (T_{-1} = T_1),
// one FMA per (j, c): no table of T_j and no per-MFMA multiply.  Against the W-first form:
// no padding of 8 rows to 16, none of n to a multiple of 4 (138 instead of 150 16-cycle
// instructions for ranks [1,8,8,8,6,1], n = 11), 180 instead of 230 other vector instructions
// per 16 points, and 80 VGPRs (6 waves per SIMD).  What bounds it (profiles/r02_tt5d_*): FP64
// MFMA and FP64/other VALU do not overlap on gfx950 -- measured cycles per 16-point batch per
// SIMD = SQ_VALU_MFMA_BUSY_CYCLES + SQ_ACTIVE_INST_VALU for every form tried
// (tools/tt_wfirst_lab.hip).
// Dimension 0 (left rank 1) is a GEMM over the nodes alone (B = T_{4s+k}(x), stride-4
// recurrence seeded through a 6-double LDS record per point, so that each lane group picks
// T_k and T_{4-k} by ADDRESS, not by selects); the last dimension (right rank 1) is a VALU dot
// product split over the four lane groups.  Node counts are dispatched to fully unrolled
// bodies so that the LDS reads run ahead of the MFMAs.
// image: dim 0: [s][16 slots][NMP]; mid dims: [j][c < RA][16 slots][NMP] with slot = 4 k + i and
//        the RA values m of a slot contiguous (NMP = 1, 2, 4 doubles: one aligned read);
//        last dim: [4 RA rows a][n] zero padded.
// =====================================================================================
typedef double pcx_d2 __attribute__((ext_vector_type(2)));

template <int RA> struct D4Frag;
template <> struct D4Frag<1> {
    double v[1];
    static constexpr int NMP = 1;
    __device__ __forceinline__ void load(const double *p) { v[0] = p[0]; }
};
template <> struct D4Frag<2> {
    double v[2];
    static constexpr int NMP = 2;
    __device__ __forceinline__ void load(const double *p) { const pcx_d2 t = *(const pcx_d2 *)p; v[0] = t.x; v[1] = t.y; }
};
template <> struct D4Frag<3> {
    double v[3];
    static constexpr int NMP = 4;
    __device__ __forceinline__ void load(const double *p) { const pcx_d2 t = *(const pcx_d2 *)p; v[0] = t.x; v[1] = t.y; v[2] = p[2]; }
};

struct TTD4Plan {
    int lds_off[PCX_MAX_DIMS];  // offset (doubles) of dim k's block inside the LDS image
    int total;                  // doubles in the LDS image
};

#define PCX_D4_MAX_NODES 16

template <int RA, int NJ>
__device__ __forceinline__ void tt_d4_mid(const double *fk, double x, double (&v)[RA]) {
    constexpr int NMP = D4Frag<RA>::NMP;
    const double x2 = x + x;
    double zc[RA], zp[RA], acc[RA];
#pragma unroll
    for (int c = 0; c < RA; ++c) { zc[c] = v[c]; zp[c] = v[c] * x; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int c = 0; c < RA; ++c) {
            D4Frag<RA> a;
            a.load(fk + (j * RA + c) * 16 * NMP);
#pragma unroll
            for (int m = 0; m < RA; ++m)
                acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[m], zc[c], (j == 0 && c == 0) ? 0.0 : acc[m], 0, 0, 0);
        }
        if (j + 1 < NJ) {
#pragma unroll
            for (int c = 0; c < RA; ++c) { const double zn = __builtin_fma(x2, zc[c], -zp[c]); zp[c] = zc[c]; zc[c] = zn; }
        }
    }
#pragma unroll
    for (int m = 0; m < RA; ++m) v[m] = acc[m];
}

// y_partial = sum_c v[c] * sum_j T_j(x) G[a = 4c + g][j]; gl -> row g of the [4 RA][NJ] table
template <int RA, int NJ>
__device__ __forceinline__ double tt_d4_last(const double *gl, double x, const double (&v)[RA]) {
    const double x2 = x + x;
    double w[RA], tp = 1.0, tc = x;
#pragma unroll
    for (int c = 0; c < RA; ++c) w[c] = 0.0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int c = 0; c < RA; ++c) w[c] = __builtin_fma(tp, gl[c * 4 * NJ + j], w[c]);
        if (j + 1 < NJ) { const double tn = __builtin_fma(x2, tc, -tp); tp = tc; tc = tn; }
    }
    double y = v[0] * w[0];
#pragma unroll
    for (int c = 1; c < RA; ++c) y = __builtin_fma(v[c], w[c], y);
    return y;
}

#define PCX_D4_NODE_SWITCH(n, CALL)                                                                     \
    switch (n) {                                                                                        \
    case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break;     \
    case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break;     \
    case 9: CALL(9); break; case 10: CALL(10); break; case 11: CALL(11); break; case 12: CALL(12); break; \
    case 13: CALL(13); break; case 14: CALL(14); break; case 15: CALL(15); break; default: CALL(16); break; }

// Persistent workgroups of 4 waves; a wave owns 16 points per batch.  The query rows of the
// NEXT batch are fetched (coalesced) while the current one is computed and mapped to [-1, 1]
// on their way into the wave's LDS slice: s = (x - a) * (2 / (b - a)) - 1 in one subtraction and
// one FMA (the reference divides, 2 (x - a) / (b - a) - 1: the two differ by at most 2 ulp of s,
// far inside the TT tolerance; the division sequence cost 70 FP64 instructions per batch).
template <int RA>
__global__ void __launch_bounds__(256, 4)
k_tt_eval_d4(TTDims dims, TTD4Plan plan, const double *__restrict__ img, const double *__restrict__ pts,
             double *__restrict__ out, long N) {
    constexpr int NMP = D4Frag<RA>::NMP;
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int d = dims.d;
    const int cnt = 16 * d;                         // doubles in a wave's block of query rows
    for (int i = threadIdx.x; i < plan.total; i += 256) lds[i] = img[i];
    // element i of a wave's block is column i % d: table of its storage dimension's bounds
    double *lo_t = lds + plan.total, *wd_t = lo_t + cnt;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int c = i % d;
        double lo = 0.0, wd = 1.0;
        for (int kk = 0; kk < d; ++kk)
            if (dims.col[kk] == c) { lo = dims.lo[kk]; wd = dims.hi[kk] - dims.lo[kk]; }
        lo_t[i] = lo;
        wd_t[i] = 2.0 / wd;
    }
    double *xs = wd_t + cnt + (size_t)wave * (cnt + 16 * 6);
    double *seed = xs + cnt + c16 * 6;              // this lane's point: {T_0 .. T_4, 2 T_4}(x_0)
    const int a_idx = ((lane >> 4) * 4 + (lane & 3)) * NMP;
    const long nbatch = (N + 63) / 64;
    constexpr int PFMAX = 4;                        // 16 d / 64 staged elements per lane, d <= 16
    const int npf = (cnt + 63) >> 6;
    double pf[PFMAX];
    auto fetch = [&](long batch) {
        const long base = (batch * 4 + wave) * 16;
        const long first = base * d, avail = (N - base) * (long)d;
#pragma unroll
        for (int r = 0; r < PFMAX; ++r)
            if (r < npf) {
                const int i = lane + 64 * r;
                pf[r] = (i < cnt && i < avail) ? pts[first + i] : 0.0;
            }
    };
    if ((long)blockIdx.x < nbatch) fetch(blockIdx.x);
    __syncthreads();
    const int ks0 = (dims.n[0] + 3) >> 2;

    for (long batch = blockIdx.x; batch < nbatch; batch += gridDim.x) {
        const long base = (batch * 4 + wave) * 16;
        // wave-private LDS slice: LDS operations of one wave execute in order, no barrier needed
#pragma unroll
        for (int r = 0; r < PFMAX; ++r)
            if (r < npf) {
                const int i = lane + 64 * r;
                if (i < cnt) xs[i] = __builtin_fma(pf[r] - lo_t[i], wd_t[i], -1.0);
            }
        if (batch + gridDim.x < nbatch) fetch(batch + gridDim.x);
        double v[RA];
        if (d > 1) {
            const double x = xs[c16 * d + dims.col[0]];
            const double x2 = x + x, t2 = __builtin_fma(x2, x, -1.0), t3 = __builtin_fma(x2, t2, -x),
                         t4 = __builtin_fma(x2, t3, -t2);
            // the compiler must not move the reads below above these writes (the hardware keeps
            // a wave's LDS operations in order)
            asm volatile("" ::: "memory");
            if (g == 0) {
                *(pcx_d2 *)(seed) = (pcx_d2){1.0, x};
                *(pcx_d2 *)(seed + 2) = (pcx_d2){t2, t3};
                *(pcx_d2 *)(seed + 4) = (pcx_d2){t4, t4 + t4};
            }
            asm volatile("" ::: "memory");
            double up = seed[4 - g], uc = seed[g];  // T_{4-g} = T_{|g-4|}, T_g
            const double c4 = seed[5];
            const double *f0 = lds + plan.lds_off[0] + a_idx;
            double acc[RA];
#pragma unroll
            for (int m = 0; m < RA; ++m) acc[m] = 0.0;
            for (int s = 0; s < ks0; ++s) {
                D4Frag<RA> a;
                a.load(f0 + s * 16 * NMP);
#pragma unroll
                for (int m = 0; m < RA; ++m) acc[m] = __builtin_amdgcn_mfma_f64_4x4x4f64(a.v[m], uc, acc[m], 0, 0, 0);
                const double un = __builtin_fma(c4, uc, -up);   // T_{4(s+1)+g} = 2 T_4 T_{4s+g} - T_{|4(s-1)+g|}
                up = uc;
                uc = un;
            }
#pragma unroll
            for (int m = 0; m < RA; ++m) v[m] = acc[m];
        } else {
#pragma unroll
            for (int m = 0; m < RA; ++m) v[m] = (m == 0 && g == 0) ? 1.0 : 0.0;
        }
        for (int k = 1; k < d - 1; ++k) {
            const double x = xs[c16 * d + dims.col[k]];
            const double *fk = lds + plan.lds_off[k] + a_idx;
#define PCX_D4_MID(NJ) tt_d4_mid<RA, NJ>(fk, x, v)
            PCX_D4_NODE_SWITCH(dims.n[k], PCX_D4_MID)
#undef PCX_D4_MID
        }
        {
            const int nl = dims.n[d - 1];
            const double *gl = lds + plan.lds_off[d - 1] + g * nl;
            const double x = xs[c16 * d + dims.col[d - 1]];
            double y;
#define PCX_D4_LAST(NJ) y = tt_d4_last<RA, NJ>(gl, x, v)
            PCX_D4_NODE_SWITCH(nl, PCX_D4_LAST)
#undef PCX_D4_LAST
            y += __shfl_xor(y, 16, 64);
            y += __shfl_xor(y, 32, 64);
            const long p = base + c16;
            if (g == 0 && p < N) out[p] = y;
        }
    }
}
