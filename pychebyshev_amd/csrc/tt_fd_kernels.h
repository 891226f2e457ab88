// tt_fd_kernels.h -- finite-difference Greeks of a tensor train on the device (gfx950), round 4.
//
// Replaces the host-side stencil columns of ChebyshevTT.eval_multi_batch; the rules are the reference's
// (tensor_train.py:2322-2463, eval_multi + _fd_*), in its order of operations, in the STORAGE frame:
//   h = (b - a) 1e-4 per dimension; before a dimension is differenced its coordinate is nudged so that
//   1.5 h stays inside the domain:  if (x - a < 1.5 h) x = a + 1.5 h;  if (b - x < 1.5 h) x = b - 1.5 h
//   exactly two active dimensions, both of order 1 (a mixed partial): both nudged, then the four-point rule
//       (f(+,+) - f(+,-) - f(-,+) + f(-,-)) / (4 h1 h2)
//   otherwise nested, first active dimension outermost:
//       order 1: (F(x + h) - F(x - h)) / (2 h)          order 2: (F(x + h) - 2 F(x) + F(x - h)) / (h h)
//   with F the same rule over the remaining active dimensions (the value itself when none is left).
//
// k_tt_fd_lpp<RCAP,NJ>  (models whose evaluation kernel is the lane-per-point form, tt_lpp_kernels.h): lane = point;
//   the lane forms every stencil point of every spec in registers -- only the coordinates of the active dimensions
//   differ from the query row -- walks the chain for it (the same tt_lpp_dim bodies, so a stencil value is
//   bit-identical to eval_batch at that point) and folds the values through the rule as they arrive.  No
//   (N x stencil, d) batch ever exists in HBM: traffic is the query rows once per stencil point (L2 hits) and
//   8 m bytes of results per point.  Spec and stencil loops are wave-uniform (scalar branches).
// k_tt_fd_points / k_tt_fd_combine  (any other model: ranks >= 16, n > 16): the stencil batch of a chunk of
//   points is written to HBM, evaluated by the model's own kernel, and combined -- same rules, same order.
#pragma once

#include "tt_lpp_kernels.h"

#define PCX_FD_MAX_ACTIVE 3       // differenced dimensions per spec on the device (27 stencil points at most)
#define PCX_FD_PACK 16            // specs per launch (passed by value in the kernel arguments)

struct TTFdSpec {
    int kind;                       // 0 value, 1 nested rule, 2 four-point mixed partial
    int nact;                       // differenced dimensions
    int nleaf;                      // stencil points
    int slot0;                      // first stencil slot of this spec in a materialised batch (generic path)
    int dim[PCX_FD_MAX_ACTIVE];     // storage dimension, ascending
    int order[PCX_FD_MAX_ACTIVE];   // 1 or 2
    int col[PCX_FD_MAX_ACTIVE];     // user column of that storage dimension
    int pad_[3];
    double lo[PCX_FD_MAX_ACTIVE], hi[PCX_FD_MAX_ACTIVE];
    double h[PCX_FD_MAX_ACTIVE];    // (hi - lo) * 1e-4
    double need[PCX_FD_MAX_ACTIVE]; // h * 1.5
};

struct TTFdPack {
    int m;                          // specs in this launch
    int slots;                      // stencil slots of all of them (generic path)
    TTFdSpec s[PCX_FD_PACK];
};

__device__ __forceinline__ double fd_nudge(double x, double lo, double hi, double need) {
    if (x - lo < need) x = lo + need;            // tensor_train.py:2331-2339 (NaN compares false: unchanged)
    if (hi - x < need) x = hi - need;
    return x;
}

// digit of level i of stencil point `leaf` (level 0 outermost / slowest); a level of order o has o + 1 children:
// order 1: +h, -h;  order 2: +h, centre, -h.  The four-point rule enumerates (+,+), (+,-), (-,+), (-,-): the same digits.
__device__ __forceinline__ void fd_digits(const TTFdSpec &sp, int leaf, int (&dig)[PCX_FD_MAX_ACTIVE]) {
    int rem = leaf;
#pragma unroll
    for (int i = PCX_FD_MAX_ACTIVE - 1; i >= 0; --i) {
        dig[i] = 0;
        if (i < sp.nact) {
            const int c = sp.order[i] + 1;
            dig[i] = rem % c;
            rem /= c;
        }
    }
}

// coordinate of active dimension i at digit g, from the nudged base xb
__device__ __forceinline__ double fd_coord(double xb, double h, int order, int g) {
    if (g == 0) return xb + h;
    if (order == 2 && g == 1) return xb;
    return xb + (-h);
}

// Folds the value of stencil point `leaf` into the rule.  cl = the lane's column of child values (stride PCX_LPP_WG
// doubles, 3 per level); returns true when `result` is the spec's final value.
__device__ __forceinline__ bool fd_push(const TTFdSpec &sp, int leaf, const int (&dig)[PCX_FD_MAX_ACTIVE], double val,
                                        double *cl, double &result) {
#pragma clang fp contract(off)
    if (sp.kind == 0) { result = val; return true; }
    if (sp.kind == 2) {
        cl[leaf * PCX_LPP_WG] = val;
        if (leaf < 3) return false;
        const double fpp = cl[0], fpm = cl[PCX_LPP_WG], fmp = cl[2 * PCX_LPP_WG], fmm = cl[3 * PCX_LPP_WG];
        result = (fpp - fpm - fmp + fmm) / (4.0 * sp.h[0] * sp.h[1]);
        return true;
    }
    double r = val;
    for (int lvl = sp.nact - 1; lvl >= 0; --lvl) {
        double *c = cl + (size_t)lvl * 3 * PCX_LPP_WG;
        c[dig[lvl] * PCX_LPP_WG] = r;
        if (dig[lvl] != sp.order[lvl]) return false;            // more children of this level to come
        const double h = sp.h[lvl];
        if (sp.order[lvl] == 1) r = (c[0] - c[PCX_LPP_WG]) / (2.0 * h);
        else r = (c[0] - 2.0 * c[PCX_LPP_WG] + c[2 * PCX_LPP_WG]) / (h * h);
    }
    result = r;
    return true;
}

// dynamic LDS = (vrows + 3 * PCX_FD_MAX_ACTIVE + 1) * 64 * 8 bytes: the chain's v' column, then the child values
template <int RCAP, int NJ>
__global__ void __launch_bounds__(PCX_LPP_WG, RCAP <= 8 ? 6 : 4)
k_tt_fd_lpp(const TTLppDim *__restrict__ tab, int d, int vrows, const double *__restrict__ img,
            const double *__restrict__ pts, double *__restrict__ out, long N, long ostride, long ooff, TTFdPack pack) {
    extern __shared__ double lds_fd[];
    double *vl = lds_fd + threadIdx.x;
    double *cl = lds_fd + (size_t)vrows * PCX_LPP_WG + threadIdx.x;
    typedef const TTLppDim __attribute__((address_space(4))) *tab_cptr;
    const tab_cptr ct = (tab_cptr)(unsigned long long)tab;
    const pcx_lpp_cptr cimg = (pcx_lpp_cptr)(unsigned long long)img;
    const long p = (long)blockIdx.x * PCX_LPP_WG + threadIdx.x;
    const long pc = p < N ? p : N - 1;
    const double *row = pts + pc * d;
    for (int s = 0; s < pack.m; ++s) {
        const TTFdSpec &sp = pack.s[s];
        double xb[PCX_FD_MAX_ACTIVE];
#pragma unroll
        for (int i = 0; i < PCX_FD_MAX_ACTIVE; ++i) {
            xb[i] = 0.0;
            if (i < sp.nact) xb[i] = fd_nudge(row[sp.col[i]], sp.lo[i], sp.hi[i], sp.need[i]);
        }
        double result = 0.0;
        for (int leaf = 0; leaf < sp.nleaf; ++leaf) {
            int dig[PCX_FD_MAX_ACTIVE];
            fd_digits(sp, leaf, dig);
            double xl[PCX_FD_MAX_ACTIVE];
#pragma unroll
            for (int i = 0; i < PCX_FD_MAX_ACTIVE; ++i) xl[i] = i < sp.nact ? fd_coord(xb[i], sp.h[i], sp.order[i], dig[i]) : 0.0;
            vl[0] = 1.0;
            // the coordinate of dimension k + 1 is fetched (and replaced, where that dimension is differenced) while
            // dimension k is contracted, as in k_tt_eval_lpp
            auto coord = [&](int k) {
                double xu = row[ct[k].col];
#pragma unroll
                for (int i = 0; i < PCX_FD_MAX_ACTIVE; ++i)
                    if (i < sp.nact && sp.dim[i] == k) xu = xl[i];
                return xu;
            };
            double xn = coord(0);
            for (int k = 0; k < d; ++k) {
                const double x = __builtin_fma(xn - ct[k].lo, ct[k].scale, -1.0);    // as k_tt_eval_lpp
                if (k + 1 < d) xn = coord(k + 1);
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
            (void)fd_push(sp, leaf, dig, vl[0], cl, result);
        }
        if (p < N) out[p * ostride + ooff + s] = result;
    }
}

// ---- generic path: materialise, evaluate with the model's own kernel, combine ---------------------------------
// batch row (slot * N + p) = query row p with the active coordinates of stencil slot `slot` (user column order kept)
__global__ void __launch_bounds__(256)
k_tt_fd_points(const double *__restrict__ pts, long N, int d, double *__restrict__ batch, TTFdPack pack) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int slot = blockIdx.y;
    if (p >= N) return;
    int s = 0;
    while (s + 1 < pack.m && pack.s[s + 1].slot0 <= slot) ++s;
    const TTFdSpec &sp = pack.s[s];
    int dig[PCX_FD_MAX_ACTIVE];
    fd_digits(sp, slot - sp.slot0, dig);
    const double *row = pts + p * d;
    double *dst = batch + ((long)slot * N + p) * d;
    for (int c = 0; c < d; ++c) {
        double x = row[c];
#pragma unroll
        for (int i = 0; i < PCX_FD_MAX_ACTIVE; ++i)
            if (i < sp.nact && sp.col[i] == c)
                x = fd_coord(fd_nudge(x, sp.lo[i], sp.hi[i], sp.need[i]), sp.h[i], sp.order[i], dig[i]);
        dst[c] = x;
    }
}

// out[p][ooff + s] from vals[(slot0_s + leaf) * N + p]; 64 threads per workgroup, child values in LDS as above
__global__ void __launch_bounds__(PCX_LPP_WG)
k_tt_fd_combine(const double *__restrict__ vals, long N, double *__restrict__ out, long ostride, long ooff, TTFdPack pack) {
    __shared__ double cl_s[(3 * PCX_FD_MAX_ACTIVE + 1) * PCX_LPP_WG];
    double *cl = cl_s + threadIdx.x;
    const long p = (long)blockIdx.x * PCX_LPP_WG + threadIdx.x;
    const long pc = p < N ? p : N - 1;
    for (int s = 0; s < pack.m; ++s) {
        const TTFdSpec &sp = pack.s[s];
        double result = 0.0;
        for (int leaf = 0; leaf < sp.nleaf; ++leaf) {
            int dig[PCX_FD_MAX_ACTIVE];
            fd_digits(sp, leaf, dig);
            (void)fd_push(sp, leaf, dig, vals[(long)(sp.slot0 + leaf) * N + pc], cl, result);
        }
        if (p < N) out[p * ostride + ooff + s] = result;
    }
}
