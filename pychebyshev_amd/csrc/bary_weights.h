// bary_weights.h -- division-free barycentric weights shared by the MFMA kernels (bary_kernels.h, bary_grid_kernels.h).
#pragma once

#include "pcx_common.h"

// b_j = w_j prod_{i != j} t_i / sum_k w_k prod_{i != k} t_i from running prefix and suffix products, one division per
// dimension; within 1e-14 of a node the first such node's slice, as the reference (barycentric.py:1039-1043).
// For dimensions of up to 64 nodes (the products stay far from over / underflow).  A short plan's prologue is a large
// part of its wave: 21^3 spends ~1,500 vector instructions on 63 IEEE divisions per point against 396 matrix instructions.
__device__ __forceinline__ void grid_weights_prod(double x, double scale, const double *__restrict__ snodes,
                                                  const double *__restrict__ wts, int n, double *dst, int stride) {
    double run = 1.0, amin = 1.0e300;
#pragma unroll 1
    for (int j = 0; j < n; ++j) {
        dst[j * stride] = wts[j] * run;
        const double t = __builtin_fma(x, scale, -snodes[j]);
        amin = __builtin_fmin(amin, __builtin_fabs(t));
        run *= t;
    }
    run = 1.0;
    double su = 0.0;
#pragma unroll 1
    for (int j = n - 1; j >= 0; --j) {
        const double cj = dst[j * stride] * run;
        dst[j * stride] = cj;
        su += cj;
        run *= __builtin_fma(x, scale, -snodes[j]);
    }
    const double r = 1.0 / su;
#pragma unroll 1
    for (int j = 0; j < n; ++j) dst[j * stride] *= r;
    if (amin < 1e-14 * scale) {
        bool found = false;
#pragma unroll 1
        for (int j = 0; j < n; ++j) {
            const bool hit = !found && __builtin_fabs(__builtin_fma(x, scale, -snodes[j])) < 1e-14 * scale;
            dst[j * stride] = hit ? 1.0 : 0.0;
            found = found || hit;
        }
    }
}

