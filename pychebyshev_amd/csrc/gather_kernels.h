// gather_kernels.h -- the column gather shared by the barycentric dim-q groups (pcx_bary.hip) and the slider
// (pcx_spline.hip).  `static`: each translation unit that includes it carries its own copy.
#pragma once

#include "pcx_common.h"

struct SliderCols {
    int nc;
    int col[PCX_MAX_DIMS];
};

// out[p][c] = pts[p][cols.col[c]]  (the column block a slide reads, packed)
static __global__ void k_gather_columns(const double *__restrict__ pts, long N, int d, SliderCols cols,
                                 double *__restrict__ out) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * cols.nc) return;
    const long p = e / cols.nc;
    const int c = (int)(e - p * cols.nc);
    out[e] = pts[p * d + cols.col[c]];
}
