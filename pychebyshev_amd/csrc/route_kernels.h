// route_kernels.h -- routing kernels of the piecewise interpolant (ChebyshevSpline.eval_batch, reference
// spline.py:633-700) and the composition kernels of the slider (reference slider.py:247-318).  gfx950.
#pragma once

#include "pcx_common.h"
#include "gather_kernels.h"

// ---------------------------------------------------------------------------------
// Piecewise interpolants (reference spline.py:633-700, ChebyshevSpline.eval_batch):
// route every point to its piece, bucket the points, then run the barycentric kernel once
// per non-empty piece on that piece's bucket (its `perm` argument).
// ---------------------------------------------------------------------------------
struct SplineDims {
    int d;
    int nknots[PCX_MAX_DIMS];   // knots per dimension
    int koff[PCX_MAX_DIMS];     // offset of dimension k's knots in knots_cat
    int shape[PCX_MAX_DIMS];    // pieces per dimension = nknots + 1
};

// One atomic per wave and distinct key instead of one per lane: the lanes of a wave that hold the same
// key elect a leader, the leader adds their count, every lane gets base + its rank among them.  (With
// 10^6 points and two pieces the per-lane form serialised a million atomics on two addresses: 10 ms.)
__device__ __forceinline__ int wave_grouped_add(int *__restrict__ counters, int key, bool active) {
    const int lane = (int)__lane_id();
    unsigned long long todo = __ballot(active);
    int slot = 0;
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int k = __shfl(key, leader);
        const bool mine = active && key == k;
        const unsigned long long same = __ballot(mine);
        int base = 0;
        if (lane == leader) base = atomicAdd(&counters[k], __popcll(same));
        base = __shfl(base, leader);
        if (mine) slot = base + __popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return slot;
}

// piece index = ravel_multi_index(clip(searchsorted(knots_k, x_k, side='right'), 0, shape_k - 1)):
// the number of knots <= x (a point exactly on a knot belongs to the piece on its right);
// NaN sorts after every knot in NumPy, i.e. lands in the last piece.
__device__ __forceinline__ int spline_flat_piece(const SplineDims &sd, const double *__restrict__ knots,
                                                 const double *__restrict__ pt) {
    int flat = 0;
    for (int k = 0; k < sd.d; ++k) {
        const double x = pt[k];
        int idx = 0;
        if (x != x) idx = sd.nknots[k];
        else
            for (int j = 0; j < sd.nknots[k]; ++j) idx += (knots[sd.koff[k] + j] <= x) ? 1 : 0;
        if (idx > sd.shape[k] - 1) idx = sd.shape[k] - 1;
        flat = flat * sd.shape[k] + idx;
    }
    return flat;
}

#define PCX_SPLINE_LDS_PIECES 4096     // histograms of up to this many pieces live in LDS (2 x 16 KB)
#define PCX_SPLINE_BLOCK_POINTS 4096   // points per workgroup of the routing kernels

// Routing + histogram.  Atomics on neighbouring global counters all land in one L2 line and retire at
// ~10 ns each (measured: 10^6 points over 64 pieces cost 7 ms that way), so each workgroup counts its
// PCX_SPLINE_BLOCK_POINTS points in LDS and adds one number per non-empty piece to the global histogram.
// lds_hist = 0: more pieces than LDS holds -- wave-grouped global atomics (the counters then spread over
// many lines).
__global__ void __launch_bounds__(256)
k_spline_piece_id(SplineDims sd, const double *__restrict__ knots, const double *__restrict__ pts, long N,
                  int *__restrict__ piece, int *__restrict__ counts, int n_pieces, int lds_hist) {
    __shared__ int hist[PCX_SPLINE_LDS_PIECES];
    const long lo = (long)blockIdx.x * PCX_SPLINE_BLOCK_POINTS;
    const long hi = lo + PCX_SPLINE_BLOCK_POINTS < N ? lo + PCX_SPLINE_BLOCK_POINTS : N;
    if (lds_hist) {
        for (int i = threadIdx.x; i < n_pieces; i += 256) hist[i] = 0;
        __syncthreads();
    }
    for (long base = lo; base < hi; base += 256) {       // every wave runs the same number of rounds
        const long p = base + threadIdx.x;
        const bool active = p < hi;
        int flat = 0;
        if (active) {
            flat = spline_flat_piece(sd, knots, pts + p * sd.d);
            piece[p] = flat;
        }
        if (lds_hist) { if (active) atomicAdd(&hist[flat], 1); }
        else (void)wave_grouped_add(counts, flat, active);
    }
    if (lds_hist) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_pieces; i += 256)
            if (hist[i]) atomicAdd(&counts[i], hist[i]);
    }
}

// perm[offset[piece] + slot] = point index (slot order inside a bucket is arbitrary: each point's result
// does not depend on its neighbours).  Same two-level scheme: count the workgroup's points per piece in
// LDS, reserve that many slots per piece with ONE global atomic, hand the slots out with LDS atomics.
__global__ void __launch_bounds__(256)
k_spline_scatter(const int *__restrict__ piece, long N, int *__restrict__ cursor, int *__restrict__ perm,
                 int n_pieces, int lds_hist) {
    __shared__ int hist[PCX_SPLINE_LDS_PIECES];
    __shared__ int base_slot[PCX_SPLINE_LDS_PIECES];
    const long lo = (long)blockIdx.x * PCX_SPLINE_BLOCK_POINTS;
    const long hi = lo + PCX_SPLINE_BLOCK_POINTS < N ? lo + PCX_SPLINE_BLOCK_POINTS : N;
    if (!lds_hist) {
        for (long base = lo; base < hi; base += 256) {
            const long p = base + threadIdx.x;
            const bool active = p < hi;
            const int key = active ? piece[p] : 0;
            const int slot = wave_grouped_add(cursor, key, active);
            if (active) perm[slot] = (int)p;
        }
        return;
    }
    for (int i = threadIdx.x; i < n_pieces; i += 256) hist[i] = 0;
    __syncthreads();
    for (long p = lo + threadIdx.x; p < hi; p += 256) atomicAdd(&hist[piece[p]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < n_pieces; i += 256) {
        const int c = hist[i];
        base_slot[i] = c ? atomicAdd(&cursor[i], c) : 0;
        hist[i] = 0;
    }
    __syncthreads();
    for (long p = lo + threadIdx.x; p < hi; p += 256) {
        const int k = piece[p];
        perm[base_slot[k] + atomicAdd(&hist[k], 1)] = (int)p;
    }
}

// ---------------------------------------------------------------------------------
// ChebyshevSlider on the device (reference slider.py:247-318): every slide is a low-dimensional
// interpolant over a column group of the points; value = pivot + sum_s (slide_s - pivot).
// ---------------------------------------------------------------------------------
// out[p * ostride + ooff] = ((pivot + (v_0 - pivot)) + (v_1 - pivot)) + ...   -- the reference's order of
// operations (result starts at the pivot value and takes the slides one by one)
__global__ void k_slider_sum(const double *__restrict__ vals, long N, int n_slides, double pivot,
                             double *__restrict__ out, long ostride, long ooff) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    double r = pivot;
    for (int s = 0; s < n_slides; ++s) r += vals[p * n_slides + s] - pivot;
    out[p * ostride + ooff] = r;
}

// out[p * ostride + ooff + j] = src[p * w + j], j < w: a group of spec columns into its place in a wider result
__global__ void k_scatter_columns(const double *__restrict__ src, long N, int w, double *__restrict__ out, long ostride,
                                  long ooff) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * w) return;
    const long p = e / w;
    const int j = (int)(e - p * w);
    out[p * ostride + ooff + j] = src[e];
}

__global__ void k_fill_strided(double *__restrict__ out, long N, long ostride, long ooff, double v) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < N) out[p * ostride + ooff] = v;
}
