// pcx_common.h -- shared host/device definitions for libpcx_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcx.h"

#define PCX_WAVE 64

typedef double pcx_d4 __attribute__((ext_vector_type(4)));

// Row / k codes are 4 x 8-bit row indices into the per-wave weight table held in LDS.  The
// table has two parts, each closed by a row of ones that pads unused fields:
//   [head dims 0..split) rows][ones][tail dims split..d) rows][ones]      (sum_n + 2 rows)
// row codes index the head part from 0, k codes the tail part from plan.tail_base, so each
// part may hold up to PCX_MAX_PART_ROWS weight rows (e.g. 64^4 = 192 + 64).
#define PCX_MAX_PART_ROWS 255
#define PCX_CODE_FIELDS 4

// ---- barycentric kernel parameters (passed by value -> kernarg / SGPRs) ----------
struct BaryDims {
    int d;                  // number of dimensions
    int sum_n;              // sum of n[]
    int n[PCX_MAX_DIMS];    // nodes per dimension
    int off[PCX_MAX_DIMS];  // prefix offsets into nodes_cat / weights_cat
};

// y[p] = sum_m wM[m,p] * sum_k T2[m,k] * wK[k,p] with T2 = tensor viewed as (M x K):
// dims [0, split) index the rows m (head), dims [split, d) the columns k (tail).
struct BaryMfmaPlan {
    int split;
    int M;    // prod n[0:split]  (1 when split == 0)
    int K;    // prod n[split:d]
    int MT;   // row tiles of 16 covering M
    int KS;   // k-steps of 4 run by the kernel instantiation (KS*4 >= K)
    int tail_base;  // first table row of the tail part (= sum of head n + 1)
    int rows;       // table rows (= sum_n + 2)
};

// k_bary_mfma_grid (bary_grid_kernels.h): row tiles = RA x RB blocks of the last two head dimensions A, B
struct BaryGridPlan {
    int RA;          // A rows per tile: 1, 2 or 4 (RB = 16 / RA rows of B)
    int gbs;         // log2(GB), GB = 4 / RA lane groups span B
    int nA, nB;      // nodes of the two tiled head dimensions (dims split-2 and split-1)
    int TA, TB;      // tiles along A and B
    int rowA, rowB;  // first table row of A and of B
    int nouter;      // head dimensions in front of A: 0, 1 or 2
    int no1;         // nodes of the second outer dimension (1 when there is none)
    int rowo0, rowo1;// first table rows of the outer dimensions
    int nchunks;     // O * TA
    int hrows;       // head rows + the slack the padded B index of a last tile reads (zeroed)
    int trows;       // table rows per wave: max(hrows, tail rows + 1)
    int af;          // 1: dimension A's weight is formed per chunk, the table has no rows for A (B sits at rowA)
    int wpb;         // waves per workgroup: 1 (fragment image in L2) or 4 (large tensors: the waves share the stream)
    int MT;          // nchunks * TB row tiles
};

// k_bary_mfma_kfold (bary_kfold_kernels.h): 3-D tensors, rows = dimension 0 (MT tiles held in the accumulators),
// K = dimensions 1 x 2 with the B operand formed per k-step
#define PCX_KFOLD_PAD 16   // fragments behind the image the prefetch ring may read (zeroed, never multiplied)
struct BaryKfoldPlan {
    int dim[3];      // the tensor dimension in each role: dim[0] -> rows, dim[1] -> the loop ("i1"), dim[2] -> the b2 registers
    long stride[3];  // element strides of those dimensions in the C-order tensor
    int n0, n1, n2;  // their node counts (the names below are ROLES: "dimension 0" = dim[0], ...)
    int MT;          // ceil(n0 / 16), at most 4
    int KS2;         // ceil(n2 / 4), at most 16
    int trows;       // table rows per wave: max(16 MT, n1 + 1, 4 KS2)
    int str;         // 1: n2 = 4 KS2 - 2, K runs over pairs of i1 without padding (P = 2 KS2 - 1 k-steps per pair)
    int P;           // k-steps per loop body: KS2, or 2 KS2 - 1
    int nbody;       // loop bodies: n1, or ceil(n1 / 2)
};

// A "dim-0 group" of a multi-spec launch: specs that differ only in their derivative order along dimension 0
// share ONE contraction of dimensions 1 .. d-1 (reference vectorized_eval_multi, barycentric.py:1098-1110:
// contract the later dimensions, THEN apply D_0, then contract dimension 0).  The GEMM rows are laid out in
// slabs of one i0 each (padded to whole row tiles), the kernel keeps the n0 per-i0 partial sums P of a point
// and finishes with g = D_0^o P and y = b_0 . g on the VALU.
#define PCX_G0_MAX 9
struct BaryG0 {
    int nmem;                // members of the group (0: ordinary launch)
    int maxorder;            // highest dim-0 order among them
    int tps;                 // row tiles per dim-0 slab
    int n0;                  // nodes of dimension 0 (= slabs)
    int order[PCX_G0_MAX];   // dim-0 derivative order of member s
    int col[PCX_G0_MAX];     // output column of member s
};

// "Plain" (C-order) tensors carry PCX_PLAIN_PAD zeroed doubles behind their end: the lane-per-point kernels read a
// row with a fixed-width run of scalar loads that may reach past the last row.
#define PCX_PLAIN_PAD 64

// lane-per-point kernels (k_bary_small, k_bary_sq): power-of-two coordinate scales of the weight products
struct BarySmallScale {
    double s[4];            // 2^e per dimension (d <= 4)
};

// one piece of a piecewise interpolant as the all-pieces-in-one-launch kernels read it (device table, scalar loads)
struct SplinePieceModel {
    const double *snodes, *nodes, *wts;
    const double *T;                    // the piece's tensor (m == 1) ...
    const double *const *T_tab;         // ... or its device table of m tensors (NULL when m == 1)
    BarySmallScale sc;
};

// ---- tensor-train kernel parameters ---------------------------------------------
struct TTDims {
    int d;
    int n[PCX_MAX_DIMS];
    int col[PCX_MAX_DIMS];        // user column read by storage position k (dim_order)
    double lo[PCX_MAX_DIMS];
    double hi[PCX_MAX_DIMS];
    double scale[PCX_MAX_DIMS];   // 2 / (hi - lo): s = fma(x - lo, scale, -1) without a division per point
    long frag_off[PCX_MAX_DIMS];  // offset (doubles) of storage dim k in the packed cores
};
