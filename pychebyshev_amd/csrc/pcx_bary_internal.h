// pcx_bary_internal.h -- the barycentric handle as the other translation units see it (pcx_spline.hip builds
// piecewise interpolants and sliders out of pcx_bary handles).  Not part of the ABI.
#pragma once

#include "pcx_internal.h"

// "Plain" (C-order) tensors carry PCX_PLAIN_PAD zeroed doubles behind their end: k_bary_small reads a
// row with a fixed-width run of scalar loads that may reach past the last row.
static int alloc_plain(DevBuf &b, long total) {
    int rc = b.alloc(((size_t)total + PCX_PLAIN_PAD) * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipMemset((char *)b.p + (size_t)total * sizeof(double), 0, PCX_PLAIN_PAD * sizeof(double)));
    return PCX_OK;
}

struct DerivedTensor {
    double *plain = nullptr;  // C-order tensor after the derivative passes (prod n doubles)
    double *frag = nullptr;   // MFMA A-fragment packing of `plain` (MT*KS*64 doubles) or NULL
    double **slot = nullptr;  // device table with the single entry `frag` (kernel's frag_tab)
    double *frag_g0 = nullptr;   // slab packing for dim-0 group launches (n0 * tps * KS * 64 doubles), built on first use
    double **slot_g0 = nullptr;  // device table with the single entry `frag_g0`
    uint64_t last_use = 0;    // handle clock at the last request (least-recently-used eviction)
    void free_all() {
        if (plain) (void)hipFree(plain);
        if (frag) (void)hipFree(frag);
        if (slot) (void)hipFree(slot);
        if (frag_g0) (void)hipFree(frag_g0);
        if (slot_g0) (void)hipFree(slot_g0);
        plain = frag = frag_g0 = nullptr;
        slot = slot_g0 = nullptr;
    }
};

struct pcx_bary {
    int device = 0;
    hipStream_t stream = nullptr;
    BaryDims dims;
    long total = 0;
    std::vector<int> doff;           // offsets of D_k in diff_cat
    double *d_nodes = nullptr, *d_wts = nullptr, *d_diff = nullptr;
    // launch plan
    bool mfma_ok = false;
    BaryMfmaPlan plan;
    int nt = 2;                      // point tiles per wave in the MFMA kernel
    unsigned *d_rowcode = nullptr, *d_kcode = nullptr;
    unsigned *d_rowcode_hi = nullptr, *d_kcode_hi = nullptr;   // fields 4..7 (wide plans only)
    bool wide = false;      // more than four head or tail dimensions
    // short plans: row tiles = RA x RB blocks of the last two head dimensions, no row codes (bary_grid_kernels.h);
    // the fragment image of every derivative tensor is then packed in that order
    bool grid_ok = false;
    BaryGridPlan gp;
    double *d_gsnodes = nullptr;     // nodes times a power of two per dimension, then the PCX_MAX_DIMS scales themselves
    bool kfold_ok = false;           // 3-D, whole row tiles along dimension 0: k_bary_mfma_kfold (bary_kfold_kernels.h)
    BaryKfoldPlan kf;
    bool grid_prod = false;          // every dimension <= 64 nodes: division-free weights (grid_weights_prod)
    // dim-0 groups (BaryG0): specs differing only in their dim-0 order share one slab-packed GEMM
    bool g0_ok = false;
    int g0_tps = 0;                  // row tiles per dim-0 slab
    int g0_nf = 2;                   // live row-code fields (head dimensions 1 .. split-1, at least two)
    unsigned *d_rowcode_g0 = nullptr;
    int g0_span = 1;                 // dim-0 orders above its base tensor's a slab GEMM serves (pcx_bary_set_group_span)
    // dim-q groups (q > 0): the same model with dimension q moved to the front, built on first use (bary_rot); a pair of
    // specs one order apart along q shares ITS dim-0 slab GEMM, on the batch with its columns in that order
    pcx_bary *rot[PCX_MAX_DIMS] = {};
    char rot_state[PCX_MAX_DIMS] = {};   // 0 untried, 1 ready, 2 not available
    Scratch s_rot, s_rot2;           // the batch in a sub-model's column order (per staging slot)
    // what the probe measured for "spec base + e_q out of base's GEMM": |shared - own GEMM| / scale (bary_pair_deviation);
    // a pair is formed when that is at most group_tol
    std::map<std::vector<int>, double> pair_dev;
    double group_tol = 3e-13;
    int lpp = 64;                    // lanes per point in the rows kernel
    bool mfma4_ok = false;           // 4x4x4_4b form available (LDS budget)
    int small_nlp = 0;               // lane-per-point kernel for small tensors: padded last-dim width, 0 = not available
    std::vector<double> dom_lo, dom_hi;   // the domain, when the handle came from a .pcb file (pcx_bary_save_pcb)
    BarySmallScale small_scale;      // its power-of-two coordinate scales and the nodes times them
    double *d_snodes = nullptr;
    bool small_preferred = false;    // auto picks it (few row tiles: the MFMA kernel would be all prologue)
    int sq_nl = 0;                   // k_bary_sq (last two dimensions of sq_nl nodes each, d <= 4): 0 = not available
    bool sq_preferred = false;       // auto picks it
    int variant = 0;                 // 0 auto, 1 rows, 2 mfma 16x16x4, 3 mfma 4x4x4_4b, 4 lane-per-point (small tensors)
    std::mutex mu;
    std::map<std::vector<int>, DerivedTensor> cache;
    uint64_t clock = 0;              // bumped per request; entries used since `call_mark` are never evicted
    uint64_t call_mark = 0;
    Scratch s_pts, s_out;
    hipStream_t stream2 = nullptr;   // second staging slot of the host-pointer pipeline (lazy)
    Scratch s_pts2, s_out2;
    double **d_tab = nullptr;        // frag table for multi-spec launches (kMaxSpecs entries)
    std::vector<double *> tab_host;  // what d_tab currently holds
    Scratch s_partial;               // per-chunk totals of split launches
    Pinned pin;                      // zero-copy staging for small host-pointer batches
};

static const int kMaxSpecs = 64;      // derivative specs evaluated by one launch (grid.z)

// pcx_bary_grid.hip
PCX_HIDDEN bool bary_plan_grid(const BaryDims &dm, const BaryMfmaPlan &plan, BaryGridPlan &gp);
PCX_HIDDEN int bary_pack_grid(pcx_bary *h, const double *plain, double *frag);
PCX_HIDDEN size_t bary_grid_lds_bytes(const pcx_bary *h, int nt);
PCX_HIDDEN long bary_plan_kfold(const BaryDims &dm, BaryKfoldPlan &kp);
PCX_HIDDEN long bary_kfold_estimate(const BaryKfoldPlan &kp, long eff);
PCX_HIDDEN bool bary_kfold_take(const BaryKfoldPlan &kp, long eff, long grid_eff, int grid_ks);
PCX_HIDDEN size_t bary_kfold_frag_count(const BaryKfoldPlan &kp);
PCX_HIDDEN size_t bary_kfold_lds_bytes(const BaryKfoldPlan &kp, int nt);
PCX_HIDDEN int bary_pack_kfold(pcx_bary *h, const double *plain, double *frag);
PCX_HIDDEN int bary_launch_kfold(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N, double *d_out,
                                 long ostride, long ooff, hipStream_t st, const int *perm);
PCX_HIDDEN int bary_launch_grid(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N, double *d_out,
                                long ostride, long ooff, hipStream_t st, Scratch *split_scratch, const int *perm);

// pcx_bary.hip
PCX_HIDDEN int bary_get_tensor(pcx_bary *h, const int32_t *deriv, DerivedTensor **out);
PCX_HIDDEN int bary_effective_variant(const pcx_bary *h);
PCX_HIDDEN int bary_launch(pcx_bary *h, DerivedTensor *const *dts, int m, const double *const *frag_tab,
                           const double *d_pts, long N, double *d_out, long ostride, long ooff, hipStream_t st,
                           Scratch *split_scratch, const int *perm = nullptr);
PCX_HIDDEN int bary_launch_small_pieces(int dout, const pcx_bary *p0, const SplinePieceModel *models, const int *blk_piece,
                                        const int *blk_first, const int *piece_end, int m, long blocks, const double *dp,
                                        double *dout_buf, const int *perm, hipStream_t st);
PCX_HIDDEN int bary_launch_sq_pieces(const pcx_bary *p0, const SplinePieceModel *models, const int *blk_piece, const int *blk_first,
                                     const int *piece_end, int m, long blocks, const double *dp, double *dout, const int *perm,
                                     hipStream_t st);
