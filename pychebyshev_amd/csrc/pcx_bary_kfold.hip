// pcx_bary_kfold.hip -- planning, packing and launching of k_bary_mfma_kfold (bary_kfold_kernels.h): the MFMA form for
// 3-D tensors one of whose dimensions fills row tiles well.  Part of the barycentric handle (pcx_bary.hip).

#include "pcx_bary_internal.h"
#include "bary_kfold_kernels.h"

// Eligible (kfold_try): three dimensions -- any of them may play the rows --, n0 <= 64 (four row tiles in the accumulators),
// n2 <= 64 (sixteen B operands of dimension 2 in registers), at least 8 fragments per i1 (the prefetch ring).  Whether an
// eligible plan is TAKEN is decided by bary_kfold_take below (priced against the grid plan; fixed bars otherwise).
// PCX_BARY_KFOLD=0 switches the form off, PCX_BARY_KFOLD_EFF=<percent> replaces the pricing by one bar (experiments).
// One assignment of the three tensor dimensions to the roles (rows, loop, b2 registers): fills kp, returns the share of real
// products in hundredths of a percent (0: not eligible).
static long kfold_try(const BaryDims &dm, int dr, int da, int db, bool straddle, BaryKfoldPlan &kp) {
    kp.dim[0] = dr; kp.dim[1] = da; kp.dim[2] = db;
    const long st[3] = {(long)dm.n[1] * dm.n[2], (long)dm.n[2], 1L};
    for (int q = 0; q < 3; ++q) kp.stride[q] = st[kp.dim[q]];
    kp.n0 = dm.n[dr]; kp.n1 = dm.n[da]; kp.n2 = dm.n[db];
    if (kp.n0 > 64 || kp.n2 > 64 || kp.n2 < 2) return 0;
    kp.MT = (kp.n0 + 15) / 16;
    kp.KS2 = (kp.n2 + 3) / 4;
    if (kp.MT * kp.KS2 < 8) return 0;
    kp.trows = std::max(std::max(16 * kp.MT, kp.n1 + 1), 4 * kp.KS2);
    // n2 = 26, 30 (and 22): two indices of dimension 1 share a k-step instead of padding each to a multiple of four
    kp.str = (kp.n2 % 4 == 2 && kp.KS2 >= 6 && kp.KS2 <= 8 && straddle) ? 1 : 0;
    if (kp.str && ((kp.n1 + 1) / 2) * (2 * kp.KS2 - 1) >= kp.n1 * kp.KS2) kp.str = 0;      // a short odd n1: nothing saved
    kp.P = kp.str ? 2 * kp.KS2 - 1 : kp.KS2;
    kp.nbody = kp.str ? (kp.n1 + 1) / 2 : kp.n1;
    const long used = (long)kp.n0 * kp.n1 * kp.n2, padded = 16L * kp.MT * kp.nbody * kp.P * 4;
    return used * 10000 / padded;                     // hundredths of a percent
}

// What the matrix pipe makes of a row-tile count (one multiply and one fragment load feed MT x NT matrix instructions):
// measured busy x clock of the unpadded shapes, 15 x 33 x 31 0.65, 32^3 0.75, 48^3 0.84, 64^3 0.91 -- relative weights for
// choosing between assignments of equal padding (20 x 16 x 64: 64 rows in four tiles, not 16 rows in one).
static long kfold_tile_weight(int MT) { return MT >= 4 ? 100 : (MT == 3 ? 93 : (MT == 2 ? 83 : 72)); }

// The best assignment of the roles: fills kp and returns the share of real products in hundredths of a percent (0: the form is
// switched off or no assignment is eligible).  Whether the form is TAKEN is the caller's decision (bary_kfold_take).
PCX_HIDDEN long bary_plan_kfold(const BaryDims &dm, BaryKfoldPlan &kp) {
    const char *env = getenv("PCX_BARY_KFOLD");                                        // read per handle
    const char *strd = getenv("PCX_BARY_KFOLD_STRADDLE");                              // =0: pad n2 = 26, 30 instead (A/B)
    if ((env && env[0] == '0') || dm.d != 3) return 0;
    // the roles go to the assignment with the least padding; the natural order (rows = dimension 0, b2 = the contiguous last
    // dimension) wins ties, so cubes are packed and summed as before
    static const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    long best = 0, best_w = 0;
    BaryKfoldPlan cand;
    for (const auto &pm : perms) {
        const long sc = kfold_try(dm, pm[0], pm[1], pm[2], !(strd && strd[0] == '0'), cand);
        if (sc * kfold_tile_weight(cand.MT) > best_w) { best_w = sc * kfold_tile_weight(cand.MT); best = sc; kp = cand; }
    }
    return best;
}

// Take the k-fold form?  eff: bary_plan_kfold's share of real products (1e-4).  grid_eff > 0: the grid form is the alternative, with
// that share of real products over its padded tiles and grid_ks k-steps per tile -- then both are priced by what the matrix pipe was
// measured to make of them (fraction of the FP64 peak at full tiles: k-fold by row tiles 0.65 / 0.745 / 0.84 / 0.91 -- 15 x 33 x 31,
// 32^3, 48^3, 64^3 --, grid by k-steps per tile 0.49 (5), 0.56, 0.58, 0.64, 0.67, 0.71 (10), 0.745, 0.78 (12), 0.84 (16) -- 20^3,
// 24^3, 28^3, 32^3, 40^3, 48^3, 64^4) and the k-fold form must be 6 % ahead (23^3, 24^3, 21^3 are level: they stay; 48^3, 0.84 / 0.78
// estimated, 0.885 / 0.805 measured, must not fall back); the estimates
// reproduce the measured pairs of profiles/r04_bary_rate_probe_kfold50.txt (26^3 0.605 / 0.465, 29^3 0.61 / 0.48, 40^3 0.70 / 0.71,
// 52^3 0.74 / 0.77, 36^3 0.63 / 0.65) and send 25^3 (0.52 / 0.41) to the k-fold form.  No grid plan (dim-0 groups or a row-code
// plan instead): the fixed bars, 85 % of the padded products real, more than 75 % with one or two row tiles.
// PCX_BARY_KFOLD_EFF=<percent> replaces all of it by one bar (experiments).
static const int kKfoldTileFrac[5] = {0, 650, 745, 840, 910};       // permille of the FP64 peak at full tiles, by row tiles

// the fraction of the FP64 peak (permille) expected of the plan: share of real products x the row-tile figure above
PCX_HIDDEN long bary_kfold_estimate(const BaryKfoldPlan &kp, long eff) {
    return eff <= 0 ? 0 : eff * kKfoldTileFrac[kp.MT < 4 ? kp.MT : 4] / 10000;
}

PCX_HIDDEN bool bary_kfold_take(const BaryKfoldPlan &kp, long eff, long grid_eff, int grid_ks) {
    if (eff <= 0) return false;
    const char *e = getenv("PCX_BARY_KFOLD_EFF");
    const int bar = e ? atoi(e) : 0;
    if (bar > 0 && bar <= 100) return eff >= bar * 100L;
    if (grid_eff > 0) {
        static const int g_ks[18] = {400, 400, 400, 420, 450, 487, 560, 580, 640, 670, 710, 745, 780, 800, 815, 830, 840, 850};
        const long est_k = eff * kKfoldTileFrac[kp.MT < 4 ? kp.MT : 4];
        const long est_g = grid_eff * g_ks[grid_ks < 17 ? grid_ks : 17];
        return est_k * 100 > est_g * 106;
    }
    return kp.MT <= 2 ? eff > 7500 : eff >= 8500;
}

PCX_HIDDEN size_t bary_kfold_frag_count(const BaryKfoldPlan &kp) {
    return ((size_t)kp.nbody * kp.P * kp.MT + PCX_KFOLD_PAD) * 64;
}

PCX_HIDDEN size_t bary_kfold_lds_bytes(const BaryKfoldPlan &kp, int nt) {       // per workgroup of four waves
    return (size_t)4 * kp.trows * 16 * nt * sizeof(double);
}

PCX_HIDDEN int bary_pack_kfold(pcx_bary *h, const double *plain, double *frag) {
    const size_t cnt = bary_kfold_frag_count(h->kf);
    hipLaunchKernelGGL(k_pack_fragments_kfold, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, plain, frag, h->kf);
    HIP_TRY(hipGetLastError());
    return PCX_OK;
}

template <int MT, int KS2, int NT, bool STR>
static int launch_kfold_t(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N, double *d_out,
                          long ostride, long ooff, hipStream_t st, const int *perm) {
    if constexpr (MT * KS2 < 8 || (STR && (KS2 < 6 || KS2 > 8))) {
        return fail(PCX_ERR_UNSUPPORTED, "no k-fold MFMA instantiation for MT=%d KS2=%d", MT, KS2);
    } else {
        // small batches: one row tile per wave (split-M launch) while that still leaves CUs idle -- a wave's chain of matrix
        // instructions is what a single query waits for
        if constexpr (NT == 1 && MT >= 2) {
            const long tiles = (N + 15) / 16;
            if (tiles * m <= 1024) {
                const size_t lds = (size_t)MT * h->kf.trows * 16 * sizeof(double) + 64 * sizeof(double);
                auto kern = k_bary_mfma_kfold<MT, KS2, 1, STR, true>;
                if (lds > 64 * 1024)
                    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3((unsigned)tiles, 1, (unsigned)m), dim3(64 * MT), lds, st, h->dims, h->kf, h->d_nodes,
                                   h->d_wts, h->grid_prod ? h->d_gsnodes : nullptr, frag_tab, d_pts, d_out, N, ostride, ooff, perm);
                HIP_TRY(hipGetLastError());
                return PCX_OK;
            }
        }
        const size_t lds = bary_kfold_lds_bytes(h->kf, NT);
        auto kern = k_bary_mfma_kfold<MT, KS2, NT, STR>;
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const long per_wg = 64L * NT;
        const long blocks = (N + per_wg - 1) / per_wg;
        if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks, 1, (unsigned)m), dim3(256), lds, st, h->dims, h->kf, h->d_nodes, h->d_wts,
                           h->grid_prod ? h->d_gsnodes : nullptr, frag_tab, d_pts, d_out, N, ostride, ooff, perm);
        HIP_TRY(hipGetLastError());
        return PCX_OK;
    }
}

template <int MT, int NT>
static int launch_kfold_mt(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N, double *d_out,
                           long ostride, long ooff, hipStream_t st, const int *perm) {
    switch (h->kf.KS2) {
#define CASE_KS(v) case v: return h->kf.str ? launch_kfold_t<MT, v, NT, true>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, perm) \
                                        : launch_kfold_t<MT, v, NT, false>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
        CASE_KS(2) CASE_KS(3) CASE_KS(4) CASE_KS(5) CASE_KS(6) CASE_KS(7) CASE_KS(8) CASE_KS(9)
        CASE_KS(10) CASE_KS(11) CASE_KS(12) CASE_KS(13) CASE_KS(14) CASE_KS(15) CASE_KS(16)
#undef CASE_KS
    }
    return fail(PCX_ERR_UNSUPPORTED, "no k-fold MFMA instantiation for KS2=%d", h->kf.KS2);
}

PCX_HIDDEN int bary_launch_kfold(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N, double *d_out,
                                 long ostride, long ooff, hipStream_t st, const int *perm) {
    // two column tiles per wave for throughput; one when the batch cannot fill the chip (same sums either way)
    const bool two = N >= 65536 && h->nt == 2;
#define GO(MTv) (two ? launch_kfold_mt<MTv, 2>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, perm) \
                     : launch_kfold_mt<MTv, 1>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, perm))
    switch (h->kf.MT) {
        case 1: return GO(1);
        case 2: return GO(2);
        case 3: return GO(3);
        case 4: return GO(4);
    }
#undef GO
    return fail(PCX_ERR_UNSUPPORTED, "no k-fold MFMA instantiation for MT=%d", h->kf.MT);
}
