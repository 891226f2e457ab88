// pcx_bary_grid.hip -- planning, packing and launching of k_bary_mfma_grid (bary_grid_kernels.h): the MFMA form for
// short plans, row tiles laid over the last two head dimensions.  Part of the barycentric handle (pcx_bary.hip).

#include "pcx_bary_internal.h"
#include "bary_grid_kernels.h"

// Is the grid form ahead of the row-code form for this plan?  Cost per point block in matrix-instruction units: a row
// tile is KS instructions plus its epilogue -- about 5 for the row-code look-ups (plan_mfma, measured on 15^4), about 1
// here (four FMAs on four table rows at fixed strides).  The grid pads dimension A to RA and B to 16 / RA rows per tile:
// RA is chosen for the fewest tiles.  A margin of 8 % keeps the long plans (11^5: 31 k-steps) where they are.
PCX_HIDDEN bool bary_plan_grid(const BaryDims &dm, const BaryMfmaPlan &plan, BaryGridPlan &gp) {
    const char *env = getenv("PCX_BARY_GRID");                      // read per handle: a process may build both forms (tests)
    if (env && env[0] == '0') return false;
    const int split = plan.split;
    if (split < 2 || split > 4 || dm.d - split > PCX_CODE_FIELDS || plan.KS > 32) return false;
    const int nA = dm.n[split - 2], nB = dm.n[split - 1];
    long O = 1;
    for (int k = 0; k < split - 2; ++k) O *= dm.n[k];
    long best = -1;
    for (int RA : {4, 2, 1}) {
        const int RB = 16 / RA;
        const long tiles = O * ((nA + RA - 1) / RA) * ((nB + RB - 1) / RB);
        if (best < 0 || tiles < best) {
            best = tiles;
            gp.RA = RA;
            gp.gbs = RA == 4 ? 0 : (RA == 2 ? 1 : 2);
        }
    }
    if (best > (1L << 26)) return false;
    gp.nA = nA; gp.nB = nB;
    gp.TA = (nA + gp.RA - 1) / gp.RA;
    gp.TB = (nB + 16 / gp.RA - 1) / (16 / gp.RA);
    gp.af = gp.TB * plan.KS >= 48 ? 1 : 0;                          // long chunks: A's weight formed per chunk, no table rows for A
    gp.rowA = dm.off[split - 2];
    gp.rowB = gp.af ? dm.off[split - 2] : dm.off[split - 1];
    gp.nouter = split - 2;
    gp.no1 = split == 4 ? dm.n[1] : 1;
    gp.rowo0 = dm.off[0]; gp.rowo1 = split == 4 ? dm.off[1] : 0;
    gp.nchunks = (int)(O * gp.TA);
    gp.MT = (int)best;
    const int tail_rows = dm.sum_n - dm.off[split];
    gp.hrows = gp.rowB + gp.TB * (16 / gp.RA);                       // outer rows, (A rows,) B rows, zeroed slack of B's last tile
    gp.trows = std::max(gp.hrows, tail_rows + 1) + 2;                // + 1 / S and the exact-node index of dimension A per point (af)
    // Four waves per workgroup wherever two such workgroups fit a CU (or the image is beyond L2 anyway): waves started
    // together walk the fragment image in step, so one wave's L2 fetch is an L1 hit for the other three.  With one wave per
    // workgroup every fragment load of every wave misses L1 (TCP_PENDING_STALL_CYCLES: half the kernel's cycles on 30^3;
    // measured 30^3 0.50 -> 0.54, 20^3 / 48^3 +3 %, the rest +1 %).
    const size_t table = (size_t)gp.trows * 32 * sizeof(double);
    gp.wpb = ((size_t)gp.MT * plan.KS * 512 > ((size_t)1 << 20) || 4 * table <= 80 * 1024) ? 4 : 1;

    // Measured (profiles/r04_bary_rate_probe.txt, fraction of the FP64 peak, row codes -> grid): 30^3 0.41 -> 0.55, 40^3
    // 0.54 -> 0.71, 48^3 0.78, 32^3 0.46 -> 0.64, 28^3 0.42 -> 0.58, 20^3 0.44 -> 0.49, 64^4 0.58 -> 0.84, 65^3 0.61 -> 0.74; 21^3
    // 0.39 -> 0.41 (18 % more row tiles), 7^5 0.66 -> 0.56 (27 % more): from 13 k-steps on the row-code kernel's
    // hand-pipelined loop stays ahead unless the grid pads little AND forms A's weight per chunk.
    const double cost_grid = (double)gp.MT * (plan.KS + 1.0), cost_codes = (double)plan.MT * (plan.KS + 5.0);
    if (env && env[0] == '2') return true;                          // experiments: every eligible plan
    if (plan.KS > 12) return gp.MT == plan.MT || (gp.af && gp.MT * 100L <= plan.MT * 112L);      // 65^3: 289 tiles for 265, 0.61 -> 0.74
    return cost_grid <= 0.85 * cost_codes;
}

PCX_HIDDEN size_t bary_grid_lds_bytes(const pcx_bary *h, int nt) {       // per workgroup
    return (size_t)h->gp.wpb * h->gp.trows * 16 * nt * sizeof(double);
}

PCX_HIDDEN int bary_pack_grid(pcx_bary *h, const double *plain, double *frag) {
    const long cnt = (long)h->gp.MT * h->plan.KS * 64;
    hipLaunchKernelGGL(k_pack_fragments_grid, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, plain, frag, h->gp,
                       h->plan.K, h->plan.KS);
    HIP_TRY(hipGetLastError());
    return PCX_OK;
}

template <int KS, int NT, int WPB, bool AF>
static int launch_grid_t(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N, double *d_out,
                         long ostride, long ooff, hipStream_t st, Scratch *split_scratch, const int *perm) {
    const size_t lds = bary_grid_lds_bytes(h, NT);
    auto kern = k_bary_mfma_grid<KS, NT, WPB, AF>;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long per_wg = 16L * NT * WPB;
    const long blocks = (N + per_wg - 1) / per_wg;
    if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
    const int nchunks = h->gp.nchunks;
    int nsplit = 1, cps = nchunks;
    const long want = 2048 / WPB;   // waves that fill 256 CUs at two per SIMD
    if (split_scratch && blocks * m < want && nchunks > 1) {
        nsplit = (int)std::min<long>(nchunks, (want + blocks * m - 1) / (blocks * m));
        cps = (nchunks + nsplit - 1) / nsplit;
        nsplit = (nchunks + cps - 1) / cps;
    }
    double *partial = nullptr;
    if (nsplit > 1) {
        int rc = split_scratch->reserve((size_t)m * nchunks * 4 * (size_t)N * sizeof(double));
        if (rc) return rc;
        partial = (double *)split_scratch->ptr;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)nsplit, (unsigned)m), dim3(64 * WPB), lds, st, h->dims, h->plan, h->gp,
                       h->d_nodes, h->d_wts, h->grid_prod ? h->d_gsnodes : nullptr, frag_tab, h->d_kcode, d_pts, d_out, N, ostride, ooff,
                       cps, partial, perm);
    HIP_TRY(hipGetLastError());
    if (nsplit > 1) {
        const long cnt = N * m;
        hipLaunchKernelGGL(k_bary_grid_reduce, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, partial, d_out, N, nchunks, m,
                           ostride, ooff, perm);
        HIP_TRY(hipGetLastError());
    }
    return PCX_OK;
}

template <int NT>
static int launch_grid_nt(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N, double *d_out,
                          long ostride, long ooff, hipStream_t st, Scratch *split_scratch, const int *perm) {
    switch (h->plan.KS) {
#define GRID_GO(v, W, A) launch_grid_t<v, NT, W, A>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm)
#define CASE_KS(v) case v: return h->gp.wpb == 4 ? (h->gp.af ? GRID_GO(v, 4, true) : GRID_GO(v, 4, false)) \
                                        : (h->gp.af ? GRID_GO(v, 1, true) : GRID_GO(v, 1, false));
        CASE_KS(1) CASE_KS(2) CASE_KS(3) CASE_KS(4) CASE_KS(5) CASE_KS(6) CASE_KS(7) CASE_KS(8)
        CASE_KS(9) CASE_KS(10) CASE_KS(11) CASE_KS(12) CASE_KS(13) CASE_KS(14) CASE_KS(15) CASE_KS(16)
        CASE_KS(17) CASE_KS(18) CASE_KS(19) CASE_KS(20) CASE_KS(21) CASE_KS(22) CASE_KS(23) CASE_KS(24)
        CASE_KS(25) CASE_KS(26) CASE_KS(27) CASE_KS(28) CASE_KS(29) CASE_KS(30) CASE_KS(31) CASE_KS(32)
#undef CASE_KS
#undef GRID_GO
    }
    return fail(PCX_ERR_UNSUPPORTED, "no grid MFMA instantiation for KS=%d", h->plan.KS);
}

PCX_HIDDEN int bary_launch_grid(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N, double *d_out,
                                long ostride, long ooff, hipStream_t st, Scratch *split_scratch, const int *perm) {
    // two column tiles per wave for throughput; one when the batch cannot fill the chip.  (Four, for plans of up to 8 k-steps,
    // were measured in round 4: 20^3 0.487 -> 0.323, 23^3 0.500 -> 0.432, 24^3 0.559 -> 0.484, 25^3 0.425 -> 0.468 -- not kept.)
    const int nt = (N >= 65536) ? h->nt : 1;
    return nt == 2 ? launch_grid_nt<2>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm)
                   : launch_grid_nt<1>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
}
