// pcx_spline.hip -- C ABI of libpcx_hip.so (see include/pcx.h): piecewise interpolants and sliders, both built
// out of barycentric handles (pcx_bary.hip).  gfx950 only.

#include "pcx_bary_internal.h"
#include "route_kernels.h"

// ---------------------------------------------------------------------------------
// spline (piecewise) handle
// ---------------------------------------------------------------------------------
struct pcx_spline {
    int device = 0;
    hipStream_t stream = nullptr;
    SplineDims sd;
    int n_pieces = 0;
    std::vector<pcx_bary *> pieces;      // borrowed
    double *d_knots = nullptr;
    int *d_counts = nullptr;             // n_pieces: histogram, then bucket cursors
    int lds_hist = 1;                    // routing kernels count per workgroup in LDS (<= PCX_SPLINE_LDS_PIECES pieces)
    // the per-piece launches of one batch are independent: they go round-robin over a few side streams so
    // that small buckets overlap instead of queueing behind each other's launch latency
    static const int kSide = 4;
    hipStream_t side[kSide] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kSide] = {nullptr, nullptr, nullptr, nullptr};
    std::mutex mu;
    Scratch s_pts, s_out, s_piece, s_perm, s_partial;
    // one launch for all pieces (pieces of equal shape on the lane-per-point kernel): per-piece model table,
    // per-workgroup (piece, first slot) lists; staged through a pinned host buffer
    bool fused_ok = false;
    Scratch s_models, s_blk;
    void *pin_stage = nullptr;
    size_t pin_cap = 0;
};

extern "C" int pcx_spline_destroy(pcx_spline *h) {
    PCX_API_BEGIN
    if (!h) return PCX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_knots);
    (void)hipFree(h->d_counts);
    h->s_pts.release(); h->s_out.release(); h->s_piece.release(); h->s_perm.release(); h->s_partial.release();
    h->s_models.release(); h->s_blk.release();
    if (h->pin_stage) (void)hipHostFree(h->pin_stage);
    for (int i = 0; i < pcx_spline::kSide; ++i) {
        if (h->side[i]) { (void)hipStreamSynchronize(h->side[i]); (void)hipStreamDestroy(h->side[i]); }
        if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_spline_create(int device, int d, const int32_t *n_knots, const double *knots_cat,
                                 pcx_bary *const *pieces, int n_pieces, pcx_spline **out) {
    PCX_API_BEGIN
    if (!out) return fail(PCX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (d < 1 || d > PCX_MAX_DIMS || !n_knots || !pieces) return fail(PCX_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    pcx_spline *h = new (std::nothrow) pcx_spline();
    if (!h) return fail(PCX_ERR_NOMEM, "out of host memory");
    h->device = device;
    h->sd.d = d;
    long total = 1;
    int nk_total = 0;
    for (int k = 0; k < PCX_MAX_DIMS; ++k) { h->sd.nknots[k] = 0; h->sd.koff[k] = 0; h->sd.shape[k] = 1; }
    for (int k = 0; k < d; ++k) {
        if (n_knots[k] < 0 || n_knots[k] > 4096) { delete h; return fail(PCX_ERR_INVALID, "n_knots[%d]=%d", k, n_knots[k]); }
        h->sd.nknots[k] = n_knots[k];
        h->sd.koff[k] = nk_total;
        h->sd.shape[k] = n_knots[k] + 1;
        for (int j = 1; j < n_knots[k]; ++j)
            if (!(knots_cat[nk_total + j - 1] <= knots_cat[nk_total + j])) { delete h; return fail(PCX_ERR_INVALID, "knots of dimension %d are not sorted", k); }
        nk_total += n_knots[k];
        total *= n_knots[k] + 1;
        if (total > (1 << 20)) { delete h; return fail(PCX_ERR_UNSUPPORTED, "more than 2^20 pieces"); }
    }
    if (n_pieces != total) { delete h; return fail(PCX_ERR_INVALID, "n_pieces=%d but the knots define %ld pieces", n_pieces, total); }
    if (nk_total > 0 && !knots_cat) { delete h; return fail(PCX_ERR_INVALID, "knots_cat is NULL"); }
    for (int i = 0; i < n_pieces; ++i) {
        if (!pieces[i] || pieces[i]->device != device || pieces[i]->dims.d != d) { delete h; return fail(PCX_ERR_INVALID, "piece %d is NULL, on another device or of another dimension", i); }
        h->pieces.push_back(pieces[i]);
    }
    h->n_pieces = n_pieces;
    {   // the one-launch path: every piece the same shape, all on the lane-per-point kernel (PCX_SPLINE_FUSED=0: off)
        const pcx_bary *p0 = h->pieces[0];
        const char *f = getenv("PCX_SPLINE_FUSED");
        bool ok = n_pieces > 1 && (p0->small_nlp > 0 || p0->sq_nl > 0) && !(f && f[0] == '0');
        for (int i = 0; ok && i < n_pieces; ++i) {
            const pcx_bary *pc = h->pieces[i];
            ok = pc->small_nlp == p0->small_nlp && pc->sq_nl == p0->sq_nl && memcmp(&pc->dims, &p0->dims, sizeof(BaryDims)) == 0;
        }
        h->fused_ok = ok;
    }
    {   // PCX_SPLINE_GLOBAL_HIST=1 forces the many-pieces routing path (tests)
        const char *g = getenv("PCX_SPLINE_GLOBAL_HIST");
        h->lds_hist = (n_pieces <= PCX_SPLINE_LDS_PIECES && !(g && g[0] == '1')) ? 1 : 0;
    }
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (n_pieces > 2) {
        for (int i = 0; i < pcx_spline::kSide && e == hipSuccess; ++i) {
            e = hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming);
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_knots, (nk_total ? nk_total : 1) * sizeof(double));
    if (e == hipSuccess && nk_total) e = hipMemcpy(h->d_knots, knots_cat, nk_total * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_counts, (size_t)n_pieces * sizeof(int));
    if (e != hipSuccess) { int c = fail(PCX_ERR_HIP, "spline create: %s", hipGetErrorString(e)); pcx_spline_destroy(h); return c; }
    *out = h;
    return PCX_OK;
    PCX_API_END
}

// Route + bucket cnt device-resident points; returns the per-piece counts/offsets on the host
// and leaves the bucket permutation in h->s_perm.  Caller holds h->mu.
static int spline_bucket(pcx_spline *h, const double *dp, long cnt, std::vector<int> &counts,
                         std::vector<int> &offsets) {
    int rc = h->s_piece.reserve((size_t)cnt * sizeof(int));
    if (rc) return rc;
    rc = h->s_perm.reserve((size_t)cnt * sizeof(int));
    if (rc) return rc;
    int *piece = (int *)h->s_piece.ptr, *perm = (int *)h->s_perm.ptr;
    HIP_TRY(hipMemsetAsync(h->d_counts, 0, (size_t)h->n_pieces * sizeof(int), h->stream));
    const unsigned blocks = (unsigned)((cnt + PCX_SPLINE_BLOCK_POINTS - 1) / PCX_SPLINE_BLOCK_POINTS);
    const int lds_hist = h->lds_hist;
    hipLaunchKernelGGL(k_spline_piece_id, dim3(blocks), dim3(256), 0, h->stream, h->sd, h->d_knots, dp, cnt, piece, h->d_counts,
                       h->n_pieces, lds_hist);
    HIP_TRY(hipGetLastError());
    counts.assign(h->n_pieces, 0);
    HIP_TRY(hipMemcpyAsync(counts.data(), h->d_counts, (size_t)h->n_pieces * sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    offsets.assign(h->n_pieces, 0);
    int acc = 0;
    for (int i = 0; i < h->n_pieces; ++i) { offsets[i] = acc; acc += counts[i]; }
    HIP_TRY(hipMemcpyAsync(h->d_counts, offsets.data(), (size_t)h->n_pieces * sizeof(int), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_spline_scatter, dim3(blocks), dim3(256), 0, h->stream, piece, cnt, h->d_counts, perm, h->n_pieces, lds_hist);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));   // `offsets` (pageable) must stay valid until copied
    return PCX_OK;
}


static int spline_launch_fused(pcx_spline *h, const double *dp, const std::vector<int> &counts,
                               const std::vector<int> &offsets, const int32_t *derivs, int m, double *dout, bool *done) {
    *done = false;
    if (!h->fused_ok || m > kMaxSpecs) return PCX_OK;
    const int d = h->sd.d;
    const int np = h->n_pieces;
    // every piece on the same lane-per-point form: 4 (k_bary_small) or 5 (k_bary_sq: equal trailing node counts)
    const int form = bary_effective_variant(h->pieces[0]);
    if (form != 4 && form != 5) return PCX_OK;
    for (int i = 0; i < np; ++i)
        if (counts[i] && bary_effective_variant(h->pieces[i]) != form) return PCX_OK;
    long blocks = 0;
    for (int i = 0; i < np; ++i) blocks += (counts[i] + 63) / 64;
    if (blocks == 0) { *done = true; return PCX_OK; }
    // host staging: [models np][piece_end np][blk_piece blocks][blk_first blocks]
    const size_t b_models = (size_t)np * sizeof(SplinePieceModel);
    const size_t b_ints = ((size_t)np + 2 * (size_t)blocks) * sizeof(int);
    const size_t need = b_models + b_ints;
    if (need > h->pin_cap) {
        if (h->pin_stage) (void)hipHostFree(h->pin_stage);
        h->pin_stage = nullptr;
        h->pin_cap = 0;
        HIP_TRY(hipHostMalloc(&h->pin_stage, need * 2, hipHostMallocDefault));
        h->pin_cap = need * 2;
    }
    int rc = h->s_models.reserve(b_models);
    if (rc) return rc;
    rc = h->s_blk.reserve(b_ints);
    if (rc) return rc;
    SplinePieceModel *hm = (SplinePieceModel *)h->pin_stage;
    int *h_end = (int *)((char *)h->pin_stage + b_models), *h_piece = h_end + np, *h_first = h_piece + blocks;
    long b = 0;
    for (int i = 0; i < np; ++i) {
        pcx_bary *pc = h->pieces[i];
        SplinePieceModel mm;
        mm.snodes = pc->d_snodes; mm.nodes = pc->d_nodes; mm.wts = pc->d_wts;
        mm.T = nullptr; mm.T_tab = nullptr; mm.sc = pc->small_scale;
        h_end[i] = offsets[i] + counts[i];
        if (counts[i]) {
            std::lock_guard<std::mutex> plk(pc->mu);
            pc->call_mark = pc->clock;
            std::vector<DerivedTensor *> dts(m);
            for (int s = 0; s < m; ++s) {
                rc = bary_get_tensor(pc, derivs ? derivs + (size_t)s * d : nullptr, &dts[s]);
                if (rc) return rc;
            }
            if (m > 1) {
                std::vector<double *> tab(m);
                for (int s = 0; s < m; ++s) tab[s] = dts[s]->plain;
                if (tab != pc->tab_host) {
                    HIP_TRY(hipDeviceSynchronize());            // earlier launches (any stream) may still read d_tab
                    HIP_TRY(hipMemcpy(pc->d_tab, tab.data(), m * sizeof(double *), hipMemcpyHostToDevice));
                    pc->tab_host = tab;
                }
                mm.T_tab = pc->d_tab;
            } else {
                mm.T = dts[0]->plain;
            }
            for (int k = 0; k < (counts[i] + 63) / 64; ++k, ++b) { h_piece[b] = i; h_first[b] = offsets[i] + 64 * k; }
        }
        hm[i] = mm;
    }
    HIP_TRY(hipMemcpyAsync(h->s_models.ptr, hm, b_models, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->s_blk.ptr, h_end, b_ints, hipMemcpyHostToDevice, h->stream));
    const int *d_end = (const int *)h->s_blk.ptr, *d_piece = d_end + np, *d_first = d_piece + blocks;
    const pcx_bary *p0 = h->pieces[0];
    const int *perm = (const int *)h->s_perm.ptr;
    const SplinePieceModel *dm = (const SplinePieceModel *)h->s_models.ptr;
    if (form == 5) rc = bary_launch_sq_pieces(p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream);
    else switch (d) {
    case 1: rc = bary_launch_small_pieces(0, p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream); break;
    case 2: rc = bary_launch_small_pieces(1, p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream); break;
    case 3: rc = bary_launch_small_pieces(2, p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream); break;
    case 4: rc = bary_launch_small_pieces(3, p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream); break;
    default: return PCX_OK;
    }
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    // the staging buffer is rewritten by the next chunk / call: its copies must have left the host
    HIP_TRY(hipStreamSynchronize(h->stream));
    *done = true;
    return PCX_OK;
}

// One chunk of device-resident points through routing, bucketing and the per-piece launches, on h->stream
// (results land in dout in point order; the launches are queued, not awaited).  Caller holds h->mu.
static int spline_eval_chunk(pcx_spline *h, const double *dp, long cnt, const int32_t *derivs, int m, double *dout) {
    const int d = h->sd.d;
    std::vector<int> counts, offsets;
    int rc = spline_bucket(h, dp, cnt, counts, offsets);
    if (rc) return rc;
    {
        bool done = false;
        rc = spline_launch_fused(h, dp, counts, offsets, derivs, m, dout, &done);
        if (rc || done) return rc;
    }
    const int *perm = (const int *)h->s_perm.ptr;
    int busy = 0;
    for (int i = 0; i < h->n_pieces; ++i) busy += counts[i] ? 1 : 0;
    // fork: with several small buckets the launches go round-robin over the side streams (each waits for the
    // bucketing on h->stream); join: h->stream waits for every side stream used.  The row kernel's split
    // scratch is per handle, so only the main stream may use it: side launches pass nullptr (no split).
    const bool fan = h->ev_fork && busy > 2 && cnt / busy < (1 << 18);
    if (fan) {
        HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
        for (int i = 0; i < pcx_spline::kSide; ++i) HIP_TRY(hipStreamWaitEvent(h->side[i], h->ev_fork, 0));
    }
    int turn = 0;
    for (int i = 0; i < h->n_pieces; ++i) {
        if (counts[i] == 0) continue;
        pcx_bary *pc = h->pieces[i];
        std::lock_guard<std::mutex> plk(pc->mu);
        pc->call_mark = pc->clock;
        std::vector<DerivedTensor *> dts(m);
        for (int s = 0; s < m; ++s) {
            rc = bary_get_tensor(pc, derivs ? derivs + (size_t)s * d : nullptr, &dts[s]);
            if (rc) return rc;
        }
        const double *const *frag_tab = dts[0]->slot;
        const int eff = bary_effective_variant(pc);
        if (m > 1 && (eff == 4 || eff == 5 || pc->mfma_ok)) {
            std::vector<double *> tab(m);
            for (int s = 0; s < m; ++s) tab[s] = (eff == 4 || eff == 5) ? dts[s]->plain : dts[s]->frag;
            if (tab != pc->tab_host) {
                HIP_TRY(hipDeviceSynchronize());            // earlier launches (any stream) may still read d_tab
                HIP_TRY(hipMemcpy(pc->d_tab, tab.data(), m * sizeof(double *), hipMemcpyHostToDevice));
                pc->tab_host = tab;
            }
            frag_tab = pc->d_tab;
        }
        hipStream_t st = fan ? h->side[turn % pcx_spline::kSide] : h->stream;
        ++turn;
        rc = bary_launch(pc, dts.data(), m, frag_tab, dp, counts[i], dout, m, 0, st, fan ? nullptr : &h->s_partial,
                         perm + offsets[i]);
        if (rc) return rc;
    }
    if (fan)
        for (int i = 0; i < pcx_spline::kSide && i < turn; ++i) {
            HIP_TRY(hipEventRecord(h->ev_join[i], h->side[i]));
            HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join[i], 0));
        }
    return PCX_OK;
}

static int spline_eval_host(pcx_spline *h, const double *pts, int64_t N, const int32_t *derivs, int m,
                            double *out) {
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0 || m < 1) return fail(PCX_ERR_INVALID, "bad N or m");
    if (N > 0 && (!pts || !out)) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (m > kMaxSpecs) {      // groups of kMaxSpecs specs, each into its columns of `out`
        if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
        std::vector<double> part;
        for (int s0 = 0; s0 < m; s0 += kMaxSpecs) {
            const int mc = std::min(kMaxSpecs, m - s0);
            part.resize((size_t)N * mc);
            int rc = spline_eval_host(h, pts, N, derivs + (size_t)s0 * h->sd.d, mc, part.data());
            if (rc) return rc;
            for (int64_t i = 0; i < N; ++i)
                memcpy(out + (size_t)i * m + s0, part.data() + (size_t)i * mc, (size_t)mc * sizeof(double));
        }
        return PCX_OK;
    }
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    const int d = h->sd.d;
    for (int64_t start = 0; start < N; start += kChunkPoints) {
        long cnt = (long)std::min<int64_t>(kChunkPoints, N - start);
        int rc = h->s_pts.reserve((size_t)cnt * d * sizeof(double));
        if (rc) return rc;
        rc = h->s_out.reserve((size_t)cnt * m * sizeof(double));
        if (rc) return rc;
        double *dp = (double *)h->s_pts.ptr, *dout = (double *)h->s_out.ptr;
        HIP_TRY(hipMemcpyAsync(dp, pts + (size_t)start * d, (size_t)cnt * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
        rc = spline_eval_chunk(h, dp, cnt, derivs, m, dout);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(out + (size_t)start * m, dout, (size_t)cnt * m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return PCX_OK;
}

// Device-resident points and results (d_pts N x d, d_out N x m, both on the handle's device).  Routing needs
// the per-piece counts on the host, so the call is synchronous: everything has finished when it returns.
static int spline_eval_dev(pcx_spline *h, const double *d_pts, int64_t N, const int32_t *derivs, int m, double *d_out) {
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0 || m < 1) return fail(PCX_ERR_INVALID, "bad N or m");
    if (N > 0 && (!d_pts || !d_out)) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (m > kMaxSpecs) {      // groups of kMaxSpecs specs (as the host-pointer path), each scattered into its columns
        if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
        HIP_TRY(hipSetDevice(h->device));
        DevBuf part;
        int rc = part.alloc((size_t)std::max<int64_t>(N, 1) * kMaxSpecs * sizeof(double));
        if (rc) return rc;
        for (int s0 = 0; s0 < m; s0 += kMaxSpecs) {
            const int mc = std::min(kMaxSpecs, m - s0);
            if ((rc = spline_eval_dev(h, d_pts, N, derivs + (size_t)s0 * h->sd.d, mc, part.as<double>()))) return rc;
            const long cnt = (long)N * mc;
            if (cnt > 0) {
                hipLaunchKernelGGL(k_scatter_columns, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream,
                                   part.as<double>(), (long)N, mc, d_out, (long)m, (long)s0);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(h->stream));
            }
        }
        return PCX_OK;
    }
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    const int d = h->sd.d;
    for (int64_t start = 0; start < N; start += kChunkPoints) {
        long cnt = (long)std::min<int64_t>(kChunkPoints, N - start);
        int rc = spline_eval_chunk(h, d_pts + (size_t)start * d, cnt, derivs, m, d_out + (size_t)start * m);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PCX_OK;
}

extern "C" int pcx_spline_eval_batch_dev(pcx_spline *h, const double *d_pts, int64_t N, const int32_t *deriv,
                                         double *d_out) {
    PCX_API_BEGIN
    return spline_eval_dev(h, d_pts, N, deriv, 1, d_out);
    PCX_API_END
}

extern "C" int pcx_spline_eval_multi_batch_dev(pcx_spline *h, const double *d_pts, int64_t N,
                                               const int32_t *derivs, int m, double *d_out) {
    PCX_API_BEGIN
    if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
    return spline_eval_dev(h, d_pts, N, derivs, m, d_out);
    PCX_API_END
}

extern "C" int pcx_spline_eval_batch(pcx_spline *h, const double *pts, int64_t N, const int32_t *deriv,
                                     double *out) {
    PCX_API_BEGIN
    return spline_eval_host(h, pts, N, deriv, 1, out);
    PCX_API_END
}

extern "C" int pcx_spline_eval_multi_batch(pcx_spline *h, const double *pts, int64_t N,
                                           const int32_t *derivs, int m, double *out) {
    PCX_API_BEGIN
    if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
    return spline_eval_host(h, pts, N, derivs, m, out);
    PCX_API_END
}

extern "C" int pcx_spline_piece_ids(pcx_spline *h, const double *pts, int64_t N, int32_t *ids_out) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0) return fail(PCX_ERR_INVALID, "N < 0");
    if (N == 0) return PCX_OK;
    if (!pts || !ids_out) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (N > kChunkPoints) return fail(PCX_ERR_UNSUPPORTED, "more than %lld points", (long long)kChunkPoints);
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    const int d = h->sd.d;
    int rc = h->s_pts.reserve((size_t)N * d * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->s_pts.ptr, pts, (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
    std::vector<int> counts, offsets;
    rc = spline_bucket(h, (const double *)h->s_pts.ptr, (long)N, counts, offsets);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(ids_out, h->s_piece.ptr, (size_t)N * sizeof(int), hipMemcpyDeviceToHost));
    return PCX_OK;
    PCX_API_END
}

// ---------------------------------------------------------------------------------
// slider handle (reference slider.py:80-341): slides are borrowed pcx_bary handles
// ---------------------------------------------------------------------------------
struct pcx_slider {
    int device = 0;
    hipStream_t stream = nullptr;
    int d = 0;
    double pivot = 0.0;
    std::vector<pcx_bary *> slides;      // borrowed
    std::vector<SliderCols> cols;        // the point columns slide s reads
    std::vector<int> owner;              // dimension -> slide
    int max_cols = 1;
    std::mutex mu;
    Scratch s_pts, s_out, s_cols, s_vals, s_partial;
};

extern "C" int pcx_slider_destroy(pcx_slider *h) {
    PCX_API_BEGIN
    if (!h) return PCX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->s_pts.release(); h->s_out.release(); h->s_cols.release(); h->s_vals.release(); h->s_partial.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_slider_create(int device, int d, int n_slides, pcx_bary *const *slides,
                                 const int32_t *group_sizes, const int32_t *group_dims_cat, double pivot_value,
                                 pcx_slider **out) {
    PCX_API_BEGIN
    if (!out) return fail(PCX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (d < 1 || d > 4096 || n_slides < 1 || !slides || !group_sizes || !group_dims_cat)
        return fail(PCX_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    pcx_slider *h = new (std::nothrow) pcx_slider();
    if (!h) return fail(PCX_ERR_NOMEM, "out of host memory");
    h->device = device;
    h->d = d;
    h->pivot = pivot_value;
    h->owner.assign(d, -1);
    long at = 0;
    for (int s = 0; s < n_slides; ++s) {
        const int g = group_sizes[s];
        if (g < 1 || g > PCX_MAX_DIMS) { delete h; return fail(PCX_ERR_INVALID, "slide %d has %d dimensions (1..%d)", s, g, PCX_MAX_DIMS); }
        if (!slides[s] || slides[s]->device != device || slides[s]->dims.d != g) { delete h; return fail(PCX_ERR_INVALID, "slide %d is NULL, on another device or not %d-dimensional", s, g); }
        SliderCols c;
        c.nc = g;
        for (int k = 0; k < PCX_MAX_DIMS; ++k) c.col[k] = 0;
        for (int k = 0; k < g; ++k) {
            const int dim = group_dims_cat[at + k];
            if (dim < 0 || dim >= d || h->owner[dim] != -1) { delete h; return fail(PCX_ERR_INVALID, "partition must cover each dimension exactly once (slide %d, entry %d)", s, k); }
            h->owner[dim] = s;
            c.col[k] = dim;
        }
        at += g;
        h->slides.push_back(slides[s]);
        h->cols.push_back(c);
        h->max_cols = std::max(h->max_cols, g);
    }
    for (int k = 0; k < d; ++k)
        if (h->owner[k] < 0) { delete h; return fail(PCX_ERR_INVALID, "partition must cover each dimension exactly once (dimension %d missing)", k); }
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { int c = fail(PCX_ERR_HIP, "slider create: %s", hipGetErrorString(e)); pcx_slider_destroy(h); return c; }
    *out = h;
    return PCX_OK;
    PCX_API_END
}

// slide s at the gathered columns of dp, spec `sub` (the slide's own dimensions), into dout[p * ostride + ooff]
static int slider_launch_slide(pcx_slider *h, int s, const double *dp, long cnt, const int32_t *sub, double *dout,
                               long ostride, long ooff) {
    pcx_bary *pc = h->slides[s];
    const SliderCols &c = h->cols[s];
    double *cols = (double *)h->s_cols.ptr;
    const long elems = cnt * c.nc;
    hipLaunchKernelGGL(k_gather_columns, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, h->stream, dp, cnt, h->d, c, cols);
    HIP_TRY(hipGetLastError());
    std::lock_guard<std::mutex> plk(pc->mu);
    pc->call_mark = pc->clock;
    DerivedTensor *dt = nullptr;
    int rc = bary_get_tensor(pc, sub, &dt);
    if (rc) return rc;
    return bary_launch(pc, &dt, 1, dt->slot, cols, cnt, dout, ostride, ooff, h->stream, &h->s_partial);
}

// One chunk of device-resident points, m specs, results into dout (cnt x m); queued on h->stream.
static int slider_eval_chunk(pcx_slider *h, const double *dp, long cnt, const int32_t *derivs, int m, double *dout) {
    const int ns = (int)h->slides.size();
    int rc = h->s_cols.reserve((size_t)cnt * h->max_cols * sizeof(double));
    if (rc) return rc;
    const unsigned blocks = (unsigned)((cnt + 255) / 256);
    bool have_values = false;
    for (int q = 0; q < m; ++q) {
        const int32_t *spec = derivs ? derivs + (size_t)q * h->d : nullptr;
        // the slides that own a differentiated dimension: more than one -> the mixed partial is identically zero
        int active = -1, n_active = 0;
        if (spec)
            for (int k = 0; k < h->d; ++k) {
                if (spec[k] < 0) return fail(PCX_ERR_INVALID, "derivative order %d at dim %d", spec[k], k);
                if (spec[k] > 0 && h->owner[k] != active) {
                    bool counted = false;
                    for (int j = 0; j < k; ++j) counted = counted || (spec[j] > 0 && h->owner[j] == h->owner[k]);
                    if (!counted) ++n_active;
                    active = h->owner[k];
                }
            }
        if (n_active > 1) {
            hipLaunchKernelGGL(k_fill_strided, dim3(blocks), dim3(256), 0, h->stream, dout, cnt, (long)m, (long)q, 0.0);
            HIP_TRY(hipGetLastError());
        } else if (n_active == 1) {
            int32_t sub[PCX_MAX_DIMS];
            for (int k = 0; k < h->cols[active].nc; ++k) sub[k] = spec[h->cols[active].col[k]];
            rc = slider_launch_slide(h, active, dp, cnt, sub, dout, m, q);
            if (rc) return rc;
        } else {
            if (!have_values) {            // the slides' values are shared by every value spec of the call
                rc = h->s_vals.reserve((size_t)cnt * ns * sizeof(double));
                if (rc) return rc;
                for (int s = 0; s < ns; ++s) {
                    rc = slider_launch_slide(h, s, dp, cnt, nullptr, (double *)h->s_vals.ptr, ns, s);
                    if (rc) return rc;
                }
                have_values = true;
            }
            hipLaunchKernelGGL(k_slider_sum, dim3(blocks), dim3(256), 0, h->stream, (const double *)h->s_vals.ptr, cnt, ns,
                               h->pivot, dout, (long)m, (long)q);
            HIP_TRY(hipGetLastError());
        }
    }
    return PCX_OK;
}

extern "C" int pcx_slider_eval_multi_batch(pcx_slider *h, const double *pts, int64_t N, const int32_t *derivs, int m,
                                           double *out) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0 || m < 1) return fail(PCX_ERR_INVALID, "bad N or m");
    if (N == 0) return PCX_OK;
    if (!pts || !out) return fail(PCX_ERR_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    const int64_t chunk = std::max<int64_t>(1, kChunkPoints / std::max(1, m));
    for (int64_t start = 0; start < N; start += chunk) {
        const long cnt = (long)std::min<int64_t>(chunk, N - start);
        int rc = h->s_pts.reserve((size_t)cnt * h->d * sizeof(double));
        if (rc) return rc;
        rc = h->s_out.reserve((size_t)cnt * m * sizeof(double));
        if (rc) return rc;
        double *dp = (double *)h->s_pts.ptr, *dout = (double *)h->s_out.ptr;
        HIP_TRY(hipMemcpyAsync(dp, pts + (size_t)start * h->d, (size_t)cnt * h->d * sizeof(double), hipMemcpyHostToDevice, h->stream));
        rc = slider_eval_chunk(h, dp, cnt, derivs, m, dout);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(out + (size_t)start * m, dout, (size_t)cnt * m * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_slider_eval_batch(pcx_slider *h, const double *pts, int64_t N, const int32_t *deriv, double *out) {
    PCX_API_BEGIN
    return pcx_slider_eval_multi_batch(h, pts, N, deriv, 1, out);
    PCX_API_END
}

// Device-resident points (N x d) and results (N x m); synchronous on return.
extern "C" int pcx_slider_eval_multi_batch_dev(pcx_slider *h, const double *d_pts, int64_t N, const int32_t *derivs,
                                               int m, double *d_out) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0 || m < 1) return fail(PCX_ERR_INVALID, "bad N or m");
    if (N == 0) return PCX_OK;
    if (!d_pts || !d_out) return fail(PCX_ERR_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    for (int64_t start = 0; start < N; start += kChunkPoints) {
        const long cnt = (long)std::min<int64_t>(kChunkPoints, N - start);
        int rc = slider_eval_chunk(h, d_pts + (size_t)start * h->d, cnt, derivs, m, d_out + (size_t)start * m);
        if (rc) return rc;
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return PCX_OK;
    PCX_API_END
}
