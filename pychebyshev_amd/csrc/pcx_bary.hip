// pcx_bary.hip -- C ABI of libpcx_hip.so (see include/pcx.h): the barycentric handle.  gfx950 only.
//
// Host side: argument validation, device buffers, launch planning, kernel launches, host-pointer pipelines and the
// single-process fan-out.  No CPU arithmetic fallback lives here: every numeric result comes from a HIP kernel.

#include "pcx_bary_internal.h"
#include "bary_kernels.h"
#include "gather_kernels.h"

// ---------------------------------------------------------------------------------
// barycentric handle
// ---------------------------------------------------------------------------------
// How many dim-0 orders above its base tensor's a slab GEMM serves.  Differentiating AFTER the contraction (as the
// reference's vectorized_eval_multi does) rounds differently from the reference's batch path, which differentiates the
// tensor first: each D_0 applied to the partial sums amplifies their rounding by ~|D_0| |P| / |result|.  One level keeps
// 5-D Black-Scholes delta / vanna within 2e-13 of the reference's batch result; two levels put gamma at 4.4e-12 --
// outside the 1e-12 bar -- so the default is 1 (price + delta share a GEMM, gamma keeps its own);
// PCX_BARY_G0_SPAN=2 trades that for one GEMM less, 0 switches the grouping off.
static const int g_g0_span_default = [] { const char *e = getenv("PCX_BARY_G0_SPAN"); return e ? std::min(8, std::max(0, atoi(e))) : 1; }();

// Largest measured deviation of a shared spec from its own GEMM (relative to the probe batch's scale) at which a pair is
// still formed.  3e-13 keeps a factor of three to the 1e-12 parity bar for whatever batch and pairing order follow
// (PCX_BARY_GROUP_TOL / pcx_bary_set_group_tolerance override it).
static const double g_group_tol_default = [] { const char *e = getenv("PCX_BARY_GROUP_TOL"); double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 3e-13; }();

static const long kSmallTensorElems = 4096;   // auto: tensors up to this size run on k_bary_small
static const int kCacheSpecs = 96;    // derivative tensors kept per handle besides the untransformed one

// every k-step count up to 32 is instantiated: no padding of the folded K axis beyond 4;
// 36..64 (one column tile per wave only: the B operands alone are up to 128 VGPRs) let two
// tail dimensions of 12..16 nodes fold into K
static const int kKsList[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22,
                              23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 36, 40, 44, 48, 52, 56, 60, 64};

static int pick_ks(int K) {
    int need = (K + 3) / 4;
    for (int ks : kKsList)
        if (ks >= need) return ks;
    return -1;
}

// choose the head/tail split minimising the estimated time: MT row tiles, each KS MFMAs plus
// an epilogue (head-weight look-ups and products) worth about 5 MFMAs (measured on 15^4,
// tools/bary_rate_probe.py); the single-column-tile kernels re-read A twice as often
static bool plan_mfma(const BaryDims &dm, BaryMfmaPlan &best) {
    bool found = false;
    long best_cost = 0;
    for (int split = std::max(0, dm.d - 2 * PCX_CODE_FIELDS); split < dm.d; ++split) {
        if (split > 2 * PCX_CODE_FIELDS) continue;  // head dims must fit the two words of a row code
        long M = 1, K = 1, head_rows = 0, tail_rows = 0;
        for (int k = 0; k < split; ++k) { M *= dm.n[k]; head_rows += dm.n[k]; }
        for (int k = split; k < dm.d; ++k) { K *= dm.n[k]; tail_rows += dm.n[k]; }
        if (K > 256 || M > (1 << 24)) continue;
        if (head_rows > PCX_MAX_PART_ROWS || tail_rows > PCX_MAX_PART_ROWS) continue;   // 8-bit code fields per table part
        int ks = pick_ks((int)K);
        if (ks < 0) continue;
        long mt = (M + 15) / 16;
        long cost = mt * (ks + 5) * (ks > 32 ? 23 : 20);
        if (!found || cost < best_cost || (cost == best_cost && K > best.K)) {
            found = true;
            best_cost = cost;
            best.split = split; best.M = (int)M; best.K = (int)K; best.MT = (int)mt; best.KS = ks;
            best.tail_base = (int)head_rows + 1;
            best.rows = dm.sum_n + 2;
        }
    }
    return found;
}

static size_t mfma4_lds_bytes(const BaryDims &dm, int ks) {
    return ((size_t)8 * (dm.sum_n + 2) * 32 + (size_t)2 * ks * 64) * sizeof(double);
}

static size_t mfma_lds_bytes(const BaryDims &dm, int nt) {
    return (size_t)4 * (dm.sum_n + 2) * 16 * nt * sizeof(double);
}

extern "C" int pcx_bary_destroy(pcx_bary *h) {
    PCX_API_BEGIN
    if (!h) return PCX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &kv : h->cache) kv.second.free_all();
    (void)hipFree(h->d_tab);
    h->s_partial.release();
    h->pin.release();
    (void)hipFree(h->d_nodes); (void)hipFree(h->d_wts); (void)hipFree(h->d_diff);
    (void)hipFree(h->d_snodes);
    (void)hipFree(h->d_gsnodes);
    (void)hipFree(h->d_rowcode); (void)hipFree(h->d_kcode);
    (void)hipFree(h->d_rowcode_hi); (void)hipFree(h->d_kcode_hi);
    (void)hipFree(h->d_rowcode_g0);
    for (pcx_bary *&r : h->rot) { if (r) pcx_bary_destroy(r); r = nullptr; }
    h->s_rot.release(); h->s_rot2.release();
    h->s_pts.release(); h->s_out.release();
    h->s_pts2.release(); h->s_out2.release();
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PCX_OK;
    PCX_API_END
}

// Packs dt.plain into MFMA fragments; on failure dt.frag / dt.slot are released again.
static int bary_pack(pcx_bary *h, DerivedTensor &dt) {
    if (!h->mfma_ok) return PCX_OK;
    const BaryMfmaPlan &p = h->plan;
    size_t cnt = h->kfold_ok ? bary_kfold_frag_count(h->kf) : (size_t)(h->grid_ok ? h->gp.MT : p.MT) * p.KS * 64;
    DevBuf frag, slot;
    int rc = frag.alloc(cnt * sizeof(double));
    if (rc) return rc;
    if (h->kfold_ok) {
        if ((rc = bary_pack_kfold(h, dt.plain, frag.as<double>()))) return rc;
    } else if (h->grid_ok) {
        if ((rc = bary_pack_grid(h, dt.plain, frag.as<double>()))) return rc;
    } else {
        int blocks = (int)((cnt + 255) / 256);
        hipLaunchKernelGGL(k_pack_fragments, dim3(blocks), dim3(256), 0, h->stream, dt.plain, frag.as<double>(),
                           p.M, p.K, p.MT, p.KS);
        HIP_TRY(hipGetLastError());
    }
    if ((rc = slot.alloc(sizeof(double *)))) return rc;
    double *fp = frag.as<double>();
    HIP_TRY(hipMemcpy(slot.p, &fp, sizeof(double *), hipMemcpyHostToDevice));
    dt.frag = frag.release<double>();
    dt.slot = slot.release<double *>();
    return PCX_OK;
}

extern "C" int pcx_bary_create(int device, int d, const int32_t *n_nodes, const double *nodes_cat,
                               const double *weights_cat, const double *diffmat_cat,
                               const double *tensor, pcx_bary **out) {
    PCX_API_BEGIN
    if (!out) return fail(PCX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (d < 1 || d > PCX_MAX_DIMS) return fail(PCX_ERR_INVALID, "d=%d outside [1, %d]", d, PCX_MAX_DIMS);
    if (!n_nodes || !nodes_cat || !weights_cat || !diffmat_cat || !tensor)
        return fail(PCX_ERR_INVALID, "NULL model array");
    int rc = use_device(device);
    if (rc) return rc;
    pcx_bary *h = new (std::nothrow) pcx_bary();
    if (!h) return fail(PCX_ERR_NOMEM, "out of host memory");
    h->device = device;
    h->dims.d = d;
    h->g0_span = g_g0_span_default;
    h->group_tol = g_group_tol_default;
    long total = 1, sum_n = 0, sum_n2 = 0;
    for (int k = 0; k < d; ++k) {
        if (n_nodes[k] < 1 || n_nodes[k] > 4096) { delete h; return fail(PCX_ERR_INVALID, "n_nodes[%d]=%d outside [1, 4096]", k, n_nodes[k]); }
        h->dims.n[k] = n_nodes[k];
        h->dims.off[k] = (int)sum_n;
        h->doff.push_back((int)sum_n2);
        sum_n += n_nodes[k];
        sum_n2 += (long)n_nodes[k] * n_nodes[k];
        total *= n_nodes[k];
        if (total > (1L << 33)) { delete h; return fail(PCX_ERR_UNSUPPORTED, "tensor larger than 2^33 elements"); }
    }
    for (int k = d; k < PCX_MAX_DIMS; ++k) { h->dims.n[k] = 1; h->dims.off[k] = 0; }
    h->dims.sum_n = (int)sum_n;
    h->total = total;

#define CREATE_TRY(expr)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            int c_ = fail(PCX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
            pcx_bary_destroy(h);                                                           \
            return c_;                                                                     \
        }                                                                                  \
    } while (0)

    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipMalloc((void **)&h->d_nodes, sum_n * sizeof(double)));
    CREATE_TRY(hipMalloc((void **)&h->d_wts, sum_n * sizeof(double)));
    CREATE_TRY(hipMalloc((void **)&h->d_diff, sum_n2 * sizeof(double)));
    CREATE_TRY(hipMemcpy(h->d_nodes, nodes_cat, sum_n * sizeof(double), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->d_wts, weights_cat, sum_n * sizeof(double), hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->d_diff, diffmat_cat, sum_n2 * sizeof(double), hipMemcpyHostToDevice));

    // rows kernel geometry: lanes per point = smallest power of two >= number of rows
    long Mrows = total / h->dims.n[d - 1];
    int lpp = 1;
    while (lpp < 64 && lpp < Mrows) lpp <<= 1;
    // its LDS weight table is (256 / lpp) points x sum_n doubles: widen the groups until it fits
    while (lpp < 64 && (size_t)(256 / lpp) * sum_n * sizeof(double) > 48 * 1024) lpp <<= 1;
    h->lpp = lpp;

    // MFMA plan + row/k codes
    h->mfma_ok = plan_mfma(h->dims, h->plan);
    if (h->mfma_ok) {
        // two column tiles per wave while the B operands fit the register file beside them: up to 32 k-steps, 36 and 40 with
        // narrow codes (232 ... 256 VGPRs at two waves per SIMD; round 4, one against two column tiles: 12^4 0.642 -> 0.683,
        // 12 x 12 x 10 x 16 0.629 -> 0.682, 6 x 6 x 6 x 12 x 12 0.644 -> 0.692 -- each fragment load feeds two matrix instructions)
        const bool narrow = h->plan.split <= PCX_CODE_FIELDS && d - h->plan.split <= PCX_CODE_FIELDS;
        h->nt = (h->plan.KS <= 32 || (narrow && h->plan.KS <= 40)) ? 2 : 1;
        if (mfma_lds_bytes(h->dims, h->nt) > 150 * 1024) h->nt = 1;
        if (mfma_lds_bytes(h->dims, h->nt) > 150 * 1024) h->mfma_ok = false;
    }
    // shapes no kernel covers fail here, at create, not at the first evaluation: the rows kernel
    // keeps (256 / lpp) x sum_n weights in LDS
    if (!h->mfma_ok && (size_t)(256 / h->lpp) * sum_n * sizeof(double) > 160 * 1024) {
        int c_ = fail(PCX_ERR_UNSUPPORTED, "sum of node counts %ld too large for any kernel (MFMA plan: each of the "
                      "head / tail parts <= %d rows and a tail product <= 256; row kernel: sum <= 5120)", sum_n, PCX_MAX_PART_ROWS);
        pcx_bary_destroy(h);
        return c_;
    }
    h->mfma4_ok = h->mfma_ok && mfma4_lds_bytes(h->dims, h->plan.KS) <= 160 * 1024 &&
                  h->plan.KS <= 32 && h->plan.split <= PCX_CODE_FIELDS && d - h->plan.split <= PCX_CODE_FIELDS;
    // lane-per-point kernel (k_bary_small): d <= 4, last dimension <= 64 nodes (weights in registers),
    // outer weights table (sum of outer n) x 64 lanes x 8 B within 64 KB.  Preferred by auto while the
    // tensor is small enough that the MFMA kernel's prologue outweighs its tiles
    // (tools/bary_rate_probe.py, profiles/r02_bary_rate_probe.txt).
    {
        // every node count up to 16 has its own instantiation (no padding, no per-node tests); classes above
        static const int kNlp[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 24, 32, 48, 64};
        const int nl = h->dims.n[d - 1];
        const long outer_rows = sum_n - nl;
        if (d <= 4 && nl <= 64 && outer_rows * 64 * (long)sizeof(double) <= 64 * 1024 && total <= (1L << 22)) {
            for (int v : kNlp)
                if (v >= nl) { h->small_nlp = v; break; }
            // ... and 2-D tensors with a last dimension of 49 ... 64 nodes while the first has at most 48 (round 4,
            // lane-per-point / MFMA in 1e9 pts/s: 20 x 64 5.8 / 2.2, 30 x 50 3.5 / 2.6, 40 x 64 2.7 / 1.9, 48 x 64 2.1 / 1.9; 50^2
            // 2.0 / 2.1, 60^2 1.6 / 1.9, 64^2 1.6 / 1.7; 3-D shapes of that kind -- 8 x 8 x 50 -- stay on the MFMA kernel)
            h->small_preferred = total <= kSmallTensorElems && (nl <= 48 || (d == 2 && n_nodes[0] <= 48));
            // mid-size tensors with equal trailing node counts: both trailing weight vectors in registers (k_bary_sq)
            if (d >= 2 && n_nodes[d - 2] == nl && ((nl >= 4 && nl <= 24) || nl == 26 || nl == 28 || nl == 30 || nl == 32) &&
                (outer_rows - nl) * 64 * (long)sizeof(double) <= 48 * 1024) {
                h->sq_nl = nl;
                static const bool sq_auto = [] { const char *e = getenv("PCX_BARY_SQ"); return !(e && e[0] == '0'); }();
                // tools/bary_rate_probe.py (profiles/r03_bary_rate_probe.txt): ahead of k_bary_small everywhere it applies
                // (12^2 +33 %, 8^3 +37 %, 11^3 +44 %, 6^4 +50 %) and of the MFMA kernel's short plans for d <= 3
                // (17^3 +48 %, 20^3 +11 %, 24^3 +15 %); from 10^4 up the MFMA kernel (K = n^2 >= 100) is ahead
                // 21 and 23 nodes: hipcc runs out of scalar registers on the odd row length (SGPR spills in the block,
                // 0.37 / 0.36 of the peak against 0.39 / 0.44 on the MFMA kernel): available, not preferred
                // 26 / 28 / 30 nodes: ahead in 2-D (26^2 0.40 against 0.21), behind the MFMA kernel in 3-D (30^3 0.32 against 0.42)
                // round 4 (k_bary_mfma_grid): 20^3 0.45 -> 0.48, 24^3 0.545 -> 0.56, 32^3 0.48 -> 0.63 on the MFMA kernel
                // (21 nodes: 0.44 here against 0.40 on the grid MFMA kernel -- preferred again; 23: 0.465 against 0.47)
                h->sq_preferred = sq_auto && (d <= 3 || total <= kSmallTensorElems) && nl != 23 &&
                                  !(d >= 3 && nl >= 24) && !(d == 3 && nl == 20 && n_nodes[0] == 20);      // 24^3: grid 0.56, here 0.545
            }
            // 2^e ~ 2 / (node span): exact to apply, keeps the prefix / suffix products of the weights in range
            std::vector<double> sn((size_t)sum_n);
            for (int k = 0; k < d; ++k) {
                const double *nd = nodes_cat + h->dims.off[k];
                const double span = nd[n_nodes[k] - 1] - nd[0];
                int e = 0;
                if (span > 0.0 && std::isfinite(span)) (void)std::frexp(2.0 / span, &e);
                const double sck = std::ldexp(1.0, e - 1);
                h->small_scale.s[k] = sck;
                for (int j = 0; j < n_nodes[k]; ++j) sn[h->dims.off[k] + j] = nd[j] * sck;
            }
            CREATE_TRY(hipMalloc((void **)&h->d_snodes, sum_n * sizeof(double)));
            CREATE_TRY(hipMemcpy(h->d_snodes, sn.data(), sum_n * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    if (h->mfma_ok) {
        const BaryMfmaPlan &p = h->plan;
        // all-ones rows: the last row of the head part (row codes) and of the tail part (k codes)
        const unsigned ones_h = (unsigned)(p.tail_base - 1), ones_t = (unsigned)(p.rows - 1 - p.tail_base);
        std::vector<unsigned> rowcode((size_t)p.MT * 16), kcode((size_t)p.KS * 4);
        std::vector<unsigned> rowcode_hi(rowcode.size()), kcode_hi(kcode.size());
        h->wide = p.split > PCX_CODE_FIELDS || d - p.split > PCX_CODE_FIELDS;
        for (long m = 0; m < (long)p.MT * 16; ++m) {
            unsigned f[2 * PCX_CODE_FIELDS] = {ones_h, ones_h, ones_h, ones_h, ones_h, ones_h, ones_h, ones_h};
            if (m < p.M) {
                long rem = m;
                for (int k = p.split - 1; k >= 0; --k) {
                    int i = (int)(rem % h->dims.n[k]);
                    rem /= h->dims.n[k];
                    f[k] = (unsigned)(h->dims.off[k] + i);
                }
            }
            rowcode[m] = f[0] | (f[1] << 8) | (f[2] << 16) | (f[3] << 24);
            rowcode_hi[m] = f[4] | (f[5] << 8) | (f[6] << 16) | (f[7] << 24);
        }
        for (long kk = 0; kk < (long)p.KS * 4; ++kk) {
            unsigned f[2 * PCX_CODE_FIELDS] = {ones_t, ones_t, ones_t, ones_t, ones_t, ones_t, ones_t, ones_t};
            if (kk < p.K) {
                long rem = kk;
                for (int k = d - 1; k >= p.split; --k) {
                    int i = (int)(rem % h->dims.n[k]);
                    rem /= h->dims.n[k];
                    f[k - p.split] = (unsigned)(h->dims.off[k] - h->dims.off[p.split] + i);   // relative to the tail part
                }
            }
            kcode[kk] = f[0] | (f[1] << 8) | (f[2] << 16) | (f[3] << 24);
            kcode_hi[kk] = f[4] | (f[5] << 8) | (f[6] << 16) | (f[7] << 24);
        }
        // device layout of the row codes: the four codes a lane needs for a tile (rows g, g+4, g+8, g+12 of
        // tile t) side by side, [t][g][j], so that one 16-byte load fetches them
        auto lane_order = [&](std::vector<unsigned> &v) {
            std::vector<unsigned> o(v.size());
            for (long t = 0; t < (long)p.MT; ++t)
                for (int g = 0; g < 4; ++g)
                    for (int j = 0; j < 4; ++j) o[(size_t)(4 * t + g) * 4 + j] = v[(size_t)16 * t + g + 4 * j];
            v.swap(o);
        };
        lane_order(rowcode);
        lane_order(rowcode_hi);
        if (h->wide) {
            CREATE_TRY(hipMalloc((void **)&h->d_rowcode_hi, rowcode_hi.size() * sizeof(unsigned)));
            CREATE_TRY(hipMalloc((void **)&h->d_kcode_hi, kcode_hi.size() * sizeof(unsigned)));
            CREATE_TRY(hipMemcpy(h->d_rowcode_hi, rowcode_hi.data(), rowcode_hi.size() * sizeof(unsigned), hipMemcpyHostToDevice));
            CREATE_TRY(hipMemcpy(h->d_kcode_hi, kcode_hi.data(), kcode_hi.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        }
        CREATE_TRY(hipMalloc((void **)&h->d_rowcode, rowcode.size() * sizeof(unsigned)));
        CREATE_TRY(hipMalloc((void **)&h->d_kcode, kcode.size() * sizeof(unsigned)));
        CREATE_TRY(hipMemcpy(h->d_rowcode, rowcode.data(), rowcode.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        CREATE_TRY(hipMemcpy(h->d_kcode, kcode.data(), kcode.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        // dim-0 groups: head = dimension 0 x (dimensions 1 .. split-1); the rows of one i0 form a slab padded to whole
        // tiles.  Needs two column tiles per wave (large batches only), narrow codes, and room for two n0-vectors
        // per point in the tail part of the LDS table (dead once the B operands are in registers); n0 <= 16 bounds the
        // rounding amplification of the D_0 step (~ n0^2 eps).
        long M1 = 1;
        for (int k = 1; k < p.split; ++k) M1 *= h->dims.n[k];
        if (!h->wide && p.split >= 2 && p.split <= PCX_CODE_FIELDS && h->nt == 2 && p.KS <= 32 && M1 >= 16 &&
            2 * h->dims.n[0] <= p.rows - p.tail_base && h->dims.n[0] >= 2 && h->dims.n[0] <= 16) {
            const int tps = (int)((M1 + 15) / 16);
            const long mtg = (long)tps * h->dims.n[0];
            // padding must stay cheap: at most 15 % more row tiles than the plain plan
            if (mtg * 100 <= (long)p.MT * 115) {
                std::vector<unsigned> rc((size_t)mtg * 16);
                for (long t = 0; t < mtg; ++t)
                    for (int r = 0; r < 16; ++r) {
                        const long m1 = (t % tps) * 16 + r;
                        unsigned f[PCX_CODE_FIELDS] = {ones_h, ones_h, ones_h, ones_h};
                        if (m1 < M1) {
                            long rem = m1;
                            for (int k = p.split - 1; k >= 1; --k) {
                                int i = (int)(rem % h->dims.n[k]);
                                rem /= h->dims.n[k];
                                f[k - 1] = (unsigned)(h->dims.off[k] + i);
                            }
                        }
                        rc[(size_t)16 * t + r] = f[0] | (f[1] << 8) | (f[2] << 16) | (f[3] << 24);
                    }
                std::vector<unsigned> o(rc.size());
                for (long t = 0; t < mtg; ++t)
                    for (int g = 0; g < 4; ++g)
                        for (int j = 0; j < 4; ++j) o[(size_t)(4 * t + g) * 4 + j] = rc[(size_t)16 * t + g + 4 * j];
                CREATE_TRY(hipMalloc((void **)&h->d_rowcode_g0, o.size() * sizeof(unsigned)));
                CREATE_TRY(hipMemcpy(h->d_rowcode_g0, o.data(), o.size() * sizeof(unsigned), hipMemcpyHostToDevice));
                h->g0_ok = true;
                h->g0_tps = tps;
                h->g0_nf = std::max(2, p.split - 1);
            }
        }
    }

    // short plans: the grid form (bary_grid_kernels.h) -- unless the shape can share GEMMs between specs one order apart
    // (dim-0 groups above), which needs the slab packing of the row-code form
    if (h->mfma_ok && !(h->plan.split > PCX_CODE_FIELDS || d - h->plan.split > PCX_CODE_FIELDS)) {
        // 3-D tensors one of whose dimensions fills whole row tiles: the k-fold form (bary_kfold_kernels.h) ahead of the grid
        // form -- and of the dim-0 groups: 9 x 48 x 30 runs at 0.30 of the peak on the row-code form a group would share, the
        // k-fold form (rows = the 48) takes every launch of it to 0.8
        long kfold_est = 0;                               // permille of the FP64 peak expected of the k-fold plan
        {
            const long keff = bary_plan_kfold(h->dims, h->kf);
            kfold_est = bary_kfold_estimate(h->kf, keff);
            long geff = 0;
            BaryGridPlan gtry;
            gtry.MT = 0;
            if (keff > 0 && !h->g0_ok) {
                // the alternative is the grid form -- or the row-code form where the grid planner declines, which it does when it
                // expects no more of its own form: priced as a grid plan either way (25 x 25 x 40: row codes 0.44, k-fold 0.6)
                (void)bary_plan_grid(h->dims, h->plan, gtry);
                if (gtry.MT > 0)
                    geff = (long)((double)h->plan.M * h->plan.K * 10000.0 / ((double)gtry.MT * 16.0 * h->plan.KS * 4.0));
            }
            h->kfold_ok = bary_kfold_take(h->kf, keff, geff, h->plan.KS);
        }
        if (h->kfold_ok) {
            const int nt_rc = h->nt;
            h->nt = 2;
            if (bary_kfold_lds_bytes(h->kf, 2) > (size_t)72 * 1024) h->nt = 1;      // two workgroups per CU
            if (bary_kfold_lds_bytes(h->kf, h->nt) > (size_t)150 * 1024) { h->kfold_ok = false; h->nt = nt_rc; }   // a very long middle dimension
        }
        // where the k-fold form is taken it is also ahead of the lane-per-point kernels (round 4, k-fold / k_bary_sq in fractions of
        // the FP64 peak: 45 x 20 x 20 0.68 / 0.31, 30 x 20 x 20 0.65 / 0.42, 32 x 16 x 16 0.65 / 0.41, 64 x 8 x 8 0.45 / 0.21, 24 x 20 x 20
        // 0.53 / 0.46, 22^3 0.49 / 0.46; without it -- 40 x 12 x 12, 26 x 14 x 14, 21^3 -- they keep their rule)
        // -- for tensors of 4,096 elements or more and plans expected at half the peak or better: (2, 17, 17) stays where it was
        if (h->kfold_ok && total >= kSmallTensorElems && kfold_est >= 500) h->sq_preferred = h->small_preferred = false;
        if (h->kfold_ok && h->g0_ok) {
            h->g0_ok = false;
            (void)hipFree(h->d_rowcode_g0);
            h->d_rowcode_g0 = nullptr;
        }
        if (!h->kfold_ok && !h->g0_ok) h->grid_ok = bary_plan_grid(h->dims, h->plan, h->gp);
        if (h->grid_ok) {
            h->nt = h->plan.KS > 32 ? 1 : 2;                       // the table is per wave and holds one part at a time
            const size_t cap = (size_t)(h->gp.wpb == 4 ? 150 : 64) * 1024;
            if (bary_grid_lds_bytes(h, h->nt) > cap) h->nt = 1;
            if (bary_grid_lds_bytes(h, h->nt) > cap) h->grid_ok = false;
        }
    }
    if (h->grid_ok || h->kfold_ok) {
        // division-free weights for the grid and k-fold forms (bary_weights.h; as k_bary_small): nodes scaled by 2^e ~ 2 / span
        // per dimension; PCX_BARY_GRID_PROD=0 (read per handle) keeps the weights by division (A/B measurements)
        std::vector<double> sn((size_t)sum_n + PCX_MAX_DIMS, 0.0);
        h->grid_prod = true;
        for (int k = 0; k < d; ++k) {
            const double *nd = nodes_cat + h->dims.off[k];
            const double span = nd[n_nodes[k] - 1] - nd[0];
            int e = 0;
            if (span > 0.0 && std::isfinite(span)) (void)std::frexp(2.0 / span, &e);
            const double sck = std::ldexp(1.0, e - 1);
            sn[(size_t)sum_n + k] = sck;
            for (int j = 0; j < n_nodes[k]; ++j) sn[h->dims.off[k] + j] = nd[j] * sck;
            if (n_nodes[k] > 64) h->grid_prod = false;
        }
        { const char *e = getenv("PCX_BARY_GRID_PROD"); if (e && e[0] == '0') h->grid_prod = false; }
        CREATE_TRY(hipMalloc((void **)&h->d_gsnodes, sn.size() * sizeof(double)));
        CREATE_TRY(hipMemcpy(h->d_gsnodes, sn.data(), sn.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (h->grid_ok || h->kfold_ok) h->mfma4_ok = false;

    // value tensor (derivative spec all-zero) enters the cache at create
    DerivedTensor dt;
    {
        DevBuf plain;
        rc = alloc_plain(plain, total);
        if (rc) { pcx_bary_destroy(h); return rc; }
        dt.plain = plain.release<double>();
    }
    { hipError_t e_ = hipMemcpy(dt.plain, tensor, total * sizeof(double), hipMemcpyHostToDevice);
      if (e_ != hipSuccess) { (void)hipFree(dt.plain); int c_ = fail(PCX_ERR_HIP, "tensor upload: %s", hipGetErrorString(e_)); pcx_bary_destroy(h); return c_; } }
    rc = bary_pack(h, dt);
    if (rc) { (void)hipFree(dt.plain); pcx_bary_destroy(h); return rc; }
    h->cache[std::vector<int>(d, 0)] = dt;
    CREATE_TRY(hipMalloc((void **)&h->d_tab, kMaxSpecs * sizeof(double *)));
    CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
    *out = h;
    return PCX_OK;
    PCX_API_END
}

// ---- .pcb loader (host-side parsing and grid metadata; no evaluation arithmetic) ------
static void host_grid_metadata(double lo, double hi, int n, double *x, double *w, double *D) {
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < n; ++k)   // numpy chebpts1: sin(0.5 pi / n * (-n + 1 + 2k)), ascending
        x[k] = 0.5 * (lo + hi) + 0.5 * (hi - lo) * std::sin(0.5 * pi / n * (double)(-n + 1 + 2 * k));
    std::sort(x, x + n);
    for (int i = 0; i < n; ++i) {
        double wi = 1.0;
        for (int j = 0; j < n; ++j)
            if (j != i) wi /= (x[i] - x[j]);
        w[i] = wi;
    }
    for (int i = 0; i < n; ++i) {
        double rowsum = 0.0;
        for (int j = 0; j < n; ++j) {
            double v = (i == j) ? 0.0 : w[j] / ((x[i] - x[j]) * w[i]);
            D[(size_t)i * n + j] = v;
            rowsum += v;
        }
        D[(size_t)i * n + i] = -rowsum;
    }
}

extern "C" int pcx_bary_create_from_pcb(int device, const char *path, pcx_bary **out) {
    PCX_API_BEGIN
    if (!out) return fail(PCX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!path) return fail(PCX_ERR_INVALID, "path is NULL");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(PCX_ERR_INVALID, "cannot open %s", path);
    auto bail = [&](const char *why) { fclose(f); return fail(PCX_ERR_INVALID, "%s: %s", path, why); };
    unsigned char head[12];
    if (fread(head, 1, 12, f) != 12) return bail("shorter than the 12-byte .pcb header");
    if (memcmp(head, "PCB\0", 4) != 0) return bail("not a PyChebyshev binary file (bad magic)");
    if (head[4] != 1) return bail("unsupported .pcb major version");
    if ((head[6] | (head[7] << 8)) != 1) return bail("class tag is not ChebyshevApproximation");
    if (head[8] | head[9] | head[10] | head[11]) return bail("reserved header bytes nonzero");
    uint32_t d = 0;
    if (fread(&d, 4, 1, f) != 1) return bail("unexpected EOF reading num_dimensions");
    if (d < 1 || d > PCX_MAX_DIMS) return bail("num_dimensions outside [1, 16]");
    std::vector<double> lo(d), hi(d);
    std::vector<uint32_t> nn(d);
    if (fread(lo.data(), 8, d, f) != d || fread(hi.data(), 8, d, f) != d || fread(nn.data(), 4, d, f) != d)
        return bail("unexpected EOF reading domain / n_nodes");
    size_t total = 1, sum_n = 0, sum_n2 = 0;
    std::vector<int32_t> n(d);
    for (uint32_t k = 0; k < d; ++k) {
        if (!(lo[k] < hi[k])) return bail("domain lo must be < hi");
        if (nn[k] < 1 || nn[k] > 4096) return bail("n_nodes outside [1, 4096]");
        n[k] = (int32_t)nn[k];
        total *= nn[k];
        sum_n += nn[k];
        sum_n2 += (size_t)nn[k] * nn[k];
        if (total > ((size_t)1 << 33)) return bail("tensor larger than 2^33 elements");
    }
    std::vector<double> tensor(total);
    if (fread(tensor.data(), 8, total, f) != total) return bail("unexpected EOF reading tensor_values");
    fclose(f);
    for (size_t i = 0; i < total; ++i)
        if (!std::isfinite(tensor[i])) return fail(PCX_ERR_INVALID, "%s: tensor_values contains NaN or Inf", path);
    std::vector<double> nodes(sum_n), wts(sum_n), diff(sum_n2);
    size_t o1 = 0, o2 = 0;
    for (uint32_t k = 0; k < d; ++k) {
        host_grid_metadata(lo[k], hi[k], n[k], nodes.data() + o1, wts.data() + o1, diff.data() + o2);
        o1 += n[k];
        o2 += (size_t)n[k] * n[k];
    }
    int rc = pcx_bary_create(device, (int)d, n.data(), nodes.data(), wts.data(), diff.data(), tensor.data(), out);
    if (rc == PCX_OK) { (*out)->dom_lo = lo; (*out)->dom_hi = hi; }
    return rc;
    PCX_API_END
}

// .pcb v1 writer (reference _binary.py:208-283, write side): 12-byte header, d, lower bounds,
// upper bounds, n_nodes, tensor_values in C order -- all little-endian, no padding.  The tensor
// is the handle's untransformed device copy, so load -> save reproduces the file byte for byte.
extern "C" int pcx_bary_save_pcb(pcx_bary *h, const char *path, const double *lo, const double *hi) {
    PCX_API_BEGIN
    if (!h || !path) return fail(PCX_ERR_INVALID, "NULL argument");
    const int d = h->dims.d;
    if ((lo == nullptr) != (hi == nullptr)) return fail(PCX_ERR_INVALID, "pass both domain bounds or neither");
    if (!lo) {
        if ((int)h->dom_lo.size() != d)
            return fail(PCX_ERR_INVALID, "the handle does not know its domain (not loaded from a .pcb file): pass lo / hi");
        lo = h->dom_lo.data();
        hi = h->dom_hi.data();
    }
    for (int k = 0; k < d; ++k)
        if (!(lo[k] < hi[k])) return fail(PCX_ERR_INVALID, "domain[%d]: lo must be < hi", k);
    std::vector<double> tensor((size_t)h->total);
    {
        HIP_TRY(hipSetDevice(h->device));
        std::lock_guard<std::mutex> lk(h->mu);
        const DerivedTensor &base = h->cache[std::vector<int>(d, 0)];
        HIP_TRY(hipMemcpy(tensor.data(), base.plain, (size_t)h->total * sizeof(double), hipMemcpyDeviceToHost));
    }
    FILE *f = fopen(path, "wb");
    if (!f) return fail(PCX_ERR_INVALID, "cannot open %s for writing", path);
    const unsigned char head[12] = {'P', 'C', 'B', 0, 1, 0, 1, 0, 0, 0, 0, 0};   // magic, major 1, minor 0, class tag 1
    const uint32_t du = (uint32_t)d;
    std::vector<uint32_t> nn(d);
    for (int k = 0; k < d; ++k) nn[k] = (uint32_t)h->dims.n[k];
    bool ok = fwrite(head, 1, 12, f) == 12 && fwrite(&du, 4, 1, f) == 1 && fwrite(lo, 8, d, f) == (size_t)d &&
              fwrite(hi, 8, d, f) == (size_t)d && fwrite(nn.data(), 4, d, f) == (size_t)d &&
              fwrite(tensor.data(), 8, tensor.size(), f) == tensor.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(PCX_ERR_INVALID, "short write to %s", path);
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_bary_shape(pcx_bary *h, int32_t *d_out, int32_t *n_nodes_out) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (d_out) *d_out = h->dims.d;
    if (n_nodes_out)
        for (int k = 0; k < PCX_MAX_DIMS; ++k) n_nodes_out[k] = k < h->dims.d ? h->dims.n[k] : 0;
    return PCX_OK;
    PCX_API_END
}

// Returns (building on first use) the derivative-transformed tensor for `deriv`.
// Caller holds h->mu.  Work is enqueued on h->stream.
PCX_HIDDEN int bary_get_tensor(pcx_bary *h, const int32_t *deriv, DerivedTensor **out) {
    const int d = h->dims.d;
    std::vector<int> key(d, 0);
    if (deriv)
        for (int k = 0; k < d; ++k) {
            if (deriv[k] < 0 || deriv[k] > 8) return fail(PCX_ERR_INVALID, "derivative order %d at dim %d outside [0, 8]", deriv[k], k);
            key[k] = deriv[k];
        }
    auto it = h->cache.find(key);
    if (it != h->cache.end()) {
        it->second.last_use = ++h->clock;
        *out = &it->second;
        return PCX_OK;
    }
    // The cache holds the untransformed tensor plus up to kCacheSpecs derivative tensors; beyond
    // that the least recently used one that the current call has not asked for is dropped.
    // Kernels reading it may still be queued (on any stream of a _dev caller): drain the device first.
    if (h->cache.size() > (size_t)kCacheSpecs) {
        auto victim = h->cache.end();
        for (auto c = h->cache.begin(); c != h->cache.end(); ++c) {
            bool is_base = true;
            for (int v : c->first) is_base = is_base && v == 0;
            if (is_base || c->second.last_use > h->call_mark) continue;
            if (victim == h->cache.end() || c->second.last_use < victim->second.last_use) victim = c;
        }
        if (victim == h->cache.end())
            return fail(PCX_ERR_UNSUPPORTED, "more than %d distinct derivative specs in one call", kCacheSpecs);
        HIP_TRY(hipDeviceSynchronize());
        if (!h->tab_host.empty()) h->tab_host.clear();      // the multi-spec table may name the victim
        victim->second.free_all();
        h->cache.erase(victim);
    }

    DerivedTensor &base = h->cache[std::vector<int>(d, 0)];
    DevBuf cur, tmp;
    int rc = alloc_plain(cur, h->total);
    if (rc) return rc;
    if ((rc = alloc_plain(tmp, h->total))) return rc;
    HIP_TRY(hipMemcpyAsync(cur.p, base.plain, h->total * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    // barycentric.py:982-989: dims descending, order[k] passes each
    for (int k = d - 1; k >= 0; --k) {
        long outer = 1, inner = 1;
        for (int q = 0; q < k; ++q) outer *= h->dims.n[q];
        for (int q = k + 1; q < d; ++q) inner *= h->dims.n[q];
        for (int r = 0; r < key[k]; ++r) {
            int blocks = (int)((h->total + 255) / 256);
            hipLaunchKernelGGL(k_mode_product, dim3(blocks), dim3(256), 0, h->stream, cur.as<double>(), tmp.as<double>(),
                               h->d_diff + h->doff[k], outer, h->dims.n[k], inner);
            std::swap(cur.p, tmp.p);
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    DerivedTensor dt;
    dt.plain = cur.as<double>();
    if ((rc = bary_pack(h, dt))) return rc;          // cur still owns the tensor: freed on this path
    (void)cur.release<double>();
    dt.last_use = ++h->clock;
    auto ins = h->cache.emplace(key, dt);
    *out = &ins.first->second;
    return PCX_OK;
}

// One MFMA launch for m specs (frag_tab: device table of m fragment pointers).  Small
// batches are split over grid.y (chunks of row tiles) so that a handful of points still
// uses the whole chip; the per-chunk totals are then added by k_bary_reduce in the fixed
// chunk order, which makes every result independent of the batch size.
template <int KS, int NT, bool WIDE, int NF = 4>
static int launch_mfma_t(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N,
                         double *d_out, long ostride, long ooff, hipStream_t st, Scratch *split_scratch,
                         const int *perm) {
    const bool allow_split = split_scratch != nullptr;
    size_t lds = mfma_lds_bytes(h->dims, NT);
    auto kern = k_bary_mfma<KS, NT, WIDE, NF>;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long per_wg = 4L * 16 * NT;
    long blocks = (N + per_wg - 1) / per_wg;
    if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
    int nchunks = (h->plan.MT + PCX_CHUNK_TILES - 1) / PCX_CHUNK_TILES;
    int nsplit = 1, cps = nchunks;
    const long want = 512;   // workgroups that fill 256 CUs at two per CU
    if (allow_split && blocks * m < want && nchunks > 1) {
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
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, (unsigned)nsplit, (unsigned)m), dim3(256), lds, st,
                       h->dims, h->plan, h->d_nodes, h->d_wts, frag_tab, h->d_rowcode, h->d_kcode,
                       h->d_rowcode_hi, h->d_kcode_hi, d_pts, d_out, N, ostride, ooff, cps, partial, perm, BaryG0{}, nullptr);
    HIP_TRY(hipGetLastError());
    if (nsplit > 1) {
        long cnt = N * m;
        hipLaunchKernelGGL(k_bary_reduce, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, partial, d_out,
                           N, nchunks, m, ostride, ooff, perm);
        HIP_TRY(hipGetLastError());
    }
    return PCX_OK;
}

// 4x4x4_4b form: 512-thread workgroups (8 waves x 32 points), row tiles staged through LDS.
template <int KS>
static int launch_mfma4_t(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N,
                          double *d_out, long ostride, long ooff, hipStream_t st, const int *perm) {
    size_t lds = mfma4_lds_bytes(h->dims, KS);
    auto kern = k_bary_mfma4<KS>;
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long blocks = (N + 255) / 256;
    if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, 1, (unsigned)m), dim3(512), lds, st, h->dims, h->plan,
                       h->d_nodes, h->d_wts, frag_tab, h->d_rowcode, h->d_kcode, d_pts, d_out, N, ostride, ooff, perm);
    HIP_TRY(hipGetLastError());
    return PCX_OK;
}

static int launch_mfma4(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N,
                        double *d_out, long ostride, long ooff, hipStream_t st, const int *perm) {
    switch (h->plan.KS) {
#define CASE_KS(v) case v: return launch_mfma4_t<v>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
        CASE_KS(1) CASE_KS(2) CASE_KS(3) CASE_KS(4) CASE_KS(5) CASE_KS(6) CASE_KS(7) CASE_KS(8)
        CASE_KS(9) CASE_KS(10) CASE_KS(11) CASE_KS(12) CASE_KS(13) CASE_KS(14) CASE_KS(15) CASE_KS(16)
        CASE_KS(17) CASE_KS(18) CASE_KS(19) CASE_KS(20) CASE_KS(21) CASE_KS(22) CASE_KS(23) CASE_KS(24)
        CASE_KS(25) CASE_KS(26) CASE_KS(27) CASE_KS(28) CASE_KS(29) CASE_KS(30) CASE_KS(31) CASE_KS(32)
#undef CASE_KS
    }
    return fail(PCX_ERR_UNSUPPORTED, "no MFMA instantiation for KS=%d", h->plan.KS);
}

// NF: live fields of a row code = head dimensions (1..4), known per handle: the kernel reads only those
// (16 LDS reads and multiplies fewer per row tile with a two-dimensional head; 11^5, head of three: +1.4 %).
template <int NT, bool WIDE, int NF>
static int launch_mfma_nf(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N,
                          double *d_out, long ostride, long ooff, hipStream_t st, Scratch *split_scratch,
                          const int *perm) {
    switch (h->plan.KS) {
#define CASE_KS(v) case v: return launch_mfma_t<v, NT, WIDE, NF>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
        CASE_KS(1) CASE_KS(2) CASE_KS(3) CASE_KS(4) CASE_KS(5) CASE_KS(6) CASE_KS(7) CASE_KS(8)
        CASE_KS(9) CASE_KS(10) CASE_KS(11) CASE_KS(12) CASE_KS(13) CASE_KS(14) CASE_KS(15) CASE_KS(16)
        CASE_KS(17) CASE_KS(18) CASE_KS(19) CASE_KS(20) CASE_KS(21) CASE_KS(22) CASE_KS(23) CASE_KS(24)
        CASE_KS(25) CASE_KS(26) CASE_KS(27) CASE_KS(28) CASE_KS(29) CASE_KS(30) CASE_KS(31) CASE_KS(32)
#undef CASE_KS
    }
    if constexpr (NT == 1) {
        switch (h->plan.KS) {
#define CASE_KS(v) case v: return launch_mfma_t<v, 1, WIDE, NF>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
            CASE_KS(36) CASE_KS(40) CASE_KS(44) CASE_KS(48) CASE_KS(52) CASE_KS(56) CASE_KS(60) CASE_KS(64)
#undef CASE_KS
        }
    }
    if constexpr (NT == 2 && !WIDE) {
        switch (h->plan.KS) {
#define CASE_KS(v) case v: return launch_mfma_t<v, 2, WIDE, NF>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
            CASE_KS(36) CASE_KS(40)
#undef CASE_KS
        }
    }
    return fail(PCX_ERR_UNSUPPORTED, "no MFMA instantiation for KS=%d, NT=%d", h->plan.KS, NT);
}

template <int NT, bool WIDE>
static int launch_mfma_nt(pcx_bary *h, const double *const *frag_tab, int m, const double *d_pts, long N,
                          double *d_out, long ostride, long ooff, hipStream_t st, Scratch *split_scratch,
                          const int *perm) {
    if constexpr (!WIDE) {
        if (h->plan.split <= 2)
            return launch_mfma_nf<NT, false, 2>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
        if (h->plan.split == 3)
            return launch_mfma_nf<NT, false, 3>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
    }
    return launch_mfma_nf<NT, WIDE, 4>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
}

static int launch_rows(pcx_bary *h, const DerivedTensor &dt, const double *d_pts, long N,
                       double *d_out, long ostride, long ooff, hipStream_t st, const int *perm) {
    int ppw = 256 / h->lpp;
    size_t lds = (size_t)ppw * h->dims.sum_n * sizeof(double);
    if (lds > 160 * 1024) return fail(PCX_ERR_UNSUPPORTED, "sum of node counts %d too large for the rows kernel", h->dims.sum_n);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)k_bary_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long blocks = (N + ppw - 1) / ppw;
    if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(k_bary_rows, dim3((unsigned)blocks), dim3(256), lds, st, h->dims, h->lpp,
                       h->d_nodes, h->d_wts, dt.plain, d_pts, d_out, N, ostride, ooff, perm);
    HIP_TRY(hipGetLastError());
    return PCX_OK;
}

// Evaluate m specs (dts[0..m)) at N device-resident points; out[p*ostride + ooff + s].
// T_tab (device table of m plain tensors) or, when NULL, the single tensor dt
template <int DOUT, int NLP>
static int launch_small_t(pcx_bary *h, const DerivedTensor &dt, const double *const *T_tab, int m, const double *d_pts,
                          long N, double *d_out, long ostride, long ooff, hipStream_t st, const int *perm) {
    auto kern = k_bary_small<DOUT, NLP>;
    size_t lds = (size_t)(h->dims.sum_n - h->dims.n[DOUT]) * 64 * sizeof(double);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long blocks = (N + 63) / 64;
    if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), lds, st, h->dims, h->small_scale, h->d_snodes, h->d_nodes,
                       h->d_wts, dt.plain, T_tab, m, d_pts, d_out, N, ostride, ooff, perm);
    HIP_TRY(hipGetLastError());
    return PCX_OK;
}

template <int DOUT>
static int launch_small_d(pcx_bary *h, const DerivedTensor &dt, const double *const *T_tab, int m, const double *d_pts,
                          long N, double *d_out, long ostride, long ooff, hipStream_t st, const int *perm) {
    switch (h->small_nlp) {
#define CASE_NLP(v) case v: return launch_small_t<DOUT, v>(h, dt, T_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
    CASE_NLP(2) CASE_NLP(3) CASE_NLP(4) CASE_NLP(5) CASE_NLP(6) CASE_NLP(7) CASE_NLP(8) CASE_NLP(9) CASE_NLP(10) CASE_NLP(11)
    CASE_NLP(12) CASE_NLP(13) CASE_NLP(14) CASE_NLP(15) CASE_NLP(16) CASE_NLP(24) CASE_NLP(32) CASE_NLP(48) CASE_NLP(64)
#undef CASE_NLP
    }
    return fail(PCX_ERR_UNSUPPORTED, "lane-per-point kernel does not cover this shape");
}

static int launch_small(pcx_bary *h, const DerivedTensor &dt, const double *const *T_tab, int m, const double *d_pts,
                        long N, double *d_out, long ostride, long ooff, hipStream_t st, const int *perm) {
    switch (h->dims.d) {
    case 1: return launch_small_d<0>(h, dt, T_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
    case 2: return launch_small_d<1>(h, dt, T_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
    case 3: return launch_small_d<2>(h, dt, T_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
    case 4: return launch_small_d<3>(h, dt, T_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
    }
    return fail(PCX_ERR_UNSUPPORTED, "lane-per-point kernel does not cover this shape");
}


template <int NL>
static int launch_sq_t(pcx_bary *h, const DerivedTensor &dt, const double *const *T_tab, int m, const double *d_pts,
                       long N, double *d_out, long ostride, long ooff, hipStream_t st, const int *perm) {
    const int d = h->dims.d;
    size_t lds = 0;
    for (int k = 0; k < d - 2; ++k) lds += (size_t)h->dims.n[k] * 64 * sizeof(double);
    long blocks = (N + 63) / 64;
    if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
#define PCX_SQ_GO(LEAD)                                                                                               \
    hipLaunchKernelGGL((k_bary_sq<NL, LEAD>), dim3((unsigned)blocks), dim3(64), lds, st, h->dims, h->small_scale,    \
                       h->d_snodes, h->d_nodes, h->d_wts, dt.plain, T_tab, m, d_pts, d_out, N, ostride, ooff, perm)
    if (d == 2) PCX_SQ_GO(0);
    else if (d == 3) PCX_SQ_GO(1);
    else PCX_SQ_GO(2);
#undef PCX_SQ_GO
    HIP_TRY(hipGetLastError());
    return PCX_OK;
}

static int launch_sq(pcx_bary *h, const DerivedTensor &dt, const double *const *T_tab, int m, const double *d_pts,
                     long N, double *d_out, long ostride, long ooff, hipStream_t st, const int *perm) {
    switch (h->sq_nl) {
#define CASE_NL(v) case v: return launch_sq_t<v>(h, dt, T_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
    CASE_NL(4) CASE_NL(5) CASE_NL(6) CASE_NL(7) CASE_NL(8) CASE_NL(9) CASE_NL(10) CASE_NL(11) CASE_NL(12) CASE_NL(13)
    CASE_NL(14) CASE_NL(15) CASE_NL(16) CASE_NL(17) CASE_NL(18) CASE_NL(19) CASE_NL(20) CASE_NL(21) CASE_NL(22)
    CASE_NL(23) CASE_NL(24) CASE_NL(26) CASE_NL(28) CASE_NL(30) CASE_NL(32)
#undef CASE_NL
    }
    return fail(PCX_ERR_UNSUPPORTED, "square-trailing lane-per-point kernel does not cover this shape");
}

// ---- dim-0 group launches (BaryG0) --------------------------------------------------------------
// Slab-packs dt.plain on first use (caller holds h->mu; enqueued on h->stream and synchronised).
static int bary_pack_g0(pcx_bary *h, DerivedTensor &dt) {
    if (dt.frag_g0) return PCX_OK;
    const BaryMfmaPlan &p = h->plan;
    const int n0 = h->dims.n[0];
    const size_t cnt = (size_t)n0 * h->g0_tps * p.KS * 64;
    DevBuf frag, slot;
    int rc = frag.alloc(cnt * sizeof(double));
    if (rc) return rc;
    hipLaunchKernelGGL(k_pack_fragments_slabs, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, dt.plain,
                       frag.as<double>(), n0, p.M / n0, p.K, h->g0_tps, p.KS);
    HIP_TRY(hipGetLastError());
    if ((rc = slot.alloc(sizeof(double *)))) return rc;
    double *fp = frag.as<double>();
    HIP_TRY(hipMemcpy(slot.p, &fp, sizeof(double *), hipMemcpyHostToDevice));
    HIP_TRY(hipStreamSynchronize(h->stream));
    dt.frag_g0 = frag.release<double>();
    dt.slot_g0 = slot.release<double *>();
    return PCX_OK;
}

template <int KS, int NF>
static int launch_g0_t(pcx_bary *h, const DerivedTensor &base, const BaryG0 &gs, const double *d_pts, long N, double *d_out,
                       long ostride, long ooff, hipStream_t st) {
    auto kern = k_bary_mfma<KS, 2, false, NF, true>;
    const size_t lds = mfma_lds_bytes(h->dims, 2);
    if (lds > 64 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long blocks = (N + 127) / 128;
    if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
    BaryMfmaPlan plan = h->plan;
    plan.MT = gs.tps * gs.n0;
    const int nchunks = (plan.MT + PCX_CHUNK_TILES - 1) / PCX_CHUNK_TILES;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, 1, 1), dim3(256), lds, st, h->dims, plan, h->d_nodes, h->d_wts,
                       (const double *const *)base.slot_g0, h->d_rowcode_g0, h->d_kcode, nullptr, nullptr, d_pts, d_out, N,
                       ostride, ooff, nchunks, nullptr, nullptr, gs, h->d_diff + h->doff[0]);
    HIP_TRY(hipGetLastError());
    return PCX_OK;
}

template <int NF>
static int launch_g0_nf(pcx_bary *h, const DerivedTensor &base, const BaryG0 &gs, const double *d_pts, long N, double *d_out,
                        long ostride, long ooff, hipStream_t st) {
    switch (h->plan.KS) {
#define CASE_KS(v) case v: return launch_g0_t<v, NF>(h, base, gs, d_pts, N, d_out, ostride, ooff, st);
        CASE_KS(1) CASE_KS(2) CASE_KS(3) CASE_KS(4) CASE_KS(5) CASE_KS(6) CASE_KS(7) CASE_KS(8)
        CASE_KS(9) CASE_KS(10) CASE_KS(11) CASE_KS(12) CASE_KS(13) CASE_KS(14) CASE_KS(15) CASE_KS(16)
        CASE_KS(17) CASE_KS(18) CASE_KS(19) CASE_KS(20) CASE_KS(21) CASE_KS(22) CASE_KS(23) CASE_KS(24)
        CASE_KS(25) CASE_KS(26) CASE_KS(27) CASE_KS(28) CASE_KS(29) CASE_KS(30) CASE_KS(31) CASE_KS(32)
#undef CASE_KS
    }
    return fail(PCX_ERR_UNSUPPORTED, "no dim-0 group instantiation for KS=%d", h->plan.KS);
}

static const long kG0MinPoints = 65536;      // below: per-spec launches (they split over row tiles and need no second pass)

// the kernel a launch will take: 1 rows, 2 MFMA 16x16x4, 3 MFMA 4x4x4, 4 lane-per-point
PCX_HIDDEN int bary_effective_variant(const pcx_bary *h) {
    if (h->variant != 0) return h->variant;
    if (h->sq_nl && h->sq_preferred) return 5;
    return (h->small_nlp && (h->small_preferred || !h->mfma_ok)) ? 4 : (h->mfma_ok ? 2 : 1);
}

// frag_tab is a device table holding the m tensor pointers of a multi-spec launch: fragment images for
// the MFMA kernels, plain tensors for the lane-per-point kernel (see bary_spec_table); for m = 1 the MFMA
// kernels read dts[0]->slot and the lane-per-point kernel takes dts[0]->plain directly.
// split_scratch (nullable): where split launches of small batches keep their per-chunk sums;
// perm (nullable): evaluate rows perm[0..N) of d_pts / d_out instead of rows 0..N.
PCX_HIDDEN int bary_launch(pcx_bary *h, DerivedTensor *const *dts, int m, const double *const *frag_tab,
                       const double *d_pts, long N, double *d_out, long ostride, long ooff,
                       hipStream_t st, Scratch *split_scratch, const int *perm) {
    if (N == 0) return PCX_OK;
    const int variant = bary_effective_variant(h);
    if (variant == 4) {
        if (!h->small_nlp) return fail(PCX_ERR_UNSUPPORTED, "lane-per-point kernel does not cover this shape");
        return launch_small(h, *dts[0], m > 1 ? frag_tab : nullptr, m, d_pts, N, d_out, ostride, ooff, st, perm);
    }
    if (variant == 5) {
        if (!h->sq_nl) return fail(PCX_ERR_UNSUPPORTED, "square-trailing lane-per-point kernel does not cover this shape");
        return launch_sq(h, *dts[0], m > 1 ? frag_tab : nullptr, m, d_pts, N, d_out, ostride, ooff, st, perm);
    }
    if (variant == 3) {
        if (!h->mfma4_ok) return fail(PCX_ERR_UNSUPPORTED, "4x4x4 MFMA kernel does not cover this shape");
        return launch_mfma4(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
    }
    if (variant == 2) {
        if (!h->mfma_ok) return fail(PCX_ERR_UNSUPPORTED, "MFMA kernel does not cover this shape");
        if (h->kfold_ok) return bary_launch_kfold(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, perm);
        if (h->grid_ok) return bary_launch_grid(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
        // two column tiles per wave for throughput; one when the batch cannot fill the chip
        int nt = (N >= 65536) ? h->nt : 1;
        if (h->wide)
            return nt == 2 ? launch_mfma_nt<2, true>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm)
                           : launch_mfma_nt<1, true>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
        return nt == 2 ? launch_mfma_nt<2, false>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm)
                       : launch_mfma_nt<1, false>(h, frag_tab, m, d_pts, N, d_out, ostride, ooff, st, split_scratch, perm);
    }
    for (int s = 0; s < m; ++s) {
        int rc = launch_rows(h, *dts[s], d_pts, N, d_out, ostride, ooff + s, st, perm);
        if (rc) return rc;
    }
    return PCX_OK;
}

// The model with dimension q moved to the front (the other dimensions keep their order), or NULL when that shape has
// no slab plan.  Built on first use from the device copies of the grid arrays and the value tensor: a transposed copy of a
// tensor of at most 2^24 elements, once per handle and dimension.  Caller holds h->mu.
static const long kRotMaxElems = 1L << 24;
static pcx_bary *bary_rot(pcx_bary *h, int q) {
    if (h->rot_state[q]) return h->rot[q];
    h->rot_state[q] = 2;
    const int d = h->dims.d;
    if (q < 1 || q >= d || h->total > kRotMaxElems || h->dims.n[q] < 2 || h->dims.n[q] > 16) return nullptr;
    const long sum_n = h->dims.sum_n;
    long sum_n2 = 0;
    for (int k = 0; k < d; ++k) sum_n2 += (long)h->dims.n[k] * h->dims.n[k];
    std::vector<double> nodes((size_t)sum_n), wts((size_t)sum_n), diff((size_t)sum_n2), T((size_t)h->total), TR((size_t)h->total);
    const DerivedTensor &val = h->cache[std::vector<int>(d, 0)];
    if (hipMemcpy(nodes.data(), h->d_nodes, sum_n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(wts.data(), h->d_wts, sum_n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(diff.data(), h->d_diff, sum_n2 * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(T.data(), val.plain, h->total * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    int pd[PCX_MAX_DIMS];                       // pd[c] = original dimension at position c of the sub-model
    pd[0] = q;
    for (int k = 0, c = 1; k < d; ++k)
        if (k != q) pd[c++] = k;
    std::vector<int32_t> nn(d);
    std::vector<double> rn, rw, rd;
    long stride[PCX_MAX_DIMS];                  // element strides of the original C-order tensor
    { long acc = 1; for (int k = d - 1; k >= 0; --k) { stride[k] = acc; acc *= h->dims.n[k]; } }
    for (int c = 0; c < d; ++c) {
        const int k = pd[c], n = h->dims.n[k];
        nn[c] = n;
        rn.insert(rn.end(), nodes.begin() + h->dims.off[k], nodes.begin() + h->dims.off[k] + n);
        rw.insert(rw.end(), wts.begin() + h->dims.off[k], wts.begin() + h->dims.off[k] + n);
        rd.insert(rd.end(), diff.begin() + h->doff[k], diff.begin() + h->doff[k] + (long)n * n);
    }
    {   // TR[i_q, i_0, ..] = T[i_0, .., i_q, ..]: an odometer over the sub-model's index, the source offset kept alongside
        int idx[PCX_MAX_DIMS] = {};
        long src = 0;
        for (long e = 0; e < h->total; ++e) {
            TR[(size_t)e] = T[(size_t)src];
            for (int c = d - 1; c >= 0; --c) {
                src += stride[pd[c]];
                if (++idx[c] < nn[c]) break;
                src -= stride[pd[c]] * nn[c];
                idx[c] = 0;
            }
        }
    }
    pcx_bary *r = nullptr;
    if (pcx_bary_create(h->device, d, nn.data(), rn.data(), rw.data(), rd.data(), TR.data(), &r) != PCX_OK || !r) return nullptr;
    if (!r->g0_ok) { pcx_bary_destroy(r); return nullptr; }
    h->rot[q] = r;
    h->rot_state[q] = 1;
    return r;
}

// One slab launch: the specs `lower` + rel[i] e_q (lower: a spec in h's dimension order whose order along q is the
// group's base) into columns col[i].  pp: the batch in the column order of the model that runs it (h for q = 0, else
// h->rot[q]).  Caller holds h->mu.
static int bary_launch_group(pcx_bary *h, int q, const int32_t *lower, const int *rel, const int *col, int nmem,
                             const double *pp, long N, double *d_out, long ostride, long ooff, hipStream_t st) {
    const int d = h->dims.d;
    pcx_bary *g = q == 0 ? h : h->rot[q];
    std::vector<int32_t> bspec(d);
    if (q == 0) bspec.assign(lower, lower + d);
    else { bspec[0] = lower[q]; for (int k = 0, c = 1; k < d; ++k) if (k != q) bspec[c++] = lower[k]; }
    if (g != h) g->call_mark = g->clock;
    DerivedTensor *base = nullptr;
    int rc = bary_get_tensor(g, bspec.data(), &base);
    if (rc) return rc;
    if ((rc = bary_pack_g0(g, *base))) return rc;
    BaryG0 gs{};
    gs.nmem = nmem;
    gs.tps = g->g0_tps;
    gs.n0 = g->dims.n[0];
    for (int i = 0; i < nmem; ++i) {
        gs.order[i] = rel[i];
        gs.col[i] = col[i];
        gs.maxorder = std::max(gs.maxorder, rel[i]);
    }
    return (g->g0_nf == 2) ? launch_g0_nf<2>(g, *base, gs, pp, N, d_out, ostride, ooff, st)
                           : launch_g0_nf<3>(g, *base, gs, pp, N, d_out, ostride, ooff, st);
}

// How far does the spec lower + e_q come out of lower's GEMM from where its own GEMM puts it?  Differentiating after
// the contraction rounds differently from the reference's batch path, by an amount that depends on the data and that
// no cheap bound predicts (5-D Black-Scholes: delta out of the price tensor 1e-13, vega 7e-13, rho 1e-12, vanna out of
// the delta tensor along sigma 6e-12).  So it is MEASURED, once per handle and (lower, q): a probe batch -- a quarter
// interior points, a quarter domain corners, half mixtures of lo / hi / interior coordinates: the roundings are
// largest where the weights are -- goes through the slab launch and through the spec's own GEMM; returned is
// max |shared - own| / max |own| over it (infinity when the pair cannot run).  Caller holds h->mu; q's model exists.
static const int kProbePoints = 2048;
static double bary_pair_deviation(pcx_bary *h, const std::vector<int> &lower, int q) {
    std::vector<int> key = lower;
    key.push_back(q);
    auto it = h->pair_dev.find(key);
    if (it != h->pair_dev.end()) return it->second;
    // a probe that could not RUN (allocation, copy or launch failed) is not remembered: the pair stays ungrouped for this
    // call and is measured again by the next one; PCX_BARY_PROBE_LOG=1 reports it
    struct Forget {
        pcx_bary *h; const std::vector<int> &key; bool keep = false;
        ~Forget() {
            if (keep) return;
            h->pair_dev.erase(key);
            (void)hipGetLastError();
            if (getenv("PCX_BARY_PROBE_LOG")) fprintf(stderr, "[pcx] pair probe could not run: not cached, retried by the next call\n");
        }
    } forget{h, key};
    double &dev = h->pair_dev[key];
    dev = INFINITY;
    const int d = h->dims.d;
    // the domain from the outer nodes (Chebyshev points of the first kind: x_0 = mid - half cos(pi / 2n))
    std::vector<double> nodes((size_t)h->dims.sum_n);
    if (hipMemcpy(nodes.data(), h->d_nodes, nodes.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return dev;
    std::vector<double> lo(d), hi(d);
    for (int k = 0; k < d; ++k) {
        const int n = h->dims.n[k];
        const double a = nodes[h->dims.off[k]], b = nodes[h->dims.off[k] + n - 1];
        const double half = n > 1 ? 0.5 * (b - a) / std::cos(3.14159265358979323846 / (2.0 * n)) : 0.0;
        lo[k] = 0.5 * (a + b) - half;
        hi[k] = 0.5 * (a + b) + half;
    }
    const long N = kProbePoints;
    std::vector<double> P((size_t)N * d);
    uint64_t state = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { state = state * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(state >> 33); };
    for (long p = 0; p < N; ++p) {
        const int mode = (int)(p % 4);              // 0 interior, 1 corner, 2 / 3 a mix of lo, hi and interior coordinates
        for (int k = 0; k < d; ++k) {
            const uint32_t r = rnd();
            const double uni = lo[k] + (hi[k] - lo[k]) * ((double)(r >> 8) / 8388608.0);
            const int pick = mode == 0 ? 2 : (mode == 1 ? (int)(r & 1) : (int)(r % 3));
            P[(size_t)p * d + k] = pick == 0 ? lo[k] : (pick == 1 ? hi[k] : uni);
        }
    }
    DevBuf dp, dr, dout;
    if (dp.alloc(P.size() * sizeof(double)) || dr.alloc(P.size() * sizeof(double)) || dout.alloc((size_t)N * 2 * sizeof(double))) return dev;
    if (hipMemcpy(dp.p, P.data(), P.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return dev;
    const double *gp = dp.as<double>();
    if (q > 0) {
        SliderCols cols{};
        cols.nc = d;
        cols.col[0] = q;
        for (int k = 0, c = 1; k < d; ++k)
            if (k != q) cols.col[c++] = k;
        hipLaunchKernelGGL(k_gather_columns, dim3((unsigned)((N * d + 255) / 256)), dim3(256), 0, h->stream, dp.as<double>(), N, d,
                           cols, dr.as<double>());
        gp = dr.as<double>();
    }
    std::vector<int32_t> lspec(lower.begin(), lower.end()), uspec(lower.begin(), lower.end());
    ++uspec[q];
    const int rel = 1, col = 0;
    if (bary_launch_group(h, q, lspec.data(), &rel, &col, 1, gp, N, dout.as<double>(), 2, 0, h->stream)) return dev;
    DerivedTensor *own = nullptr;
    if (bary_get_tensor(h, uspec.data(), &own)) return dev;
    DerivedTensor *one[1] = {own};
    if (bary_launch(h, one, 1, own->slot, dp.as<double>(), N, dout.as<double>(), 2, 1, h->stream, &h->s_partial)) return dev;
    std::vector<double> R((size_t)N * 2);
    if (hipStreamSynchronize(h->stream) != hipSuccess ||
        hipMemcpy(R.data(), dout.p, R.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return dev;
    }
    double scale = 0.0, diff = 0.0;
    for (long p = 0; p < N; ++p) {
        scale = std::max(scale, std::fabs(R[2 * p + 1]));
        diff = std::max(diff, std::fabs(R[2 * p] - R[2 * p + 1]));
    }
    if (std::isfinite(diff) && std::isfinite(scale)) dev = scale > 0.0 ? diff / scale : (diff == 0.0 ? 0.0 : INFINITY);
    forget.keep = true;                           // measured (an infinite deviation included: non-finite model values)
    static const bool log = getenv("PCX_BARY_PROBE_LOG") != nullptr;
    if (log) {
        fprintf(stderr, "[pcx] pair probe (");
        for (int k = 0; k < d; ++k) fprintf(stderr, "%d%s", lower[k], k + 1 < d ? "," : "");
        fprintf(stderr, ") + e_%d out of one GEMM: %.3g of the scale from its own GEMM (tolerance %.3g)\n", q, dev, h->group_tol);
    }
    return dev;
}

// a group = the specs one slab GEMM serves: equal orders off dimension q, orders along q in [base, base + span]
struct BaryGroup { int q; int base; std::vector<int> members; };

// Which specs of a multi-spec launch share a GEMM (caller holds h->mu; may build sub-models and run probes).
//  * span >= 2 (opt-in, not probed): specs with equal orders along dimensions 1 .. d-1 and dim-0 orders within
//    [base, base + span].
//  * then PAIRS: a spec and the spec one order below it along any one dimension q (delta / gamma from the delta
//    tensor's GEMM, price / vega along the volatility axis, ...), found greedily from the highest total order down,
//    dimension 0 first; q > 0 runs on the sub-model with q in front (bary_rot), own streams only (its column-permuted
//    batch lives in the handle).  A pair is formed only when the probe has MEASURED the derived member within
//    h->group_tol of its own GEMM (bary_pair_deviation).
// N = the size of the CALL's batch: every piece of a pipelined host batch and every block of a fan-out is planned with it,
// so that a (point, spec) gets the same rounding whichever piece it lands in.
static void bary_plan_groups(pcx_bary *h, const int32_t *derivs, int m, long N, bool own_stream, std::vector<BaryGroup> &subs,
                             std::vector<char> &grouped) {
    const int d = h->dims.d;
    grouped.assign(m, 0);
    const int span = h->g0_span;
    if (!(span > 0 && derivs && m > 1 && N >= kG0MinPoints && bary_effective_variant(h) == 2)) return;
    auto spec = [&](int s) { return std::vector<int>(derivs + (size_t)s * d, derivs + (size_t)(s + 1) * d); };
    if (span >= 2 && h->g0_ok) {
        std::map<std::vector<int>, std::vector<int>> by_key;       // orders[1:] -> specs, in column order
        for (int s = 0; s < m; ++s)
            by_key[std::vector<int>(derivs + (size_t)s * d + 1, derivs + (size_t)(s + 1) * d)].push_back(s);
        for (auto &kv : by_key) {
            std::vector<int> &mem = kv.second;
            if (mem.size() < 2) continue;
            std::sort(mem.begin(), mem.end(), [&](int a, int b) {
                const int oa = derivs[(size_t)a * d], ob = derivs[(size_t)b * d];
                return oa != ob ? oa < ob : a < b;
            });
            for (size_t i = 0; i < mem.size();) {
                const int base = derivs[(size_t)mem[i] * d];
                size_t e = i;
                while (e < mem.size() && derivs[(size_t)mem[e] * d] <= base + span && e - i < PCX_G0_MAX) ++e;
                if (e - i >= 2) {
                    subs.push_back(BaryGroup{0, base, std::vector<int>(mem.begin() + i, mem.begin() + e)});
                    for (size_t q = i; q < e; ++q) grouped[mem[q]] = 1;
                }
                i = e;
            }
        }
    }
    std::map<std::vector<int>, int> first;                          // orders -> first column still on its own
    std::vector<int> by_order;
    for (int s = 0; s < m; ++s)
        if (!grouped[s] && first.emplace(spec(s), s).second) by_order.push_back(s);
    auto total_order = [&](int s) { int t = 0; for (int k = 0; k < d; ++k) t += derivs[(size_t)s * d + k]; return t; };
    std::stable_sort(by_order.begin(), by_order.end(), [&](int a, int b) { return total_order(a) > total_order(b); });
    for (int b : by_order) {
        if (grouped[b]) continue;
        std::vector<int> lower = spec(b);
        for (int q = 0; q < d; ++q) {
            if (lower[q] < 1) continue;
            --lower[q];
            auto it = first.find(lower);
            ++lower[q];
            if (it == first.end() || grouped[it->second]) continue;
            if (q == 0 ? !h->g0_ok : !(own_stream && bary_rot(h, q))) continue;
            --lower[q];
            const double dev = bary_pair_deviation(h, lower, q);
            ++lower[q];
            if (!(dev <= h->group_tol)) continue;
            subs.push_back(BaryGroup{q, lower[q] - 1, {it->second, b}});
            grouped[it->second] = grouped[b] = 1;
            break;
        }
    }
}

// Multi-spec launch with shared contractions: specs (rows of `derivs`, m x d) one order apart along one dimension --
// price / delta, delta / gamma, price / vega -- share one slab-packed GEMM over the tensor of the lower order (the
// reference's own order in vectorized_eval_multi, barycentric.py:1098-1110: contract the other dimensions, then apply
// D_q); every other spec keeps its own GEMM, launched in runs of consecutive columns.  Large batches on the MFMA
// kernel only; results of grouped specs differ from the per-spec path by rounding (<= 2e-13 of the scale on 5-D
// Black-Scholes), as the reference's multi and batch paths do.  Caller holds h->mu.
static int bary_launch_specs(pcx_bary *h, const int32_t *derivs, DerivedTensor *const *dts, int m,
                             const double *const *frag_tab, const double *d_pts, long N, double *d_out, long ostride,
                             long ooff, hipStream_t st, Scratch *split_scratch, long N_call = -1) {
    const int d = h->dims.d;
    std::vector<BaryGroup> subs;
    std::vector<char> grouped;
    const bool own_stream = st == h->stream || (h->stream2 && st == h->stream2);
    bary_plan_groups(h, derivs, m, N_call > 0 ? N_call : N, own_stream, subs, grouped);
    // runs of consecutive ungrouped specs: ordinary launches
    for (int s = 0; s < m;) {
        if (grouped[s]) { ++s; continue; }
        int e = s;
        while (e < m && !grouped[e]) ++e;
        int rc = bary_launch(h, dts + s, e - s, (e - s == 1) ? (const double *const *)dts[s]->slot : frag_tab + s, d_pts, N,
                             d_out, ostride, ooff + s, st, split_scratch);
        if (rc) return rc;
        s = e;
    }
    // the batch in the column order of every sub-model this call uses: one gather per dimension
    const double *rpts[PCX_MAX_DIMS] = {};
    {
        int nq = 0;
        for (const BaryGroup &sub : subs)
            if (sub.q > 0 && !rpts[sub.q]) { rpts[sub.q] = d_pts; ++nq; }
        if (nq) {
            Scratch &sc = (h->stream2 && st == h->stream2) ? h->s_rot2 : h->s_rot;
            int rc = sc.reserve((size_t)nq * N * d * sizeof(double));
            if (rc) return rc;
            double *dst = (double *)sc.ptr;
            for (int q = 1; q < d; ++q) {
                if (!rpts[q]) continue;
                SliderCols cols{};
                cols.nc = d;
                cols.col[0] = q;
                for (int k = 0, c = 1; k < d; ++k)
                    if (k != q) cols.col[c++] = k;
                const long elems = N * d;
                hipLaunchKernelGGL(k_gather_columns, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, d_pts, N, d, cols, dst);
                HIP_TRY(hipGetLastError());
                rpts[q] = dst;
                dst += (size_t)N * d;
            }
        }
    }
    for (const BaryGroup &sub : subs) {
        std::vector<int32_t> lower(derivs + (size_t)sub.members[0] * d, derivs + (size_t)(sub.members[0] + 1) * d);
        lower[sub.q] = sub.base;
        int rel[PCX_G0_MAX], col[PCX_G0_MAX];
        const int nmem = (int)sub.members.size();
        for (int i = 0; i < nmem; ++i) {
            rel[i] = derivs[(size_t)sub.members[i] * d + sub.q] - sub.base;
            col[i] = sub.members[i];
        }
        int rc = bary_launch_group(h, sub.q, lower.data(), rel, col, nmem, sub.q == 0 ? d_pts : rpts[sub.q], N, d_out, ostride,
                                   ooff, st);
        if (rc) return rc;
    }
    return PCX_OK;
}

// GEMM launches a multi-spec call of N points would execute (groups count once); builds what the call would build.
extern "C" int pcx_bary_count_gemms(pcx_bary *h, const int32_t *derivs, int m, int64_t N, int32_t *gemms_out) {
    PCX_API_BEGIN
    if (!h || !derivs || !gemms_out || m < 1) return fail(PCX_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    std::vector<BaryGroup> subs;
    std::vector<char> grouped;
    bary_plan_groups(h, derivs, m, (long)N, true, subs, grouped);
    int count = (int)subs.size();
    for (int s = 0; s < m; ++s) count += !grouped[s];
    *gemms_out = count;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_bary_eval_batch_dev(pcx_bary *h, const double *d_pts, int64_t N,
                                       const int32_t *deriv, double *d_out, void *stream) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0) return fail(PCX_ERR_INVALID, "N=%lld < 0", (long long)N);
    if (N > 0 && (!d_pts || !d_out)) return fail(PCX_ERR_INVALID, "NULL device buffer");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    h->call_mark = h->clock;
    DerivedTensor *dt = nullptr;
    int rc = bary_get_tensor(h, deriv, &dt);
    if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    // split launches share the handle's scratch: only on the handle's own stream
    return bary_launch(h, &dt, 1, dt->slot, d_pts, (long)N, d_out, 1, 0, st,
                       st == h->stream ? &h->s_partial : nullptr);
    PCX_API_END
}

// m specs at N device-resident points into d_out (N x m row-major); groups of kMaxSpecs specs per launch.
extern "C" int pcx_bary_eval_multi_batch_dev(pcx_bary *h, const double *d_pts, int64_t N, const int32_t *derivs,
                                             int m, double *d_out, void *stream) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0 || m < 1) return fail(PCX_ERR_INVALID, "bad N or m");
    if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
    if (N == 0) return PCX_OK;
    if (!d_pts || !d_out) return fail(PCX_ERR_INVALID, "NULL device buffer");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    h->call_mark = h->clock;
    const int d = h->dims.d;
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    for (int s0 = 0; s0 < m; s0 += kMaxSpecs) {
        const int mc = std::min(kMaxSpecs, m - s0);
        std::vector<DerivedTensor *> dts(mc);
        for (int s = 0; s < mc; ++s) {
            int rc = bary_get_tensor(h, derivs + (size_t)(s0 + s) * d, &dts[s]);
            if (rc) return rc;
        }
        const double *const *frag_tab = dts[0]->slot;
        const int eff = bary_effective_variant(h);
        if (mc > 1 && (eff == 4 || eff == 5 || h->mfma_ok)) {
            std::vector<double *> tab(mc);
            for (int s = 0; s < mc; ++s) tab[s] = (eff == 4 || eff == 5) ? dts[s]->plain : dts[s]->frag;
            if (tab != h->tab_host) {
                HIP_TRY(hipDeviceSynchronize());        // launches still in flight on any stream may read d_tab
                HIP_TRY(hipMemcpy(h->d_tab, tab.data(), mc * sizeof(double *), hipMemcpyHostToDevice));
                h->tab_host = tab;
            }
            frag_tab = h->d_tab;
        }
        int rc = bary_launch_specs(h, derivs + (size_t)s0 * d, dts.data(), mc, frag_tab, d_pts, (long)N, d_out, m, s0, st,
                                   st == h->stream ? &h->s_partial : nullptr);
        if (rc) return rc;
    }
    return PCX_OK;
    PCX_API_END
}

// N_call: the batch of the API call this block belongs to (fan-out blocks pass the whole call's N), -1 = N itself
static int bary_eval_host(pcx_bary *h, const double *pts, int64_t N, const int32_t *derivs, int m,
                          double *out, int64_t N_call = -1) {
    if (N_call < N) N_call = N;
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0 || m < 1) return fail(PCX_ERR_INVALID, "bad N or m");
    if (N > 0 && (!pts || !out)) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (m > kMaxSpecs) {
        // more specs than one launch takes (the reference has no limit: a gradient plus full
        // Hessian in 10-D is 65): groups of kMaxSpecs, each into its columns of `out`
        if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
        const int d0 = h->dims.d;
        std::vector<double> part;
        for (int s0 = 0; s0 < m; s0 += kMaxSpecs) {
            const int mc = std::min(kMaxSpecs, m - s0);
            part.resize((size_t)N * mc);
            int rc = bary_eval_host(h, pts, N, derivs + (size_t)s0 * d0, mc, part.data(), N_call);
            if (rc) return rc;
            for (int64_t i = 0; i < N; ++i)
                memcpy(out + (size_t)i * m + s0, part.data() + (size_t)i * mc, (size_t)mc * sizeof(double));
        }
        return PCX_OK;
    }
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    h->call_mark = h->clock;
    const int d = h->dims.d;
    std::vector<DerivedTensor *> dts(m);
    for (int s = 0; s < m; ++s) {
        int rc = bary_get_tensor(h, derivs ? derivs + (size_t)s * d : nullptr, &dts[s]);
        if (rc) return rc;
    }
    const double *const *frag_tab = dts[0]->slot;
    const int eff = bary_effective_variant(h);
    if (m > 1 && (eff == 4 || eff == 5 || h->mfma_ok)) {
        std::vector<double *> tab(m);
        for (int s = 0; s < m; ++s) tab[s] = (eff == 4 || eff == 5) ? dts[s]->plain : dts[s]->frag;
        if (tab != h->tab_host) {   // every earlier launch on this handle has been synchronised
            HIP_TRY(hipMemcpy(h->d_tab, tab.data(), m * sizeof(double *), hipMemcpyHostToDevice));
            h->tab_host = tab;
        }
        frag_tab = h->d_tab;
    }
    if (N > 0 && (size_t)N * d * sizeof(double) <= kPinnedBytes && (size_t)N * m * sizeof(double) <= kPinnedBytes &&
        h->pin.ready()) {
        memcpy(h->pin.in, pts, (size_t)N * d * sizeof(double));
        int rc = bary_launch(h, dts.data(), m, frag_tab, (const double *)h->pin.in, (long)N, (double *)h->pin.out, m, 0,
                             h->stream, &h->s_partial);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(out, h->pin.out, (size_t)N * m * sizeof(double));
        return PCX_OK;
    }
    // Two staging slots on two streams: the H2D copy of chunk i+1 and the D2H copy of chunk i-1
    // overlap the kernel of chunk i.  A slot is reused only after its stream has drained.
    // pieces of 2^18 points (10 MB of 5-D coordinates); low-dimensional models take more points per piece so that a
    // piece still moves ~10 MB (12 x 12 at 2x10^7 points: 4 MB pieces ran the path at 13 GB/s)
    const int64_t piece = std::min<int64_t>((int64_t)1 << 21, std::max<int64_t>(kPipeChunkPoints, (((int64_t)10 << 20) / (d * 8)) & ~(int64_t)65535));
    const bool piped = N >= 2 * piece;
    const int64_t chunk = piped ? piece : kChunkPoints;
    if (piped && !h->stream2)
        HIP_TRY(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    // The copies queued below read and write the CALLER's arrays: whatever happens, the helper thread is joined and both
    // streams are drained before this call returns.
    Downloader dl(h->device);
    auto pipeline = [&]() -> int {
        int slot = 0;
        // a short first piece (one round of workgroups) so that the first kernel starts after 2.6 MB instead of
        // 10 MB of upload: nothing overlaps the first upload
        const int64_t first_piece = piped ? (1 << 16) : chunk;
        long piece_no = 0;
        for (int64_t start = 0, step = first_piece; start < N; start += step, step = chunk, ++piece_no) {
            long cnt = (long)std::min<int64_t>(step, N - start);
            const bool second = piped && slot == 1;
            hipStream_t st = second ? h->stream2 : h->stream;
            Scratch &sp = second ? h->s_pts2 : h->s_pts, &so = second ? h->s_out2 : h->s_out;
            if (piped && piece_no >= 2) dl.wait_issued(piece_no - 1);     // this slot's last download is behind its kernel
            int rc = sp.reserve((size_t)cnt * d * sizeof(double));
            if (rc) return rc;
            rc = so.reserve((size_t)cnt * m * sizeof(double));
            if (rc) return rc;
            double *dp = (double *)sp.ptr, *dout = (double *)so.ptr;
            HIP_TRY(hipMemcpyAsync(dp, pts + (size_t)start * d, (size_t)cnt * d * sizeof(double), hipMemcpyHostToDevice, st));
            rc = bary_launch_specs(h, derivs, dts.data(), m, frag_tab, dp, cnt, dout, m, 0, st, second ? nullptr : &h->s_partial,
                                   (long)N_call);
            if (rc) return rc;
            if (!piped) {                             // single slot: download here, drain before its buffers are reused
                HIP_TRY(hipMemcpyAsync(out + (size_t)start * m, dout, (size_t)cnt * m * sizeof(double), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                continue;
            }
            dl.push(out + (size_t)start * m, dout, (size_t)cnt * m * sizeof(double), st);
            slot ^= 1;
        }
        return PCX_OK;
    };
    const int rc_pipe = pipeline();
    const int rc_dl = dl.finish();
    const hipError_t e1 = hipStreamSynchronize(h->stream);
    const hipError_t e2 = h->stream2 ? hipStreamSynchronize(h->stream2) : hipSuccess;
    if (rc_pipe) return rc_pipe;
    if (rc_dl) return rc_dl;
    HIP_TRY(e1);
    HIP_TRY(e2);
    return PCX_OK;
}

extern "C" int pcx_bary_eval_batch(pcx_bary *h, const double *pts, int64_t N, const int32_t *deriv,
                                   double *out) {
    PCX_API_BEGIN
    return bary_eval_host(h, pts, N, deriv, 1, out);
    PCX_API_END
}

extern "C" int pcx_bary_eval_multi_batch(pcx_bary *h, const double *pts, int64_t N,
                                         const int32_t *derivs, int m, double *out) {
    PCX_API_BEGIN
    if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
    return bary_eval_host(h, pts, N, derivs, m, out);
    PCX_API_END
}

// ---- all equal-shape pieces of a piecewise interpolant in one launch (called from pcx_spline.hip) ---------------
template <int DOUT, int NLP>
static void launch_small_pieces_t(const pcx_bary *p0, const SplinePieceModel *models, const int *blk_piece,
                                  const int *blk_first, const int *piece_end, int m, long blocks, const double *dp,
                                  double *dout, const int *perm, hipStream_t st) {
    auto kern = k_bary_small_pieces<DOUT, NLP>;
    const size_t lds = (size_t)(p0->dims.sum_n - p0->dims.n[DOUT]) * 64 * sizeof(double);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), lds, st, p0->dims, models, blk_piece, blk_first, piece_end, m,
                       dp, dout, (long)m, 0L, perm);
}

template <int DOUT>
static int launch_small_pieces_d(const pcx_bary *p0, const SplinePieceModel *models, const int *blk_piece,
                                 const int *blk_first, const int *piece_end, int m, long blocks, const double *dp,
                                 double *dout, const int *perm, hipStream_t st) {
    switch (p0->small_nlp) {
#define CASE_NLP(v) case v: launch_small_pieces_t<DOUT, v>(p0, models, blk_piece, blk_first, piece_end, m, blocks, dp, dout, perm, st); return PCX_OK;
    CASE_NLP(2) CASE_NLP(3) CASE_NLP(4) CASE_NLP(5) CASE_NLP(6) CASE_NLP(7) CASE_NLP(8) CASE_NLP(9) CASE_NLP(10) CASE_NLP(11)
    CASE_NLP(12) CASE_NLP(13) CASE_NLP(14) CASE_NLP(15) CASE_NLP(16) CASE_NLP(24) CASE_NLP(32) CASE_NLP(48) CASE_NLP(64)
#undef CASE_NLP
    }
    return fail(PCX_ERR_UNSUPPORTED, "lane-per-point kernel does not cover this shape");
}

// All non-empty pieces in one launch.  Returns PCX_OK with *done = false when the batch does not qualify
// (a piece forced onto another kernel form): the caller then launches per piece.
template <int NL>
static int launch_sq_pieces_t(const pcx_bary *p0, const SplinePieceModel *models, const int *blk_piece, const int *blk_first,
                              const int *piece_end, int m, long blocks, const double *dp, double *dout, const int *perm,
                              hipStream_t st) {
    const int d = p0->dims.d;
    size_t lds = 0;
    for (int k = 0; k < d - 2; ++k) lds += (size_t)p0->dims.n[k] * 64 * sizeof(double);
#define PCX_SQP_GO(LEAD)                                                                                              \
    hipLaunchKernelGGL((k_bary_sq_pieces<NL, LEAD>), dim3((unsigned)blocks), dim3(64), lds, st, p0->dims, models, blk_piece, \
                       blk_first, piece_end, m, dp, dout, (long)m, 0L, perm)
    if (d == 2) PCX_SQP_GO(0);
    else if (d == 3) PCX_SQP_GO(1);
    else PCX_SQP_GO(2);
#undef PCX_SQP_GO
    return PCX_OK;
}

PCX_HIDDEN int bary_launch_sq_pieces(const pcx_bary *p0, const SplinePieceModel *models, const int *blk_piece, const int *blk_first,
                            const int *piece_end, int m, long blocks, const double *dp, double *dout, const int *perm,
                            hipStream_t st) {
    switch (p0->sq_nl) {
#define CASE_NL(v) case v: return launch_sq_pieces_t<v>(p0, models, blk_piece, blk_first, piece_end, m, blocks, dp, dout, perm, st);
    CASE_NL(4) CASE_NL(5) CASE_NL(6) CASE_NL(7) CASE_NL(8) CASE_NL(9) CASE_NL(10) CASE_NL(11) CASE_NL(12) CASE_NL(13)
    CASE_NL(14) CASE_NL(15) CASE_NL(16) CASE_NL(17) CASE_NL(18) CASE_NL(19) CASE_NL(20) CASE_NL(21) CASE_NL(22)
    CASE_NL(23) CASE_NL(24) CASE_NL(26) CASE_NL(28) CASE_NL(30) CASE_NL(32)
#undef CASE_NL
    }
    return fail(PCX_ERR_UNSUPPORTED, "square-trailing lane-per-point kernel does not cover this piece shape");
}
PCX_HIDDEN int bary_launch_small_pieces(int dout, const pcx_bary *p0, const SplinePieceModel *models, const int *blk_piece,
                                        const int *blk_first, const int *piece_end, int m, long blocks, const double *dp,
                                        double *dout_buf, const int *perm, hipStream_t st) {
    switch (dout) {
    case 0: return launch_small_pieces_d<0>(p0, models, blk_piece, blk_first, piece_end, m, blocks, dp, dout_buf, perm, st);
    case 1: return launch_small_pieces_d<1>(p0, models, blk_piece, blk_first, piece_end, m, blocks, dp, dout_buf, perm, st);
    case 2: return launch_small_pieces_d<2>(p0, models, blk_piece, blk_first, piece_end, m, blocks, dp, dout_buf, perm, st);
    case 3: return launch_small_pieces_d<3>(p0, models, blk_piece, blk_first, piece_end, m, blocks, dp, dout_buf, perm, st);
    }
    return fail(PCX_ERR_UNSUPPORTED, "lane-per-point kernel covers d <= 4");
}

// ---------------------------------------------------------------------------------
// Single-process fan-out over several devices (contiguous row blocks, model replicated, "G parallel
// D2H copies straight into the host result", no collective).  handles[g] is the same model on device g (the same
// device may appear twice: two handles then pipeline on it); block g = rows [g ceil(N/G), min(N, (g+1) ceil(N/G)))
// is evaluated by the ordinary host-pointer path of handle g on its own host thread, its download landing in the
// caller's `out` slice.  pin != 0 page-locks the caller's arrays for the duration of the call (hipHostRegister,
// portable): the copies then run asynchronously at PCIe rate instead of through the driver's pageable staging.
// A point's result does not depend on the block it lands in (for grouped multi-spec launches: as long as every
// block stays above the 65,536-point threshold of that path).
// ---------------------------------------------------------------------------------
extern "C" int pcx_bary_group_eval_multi_batch(pcx_bary *const *handles, int n_handles, const double *pts, int64_t N,
                                               const int32_t *derivs, int m, double *out, int pin) {
    PCX_API_BEGIN
    if (!handles || n_handles < 1) return fail(PCX_ERR_INVALID, "no handles");
    for (int g = 0; g < n_handles; ++g) {
        if (!handles[g]) return fail(PCX_ERR_INVALID, "handle %d is NULL", g);
        if (handles[g]->dims.d != handles[0]->dims.d || handles[g]->total != handles[0]->total)
            return fail(PCX_ERR_INVALID, "handle %d holds a different model", g);
    }
    if (N < 0 || m < 1) return fail(PCX_ERR_INVALID, "bad N or m");
    if (N > 0 && (!pts || !out)) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (n_handles == 1 || N == 0) return bary_eval_host(handles[0], pts, N, derivs, m, out);
    const int d = handles[0]->dims.d;
    HostPin hp;
    HIP_TRY(hipSetDevice(handles[0]->device));
    if (!fanout_arrays_locked(hp, pin, pts, (size_t)N * d * sizeof(double), out, (size_t)N * m * sizeof(double)))
        return bary_eval_host(handles[0], pts, N, derivs, m, out);
    return fan_out(n_handles, N, [&](int g, int64_t lo, int64_t cnt) {
        return bary_eval_host(handles[g], pts + (size_t)lo * d, cnt, derivs, m, out + (size_t)lo * m, N);
    });
    PCX_API_END
}

extern "C" int pcx_bary_derivative_tensor(pcx_bary *h, const int32_t *deriv, double *tensor_out) {
    PCX_API_BEGIN
    if (!h || !tensor_out) return fail(PCX_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    h->call_mark = h->clock;
    DerivedTensor *dt = nullptr;
    int rc = bary_get_tensor(h, deriv, &dt);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(tensor_out, dt->plain, h->total * sizeof(double), hipMemcpyDeviceToHost));
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_tensor_contract_axis(int device, int d, const int32_t *n_nodes, const double *tensor,
                                        int axis, const double *vec, double *out) {
    PCX_API_BEGIN
    if (d < 1 || d > PCX_MAX_DIMS || !n_nodes || !tensor || !vec || !out) return fail(PCX_ERR_INVALID, "bad argument");
    if (axis < 0 || axis >= d) return fail(PCX_ERR_INVALID, "axis %d outside [0, %d)", axis, d);
    long outer = 1, inner = 1;
    for (int k = 0; k < d; ++k) {
        if (n_nodes[k] < 1) return fail(PCX_ERR_INVALID, "n_nodes[%d] < 1", k);
        if (k < axis) outer *= n_nodes[k];
        if (k > axis) inner *= n_nodes[k];
    }
    const int na = n_nodes[axis];
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf din, dvec, dout;
    HIP_TRY(hipMalloc(&din.p, (size_t)outer * na * inner * sizeof(double)));
    HIP_TRY(hipMalloc(&dvec.p, (size_t)na * sizeof(double)));
    HIP_TRY(hipMalloc(&dout.p, (size_t)outer * inner * sizeof(double)));
    HIP_TRY(hipMemcpy(din.p, tensor, (size_t)outer * na * inner * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dvec.p, vec, (size_t)na * sizeof(double), hipMemcpyHostToDevice));
    long cnt = outer * inner;
    hipLaunchKernelGGL(k_contract_axis, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, (const double *)din.p,
                       (double *)dout.p, (const double *)dvec.p, outer, na, inner);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost));
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_bary_set_kernel(pcx_bary *h, int variant) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (variant < 0 || variant > 5) return fail(PCX_ERR_INVALID, "variant %d outside [0, 5]", variant);
    if (variant == 4 && !h->small_nlp) return fail(PCX_ERR_UNSUPPORTED, "lane-per-point kernel does not cover this shape");
    if (variant == 5 && !h->sq_nl) return fail(PCX_ERR_UNSUPPORTED, "square-trailing lane-per-point kernel does not cover this shape");
    if (variant == 2 && !h->mfma_ok) return fail(PCX_ERR_UNSUPPORTED, "MFMA kernel does not cover this shape");
    if (variant == 3 && !h->mfma4_ok) return fail(PCX_ERR_UNSUPPORTED, "4x4x4 MFMA kernel does not cover this shape");
    std::lock_guard<std::mutex> lk(h->mu);
    h->variant = variant;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_bary_set_group_span(pcx_bary *h, int span) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (span < 0 || span > 8) return fail(PCX_ERR_INVALID, "span %d outside [0, 8]", span);
    std::lock_guard<std::mutex> lk(h->mu);
    h->g0_span = span;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_bary_set_group_tolerance(pcx_bary *h, double tol) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (!(tol >= 0.0)) return fail(PCX_ERR_INVALID, "tolerance must be >= 0");
    std::lock_guard<std::mutex> lk(h->mu);
    h->group_tol = tol;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_bary_kernel_info(pcx_bary *h, int32_t *info) {
    PCX_API_BEGIN
    if (!h || !info) return fail(PCX_ERR_INVALID, "NULL argument");
    { const int keep = h->variant; h->variant = 0; info[0] = bary_effective_variant(h); h->variant = keep; }
    info[1] = h->mfma_ok ? (h->kfold_ok ? h->kf.MT : (h->grid_ok ? h->gp.MT : h->plan.MT)) : 0;
    info[2] = h->mfma_ok ? (h->kfold_ok ? h->kf.nbody * h->kf.P : h->plan.KS) : 0;
    info[3] = h->mfma_ok ? (int32_t)(h->kfold_ok ? bary_kfold_lds_bytes(h->kf, h->nt)
                                                  : (h->grid_ok ? bary_grid_lds_bytes(h, h->nt) : mfma_lds_bytes(h->dims, h->nt)))
                         : (256 / h->lpp) * h->dims.sum_n * 8;
    info[4] = h->mfma_ok ? 64 * h->nt : 256 / h->lpp;
    info[5] = h->mfma_ok ? h->plan.split : h->dims.d - 1;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_bary_grid_info(pcx_bary *h, int32_t *info) {
    PCX_API_BEGIN
    if (!h || !info) return fail(PCX_ERR_INVALID, "NULL argument");
    info[0] = h->grid_ok ? 1 : (h->kfold_ok ? 2 : 0);
    info[1] = h->grid_ok ? h->gp.RA : 0;
    info[2] = h->grid_ok ? h->gp.MT : (h->kfold_ok ? h->kf.MT : 0);
    info[3] = h->grid_ok ? h->gp.nchunks : (h->kfold_ok ? h->kf.KS2 : 0);
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_bary_stream(pcx_bary *h, void **stream) {
    PCX_API_BEGIN
    if (!h || !stream) return fail(PCX_ERR_INVALID, "NULL argument");
    *stream = (void *)h->stream;
    return PCX_OK;
    PCX_API_END
}
