// pcx_tt.hip -- C ABI of libpcx_hip.so (see include/pcx.h): tensor-train evaluation.  gfx950 only.

#include "pcx_internal.h"
#include "tt_kernels.h"
#include "tt_lpp_kernels.h"
#include "tt_fd_kernels.h"

// ---------------------------------------------------------------------------------
// tensor-train handle
// ---------------------------------------------------------------------------------
struct pcx_tt {
    int device = 0;
    hipStream_t stream = nullptr;
    TTDims dims;
    TTRanks rk;
    int rmax = 1;
    int cls = 0;  // 0: RC<=4,RT=1,NT=4   1: RC<=8,RT=2,NT=2   2: RC<=16,RT=4,NT=1
    double *d_frag = nullptr;
    double *d_last = nullptr;   // last core [a][j] (right rank 1), zero-padded to 4 RC rows, for the VALU tail
    long last_lds_doubles = 0;  // its size when it is small enough to be copied to LDS, else 0
    int rl_last = 1;
    // small-rank "W first" form (ranks <= 12, packed cores resident in LDS)
    int wR = 0;           // 0 = not available, else padded rank 4 / 8 / 12
    TTWPlan wplan;
    long w_resident = 0;  // workgroups of the W-first kernel the device keeps resident (lazy)
    double *d_img = nullptr;
    // small-rank direct form on the 4x4x4 MFMA (ranks <= 12, n <= 16, packed cores resident in LDS)
    int d4RA = 0;         // 0 = not available, else left/right chunks of 4: 1, 2, 3
    bool d4_preferred = true;   // auto: the cheaper of this form and the W-first form (cycle estimate at create)
    TTD4Plan d4plan;
    long d4_resident = 0;
    double *d_img4 = nullptr;
    // lane-per-point VALU form (tt_lpp_kernels.h; ranks <= 16, n <= 16): exact image [b][a][j] + per-dim table
    int lppCap = 0;       // 0 = not available, else the instantiation's rank cap: 8, 12 or 16
    int lpp_nodes = 0;    // the node count every dimension shares (instantiations with one switch level), 0 = they differ
    bool lpp_preferred = false;  // auto takes it (ranks <= 15: measured ahead of every MFMA form, profiles/r03_tt_rate_probe.txt)
    double *d_lpp_img = nullptr;
    TTLppDim *d_lpp_tab = nullptr;
    int variant = 0;      // 0 auto, 1 direct form (16x16x4), 2 W-first form, 3 direct form (4x4x4), 4 lane per point
    bool generic = false; // ranks > 64: wave-per-point kernel on the plain cores
    TTGeneric gi;
    double *d_cores = nullptr;
    std::mutex mu;
    Scratch s_pts, s_out;
    hipStream_t stream2 = nullptr;   // second staging slot of the host-pointer pipeline (lazy)
    Scratch s_pts2, s_out2;
    Pinned pin;           // zero-copy staging for small host-pointer batches
    Scratch s_fd_batch, s_fd_vals;   // finite-difference stencil batch and its values (models off the lane-per-point kernel)
};

extern "C" int pcx_tt_destroy(pcx_tt *h) {
    PCX_API_BEGIN
    if (!h) return PCX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->d_frag);
    (void)hipFree(h->d_last);
    (void)hipFree(h->d_img);
    (void)hipFree(h->d_img4);
    (void)hipFree(h->d_lpp_img);
    (void)hipFree(h->d_lpp_tab);
    (void)hipFree(h->d_cores);
    h->s_pts.release(); h->s_out.release();
    h->s_pts2.release(); h->s_out2.release();
    h->s_fd_batch.release(); h->s_fd_vals.release();
    h->pin.release();
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_tt_create(int device, int d, const int32_t *n_nodes, const int32_t *ranks,
                             const double *lo, const double *hi, const double *cores_cat,
                             const int32_t *dim_order, pcx_tt **out) {
    PCX_API_BEGIN
    if (!out) return fail(PCX_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (d < 1 || d > PCX_MAX_DIMS) return fail(PCX_ERR_INVALID, "d=%d outside [1, %d]", d, PCX_MAX_DIMS);
    if (!n_nodes || !ranks || !lo || !hi || !cores_cat) return fail(PCX_ERR_INVALID, "NULL model array");
    if (ranks[0] != 1 || ranks[d] != 1) return fail(PCX_ERR_INVALID, "boundary TT ranks must be 1");
    int rc = use_device(device);
    if (rc) return rc;
    pcx_tt *h = new (std::nothrow) pcx_tt();
    if (!h) return fail(PCX_ERR_NOMEM, "out of host memory");
    h->device = device;
    h->dims.d = d;
    std::vector<char> seen(d, 0);
    long frag_total = 0, core_total = 0;
    std::vector<long> coff(d);
    for (int k = 0; k < d; ++k) {
        if (n_nodes[k] < 1 || n_nodes[k] > 4096 || ranks[k] < 1 || ranks[k + 1] < 1) { delete h; return fail(PCX_ERR_INVALID, "bad n_nodes/ranks at dim %d", k); }
        if (!(lo[k] < hi[k])) { delete h; return fail(PCX_ERR_INVALID, "domain[%d]: lo must be < hi", k); }
        int col = dim_order ? dim_order[k] : k;
        if (col < 0 || col >= d || seen[col]) { delete h; return fail(PCX_ERR_INVALID, "dim_order is not a permutation"); }
        seen[col] = 1;
        h->dims.n[k] = n_nodes[k];
        h->dims.col[k] = col;
        h->dims.lo[k] = lo[k];
        h->dims.hi[k] = hi[k];
        h->dims.scale[k] = 2.0 / (hi[k] - lo[k]);
        coff[k] = core_total;
        core_total += (long)ranks[k] * n_nodes[k] * ranks[k + 1];
        h->rmax = std::max(h->rmax, std::max(ranks[k], ranks[k + 1]));
    }
    if (h->rmax > 64) {
        // outside the MFMA tilings: the generic wave-per-point kernel on the plain cores
        if (h->rmax > 4096) { delete h; return fail(PCX_ERR_UNSUPPORTED, "TT rank %d > 4096", h->rmax); }
        h->generic = true;
        h->gi.rmax = h->rmax;
        h->gi.nmax = 1;
        for (int k = 0; k < d; ++k) {
            h->gi.rank[k] = ranks[k];
            h->gi.coff[k] = coff[k];
            h->gi.nmax = std::max(h->gi.nmax, (int)n_nodes[k]);
        }
        // the generic kernel keeps 2 rmax + nmax doubles per wave in LDS (4 waves per workgroup)
        if ((size_t)4 * (2 * h->gi.rmax + h->gi.nmax) * sizeof(double) > 160 * 1024) {
            const int rm = h->gi.rmax, nm = h->gi.nmax;
            delete h;
            return fail(PCX_ERR_UNSUPPORTED, "TT rank %d with %d nodes exceeds the generic kernel's LDS budget "
                        "(2 rank + nodes <= 5120)", rm, nm);
        }
        h->gi.rank[d] = ranks[d];
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void **)&h->d_cores, core_total * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(h->d_cores, cores_cat, core_total * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            int c = fail(PCX_ERR_HIP, "TT create (generic): %s", hipGetErrorString(e));
            pcx_tt_destroy(h);
            return c;
        }
        *out = h;
        return PCX_OK;
    }
    h->cls = h->rmax <= 16 ? 0 : (h->rmax <= 32 ? 1 : 2);
    // direct form: dim 0 stores one left chunk; later dims are padded to the kernel's
    // compile-time RC chunks x RT tiles so that its node loop is branch-free
    {
        const int RCk = h->cls == 0 ? (h->rmax + 3) / 4 : (h->cls == 1 ? 8 : 16);
        const int RTk = h->cls == 0 ? 1 : (h->cls == 1 ? 2 : 4);
        for (int k = 0; k < d; ++k) {
            h->rk.rc[k] = (k == 0) ? 1 : RCk;
            h->rk.rt[k] = RTk;
            h->dims.frag_off[k] = frag_total;
            frag_total += (long)n_nodes[k] * h->rk.rc[k] * h->rk.rt[k] * 64;
        }
    }

#define CREATE_TRY(expr)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            int c_ = fail(PCX_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));     \
            pcx_tt_destroy(h);                                                             \
            return c_;                                                                     \
        }                                                                                  \
    } while (0)
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipMalloc((void **)&h->d_frag, frag_total * sizeof(double)));
    double *d_cores = nullptr;
    CREATE_TRY(hipMalloc((void **)&d_cores, core_total * sizeof(double)));
    CREATE_TRY(hipMemcpy(d_cores, cores_cat, core_total * sizeof(double), hipMemcpyHostToDevice));
    h->rl_last = ranks[d - 1];
    {
        // the last core as an [a][j] table zero-padded to the direct kernel's 4 RC rows
        const int RCk = h->cls == 0 ? (h->rmax + 3) / 4 : (h->cls == 1 ? 8 : 16);
        const size_t rows = (size_t)4 * RCk, nl = (size_t)n_nodes[d - 1];
        std::vector<double> padded(rows * nl, 0.0);
        for (size_t a = 0; a < (size_t)ranks[d - 1]; ++a)
            for (size_t j = 0; j < nl; ++j) padded[a * nl + j] = cores_cat[coff[d - 1] + a * nl + j];
        CREATE_TRY(hipMalloc((void **)&h->d_last, padded.size() * sizeof(double)));
        CREATE_TRY(hipMemcpy(h->d_last, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice));
        h->last_lds_doubles = (padded.size() * sizeof(double) <= 16 * 1024) ? (long)padded.size() : 0;
    }
    for (int k = 0; k < d; ++k) {
        long cnt = (long)n_nodes[k] * h->rk.rc[k] * h->rk.rt[k] * 64;
        hipLaunchKernelGGL(k_tt_pack_core, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream,
                           d_cores + coff[k], h->d_frag + h->dims.frag_off[k], ranks[k], n_nodes[k],
                           ranks[k + 1], h->rk.rc[k], h->rk.rt[k]);
    }
    // W-first image (small ranks, n <= 32): every dim but the last as (R*R) x n GEMM fragments
    // padded to a common k-step count, the last dim as an [R][4 ks] table; the whole image
    // must fit the kernel's LDS budget.
    int nmax = 1;
    for (int k = 0; k < d; ++k) nmax = std::max(nmax, (int)n_nodes[k]);
    if (h->rmax <= 12 && nmax <= 32) {
        int R = 4 * ((h->rmax + 3) / 4);
        const int ks = (nmax + 3) / 4;
        long total = 0;
        for (int k = 0; k < d; ++k) {
            h->wplan.ntiles[k] = (k == 0) ? (R + 15) / 16 : R * R / 16;   // compile-time tile counts of the kernel
            h->wplan.lds_off[k] = (int)total;
            total += (k == d - 1) ? (long)R * 4 * ks : (long)ks * h->wplan.ntiles[k] * 64;
        }
        for (int k = d; k < PCX_MAX_DIMS; ++k) { h->wplan.ntiles[k] = 0; h->wplan.lds_off[k] = 0; }
        h->wplan.ks = ks;
        h->wplan.total = (int)total;
        if (total * (long)sizeof(double) <= 96 * 1024) {
            h->wR = R;
            if (hipMalloc((void **)&h->d_img, total * sizeof(double)) != hipSuccess) h->wR = 0;
        }
        for (int k = 0; k < d && h->wR; ++k) {
            int last = (k == d - 1);
            long cnt = last ? (long)R * 4 * ks : (long)ks * h->wplan.ntiles[k] * 64;
            dim3 grid((unsigned)((cnt + 255) / 256)), block(256);
            double *dst = h->d_img + h->wplan.lds_off[k];
            const double *src = d_cores + coff[k];
            if (R == 4) hipLaunchKernelGGL(k_tt_pack_wfirst<4>, grid, block, 0, h->stream, src, dst, ranks[k], n_nodes[k], ranks[k + 1], ks, h->wplan.ntiles[k], last);
            else if (R == 8) hipLaunchKernelGGL(k_tt_pack_wfirst<8>, grid, block, 0, h->stream, src, dst, ranks[k], n_nodes[k], ranks[k + 1], ks, h->wplan.ntiles[k], last);
            else hipLaunchKernelGGL(k_tt_pack_wfirst<12>, grid, block, 0, h->stream, src, dst, ranks[k], n_nodes[k], ranks[k + 1], ks, h->wplan.ntiles[k], last);
        }
    }
    // 4x4x4 direct-form image (tt_kernels.h, k_tt_eval_d4): packed on the host from the caller's
    // cores -- dim 0: [s][slot][NMP], mid dims: [j][c][slot][NMP], last dim: [4 RA][n].
    if (h->rmax <= 12 && nmax <= PCX_D4_MAX_NODES) {
        const int RA = (h->rmax + 3) / 4;
        const int NMP = RA == 3 ? 4 : RA;
        long total = 0;
        for (int k = 0; k < d; ++k) {
            h->d4plan.lds_off[k] = (int)total;
            const long nk = n_nodes[k];
            total += (k == d - 1) ? 4L * RA * nk : (k == 0) ? ((nk + 3) / 4) * 16L * NMP : nk * RA * 16L * NMP;
        }
        for (int k = d; k < PCX_MAX_DIMS; ++k) h->d4plan.lds_off[k] = 0;
        h->d4plan.total = (int)total;
        const size_t per_wave = (size_t)16 * d + 16 * 6;
        const size_t lds_bytes = ((size_t)total + 2 * 16 * d + 4 * per_wave) * sizeof(double);
        if (lds_bytes <= 72 * 1024) {     // two workgroups per CU at least
            std::vector<double> img((size_t)total, 0.0);
            auto G = [&](int k, int a, int j, int b) -> double {
                if (a >= ranks[k] || b >= ranks[k + 1] || j >= n_nodes[k]) return 0.0;
                return cores_cat[coff[k] + ((long)a * n_nodes[k] + j) * ranks[k + 1] + b];
            };
            for (int k = 0; k < d - 1; ++k) {
                double *dst = img.data() + h->d4plan.lds_off[k];
                if (k == 0) {
                    const int ks0 = (n_nodes[0] + 3) / 4;
                    for (int s0 = 0; s0 < ks0; ++s0)
                        for (int k4 = 0; k4 < 4; ++k4)
                            for (int i = 0; i < 4; ++i)
                                for (int m = 0; m < RA; ++m)
                                    dst[s0 * 16 * NMP + (k4 * 4 + i) * NMP + m] = G(0, 0, 4 * s0 + k4, 4 * m + i);
                } else {
                    for (int j = 0; j < n_nodes[k]; ++j)
                        for (int c = 0; c < RA; ++c)
                            for (int k4 = 0; k4 < 4; ++k4)
                                for (int i = 0; i < 4; ++i)
                                    for (int m = 0; m < RA; ++m)
                                        dst[(j * RA + c) * 16 * NMP + (k4 * 4 + i) * NMP + m] = G(k, 4 * c + k4, j, 4 * m + i);
                }
            }
            {
                double *dst = img.data() + h->d4plan.lds_off[d - 1];
                const int nl = n_nodes[d - 1];
                for (int a = 0; a < 4 * RA; ++a)
                    for (int j = 0; j < nl; ++j) dst[a * nl + j] = G(d - 1, a, j, 0);
            }
            if (hipMalloc((void **)&h->d_img4, img.size() * sizeof(double)) == hipSuccess &&
                hipMemcpy(h->d_img4, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess)
                h->d4RA = RA;
            // which small-rank form auto takes: FP64-pipe cycles per 16 points and middle dimension,
            // from tools/tt_rate_probe.py (profiles/r02_tt_rate_probe.txt): the 4x4x4 form issues
            // n RA^2 instructions of ~20 cycles and pads nothing; the W-first form R^2/16 ceil(n/4)
            // instructions of ~75 cycles plus a fold.  Ranks <= 4 always favour the 4x4x4 form.
            if (h->wR && RA > 1) {
                long c4 = 0, cw = 0;
                for (int k = 1; k < d - 1; ++k) {
                    c4 += (long)n_nodes[k] * RA * RA * 20 + 100;
                    cw += (long)(h->wR * h->wR / 16) * h->wplan.ks * 75 + 160;
                }
                h->d4_preferred = c4 <= cw;
            }
        }
    }
    // lane-per-point image (tt_lpp_kernels.h): img[off_k + (b rl + a) n + j] = G_k[a][j][b], nothing padded;
    // 64 zeroed doubles behind the end (scalar loads are merged into 64-byte reads).
    if (h->rmax <= PCX_LPP_MAX_RANK && nmax <= PCX_LPP_MAX_NODES) {
        std::vector<TTLppDim> tab(d);
        std::vector<double> img((size_t)core_total + 64, 0.0);
        for (int k = 0; k < d; ++k) {
            const int rl = ranks[k], rr = ranks[k + 1], nk = n_nodes[k];
            tab[k] = TTLppDim{(int)coff[k], rl, rr, nk, h->dims.col[k], 0, lo[k], h->dims.scale[k]};
            double *dst = img.data() + coff[k];
            const double *G = cores_cat + coff[k];
            for (int b = 0; b < rr; ++b)
                for (int a = 0; a < rl; ++a)
                    for (int j = 0; j < nk; ++j) dst[((size_t)b * rl + a) * nk + j] = G[((size_t)a * nk + j) * rr + b];
        }
        if (hipMalloc((void **)&h->d_lpp_img, img.size() * sizeof(double)) == hipSuccess &&
            hipMalloc((void **)&h->d_lpp_tab, tab.size() * sizeof(TTLppDim)) == hipSuccess &&
            hipMemcpy(h->d_lpp_img, img.data(), img.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(h->d_lpp_tab, tab.data(), tab.size() * sizeof(TTLppDim), hipMemcpyHostToDevice) == hipSuccess) {
            h->lppCap = h->rmax <= 8 ? 8 : (h->rmax <= 12 ? 12 : 16);
            h->lpp_nodes = n_nodes[0];
            for (int k = 1; k < d; ++k)
                if (n_nodes[k] != n_nodes[0]) h->lpp_nodes = 0;
            h->lpp_preferred = h->rmax <= 15;       // rank 16: the 16x16x4 direct form is ahead (0.70-0.79 vs 0.66-0.69)
        }
    }
    hipError_t e1 = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(h->stream);
    (void)hipFree(d_cores);
    CREATE_TRY(e1);
    CREATE_TRY(e2);
#undef CREATE_TRY
    *out = h;
    return PCX_OK;
    PCX_API_END
}

template <int R, int KS, int NT>
static int tt_launch_wfirst(pcx_tt *h, const double *d_pts, long N, double *d_out, hipStream_t st) {
    auto kern = k_tt_eval_wfirst<R, KS, NT>;
    size_t lds = ((size_t)h->wplan.total + (size_t)(4 + 2) * 16 * NT * h->dims.d) * sizeof(double);
    if (h->w_resident == 0) {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, h->device));
        h->w_resident = std::max(1, per_cu) * std::max(1, prop.multiProcessorCount);
    }
    long per_wg = 4L * 16 * NT;
    long batches = (N + per_wg - 1) / per_wg;
    // persistent workgroups, four per resident slot (a second and third wave of workgroups
    // evens out the tail: +4 % over exactly-resident on 10^7 points), each walks a grid-stride
    // range of batches so the LDS image is loaded once per workgroup
    long blocks = std::min<long>(batches, h->w_resident * 4);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, h->dims, h->wplan, h->d_img, d_pts, d_out, N);
    HIP_TRY(hipGetLastError());
    return PCX_OK;
}

template <int RA>
static int tt_launch_d4(pcx_tt *h, const double *d_pts, long N, double *d_out, hipStream_t st) {
    auto kern = k_tt_eval_d4<RA>;
    const size_t per_wave = (size_t)16 * h->dims.d + 16 * 6;
    size_t lds = ((size_t)h->d4plan.total + (size_t)2 * 16 * h->dims.d + 4 * per_wave) * sizeof(double);
    if (h->d4_resident == 0) {
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int per_cu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, h->device));
        h->d4_resident = std::max(1, per_cu) * std::max(1, prop.multiProcessorCount);
    }
    long batches = (N + 63) / 64;
    // persistent workgroups, four per resident slot: later rounds of workgroups even out the tail
    static const int mult = [] { const char *e = getenv("PCX_D4_MULT"); return e ? std::max(1, atoi(e)) : 4; }();
    long blocks = std::min<long>(batches, h->d4_resident * mult);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, h->dims, h->d4plan, h->d_img4, d_pts, d_out, N);
    HIP_TRY(hipGetLastError());
    return PCX_OK;
}

template <int R, int NT>
static int tt_launch_wfirst_ks(pcx_tt *h, const double *d_pts, long N, double *d_out, hipStream_t st) {
    switch (h->wplan.ks) {
    case 1: return tt_launch_wfirst<R, 1, NT>(h, d_pts, N, d_out, st);
    case 2: return tt_launch_wfirst<R, 2, NT>(h, d_pts, N, d_out, st);
    case 3: return tt_launch_wfirst<R, 3, NT>(h, d_pts, N, d_out, st);
    case 4: return tt_launch_wfirst<R, 4, NT>(h, d_pts, N, d_out, st);
    case 5: return tt_launch_wfirst<R, 5, NT>(h, d_pts, N, d_out, st);
    case 6: return tt_launch_wfirst<R, 6, NT>(h, d_pts, N, d_out, st);
    case 7: return tt_launch_wfirst<R, 7, NT>(h, d_pts, N, d_out, st);
    case 8: return tt_launch_wfirst<R, 8, NT>(h, d_pts, N, d_out, st);
    }
    return fail(PCX_ERR_UNSUPPORTED, "W-first TT kernel: %d k-steps not instantiated", h->wplan.ks);
}

static int tt_launch(pcx_tt *h, const double *d_pts, long N, double *d_out, hipStream_t st) {
    if (N == 0) return PCX_OK;
    if (h->generic) {
        size_t lds = (size_t)4 * (2 * h->gi.rmax + h->gi.nmax) * sizeof(double);
        if (lds > 64 * 1024)
            HIP_TRY(hipFuncSetAttribute((const void *)k_tt_eval_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        long blocks = std::min<long>((N + 3) / 4, 256L * 8);
        hipLaunchKernelGGL(k_tt_eval_generic, dim3((unsigned)blocks), dim3(256), lds, st, h->dims, h->gi, h->d_cores, d_pts, d_out, N);
        HIP_TRY(hipGetLastError());
        return PCX_OK;
    }
    if (h->variant == 2 && !h->wR) return fail(PCX_ERR_UNSUPPORTED, "W-first TT kernel does not cover this model");
    if (h->variant == 3 && !h->d4RA) return fail(PCX_ERR_UNSUPPORTED, "4x4x4 direct TT kernel does not cover this model");
    if (h->variant == 4 && !h->lppCap) return fail(PCX_ERR_UNSUPPORTED, "lane-per-point TT kernel does not cover this model");
    if (h->lppCap && (h->variant == 4 || (h->variant == 0 && h->lpp_preferred))) {
        const long blocks = (N + PCX_LPP_WG - 1) / PCX_LPP_WG;
        if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
        const size_t lds = (size_t)h->rmax * PCX_LPP_WG * sizeof(double);
#define PCX_LPP_GO(RCAP, NJ)                                                                                        \
        hipLaunchKernelGGL((k_tt_eval_lpp<RCAP, NJ>), dim3((unsigned)blocks), dim3(PCX_LPP_WG), lds, st, h->d_lpp_tab, \
                           h->dims.d, h->d_lpp_img, d_pts, d_out, N)
#define PCX_LPP_GO_N(RCAP)                                                                                           \
        switch (h->lpp_nodes) {                                                                                      \
        case 1: PCX_LPP_GO(RCAP, 1); break; case 2: PCX_LPP_GO(RCAP, 2); break; case 3: PCX_LPP_GO(RCAP, 3); break;  \
        case 4: PCX_LPP_GO(RCAP, 4); break; case 5: PCX_LPP_GO(RCAP, 5); break; case 6: PCX_LPP_GO(RCAP, 6); break;  \
        case 7: PCX_LPP_GO(RCAP, 7); break; case 8: PCX_LPP_GO(RCAP, 8); break; case 9: PCX_LPP_GO(RCAP, 9); break;  \
        case 10: PCX_LPP_GO(RCAP, 10); break; case 11: PCX_LPP_GO(RCAP, 11); break; case 12: PCX_LPP_GO(RCAP, 12); break; \
        case 13: PCX_LPP_GO(RCAP, 13); break; case 14: PCX_LPP_GO(RCAP, 14); break; case 15: PCX_LPP_GO(RCAP, 15); break; \
        case 16: PCX_LPP_GO(RCAP, 16); break; default: PCX_LPP_GO(RCAP, 0); break;                                   \
        }
        if (h->lppCap == 8) { PCX_LPP_GO_N(8) } else if (h->lppCap == 12) { PCX_LPP_GO_N(12) } else { PCX_LPP_GO_N(16) }
#undef PCX_LPP_GO_N
#undef PCX_LPP_GO
        HIP_TRY(hipGetLastError());
        return PCX_OK;
    }
    if (h->d4RA && (h->variant == 3 || (h->variant == 0 && (!h->wR || h->d4_preferred)))) {
        if (h->d4RA == 1) return tt_launch_d4<1>(h, d_pts, N, d_out, st);
        if (h->d4RA == 2) return tt_launch_d4<2>(h, d_pts, N, d_out, st);
        return tt_launch_d4<3>(h, d_pts, N, d_out, st);
    }
    if (h->wR && h->variant != 1) {
        if (h->wR == 4) return tt_launch_wfirst_ks<4, 4>(h, d_pts, N, d_out, st);
        if (h->wR == 8) return tt_launch_wfirst_ks<8, 1>(h, d_pts, N, d_out, st);
        return tt_launch_wfirst_ks<12, 1>(h, d_pts, N, d_out, st);
    }
    auto go = [&](auto kern, int nt) -> int {
        long per_wg = 4L * 16 * nt;
        long blocks = (N + per_wg - 1) / per_wg;
        if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
        size_t lds = ((size_t)4 * 16 * nt * h->dims.d + (size_t)h->last_lds_doubles) * sizeof(double);   // query rows + last core
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, h->dims, h->rk, h->d_frag, h->d_last,
                           h->last_lds_doubles ? h->rl_last : 0, d_pts, d_out, N);
        HIP_TRY(hipGetLastError());
        return PCX_OK;
    };
    if (h->cls == 0) {
        if (h->rmax <= 4) return go(k_tt_eval_mfma<1, 1, 4>, 4);
        if (h->rmax <= 8) return go(k_tt_eval_mfma<2, 1, 4>, 4);
        if (h->rmax <= 12) return go(k_tt_eval_mfma<3, 1, 4>, 4);
        return go(k_tt_eval_mfma<4, 1, 4>, 4);
    }
    if (h->cls == 1) return go(k_tt_eval_mfma<8, 2, 2>, 2);
    return go(k_tt_eval_mfma<16, 4, 1>, 1);
}

extern "C" int pcx_tt_eval_batch_dev(pcx_tt *h, const double *d_pts, int64_t N, double *d_out,
                                     void *stream) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0) return fail(PCX_ERR_INVALID, "N < 0");
    if (N > 0 && (!d_pts || !d_out)) return fail(PCX_ERR_INVALID, "NULL device buffer");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);      // tt_launch reads h->variant and fills the lazy residency counts
    return tt_launch(h, d_pts, (long)N, d_out, stream ? (hipStream_t)stream : h->stream);
    PCX_API_END
}

extern "C" int pcx_tt_eval_batch(pcx_tt *h, const double *pts, int64_t N, double *out) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0) return fail(PCX_ERR_INVALID, "N < 0");
    if (N > 0 && (!pts || !out)) return fail(PCX_ERR_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    const int d = h->dims.d;
    if (N > 0 && (size_t)N * d * sizeof(double) <= kPinnedBytes && h->pin.ready()) {
        memcpy(h->pin.in, pts, (size_t)N * d * sizeof(double));
        int rc = tt_launch(h, (const double *)h->pin.in, (long)N, (double *)h->pin.out, h->stream);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(out, h->pin.out, (size_t)N * sizeof(double));
        return PCX_OK;
    }
    // The path is transfer-bound (48 .. 88 bytes per point against ~0.1 ns of kernel): pieces of ~10 MB of coordinates alternate
    // between two staging slots on two streams, so the upload of piece i+1 runs while piece i is evaluated and piece
    // i-1 is downloaded (both PCIe directions busy; the downloads are issued by a helper thread, see Downloader).  From page-locked caller memory (pcx_host_register, or the `pin`
    // flag of pcx_tt_group_eval_batch) the copies are asynchronous DMA; from pageable memory the driver stages them.
    // ~10 MB of coordinates per piece for batches of a few pieces (N = 10^6: 1.08 -> 1.00 ms), up to ~40 MB for long ones
    // (N = 10^7: 8.9 ms with 40 MB pieces against 9.4 ms with 10 MB pieces)
    const int64_t piece_lo = std::max<int64_t>(65536, (((int64_t)10 << 20) / (d * 8)) & ~(int64_t)65535);
    const int64_t kTTPipePoints = std::min<int64_t>(4 * piece_lo, std::max<int64_t>(piece_lo, (N / 8) & ~(int64_t)65535));
    const bool piped = N >= 2 * kTTPipePoints;
    const int64_t chunk = piped ? kTTPipePoints : kChunkPoints;
    if (piped && !h->stream2) HIP_TRY(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    // the copies queued below read and write the CALLER's arrays: whatever happens, the helper thread (Downloader) is
    // joined and both streams are drained before this call returns
    Downloader dl(h->device);
    auto pipeline = [&]() -> int {
        int slot = 0;
        long piece_no = 0;
        for (int64_t start = 0; start < N; start += chunk, ++piece_no) {
            long cnt = (long)std::min<int64_t>(chunk, N - start);
            const bool second = piped && slot == 1;
            hipStream_t st = second ? h->stream2 : h->stream;
            Scratch &sp = second ? h->s_pts2 : h->s_pts, &so = second ? h->s_out2 : h->s_out;
            if (piped && piece_no >= 2) dl.wait_issued(piece_no - 1);     // this slot's last download is behind its kernel
            int rc = sp.reserve((size_t)cnt * d * sizeof(double));
            if (rc) return rc;
            if ((rc = so.reserve((size_t)cnt * sizeof(double)))) return rc;
            HIP_TRY(hipMemcpyAsync(sp.ptr, pts + (size_t)start * d, (size_t)cnt * d * sizeof(double), hipMemcpyHostToDevice, st));
            rc = tt_launch(h, (const double *)sp.ptr, cnt, (double *)so.ptr, st);
            if (rc) return rc;
            if (!piped) {                           // single slot: download here, drain before its buffers are reused
                HIP_TRY(hipMemcpyAsync(out + start, so.ptr, (size_t)cnt * sizeof(double), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                continue;
            }
            dl.push(out + start, so.ptr, (size_t)cnt * sizeof(double), st);
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
    PCX_API_END
}

// ---------------------------------------------------------------------------------
// Finite-difference Greeks, batched (reference eval_multi + _fd_*, tensor_train.py:2267-2463; tt_fd_kernels.h)
// ---------------------------------------------------------------------------------
// derivs: m x d orders in the USER's dimension order (as ChebyshevTT.eval_multi takes them); the rules run in the
// storage frame: storage dimension k is user column dims.col[k].
static int tt_fd_plan(const pcx_tt *h, const int32_t *derivs, int m, std::vector<TTFdSpec> &specs) {
    const int d = h->dims.d;
    specs.assign((size_t)m, TTFdSpec{});
    for (int s = 0; s < m; ++s) {
        TTFdSpec &sp = specs[(size_t)s];
        sp.nact = 0;
        sp.nleaf = 1;
        for (int k = 0; k < d; ++k) {
            const int o = derivs ? derivs[(size_t)s * d + h->dims.col[k]] : 0;
            if (o == 0) continue;
            if (o != 1 && o != 2) return fail(PCX_ERR_INVALID, "Derivative order %d not supported (use 1 or 2)", o);
            if (sp.nact == PCX_FD_MAX_ACTIVE)
                return fail(PCX_ERR_UNSUPPORTED, "more than %d differenced dimensions in one spec: evaluate the stencil on the host",
                            PCX_FD_MAX_ACTIVE);
            const int i = sp.nact++;
            sp.dim[i] = k;
            sp.order[i] = o;
            sp.col[i] = h->dims.col[k];
            sp.lo[i] = h->dims.lo[k];
            sp.hi[i] = h->dims.hi[k];
            sp.h[i] = (h->dims.hi[k] - h->dims.lo[k]) * 1e-4;          // tensor_train.py:2324
            sp.need[i] = sp.h[i] * 1.5;                                 // :2331
            sp.nleaf *= o + 1;
        }
        sp.kind = sp.nact == 0 ? 0 : 1;
        if (sp.nact == 2 && sp.order[0] == 1 && sp.order[1] == 1) sp.kind = 2;       // :2413-2441, the 4-point mixed partial
    }
    return PCX_OK;
}

static bool tt_runs_lpp(const pcx_tt *h) {
    return !h->generic && h->lppCap && (h->variant == 4 || (h->variant == 0 && h->lpp_preferred));
}

// N device-resident points x m specs -> d_out[p * ostride + s]; queued on st, not awaited.  Caller holds h->mu.
static int tt_fd_launch(pcx_tt *h, const double *d_pts, long N, const std::vector<TTFdSpec> &specs, double *d_out,
                        long ostride, hipStream_t st) {
    if (N == 0) return PCX_OK;
    const int m = (int)specs.size(), d = h->dims.d;
    for (int s0 = 0; s0 < m; s0 += PCX_FD_PACK) {
        TTFdPack pack;
        pack.m = std::min(PCX_FD_PACK, m - s0);
        pack.slots = 0;
        for (int s = 0; s < pack.m; ++s) {
            pack.s[s] = specs[(size_t)(s0 + s)];
            pack.s[s].slot0 = pack.slots;
            pack.slots += pack.s[s].nleaf;
        }
        if (tt_runs_lpp(h)) {
            const long blocks = (N + PCX_LPP_WG - 1) / PCX_LPP_WG;
            if (blocks > 0x7fffffffL) return fail(PCX_ERR_UNSUPPORTED, "batch too large for one launch");
            const size_t lds = (size_t)(h->rmax + 3 * PCX_FD_MAX_ACTIVE + 1) * PCX_LPP_WG * sizeof(double);
#define PCX_FD_GO(RCAP, NJ)                                                                                             \
            hipLaunchKernelGGL((k_tt_fd_lpp<RCAP, NJ>), dim3((unsigned)blocks), dim3(PCX_LPP_WG), lds, st, h->d_lpp_tab, d,    \
                               h->rmax, h->d_lpp_img, d_pts, d_out, N, ostride, (long)s0, pack)
#define PCX_FD_GO_N(RCAP)                                                                                               \
            switch (h->lpp_nodes) {                                                                                     \
            case 1: PCX_FD_GO(RCAP, 1); break; case 2: PCX_FD_GO(RCAP, 2); break; case 3: PCX_FD_GO(RCAP, 3); break;    \
            case 4: PCX_FD_GO(RCAP, 4); break; case 5: PCX_FD_GO(RCAP, 5); break; case 6: PCX_FD_GO(RCAP, 6); break;    \
            case 7: PCX_FD_GO(RCAP, 7); break; case 8: PCX_FD_GO(RCAP, 8); break; case 9: PCX_FD_GO(RCAP, 9); break;    \
            case 10: PCX_FD_GO(RCAP, 10); break; case 11: PCX_FD_GO(RCAP, 11); break; case 12: PCX_FD_GO(RCAP, 12); break; \
            case 13: PCX_FD_GO(RCAP, 13); break; case 14: PCX_FD_GO(RCAP, 14); break; case 15: PCX_FD_GO(RCAP, 15); break; \
            case 16: PCX_FD_GO(RCAP, 16); break; default: PCX_FD_GO(RCAP, 0); break;                                     \
            }
            if (h->lppCap == 8) { PCX_FD_GO_N(8) } else if (h->lppCap == 12) { PCX_FD_GO_N(12) } else { PCX_FD_GO_N(16) }
#undef PCX_FD_GO_N
#undef PCX_FD_GO
            HIP_TRY(hipGetLastError());
            continue;
        }
        // any other evaluation kernel: stencil batch in HBM, in pieces that keep it under ~256 MB
        const long piece = std::max<long>(4096, std::min<long>(N, (256L << 20) / ((long)pack.slots * d * 8)));
        int rc = h->s_fd_batch.reserve((size_t)pack.slots * piece * d * sizeof(double));
        if (rc) return rc;
        if ((rc = h->s_fd_vals.reserve((size_t)pack.slots * piece * sizeof(double)))) return rc;
        for (long p0 = 0; p0 < N; p0 += piece) {
            const long cnt = std::min(piece, N - p0);
            hipLaunchKernelGGL(k_tt_fd_points, dim3((unsigned)((cnt + 255) / 256), (unsigned)pack.slots), dim3(256), 0, st,
                               d_pts + (size_t)p0 * d, cnt, d, (double *)h->s_fd_batch.ptr, pack);
            HIP_TRY(hipGetLastError());
            rc = tt_launch(h, (const double *)h->s_fd_batch.ptr, (long)pack.slots * cnt, (double *)h->s_fd_vals.ptr, st);
            if (rc) return rc;
            hipLaunchKernelGGL(k_tt_fd_combine, dim3((unsigned)((cnt + PCX_LPP_WG - 1) / PCX_LPP_WG)), dim3(PCX_LPP_WG), 0, st,
                               (const double *)h->s_fd_vals.ptr, cnt, d_out + (size_t)p0 * ostride, ostride, (long)s0, pack);
            HIP_TRY(hipGetLastError());
        }
    }
    return PCX_OK;
}

extern "C" int pcx_tt_eval_multi_batch_dev(pcx_tt *h, const double *d_pts, int64_t N, const int32_t *derivs, int m,
                                           double *d_out, void *stream) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0 || m < 1) return fail(PCX_ERR_INVALID, "bad N or m");
    if (N > 0 && (!d_pts || !d_out)) return fail(PCX_ERR_INVALID, "NULL device buffer");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    std::vector<TTFdSpec> specs;
    int rc = tt_fd_plan(h, derivs, m, specs);
    if (rc) return rc;
    // the generic path stages through the handle's scratch: its launches must stay on the handle's own stream
    hipStream_t st = (stream && tt_runs_lpp(h)) ? (hipStream_t)stream : h->stream;
    if (stream && st != (hipStream_t)stream) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    rc = tt_fd_launch(h, d_pts, (long)N, specs, d_out, m, st);
    if (rc) return rc;
    if (stream && st != (hipStream_t)stream) HIP_TRY(hipStreamSynchronize(st));
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_tt_eval_multi_batch(pcx_tt *h, const double *pts, int64_t N, const int32_t *derivs, int m, double *out) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0 || m < 1) return fail(PCX_ERR_INVALID, "bad N or m");
    if (N > 0 && (!pts || !out)) return fail(PCX_ERR_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);
    std::vector<TTFdSpec> specs;
    int rc = tt_fd_plan(h, derivs, m, specs);
    if (rc) return rc;
    const int d = h->dims.d;
    if (N > 0 && (size_t)N * d * sizeof(double) <= kPinnedBytes && (size_t)N * m * sizeof(double) <= kPinnedBytes && h->pin.ready()) {
        memcpy(h->pin.in, pts, (size_t)N * d * sizeof(double));
        rc = tt_fd_launch(h, (const double *)h->pin.in, (long)N, specs, (double *)h->pin.out, m, h->stream);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(h->stream));
        memcpy(out, h->pin.out, (size_t)N * m * sizeof(double));
        return PCX_OK;
    }
    // pieces of 2^18 points alternate between the two staging slots: the upload of piece i + 1 overlaps the kernel of
    // piece i (8 m bytes per point come back against 8 d going in; the kernel is (stencil points) chains per point)
    const int64_t chunk = 1 << 18;
    const bool two = N > chunk && tt_runs_lpp(h);        // the generic path's scratch is single
    if (two && !h->stream2) HIP_TRY(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
    auto pipeline = [&]() -> int {
        int slot = 0;
        for (int64_t start = 0; start < N; start += chunk, slot ^= 1) {
            const long cnt = (long)std::min<int64_t>(chunk, N - start);
            const bool second = two && slot == 1;
            hipStream_t st = second ? h->stream2 : h->stream;
            Scratch &sp = second ? h->s_pts2 : h->s_pts, &so = second ? h->s_out2 : h->s_out;
            HIP_TRY(hipStreamSynchronize(st));                       // the slot's previous download has left its buffer
            int rc2 = sp.reserve((size_t)cnt * d * sizeof(double));
            if (rc2) return rc2;
            if ((rc2 = so.reserve((size_t)cnt * m * sizeof(double)))) return rc2;
            HIP_TRY(hipMemcpyAsync(sp.ptr, pts + (size_t)start * d, (size_t)cnt * d * sizeof(double), hipMemcpyHostToDevice, st));
            rc2 = tt_fd_launch(h, (const double *)sp.ptr, cnt, specs, (double *)so.ptr, m, st);
            if (rc2) return rc2;
            HIP_TRY(hipMemcpyAsync(out + (size_t)start * m, so.ptr, (size_t)cnt * m * sizeof(double), hipMemcpyDeviceToHost, st));
        }
        return PCX_OK;
    };
    const int rc_pipe = pipeline();
    const hipError_t e1 = hipStreamSynchronize(h->stream);
    const hipError_t e2 = h->stream2 ? hipStreamSynchronize(h->stream2) : hipSuccess;
    if (rc_pipe) return rc_pipe;
    HIP_TRY(e1);
    HIP_TRY(e2);
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_tt_group_eval_batch(pcx_tt *const *handles, int n_handles, const double *pts, int64_t N, double *out,
                                       int pin) {
    PCX_API_BEGIN
    if (!handles || n_handles < 1) return fail(PCX_ERR_INVALID, "no handles");
    for (int g = 0; g < n_handles; ++g) {
        if (!handles[g]) return fail(PCX_ERR_INVALID, "handle %d is NULL", g);
        if (handles[g]->dims.d != handles[0]->dims.d) return fail(PCX_ERR_INVALID, "handle %d holds a different model", g);
    }
    if (N < 0) return fail(PCX_ERR_INVALID, "N < 0");
    if (N > 0 && (!pts || !out)) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (n_handles == 1 || N == 0) return pcx_tt_eval_batch(handles[0], pts, N, out);
    const int d = handles[0]->dims.d;
    HostPin hp;
    HIP_TRY(hipSetDevice(handles[0]->device));
    if (!fanout_arrays_locked(hp, pin, pts, (size_t)N * d * sizeof(double), out, (size_t)N * sizeof(double)))
        return pcx_tt_eval_batch(handles[0], pts, N, out);
    return fan_out(n_handles, N, [&](int g, int64_t lo, int64_t cnt) {
        return pcx_tt_eval_batch(handles[g], pts + (size_t)lo * d, cnt, out + lo);
    });
    PCX_API_END
}

extern "C" int pcx_tt_set_kernel(pcx_tt *h, int variant) {
    PCX_API_BEGIN
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (variant < 0 || variant > 4) return fail(PCX_ERR_INVALID, "variant %d outside [0, 4]", variant);
    if (variant == 2 && !h->wR) return fail(PCX_ERR_UNSUPPORTED, "W-first TT kernel does not cover this model");
    if (variant == 3 && !h->d4RA) return fail(PCX_ERR_UNSUPPORTED, "4x4x4 direct TT kernel does not cover this model");
    if (variant == 4 && !h->lppCap) return fail(PCX_ERR_UNSUPPORTED, "lane-per-point TT kernel does not cover this model");
    if (variant != 0 && h->generic) return fail(PCX_ERR_UNSUPPORTED, "ranks above 64 run on the generic kernel only");
    std::lock_guard<std::mutex> lk(h->mu);
    h->variant = variant;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_tt_stream(pcx_tt *h, void **stream) {
    PCX_API_BEGIN
    if (!h || !stream) return fail(PCX_ERR_INVALID, "NULL argument");
    *stream = (void *)h->stream;
    return PCX_OK;
    PCX_API_END
}
