// pcx_ttbuild.hip -- C ABI of libpcx_hip.so (see include/pcx.h): the dense steps of the TT-Cross and TT-SVD builds.
// gfx950 only.

#include "pcx_internal.h"
#include "ttcross_kernels.h"
#include "ttsvd_kernels.h"

// ---------------------------------------------------------------------------------
// TT-Cross build steps
// ---------------------------------------------------------------------------------

extern "C" int pcx_tt_value_to_coeff_core(int device, const double *value_core, int rl, int n, int rr,
                                          double *coeff_core) {
    PCX_API_BEGIN
    if (!value_core || !coeff_core || rl < 1 || n < 1 || rr < 1) return fail(PCX_ERR_INVALID, "bad argument");
    int rc = use_device(device);
    if (rc) return rc;
    size_t cnt = (size_t)rl * n * rr;
    DevBuf in, out;
    if ((rc = in.alloc(cnt * sizeof(double)))) return rc;
    if ((rc = out.alloc(cnt * sizeof(double)))) return rc;
    HIP_TRY(hipMemcpy(in.p, value_core, cnt * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_value_to_coeff_core, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0,
                       in.as<double>(), out.as<double>(), rl, n, rr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(coeff_core, out.p, cnt * sizeof(double), hipMemcpyDeviceToHost));
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_tt_grid_eval(int device, int d, const int32_t *n_nodes, const int32_t *ranks,
                                const double *value_cores_cat, const int32_t *idx, int count,
                                double *out) {
    PCX_API_BEGIN
    if (d < 1 || d > PCX_MAX_DIMS || !n_nodes || !ranks || !value_cores_cat || count < 0) return fail(PCX_ERR_INVALID, "bad argument");
    if (count == 0) return PCX_OK;
    if (!idx || !out) return fail(PCX_ERR_INVALID, "NULL buffer");
    int rc = use_device(device);
    if (rc) return rc;
    std::vector<long> coff(d);
    long core_total = 0;
    int rmax = 1;
    for (int k = 0; k < d; ++k) {
        if (n_nodes[k] < 1 || ranks[k] < 1 || ranks[k + 1] < 1) return fail(PCX_ERR_INVALID, "bad n_nodes/ranks at dim %d", k);
        coff[k] = core_total;
        core_total += (long)ranks[k] * n_nodes[k] * ranks[k + 1];
        rmax = std::max(rmax, std::max(ranks[k], ranks[k + 1]));
    }
    for (long i = 0; i < (long)count * d; ++i)
        if (idx[i] < 0 || idx[i] >= n_nodes[i % d]) return fail(PCX_ERR_INVALID, "grid index out of range");
    DevBuf dn, dr, dc, dcores, didx, dout, dwork;
    if ((rc = dn.alloc(d * sizeof(int)))) return rc;
    if ((rc = dr.alloc((d + 1) * sizeof(int)))) return rc;
    if ((rc = dc.alloc(d * sizeof(long)))) return rc;
    if ((rc = dcores.alloc(core_total * sizeof(double)))) return rc;
    if ((rc = didx.alloc((size_t)count * d * sizeof(int)))) return rc;
    if ((rc = dout.alloc((size_t)count * sizeof(double)))) return rc;
    if ((rc = dwork.alloc((size_t)count * 2 * rmax * sizeof(double)))) return rc;
    HIP_TRY(hipMemcpy(dn.p, n_nodes, d * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dr.p, ranks, (d + 1) * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dc.p, coff.data(), d * sizeof(long), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dcores.p, value_cores_cat, core_total * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(didx.p, idx, (size_t)count * d * sizeof(int), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_tt_grid_eval, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, 0, d, dn.as<int>(),
                       dr.as<int>(), dc.as<long>(), dcores.as<double>(), didx.as<int>(), count,
                       dout.as<double>(), dwork.as<double>(), rmax);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout.p, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    return PCX_OK;
    PCX_API_END
}

// TT-SVD of a dense value tensor (reference _tt_svd_from_tensor, tensor_train.py:638-690).
extern "C" int pcx_tt_svd(int device, int d, const int32_t *n_nodes, const double *tensor, int max_rank,
                          double tol, int32_t *ranks_out, double *cores_out, int64_t cores_cap,
                          int64_t *cores_len, int32_t *sweeps_out) {
    PCX_API_BEGIN
    if (d < 1 || d > PCX_MAX_DIMS || !n_nodes || !tensor || !ranks_out || !cores_out || !cores_len)
        return fail(PCX_ERR_INVALID, "bad argument");
    if (max_rank < 1) return fail(PCX_ERR_INVALID, "max_rank must be >= 1");
    long total = 1;
    for (int k = 0; k < d; ++k) {
        if (n_nodes[k] < 1) return fail(PCX_ERR_INVALID, "n_nodes[%d] < 1", k);
        total *= n_nodes[k];
        if (total > (1L << 33)) return fail(PCX_ERR_UNSUPPORTED, "dense tensor too large for TT-SVD");
    }
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf cur, nxt, drot;
    Scratch sU, snrm, srows, sG;          // grow-only work buffers shared by the unfoldings
    struct Release { Scratch &a, &b, &c, &d; ~Release() { a.release(); b.release(); c.release(); d.release(); } } rel{sU, snrm, srows, sG};
    if ((rc = cur.alloc((size_t)total * sizeof(double)))) return rc;
    if ((rc = nxt.alloc((size_t)total * sizeof(double)))) return rc;
    if ((rc = drot.alloc(2 * sizeof(int)))) return rc;      // {pairs rotated, pairs rotated that were > 1e-8 from orthogonal}
    HIP_TRY(hipMemcpy(cur.p, tensor, (size_t)total * sizeof(double), hipMemcpyHostToDevice));
    long elems = total;
    int r_prev = 1;
    int64_t written = 0;
    int sweeps_total = 0;
    ranks_out[0] = 1;
    std::vector<double> hU, hnorm;
    std::vector<int> order;
    for (int k = 0; k < d - 1; ++k) {
        const long m_l = (long)r_prev * n_nodes[k];
        if (m_l > 8192) return fail(PCX_ERR_UNSUPPORTED, "TT-SVD unfolding with %ld rows (> 8192)", m_l);
        const int m = (int)m_l;
        const long N = elems / m;
        if ((rc = sU.reserve((size_t)m * m * sizeof(double)))) return rc;
        if ((rc = snrm.reserve((size_t)m * sizeof(double)))) return rc;
        if ((rc = srows.reserve((size_t)m * sizeof(int)))) return rc;
        DevView U{sU.ptr}, nrm{snrm.ptr}, rows{srows.ptr};
        hipLaunchKernelGGL(k_set_identity, dim3((unsigned)(((long)m * m + 255) / 256)), dim3(256), 0, 0, U.as<double>(), m);
        const int mp = (m + 1) & ~1;
        // squared norm of the largest row bounds sigma_max^2 from below (and sigma_max^2 <= m times it)
        hipLaunchKernelGGL(k_row_sqnorms, dim3(m), dim3(TTSVD_THREADS), 0, 0, cur.as<double>(), N, N, nrm.as<double>());
        hnorm.resize(m);
        HIP_TRY(hipMemcpy(hnorm.data(), nrm.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
        double fro2 = 0.0;
        for (int i = 0; i < m; ++i) fro2 += hnorm[i];
        const double eps64 = 8.0 * 2.220446049250313e-16;   // rows below 8 eps ||C||_F: noise
        const double floor2 = eps64 * eps64 * fro2;
        // pairs of rows whose squared norms add up to less than (tol * largest row norm)^2 / m are not rotated
        // against each other: see k_rowjacobi_step
        double row_max2 = 0.0;
        for (int i = 0; i < m; ++i) row_max2 = std::max(row_max2, hnorm[i]);
        const double sig2 = tol * tol * row_max2 / (double)m;
        const double rot_tol = std::max(1e-15, 2.0 * 2.220446049250313e-16 * std::sqrt((double)N));
        size_t lds_rows = (size_t)m * N * sizeof(double);
        if (m > 1 && lds_rows <= 144 * 1024) {
            // small unfolding: the whole iteration in one workgroup, rows (and U when it fits) in LDS, one launch
            const int u_in_lds = (lds_rows + (size_t)m * m * sizeof(double) <= 156 * 1024) ? 1 : 0;
            if (u_in_lds) lds_rows += (size_t)m * m * sizeof(double);
            if (lds_rows > 48 * 1024)
                HIP_TRY(hipFuncSetAttribute((const void *)k_rowjacobi_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_rows));
            HIP_TRY(hipMemsetAsync(drot.p, 0, sizeof(int), 0));
            hipLaunchKernelGGL(k_rowjacobi_lds, dim3(1), dim3(TTSVD_LDS_THREADS), lds_rows, 0, cur.as<double>(), m, (int)N,
                               U.as<double>(), floor2, rot_tol, sig2, 60, drot.as<int>(), u_in_lds);
            HIP_TRY(hipGetLastError());
            int sw = 0;
            HIP_TRY(hipMemcpy(&sw, drot.p, sizeof(int), hipMemcpyDeviceToHost));
            sweeps_total += sw;
        } else if (m > 1) {
            // large unfolding: one launch per tournament step, a sweep's (mp - 1) launches recorded once
            // in a hipGraph and replayed per sweep (the host could not issue ~100 tiny launches per
            // sweep at the rate the device finishes them: ~4 us each against ~1.5 us per boundary)
            // Gram preconditioner (ttsvd_kernels.h): rotations found on the m x m matrix C C^T in LDS make the
            // rows nearly orthogonal before the accurate row iteration starts
            const size_t lds_sym = ((size_t)2 * m * m + 2 * ((m + 1) / 2 + 1)) * sizeof(double) + (size_t)(m + 2) * sizeof(int);
            if (N > 2L * m && lds_sym <= 156 * 1024) {
                if ((rc = sG.reserve((size_t)m * m * sizeof(double)))) return rc;
                DevView G{sG.ptr};
                hipLaunchKernelGGL(k_gram_rows, dim3(m, m), dim3(TTSVD_THREADS), 0, 0, cur.as<double>(), N, m, N, G.as<double>());
                if (lds_sym > 48 * 1024)
                    HIP_TRY(hipFuncSetAttribute((const void *)k_symjacobi_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sym));
                hipLaunchKernelGGL(k_symjacobi_lds, dim3(1), dim3(TTSVD_LDS_THREADS), lds_sym, 0, G.as<double>(), m, U.as<double>(),
                                   std::max(floor2, 1e-13 * fro2), 1e-9, 30);
                hipLaunchKernelGGL(k_apply_vt, dim3((unsigned)((N + 255) / 256), m), dim3(256), 0, 0, cur.as<double>(), N, m, N,
                                   U.as<double>(), nxt.as<double>());
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipDeviceSynchronize());
                std::swap(cur.p, nxt.p);
            }
            if (mp - 1 <= 16) {
                // a short tournament (the 11-row first unfolding): plain launches, no graph to build
                for (int sweep = 0; sweep < 60; ++sweep) {
                    HIP_TRY(hipMemsetAsync(drot.p, 0, 2 * sizeof(int), 0));
                    for (int step = 0; step < mp - 1; ++step)
                        hipLaunchKernelGGL(k_rowjacobi_step, dim3(mp / 2), dim3(TTSVD_THREADS), 0, 0, cur.as<double>(), N, m, N,
                                           U.as<double>(), step, drot.as<int>(), floor2, rot_tol, sig2);
                    HIP_TRY(hipGetLastError());
                    int rotated[2] = {0, 0};
                    HIP_TRY(hipMemcpy(rotated, drot.p, 2 * sizeof(int), hipMemcpyDeviceToHost));
                    ++sweeps_total;
                    if (rotated[1] == 0) break;
                }
            } else {
            hipStream_t cs = nullptr;
            HIP_TRY(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            auto cleanup = [&]() {
                if (exec) (void)hipGraphExecDestroy(exec);
                if (graph) (void)hipGraphDestroy(graph);
                (void)hipStreamDestroy(cs);
            };
            HIP_TRY(hipDeviceSynchronize());       // the identity / norm kernels above ran on the NULL stream
            hipError_t ge = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
            if (ge == hipSuccess) {
                for (int step = 0; step < mp - 1; ++step)
                    hipLaunchKernelGGL(k_rowjacobi_step, dim3(mp / 2), dim3(TTSVD_THREADS), 0, cs, cur.as<double>(), N, m, N,
                                       U.as<double>(), step, drot.as<int>(), floor2, rot_tol, sig2);
                ge = hipStreamEndCapture(cs, &graph);
            }
            if (ge == hipSuccess) ge = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            if (ge != hipSuccess) { cleanup(); return fail(PCX_ERR_HIP, "TT-SVD sweep graph: %s", hipGetErrorString(ge)); }
            for (int sweep = 0; sweep < 60; ++sweep) {
                int rotated[2] = {0, 0};
                hipError_t e = hipMemsetAsync(drot.p, 0, 2 * sizeof(int), cs);
                if (e == hipSuccess) e = hipGraphLaunch(exec, cs);
                if (e == hipSuccess) e = hipMemcpyAsync(rotated, drot.p, 2 * sizeof(int), hipMemcpyDeviceToHost, cs);
                if (e == hipSuccess) e = hipStreamSynchronize(cs);
                if (e != hipSuccess) { cleanup(); return fail(PCX_ERR_HIP, "TT-SVD sweep: %s", hipGetErrorString(e)); }
                ++sweeps_total;
                if (rotated[1] == 0) break;
            }
            cleanup();
            }
        }
        hipLaunchKernelGGL(k_row_sqnorms, dim3(m), dim3(TTSVD_THREADS), 0, 0, cur.as<double>(), N, N, nrm.as<double>());
        HIP_TRY(hipGetLastError());
        hnorm.resize(m);
        HIP_TRY(hipMemcpy(hnorm.data(), nrm.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost));
        order.resize(m);
        for (int i = 0; i < m; ++i) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return hnorm[a] > hnorm[b]; });
        // rank rule of the reference (:673-678): cap, then drop S <= tol * S[0]
        const long len_s = std::min<long>(m, N);
        int rank = (int)std::min<long>(max_rank, len_s);
        const double s0 = std::sqrt(hnorm[order[0]]);
        if (s0 > 0.0) {
            int effective = 0;
            for (int i = 0; i < len_s; ++i) effective += (std::sqrt(hnorm[order[i]]) > tol * s0) ? 1 : 0;
            rank = std::max(1, std::min(rank, effective));
        }
        if (written + (int64_t)m * rank > cores_cap) return fail(PCX_ERR_INVALID, "cores_out too small");
        hU.resize((size_t)m * m);
        HIP_TRY(hipMemcpy(hU.data(), U.p, (size_t)m * m * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < m; ++i)
            for (int c = 0; c < rank; ++c) cores_out[written + (int64_t)i * rank + c] = hU[(size_t)i * m + order[c]];
        written += (int64_t)m * rank;
        HIP_TRY(hipMemcpy(rows.p, order.data(), (size_t)rank * sizeof(int), hipMemcpyHostToDevice));
        unsigned gx = (unsigned)std::min<long>((N + 255) / 256, 1024);
        hipLaunchKernelGGL(k_gather_rows, dim3(gx, rank), dim3(256), 0, 0, cur.as<double>(), N, N, rows.as<int>(), nxt.as<double>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        std::swap(cur.p, nxt.p);
        elems = (long)rank * N;
        r_prev = rank;
        ranks_out[k + 1] = rank;
    }
    ranks_out[d] = 1;
    if (written + elems > cores_cap) return fail(PCX_ERR_INVALID, "cores_out too small");
    HIP_TRY(hipMemcpy(cores_out + written, cur.p, (size_t)elems * sizeof(double), hipMemcpyDeviceToHost));
    written += elems;
    *cores_len = written;
    if (sweeps_out) *sweeps_out = sweeps_total;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_maxvol(int device, const double *A, int m, int r, double tol, int max_iters,
                          int64_t *idx_out) {
    PCX_API_BEGIN
    if (!A || !idx_out || m < 1 || r < 1) return fail(PCX_ERR_INVALID, "bad argument");
    if (m <= r) {  // tensor_train.py:85-86
        for (int i = 0; i < m; ++i) idx_out[i] = i;
        return PCX_OK;
    }
    if (r > TTX_MAX_R || m > TTX_MAX_M) return fail(PCX_ERR_UNSUPPORTED, "maxvol: %d x %d exceeds %d x %d", m, r, TTX_MAX_M, TTX_MAX_R);
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf dA, dB, didx;
    if ((rc = dA.alloc((size_t)m * r * sizeof(double)))) return rc;
    if ((rc = dB.alloc((size_t)m * r * sizeof(double)))) return rc;
    if ((rc = didx.alloc((size_t)r * sizeof(long long)))) return rc;
    HIP_TRY(hipMemcpy(dA.p, A, (size_t)m * r * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_maxvol, dim3(1), dim3(TTX_THREADS), 0, 0, dA.as<double>(), m, r, tol, max_iters,
                       dB.as<double>(), didx.as<long long>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(idx_out, didx.p, (size_t)r * sizeof(long long), hipMemcpyDeviceToHost));
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_tt_cross_step(int device, const double *C, int m, int c, int cap, double rel_thresh,
                                 double *chat, int64_t *pivots, int32_t *rank_out) {
    PCX_API_BEGIN
    if (!C || !chat || !pivots || !rank_out || m < 1 || c < 1 || cap < 1) return fail(PCX_ERR_INVALID, "bad argument");
    if (c > TTX_MAX_R || m > TTX_MAX_M) return fail(PCX_ERR_UNSUPPORTED, "cross step: %d x %d exceeds %d x %d", m, c, TTX_MAX_M, TTX_MAX_R);
    int rc = use_device(device);
    if (rc) return rc;
    DevBuf dC, dU, dB, dchat, dpiv, drank;
    if ((rc = dC.alloc((size_t)m * c * sizeof(double)))) return rc;
    if ((rc = dU.alloc((size_t)m * c * sizeof(double)))) return rc;
    if ((rc = dB.alloc((size_t)m * c * sizeof(double)))) return rc;
    if ((rc = dchat.alloc((size_t)m * c * sizeof(double)))) return rc;
    if ((rc = dpiv.alloc((size_t)c * sizeof(long long)))) return rc;
    if ((rc = drank.alloc(sizeof(int)))) return rc;
    HIP_TRY(hipMemcpy(dC.p, C, (size_t)m * c * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cross_step, dim3(1), dim3(TTX_THREADS), 0, 0, dC.as<double>(), m, c, cap, rel_thresh,
                       dU.as<double>(), dB.as<double>(), dchat.as<double>(), dpiv.as<long long>(),
                       drank.as<int>());
    HIP_TRY(hipGetLastError());
    int rank = 0;
    HIP_TRY(hipMemcpy(&rank, drank.p, sizeof(int), hipMemcpyDeviceToHost));
    if (rank < 1 || rank > c) return fail(PCX_ERR_HIP, "cross step returned rank %d", rank);
    HIP_TRY(hipMemcpy(chat, dchat.p, (size_t)m * rank * sizeof(double), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pivots, dpiv.p, (size_t)rank * sizeof(long long), hipMemcpyDeviceToHost));
    *rank_out = rank;
    return PCX_OK;
    PCX_API_END
}

