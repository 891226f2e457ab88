/*
 * pcx.h -- C ABI of libpcx_hip.so: MI355X (gfx950) batched evaluation of Chebyshev
 * interpolants, the drop-in for PyChebyshev's evaluation hot path.
 *
 * The reference (PyChebyshev v0.21.1) is pure Python/NumPy and has NO FFI seam; the
 * seam is its Python class surface.  Each entry point below therefore names the
 * reference *method* (file:line under /root/reference/src/pychebyshev/) whose work it
 * replaces.  The Python classes in pychebyshev_amd/ are the only intended callers;
 * INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no C++/torch types cross the boundary.
 *   - return 0 (PCX_OK) or a negative PCX_ERR_* code; never throws: every entry point catches what its C++ body may
 *     raise (std::bad_alloc -> PCX_ERR_NOMEM, anything else -> PCX_ERR_HIP; csrc/pcx_internal.h, PCX_API_BEGIN / _END).
 *     pcx_last_error() returns a thread-local description of the last failure on the calling thread.
 *   - all floating point is IEEE float64; tensors are C-order (last index fastest);
 *     `pts` is an (N, d) row-major array, exactly the ndarray the reference takes.
 *   - the library copies model data at create; callers keep ownership of every buffer
 *     they pass.  Handles are immutable after create except for an internal cache of
 *     derivative-transformed tensors (mutex-protected); one handle may be used from
 *     several host threads.
 *   - one handle lives on ONE device (one process per GPU; shard batches across
 *     processes/handles -- the path has no cross-device exchange).
 *   - "_dev" variants take DEVICE pointers and only enqueue work on the given HIP
 *     stream (NULL = the handle's own stream); they do not synchronize.
 */
#ifndef PCX_H
#define PCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCX_ABI_VERSION 1

#define PCX_OK 0
#define PCX_ERR_INVALID (-1)     /* bad argument (shape, NULL, range)               */
#define PCX_ERR_NO_DEVICE (-2)   /* no usable HIP device / device index out of range */
#define PCX_ERR_HIP (-3)         /* a HIP runtime call failed                        */
#define PCX_ERR_UNSUPPORTED (-4) /* valid request outside what the kernels cover     */
#define PCX_ERR_NOMEM (-5)

#define PCX_MAX_DIMS 16

typedef struct pcx_bary pcx_bary; /* device-resident ChebyshevApproximation state */
typedef struct pcx_tt pcx_tt;     /* device-resident ChebyshevTT coefficient cores */
typedef struct pcx_spline pcx_spline; /* knots + piece handles of a ChebyshevSpline   */
typedef struct pcx_slider pcx_slider; /* partition + slide handles of a ChebyshevSlider */

/* ---- library / device ------------------------------------------------------ */
int pcx_abi_version(void);
const char *pcx_last_error(void);
int pcx_device_count(int *n);
int pcx_device_info(int device, char *name, int name_len, int *compute_units, int64_t *hbm_bytes);
/* PCI bus id of the device ("0000:05:00.0"): lets a multi-rank run prove that its ranks sat on distinct GPUs */
int pcx_device_pci_bus_id(int device, char *buf, int len);

/* Device-memory plumbing for callers that keep query batches resident in HBM
 * (bench.py, multi-GPU drivers).  `stream` arguments are hipStream_t passed as void*. */
int pcx_dev_malloc(int device, size_t bytes, void **dptr);
int pcx_dev_free(int device, void *dptr);
/* The device a pointer lives on (hipPointerGetAttributes); PCX_ERR_INVALID for host or
 * unknown memory.  Lets a host that received a device array from another library
 * (`__cuda_array_interface__`) check it before handing it to a `_dev` entry point.    */
int pcx_pointer_device(const void *ptr, int *device);
int pcx_memcpy_h2d(int device, void *dst, const void *src, size_t bytes);
int pcx_memcpy_d2h(int device, void *dst, const void *src, size_t bytes);
int pcx_device_synchronize(int device);
int pcx_event_create(int device, void **event);
int pcx_event_record(void *event, void *stream);
int pcx_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronizes on `stop` */
int pcx_event_destroy(void *event);
/* Streams and asynchronous copies for drivers that overlap the gather / download of one
 * step with the kernel of the next (bench.py, pychebyshev_amd/distributed.py).  A stream
 * belongs to the device it was created on.  pcx_host_register pins an existing host range
 * (e.g. a POSIX shared-memory result buffer every rank downloads its block into) so that
 * device-to-host copies into it run at full PCIe rate and asynchronously.               */
int pcx_stream_create(int device, void **stream);
int pcx_stream_destroy(void *stream);
int pcx_stream_synchronize(void *stream);
int pcx_stream_wait_event(void *stream, void *event);
int pcx_memcpy_h2d_async(void *dst, const void *src, size_t bytes, void *stream);
int pcx_memcpy_d2h_async(void *dst, const void *src, size_t bytes, void *stream);
int pcx_host_register(int device, void *ptr, size_t bytes);
int pcx_host_unregister(void *ptr);

/* ---- barycentric full-tensor interpolant ----------------------------------- */
/* State of ChebyshevApproximation (barycentric.py:401-414): per-dimension nodes,
 * barycentric weights and differentiation matrices (concatenated: sum n_d, sum n_d,
 * sum n_d^2 doubles, each D_d row-major) and tensor_values (prod n_d doubles, C-order).
 * Built by the host exactly as barycentric.py:440-452, :30-49, :52-77 do.            */
int pcx_bary_create(int device, int d, const int32_t *n_nodes, const double *nodes_cat,
                    const double *weights_cat, const double *diffmat_cat, const double *tensor,
                    pcx_bary **out);
int pcx_bary_destroy(pcx_bary *h);

/* Load a ChebyshevApproximation from a .pcb v1 file (reference _binary.py:208-283;
 * examples/binary_reader/reader.c there is the reference's C reader) straight into a
 * device handle.  Nodes / weights / differentiation matrices are rebuilt on the host
 * with the formulas of barycentric.py:440-452, :30-49, :52-77.  For C/C++ callers.    */
int pcx_bary_create_from_pcb(int device, const char *path, pcx_bary **out);
/* The write side of the same format (_binary.py:208-283): header, d, domain bounds, n_nodes
 * and the untransformed tensor (copied back from the device), little-endian, byte-identical
 * to what the reference writes for the same model.  lo / hi (d doubles each) give the domain;
 * both may be NULL for a handle that came from pcx_bary_create_from_pcb (it remembers its own). */
int pcx_bary_save_pcb(pcx_bary *h, const char *path, const double *lo, const double *hi);
/* Shape of a handle: d (may be NULL) and n_nodes (PCX_MAX_DIMS ints, may be NULL).   */
int pcx_bary_shape(pcx_bary *h, int32_t *d_out, int32_t *n_nodes_out);

/* vectorized_eval_batch (barycentric.py:992-1047): _apply_derivative_passes
 * (:951-990) once for `deriv` (d orders, NULL = all zero; cached per handle), then per
 * point the exact-node test |x - node| < 1e-14 / (T . w/diff) / sum(w/diff) reduction
 * over all dimensions.  Host-pointer form: copies pts in, result out, synchronizes.   */
int pcx_bary_eval_batch(pcx_bary *h, const double *pts, int64_t N, const int32_t *deriv,
                        double *out);
int pcx_bary_eval_batch_dev(pcx_bary *h, const double *d_pts, int64_t N, const int32_t *deriv,
                            double *d_out, void *stream);

/* vectorized_eval_multi (barycentric.py:1049-1112) batched over points: m derivative
 * specs (m x d orders) at each of N points; out is (N, m) row-major.                  */
int pcx_bary_eval_multi_batch(pcx_bary *h, const double *pts, int64_t N, const int32_t *derivs,
                              int m, double *out);
/* The same over SEVERAL handles of one process (the same model created on several devices -- the
 * "devices[] list" of a drop-in create): handle g evaluates the contiguous row
 * block [g ceil(N/G), ...) on its own device from its own host thread, every download lands in its slice of
 * `out`; no collective.  The blocks run concurrently only over PAGE-LOCKED arrays: pin != 0 registers the caller's
 * arrays for the call (arrays the caller has page-locked itself -- pcx_host_register -- are taken as they are);
 * with pin = 0 and pageable arrays, or when the registration fails, the whole batch goes through handles[0]
 * (several host threads never copy to or from one pageable allocation; PCX_FANOUT_LOG=1 reports the fallback).
 * Prefer pin = 0 over arrays registered ONCE (pcx_host_register, for the arrays' lifetime): a heap range registered and
 * released per call has ended later calls over the same addresses in a GPU memory access fault; the
 * Python layer passes pin = 0 unless asked (to_device(devices=..., pin=True)).
 * Replaces the reference's single-process vectorized_eval_batch (barycentric.py:992) when the process sees more
 * than one GPU.                                                                                              */
int pcx_bary_group_eval_multi_batch(pcx_bary *const *handles, int n_handles, const double *pts, int64_t N,
                                    const int32_t *derivs, int m, double *out, int pin);
/* Device-resident form: d_pts (N x d) and d_out (N x m) in HBM; enqueues on `stream`
 * (NULL = the handle's own) and returns without synchronizing.                         */
int pcx_bary_eval_multi_batch_dev(pcx_bary *h, const double *d_pts, int64_t N,
                                  const int32_t *derivs, int m, double *d_out, void *stream);

/* The derivative-transformed tensor of _apply_derivative_passes (:951-990), copied to
 * the host (prod n_d doubles) -- lets tests check kernel K3 on its own.               */
int pcx_bary_derivative_tensor(pcx_bary *h, const int32_t *deriv, double *tensor_out);

/* Contract axis `axis` of a C-order tensor (d dims, n_nodes) with `vec` (n_nodes[axis]
 * doubles): out has the remaining d-1 dims.  The device half of
 * ChebyshevApproximation.slice (barycentric.py:2064-2154, _extrude_slice.py:79-92).   */
int pcx_tensor_contract_axis(int device, int d, const int32_t *n_nodes, const double *tensor,
                             int axis, const double *vec, double *out);

/* Kernel selection and introspection.  variant: 0 = auto, 1 = row-parallel VALU kernel
 * (any shape), 2 = MFMA kernel (v_mfma_f64_16x16x4_f64), 3 = the same contraction on
 * v_mfma_f64_4x4x4_4b_f64 with LDS-staged tiles (bit-identical to 2; opt-in, never picked
 * by auto), 5 = lane-per-point kernel for mid-size tensors whose last two dimensions have the same node count
 * (4 .. 24 or 32; d <= 4: both trailing weight vectors in registers, what auto picks there above 4096 elements),
 * 4 = lane-per-point kernel for small tensors (d <= 4, last dimension <= 64 nodes;
 * what auto picks up to 4096 elements; it sums in the reference's own nesting order and
 * forms the barycentric weights from prefix / suffix products, one division per dimension).
 * PCX_ERR_UNSUPPORTED when the shape is not covered.  info_out receives
 * {variant used by auto, row tiles, k-steps, lds bytes, points per workgroup, split}.  */
int pcx_bary_set_kernel(pcx_bary *h, int variant);
/* Multi-spec batches (pcx_bary_eval_multi_batch[_dev], N >= 65,536 on the MFMA kernel): a spec and the spec ONE order
 * below it along any one dimension q (n_q <= 16) may share ONE contraction of the other dimensions, finished with D_q
 * on the per-node partial sums -- the order of operations of the reference's vectorized_eval_multi
 * (barycentric.py:1098-1110).  That rounds differently from the reference's batch path (which differentiates the
 * tensor first) by a data-dependent amount, so every candidate pair is MEASURED once per handle: a probe batch of
 * 2,048 points (domain corners, edges, interior) goes through the shared launch and through the spec's own GEMM, and
 * the pair is formed only when the two agree to `tol` of the batch's scale -- default 3e-13 (PCX_BARY_GROUP_TOL), a
 * factor of three inside the 1e-12 parity bar.  5-D Black-Scholes: delta out of the price tensor 1e-13 and gamma out
 * of the delta tensor 1e-13 (shared), vega 7e-13, dV/dT 1e-12, rho 1e-12 (own GEMMs unless tol is raised).
 * q > 0 runs on a copy of the model with q in front, built on first use (tensors up to 2^24 elements; launches on
 * the handle's own streams only).  PCX_BARY_PROBE_LOG=1 prints every measurement.
 * pcx_bary_set_group_span: 1 = the above (default; PCX_BARY_G0_SPAN overrides it at load); 2 first lets specs up to
 * two dim-0 orders apart share without a probe (price / delta / gamma in one GEMM: gamma then 4e-12 from the
 * reference's batch path); 0: every spec its own GEMM.
 * pcx_bary_count_gemms: the number of GEMM launches a call with these specs and N would execute.                 */
int pcx_bary_count_gemms(pcx_bary *h, const int32_t *derivs, int m, int64_t N, int32_t *gemms_out);
int pcx_bary_set_group_tolerance(pcx_bary *h, double tol);
int pcx_bary_set_group_span(pcx_bary *h, int span);
int pcx_bary_kernel_info(pcx_bary *h, int32_t *info_out /* 6 ints */);
/* Short MFMA plans (3-D tensors of 17 .. 65 nodes, 64^4: spline pieces, auto-N builds) lay their row tiles over the last two
 * head dimensions instead of taking 16 consecutive rows, so the head weights need no row codes (k_bary_mfma_grid).
 * info_out: {1 when the handle's MFMA plan is of that kind else 0, rows of the first tiled dimension per tile (1, 2, 4),
 * row tiles, chunks}.  PCX_BARY_GRID=0 in the environment keeps every handle on the row-code form.
 * 3-D tensors one of whose dimensions fills row tiles well (25 .. 32, 44 .. 48, 59 .. 64 nodes, or whatever prices ahead of the
 * grid plan by the measured rule of DESIGN.md 3.1f) keep that dimension in the accumulators and fold the other two into K
 * with the B operand formed per k-step (k_bary_mfma_kfold): info_out = {2, 0, row tiles, k-steps per index of the loop
 * dimension}.  PCX_BARY_KFOLD=0 switches
 * that form off. */
int pcx_bary_grid_info(pcx_bary *h, int32_t *info_out /* 4 ints */);
int pcx_bary_stream(pcx_bary *h, void **stream);

/* ---- piecewise (spline) interpolant ---------------------------------------- */
/* ChebyshevSpline (spline.py:35-700): per-dimension sorted interior knots (concatenated)
 * and one pcx_bary handle per piece in C order over the per-dimension intervals
 * (n_pieces = prod(n_knots[k] + 1)).  The spline handle BORROWS the piece handles: they
 * must outlive it and live on the same device.                                        */
int pcx_spline_create(int device, int d, const int32_t *n_knots, const double *knots_cat,
                      pcx_bary *const *pieces, int n_pieces, pcx_spline **out);
int pcx_spline_destroy(pcx_spline *h);
/* eval_batch (spline.py:633-700): piece = searchsorted(knots, x, side='right') clipped,
 * points bucketed per piece on the device, one barycentric launch per non-empty piece. */
int pcx_spline_eval_batch(pcx_spline *h, const double *pts, int64_t N, const int32_t *deriv,
                          double *out);
int pcx_spline_eval_multi_batch(pcx_spline *h, const double *pts, int64_t N, const int32_t *derivs,
                                int m, double *out);
/* The piece index of every point (spline.py:414-446, _find_piece) -- for tests/tools.  */
int pcx_spline_piece_ids(pcx_spline *h, const double *pts, int64_t N, int32_t *ids_out);
/* Device-resident points (N x d) and results (N, or N x m) on the handle's device.  Routing needs
 * the per-piece counts on the host, so these calls are synchronous: all work is done on return;
 * m <= 64. */
int pcx_spline_eval_batch_dev(pcx_spline *h, const double *d_pts, int64_t N, const int32_t *deriv,
                              double *d_out);
int pcx_spline_eval_multi_batch_dev(pcx_spline *h, const double *d_pts, int64_t N,
                                    const int32_t *derivs, int m, double *d_out);

/* ---- slider: sum of low-dimensional slides around a pivot -------------------- */
/* State of ChebyshevSlider (slider.py:80-341): the slides are barycentric handles over the
 * dimension groups of `partition` (group_sizes[s] entries of group_dims_cat each, every
 * dimension exactly once), pivot_value = f(pivot_point).  The slides are BORROWED: they must
 * outlive the slider handle.  Evaluation (slider.py:247-318, eq. 7.5 of Ruiz & Zeron):
 *   value:       pivot + sum_s (slide_s(x_group_s) - pivot), summed in slide order;
 *   derivative:  all differentiated dimensions in one slide -> that slide's derivative;
 *                spread over more than one slide -> 0.
 * The reference evaluates one point per call; these evaluate N points with one launch per
 * slide (plus a column gather and the sum), points uploaded once for all m specs. */
int pcx_slider_create(int device, int d, int n_slides, pcx_bary *const *slides,
                      const int32_t *group_sizes, const int32_t *group_dims_cat,
                      double pivot_value, pcx_slider **out);
int pcx_slider_destroy(pcx_slider *h);
int pcx_slider_eval_batch(pcx_slider *h, const double *pts, int64_t N, const int32_t *deriv,
                          double *out);
int pcx_slider_eval_multi_batch(pcx_slider *h, const double *pts, int64_t N,
                                const int32_t *derivs, int m, double *out /* N x m */);
/* Device-resident points and results; synchronous on return. */
int pcx_slider_eval_multi_batch_dev(pcx_slider *h, const double *d_pts, int64_t N,
                                    const int32_t *derivs, int m, double *d_out);

/* ---- tensor-train interpolant ---------------------------------------------- */
/* State of ChebyshevTT (tensor_train.py:1117-1138): Chebyshev COEFFICIENT cores
 * (r_{k-1}, n_k, r_k) C-order concatenated, ranks (d+1, ranks[0]=ranks[d]=1), domain,
 * and dim_order (storage position k reads user column dim_order[k]; NULL = identity). */
int pcx_tt_create(int device, int d, const int32_t *n_nodes, const int32_t *ranks,
                  const double *lo, const double *hi, const double *coeff_cores_cat,
                  const int32_t *dim_order, pcx_tt **out);
int pcx_tt_destroy(pcx_tt *h);

/* eval_batch (tensor_train.py:2217-2265): per storage dim scale to [-1,1], Chebyshev
 * polynomials T_0..T_{n-1}, contract with the core, chain-multiply.                   */
int pcx_tt_eval_batch(pcx_tt *h, const double *pts, int64_t N, double *out);
/* eval_batch fanned out over several handles (one per device), as pcx_bary_group_eval_multi_batch
 * (reference entry point: tensor_train.py:2217).                                            */
int pcx_tt_group_eval_batch(pcx_tt *const *handles, int n_handles, const double *pts, int64_t N, double *out, int pin);
int pcx_tt_eval_batch_dev(pcx_tt *h, const double *d_pts, int64_t N, double *d_out, void *stream);
/* eval_multi batched (tensor_train.py:2267-2463; the reference evaluates one point per call): value and central
 * finite-difference derivatives, out[p * m + s] for spec s.  derivs = m x d orders (0, 1 or 2; anything else is
 * PCX_ERR_INVALID, "... not supported ...") in the USER's dimension order.  The reference's rules in its order of
 * operations: h = (b - a) 1e-4, the coordinate nudged so that 1.5 h stays inside the domain, 2-point / 3-point rules
 * nested over the differenced dimensions, the 4-point rule for a mixed (1, 1) partial.  Every stencil point is formed
 * and evaluated on the device (for lane-per-point models in registers: no stencil batch in HBM); a spec with more
 * than three differenced dimensions is PCX_ERR_UNSUPPORTED.  Row p equals the per-point stencil evaluated through
 * pcx_tt_eval_batch bit for bit. */
int pcx_tt_eval_multi_batch(pcx_tt *h, const double *pts, int64_t N, const int32_t *derivs, int m, double *out);
int pcx_tt_eval_multi_batch_dev(pcx_tt *h, const double *d_pts, int64_t N, const int32_t *derivs, int m, double *d_out,
                                void *stream);
int pcx_tt_stream(pcx_tt *h, void **stream);
/* Kernel selection: 0 = auto, 1 = direct form on v_mfma_f64_16x16x4 (one GEMM over (node, left
 * rank) per dimension; ranks <= 64), 2 = small-rank "W first" form (ranks <= 12, cores in LDS),
 * 3 = small-rank direct form on v_mfma_f64_4x4x4_4b (ranks <= 12, n <= 16, cores in LDS),
 * 4 = lane-per-point VALU form (ranks <= 16, n <= 16: one point per lane, core elements as scalar
 * operands of v_fma_f64; what auto picks for ranks <= 15 since round 3).
 * PCX_ERR_UNSUPPORTED when the model is outside a form's range.
 * Models with a rank above 64 always run on a generic wave-per-point kernel.            */
int pcx_tt_set_kernel(pcx_tt *h, int variant);

/* ---- TT-Cross build steps (tensor_train.py:123-540) ------------------------- */
/* One unfolding step of _tt_cross (:332-362 and :449-474): thin SVD of the m x c cross
 * matrix C (row-major), rank = max(1, min(cap, #{S > rel_thresh*S0}, min(m,c))), maxvol
 * rows of U[:, :rank] when m > rank, C_hat = U inv(U[piv]).  Outputs: chat (m x rank
 * row-major), pivots (rank), rank.  Runs on `device` (single-workgroup HIP kernels).  */
int pcx_tt_cross_step(int device, const double *C, int m, int c, int cap, double rel_thresh,
                      double *chat, int64_t *pivots, int32_t *rank_out);

/* _maxvol (:38-120) on a row-major m x r matrix; idx_out receives r row indices.      */
int pcx_maxvol(int device, const double *A, int m, int r, double tol, int max_iters,
               int64_t *idx_out);

/* _value_core_to_coeff_core (:997-1016): DCT-II along the node axis, /n, c_0 halved.  */
int pcx_tt_value_to_coeff_core(int device, const double *value_core, int rl, int n, int rr,
                               double *coeff_core);

/* _eval_tt (:223-228) batched: TT value at `count` integer grid index tuples
 * (count x d int32, row-major) through the chain of VALUE cores (same layout as
 * coeff_cores_cat).  Used by the convergence check (:287-297).                        */
int pcx_tt_grid_eval(int device, int d, const int32_t *n_nodes, const int32_t *ranks,
                     const double *value_cores_cat, const int32_t *idx, int count, double *out);

/* TT-SVD compression of a dense C-order value tensor (d dims, n_nodes): replaces
 * _tt_svd_from_tensor (tensor_train.py:638-690), used by ChebyshevTT.from_values (:2871-2965)
 * and build(method="svd") (:1235-1243).  Each unfolding is factored on the device by a
 * one-sided Jacobi iteration on its rows; the rank rule is the reference's (cap at
 * max_rank, drop singular values <= tol * sigma_max, at least 1).  Outputs: ranks_out[d+1];
 * the VALUE cores (r_{k-1}, n_k, r_k), C order, concatenated in cores_out (capacity
 * cores_cap doubles, used length in *cores_len); sweeps_out (optional) = Jacobi sweeps run. */
int pcx_tt_svd(int device, int d, const int32_t *n_nodes, const double *tensor, int max_rank,
               double tol, int32_t *ranks_out, double *cores_out, int64_t cores_cap,
               int64_t *cores_len, int32_t *sweeps_out);

/* ---- multi-GPU: the final gather of the per-rank result blocks ----------------- */
/* The reference has no multi-device path (docs/roadmap.md:245); SURVEY.md 8(e) defines
 * it: query rows are sharded in contiguous blocks over one process per GPU, the model is
 * replicated, and the ONLY exchange is the gather of the result blocks on one rank --
 * here grouped ncclSend/ncclRecv of RCCL over xGMI (one direct link per peer, no ring).
 * RCCL (librccl.so.1 of the ROCm this library was built with) is dlopen'ed on the first
 * pcx_comm_* call: single-GPU users never load it.  No PyTorch anywhere.
 *
 * Bootstrap: rank 0 calls pcx_comm_unique_id and hands the PCX_COMM_ID_BYTES opaque bytes
 * to the other ranks by any host channel (pychebyshev_amd.distributed.HostGroup uses a
 * shared-memory file); every rank then calls pcx_comm_create (collective, blocking).      */
#define PCX_COMM_ID_BYTES 128
typedef struct pcx_comm pcx_comm;
int pcx_comm_unique_id(void *id_out /* PCX_COMM_ID_BYTES */);
int pcx_comm_create(int device, int rank, int world, const void *id, pcx_comm **out);
int pcx_comm_destroy(pcx_comm *c);
/* rank, world, device, RCCL version code (any pointer may be NULL)                      */
int pcx_comm_info(pcx_comm *c, int32_t *rank, int32_t *world, int32_t *device, int32_t *rccl_version);
/* Gather blocks of doubles on `root`: rank r contributes counts[r] doubles from d_send,
 * which land at d_recv + offsets[r] on root (d_recv is ignored elsewhere; counts/offsets
 * have `world` entries and must be identical on every rank).  Enqueues on `stream`
 * (NULL = the communicator's own stream) and returns without synchronizing: ordered
 * behind the kernel that produced d_send when that ran on the same stream.              */
int pcx_comm_gatherv_dev(pcx_comm *c, const double *d_send, double *d_recv, const int64_t *counts,
                         const int64_t *offsets, int root, void *stream);
/* Host-level helpers for drivers: max over ranks of one double (the benchmark's
 * max-over-ranks step time) and a barrier; both synchronize the communicator's stream.  */
int pcx_comm_allreduce_max(pcx_comm *c, double *value_inout);
int pcx_comm_barrier(pcx_comm *c);
int pcx_comm_stream(pcx_comm *c, void **stream);

#ifdef __cplusplus
}
#endif
#endif /* PCX_H */
