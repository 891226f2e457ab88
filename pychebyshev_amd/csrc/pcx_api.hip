// pcx_api.hip -- C ABI of libpcx_hip.so (see include/pcx.h).  gfx950 only.
//
// Host side: argument validation, device buffers, launch planning, kernel launches.
// No CPU arithmetic fallback lives here: every numeric result comes from a HIP kernel.

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "bary_kernels.h"
#include "tt_kernels.h"
#include "tt_lpp_kernels.h"
#include "ttcross_kernels.h"
#include "ttsvd_kernels.h"

// ---------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// the other translation units of the library (pcx_comm.hip) report through the same buffer
__attribute__((visibility("hidden"))) int pcx_fail_v(int code, const char *fmt, va_list ap) {
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    return code;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(PCX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

static int use_device(int device) {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(PCX_ERR_NO_DEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= cnt)
        return fail(PCX_ERR_NO_DEVICE, "device %d out of range [0, %d)", device, cnt);
    HIP_TRY(hipSetDevice(device));
    return PCX_OK;
}

extern "C" int pcx_abi_version(void) { return PCX_ABI_VERSION; }
extern "C" const char *pcx_last_error(void) { return g_err; }

extern "C" int pcx_device_count(int *n) {
    if (!n) return fail(PCX_ERR_INVALID, "n is NULL");
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess) { *n = 0; return fail(PCX_ERR_NO_DEVICE, "%s", hipGetErrorString(e)); }
    *n = cnt;
    return PCX_OK;
}

extern "C" int pcx_device_info(int device, char *name, int name_len, int *cus, int64_t *hbm) {
    int rc = use_device(device);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = (int64_t)prop.totalGlobalMem;
    return PCX_OK;
}

extern "C" int pcx_dev_malloc(int device, size_t bytes, void **dptr) {
    if (!dptr) return fail(PCX_ERR_INVALID, "dptr is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 8));
    return PCX_OK;
}
extern "C" int pcx_pointer_device(const void *ptr, int *device) {
    if (!ptr || !device) return fail(PCX_ERR_INVALID, "NULL argument");
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(PCX_ERR_INVALID, "not a HIP pointer: %s", hipGetErrorString(e)); }
    if (attr.type != hipMemoryTypeDevice) return fail(PCX_ERR_INVALID, "pointer is not device memory (memory type %d)", (int)attr.type);
    *device = attr.device;
    return PCX_OK;
}

extern "C" int pcx_dev_free(int device, void *dptr) {
    int rc = use_device(device);
    if (rc) return rc;
    if (dptr) HIP_TRY(hipFree(dptr));
    return PCX_OK;
}
extern "C" int pcx_memcpy_h2d(int device, void *dst, const void *src, size_t bytes) {
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return PCX_OK;
}
extern "C" int pcx_memcpy_d2h(int device, void *dst, const void *src, size_t bytes) {
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return PCX_OK;
}
extern "C" int pcx_device_synchronize(int device) {
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return PCX_OK;
}
extern "C" int pcx_event_create(int device, void **event) {
    if (!event) return fail(PCX_ERR_INVALID, "event is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    *event = (void *)ev;
    return PCX_OK;
}
extern "C" int pcx_event_record(void *event, void *stream) {
    if (!event) return fail(PCX_ERR_INVALID, "event is NULL");
    HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return PCX_OK;
}
extern "C" int pcx_event_elapsed_ms(void *start, void *stop, float *ms) {
    if (!start || !stop || !ms) return fail(PCX_ERR_INVALID, "NULL argument");
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PCX_OK;
}
extern "C" int pcx_event_destroy(void *event) {
    if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event));
    return PCX_OK;
}
extern "C" int pcx_stream_create(int device, void **stream) {
    if (!stream) return fail(PCX_ERR_INVALID, "stream is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void *)st;
    return PCX_OK;
}
extern "C" int pcx_stream_destroy(void *stream) {
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return PCX_OK;
}
extern "C" int pcx_stream_synchronize(void *stream) {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return PCX_OK;
}
extern "C" int pcx_stream_wait_event(void *stream, void *event) {
    if (!event) return fail(PCX_ERR_INVALID, "event is NULL");
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return PCX_OK;
}
extern "C" int pcx_memcpy_h2d_async(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes && (!dst || !src)) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return PCX_OK;
}
extern "C" int pcx_memcpy_d2h_async(void *dst, const void *src, size_t bytes, void *stream) {
    if (bytes && (!dst || !src)) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PCX_OK;
}
extern "C" int pcx_host_register(int device, void *ptr, size_t bytes) {
    if (!ptr || !bytes) return fail(PCX_ERR_INVALID, "empty host range");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
    return PCX_OK;
}
extern "C" int pcx_host_unregister(void *ptr) {
    if (ptr) HIP_TRY(hipHostUnregister(ptr));
    return PCX_OK;
}

// grow-only device scratch used by the host-pointer entry points
struct Scratch {
    void *ptr = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes) {
        if (bytes <= cap) return PCX_OK;
        if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&ptr, bytes);
        if (e != hipSuccess) { ptr = nullptr; return fail(PCX_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
        cap = bytes;
        return PCX_OK;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

// RAII device buffer: freed on every exit path unless release() hands the pointer on
struct DevBuf {
    void *p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
        if (e != hipSuccess) { p = nullptr; return fail(PCX_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); }
        return PCX_OK;
    }
    template <typename T> T *as() { return (T *)p; }
    template <typename T> T *release() { T *q = (T *)p; p = nullptr; return q; }
};

// typed view of memory somebody else owns (a Scratch)
struct DevView {
    void *p;
    template <typename T> T *as() { return (T *)p; }
};

// "Plain" (C-order) tensors carry PCX_PLAIN_PAD zeroed doubles behind their end: k_bary_small reads a
// row with a fixed-width run of scalar loads that may reach past the last row.
static int alloc_plain(DevBuf &b, long total) {
    int rc = b.alloc(((size_t)total + PCX_PLAIN_PAD) * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipMemset((char *)b.p + (size_t)total * sizeof(double), 0, PCX_PLAIN_PAD * sizeof(double)));
    return PCX_OK;
}

// Host-pointer batches are processed in chunks so the staging buffers stay bounded.
static const int64_t kChunkPoints = 1 << 23;
// ... and, from two such pieces on, in 256 Ki-point pieces alternating between two streams
static const int64_t kPipeChunkPoints = 1 << 18;

// Small host-pointer batches skip the H2D/D2H copies: the points are memcpy'd into a pinned,
// device-mapped buffer the kernel reads directly over PCIe, and the results land in a second
// pinned buffer (coherent host memory: visible after the stream sync).  Two API calls fewer
// per query; this is the single-query latency path.
static const size_t kPinnedBytes = 512 * 1024;
struct Pinned {
    void *in = nullptr, *out = nullptr;
    bool tried = false;
    bool ready() {
        if (!tried) {
            tried = true;
            if (hipHostMalloc(&in, kPinnedBytes, hipHostMallocMapped) != hipSuccess) in = nullptr;
            if (hipHostMalloc(&out, kPinnedBytes, hipHostMallocMapped) != hipSuccess) out = nullptr;
            (void)hipGetLastError();
        }
        return in && out;
    }
    void release() {
        if (in) (void)hipHostFree(in);
        if (out) (void)hipHostFree(out);
        in = out = nullptr;
    }
};

// A copy from or to PAGEABLE caller memory blocks the calling thread until it is done, so a single host thread runs a
// piece's upload and the previous piece's download one after the other (40 + 8 bytes per TT point at the pageable rate:
// the whole host-pointer path).  The downloads of a pipelined batch therefore go to a helper thread: it issues each one
// on the piece's own stream -- behind that piece's kernel -- while the caller's thread is inside the next upload.  The
// two threads touch different allocations (points / results).  Jobs are issued in order; the caller's thread waits for
// job i to have been ISSUED before it queues anything else on that stream or frees its source buffer.
struct Downloader {
    struct Job { void *dst; const void *src; size_t bytes; hipStream_t st; };
    int device;
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    long pushed = 0, issued = 0;
    bool stop = false;
    int rc = PCX_OK;
    std::string err;
    explicit Downloader(int dev) : device(dev) {}
    Downloader(const Downloader &) = delete;
    Downloader &operator=(const Downloader &) = delete;
    void run() {
        (void)hipSetDevice(device);
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                j = q.front();
                q.pop_front();
            }
            hipError_t e = hipSuccess;
            if (rc == PCX_OK) e = hipMemcpyAsync(j.dst, j.src, j.bytes, hipMemcpyDeviceToHost, j.st);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (e != hipSuccess && rc == PCX_OK) { rc = PCX_ERR_HIP; err = std::string("download: ") + hipGetErrorString(e); }
                ++issued;
            }
            cv.notify_all();
        }
    }
    void push(void *dst, const void *src, size_t bytes, hipStream_t st) {
        {
            std::lock_guard<std::mutex> lk(mu);
            q.push_back(Job{dst, src, bytes, st});
            ++pushed;
        }
        if (!th.joinable()) th = std::thread([this] { run(); });
        cv.notify_all();
    }
    void wait_issued(long count) {               // until the first `count` jobs have been issued
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return issued >= count; });
    }
    int finish() {                               // every job issued, the thread gone; the streams are the caller's to drain
        if (th.joinable()) {
            wait_issued(pushed);
            { std::lock_guard<std::mutex> lk(mu); stop = true; }
            cv.notify_all();
            th.join();
        }
        return rc == PCX_OK ? PCX_OK : fail(rc, "%s", err.c_str());
    }
    ~Downloader() { (void)finish(); }
};

// ---------------------------------------------------------------------------------
// barycentric handle
// ---------------------------------------------------------------------------------
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
static const int kMaxSpecs = 64;      // derivative specs evaluated by one launch (grid.z)
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
    if (!h) return PCX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &kv : h->cache) kv.second.free_all();
    (void)hipFree(h->d_tab);
    h->s_partial.release();
    h->pin.release();
    (void)hipFree(h->d_nodes); (void)hipFree(h->d_wts); (void)hipFree(h->d_diff);
    (void)hipFree(h->d_snodes);
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
}

// Packs dt.plain into MFMA fragments; on failure dt.frag / dt.slot are released again.
static int bary_pack(pcx_bary *h, DerivedTensor &dt) {
    if (!h->mfma_ok) return PCX_OK;
    const BaryMfmaPlan &p = h->plan;
    size_t cnt = (size_t)p.MT * p.KS * 64;
    DevBuf frag, slot;
    int rc = frag.alloc(cnt * sizeof(double));
    if (rc) return rc;
    int blocks = (int)((cnt + 255) / 256);
    hipLaunchKernelGGL(k_pack_fragments, dim3(blocks), dim3(256), 0, h->stream, dt.plain, frag.as<double>(),
                       p.M, p.K, p.MT, p.KS);
    HIP_TRY(hipGetLastError());
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
        h->nt = (h->plan.KS > 32) ? 1 : 2;
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
            h->small_preferred = total <= kSmallTensorElems && nl <= 48;
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
                h->sq_preferred = sq_auto && (d <= 3 || total <= kSmallTensorElems) && nl != 21 && nl != 23 &&
                                  !(d >= 3 && nl > 24 && nl != 32);
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
}

// .pcb v1 writer (reference _binary.py:208-283, write side): 12-byte header, d, lower bounds,
// upper bounds, n_nodes, tensor_values in C order -- all little-endian, no padding.  The tensor
// is the handle's untransformed device copy, so load -> save reproduces the file byte for byte.
extern "C" int pcx_bary_save_pcb(pcx_bary *h, const char *path, const double *lo, const double *hi) {
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
}

extern "C" int pcx_bary_shape(pcx_bary *h, int32_t *d_out, int32_t *n_nodes_out) {
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (d_out) *d_out = h->dims.d;
    if (n_nodes_out)
        for (int k = 0; k < PCX_MAX_DIMS; ++k) n_nodes_out[k] = k < h->dims.d ? h->dims.n[k] : 0;
    return PCX_OK;
}

// Returns (building on first use) the derivative-transformed tensor for `deriv`.
// Caller holds h->mu.  Work is enqueued on h->stream.
static int bary_get_tensor(pcx_bary *h, const int32_t *deriv, DerivedTensor **out) {
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
static int bary_effective_variant(const pcx_bary *h) {
    if (h->variant != 0) return h->variant;
    if (h->sq_nl && h->sq_preferred) return 5;
    return (h->small_nlp && (h->small_preferred || !h->mfma_ok)) ? 4 : (h->mfma_ok ? 2 : 1);
}

// frag_tab is a device table holding the m tensor pointers of a multi-spec launch: fragment images for
// the MFMA kernels, plain tensors for the lane-per-point kernel (see bary_spec_table); for m = 1 the MFMA
// kernels read dts[0]->slot and the lane-per-point kernel takes dts[0]->plain directly.
// split_scratch (nullable): where split launches of small batches keep their per-chunk sums;
// perm (nullable): evaluate rows perm[0..N) of d_pts / d_out instead of rows 0..N.
static int bary_launch(pcx_bary *h, DerivedTensor *const *dts, int m, const double *const *frag_tab,
                       const double *d_pts, long N, double *d_out, long ostride, long ooff,
                       hipStream_t st, Scratch *split_scratch, const int *perm = nullptr) {
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
                             long ooff, hipStream_t st, Scratch *split_scratch) {
    const int d = h->dims.d;
    std::vector<BaryGroup> subs;
    std::vector<char> grouped;
    const bool own_stream = st == h->stream || (h->stream2 && st == h->stream2);
    bary_plan_groups(h, derivs, m, N, own_stream, subs, grouped);
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
}

extern "C" int pcx_bary_eval_batch_dev(pcx_bary *h, const double *d_pts, int64_t N,
                                       const int32_t *deriv, double *d_out, void *stream) {
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
}

// m specs at N device-resident points into d_out (N x m row-major); groups of kMaxSpecs specs per launch.
extern "C" int pcx_bary_eval_multi_batch_dev(pcx_bary *h, const double *d_pts, int64_t N, const int32_t *derivs,
                                             int m, double *d_out, void *stream) {
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
}

static int bary_eval_host(pcx_bary *h, const double *pts, int64_t N, const int32_t *derivs, int m,
                          double *out) {
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
            int rc = bary_eval_host(h, pts, N, derivs + (size_t)s0 * d0, mc, part.data());
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
            rc = bary_launch_specs(h, derivs, dts.data(), m, frag_tab, dp, cnt, dout, m, 0, st, second ? nullptr : &h->s_partial);
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
    return bary_eval_host(h, pts, N, deriv, 1, out);
}

extern "C" int pcx_bary_eval_multi_batch(pcx_bary *h, const double *pts, int64_t N,
                                         const int32_t *derivs, int m, double *out) {
    if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
    return bary_eval_host(h, pts, N, derivs, m, out);
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
struct HostPin {
    void *a = nullptr, *b = nullptr;
    // true when [p, p + bytes) is page-locked afterwards: registered here (released by the destructor) or already
    // page-locked by the caller (pcx_host_register, hipHostMalloc)
    bool pin(const void *p, size_t bytes, void **slot) {
        if (!p || !bytes) return true;
        hipPointerAttribute_t at{};
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost) {
            hipPointerAttribute_t ae{};
            if (hipPointerGetAttributes(&ae, (const char *)p + bytes - 1) == hipSuccess && ae.type == hipMemoryTypeHost) return true;
        }
        (void)hipGetLastError();
        if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterPortable) == hipSuccess) {
            *slot = const_cast<void *>(p);
            return true;
        }
        (void)hipGetLastError();
        return false;
    }
    ~HostPin() {
        if (a) (void)hipHostUnregister(a);
        if (b) (void)hipHostUnregister(b);
    }
};

// Several host threads may copy between ONE pageable allocation and their devices only when that allocation is
// page-locked as a whole.  From pageable memory the runtime page-locks each copy's range on the fly, rounded to pages;
// two threads working on adjacent row blocks then share the boundary page, and a copy that found the neighbour's
// lock went on past its end: a GPU memory access fault in the middle of the caller's result array (found by
// tools/soak.py --pin, round 3: barycentric value, 2^19 rows over two handles).  So the fan-out runs only over
// page-locked arrays: `pin` registers them for the call; when that is declined or fails, the whole batch goes
// through the first handle.
static bool fanout_arrays_locked(HostPin &hp, int pin, const void *in, size_t in_bytes, void *out, size_t out_bytes) {
    if (!pin) {
        hipPointerAttribute_t ai{}, ao{};
        const bool ok = hipPointerGetAttributes(&ai, in) == hipSuccess && ai.type == hipMemoryTypeHost &&
                        hipPointerGetAttributes(&ao, out) == hipSuccess && ao.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        return ok;
    }
    const bool ok_in = hp.pin(in, in_bytes, &hp.a);
    const bool ok_out = hp.pin(out, out_bytes, &hp.b);
    static const bool log = getenv("PCX_FANOUT_LOG") != nullptr;
    if (log && !(ok_in && ok_out))
        fprintf(stderr, "[pcx] fan-out: could not page-lock the caller's arrays (points %d, results %d): one handle takes the batch\n",
                (int)ok_in, (int)ok_out);
    return ok_in && ok_out;
}

template <typename Fn>
static int fan_out(int n_handles, int64_t N, Fn &&block_call) {
    const int64_t per = (N + n_handles - 1) / n_handles;
    std::vector<int> rcs(n_handles, PCX_OK);
    std::vector<std::string> errs(n_handles);
    std::vector<std::thread> th;
    for (int g = 0; g < n_handles; ++g) {
        const int64_t lo = std::min<int64_t>(N, (int64_t)g * per), hi = std::min<int64_t>(N, lo + per);
        if (hi <= lo) continue;
        th.emplace_back([&, g, lo, hi] {
            rcs[g] = block_call(g, lo, hi - lo);
            if (rcs[g]) errs[g] = g_err;            // g_err is thread-local: carry the message to the caller's thread
        });
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < n_handles; ++g)
        if (rcs[g]) return fail(rcs[g], "device block %d: %s", g, errs[g].c_str());
    return PCX_OK;
}

extern "C" int pcx_bary_group_eval_multi_batch(pcx_bary *const *handles, int n_handles, const double *pts, int64_t N,
                                               const int32_t *derivs, int m, double *out, int pin) {
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
        return bary_eval_host(handles[g], pts + (size_t)lo * d, cnt, derivs, m, out + (size_t)lo * m);
    });
}

extern "C" int pcx_bary_derivative_tensor(pcx_bary *h, const int32_t *deriv, double *tensor_out) {
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
}

extern "C" int pcx_tensor_contract_axis(int device, int d, const int32_t *n_nodes, const double *tensor,
                                        int axis, const double *vec, double *out) {
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
}

extern "C" int pcx_bary_set_kernel(pcx_bary *h, int variant) {
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (variant < 0 || variant > 5) return fail(PCX_ERR_INVALID, "variant %d outside [0, 5]", variant);
    if (variant == 4 && !h->small_nlp) return fail(PCX_ERR_UNSUPPORTED, "lane-per-point kernel does not cover this shape");
    if (variant == 5 && !h->sq_nl) return fail(PCX_ERR_UNSUPPORTED, "square-trailing lane-per-point kernel does not cover this shape");
    if (variant == 2 && !h->mfma_ok) return fail(PCX_ERR_UNSUPPORTED, "MFMA kernel does not cover this shape");
    if (variant == 3 && !h->mfma4_ok) return fail(PCX_ERR_UNSUPPORTED, "4x4x4 MFMA kernel does not cover this shape");
    std::lock_guard<std::mutex> lk(h->mu);
    h->variant = variant;
    return PCX_OK;
}

extern "C" int pcx_bary_set_group_span(pcx_bary *h, int span) {
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (span < 0 || span > 8) return fail(PCX_ERR_INVALID, "span %d outside [0, 8]", span);
    std::lock_guard<std::mutex> lk(h->mu);
    h->g0_span = span;
    return PCX_OK;
}

extern "C" int pcx_bary_set_group_tolerance(pcx_bary *h, double tol) {
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (!(tol >= 0.0)) return fail(PCX_ERR_INVALID, "tolerance must be >= 0");
    std::lock_guard<std::mutex> lk(h->mu);
    h->group_tol = tol;
    return PCX_OK;
}

extern "C" int pcx_bary_kernel_info(pcx_bary *h, int32_t *info) {
    if (!h || !info) return fail(PCX_ERR_INVALID, "NULL argument");
    { const int keep = h->variant; h->variant = 0; info[0] = bary_effective_variant(h); h->variant = keep; }
    info[1] = h->mfma_ok ? h->plan.MT : 0;
    info[2] = h->mfma_ok ? h->plan.KS : 0;
    info[3] = h->mfma_ok ? (int32_t)mfma_lds_bytes(h->dims, h->nt) : (256 / h->lpp) * h->dims.sum_n * 8;
    info[4] = h->mfma_ok ? 64 * h->nt : 256 / h->lpp;
    info[5] = h->mfma_ok ? h->plan.split : h->dims.d - 1;
    return PCX_OK;
}

extern "C" int pcx_bary_stream(pcx_bary *h, void **stream) {
    if (!h || !stream) return fail(PCX_ERR_INVALID, "NULL argument");
    *stream = (void *)h->stream;
    return PCX_OK;
}

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
}

extern "C" int pcx_spline_create(int device, int d, const int32_t *n_knots, const double *knots_cat,
                                 pcx_bary *const *pieces, int n_pieces, pcx_spline **out) {
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

static int launch_sq_pieces(const pcx_bary *p0, const SplinePieceModel *models, const int *blk_piece, const int *blk_first,
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
    if (form == 5) rc = launch_sq_pieces(p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream);
    else switch (d) {
    case 1: rc = launch_small_pieces_d<0>(p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream); break;
    case 2: rc = launch_small_pieces_d<1>(p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream); break;
    case 3: rc = launch_small_pieces_d<2>(p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream); break;
    case 4: rc = launch_small_pieces_d<3>(p0, dm, d_piece, d_first, d_end, m, blocks, dp, dout, perm, h->stream); break;
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
    return spline_eval_dev(h, d_pts, N, deriv, 1, d_out);
}

extern "C" int pcx_spline_eval_multi_batch_dev(pcx_spline *h, const double *d_pts, int64_t N,
                                               const int32_t *derivs, int m, double *d_out) {
    if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
    return spline_eval_dev(h, d_pts, N, derivs, m, d_out);
}

extern "C" int pcx_spline_eval_batch(pcx_spline *h, const double *pts, int64_t N, const int32_t *deriv,
                                     double *out) {
    return spline_eval_host(h, pts, N, deriv, 1, out);
}

extern "C" int pcx_spline_eval_multi_batch(pcx_spline *h, const double *pts, int64_t N,
                                           const int32_t *derivs, int m, double *out) {
    if (!derivs) return fail(PCX_ERR_INVALID, "derivs is NULL");
    return spline_eval_host(h, pts, N, derivs, m, out);
}

extern "C" int pcx_spline_piece_ids(pcx_spline *h, const double *pts, int64_t N, int32_t *ids_out) {
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
    if (!h) return PCX_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->s_pts.release(); h->s_out.release(); h->s_cols.release(); h->s_vals.release(); h->s_partial.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PCX_OK;
}

extern "C" int pcx_slider_create(int device, int d, int n_slides, pcx_bary *const *slides,
                                 const int32_t *group_sizes, const int32_t *group_dims_cat, double pivot_value,
                                 pcx_slider **out) {
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
}

extern "C" int pcx_slider_eval_batch(pcx_slider *h, const double *pts, int64_t N, const int32_t *deriv, double *out) {
    return pcx_slider_eval_multi_batch(h, pts, N, deriv, 1, out);
}

// Device-resident points (N x d) and results (N x m); synchronous on return.
extern "C" int pcx_slider_eval_multi_batch_dev(pcx_slider *h, const double *d_pts, int64_t N, const int32_t *derivs,
                                               int m, double *d_out) {
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
}

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
};

extern "C" int pcx_tt_destroy(pcx_tt *h) {
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
    h->pin.release();
    if (h->stream2) { (void)hipStreamSynchronize(h->stream2); (void)hipStreamDestroy(h->stream2); }
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PCX_OK;
}

extern "C" int pcx_tt_create(int device, int d, const int32_t *n_nodes, const int32_t *ranks,
                             const double *lo, const double *hi, const double *cores_cat,
                             const int32_t *dim_order, pcx_tt **out) {
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
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (N < 0) return fail(PCX_ERR_INVALID, "N < 0");
    if (N > 0 && (!d_pts || !d_out)) return fail(PCX_ERR_INVALID, "NULL device buffer");
    HIP_TRY(hipSetDevice(h->device));
    std::lock_guard<std::mutex> lk(h->mu);      // tt_launch reads h->variant and fills the lazy residency counts
    return tt_launch(h, d_pts, (long)N, d_out, stream ? (hipStream_t)stream : h->stream);
}

extern "C" int pcx_tt_eval_batch(pcx_tt *h, const double *pts, int64_t N, double *out) {
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
}

extern "C" int pcx_tt_group_eval_batch(pcx_tt *const *handles, int n_handles, const double *pts, int64_t N, double *out,
                                       int pin) {
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
}

extern "C" int pcx_tt_set_kernel(pcx_tt *h, int variant) {
    if (!h) return fail(PCX_ERR_INVALID, "handle is NULL");
    if (variant < 0 || variant > 4) return fail(PCX_ERR_INVALID, "variant %d outside [0, 4]", variant);
    if (variant == 2 && !h->wR) return fail(PCX_ERR_UNSUPPORTED, "W-first TT kernel does not cover this model");
    if (variant == 3 && !h->d4RA) return fail(PCX_ERR_UNSUPPORTED, "4x4x4 direct TT kernel does not cover this model");
    if (variant == 4 && !h->lppCap) return fail(PCX_ERR_UNSUPPORTED, "lane-per-point TT kernel does not cover this model");
    if (variant != 0 && h->generic) return fail(PCX_ERR_UNSUPPORTED, "ranks above 64 run on the generic kernel only");
    std::lock_guard<std::mutex> lk(h->mu);
    h->variant = variant;
    return PCX_OK;
}

extern "C" int pcx_tt_stream(pcx_tt *h, void **stream) {
    if (!h || !stream) return fail(PCX_ERR_INVALID, "NULL argument");
    *stream = (void *)h->stream;
    return PCX_OK;
}

// ---------------------------------------------------------------------------------
// TT-Cross build steps
// ---------------------------------------------------------------------------------

extern "C" int pcx_tt_value_to_coeff_core(int device, const double *value_core, int rl, int n, int rr,
                                          double *coeff_core) {
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
}

extern "C" int pcx_tt_grid_eval(int device, int d, const int32_t *n_nodes, const int32_t *ranks,
                                const double *value_cores_cat, const int32_t *idx, int count,
                                double *out) {
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
}

// TT-SVD of a dense value tensor (reference _tt_svd_from_tensor, tensor_train.py:638-690).
extern "C" int pcx_tt_svd(int device, int d, const int32_t *n_nodes, const double *tensor, int max_rank,
                          double tol, int32_t *ranks_out, double *cores_out, int64_t cores_cap,
                          int64_t *cores_len, int32_t *sweeps_out) {
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
}

extern "C" int pcx_maxvol(int device, const double *A, int m, int r, double tol, int max_iters,
                          int64_t *idx_out) {
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
}

extern "C" int pcx_tt_cross_step(int device, const double *C, int m, int c, int cap, double rel_thresh,
                                 double *chat, int64_t *pivots, int32_t *rank_out) {
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
}
