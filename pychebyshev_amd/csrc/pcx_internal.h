// pcx_internal.h -- host-side internals shared by the translation units of libpcx_hip.so (not part of the ABI).
#pragma once

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <exception>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pcx_common.h"

#define PCX_HIDDEN __attribute__((visibility("hidden")))

// ---------------------------------------------------------------------------------
// errors: one thread-local message buffer for the whole library (defined in pcx_core.hip)
// ---------------------------------------------------------------------------------
#define PCX_ERR_LEN 512
PCX_HIDDEN char *pcx_err_buf() noexcept;        // the calling thread's message buffer (PCX_ERR_LEN bytes)

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(pcx_err_buf(), PCX_ERR_LEN, fmt, ap);
    va_end(ap);
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

// ---------------------------------------------------------------------------------
// No C++ exception crosses the C ABI (include/pcx.h: "never throws"): every extern "C" body runs inside
// PCX_API_BEGIN / PCX_API_END.  std::bad_alloc (a std::vector, std::map node or handle that could not be
// allocated) becomes PCX_ERR_NOMEM, anything else (std::system_error from a thread that could not start, ...)
// PCX_ERR_HIP, both with pcx_last_error() set.  PCX_FAULT_INJECT=<entry point name> makes that entry point
// throw std::bad_alloc at its start: the CPU-side test of this guard (tests/test_host_logic.py).
// ---------------------------------------------------------------------------------
PCX_HIDDEN int pcx_guard_caught(const char *fn) noexcept;     // call inside a catch (...) block only
PCX_HIDDEN void pcx_fault_inject(const char *fn);             // throws when the environment names fn

#define PCX_API_BEGIN try { pcx_fault_inject(__func__);
#define PCX_API_END } catch (...) { return pcx_guard_caught(__func__); }

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

struct HostPin {
    void *a = nullptr, *b = nullptr;
    // true when [p, p + bytes) is page-locked afterwards: registered here (released by the destructor) or already
    // page-locked by the caller (pcx_host_register, hipHostMalloc)
    bool pin(const void *p, size_t bytes, void **slot) {
        if (!p || !bytes) return true;
        // Ask the runtime about the range FIRST, register only what is not page-locked yet.  (Round 4 tried the other order
        // -- always attempt hipHostRegister and read "already registered" off the error -- and tools/soak.py --pin ended in
        // the GPU memory access fault of round 3 again, at a host address, in the single-handle call FOLLOWING a fan-out call
        // over the same arrays: a heap range registered, unregistered and then copied from as pageable memory.  With the
        // query in front the same soak has run clean since round 3; why that is enough is still not established.)
        // Both ends and up to 14 interior probes (two separate registrations could cover just the ends).
        bool locked = true;
        const size_t probes = bytes > (size_t)16 ? 16 : 2;
        for (size_t i = 0; i < probes && locked; ++i) {
            const size_t off = i + 1 == probes ? bytes - 1 : (bytes / (probes - 1)) * i;
            hipPointerAttribute_t at{};
            locked = hipPointerGetAttributes(&at, (const char *)p + off) == hipSuccess && at.type == hipMemoryTypeHost;
        }
        (void)hipGetLastError();
        if (locked) return true;
        if (hipHostRegister(const_cast<void *>(p), bytes, hipHostRegisterPortable) == hipSuccess) {
            *slot = const_cast<void *>(p);
            return true;
        }
        (void)hipGetLastError();
        return false;
    }
    ~HostPin() {
        for (void *p : {a, b})
            if (p && hipHostUnregister(p) != hipSuccess) {
                (void)hipGetLastError();
                fprintf(stderr, "libpcx_hip: hipHostUnregister(%p) failed -- the range stays page-locked\n", p);
            }
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
            if (rcs[g]) errs[g] = pcx_err_buf();    // the buffer is thread-local: carry the message to the caller's thread
        });
    }
    for (auto &t : th) t.join();
    for (int g = 0; g < n_handles; ++g)
        if (rcs[g]) return fail(rcs[g], "device block %d: %s", g, errs[g].c_str());
    return PCX_OK;
}
