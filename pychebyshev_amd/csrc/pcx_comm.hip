// pcx_comm.hip -- the path's only multi-GPU exchange: the gather of the per-rank result
// blocks, on RCCL over xGMI (include/pcx.h, "multi-GPU").  One process per GPU.
//
// The reference has no multi-device code at all (docs/roadmap.md:245 "Not pursuing GPU
// acceleration, multi-threading"); the contract is SURVEY.md 8(e): contiguous row blocks,
// replicated model, one gather to root.  MI355X's xGMI is a full mesh of point-to-point
// links, so the gather is `world - 1` concurrent ncclSend/ncclRecv pairs in one group --
// each on its own link -- not a ring.
//
// librccl.so.1 is dlopen'ed on first use so that single-GPU users never map it (it is a
// 570 MB library) and libpcx_hip.so keeps linking only libamdhip64.

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums only: every function is resolved with dlsym

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/pcx.h"

int pcx_fail_v(int code, const char *fmt, va_list ap);   // pcx_core.hip (thread-local message)
// pcx_core.hip: no C++ exception crosses the C ABI (see pcx_internal.h)
int pcx_guard_caught(const char *fn) noexcept;
void pcx_fault_inject(const char *fn);
#define PCX_API_BEGIN try { pcx_fault_inject(__func__);
#define PCX_API_END } catch (...) { return pcx_guard_caught(__func__); }

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    int rc = pcx_fail_v(code, fmt, ap);
    va_end(ap);
    return rc;
}

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(PCX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// ---------------------------------------------------------------------------------
// RCCL entry points, resolved once
// ---------------------------------------------------------------------------------
struct Rccl {
    void *so = nullptr;
    std::string path, error;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
};

static Rccl g_rccl;
static std::once_flag g_rccl_once;

static void rccl_load() {
    Rccl &r = g_rccl;
    // the RCCL of the ROCm this library was built against first: mixing it with the copy
    // a Python wheel bundles for another HIP runtime puts two ROCm stacks in one process
    const char *env = getenv("PCX_RCCL_LIBRARY");
    std::string rocm = getenv("ROCM_PATH") ? getenv("ROCM_PATH") : "/opt/rocm";
    const std::string cands[] = {env ? env : "", rocm + "/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    for (const std::string &c : cands) {
        if (c.empty()) continue;
        r.so = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.so) { r.path = c; break; }
        const char *msg = dlerror();
        r.error += c + ": " + (msg ? msg : "?") + "; ";
    }
    if (!r.so) return;
    bool ok = true;
#define SYM(field, name)                                                         \
    do {                                                                         \
        r.field = (decltype(r.field))dlsym(r.so, name);                          \
        if (!r.field) { ok = false; r.error += std::string(name) + " missing; "; } \
    } while (0)
    SYM(GetVersion, "ncclGetVersion");
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GetErrorString, "ncclGetErrorString");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
#undef SYM
    if (!ok) { dlclose(r.so); r.so = nullptr; }
}

static int rccl_ready() {
    std::call_once(g_rccl_once, rccl_load);
    if (!g_rccl.so) return fail(PCX_ERR_UNSUPPORTED, "RCCL is not loadable: %s", g_rccl.error.c_str());
    return PCX_OK;
}

#define NCCL_TRY(expr)                                                                       \
    do {                                                                                     \
        ncclResult_t r_ = (expr);                                                            \
        if (r_ != ncclSuccess)                                                               \
            return fail(PCX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(r_), \
                        __FILE__, __LINE__);                                                 \
    } while (0)

// ---------------------------------------------------------------------------------
struct pcx_comm {
    int device = 0, rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    double *d_scratch = nullptr;   // two doubles for the host-level reductions
};

extern "C" int pcx_comm_unique_id(void *id_out) {
    PCX_API_BEGIN
    if (!id_out) return fail(PCX_ERR_INVALID, "id_out is NULL");
    static_assert(sizeof(ncclUniqueId) == PCX_COMM_ID_BYTES, "RCCL unique id size");
    int rc = rccl_ready();
    if (rc) return rc;
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_comm_create(int device, int rank, int world, const void *id, pcx_comm **out) {
    PCX_API_BEGIN
    if (!out || !id) return fail(PCX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world)
        return fail(PCX_ERR_INVALID, "bad rank/world %d/%d", rank, world);
    int rc = rccl_ready();
    if (rc) return rc;
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
        return fail(PCX_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= cnt)
        return fail(PCX_ERR_NO_DEVICE, "device %d out of range [0, %d)", device, cnt);
    HIP_TRY(hipSetDevice(device));
    pcx_comm *c = new (std::nothrow) pcx_comm;
    if (!c) return fail(PCX_ERR_NOMEM, "out of host memory");
    c->device = device; c->rank = rank; c->world = world;
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(PCX_ERR_HIP, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world,
                    device, g_rccl.GetErrorString(r));
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_scratch, 2 * sizeof(double));
    if (e != hipSuccess) {
        (void)pcx_comm_destroy(c);
        return fail(PCX_ERR_HIP, "communicator resources: %s", hipGetErrorString(e));
    }
    *out = c;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_comm_destroy(pcx_comm *c) {
    PCX_API_BEGIN
    if (!c) return PCX_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && g_rccl.so) (void)g_rccl.CommDestroy(c->comm);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_comm_info(pcx_comm *c, int32_t *rank, int32_t *world, int32_t *device,
                             int32_t *rccl_version) {
    PCX_API_BEGIN
    if (!c) return fail(PCX_ERR_INVALID, "comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (device) *device = c->device;
    if (rccl_version) {
        int v = 0;
        NCCL_TRY(g_rccl.GetVersion(&v));
        *rccl_version = v;
    }
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_comm_stream(pcx_comm *c, void **stream) {
    PCX_API_BEGIN
    if (!c || !stream) return fail(PCX_ERR_INVALID, "NULL argument");
    *stream = (void *)c->stream;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_comm_gatherv_dev(pcx_comm *c, const double *d_send, double *d_recv,
                                    const int64_t *counts, const int64_t *offsets, int root,
                                    void *stream) {
    PCX_API_BEGIN
    if (!c || !counts || !offsets) return fail(PCX_ERR_INVALID, "NULL argument");
    if (root < 0 || root >= c->world) return fail(PCX_ERR_INVALID, "root %d outside [0, %d)", root, c->world);
    for (int r = 0; r < c->world; ++r)
        if (counts[r] < 0 || offsets[r] < 0) return fail(PCX_ERR_INVALID, "negative count/offset for rank %d", r);
    const int64_t mine = counts[c->rank];
    if (mine > 0 && !d_send) return fail(PCX_ERR_INVALID, "d_send is NULL");
    if (c->rank == root && !d_recv) return fail(PCX_ERR_INVALID, "d_recv is NULL on the root");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = stream ? (hipStream_t)stream : c->stream;
    if (c->rank == root) {
        // the root's own block: a device-to-device copy on the same stream
        if (mine > 0 && d_recv + offsets[root] != d_send)
            HIP_TRY(hipMemcpyAsync(d_recv + offsets[root], d_send, (size_t)mine * sizeof(double),
                                   hipMemcpyDeviceToDevice, st));
        if (c->world == 1) return PCX_OK;
        NCCL_TRY(g_rccl.GroupStart());
        for (int r = 0; r < c->world; ++r) {
            if (r == root || counts[r] == 0) continue;
            ncclResult_t rr = g_rccl.Recv(d_recv + offsets[r], (size_t)counts[r], ncclDouble, r, c->comm, st);
            if (rr != ncclSuccess) {
                (void)g_rccl.GroupEnd();
                return fail(PCX_ERR_HIP, "ncclRecv from rank %d failed: %s", r, g_rccl.GetErrorString(rr));
            }
        }
        NCCL_TRY(g_rccl.GroupEnd());
    } else if (mine > 0) {
        NCCL_TRY(g_rccl.GroupStart());
        ncclResult_t rr = g_rccl.Send(d_send, (size_t)mine, ncclDouble, root, c->comm, st);
        if (rr != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            return fail(PCX_ERR_HIP, "ncclSend to rank %d failed: %s", root, g_rccl.GetErrorString(rr));
        }
        NCCL_TRY(g_rccl.GroupEnd());
    }
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_comm_allreduce_max(pcx_comm *c, double *value) {
    PCX_API_BEGIN
    if (!c || !value) return fail(PCX_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMemcpyAsync(c->d_scratch, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCL_TRY(g_rccl.AllReduce(c->d_scratch, c->d_scratch + 1, 1, ncclDouble, ncclMax, c->comm, c->stream));
    HIP_TRY(hipMemcpyAsync(value, c->d_scratch + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_comm_barrier(pcx_comm *c) {
    PCX_API_BEGIN
    double v = 0.0;
    return pcx_comm_allreduce_max(c, &v);
    PCX_API_END
}
