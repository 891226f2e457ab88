// pcx_core.hip -- C ABI of libpcx_hip.so (see include/pcx.h): errors, devices, memory, streams and events.  gfx950 only.
//
// The library's host side is split along the handle types: pcx_core.hip (this file), pcx_bary.hip (barycentric
// handle, its launch planning and host pipelines), pcx_spline.hip (piecewise interpolant and slider on top of
// barycentric handles), pcx_tt.hip (tensor-train evaluation), pcx_ttbuild.hip (TT-Cross / TT-SVD build steps),
// pcx_comm.hip (RCCL gather).  No CPU arithmetic fallback lives in any of them: every numeric result comes from a
// HIP kernel.

#include <cstdlib>
#include <system_error>

#include "pcx_internal.h"

static thread_local char g_err[PCX_ERR_LEN] = "";
PCX_HIDDEN char *pcx_err_buf() noexcept { return g_err; }

// pcx_comm.hip reports through the same buffer
__attribute__((visibility("hidden"))) int pcx_fail_v(int code, const char *fmt, va_list ap) {
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    return code;
}

PCX_HIDDEN int pcx_guard_caught(const char *fn) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc &) {
        return fail(PCX_ERR_NOMEM, "%s: out of host memory (std::bad_alloc)", fn);
    } catch (const std::exception &e) {
        return fail(PCX_ERR_HIP, "%s: %s", fn, e.what());
    } catch (...) {
        return fail(PCX_ERR_HIP, "%s: unknown C++ exception", fn);
    }
}

PCX_HIDDEN void pcx_fault_inject(const char *fn) {
    const char *e = getenv("PCX_FAULT_INJECT");
    if (!e || !*e) return;
    const size_t n = strlen(fn);
    if (strncmp(e, fn, n) != 0) return;
    if (e[n] == '\0') throw std::bad_alloc();
    if (strcmp(e + n, ":system") == 0) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "injected");
}

extern "C" int pcx_abi_version(void) { return PCX_ABI_VERSION; }
extern "C" const char *pcx_last_error(void) { return g_err; }

extern "C" int pcx_device_count(int *n) {
    PCX_API_BEGIN
    if (!n) return fail(PCX_ERR_INVALID, "n is NULL");
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess) { *n = 0; return fail(PCX_ERR_NO_DEVICE, "%s", hipGetErrorString(e)); }
    *n = cnt;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_device_info(int device, char *name, int name_len, int *cus, int64_t *hbm) {
    PCX_API_BEGIN
    int rc = use_device(device);
    if (rc) return rc;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", prop.name, prop.gcnArchName);
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm) *hbm = (int64_t)prop.totalGlobalMem;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_device_pci_bus_id(int device, char *buf, int len) {
    PCX_API_BEGIN
    if (!buf || len < 16) return fail(PCX_ERR_INVALID, "buffer of at least 16 bytes needed");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipDeviceGetPCIBusId(buf, len, device));
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_dev_malloc(int device, size_t bytes, void **dptr) {
    PCX_API_BEGIN
    if (!dptr) return fail(PCX_ERR_INVALID, "dptr is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMalloc(dptr, bytes ? bytes : 8));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_pointer_device(const void *ptr, int *device) {
    PCX_API_BEGIN
    if (!ptr || !device) return fail(PCX_ERR_INVALID, "NULL argument");
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(PCX_ERR_INVALID, "not a HIP pointer: %s", hipGetErrorString(e)); }
    if (attr.type != hipMemoryTypeDevice) return fail(PCX_ERR_INVALID, "pointer is not device memory (memory type %d)", (int)attr.type);
    *device = attr.device;
    return PCX_OK;
    PCX_API_END
}

extern "C" int pcx_dev_free(int device, void *dptr) {
    PCX_API_BEGIN
    int rc = use_device(device);
    if (rc) return rc;
    if (dptr) HIP_TRY(hipFree(dptr));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_memcpy_h2d(int device, void *dst, const void *src, size_t bytes) {
    PCX_API_BEGIN
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_memcpy_d2h(int device, void *dst, const void *src, size_t bytes) {
    PCX_API_BEGIN
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_device_synchronize(int device) {
    PCX_API_BEGIN
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipDeviceSynchronize());
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_event_create(int device, void **event) {
    PCX_API_BEGIN
    if (!event) return fail(PCX_ERR_INVALID, "event is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    *event = (void *)ev;
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_event_record(void *event, void *stream) {
    PCX_API_BEGIN
    if (!event) return fail(PCX_ERR_INVALID, "event is NULL");
    HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_event_elapsed_ms(void *start, void *stop, float *ms) {
    PCX_API_BEGIN
    if (!start || !stop || !ms) return fail(PCX_ERR_INVALID, "NULL argument");
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_event_destroy(void *event) {
    PCX_API_BEGIN
    if (event) HIP_TRY(hipEventDestroy((hipEvent_t)event));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_stream_create(int device, void **stream) {
    PCX_API_BEGIN
    if (!stream) return fail(PCX_ERR_INVALID, "stream is NULL");
    int rc = use_device(device);
    if (rc) return rc;
    hipStream_t st;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    *stream = (void *)st;
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_stream_destroy(void *stream) {
    PCX_API_BEGIN
    if (stream) HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_stream_synchronize(void *stream) {
    PCX_API_BEGIN
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_stream_wait_event(void *stream, void *event) {
    PCX_API_BEGIN
    if (!event) return fail(PCX_ERR_INVALID, "event is NULL");
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_memcpy_h2d_async(void *dst, const void *src, size_t bytes, void *stream) {
    PCX_API_BEGIN
    if (bytes && (!dst || !src)) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_memcpy_d2h_async(void *dst, const void *src, size_t bytes, void *stream) {
    PCX_API_BEGIN
    if (bytes && (!dst || !src)) return fail(PCX_ERR_INVALID, "NULL buffer");
    if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_host_register(int device, void *ptr, size_t bytes) {
    PCX_API_BEGIN
    if (!ptr || !bytes) return fail(PCX_ERR_INVALID, "empty host range");
    int rc = use_device(device);
    if (rc) return rc;
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable));
    return PCX_OK;
    PCX_API_END
}
extern "C" int pcx_host_unregister(void *ptr) {
    PCX_API_BEGIN
    if (ptr) HIP_TRY(hipHostUnregister(ptr));
    return PCX_OK;
    PCX_API_END
}
