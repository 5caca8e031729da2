// Device, memory, stream, event and copy entry points of the C-ABI
// (include/katsdpsigproc_hip.h). These stand in for what the reference gets from
// PyCUDA (reference: src/katsdpsigproc/cuda.py:54-479).
#include <stdarg.h>
#include <string.h>

#include "ksp_common.h"

static thread_local char g_error[1024] = "";

void ksp_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

extern "C" {

int ksp_abi_version(void) { return KSP_ABI_VERSION; }

const char *ksp_last_error(void) { return g_error; }

int ksp_device_count(int *count)
{
    KSP_REQUIRE(count != nullptr, "count is NULL");
    hipError_t e = hipGetDeviceCount(count);
    if (e == hipErrorNoDevice) {
        *count = 0;
        return 0;
    }
    KSP_CHECK(e);
    return 0;
}

int ksp_device_get_props(int device, ksp_device_props *props)
{
    KSP_REQUIRE(props != nullptr, "props is NULL");
    hipDeviceProp_t p;
    KSP_CHECK(hipGetDeviceProperties(&p, device));
    memset(props, 0, sizeof(*props));
    strncpy(props->name, p.name, sizeof(props->name) - 1);
    strncpy(props->arch, p.gcnArchName, sizeof(props->arch) - 1);
    props->compute_units = p.multiProcessorCount;
    props->wavefront_size = p.warpSize;
    props->max_threads_per_block = p.maxThreadsPerBlock;
    props->lds_bytes_per_block = (int32_t)p.sharedMemPerBlock;
    props->clock_khz = p.clockRate;
    props->total_memory = (int64_t)p.totalGlobalMem;
    int v = 0;
    if (hipDriverGetVersion(&v) == hipSuccess) props->driver_version = v;
    if (hipRuntimeGetVersion(&v) == hipSuccess) props->runtime_version = v;
    return 0;
}

int ksp_malloc(int device, size_t bytes, void **ptr)
{
    KSP_REQUIRE(ptr != nullptr, "ptr is NULL");
    KSP_CHECK(hipSetDevice(device));
    if (bytes == 0) bytes = 1;
    KSP_CHECK(hipMalloc(ptr, bytes));
    return 0;
}

int ksp_free(int device, void *ptr)
{
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipFree(ptr));
    return 0;
}

int ksp_host_alloc(size_t bytes, void **ptr)
{
    KSP_REQUIRE(ptr != nullptr, "ptr is NULL");
    if (bytes == 0) bytes = 1;
    KSP_CHECK(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return 0;
}

int ksp_host_free(void *ptr)
{
    KSP_CHECK(hipHostFree(ptr));
    return 0;
}

int ksp_stream_create(int device, void **stream)
{
    KSP_REQUIRE(stream != nullptr, "stream is NULL");
    KSP_CHECK(hipSetDevice(device));
    hipStream_t s;
    KSP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return 0;
}

int ksp_stream_destroy(int device, void *stream)
{
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return 0;
}

int ksp_stream_synchronize(int device, void *stream)
{
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int ksp_event_create(int device, void **event)
{
    KSP_REQUIRE(event != nullptr, "event is NULL");
    KSP_CHECK(hipSetDevice(device));
    hipEvent_t e;
    // Blocking-sync events, as the reference creates them (cuda.py:463).
    KSP_CHECK(hipEventCreateWithFlags(&e, hipEventBlockingSync));
    *event = (void *)e;
    return 0;
}

int ksp_event_create_ordering(int device, void **event)
{
    KSP_REQUIRE(event != nullptr, "event is NULL");
    KSP_CHECK(hipSetDevice(device));
    hipEvent_t e;
    KSP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
    *event = (void *)e;
    return 0;
}

int ksp_event_destroy(int device, void *event)
{
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipEventDestroy((hipEvent_t)event));
    return 0;
}

int ksp_event_record(int device, void *event, void *stream)
{
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return 0;
}

int ksp_event_synchronize(int device, void *event)
{
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipEventSynchronize((hipEvent_t)event));
    return 0;
}

int ksp_event_elapsed_ms(int device, void *start, void *end, float *ms)
{
    KSP_REQUIRE(ms != nullptr, "ms is NULL");
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)end));
    return 0;
}

int ksp_stream_wait_event(int device, void *stream, void *event)
{
    KSP_CHECK(hipSetDevice(device));
    KSP_CHECK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return 0;
}

static hipMemcpyKind copy_kind(int kind)
{
    return kind == 0 ? hipMemcpyHostToDevice
                     : (kind == 1 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice);
}

int ksp_memcpy_async(int device, void *dst, const void *src, size_t bytes, int kind, void *stream)
{
    KSP_REQUIRE(kind >= 0 && kind <= 2, "bad copy kind");
    KSP_CHECK(hipSetDevice(device));
    if (bytes == 0) return 0;
    KSP_CHECK(hipMemcpyAsync(dst, src, bytes, copy_kind(kind), (hipStream_t)stream));
    return 0;
}

int ksp_memcpy_rect_async(int device, void *dst, size_t dst_origin, const size_t dst_strides[3],
                          const void *src, size_t src_origin, const size_t src_strides[3],
                          const size_t shape[3], int ndim, int kind, void *stream)
{
    KSP_REQUIRE(ndim >= 1 && ndim <= 3, "ndim must be 1..3");
    KSP_REQUIRE(kind >= 0 && kind <= 2, "bad copy kind");
    KSP_REQUIRE(dst_strides[0] == 1 && src_strides[0] == 1, "innermost stride must be 1 byte");
    KSP_CHECK(hipSetDevice(device));
    char *d = (char *)dst + dst_origin;
    const char *s = (const char *)src + src_origin;
    size_t n1 = ndim >= 2 ? shape[1] : 1;
    size_t n2 = ndim >= 3 ? shape[2] : 1;
    if (shape[0] == 0 || n1 == 0 || n2 == 0) return 0;
    size_t dp = ndim >= 2 ? dst_strides[1] : shape[0];
    size_t sp = ndim >= 2 ? src_strides[1] : shape[0];
    for (size_t k = 0; k < n2; k++) {
        char *dk = d + (ndim >= 3 ? k * dst_strides[2] : 0);
        const char *sk = s + (ndim >= 3 ? k * src_strides[2] : 0);
        if (n1 == 1)
            KSP_CHECK(hipMemcpyAsync(dk, sk, shape[0], copy_kind(kind), (hipStream_t)stream));
        else
            KSP_CHECK(hipMemcpy2DAsync(dk, dp, sk, sp, shape[0], n1, copy_kind(kind),
                                       (hipStream_t)stream));
    }
    return 0;
}

int ksp_memset_async(int device, void *ptr, int value, size_t bytes, void *stream)
{
    KSP_CHECK(hipSetDevice(device));
    if (bytes == 0) return 0;
    KSP_CHECK(hipMemsetAsync(ptr, value, bytes, (hipStream_t)stream));
    return 0;
}

}  // extern "C"
