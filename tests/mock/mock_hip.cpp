// mock_hip.cpp — a CPU stand-in for the handful of HIP runtime calls the library's HOST code makes, so that the host
// logic (staging, the graph searcher, the device-traversal driver and its fallback, training sequencing, error paths) can
// run in the CPU test suite.  "Device" memory is malloc'ed memory remembered in a set (hipPointerGetAttributes answers
// from it), streams and events are inert, copies are memcpy.  TEST HARNESS: linked only into build/mock/libjvector_hip_mock.so.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_set>

namespace {
std::mutex g_mu;
std::unordered_set<const void *> g_dev;
long g_dev_bytes_live = 0;
}  // namespace

extern "C" {

long mock_hip_live_device_allocations()
{
    std::lock_guard<std::mutex> lk(g_mu);
    return (long)g_dev.size();
}

// test hook: treat a caller-owned buffer (numpy / torch CPU memory) as device memory, e.g. a mutable adjacency handed to
// jv_hip_graph_set_level0_device
void mock_hip_register_device(const void *p)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev.insert(p);
}
void mock_hip_unregister_device(const void *p)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev.erase(p);
}

// JV_MOCK_DEVICES=n: the mock box has n (identical, independent) devices — one per rank of a multi-process dry run
static int mock_device_count()
{
    const char *e = getenv("JV_MOCK_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n >= 1 && n <= 64 ? n : 1;
}
hipError_t hipGetDeviceCount(int *n)
{
    *n = mock_device_count();
    return hipSuccess;
}
hipError_t hipSetDevice(int d) { return d >= 0 && d < mock_device_count() ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t *p, int d)
{
    if (d < 0 || d >= mock_device_count()) return hipErrorInvalidDevice;
    memset(p, 0, sizeof(*p));
    strcpy(p->name, "CPU mock device");
    strcpy(p->gcnArchName, "gfx950:mock");
    p->multiProcessorCount = 4;
    p->sharedMemPerBlock = 65536;
    p->maxSharedMemoryPerMultiProcessor = 163840;
    return hipSuccess;
}
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "mock HIP error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }

hipError_t hipMalloc(void **p, size_t bytes)
{
    void *m = aligned_alloc(256, (bytes + 255) & ~(size_t)255);
    if (!m) return hipErrorOutOfMemory;
    memset(m, 0xCD, bytes);  // device memory starts as garbage
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev.insert(m);
    *p = m;
    return hipSuccess;
}
hipError_t hipFree(void *p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_dev.erase(p)) return hipErrorInvalidValue;
    }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned int)
{
    *p = aligned_alloc(256, (bytes + 255) & ~(size_t)255);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p)
{
    free(p);
    return hipSuccess;
}
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p)
{
    // device allocations are tracked by BASE address only (the library never asks about interior pointers of its own
    // buffers; user pointers are either whole allocations or host memory)
    std::lock_guard<std::mutex> lk(g_mu);
    memset(a, 0, sizeof(*a));
    if (g_dev.count(p)) {
        a->type = hipMemoryTypeDevice;
        return hipSuccess;
    }
    return hipErrorInvalidValue;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind)
{
    memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t)
{
    for (size_t r = 0; r < height; ++r) memcpy((char *)d + r * dpitch, (const char *)s + r * spitch, width);
    return hipSuccess;
}
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t)
{
    memcpy(d, s, n);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t)
{
    memset(d, v, n);
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int)
{
    *s = (hipStream_t)malloc(8);
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s)
{
    free((void *)s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e)
{
    *e = (hipEvent_t)malloc(8);
    return hipSuccess;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e)
{
    free((void *)e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t)
{
    *ms = 0.0f;
    return hipSuccess;
}
}
