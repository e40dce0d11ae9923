// A stand-in for the HIP runtime, linked INTO the sanitizer test executable (tests/test_host_sanitizers.py):
// "device" memory is plain heap memory, copies are memmove, kernel launches do nothing.  It lets the host side of
// libhsw -- engine and gadget lifetimes, staging buffers, every hipMemcpy the library issues, the bookkeeping
// of pinned allocations -- run under AddressSanitizer / UBSan on a machine without a GPU: a copy that is longer
// than the allocation it reads or writes, a free of something never allocated, a use after destroy are then
// ordinary heap errors the sanitizer reports.  Test infrastructure only; the product links libamdhip64.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

namespace {
std::map<void *, size_t> g_device;        // hipMalloc
std::map<void *, size_t> g_pinned;        // hipHostMalloc
std::set<void *> g_events, g_streams;
int g_launches = 0;
int g_current_device = 0;

bool inside(const std::map<void *, size_t> &m, const void *p) {
    auto it = m.upper_bound(const_cast<void *>(p));
    if (it == m.begin()) return false;
    --it;
    return static_cast<const char *>(p) < static_cast<const char *>(it->first) + it->second;
}
[[noreturn]] void die(const char *what) {
    std::fprintf(stderr, "hip_stub: %s\n", what);
    std::abort();
}
}  // namespace

extern "C" {

// what the lifecycle test asks the stub
int hip_stub_launches() { return g_launches; }
size_t hip_stub_live_device_allocations() { return g_device.size(); }
size_t hip_stub_live_pinned_allocations() { return g_pinned.size(); }
size_t hip_stub_live_events() { return g_events.size(); }

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = g_current_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d != 0) return hipErrorInvalidDevice; g_current_device = d; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "hip_stub error"; }

hipError_t hipMalloc(void **p, size_t bytes) {
    *p = std::malloc(bytes ? bytes : 1);
    if (!*p) return hipErrorOutOfMemory;
    g_device[*p] = bytes;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    if (!g_device.erase(p)) die("hipFree of a pointer hipMalloc did not return");
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned int) {
    *p = std::malloc(bytes ? bytes : 1);
    if (!*p) return hipErrorOutOfMemory;
    g_pinned[*p] = bytes;
    return hipSuccess;
}
hipError_t hipHostFree(void *p) {
    if (!p) return hipSuccess;
    if (!g_pinned.erase(p)) die("hipHostFree of a pointer hipHostMalloc did not return");
    std::free(p);
    return hipSuccess;
}
// the real runtime knows which host memory is mapped: anything else is an error
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned int) {
    if (!inside(g_pinned, host)) { *dev = nullptr; return hipErrorInvalidValue; }
    *dev = host;
    return hipSuccess;
}
hipError_t hipHostRegister(void *, size_t, unsigned int) { die("the library must not register caller memory"); }

hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(dst, src, n); return hipSuccess; }
hipError_t hipMemset(void *dst, int v, size_t n) { std::memset(dst, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *dst, int v, size_t n, hipStream_t) { std::memset(dst, v, n); return hipSuccess; }

hipError_t hipStreamCreate(hipStream_t *s) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); g_streams.insert(*s); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) { return hipStreamCreate(s); }
hipError_t hipStreamDestroy(hipStream_t s) {
    if (!g_streams.erase(s)) die("hipStreamDestroy of an unknown stream");
    std::free(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) {
    if (s && !g_streams.count(s)) die("hipStreamSynchronize on a destroyed stream");
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t ev, unsigned int) {
    if (!g_events.count(ev)) die("hipStreamWaitEvent on a destroyed event");
    return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t *e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); g_events.insert(*e); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned int) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) {
    if (!g_events.erase(e)) die("hipEventDestroy of an unknown event");
    std::free(e);
    return hipSuccess;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { if (!g_events.count(e)) die("hipEventRecord on a destroyed event"); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { if (!g_events.count(e)) die("hipEventSynchronize on a destroyed event"); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
    if (!g_events.count(a) || !g_events.count(b)) die("hipEventElapsedTime on a destroyed event");
    *ms = 0.0f;
    return hipSuccess;
}

// kernel launches: the compiler's launch stubs pop the configuration and call hipLaunchKernel
static dim3 g_grid, g_block;
static size_t g_shmem;
static hipStream_t g_stream;
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    g_grid = grid; g_block = block; g_shmem = shmem; g_stream = stream;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream) {
    *grid = g_grid; *block = g_block; *shmem = g_shmem; *stream = g_stream;
    return hipSuccess;
}
// virtual memory management (hsw_devmem.cpp): a reserved range is heap memory, physical handles are counted
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipMemGetAllocationGranularity(size_t *g, const hipMemAllocationProp *, hipMemAllocationGranularity_flags) { *g = 4096; return hipSuccess; }
hipError_t hipMemAddressReserve(void **p, size_t bytes, size_t, void *, unsigned long long) { return hipMalloc(p, bytes); }
hipError_t hipMemAddressFree(void *p, size_t bytes) {
    auto it = g_device.find(p);
    if (it == g_device.end() || it->second != bytes) die("hipMemAddressFree of a range that was not reserved like this");
    return hipFree(p);
}
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *h, size_t bytes, const hipMemAllocationProp *, unsigned long long) {
    void *q = std::malloc(16);
    g_device[q] = bytes;                                   // a live physical allocation: leaks show up like any other
    *h = reinterpret_cast<hipMemGenericAllocationHandle_t>(q);
    return hipSuccess;
}
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) { return hipFree(reinterpret_cast<void *>(h)); }
hipError_t hipMemMap(void *p, size_t bytes, size_t, hipMemGenericAllocationHandle_t, unsigned long long) {
    if (!inside(g_device, p) || !inside(g_device, static_cast<char *>(p) + bytes - 1)) die("hipMemMap outside a reserved range");
    return hipSuccess;
}
hipError_t hipMemUnmap(void *p, size_t) { if (!inside(g_device, p)) die("hipMemUnmap outside a reserved range"); return hipSuccess; }
hipError_t hipMemSetAccess(void *p, size_t, const hipMemAccessDesc *, size_t) { if (!inside(g_device, p)) die("hipMemSetAccess outside a reserved range"); return hipSuccess; }

hipError_t hipLaunchKernel(const void *, dim3 grid, dim3 block, void **, size_t, hipStream_t) {
    if (grid.x == 0 || block.x == 0 || block.x * block.y * block.z > 1024) return hipErrorInvalidConfiguration;
    g_launches++;
    return hipSuccess;
}
// code-object registration emitted for every translation unit with kernels
void **__hipRegisterFatBinary(const void *) { static void *handle[1]; return handle; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned int, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}
void __hipRegisterManagedVar(void *, void **, void *, const char *, size_t, unsigned) {}

}  // extern "C"
