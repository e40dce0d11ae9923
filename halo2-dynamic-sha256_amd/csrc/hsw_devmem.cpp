// hsw_devmem.cpp -- device memory for witness streams: hsw_device_alloc / hsw_device_free (include/hsw.h).
//
// One contiguous VIRTUAL range backed by separate physical allocations of (up to) `chunk_bytes` each, through HIP's
// virtual memory management.  Why: on MI355X the write rate of the witness launch depends on where its output
// buffers sit, and a 9 GB stream backed by 4 GiB physical allocations ran 2-4 % faster than the same
// stream in one plain hipMalloc buffer in every process tried (tools/vmmprobe: 1.58-1.63 against 1.64-1.65 ms per
// 4,096 blocks), and was far less sensitive to where the chip columns are.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/hsw.h"
#include "hsw_nounwind.hpp"

namespace {

struct Range {
    size_t size = 0;
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    bool mapped = false;
};
std::mutex g_mu;
std::map<void *, Range> g_ranges;

// undo whatever of a range exists (any order of failure in hsw_device_alloc)
void release(void *ptr, Range &r) {
    if (r.mapped) (void)hipMemUnmap(ptr, r.size);
    for (auto h : r.handles) (void)hipMemRelease(h);
    if (ptr) (void)hipMemAddressFree(ptr, r.size);
}

}  // namespace

extern "C" {

int hsw_device_alloc(int device, size_t bytes, size_t chunk_bytes, void **out) try {
    if (!out || bytes == 0) return HSW_ERR_INVALID_ARG;
    *out = nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return HSW_ERR_NO_DEVICE;
    struct Back { int prev; ~Back() { if (prev >= 0) (void)hipSetDevice(prev); } } back{prev};
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) return HSW_ERR_HIP;
    if (chunk_bytes == 0) chunk_bytes = (size_t)4 << 30;
    const size_t chunk = (chunk_bytes + gran - 1) / gran * gran;
    Range r;
    r.device = device;
    r.size = (bytes + gran - 1) / gran * gran;
    void *ptr = nullptr;
    hipError_t he = hipMemAddressReserve(&ptr, r.size, 0, nullptr, 0);
    if (he != hipSuccess) return he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP;
    for (size_t off = 0; off < r.size && he == hipSuccess; off += chunk) {
        const size_t len = r.size - off < chunk ? r.size - off : chunk;
        hipMemGenericAllocationHandle_t h;
        he = hipMemCreate(&h, len, &prop, 0);
        if (he != hipSuccess) break;
        r.handles.push_back(h);
        he = hipMemMap(static_cast<char *>(ptr) + off, len, 0, h, 0);
        if (he == hipSuccess) r.mapped = true;          // (hipMemUnmap of the whole range undoes every piece mapped so far)
    }
    if (he == hipSuccess) {
        hipMemAccessDesc acc{};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        he = hipMemSetAccess(ptr, r.size, &acc, 1);
    }
    if (he != hipSuccess) {
        release(ptr, r);
        (void)hipGetLastError();
        return he == hipErrorOutOfMemory ? HSW_ERR_NOMEM : HSW_ERR_HIP;
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_ranges[ptr] = r;
    }
    *out = ptr;
    return HSW_OK;
} HSW_NO_UNWIND

int hsw_device_free(void *ptr) try {
    if (!ptr) return HSW_OK;
    Range r;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ranges.find(ptr);
        if (it == g_ranges.end()) return HSW_ERR_INVALID_ARG;      // not from hsw_device_alloc (or freed already)
        r = it->second;
        g_ranges.erase(it);
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(r.device);
    (void)hipDeviceSynchronize();                                   // nothing may still write the range
    release(ptr, r);
    if (prev >= 0) (void)hipSetDevice(prev);
    return HSW_OK;
} HSW_NO_UNWIND

}  // extern "C"
