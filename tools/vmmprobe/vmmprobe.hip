// Does it matter how the output buffers are BACKED?  (tuning aid)
//   hipcc --offload-arch=gfx950 -O3 -I../../include vmmprobe.hip -L../../halo2-dynamic-sha256_amd -lhsw -o vmmprobe
// The 4,096-block witness launch (through the C ABI) with its gate stream in (a) a plain hipMalloc buffer and (b)
// one contiguous VIRTUAL range backed by many separate physical allocations of `chunk` bytes (HIP virtual memory
// management: hipMemAddressReserve / hipMemCreate / hipMemMap); chip columns in plain buffers.  Follow-up to
// profiles/r03_placement_probe.log: two write streams in two allocations run at 6.26 TB/s, in one at 5.7.
// CAUTION: with two more configurations in the list (8 and 16 GiB pieces: two more unmap / re-create cycles) this
// probe ended in 'Memory access fault by GPU' in three of three processes, at the first launch after a new range
// had been created; as it is (five cycles) it ran clean in eight.  Hence: ranges are allocated once and kept.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "hsw.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define HK(x) do { int r_ = (x); if (r_ != HSW_OK) { printf("%s: %s\n", #x, hsw_strerror(r_)); exit(1); } } while (0)

struct Vmm { void *ptr = nullptr; size_t size = 0; std::vector<hipMemGenericAllocationHandle_t> handles; };
static Vmm vmm_alloc(size_t bytes, size_t chunk, int device) {
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    chunk = ((chunk + gran - 1) / gran) * gran;
    Vmm v;
    v.size = ((bytes + chunk - 1) / chunk) * chunk;
    CK(hipMemAddressReserve(&v.ptr, v.size, 0, nullptr, 0));
    for (size_t off = 0; off < v.size; off += chunk) {
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, chunk, &prop, 0));
        CK(hipMemMap((char *)v.ptr + off, chunk, 0, h, 0));
        v.handles.push_back(h);
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(v.ptr, v.size, &acc, 1));
    return v;
}
static void vmm_free(Vmm &v) {
    CK(hipMemUnmap(v.ptr, v.size));
    for (auto h : v.handles) CK(hipMemRelease(h));
    CK(hipMemAddressFree(v.ptr, v.size));
}

int main() {
    const size_t n = 4096;
    hsw_engine *e = nullptr;
    HK(hsw_engine_create(0, nullptr, 8, 2, &e));
    hsw_shape s;
    HK(hsw_engine_shape(e, &s));
    const size_t gate_bytes = n * (size_t)s.gate_cells_per_block * 32, rows = (size_t)hsw_chip_rows(&s, 0, n), col_bytes = 2 * rows * 32;
    uint8_t *d_blocks; uint32_t *d_pre, *d_next;
    CK(hipMalloc(&d_blocks, n * 64)); CK(hipMalloc(&d_pre, n * 32)); CK(hipMalloc(&d_next, n * 32));
    std::vector<uint8_t> hb(n * 64);
    srand(3);
    for (auto &b : hb) b = (uint8_t)rand();
    CK(hipMemcpy(d_blocks, hb.data(), hb.size(), hipMemcpyHostToDevice));
    CK(hipMemset(d_pre, 0x5a, n * 32));
    HK(hsw_set_timing(e, 1));
    auto run = [&](void *gate, void *cd, void *cs) {
        std::vector<float> ms;
        for (int i = 0; i < 9; i++) {
            HK(hsw_witness_blocks(e, d_blocks, d_pre, n, 0, gate, cd, cs, rows, d_next, 0));
            float t; HK(hsw_last_kernel_ms(e, &t));
            if (i >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        return ms[ms.size() / 2];
    };
    // chip columns: four plain candidates
    void *cd[4], *cs[4];
    for (int k = 0; k < 4; k++) { CK(hipMalloc(&cd[k], col_bytes)); CK(hipMalloc(&cs[k], col_bytes)); }
    void *plain[2];
    for (int k = 0; k < 2; k++) CK(hipMalloc(&plain[k], gate_bytes));
    for (int k = 0; k < 2; k++) {
        printf("gate stream in plain hipMalloc buffer %d, chip columns in candidates 0..3:", k);
        for (int j = 0; j < 4; j++) printf("  %.3f", run(plain[k], cd[j], cs[j]));
        printf(" ms\n");
    }
    const size_t chunks[] = {(size_t)2 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30};
    for (size_t chunk : chunks) {
        Vmm v = vmm_alloc(gate_bytes, chunk, 0);
        printf("gate stream in ONE virtual range of %zu physical allocations of %zu MiB, chip columns in candidates 0..3:", v.handles.size(), chunk >> 20);
        for (int j = 0; j < 4; j++) printf("  %.3f", run(v.ptr, cd[j], cs[j]));
        printf(" ms\n");
        fflush(stdout);
        vmm_free(v);
    }
    // ... and the chip columns in such a range too (every column pair in its own allocations)
    {
        Vmm g = vmm_alloc(gate_bytes, (size_t)256 << 20, 0), a = vmm_alloc(col_bytes, (size_t)32 << 20, 0), b = vmm_alloc(col_bytes, (size_t)32 << 20, 0);
        printf("gate stream in 256 MiB allocations, chip columns in 32 MiB allocations: %.3f ms\n", run(g.ptr, a.ptr, b.ptr));
        vmm_free(g); vmm_free(a); vmm_free(b);
    }
    hsw_engine_destroy(e);
    return 0;
}
