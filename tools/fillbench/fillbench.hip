// Write-bandwidth probe (tuning aid, not product): which store pattern does
// the MI355X memory system like best for a pure 9.77 GB write stream?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// A: grid-stride, consecutive 16 B per lane (1 KiB per wave-instruction, next one stride apart)
__global__ void fill_gridstride(uint4 *dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = v;
}
// B: blocked: every wave owns one contiguous chunk of `chunk16` pieces and streams through it
__global__ void fill_blocked(uint4 *dst, size_t n16, size_t chunk16) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t c = wave; c * chunk16 < n16; c += nwaves) {
        const size_t base = c * chunk16;
        for (size_t i = lane; i < chunk16 && base + i < n16; i += 64) dst[base + i] = v;
    }
}
// C: like the expand kernel: every wave owns a 2.1 MB region and writes it as R rows x runs of `run16`
//    pieces, rows `rowstride16` apart (tile transposition order)
__global__ void fill_rows(uint4 *dst, size_t region16, int rows, size_t rowstride16, int run16) {
    const size_t wave = blockIdx.x;          // 64 threads per block
    const unsigned lane = threadIdx.x;
    uint4 *base = dst + wave * region16;
    uint4 v = make_uint4(1, 2, 3, 4);
    for (size_t seg = 0; seg + run16 <= rowstride16; seg += run16)
        for (int r = 0; r < rows; r++)
            for (int i = lane; i < run16; i += 64) base[(size_t)r * rowstride16 + seg + i] = v;
}

int main() {
    const size_t bytes = 9771155456ull;      // 4096 blocks x 74,548 cells x 32 B
    const size_t n16 = bytes / 16;
    uint4 *d;
    CK(hipMalloc(&d, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto launch) {
        std::vector<float> ms;
        for (int it = 0; it < 8; it++) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); if (it >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-46s median %.3f ms  %.0f GB/s\n", name, ms[ms.size() / 2], bytes / 1e6 / ms[ms.size() / 2]);
        return 0;
    };
    for (int blocks_per_cu : {2, 4, 8}) {
        char nm[96]; snprintf(nm, sizeof nm, "gridstride 256thr x %d blocks/CU", blocks_per_cu);
        timeit(nm, [&] { hipLaunchKernelGGL(fill_gridstride, dim3(256 * blocks_per_cu), dim3(256), 0, 0, d, n16); });
    }
    for (size_t chunk_kb : {4, 16, 64, 256, 2048}) {
        for (int waves_per_cu : {8, 16, 32}) {
            char nm[96]; snprintf(nm, sizeof nm, "blocked chunk %zu KiB, %d waves/CU", chunk_kb, waves_per_cu);
            timeit(nm, [&] { hipLaunchKernelGGL(fill_blocked, dim3(256 * waves_per_cu), dim3(64), 0, 0, d, n16, chunk_kb * 64); });
        }
    }
    // expand-like: 4096 regions of 2,385,536 B; rounds-like rows of 24,320 B
    {
        const size_t region16 = 2385536 / 16;
        for (int run_b : {1024, 2048, 4096}) {
            for (int rows : {64, 32, 16}) {
                char nm[96]; snprintf(nm, sizeof nm, "rows: %d rows x %d B runs (4096 waves)", rows, run_b);
                // region = rows * rowstride; keep bytes identical: rowstride = region / rows
                const size_t rowstride16 = (region16 / rows / (run_b / 16)) * (run_b / 16);
                timeit(nm, [&] { hipLaunchKernelGGL(fill_rows, dim3(4096), dim3(64), 0, 0, d, region16, rows, rowstride16, run_b / 16); });
            }
        }
    }
    return 0;
}
