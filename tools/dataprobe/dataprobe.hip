// Does HBM write throughput on this part depend on the DATA?  (tuning aid; not part of the product)
//   hipcc --offload-arch=gfx950 -O3 dataprobe.hip -o dataprobe
// One streaming-store kernel (every wave writes its own 64 KiB chunks, 16 bytes per lane and instruction -- the
// best pure-write pattern found, hsw_fill_kernel), 9.77 GB like a 4,096-block gate stream, with four kinds of
// 32-byte cells: all zero; a small integer in the low 8 bytes and 24 zero bytes (what canonical cells look like);
// eight pseudo-random 32-bit limbs (what Montgomery cells look like); the same random cell everywhere.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned mix(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int KIND>
__global__ __launch_bounds__(64) void fill(v4u *dst, size_t n16, unsigned chunks) {
    const unsigned lane = threadIdx.x;
    for (size_t c = blockIdx.x; c < chunks; c += gridDim.x) {
        v4u *p = dst + c * 4096;                              // 64 KiB = 4,096 pieces of 16 bytes
#pragma unroll 4
        for (unsigned i = lane; i < 4096u; i += 64u) {
            const size_t idx = c * 4096 + i;
            if (idx >= n16) break;
            const unsigned s = (unsigned)idx;
            v4u v;
            if (KIND == 0) v = v4u{0, 0, 0, 0};
            else if (KIND == 1) v = (i & 1u) ? v4u{0, 0, 0, 0} : v4u{mix(s) & 0xffffu, 0, 0, 0};     // low half: a 16-bit value; high half: zeros
            else if (KIND == 2) v = v4u{mix(s), mix(s + 0x9e3779b9u), mix(s ^ 0x85ebca6bu), mix(s * 3u + 1u)};
            else v = v4u{0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u};
            p[i] = v;
        }
    }
}
int main() {
    const size_t bytes = 9771679744ull, n16 = bytes / 16;
    const unsigned chunks = (unsigned)((n16 + 4095) / 4096);
    v4u *d;
    if (hipMalloc(&d, (size_t)chunks * 65536) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[4] = {"all-zero cells", "16-bit value + 30 zero bytes per cell (canonical-like)", "eight random 32-bit limbs per cell (Montgomery-like)", "one fixed 16-byte pattern everywhere"};
    for (int rep = 0; rep < 3; rep++)
        for (int k = 0; k < 4; k++) {
            std::vector<float> ms;
            for (int it = 0; it < 9; it++) {
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(fill<0>, dim3(4096), dim3(64), 0, 0, d, n16, chunks);
                else if (k == 1) hipLaunchKernelGGL(fill<1>, dim3(4096), dim3(64), 0, 0, d, n16, chunks);
                else if (k == 2) hipLaunchKernelGGL(fill<2>, dim3(4096), dim3(64), 0, 0, d, n16, chunks);
                else hipLaunchKernelGGL(fill<3>, dim3(4096), dim3(64), 0, 0, d, n16, chunks);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                if (it >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("%-62s %.3f ms  %.0f GB/s (best %.0f)\n", names[k], ms[ms.size() / 2], bytes / 1e6 / ms[ms.size() / 2], bytes / 1e6 / ms[0]);
        }
    return 0;
}
