// Store-flavour probe: plain vs nt vs sc1 vs sc0 sc1 global_store_dwordx4 on a blocked streaming fill.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
template <int FLAVOUR>
__global__ void fill_blocked(uint4 *dst, size_t n16, size_t chunk16) {
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const unsigned lane = threadIdx.x & 63;
    const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    v4u v = {1, 2, 3, 4};
    for (size_t c = wave; c * chunk16 < n16; c += nwaves) {
        const size_t base = c * chunk16;
        for (size_t i = lane; i < chunk16 && base + i < n16; i += 64) {
            v4u *p = reinterpret_cast<v4u *>(dst + base + i);
            if (FLAVOUR == 0) *p = v;
            else if (FLAVOUR == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
            else if (FLAVOUR == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
            else if (FLAVOUR == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
        }
    }
}
int main() {
    const size_t bytes = 9771155456ull, n16 = bytes / 16;
    uint4 *d; if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[5] = {"plain", "nt", "sc1", "sc0 sc1", "sc0"};
    for (int rep = 0; rep < 2; rep++)
    for (int fl = 0; fl < 5; fl++) {
        std::vector<float> ms;
        for (int it = 0; it < 8; it++) {
            hipEventRecord(e0);
            if (fl == 0) hipLaunchKernelGGL(fill_blocked<0>, dim3(256 * 16), dim3(64), 0, 0, d, n16, (size_t)64 * 64);
            if (fl == 1) hipLaunchKernelGGL(fill_blocked<1>, dim3(256 * 16), dim3(64), 0, 0, d, n16, (size_t)64 * 64);
            if (fl == 2) hipLaunchKernelGGL(fill_blocked<2>, dim3(256 * 16), dim3(64), 0, 0, d, n16, (size_t)64 * 64);
            if (fl == 3) hipLaunchKernelGGL(fill_blocked<3>, dim3(256 * 16), dim3(64), 0, 0, d, n16, (size_t)64 * 64);
            if (fl == 4) hipLaunchKernelGGL(fill_blocked<4>, dim3(256 * 16), dim3(64), 0, 0, d, n16, (size_t)64 * 64);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float t; hipEventElapsedTime(&t, e0, e1); if (it >= 2) ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        printf("%-8s median %.3f ms  %.0f GB/s\n", names[fl], ms[ms.size() / 2], bytes / 1e6 / ms[ms.size() / 2]);
    }
    return 0;
}
