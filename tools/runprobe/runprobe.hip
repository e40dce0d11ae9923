// Write-pattern probe for the 32-byte-cell tile question: 4,096 block regions of 64 rows x 34,816
// bytes; a wave owns R consecutive rows of one region and writes them in "flushes" of RUN contiguous bytes per row
// (what a [R][RUN/32] tile of 32-byte cells, or a [R][RUN/8] tile of u64 expanded at write-out, does).  LDS bytes
// per workgroup limit the occupancy like the real tiles would.  Prints ms and GB/s per (R, RUN, LDS).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void pattern(char *dst, unsigned rows_per_wave, unsigned run, unsigned row_bytes, unsigned spin) {
    extern __shared__ char lds[];
    const unsigned lane = threadIdx.x, parts = 64u / rows_per_wave;
    const size_t blk = blockIdx.x / parts, part = blockIdx.x % parts;
    char *base = dst + blk * (size_t)64 * row_bytes + (size_t)part * rows_per_wave * row_bytes;
    v4u v = {lane, 2, 3, 4};
    if (spin && lane == 999) lds[0] = 1;
    const unsigned lanes_per_row = run / 16u;             // 16-byte pieces per row and flush
    for (unsigned off = 0; off < row_bytes; off += run) {
        // emulate the emission between two flushes: `spin` dependent VALU instructions
        unsigned x = lane;
        for (unsigned s = 0; s < spin; s++) x = (x ^ s) + 0x9e3779b9u;
        v.y = x;
        if (lanes_per_row >= 64u) {
            for (unsigned r = 0; r < rows_per_wave; r++)
                for (unsigned q = lane; q < lanes_per_row; q += 64)
                    *reinterpret_cast<v4u *>(base + (size_t)r * row_bytes + off + q * 16u) = v;
        } else {
            const unsigned rpi = 64u / lanes_per_row;     // rows per store instruction
            for (unsigned r0 = 0; r0 < rows_per_wave; r0 += rpi) {
                const unsigned r = r0 + lane / lanes_per_row, q = lane % lanes_per_row;
                *reinterpret_cast<v4u *>(base + (size_t)r * row_bytes + off + q * 16u) = v;
            }
        }
    }
}
int main(int argc, char **argv) {
    const unsigned row_bytes = 34816, nblk = 4096;
    const size_t bytes = (size_t)nblk * 64 * row_bytes;
    char *d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { unsigned rows, run, lds, spin; };
    std::vector<Cfg> cfgs;
    // spin = dependent plain VALU instructions between two flushes, scaled with the run so that every configuration
    // does the same total arithmetic per block (~ what an emitter does between write-outs)
    for (unsigned per256 : {0u, 150u, 400u})
        for (Cfg c : {Cfg{32, 2048, 19000, 0}, Cfg{64, 1024, 19000, 0}, Cfg{64, 512, 19000, 0}, Cfg{64, 256, 19000, 0},
                      Cfg{64, 512, 37000, 0}, Cfg{64, 256, 37000, 0}, Cfg{32, 2048, 37000, 0}, Cfg{16, 4096, 19000, 0}}) {
            c.spin = per256 * (c.run / 256u) * (c.rows / 16u) / 4u; cfgs.push_back(c);
        }
    for (int rep = 0; rep < 2; rep++)
        for (const Cfg &c : cfgs) {
            if (row_bytes % c.run) continue;
            hipFuncSetAttribute(reinterpret_cast<const void *>(pattern), hipFuncAttributeMaxDynamicSharedMemorySize, 80000);
            std::vector<float> ms;
            for (int it = 0; it < 7; it++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(pattern, dim3(nblk * (64 / c.rows)), dim3(64), c.lds, 0, d, c.rows, c.run, row_bytes, c.spin);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float t; hipEventElapsedTime(&t, e0, e1);
                if (it >= 2) ms.push_back(t);
            }
            std::sort(ms.begin(), ms.end());
            printf("rows %2u run %4u B  lds %5u B  spin %3u : %.3f ms  %.0f GB/s\n", c.rows, c.run, c.lds, c.spin, ms[ms.size() / 2],
                   bytes / 1e6 / ms[ms.size() / 2]);
        }
    return 0;
}
