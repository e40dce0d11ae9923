// Do two write streams in two allocations disturb each other, and does it depend on WHICH two?  (tuning aid)
//   hipcc --offload-arch=gfx950 -O3 pairprobe.hip -o pairprobe
// The witness launch runs 8 % faster or slower depending on the pair of allocations of its gate stream (8.69 GB) and
// its chip columns (1.08 GB) (profiles/r03_placement_probe.log).  This probe takes the kernel out of the question:
// one pure-store kernel, workgroup i streams 64 KiB chunks of buffer A, every ninth workgroup chunks of buffer B
// instead (B gets 1/9 of the bytes, like the chip columns), for every pair of NA big and NB small allocations.
// mode 1: B is written in 1 KiB pieces scattered with a 65,920-byte stride (a chip column's row pitch per block)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(64) void two_streams(v4u *a, size_t a_chunks, v4u *b, size_t b_chunks, int mode) {
    const unsigned lane = threadIdx.x;
    const v4u v = {lane, blockIdx.x, 3u, 4u};
    size_t ia = blockIdx.x - blockIdx.x / 9, ib = blockIdx.x / 9;          // this workgroup's first chunk of A / of B
    const bool for_b = blockIdx.x % 9 == 8;
    const size_t step_a = gridDim.x - gridDim.x / 9, step_b = gridDim.x / 9;
    if (!for_b) {
        for (size_t c = ia; c < a_chunks; c += step_a) {
            v4u *p = a + c * 4096;
#pragma unroll 4
            for (unsigned i = lane; i < 4096u; i += 64u) p[i] = v;
        }
    } else {
        for (size_t c = ib; c < b_chunks; c += step_b) {
            if (mode == 0) {
                v4u *p = b + c * 4096;
#pragma unroll 4
                for (unsigned i = lane; i < 4096u; i += 64u) p[i] = v;
            } else {
                // 64 pieces of 1 KiB, 65,920 bytes (4,120 pieces of 16 B) apart, wrapped into the buffer
                for (unsigned k = 0; k < 64u; k++) {
                    const size_t piece = (c * 64 + (size_t)k * 4120u * 16u) % (b_chunks * 64);      // in 1 KiB units
                    b[piece * 64 + lane] = v;
                }
            }
        }
    }
}
int main(int argc, char **argv) {
    const int NA = argc > 1 ? atoi(argv[1]) : 6, NB = argc > 2 ? atoi(argv[2]) : 6;
    const size_t a_bytes = 8691122176ull, b_bytes = 1080033280ull;
    const size_t a_chunks = a_bytes / 65536, b_chunks = b_bytes / 65536;
    std::vector<v4u *> A(NA), B(NB);
    for (int j = 0; j < NB; j++) CK(hipMalloc(&B[j], b_chunks * 65536));
    for (int k = 0; k < NA; k++) CK(hipMalloc(&A[k], a_chunks * 65536));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; mode++) {
        printf("mode %d (%s): ms per launch, rows = big buffer, columns = small buffer\n", mode, mode ? "B scattered in 1 KiB pieces" : "B streamed");
        for (int k = 0; k < NA; k++) {
            printf("  A%-2d", k);
            for (int j = 0; j < NB; j++) {
                std::vector<float> ms;
                for (int it = 0; it < 7; it++) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(two_streams, dim3(4608), dim3(64), 0, 0, A[k], a_chunks, B[j], b_chunks, mode);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float t; CK(hipEventElapsedTime(&t, e0, e1));
                    if (it >= 2) ms.push_back(t);
                }
                std::sort(ms.begin(), ms.end());
                printf("  %.3f", ms[ms.size() / 2]);
            }
            printf("\n");
        }
    }
    return 0;
}
