// Host-visible floor of "launch one kernel, wait for it" (tuning aid; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 floor.hip -o floor
// Times, on the host clock, an empty 600-workgroup kernel launched and waited for in four ways:
// hipStreamSynchronize (default device flags), the same after hipSetDeviceFlags(hipDeviceScheduleSpin),
// hipEventSynchronize on a recorded event, and a flag the LAST workgroup writes into pinned host memory
// (device-scope counter, system-scope store) that the host spins on.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 1000) *p = 1; }

__global__ void flag_kernel(unsigned *counter, volatile unsigned *host_flag, unsigned seq, int spin) {
    for (int i = 0; i < spin; i++) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned done = atomicAdd(counter, 1u) + 1u;
        if (done == gridDim.x) {
            *counter = 0;
            __threadfence_system();
            *host_flag = seq;
        }
    }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <class F>
static void timeit(const char *name, F f) {
    std::vector<double> v;
    for (int i = 0; i < 320; i++) { const double t0 = now_us(); f(i); const double t1 = now_us(); if (i >= 20) v.push_back(t1 - t0); }
    std::sort(v.begin(), v.end());
    printf("%-72s median %5.1f us  min %5.1f  p90 %5.1f\n", name, v[v.size() / 2], v[0], v[v.size() * 9 / 10]);
}

int main(int argc, char **argv) {
    const int spin_flag = argc > 1 ? atoi(argv[1]) : 0;
    if (spin_flag) CK(hipSetDeviceFlags(hipDeviceScheduleSpin));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    unsigned *d_counter; CK(hipMalloc(&d_counter, 4)); CK(hipMemset(d_counter, 0, 4));
    unsigned *h_flag, *d_flag; CK(hipHostMalloc(&h_flag, 64, hipHostMallocMapped)); *h_flag = 0;
    CK(hipHostGetDevicePointer((void **)&d_flag, h_flag, 0));
    printf("device flags: %s\n", spin_flag ? "hipDeviceScheduleSpin" : "default");
    timeit("empty kernel (600 x 64) + hipStreamSynchronize", [&](int) {
        hipLaunchKernelGGL(empty_kernel, dim3(600), dim3(64), 0, s, nullptr); CK(hipStreamSynchronize(s)); });
    timeit("empty kernel + hipEventRecord + hipEventSynchronize", [&](int) {
        hipLaunchKernelGGL(empty_kernel, dim3(600), dim3(64), 0, s, nullptr); CK(hipEventRecord(ev, s)); CK(hipEventSynchronize(ev)); });
    timeit("empty kernel + hipStreamQuery spin", [&](int) {
        hipLaunchKernelGGL(empty_kernel, dim3(600), dim3(64), 0, s, nullptr); while (hipStreamQuery(s) == hipErrorNotReady) {} });
    timeit("flag kernel (600 x 64), host spins on pinned flag", [&](int i) {
        hipLaunchKernelGGL(flag_kernel, dim3(600), dim3(64), 0, s, d_counter, d_flag, (unsigned)i + 1u, 0);
        while (*(volatile unsigned *)h_flag != (unsigned)i + 1u) {} });
    CK(hipStreamSynchronize(s));
    timeit("flag kernel, ~10 us of s_sleep per wave, host spins on pinned flag", [&](int i) {
        hipLaunchKernelGGL(flag_kernel, dim3(600), dim3(64), 0, s, d_counter, d_flag, (unsigned)i + 1000u, 380);
        while (*(volatile unsigned *)h_flag != (unsigned)i + 1000u) {} });
    CK(hipStreamSynchronize(s));
    timeit("same kernel (~10 us) + hipStreamSynchronize", [&](int i) {
        hipLaunchKernelGGL(flag_kernel, dim3(600), dim3(64), 0, s, d_counter, d_flag, (unsigned)i + 5000u, 380); CK(hipStreamSynchronize(s)); });
    timeit("launch call alone (kernel left running; drained outside the timing)", [&](int i) {
        hipLaunchKernelGGL(empty_kernel, dim3(600), dim3(64), 0, s, nullptr); });
    CK(hipStreamSynchronize(s));
    return 0;
}
