// Issue cost of the VALU instructions the Montgomery write-out is made of, on gfx950: each kernel runs
// 8 independent chains of one instruction per lane, 4 waves per SIMD, so the time per instruction is the
// issue cost (cycles per wave-instruction per SIMD), not the latency.
//   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32; typedef uint64_t u64;
#define ITER 4096
#define CHAINS 8

// inline asm keeps the compiler from folding the loops
#define KERNEL32(NAME, ASM)                                                         \
__global__ void NAME(u32 *out, u32 seed) {                                          \
    u32 x[CHAINS]; u32 k = seed | 1u;                                               \
    for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 2654435761u + c + seed;   \
    for (int it = 0; it < ITER; it++) {                                             \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                          \
            asm volatile(ASM : "+v"(x[c]) : "v"(k) : "vcc");                        \
    }                                                                               \
    u32 acc = 0;                                                                    \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) acc ^= x[c];                 \
    if (acc == 0x12345u) out[threadIdx.x] = acc;                                    \
}
#define KERNEL64(NAME, ASM)                                                         \
__global__ void NAME(u32 *out, u32 seed) {                                          \
    u64 x[CHAINS]; u32 k = seed | 1u;                                               \
    for (int c = 0; c < CHAINS; c++) x[c] = threadIdx.x * 2654435761ull + c + seed; \
    for (int it = 0; it < ITER; it++) {                                             \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++)                          \
            asm volatile(ASM : "+v"(x[c]) : "v"(k) : "vcc");                        \
    }                                                                               \
    u32 acc = 0;                                                                    \
    _Pragma("unroll") for (int c = 0; c < CHAINS; c++) acc ^= (u32)x[c];            \
    if (acc == 0x12345u) out[threadIdx.x] = acc;                                    \
}
KERNEL32(k_add, "v_add_u32 %0, %0, %1")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %0")
KERNEL32(k_addco, "v_add_co_u32 %0, vcc, %0, %1")
KERNEL32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc")
KERNEL32(k_addc_s, "v_addc_co_u32 %0, s[10:11], %0, %1, s[10:11]")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_cndmask_s, "v_cndmask_b32_e64 %0, %0, %1, s[10:11]")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %0")
KERNEL32(k_cvt_f32, "v_cvt_f32_u32 %0, %0")
KERNEL32(k_fma32, "v_fma_f32 %0, %0, %1, %0")
KERNEL64(k_mad64, "v_mad_u64_u32 %0, vcc, %1, %1, %0")
KERNEL64(k_fma64, "v_fma_f64 %0, %0, %0, %0")
KERNEL64(k_mul64f, "v_mul_f64 %0, %0, %0")
KERNEL64(k_add64, "v_lshl_add_u64 %0, %0, 0, %0")
KERNEL64(k_cvt_f64, "v_cvt_f64_u32 %0, %1")

__global__ void k_lat_mad64(u32 *out, u32 seed) {
    u64 x = threadIdx.x + seed; u32 k = seed | 1u;
    for (int it = 0; it < ITER * 8; it++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(x) : "v"(k) : "vcc");
    if ((u32)x == 0x12345u) out[threadIdx.x] = (u32)x;
}
__global__ void k_lat_mul_lo(u32 *out, u32 seed) {
    u32 x = threadIdx.x + seed; u32 k = seed | 1u;
    for (int it = 0; it < ITER * 8; it++) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(k));
    if (x == 0x12345u) out[threadIdx.x] = x;
}
__global__ void k_lat_add(u32 *out, u32 seed) {
    u32 x = threadIdx.x + seed; u32 k = seed | 1u;
    for (int it = 0; it < ITER * 8; it++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k));
    if (x == 0x12345u) out[threadIdx.x] = x;
}
__global__ void k_lat_addc(u32 *out, u32 seed) {
    u32 x = threadIdx.x + seed; u32 k = seed | 1u;
    for (int it = 0; it < ITER * 8; it++) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc");
    if (x == 0x12345u) out[threadIdx.x] = x;
}
template <class K>
static void run_lat(const char *name, K kern) {
    u32 *d; (void)hipMalloc(&d, 4096);
    hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(prop.multiProcessorCount), dim3(256), 0, 0, d, 1u);   // 1 wave per SIMD
    (void)hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(prop.multiProcessorCount), dim3(256), 0, 0, d, (u32)r + 2u);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-14s dependent chain: %6.2f cycles per instruction (at %.0f MHz)\n", name,
           best * 1e-3 * prop.clockRate * 1e3 / (ITER * 8.0), prop.clockRate / 1e3);
    (void)hipFree(d);
}
template <class K>
static void run(const char *name, K kern, int per_iter) {
    u32 *d;
    hipMalloc(&d, 4096);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * 4;                 // 4 workgroups of 256 threads per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d, (u32)r + 2u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // wave-instructions per SIMD = 4 waves * ITER * per_iter; cycles = time * clock
    const double clock_hz = prop.clockRate * 1e3;
    const double instr = 4.0 * ITER * per_iter;
    printf("%-14s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.0f MHz)\n", name, best,
           best * 1e-3 * clock_hz / instr, clock_hz / 1e6);
    hipFree(d);
}

int main() {
    run("v_add_u32", k_add, CHAINS);
    run("v_xor_b32", k_xor, CHAINS);
    run("v_mul_lo_u32", k_mul_lo, CHAINS);
    run("v_mul_hi_u32", k_mul_hi, CHAINS);
    run("v_mul_u32_u24", k_mul24, CHAINS);
    run("v_mad_u32_u24", k_mad24, CHAINS);
    run("v_mad_u64_u32", k_mad64, CHAINS);
    run("v_add_co_u32", k_addco, CHAINS);
    run("v_addc_co vcc", k_addc, CHAINS);
    run("v_addc_co sgpr", k_addc_s, CHAINS);
    run("v_cndmask vcc", k_cndmask, CHAINS);
    run("v_cndmask sgpr", k_cndmask_s, CHAINS);
    run("v_mov_b32", k_mov, CHAINS);
    run("v_add3_u32", k_add3, CHAINS);
    run("v_lshl_add_u64", k_add64, CHAINS);
    run("v_cvt_f32_u32", k_cvt_f32, CHAINS);
    run("v_cvt_f64_u32", k_cvt_f64, CHAINS);
    run("v_fma_f32", k_fma32, CHAINS);
    run("v_fma_f64", k_fma64, CHAINS);
    run("v_mul_f64", k_mul64f, CHAINS);
    run_lat("v_add_u32", k_lat_add);
    run_lat("v_mul_lo_u32", k_lat_mul_lo);
    run_lat("v_mad_u64_u32", k_lat_mad64);
    run_lat("v_addc_co_u32", k_lat_addc);
    return 0;
}
