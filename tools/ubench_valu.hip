// Developer tool (GPU box): issue rate of a few VALU instructions on gfx950 - is v_mul_lo_u32 quarter rate?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_valu tools/ubench_valu.hip && tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t a, uint32_t b, int iters)
{
    uint32_t x0 = threadIdx.x, x1 = threadIdx.x + 1, x2 = threadIdx.x + 2, x3 = threadIdx.x + 3;
    uint32_t x4 = threadIdx.x + 4, x5 = threadIdx.x + 5, x6 = threadIdx.x + 6, x7 = threadIdx.x + 7;
    for (int i = 0; i < iters; ++i) {
#define STEP(x)                                                                                       \
    if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(a));                          \
    else if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(a));                        \
    else if (OP == 2) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(a));                    \
    else if (OP == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));           \
    else if (OP == 4) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(a));                 \
    else if (OP == 5) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(x));                               \
    else if (OP == 6) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));        \
    else if (OP == 7) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));         \
    else if (OP == 8) { uint64_t y = ((uint64_t)x << 32) | x; asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(y) : "v"(a)); x = (uint32_t)(y >> 32); } \
    else if (OP == 9) { uint64_t y = ((uint64_t)x << 32) | x; asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(y) : "v"(a)); x = (uint32_t)y; }         \
    else if (OP == 10) asm volatile("v_lshlrev_b32 %0, %1, %0" : "+v"(x) : "v"(a));                    \
    else if (OP == 11) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));            \
    else if (OP == 12) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
        STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7)
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7;
}

template <int OP>
static void run(const char *name, uint32_t *d)
{
    const int iters = 4096, blocks = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, 16);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 8;   // wave-instructions
    // per SIMD: winstr / 1024 SIMDs; cycles at 2.4 GHz
    printf("%-16s %8.3f ms  %6.2f cycles per wave-instruction and SIMD (2.4 GHz assumed)\n", name, ms, ms * 1e-3 * 2.4e9 / (winstr / 1024));
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<1>("v_add_u32", d);
    run<0>("v_mul_lo_u32", d);
    run<2>("v_mul_u32_u24", d);
    run<6>("v_mad_u32_u24", d);
    run<3>("v_perm_b32", d);
    run<4>("v_lshl_or_b32", d);
    run<5>("v_bfe_u32", d);
    run<7>("v_alignbit_b32", d);
    run<10>("v_lshlrev_b32", d);
    run<11>("v_bfi_b32", d);
    run<12>("v_and_or_b32", d);
    run<8>("v_lshlrev_b64 (+mov)", d);
    run<9>("v_lshrrev_b64 (+mov)", d);
    return 0;
}
