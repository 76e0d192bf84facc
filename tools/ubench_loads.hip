// ubench_loads.hip - what do the load patterns of the record kernels cost on gfx950?  (measurement helper, not product)
//
// One lane per 123-byte "record" of a 1 GB buffer, as in kernels_sq.hip / kernels_title.hip; every variant fetches the same
// bytes of the record (its bases and qualities: offsets 48..123) in a different way and adds them up.  Prints the time per
// pass and the cycles of CU time per wave-level load instruction (time x clock x CUs / (waves x loads)).
//   hipcc --offload-arch=gfx950 -O3 -o ubench_loads ubench_loads.hip && ./ubench_loads
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U2u { uint32_t x, y; };
struct __attribute__((packed, aligned(1))) U1u { uint32_t x; };

constexpr uint32_t REC = 123;

__device__ __forceinline__ uint32_t sum4(uint4 v) { return v.x + v.y + v.z + v.w; }

// mode 0: six unaligned 16-byte loads (bases 48/64/80, qualities 87/103/119) - what k_sq_stats_reg does
// mode 1: the same addresses rounded down to 16 (aligned, scattered)
// mode 2: six consecutive aligned 16-byte chunks from (48 & ~15)
// mode 3: six unaligned 8-byte loads
// mode 4: six unaligned 4-byte loads
// mode 5: coalesced: the wave's 64 records (7872 bytes) as eight 1 KB rows, lane l takes 16 bytes of each
// mode 6: three unaligned 16-byte loads (half the instructions of mode 0)
// mode 7: nine aligned chunks = the whole record, scattered
template <int MODE>
__global__ __launch_bounds__(256) void k_loads(const uint8_t *buf, uint32_t n_rec, uint32_t *out)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_rec) return;
    const uint32_t a = g * REC;
    uint32_t acc = 0;
    if (MODE == 0 || MODE == 6) {
        const uint32_t offs[6] = {48, 64, 80, 87, 103, 119};
#pragma unroll
        for (int k = 0; k < (MODE == 0 ? 6 : 3); ++k) {
            const U4u v = *reinterpret_cast<const U4u *>(buf + a + offs[k]);
            acc += v.x + v.y + v.z + v.w;
        }
    } else if (MODE == 1) {
        const uint32_t offs[6] = {48, 64, 80, 87, 103, 119};
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += sum4(*reinterpret_cast<const uint4 *>(buf + ((a + offs[k]) & ~15u)));
    } else if (MODE == 2) {
        const uint32_t b = (a + 48) & ~15u;
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += sum4(*reinterpret_cast<const uint4 *>(buf + b + 16 * k));
    } else if (MODE == 3) {
        const uint32_t offs[6] = {48, 64, 80, 87, 103, 119};
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const U2u v = *reinterpret_cast<const U2u *>(buf + a + offs[k]);
            acc += v.x + v.y;
        }
    } else if (MODE == 4) {
        const uint32_t offs[6] = {48, 64, 80, 87, 103, 119};
#pragma unroll
        for (int k = 0; k < 6; ++k) acc += reinterpret_cast<const U1u *>(buf + a + offs[k])->x;
    } else if (MODE == 5) {
        const uint32_t wave_first = (g & ~63u) * REC, lane = threadIdx.x & 63;
        const uint32_t b = (wave_first & ~15u) + 16 * lane;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += sum4(*reinterpret_cast<const uint4 *>(buf + b + 1024 * k));
    } else if (MODE == 7) {
        const uint32_t b = a & ~15u;
#pragma unroll
        for (int k = 0; k < 9; ++k) acc += sum4(*reinterpret_cast<const uint4 *>(buf + b + 16 * k));
    }
    if (acc == 0x12345678u) out[g & 1023] = acc;
}

// LDS: a wave stages its 64 records (8 rows of 1 KB, coalesced) and every lane reads its own record's six chunks back with
// unaligned 16-byte LDS reads; check = the same sum as mode 0.
template <int CHECK>
__global__ __launch_bounds__(256) void k_lds_stage(const uint8_t *buf, uint32_t n_rec, uint32_t *out, uint32_t *sums)
{
    __shared__ __align__(16) uint8_t s_span[4][8192 + 64];
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wave_first = (g & ~63u) * REC;
    const uint32_t b = wave_first & ~15u;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        *reinterpret_cast<uint4 *>(&s_span[wave][1024 * k + 16 * lane]) = *reinterpret_cast<const uint4 *>(buf + b + 1024 * k + 16 * lane);
    __builtin_amdgcn_wave_barrier();
    __syncthreads();
    const uint32_t o = g * REC - b;
    const uint32_t offs[6] = {48, 64, 80, 87, 103, 119};
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const U4u v = *reinterpret_cast<const U4u *>(&s_span[wave][o + offs[k]]);
        acc += v.x + v.y + v.z + v.w;
    }
    if (CHECK) { if (g < n_rec) sums[g] = acc; }
    else if (acc == 0x12345678u) out[g & 1023] = acc;
}

__global__ void k_ref_sums(const uint8_t *buf, uint32_t n_rec, uint32_t *sums)
{
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_rec) return;
    const uint32_t offs[6] = {48, 64, 80, 87, 103, 119};
    uint32_t acc = 0;
    for (int k = 0; k < 6; ++k)
        for (int w = 0; w < 4; ++w) {
            uint32_t x = 0;
            for (int b = 0; b < 4; ++b) x |= (uint32_t)buf[g * REC + offs[k] + 4 * w + b] << (8 * b);
            acc += x;
        }
    sums[g] = acc;
}

template <class F>
static float time_it(F &&launch, int reps)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    const uint32_t n_rec = 8u << 20;            // 8.4 M records, 1.03 GB
    const size_t bytes = (size_t)n_rec * REC + 16384;
    uint8_t *d = nullptr;
    uint32_t *out = nullptr, *s0 = nullptr, *s1 = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&s0, 4 * (size_t)n_rec));
    CK(hipMalloc(&s1, 4 * (size_t)n_rec));
    {
        std::vector<uint8_t> h(bytes);
        uint32_t x = 12345;
        for (size_t i = 0; i < bytes; ++i) { x = x * 1664525u + 1013904223u; h[i] = (uint8_t)(x >> 24); }
        CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    }
    const dim3 grid(n_rec / 256), blk(256);
    const double waves = n_rec / 64.0, clk = 2.4e9, cus = 256;
    struct { const char *name; int loads; float ms; } r[] = {
        {"0 six unaligned 16 B (k_sq_stats_reg today)", 6, 0}, {"1 six aligned-down 16 B, scattered", 6, 0},
        {"2 six consecutive aligned 16 B chunks", 6, 0}, {"3 six unaligned 8 B", 6, 0}, {"4 six unaligned 4 B", 6, 0},
        {"5 eight coalesced 16 B rows (whole span)", 8, 0}, {"6 three unaligned 16 B", 3, 0}, {"7 nine aligned chunks (whole record)", 9, 0},
    };
    r[0].ms = time_it([&] { hipLaunchKernelGGL(k_loads<0>, grid, blk, 0, 0, d, n_rec, out); }, 10);
    r[1].ms = time_it([&] { hipLaunchKernelGGL(k_loads<1>, grid, blk, 0, 0, d, n_rec, out); }, 10);
    r[2].ms = time_it([&] { hipLaunchKernelGGL(k_loads<2>, grid, blk, 0, 0, d, n_rec, out); }, 10);
    r[3].ms = time_it([&] { hipLaunchKernelGGL(k_loads<3>, grid, blk, 0, 0, d, n_rec, out); }, 10);
    r[4].ms = time_it([&] { hipLaunchKernelGGL(k_loads<4>, grid, blk, 0, 0, d, n_rec, out); }, 10);
    r[5].ms = time_it([&] { hipLaunchKernelGGL(k_loads<5>, grid, blk, 0, 0, d, n_rec, out); }, 10);
    r[6].ms = time_it([&] { hipLaunchKernelGGL(k_loads<6>, grid, blk, 0, 0, d, n_rec, out); }, 10);
    r[7].ms = time_it([&] { hipLaunchKernelGGL(k_loads<7>, grid, blk, 0, 0, d, n_rec, out); }, 10);
    for (auto &x : r)
        printf("%-48s %.3f ms  %6.1f CU-cycles per wave load  (%.2f TB/s of file)\n", x.name, x.ms,
               x.ms * 1e-3 * clk * cus / (waves * x.loads), (double)n_rec * REC / (x.ms * 1e-3) / 1e12);
    // LDS staging: correctness of unaligned 16-byte LDS reads, then time
    hipLaunchKernelGGL(k_ref_sums, grid, blk, 0, 0, d, n_rec, s0);
    hipLaunchKernelGGL(k_lds_stage<1>, grid, blk, 0, 0, d, n_rec, out, s1);
    CK(hipDeviceSynchronize());
    {
        std::vector<uint32_t> a(n_rec), b(n_rec);
        CK(hipMemcpy(a.data(), s0, 4 * (size_t)n_rec, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), s1, 4 * (size_t)n_rec, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (uint32_t i = 0; i < n_rec; ++i) bad += a[i] != b[i];
        printf("LDS-staged sums vs byte-wise reference: %zu mismatches of %u\n", bad, n_rec);
    }
    const float ms = time_it([&] { hipLaunchKernelGGL(k_lds_stage<0>, grid, blk, 0, 0, d, n_rec, out, s1); }, 10);
    printf("%-48s %.3f ms  (%.2f TB/s of file)\n", "8 coalesced rows -> LDS -> six unaligned LDS reads", ms, (double)n_rec * REC / (ms * 1e-3) / 1e12);
    return 0;
}
