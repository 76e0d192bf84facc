// ngsc_swar.h - device helpers shared by the register-streaming kernels: unaligned global loads and
// four-bytes-per-instruction (SWAR) mask tricks.
#pragma once
#include "ngsc_common.h"
#include <hip/hip_runtime.h>

namespace ngsc {

// gfx950 global memory takes unaligned dword loads; the packed types tell the compiler so (one
// global_load_dwordx4 / dwordx2 instead of byte loads).  Never reads past buf_len.
struct __attribute__((packed, aligned(1))) U4u { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(1))) U2u { uint32_t x, y; };

__device__ __forceinline__ uint4 load16_unaligned(const uint8_t *buf, uint32_t a, uint32_t buf_len)
{
    if (a + 16 <= buf_len) { const U4u v = *reinterpret_cast<const U4u *>(buf + a); return make_uint4(v.x, v.y, v.z, v.w); }
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t x = 0; x < 16; ++x)
        if (a + x < buf_len) w[x >> 2] |= (uint32_t)buf[a + x] << (8 * (x & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ uint2 load8_unaligned(const uint8_t *buf, uint32_t a, uint32_t buf_len)
{
    if (a + 8 <= buf_len) { const U2u v = *reinterpret_cast<const U2u *>(buf + a); return make_uint2(v.x, v.y); }
    uint32_t w[2] = {0, 0};
    for (uint32_t x = 0; x < 8; ++x)
        if (a + x < buf_len) w[x >> 2] |= (uint32_t)buf[a + x] << (8 * (x & 3));
    return make_uint2(w[0], w[1]);
}

// Bytes a kernel reads ONCE and nothing reads soon after: loads with the non-temporal policy (`nt`) leave the L2 and the Infinity Cache to
// what is read again - masks, record tables.  The newline count reads the job that way: 0.212 -> 0.19 ms per GB of its own, the cut
// planner behind it 0.075 -> 0.03-0.05 ms, the step 1.887 -> 1.814 ms per GB of 36-base reads, 1.108 -> 1.084 per 400 MB of 150-base
// reads, 1.722 -> 1.688 of 36..300-base reads.  NOT the statistics kernels' reads of the same bytes (the sequence/quality kernel's
// spans: 1.814 -> 1.839; the title kernel's chunks on top: no change): what they leave in the caches the next pass over the file
// finds (profiles/r04_ab_nontemporal_loads.txt).  The measure and pack kernels read the chunk-major copies that way too where the reads
// are long (kernels_sq.hip: sq_chunk_load<true>): 150-base reads 1.109 -> 1.066 ms per 400 MB, 36..300-base reads 1.708 -> 1.648; at
// 36 bases the kernels gain nothing and the measure kernel, beside the title kernels, takes 0.37 instead of 0.31 ms: left as it was.
// Measured and not kept: non-temporal STORES for the early clearing of the output (36-base reads 1.806 -> 1.832: the pack kernel's
// read-modify-writes find the zeros in the caches) and for the pack kernel's output words (no gain, 150-base reads 1.073 -> 1.093);
// non-temporal loads of the title statistics kernel's chunks and of the newline masks in k_nl_write (within +-0.3 %).
typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld16_nt(const uint8_t *p)   // p: a multiple of 16
{
    const nt_u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
}

struct __attribute__((packed, aligned(1))) U1u { uint32_t x; };
__device__ __forceinline__ uint32_t load4_unaligned(const uint8_t *buf, uint32_t a) { return reinterpret_cast<const U1u *>(buf + a)->x; }

// 0x01 flags in bytes 0..3 -> bits 0..3
__device__ __forceinline__ uint32_t t_movemask(uint32_t flags01) { return (flags01 * 0x01020408u) >> 24; }
// per byte: 0x01 where the byte of x is non-zero
__device__ __forceinline__ uint32_t t_nonzero01(uint32_t x)
{
    return ((((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) >> 7) & 0x01010101u;
}
__device__ __forceinline__ uint64_t t_lowbytes(uint32_t n)   // 0xFF in the n lowest bytes, n clamped to 0..8
{
    return n >= 8 ? ~0ull : ((1ull << (8 * n)) - 1ull);
}
__device__ __forceinline__ uint64_t t_lowbits(uint32_t n)    // n lowest bits, n in 0..64
{
    return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
}

// ---- masks with one bit per title byte, W 64-bit words (titles of up to 64 * W bytes) ----
// the 64 mask bits that start at bit pos
template <uint32_t W>
__device__ __forceinline__ uint64_t t_window(const uint64_t (&m)[W], uint32_t pos)
{
    if constexpr (W == 1) {
        return pos < 64 ? m[0] >> pos : 0ull;
    } else {
        if (pos >= 128) return 0ull;
        if (pos >= 64) return m[1] >> (pos - 64);
        return (m[0] >> pos) | (pos ? m[1] << (64 - pos) : 0ull);
    }
}
// distance from bit pos to the next set bit at or behind it (0 if there is none: a title always ends in one)
template <uint32_t W>
__device__ __forceinline__ uint32_t t_next_set(const uint64_t (&m)[W], uint32_t pos)
{
    const uint64_t lo = t_window<W>(m, pos);
    if (lo) return (uint32_t)__builtin_ctzll(lo);
    if constexpr (W > 1) {
        const uint64_t hi = t_window<W>(m, pos + 64);
        if (hi) return 64u + (uint32_t)__builtin_ctzll(hi);
    }
    return 0u;
}
// are the len bits from bit pos all set?
template <uint32_t W>
__device__ __forceinline__ bool t_all_set(const uint64_t (&m)[W], uint32_t pos, uint32_t len)
{
    const uint64_t l0 = t_lowbits(len < 64 ? len : 64);
    bool ok = (t_window<W>(m, pos) & l0) == l0;
    if constexpr (W > 1) {
        if (len > 64) {
            const uint64_t l1 = t_lowbits(len - 64);
            ok = ok && (t_window<W>(m, pos + 64) & l1) == l1;
        }
    }
    return ok;
}

// 0xFF in the n lowest bytes of a dword, n clamped to 0..4
__device__ __forceinline__ uint32_t t_lowbytes32(int32_t n)
{
    return n >= 4 ? ~0u : (n <= 0 ? 0u : ((1u << (8 * n)) - 1u));
}

}  // namespace ngsc
