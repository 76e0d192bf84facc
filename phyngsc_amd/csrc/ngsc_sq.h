// ngsc_sq.h - device helpers of the sequence / quality kernels (kernels_sq.hip: one lane per record from
// global memory): SWAR base / quality
// classification, wave reductions, the per-lane MSB-first bit writer and the 256-thread exclusive scan.
#pragma once
#include "ngsc_swar.h"

namespace ngsc {

constexpr int SR_THREADS = 256;
constexpr uint32_t SR_HCOLS = 48;     // LDS columns: quality bytes 33..80 ('!'..'P'); anything else takes the global tables
constexpr uint32_t NB = 4;           // 16-byte chunks a lane fetches back to back (eight are no faster on 150-base reads)
constexpr uint32_t SR_HSTRIDE = SR_HCOLS + 1;   // dwords per LDS histogram row (odd: rows land in different banks)

__device__ __forceinline__ uint32_t sr_wsum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ uint32_t sr_wmax(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}

// (c >> 1) & 3 maps A, C, T, G to 0, 1, 2, 3; a byte is one of them iff the table gives the byte back
__device__ __forceinline__ uint32_t sr_idx(uint32_t w) { return (w >> 1) & 0x03030303u; }
__device__ __forceinline__ uint32_t sr_not_acgt(uint32_t w, uint32_t idx)   // non-zero byte <=> not A/C/G/T
{
    return __builtin_amdgcn_perm(0u, 0x47544341u, idx) ^ w;
}
// 0x80 per byte where the quality byte is outside 33 .. 32 + SR_HCOLS (the columns of the LDS tables)
__device__ __forceinline__ uint32_t sr_q_out(uint32_t w)
{
    const uint32_t t1 = (w | 0x80808080u) - 0x21212121u;   // bit 7 <=> (b & 0x7f) >= 33
    const uint32_t t2 = (w & 0x7F7F7F7Fu) + (0x7Fu - (32u + SR_HCOLS)) * 0x01010101u;   // bit 7 <=> (b & 0x7f) > 32 + SR_HCOLS
    return (~t1 | t2 | w) & 0x80808080u;
}
__device__ __forceinline__ uint32_t sr_byte(const uint32_t (&w)[4], uint32_t u) { return (w[u >> 2] >> (8 * (u & 3))) & 0xffu; }

// 16-bit mask of the ambiguous (non-ACGT) bases among the nv valid bytes of a chunk
__device__ __forceinline__ uint32_t sr_amb_mask(const uint32_t (&sw)[4], uint32_t nv)
{
    uint32_t m = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
        const uint32_t w = (sw[k] & vm) | (0x41414141u & ~vm);
        m |= t_movemask(t_nonzero01(sr_not_acgt(w, sr_idx(w)))) << (4 * k);
    }
    return m;
}

// One lane's MSB-first bit writer into LDS staging (or, for a tile whose stream outgrows the staging area,
// straight into the zeroed output).  Fewer than 32 bits stay pending, right-aligned in a 32-bit accumulator; a put of
// up to 32 bits completes at most one big-endian word, which leaves with an OR (a lane's first and last word are shared).
// All 32-bit operations: the word is a funnel shift of (pending : new bits) - the 64-bit accumulator of rounds 1-3 paid
// three 64-bit shifts per put, each several times the issue cost of a 32-bit instruction (tools/ubench_valu.hip).
struct SqWriter {
    uint32_t *words;
    uint64_t widx;
    uint32_t acc;    // the pending bits (the bits above them are zero)
    uint32_t nacc;   // how many: 0..31
    __device__ __forceinline__ void init(uint32_t *w, uint64_t word0, uint64_t bitpos)
    {
        words = w;
        widx = (bitpos >> 5) - word0;
        acc = 0;
        nacc = (uint32_t)(bitpos & 31);
    }
    __device__ __forceinline__ void put(uint32_t v, uint32_t nb)   // nb <= 32, v < 2^nb
    {
        const uint32_t tot = nacc + nb;
        if (tot >= 32) {
            // the string (acc : v) has tot bits; its first 32 are the word, the last r = tot - 32 (of v) stay pending
            const uint32_t word = __builtin_amdgcn_alignbit(acc, v << ((32u - nb) & 31u), nacc);
            if (word) atomicOr(&words[widx], __builtin_bswap32(word));
            ++widx;
            nacc = tot - 32;
            acc = __builtin_amdgcn_ubfe(v, 0u, nacc);
        } else {
            acc = (acc << nb) | v;
            nacc = tot;
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nacc) {
            const uint32_t word = acc << (32 - nacc);
            if (word) atomicOr(&words[widx], __builtin_bswap32(word));
            nacc = 0;
        }
    }
};

__device__ __forceinline__ uint32_t sq_scan_excl256(uint32_t v, uint32_t *s_part)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    __syncthreads();   // s_part may still be read from the previous call
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    uint32_t basev = 0;
#pragma unroll
    for (int w = 0; w < SR_THREADS / 64; ++w) basev += w < wave ? s_part[w] : 0;
    return basev + incl - v;
}


}  // namespace ngsc
