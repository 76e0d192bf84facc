// kernels_index.hip - record index of a FASTQ job and the sub-block cut plan.
//
// Replaces the per-thread newline scan of phyNGSC.cpp:254-331 (K0 in SURVEY.md): instead of walking
// each 8 MiB window serially, every newline of the job is found in one coalesced pass, numbered by a
// device-wide scan, and lines 4r / 4r+1 / 4r+3 give record r's title end, sequence end and the next
// record's line start.  HBM-bound: 1 B read per input byte (twice: count + write), 12 B written per
// record.  No MFMA - integer byte classification.
#include "ngsc_kernels.h"
#include "ngsc_swar.h"
#include <cstdlib>

namespace ngsc {

constexpr int TILE_THREADS = 256;
constexpr int BYTES_PER_THREAD = 32;
constexpr uint32_t PIECE_BYTES = TILE_THREADS * BYTES_PER_THREAD;  // 8 KiB per coalesced piece
constexpr uint32_t PIECES = 8;                                     // pieces per workgroup
constexpr uint32_t TILE_BYTES = PIECE_BYTES * PIECES;              // 64 KiB per workgroup: fewer tile counts for the single-workgroup scan

// exact per-byte "== 0" flags (bit 7 of each byte) of a 32-bit word
__device__ __forceinline__ uint32_t zero_bytes(uint32_t y)
{
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

// Newline flags of the 16 bytes starting at `base` (restricted to [start, len)) as a 16-bit mask, bit k = byte base + k.
__device__ __forceinline__ uint32_t load_nl_mask16(const uint8_t *buf, uint32_t base, uint32_t start, uint32_t len)
{
    uint32_t m = 0;
    if (base >= start && base + 16 <= len) {
        const uint4 a = *reinterpret_cast<const uint4 *>(buf + base);
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) m |= (((zero_bytes(w[i] ^ 0x0A0A0A0Au) >> 7) * 0x01020408u) >> 24) << (4 * i);
    } else {
        for (uint32_t k = 0; k < 16; ++k) {
            const uint32_t pos = base + k;
            if (pos >= start && pos < len && buf[pos] == '\n') m |= 1u << k;
        }
    }
    return m;
}

__device__ __forceinline__ uint32_t wave_reduce_add(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// at_file_end: the job buffer ends where the file ends.  A newline is assumed just past the last byte: it closes a
// last quality line that has none (also an empty one, after "+\n"), as the reference does by working from positions
// (phyNGSC.cpp:308-312); behind a complete file it opens a fifth line that no record owns and nothing reads.
__global__ __launch_bounds__(TILE_THREADS) void k_nl_count(const uint8_t *buf, uint32_t len, const uint32_t *start_ptr,
                                                           uint32_t *tile_cnt, uint32_t *nl_mask, uint32_t at_file_end)
{
    __shared__ uint32_t s_part[TILE_THREADS / 64];
    const uint32_t start = *start_ptr;
    uint32_t c = 0;
    // Every load instruction of a wave covers one contiguous KB (16 bytes per lane), and a tile that lies inside [start, len)
    // - all but the job's first and last - takes a branch-free path with eight loads in flight per lane: behind the
    // per-load range tests the compiler waited for every load before issuing the next (one KB in flight per wave).  One
    // bit per byte is kept: the write pass reads 1/8 of a byte per input byte instead of the input again; a lane's 16 bits
    // are half a mask word.
    uint16_t *mask16 = reinterpret_cast<uint16_t *>(nl_mask);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t tile0 = blockIdx.x * TILE_BYTES;
    if (tile0 >= start && tile0 + TILE_BYTES <= len && tile0 + TILE_BYTES > tile0) {
        // (round 3) A wave takes 16 contiguous KB of the tile, eight rows of 1 KB per trip.  Its 16-bit masks go through LDS, so that
        // a lane leaves with the masks of eight CONSECUTIVE chunks: one 16-byte store per lane and trip (1 KB per wave
        // instruction) instead of eight 2-byte ones (128 B per instruction - as many store instructions as loads, and the
        // texture addresser was this kernel's busiest unit).
        __shared__ __align__(16) uint16_t s_m[TILE_THREADS / 64][512];
        const uint32_t wbase = tile0 + wave * (TILE_BYTES / (TILE_THREADS / 64));
#pragma unroll
        for (uint32_t half = 0; half < 2; ++half) {
            uint4 v[PIECES];
#pragma unroll
            for (uint32_t k = 0; k < PIECES; ++k)
                v[k] = ld16_nt(buf + wbase + (half * PIECES + k) * 1024u + lane * 16u);   // (read once: ngsc_swar.h)
#pragma unroll
            for (uint32_t k = 0; k < PIECES; ++k) {
                const uint32_t w[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                uint32_t m = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) m |= (((zero_bytes(w[i] ^ 0x0A0A0A0Au) >> 7) * 0x01020408u) >> 24) << (4 * i);
                c += (uint32_t)__popc(m);
                s_m[wave][k * 64 + lane] = (uint16_t)m;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 mm = *reinterpret_cast<const uint4 *>(&s_m[wave][lane * 8]);
            *reinterpret_cast<uint4 *>(mask16 + (wbase + half * PIECES * 1024u) / 16 + lane * 8) = mm;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the next trip writes s_m again)
            __builtin_amdgcn_wave_barrier();
        }
    } else {
        for (uint32_t pc = 0; pc < PIECES; ++pc) {
            for (uint32_t h = 0; h < 2; ++h) {
                const uint32_t base = tile0 + pc * PIECE_BYTES + (wave * 2 + h) * 1024u + lane * 16u;
                uint32_t m = (base < len && base + 16 > start) ? load_nl_mask16(buf, base, start, len) : 0u;
                if (at_file_end && base <= len && len - base < 16 && len > start) m |= 1u << (len - base);   // the assumed newline at position len
                c += (uint32_t)__popc(m);
                mask16[base / 16] = (uint16_t)m;
            }
        }
    }
    c = wave_reduce_add(c);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_cnt[blockIdx.x] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

// Single workgroup exclusive scan of the tile counts (n up to ~64k: a job is < 4 GiB); total -> *total_out.
// (round 3) the counts come into LDS with coalesced loads and leave the same way; a thread scans its stretch there - with the
// stretches read straight from global memory every thread walked 15 addresses 60 bytes apart, twice (0.017 ms on the index chain)
constexpr uint32_t SCAN_LDS = 15 * 1024;   // counts held in LDS (a 1 GB job has 15 259 tiles); larger jobs take several rounds
__global__ __launch_bounds__(1024) void k_scan_tiles(uint32_t *tile_cnt, uint32_t n, uint32_t *total_out)
{
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_c[SCAN_LDS];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n; base += SCAN_LDS) {
        const uint32_t m = n - base < SCAN_LDS ? n - base : SCAN_LDS;
        for (uint32_t k = threadIdx.x; k < m; k += 1024) s_c[k] = tile_cnt[base + k];
        __syncthreads();
        const uint32_t per = (m + 1023) / 1024;
        const uint32_t lo = threadIdx.x * per < m ? threadIdx.x * per : m;
        const uint32_t hi = lo + per < m ? lo + per : m;
        uint32_t sum = 0;
        for (uint32_t i = lo; i < hi; ++i) sum += s_c[i];
        const uint32_t incl = wave_scan_incl(sum);
        if ((threadIdx.x & 63) == 63) s_wave[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint32_t round_total = 0;
        if (threadIdx.x < 64) {
            const uint32_t v = threadIdx.x < 16 ? s_wave[threadIdx.x] : 0;
            const uint32_t sc = wave_scan_incl(v);
            if (threadIdx.x < 16) s_wave[threadIdx.x] = sc - v;  // exclusive wave offsets
        }
        __syncthreads();
        uint32_t run = carry + s_wave[threadIdx.x >> 6] + incl - sum;
        for (uint32_t i = lo; i < hi; ++i) {
            const uint32_t c = s_c[i];
            s_c[i] = run;
            run += c;
        }
        // the round's total = the last thread's running sum (every stretch is accounted for: empty ones add nothing)
        __syncthreads();
        if (threadIdx.x == 1023) s_wave[0] = run;
        __syncthreads();
        round_total = s_wave[0];
        for (uint32_t k = threadIdx.x; k < m; k += 1024) tile_cnt[base + k] = s_c[k];
        carry = round_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

// Writes the record table.  Line L (0-based from *start_ptr) ends at a newline; L&3 == 0: title end of
// record L>>2, 1: sequence end, 3: the next record's line start.
__global__ __launch_bounds__(TILE_THREADS) void k_nl_write(const uint32_t *nl_mask, const uint32_t *start_ptr,
                                                           const uint32_t *tile_off, uint32_t *te, uint32_t *se,
                                                           uint32_t *ls, uint32_t rec_cap)
{
    __shared__ uint32_t s_wave[2][TILE_THREADS / 64];
    const uint32_t start = *start_ptr;
    if (blockIdx.x == 0 && threadIdx.x == 0) ls[0] = start;
    uint32_t line_base = tile_off[blockIdx.x];
    // (round 3) four mask words - 128 input bytes, four or five newlines of a short-read file - per thread and round: two scans and
    // two barriers per 64 KiB tile instead of eight of each (0.083 -> ms per GB see DESIGN.md)
    constexpr uint32_t WPT = 4, ROUNDS = TILE_BYTES / (TILE_THREADS * WPT * 32);
    static_assert(ROUNDS * TILE_THREADS * WPT * 32 == TILE_BYTES, "a tile is a whole number of rounds");
#pragma unroll
    for (uint32_t pc = 0; pc < ROUNDS; ++pc) {
        const uint32_t widx = (blockIdx.x * ROUNDS + pc) * (TILE_THREADS * WPT) + threadIdx.x * WPT;   // (the mask array covers whole tiles)
        const uint4 mm = *reinterpret_cast<const uint4 *>(nl_mask + widx);
        uint32_t m[WPT] = {mm.x, mm.y, mm.z, mm.w};
        const uint32_t c = (uint32_t)(__popc(m[0]) + __popc(m[1]) + __popc(m[2]) + __popc(m[3]));
        const uint32_t incl = wave_scan_incl(c);
        if ((threadIdx.x & 63) == 63) s_wave[pc & 1][threadIdx.x >> 6] = incl;
        __syncthreads();   // (two buffers: the next round's writes cannot overtake this round's reads)
        uint32_t wbase = 0, total = 0;
#pragma unroll
        for (uint32_t w = 0; w < TILE_THREADS / 64; ++w) {
            if (w < (threadIdx.x >> 6)) wbase += s_wave[pc & 1][w];
            total += s_wave[pc & 1][w];
        }
        uint32_t line = line_base + wbase + incl - c;
        line_base += total;
#pragma unroll
        for (uint32_t k4 = 0; k4 < WPT; ++k4) {
            const uint32_t base = (widx + k4) * 32;
            uint32_t x = m[k4];
            while (x) {
                const uint32_t k = (uint32_t)__ffs(x) - 1;
                x &= x - 1;
                const uint32_t pos = base + k;
                const uint32_t r = line >> 2, kind = line & 3;
                if (r < rec_cap) {
                    if (kind == 0) te[r] = pos;
                    else if (kind == 1) se[r] = pos;
                    else if (kind == 3) ls[r + 1] = pos + 1;
                }
                ++line;
            }
        }
    }
}

// Structural validation of every record + job-wide max quality length and record size.
//   quality length must equal sequence length (README.md:97-103); line 2 == "+" is checked later.
// The reference assumes both and runs into undefined behaviour otherwise; we report ERR_FORMAT.
__device__ __forceinline__ void validate_records(const uint32_t *te, const uint32_t *se, const uint32_t *ls, uint32_t n_rec,
                                                 uint32_t first, uint32_t stride, uint32_t *max_q, uint32_t *max_rec,
                                                 uint32_t *max_title, int32_t *err)
{
    uint32_t mq = 0, mr = 0, mt = 0;
    bool bad = false;
    auto one = [&](uint32_t t, uint32_t s, uint32_t l0, uint32_t l1) __attribute__((always_inline)) {
        const uint32_t L = s - t - 1, RB = l1 - l0, TL = t - l0 + 1;
        // the bare '+' line is checked by the statistics kernels, which read the bytes anyway
        bad = bad || !(t >= l0 && s > t) || l1 != 2 * s - t + 3;
        mq = L > mq ? L : mq;
        mr = RB > mr ? RB : mr;
        mt = TL > mt ? TL : mt;
    };
    for (uint32_t r = first; r < n_rec; r += stride) one(te[r], se[r], ls[r], ls[r + 1]);
    if (__any(bad)) { if ((threadIdx.x & 63) == 0) atomicMin(err, (int32_t)ERR_FORMAT); }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        uint32_t o = __shfl_xor(mq, d); mq = o > mq ? o : mq;
        o = __shfl_xor(mr, d); mr = o > mr ? o : mr;
        o = __shfl_xor(mt, d); mt = o > mt ? o : mt;
    }
    if ((threadIdx.x & 63) == 0) {   // same-address atomics serialise: only waves that raise a maximum issue one
        if (mq > *max_q) atomicMax(max_q, mq);
        if (mr > *max_rec) atomicMax(max_rec, mr);
        if (mt > *max_title) atomicMax(max_title, mt);
    }
}

// First-record synchronisation of every virtual rank > 0 (phyNGSC.cpp:131-156): first '@' at or after
// the region start, then the first newline after it; if the byte after that newline is '@' the record
// starts there, otherwise at the first '@'.  One wave per rank, 64 bytes per step.
// resume_abs != 0: the rank is taken up at this file offset (the first record of its next window): nothing to search.
__global__ __launch_bounds__(64) void k_zero_words(uint32_t *p, uint32_t n)
{
    for (uint32_t k = threadIdx.x; k < n; k += 64) p[k] = 0;
}

// b: the rank's index in the call (rank_lo + b is the rank), one wave.
__device__ __forceinline__ void sync_first_rank(const uint8_t *buf, uint32_t len, uint64_t buf_off, uint64_t file_size, uint32_t P, uint32_t rank_lo,
                                                uint32_t b, uint64_t *first_abs, uint32_t *start_rel, int32_t *err, uint64_t resume_abs)
{
    const uint32_t r = rank_lo + b;
    const int lane = threadIdx.x & 63;
    if (resume_abs) {
        if (lane == 0) {
            if (resume_abs < buf_off || resume_abs - buf_off >= len) atomicMin(err, (int32_t)ERR_ARG);
            first_abs[b] = resume_abs;
            if (b == 0) *start_rel = (uint32_t)(resume_abs - buf_off);
        }
        return;
    }
    RankGeom g = rank_geometry(file_size, P, r);
    if (r == 0) {
        if (lane == 0) { first_abs[b] = 0; if (b == 0) *start_rel = 0; }
        return;
    }
    int64_t rel0 = g.wr_start - (int64_t)buf_off;
    int64_t lim64 = rel0 + g.bsize;
    if (rel0 < 0) { if (lane == 0) atomicMin(err, (int32_t)ERR_ARG); return; }
    uint32_t limit = lim64 < (int64_t)len ? (uint32_t)lim64 : len;
    uint32_t pos = (uint32_t)rel0;
    int64_t at = -1, nl = -1;
    for (; pos < limit && at < 0; pos += 64) {
        uint32_t p = pos + lane;
        bool hit = p < limit && buf[p] == '@';
        unsigned long long m = __ballot(hit);
        if (m) at = (int64_t)pos + (__ffsll((long long)m) - 1);
    }
    if (at >= 0) {
        for (pos = (uint32_t)at; pos < limit && nl < 0; pos += 64) {
            uint32_t p = pos + lane;
            bool hit = p < limit && buf[p] == '\n';
            unsigned long long m = __ballot(hit);
            if (m) nl = (int64_t)pos + (__ffsll((long long)m) - 1);
        }
    }
    if (lane == 0) {
        if (at < 0 || nl < 0) { atomicMin(err, (int32_t)ERR_FORMAT); first_abs[b] = 0; return; }
        // (the look at the byte behind the newline stays inside the rank's first window, like the two scans: phyNGSC.cpp:131-156)
        uint64_t res = (nl + 1 < (int64_t)limit && buf[nl + 1] == '@') ? (uint64_t)(nl + 1) : (uint64_t)at;
        first_abs[b] = res + buf_off;
        if (b == 0) *start_rel = (uint32_t)res;
    }
}

// A wave per rank.  zero / n_zero (a call of up to SYNC_ONE_WG ranks: ONE workgroup): the words of the step's small state - the
// error word first - are cleared here, in front of a workgroup barrier, instead of by a launch of their own in front of this one
// (the head of a step is bound by the host's launches: 4 us each with the GPU waiting).
constexpr uint32_t SYNC_ONE_WG = 64, SYNC_WAVES = 16;
__global__ __launch_bounds__(64 * SYNC_WAVES) void k_sync_first(const uint8_t *buf, uint32_t len, uint64_t buf_off, uint64_t file_size,
                                                                uint32_t P, uint32_t rank_lo, uint32_t n_ranks, uint64_t *first_abs, uint32_t *start_rel,
                                                                int32_t *err, uint64_t resume_abs, uint32_t *zero, uint32_t n_zero)
{
    if (zero) {   // (gridDim.x == 1)
        for (uint32_t k = threadIdx.x; k < n_zero; k += blockDim.x) zero[k] = 0;
        __syncthreads();   // (workgroup-scope release / acquire: the stores have left for the L2 before any wave's atomic on the error word)
    }
    const uint32_t waves = blockDim.x >> 6;
    for (uint32_t b = blockIdx.x * waves + (threadIdx.x >> 6); b < n_ranks; b += gridDim.x * waves)
        sync_first_rank(buf, len, buf_off, file_size, P, rank_lo, b, first_abs, start_rel, err, resume_abs);
}

// plan_rank() search policy for a whole wave: 64 probes per round (a 2^21-record window needs 4 rounds instead of 21
// binary-search steps; every round is one memory latency, and the planner's chain of searches is serial).
struct WaveSearch {
    __device__ uint32_t one(const uint32_t *a, uint32_t n, uint64_t key) const
    {
        uint32_t r0, r1;
        pair(a, n, key, a, 0, 0, 0, n, r0, r1);
        return r0;
    }
    __device__ uint32_t at(const uint32_t *a, uint32_t idx) const { return a[idx]; }
    // hint: where both bounds are expected, as a distance from lo (the previous window's record count; windows of one
    // file hold nearly the same number of records).  The first round then probes 64 positions around lo + hint, a
    // 1/2048 of hint apart: if the bound lies between two of them - it nearly always does - one more round finishes
    // (two dependent memory latencies per window instead of four); if not, the probes still narrow the range from one side.
    __device__ void pair(const uint32_t *a, uint32_t na, uint64_t ka, const uint32_t *b, uint32_t nb, uint64_t kb, uint32_t lo,
                         uint32_t span, uint32_t &ra, uint32_t &rb, uint32_t hint = 0) const
    {
        const uint32_t lane = threadIdx.x & 63;
        uint32_t la = lo < na ? lo : na, ha = (uint64_t)lo + span < na ? lo + span : na;
        uint32_t lb = lo < nb ? lo : nb, hb = (uint64_t)lo + span < nb ? lo + span : nb;
        if (hint > 64 && la < ha && lb < hb) {
            const uint32_t step = hint / 2048 + 1;
            const uint64_t c0 = (uint64_t)lo + hint;
            const uint64_t p64 = c0 + (uint64_t)lane * step;
            const uint64_t pp = p64 > 32ull * step ? p64 - 32ull * step : 0;                // lo + hint + (lane - 32) * step
            const uint32_t pa = (uint32_t)(pp < la ? la : (pp >= ha ? ha - 1 : pp)), pb = (uint32_t)(pp < lb ? lb : (pp >= hb ? hb - 1 : pp));
            const uint32_t va = a[pa], vb = b[pb];
            const uint64_t ma = __ballot((uint64_t)va < ka), mb = __ballot((uint64_t)vb < kb);   // sorted arrays: a run of ones from lane 0
            const uint32_t ca = (uint32_t)__popcll(ma), cb = (uint32_t)__popcll(mb);
            // bound in (probe[c-1], probe[c]]: the probes' positions come back through the lanes that hold them
            const uint32_t pa_lo = (uint32_t)__shfl((int)pa, ca ? (int)ca - 1 : 0), pa_hi = (uint32_t)__shfl((int)pa, ca < 64 ? (int)ca : 63);
            const uint32_t pb_lo = (uint32_t)__shfl((int)pb, cb ? (int)cb - 1 : 0), pb_hi = (uint32_t)__shfl((int)pb, cb < 64 ? (int)cb : 63);
            if (ca) la = pa_lo + 1;
            if (ca < 64) ha = pa_hi;
            if (cb) lb = pb_lo + 1;
            if (cb < 64) hb = pb_hi;
        }
        while (la < ha || lb < hb) {
            const uint32_t sa = (ha - la + 63) >> 6, sb = (hb - lb + 63) >> 6;
            const uint32_t pa = la + lane * sa, pb = lb + lane * sb;
            const bool ina = la < ha && pa < ha, inb = lb < hb && pb < hb;
            const uint32_t va = ina ? a[pa] : 0u, vb = inb ? b[pb] : 0u;   // both rounds' loads are in flight together
            const uint32_t ca = (uint32_t)__popcll(__ballot(ina && (uint64_t)va < ka));
            const uint32_t cb = (uint32_t)__popcll(__ballot(inb && (uint64_t)vb < kb));
            if (la < ha) {
                if (ca == 0) ha = la;
                else { const uint32_t top = la + ca * sa; la = la + (ca - 1) * sa + 1; ha = top < ha ? top : ha; }
            }
            if (lb < hb) {
                if (cb == 0) hb = lb;
                else { const uint32_t top = lb + cb * sb; lb = lb + (cb - 1) * sb + 1; hb = top < hb ? top : hb; }
            }
        }
        ra = la;
        rb = lb;
    }
};

// (round 3) The planner's chain - window k+1 starts at the first record behind window k's stop position - is serial, and
// every link used to cost three dependent trips to memory (two search rounds, then ls[last + 1]): 15 windows of a 125 MB
// rank took 65-95 us, the tail of the index phase.  But the links are predictable: a window stops 8 MiB - 500 B behind
// its start and starts at most one record behind the previous stop, so window k's stop lies within k records' bytes behind
// x0 + (k + 1) * (8 MiB - 500).  The rank's four waves look those places up IN PARALLEL first (a search each, hinted by
// the job's record density) and park PL_ENT record-table entries around each in LDS; the chain then runs out of LDS.
// Whatever the cache does not hold (long records, later windows of a long rank) is searched in memory as before.
constexpr uint32_t PL_WIN = 24;     // windows looked up ahead per rank
constexpr uint32_t PL_ENT = 128;    // record table entries parked per window (two per lane)
struct PlanCache {
    uint32_t ls[PL_WIN][PL_ENT], te[PL_WIN][PL_ENT];
    uint32_t base[PL_WIN];          // record index of entry 0; 0xffffffff: nothing parked
};

struct CachedSearch {
    const PlanCache *pc;
    const uint32_t *te_ptr, *ls_ptr;
    uint32_t nwin;
    mutable uint32_t next;          // the stretch the next window's bounds are expected in
    WaveSearch ws;
    const uint32_t *tile_off, *total_lines;   // the index kernel's line count in front of every 64 KiB tile (or nullptr)
    uint32_t n_tiles;
#ifdef NGSC_PLAN_TIMING
    mutable uint32_t miss_pair = 0, miss_at = 0;
#endif
    // the records among which ls[r] >= key begins: those of the tile that holds byte key - 1 (ls[r] >= key <=> r >= (lines that end
    // in front of byte key - 1, + 1) / 4, rounded up, and the tile's line counts bracket that number)
    __device__ bool bracket(uint64_t key, uint32_t &lo, uint32_t &span) const
    {
        const uint64_t t = key ? (key - 1) / TILE_BYTES : 0;
        if (!tile_off || t >= n_tiles) return false;
        const uint32_t l_lo = tile_off[t], l_hi = t + 1 < n_tiles ? tile_off[t + 1] : *total_lines;
        lo = l_lo / 4 ? l_lo / 4 - 1 : 0;
        span = l_hi / 4 + 3 - lo;
        return true;
    }
    __device__ uint32_t one(const uint32_t *a, uint32_t n, uint64_t key) const
    {
        uint32_t lo, span, r0, r1;
        if (a == ls_ptr && bracket(key, lo, span)) { ws.pair(a, n, key, a, 0, 0, lo, span, r0, r1); return r0; }
        return ws.one(a, n, key);
    }
    // lower bound of key among the parked entries of stretch k, if they decide it (sorted; entries behind the table's end are ~0)
    __device__ bool lookup(const uint32_t (*tab)[PL_ENT], uint32_t k, uint64_t key, uint32_t &res) const
    {
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t v0 = tab[k][lane], v1 = tab[k][lane + 64];
        const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)v0, 0), last = (uint32_t)__builtin_amdgcn_readlane((int)v1, 63);
        if (!((uint64_t)first < key && key <= (uint64_t)last)) return false;
        res = pc->base[k] + (uint32_t)__popcll(__ballot((uint64_t)v0 < key)) + (uint32_t)__popcll(__ballot((uint64_t)v1 < key));
        return true;
    }
    __device__ void pair(const uint32_t *a, uint32_t na, uint64_t ka, const uint32_t *b, uint32_t nb, uint64_t kb, uint32_t lo,
                         uint32_t span, uint32_t &ra, uint32_t &rb, uint32_t hint = 0) const
    {
        bool ha = false, hb = false;
        uint32_t fa = 0, fb = 0;
        if (a == te_ptr && b == ls_ptr) {
            for (uint32_t t = 0; t < 2 && !(ha && hb); ++t) {
                const uint32_t k = next + t;
                if (k >= nwin || pc->base[k] == 0xffffffffu) break;
                if (!ha) ha = lookup(pc->te, k, ka, fa);
                if (!hb) hb = lookup(pc->ls, k, kb, fb);
            }
        }
        ++next;
        if (ha && hb) {
            const uint32_t la = lo < na ? lo : na, hia = (uint64_t)lo + span < na ? lo + span : na;
            const uint32_t lb = lo < nb ? lo : nb, hib = (uint64_t)lo + span < nb ? lo + span : nb;
            ra = fa < la ? la : (fa > hia ? hia : fa);
            rb = fb < lb ? lb : (fb > hib ? hib : fb);
            return;
        }
#ifdef NGSC_PLAN_TIMING
        ++miss_pair;
#endif
        ws.pair(a, na, ka, b, nb, kb, lo, span, ra, rb, hint);
    }
    __device__ uint32_t at(const uint32_t *a, uint32_t idx) const
    {
        if (a == ls_ptr) {
            for (uint32_t t = 0; t < 2; ++t) {   // (the window just planned: stretch next - 1, or the one behind it)
                const uint32_t k = next - 1 + t;
                if (next == 0 || k >= nwin) continue;
                const uint32_t b0 = pc->base[k];
                if (b0 != 0xffffffffu && idx >= b0 && idx - b0 < PL_ENT) {
                    const uint32_t v = pc->ls[k][idx - b0];
                    if (v != 0xffffffffu) return v;
                }
            }
        }
#ifdef NGSC_PLAN_TIMING
        ++miss_at;
#endif
        return a[idx];
    }
};

// One launch, two jobs that do not depend on each other: workgroups [0, n_ranks) plan the sub-block cuts of
// one rank each (a serial chain of binary searches over the record table, see plan_rank()), the others
// validate the structure of every record and reduce the job-wide maxima while those chains run.
constexpr uint32_t CV_THREADS = 1024;   // (round 4) sixteen waves: a window of the planner's look-ahead each, see below
__global__ __launch_bounds__(CV_THREADS) void k_cuts_validate(uint64_t file_size, uint32_t P, uint32_t rank_lo, uint32_t n_ranks,
                                                       uint64_t buf_off, const uint32_t *te, const uint32_t *se, const uint32_t *ls,
                                                       const uint32_t *total_lines, const uint64_t *first_abs, Cut *cuts,
                                                       uint32_t max_cuts, RankPlan *plans, uint32_t *max_q, uint32_t *max_rec,
                                                       uint32_t *max_title, int32_t *err, const uint8_t *tail_buf, PlanResume from,
                                                       uint64_t buf_end_abs, const uint32_t *tile_off, uint32_t n_tiles)
{
    const uint32_t n_rec_total = *total_lines / 4;
    // tail_buf: the job buffer when it ends where the file ends.  Two real newlines behind the last complete record (the
    // assumed one makes three lines) are a title and a read whose quality line is missing: the reference's record loop
    // takes such a record as soon as both newlines lie behind the first leftover byte (phyNGSC.cpp:303-312) and then
    // reads its quality past the end of the data.  Two bare blank lines do not qualify (the loop skips the first byte).
    if (tail_buf && blockIdx.x == n_ranks && threadIdx.x == 0 && (*total_lines & 3u) == 3u && tail_buf[ls[n_rec_total]] != '\n')
        atomicMin(err, (int32_t)ERR_FORMAT);
    if (blockIdx.x < n_ranks) {
        const uint32_t i = blockIdx.x;
        __shared__ PlanCache s_pc;
        const CachedSearch cs0{&s_pc, te, ls, 0, 0, WaveSearch(), tile_off, total_lines, n_tiles};
#ifdef NGSC_PLAN_TIMING
        const uint64_t tm0 = wall_clock64();
#endif
        const RankGeom g = rank_geometry(file_size, P, rank_lo + i);
        uint32_t nwin = 0;
        {
            // stop position of window k, were every window to start exactly where the one before stopped
            const uint64_t x0 = from.resume ? from.next_pos : (uint64_t)g.wr_start;   // (a rank's first window starts with its region)
            const uint64_t bs = from.resume ? from.bsize : (uint64_t)g.bsize, ov = from.resume ? from.overlap : g.overlap;
            const uint64_t D = bs > ov ? bs - ov : 0;
            const uint64_t n_ls = (uint64_t)n_rec_total + 1;
            const uint32_t last_ls = ls[n_rec_total];
            if (bs == WINDOW && D && x0 >= buf_off && n_rec_total > 256) {
                const uint64_t w = ((uint64_t)g.region + D - 1) / D + 1;
                nwin = w < PL_WIN ? (uint32_t)w : PL_WIN;
            }
            // (round 4) The look-ahead was 55 of the kernel's 95 us on the 1 GB job: four waves took four windows each, one after the
            // other, and a window's search by record density went wrong often enough to cost eight trips to memory.  Now a wave per
            // window, and the search starts from the index kernel's line count in front of the window's 64 KiB tile: the bound
            // lies among that tile's records (two 64-way rounds).  (Whatever is parked is checked when it is used: a wrong range
            // here costs a search later, never a wrong cut.)
            const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
            for (uint32_t k = wave; k < nwin; k += nwaves) {
                // (the rank's last window is cut short at the end of its region and stops an overlap in front of it - no overlap for
                // the file's last rank, plan_rank_with(): the stretch that would lie behind that place is parked around it)
                const uint64_t stop_last = (uint64_t)g.wr_end - (rank_lo + i == P - 1 ? 0u : g.overlap);
                uint64_t key_abs = x0 + (uint64_t)(k + 1) * D;
                if (key_abs > stop_last && stop_last > x0) key_abs = stop_last;
                const uint64_t key = key_abs - buf_off;
                uint32_t b0 = 0xffffffffu;
                if (key <= (uint64_t)last_ls) {
                    uint32_t r0, r1, lo, span;
                    if (cs0.bracket(key, lo, span)) {
                        WaveSearch().pair(ls, (uint32_t)n_ls, key, ls, (uint32_t)n_ls, key, lo, span, r0, r1);
                    } else {
                        // where the job's record density puts the bound; the probes of the first round lie around it
                        const uint64_t guess = key * n_rec_total / ((uint64_t)last_ls + 1);
                        WaveSearch().pair(ls, (uint32_t)n_ls, key, ls, (uint32_t)n_ls, key, 0, (uint32_t)n_ls, r0, r1, (uint32_t)guess);
                    }
                    b0 = r0 >= 8 ? r0 - 8 : 0;
                    for (uint32_t j = lane; j < PL_ENT; j += 64) {
                        const uint64_t e = (uint64_t)b0 + j;
                        s_pc.ls[k][j] = e < n_ls ? ls[e] : 0xffffffffu;
                        s_pc.te[k][j] = e < n_rec_total ? te[e] : 0xffffffffu;
                    }
                }
                if (lane == 0) s_pc.base[k] = b0;
            }
        }
        __syncthreads();
        if (threadIdx.x >= 64) return;   // one wave per rank runs the chain; all its lanes run the (uniform) planner, the searches are wave-wide
#ifdef NGSC_PLAN_TIMING
        const uint64_t tm1 = wall_clock64();
#endif
        CachedSearch cs{&s_pc, te, ls, nwin, 0, WaveSearch(), tile_off, total_lines, n_tiles};
        int rc = plan_rank_with(cs, file_size, P, rank_lo + i, buf_off, te, ls, n_rec_total, first_abs[i],
                                cuts + (size_t)i * max_cuts, max_cuts, &plans[i], &from, buf_end_abs);
#ifdef NGSC_PLAN_TIMING
        if (threadIdx.x == 0 && (i == 0 || i + 1 == n_ranks))
            printf("plan rank %u: nwin %u, cache fill %.2f us, chain %.2f us, cuts %u, searches not answered by the cache %u, loads %u\n", i, nwin,
                   (double)(tm1 - tm0) / 100.0, (double)(wall_clock64() - tm1) / 100.0, plans[i].n_cuts, cs.miss_pair, cs.miss_at);
#endif
        // ... and behind the first record of a window the loop does not skip that byte (phyNGSC.cpp:295-300): there two
        // blank lines qualify as well
        if (!rc && tail_buf && (*total_lines & 3u) == 3u && n_rec_total) {
            const uint32_t nc = plans[i].n_cuts;
            for (uint32_t k = 0; k < nc && k < max_cuts; ++k)
                if (cuts[(size_t)i * max_cuts + k].rec_lo + 1 == n_rec_total) rc = ERR_FORMAT;
        }
        if (threadIdx.x == 0) {
            plans[i].err = rc;
            if (rc) atomicMin(err, (int32_t)rc);
        }
        return;
    }
    const uint32_t vb = blockIdx.x - n_ranks, nvb = gridDim.x - n_ranks;
    validate_records(te, se, ls, n_rec_total, vb * blockDim.x + threadIdx.x, nvb * blockDim.x, max_q, max_rec, max_title, err);
}

// ---- launchers ---------------------------------------------------------------------------------------
uint32_t index_num_tiles(uint32_t len) { return len / TILE_BYTES + 1; }   // covers position len itself (assumed final newline)

// zero / zero_bytes: the step's small state in front of the per-rank words (a multiple of 4 bytes), cleared by this launch when the
// call has few ranks, by a launch of its own in front of it otherwise
void launch_sync_first(hipStream_t st, const uint8_t *buf, uint32_t len, uint64_t buf_off, uint64_t file_size, uint32_t P,
                       uint32_t rank_lo, uint32_t n_ranks, uint64_t *first_abs, uint32_t *start_rel, int32_t *err, uint64_t resume_abs,
                       void *zero, size_t zero_bytes)
{
    if (n_ranks <= SYNC_ONE_WG) {
        const uint32_t waves = n_ranks < SYNC_WAVES ? n_ranks : SYNC_WAVES;
        hipLaunchKernelGGL(k_sync_first, dim3(1), dim3(64 * waves), 0, st, buf, len, buf_off, file_size, P, rank_lo, n_ranks, first_abs,
                           start_rel, err, resume_abs, reinterpret_cast<uint32_t *>(zero), (uint32_t)(zero_bytes / 4));
        return;
    }
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, st, reinterpret_cast<uint32_t *>(zero), (uint32_t)(zero_bytes / 4));
    hipLaunchKernelGGL(k_sync_first, dim3(n_ranks), dim3(64), 0, st, buf, len, buf_off, file_size, P, rank_lo, n_ranks, first_abs,
                       start_rel, err, resume_abs, (uint32_t *)nullptr, 0u);
}

void launch_nl_count(hipStream_t st, const uint8_t *buf, uint32_t len, const uint32_t *start_rel, uint32_t *tile_cnt, uint32_t *nl_mask,
                     bool at_file_end)
{
    uint32_t nt = index_num_tiles(len);
    hipLaunchKernelGGL(k_nl_count, dim3(nt), dim3(TILE_THREADS), 0, st, buf, len, start_rel, tile_cnt, nl_mask, at_file_end ? 1u : 0u);
}

size_t index_mask_words(uint32_t len) { return (size_t)index_num_tiles(len) * (TILE_BYTES / BYTES_PER_THREAD); }

void launch_scan_tiles(hipStream_t st, uint32_t *tile_cnt, uint32_t n_tiles, uint32_t *total_lines)
{
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(1024), 0, st, tile_cnt, n_tiles, total_lines);
}

void launch_nl_write(hipStream_t st, const uint32_t *nl_mask, uint32_t len, const uint32_t *start_rel, const uint32_t *tile_off,
                     uint32_t *te, uint32_t *se, uint32_t *ls, uint32_t rec_cap)
{
    uint32_t nt = index_num_tiles(len);
    hipLaunchKernelGGL(k_nl_write, dim3(nt), dim3(TILE_THREADS), 0, st, nl_mask, start_rel, tile_off, te, se, ls, rec_cap);
}

void launch_cuts_validate(hipStream_t st, uint64_t file_size, uint32_t P, uint32_t rank_lo, uint32_t n_ranks, uint64_t buf_off,
                          const uint32_t *te, const uint32_t *se, const uint32_t *ls, uint32_t n_rec, const uint32_t *total_lines,
                          const uint64_t *first_abs, Cut *cuts, uint32_t max_cuts, RankPlan *plans, uint32_t *max_q,
                          uint32_t *max_rec, uint32_t *max_title, int32_t *err, const uint8_t *tail_buf, const PlanResume &from,
                          uint64_t buf_end_abs, const uint32_t *tile_off, uint32_t n_tiles)
{
    // ~8 records per validating thread (grid-stride, so any estimate works): few fat workgroups instead of one tiny
    // wave per 64 records
    uint32_t vblocks = (n_rec + 8 * CV_THREADS - 1) / (8 * CV_THREADS);
    if (vblocks < 1) vblocks = 1;
    if (vblocks > 4096) vblocks = 4096;
    hipLaunchKernelGGL(k_cuts_validate, dim3(n_ranks + vblocks), dim3(CV_THREADS), 0, st, file_size, P, rank_lo, n_ranks, buf_off, te, se,
                       ls, total_lines, first_abs, cuts, max_cuts, plans, max_q, max_rec, max_title, err, tail_buf, from, buf_end_abs,
                       tile_off, n_tiles);
}

}  // namespace ngsc
