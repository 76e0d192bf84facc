// kernels_pack.hip - bit-length measurement, offset scans, stream layout and the packed emit.
//
// Replaces the BitStream::PutBits hot loops of StoreQuality / StoreDNA / StoreTitle (tasks.cpp:609-619,
// 544-557, 427-507), the info header + length table (phyNGSC.cpp:719-742) and the stream concatenation
// (phyNGSC.cpp:793-840).  Output bytes are produced directly at their final place:
//   sub-block = info | title | quality | DNA, sub-blocks back to back in (rank, sequence) order.
//
// Quality / DNA: one wavefront per record, one lane per symbol (coalesced byte loads).  Code tables
// live in LDS ((max_len+1) x nqual words).  Bit offsets come from a wave inclusive scan of the code
// lengths on top of the record's offset (workgroup scan of per-record bit counts).  Symbols are merged
// into whole big-endian 32-bit words by a segmented OR-scan across lanes; complete interior words are
// stored plainly, the first/last word of every 64-symbol step is merged with atomicOr.  HBM-bound
// integer work, no MFMA: algorithmic traffic per record = its bytes read once + its code bits written.
#include "ngsc_kernels.h"
#include <algorithm>

namespace ngsc {

constexpr int PK_THREADS = 256;
constexpr int PK_WAVES = PK_THREADS / 64;
constexpr uint32_t PK_REC_PER_WG = 1024;
constexpr uint32_t PK_LDS_MAX_ENTRIES = 24576;  // 96 KB of quality code table

__device__ __forceinline__ uint32_t amb_fast2(uint8_t c)
{
    return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? 1u : amb_code(c);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

__device__ __forceinline__ uint32_t wave_scan_incl_u32(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// Shared table setup for measure and pack.
struct SqTables {
    uint32_t *qtab;      // LDS [(max_q+1)][nq] or nullptr
    const uint32_t *qglob;  // global [(qrows)][256]
    uint32_t nq;
    bool in_lds;
};

__device__ __forceinline__ uint32_t qentry(const SqTables &T, uint32_t row, uint32_t idx)
{
    return T.in_lds ? T.qtab[row * T.nq + idx] : T.qglob[(size_t)row * 256 + idx];
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PK_THREADS) void k_measure_sq(Arena A, uint32_t lds_entries)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_qtab[];
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t slot_bias = sb.slot_bias;   // (read once: the stores below could alias A.sb)
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t r_begin = blockIdx.x * PK_REC_PER_WG;
    if (r_begin >= n) return;
    const uint32_t r_end = r_begin + PK_REC_PER_WG < n ? r_begin + PK_REC_PER_WG : n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *buf = A.buf;
    SqTables T;
    T.nq = sb.nqual;
    T.qglob = A.qcodes + (size_t)s * A.qrows * 256;
    T.qtab = s_qtab;
    const uint32_t entries = (sb.max_q + 1) * T.nq;
    T.in_lds = entries <= lds_entries;
    if (T.in_lds)
        for (uint32_t k = tid; k < entries; k += PK_THREADS) s_qtab[k] = T.qglob[(size_t)(k / T.nq) * 256 + (k % T.nq)];
    s_dcode[tid] = A.dcodes[(size_t)s * 256 + tid];
    s_qcode[tid] = sb.qua_code[tid];
    __syncthreads();
    for (uint32_t r = r_begin + wave; r < r_end; r += PK_WAVES) {
        const uint32_t g = rec_lo + r;
        const uint32_t t = A.te[g], e = A.se[g];
        const uint32_t L = e - t - 1, s0 = t + 1, q0 = e + 3;
        const bool xfer = A.slen[g] >> 31;
        uint32_t qb = 0, db = 0;
        for (uint32_t base = 0; base < L; base += 64) {
            uint32_t j = base + lane;
            if (j < L) {
                uint8_t c = buf[s0 + j], q = buf[q0 + j];
                uint32_t a = xfer ? amb_fast2(c) : 1u;
                bool drop = xfer && a > 1;
                uint32_t qv = drop ? transfer_quality(a, q) : q;
                qb += qentry(T, j + 1, s_qcode[qv]) & 31;
                if (!drop) db += s_dcode[c] & 31;
            }
        }
        qb = wave_sum(qb);
        db = wave_sum(db);
        if (lane == 0) { A.qoff[g + slot_bias] = qb; A.doff[g + slot_bias] = db; }
    }
}

// ---------------------------------------------------------------------------------------------------
// In-place exclusive scan of a[0..n) by one workgroup of 1024 threads; returns the total (all threads).
__device__ uint32_t wg_scan_excl(uint32_t *a, uint32_t n, uint32_t *s_wave /*[17]*/)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t run = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + tid;
        uint32_t v = i < n ? a[i] : 0;
        uint32_t incl = wave_scan_incl_u32(v);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        if (tid < 64) {
            uint32_t w = tid < 16 ? s_wave[tid] : 0;
            uint32_t sc = wave_scan_incl_u32(w);
            if (tid < 16) s_wave[tid] = sc - w;
            if (tid == 15) s_wave[16] = sc;
        }
        __syncthreads();
        if (i < n) a[i] = run + s_wave[wave] + incl - v;
        run += s_wave[16];
        __syncthreads();
    }
    return run;
}

__global__ __launch_bounds__(1024) void k_scan_sb(Arena A)
{
    if (*A.err) return;
    __shared__ uint32_t s_wave[17];
    const uint32_t s = blockIdx.x;
    SbState &sb = A.sb[s];
    const uint32_t slot_bias = sb.slot_bias;   // (read once: the stores below could alias A.sb)
    uint32_t tq = wg_scan_excl(A.qoff + sb.rec_lo + slot_bias, sb.n_rec, s_wave);
    uint32_t td = wg_scan_excl(A.doff + sb.rec_lo + slot_bias, sb.n_rec, s_wave);
    uint32_t tg = wg_scan_excl(A.goff + A.grp_lo[s], (sb.n_rec + GROUP - 1) / GROUP, s_wave);
    if (threadIdx.x == 0) { sb.qbits = tq; sb.dbits = td; sb.title_body = tg; }
}

// The same for the lane-per-record kernels (kernels_sq.hip), which leave payload bits per tile of R records.
__global__ __launch_bounds__(1024) void k_scan_tile(Arena A, uint32_t R)
{
    if (*A.err) return;
    __shared__ uint32_t s_wave[17];
    const uint32_t s = blockIdx.x;
    SbState &sb = A.sb[s];
    const uint32_t ntiles = (sb.n_rec + R - 1) / R;
    uint32_t tq = wg_scan_excl(A.wq + A.tile_lo[s], ntiles, s_wave);
    uint32_t td = wg_scan_excl(A.wd + A.tile_lo[s], ntiles, s_wave);
    uint32_t tg = wg_scan_excl(A.goff + A.grp_lo[s], (sb.n_rec + GROUP - 1) / GROUP, s_wave);
    if (threadIdx.x == 0) { sb.qbits = tq; sb.dbits = td; sb.title_body = tg; }
}

// One wave per sub-block: stream sizes.  qrow_off[s][row] = byte offset of tree `row` inside the quality header (a
// prefix sum over up to max_q + 1 tree blobs: as a serial loop of one thread it was 150 dependent loads for 150-base reads).
__device__ __forceinline__ void sizes_wave(const Arena &A, uint32_t s, uint32_t lane, uint32_t *qrow_off)
{
    SbState &sb = A.sb[s];
    const uint32_t rows = sb.max_q + 1;
    uint32_t off = sb.nqual;  // alphabet bytes (tasks.cpp:577-581)
    for (uint32_t base = 0; base < rows; base += 64) {
        const uint32_t row = base + lane;
        const uint32_t v = row < rows ? 4 + A.qblob_len[(size_t)s * A.qrows + row] : 0u;
        const uint32_t incl = wave_scan_incl_u32(v);
        if (row < rows) qrow_off[(size_t)s * A.qrows + row] = off + incl - v;
        off += (uint32_t)__shfl((int)incl, 63);
    }
    if (lane) return;
    sb.qual_hdr = off;
    sb.qual_len = off + (sb.qbits + 7) / 8;
    uint32_t dh = sb.nsym;
    if (!(sb.flags & F_DNA_PLAIN)) dh += 4 + A.dblob_len[s];
    sb.dna_hdr = dh;
    sb.dna_len = dh + (sb.dbits + 7) / 8;
    BitCounter bc;
    title_header(A, s, bc);
    sb.title_hdr = (uint32_t)bc.nbytes();
    sb.title_len = sb.title_hdr + sb.title_body;
    sb.info_len = 19 + (uint32_t)(((uint64_t)sb.n_rec * sb.lt_bits + 7) / 8);  // phyNGSC.cpp:719-742
    sb.total = sb.info_len + sb.title_len + sb.qual_len + sb.dna_len;
}

__global__ __launch_bounds__(64) void k_sizes(Arena A, uint32_t *qrow_off)
{
    if (*A.err) return;
    if (blockIdx.x >= A.n_sb) return;
    sizes_wave(A, blockIdx.x, threadIdx.x, qrow_off);
}

// One wave: exclusive prefix sum of the sub-block sizes = their offsets in the output.
// (fresh: the totals were written by other workgroups of the same launch - read them past the L1)
template <bool FRESH>
__device__ __forceinline__ uint64_t offsets_wave(const Arena &A, uint32_t lane, uint64_t out_cap, uint64_t *total_out)
{
    uint64_t off = 0;
    for (uint32_t base = 0; base < A.n_sb; base += 64) {
        const uint32_t s = base + lane;
        uint64_t v = 0;
        if (s < A.n_sb) v = FRESH ? (uint64_t)__hip_atomic_load(&A.sb[s].total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (uint64_t)A.sb[s].total;
        uint64_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t o = __shfl_up(incl, d);
            if ((int)lane >= d) incl += o;
        }
        if (s < A.n_sb) A.sb[s].out_off = off + incl - v;
        off += __shfl(incl, 63);
    }
    if (lane == 0) {
        *total_out = off;
        if (off > out_cap) atomicMin(A.err, (int32_t)ERR_CAPACITY);
    }
    return off;
}

__global__ __launch_bounds__(64) void k_offsets(Arena A, uint64_t out_cap, uint64_t *total_out)
{
    if (*A.err) return;
    if (blockIdx.x) return;
    offsets_wave<false>(A, threadIdx.x, out_cap, total_out);
}

// (round 3) k_scan_tile, k_sizes and k_offsets as ONE launch for the lane-per-record path - three tiny kernels in a row on the
// step's critical path cost more in launch gaps than in work: a workgroup scans its sub-block's tile and group totals, its first
// wave derives the stream sizes, and the workgroup that finishes last (a counter in the step's small state) lays the
// sub-blocks out.
constexpr uint64_t LAYOUT_CLEARS = 1ull << 20;   // bytes one wave clears in passing (a few us per 100 KB)
// zero_out / cleared (a predicted step): the output buffer and the bytes of it - a multiple of 16 - that were cleared early; what the
// sub-blocks take beyond that is cleared here when it is little (the step then needs no k_zero_output), and is a missed prediction
// when it is not.
__global__ __launch_bounds__(1024) void k_layout(Arena A, uint32_t R, uint32_t *qrow_off, uint64_t out_cap, uint64_t *total_out, uint32_t *done,
                                                uint4 *zero_out, uint64_t cleared)
{
    if (*A.err) return;
    __shared__ uint32_t s_wave[17];
    const uint32_t s = blockIdx.x;
    {
        SbState &sb = A.sb[s];
        const uint32_t ntiles = (sb.n_rec + R - 1) / R;
        const uint32_t tq = wg_scan_excl(A.wq + A.tile_lo[s], ntiles, s_wave);
        const uint32_t td = wg_scan_excl(A.wd + A.tile_lo[s], ntiles, s_wave);
        const uint32_t tg = wg_scan_excl(A.goff + A.grp_lo[s], (sb.n_rec + GROUP - 1) / GROUP, s_wave);
        if (threadIdx.x == 0) { sb.qbits = tq; sb.dbits = td; sb.title_body = tg; }
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;   // the rest is one wave's work (no barrier behind this line)
    sizes_wave(A, s, threadIdx.x, qrow_off);
    __threadfence();
    uint32_t last = 0;
    if (threadIdx.x == 0) last = atomicAdd(done, 1u) == gridDim.x - 1 ? 1u : 0u;
    last = (uint32_t)__shfl((int)last, 0);
    if (!last) return;
    __threadfence();
    const uint64_t total = offsets_wave<true>(A, threadIdx.x, out_cap, total_out);
    if (zero_out && total > cleared && total <= out_cap) {
        if (total - cleared <= LAYOUT_CLEARS) {
            for (uint64_t k = (cleared >> 4) + threadIdx.x; k < (total + 15) >> 4; k += 64) zero_out[k] = make_uint4(0, 0, 0, 0);
        }
        else if (threadIdx.x == 0) atomicMin(A.err, (int32_t)ERR_SHAPE);
    }
}

// ---------------------------------------------------------------------------------------------------
// Headers: info (19 B), title header, quality alphabet + trees, DNA alphabet + tree.
// (round 4) The title header - field after field, a text field with up to 129 tree blobs of ~300 bytes - was ONE thread's
// byte-by-byte walk (84 us per launch on titles with text fields, longer than the title pack kernel it runs beside).  Its
// per-field parts are byte aligned (ngsc_common.h: title_header_field): a thread per field measures its part, a scan places
// it, the thread writes it - and byte runs of 32 bytes and more (constant fields, tree blobs) are only queued, for the whole
// workgroup to copy afterwards.
struct HdrCopy { uint32_t dst, len; const uint8_t *src; };
constexpr uint32_t HDR_JOBS = 640;
struct DeferWriter {   // ByteWriter whose long byte runs go to the workgroup's copy queue
    ByteWriter w;
    uint8_t *base;
    HdrCopy *jobs;
    uint32_t *n_jobs;
    __device__ DeferWriter(uint8_t *dst, uint8_t *b, HdrCopy *q, uint32_t *qn) : w(dst), base(b), jobs(q), n_jobs(qn) {}
    __device__ void put(uint32_t v, uint32_t nb) { w.put(v, nb); }
    __device__ void align() { w.align(); }
    __device__ void byte(uint8_t b) { w.byte(b); }
    __device__ void word(uint32_t v) { w.word(v); }
    __device__ void bytes(const uint8_t *d, uint32_t len)
    {
        if (len >= 32) {
            const uint32_t k = atomicAdd(n_jobs, 1u);
            if (k < HDR_JOBS) {
                w.align();
                jobs[k] = HdrCopy{(uint32_t)((w.p + w.n) - base), len, d};
                w.n += len;
                return;
            }
        }
        w.bytes(d, len);
    }
};

__global__ __launch_bounds__(256) void k_write_headers(Arena A, const uint32_t *qrow_off)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.x;
    const SbState &sb = A.sb[s];
    const int tid = threadIdx.x;
    uint8_t *o = A.out + sb.out_off;
    __shared__ HdrCopy s_jobs[HDR_JOBS];
    __shared__ uint32_t s_njobs, s_wsum[4];
    if (tid == 0) s_njobs = 0;
    __syncthreads();
    if (tid == 0) {
        ByteWriter w(o);  // phyNGSC.cpp:719-731
        w.word(sb.n_rec); w.word(sb.max_q); w.word(sb.max_s);
        w.byte((uint8_t)sb.nsym); w.byte(0 /* QUALITY_PLAIN */); w.byte((uint8_t)sb.nqual);
        w.word(sb.flags);
    } else if (tid == 128) {
        ByteWriter w(o + sb.info_len + sb.title_len + sb.qual_len);  // tasks.cpp:519-542
        w.bytes(sb.symbols, sb.nsym);
        if (!(sb.flags & F_DNA_PLAIN)) {
            w.word(A.dblob_len[s]);
            w.bytes(A.dblob + (size_t)s * BLOB_SLOT_256, A.dblob_len[s]);
        }
    }
    // title header (tasks.cpp:302-390): fields tid, tid + 256, ... of the sub-block (more than 256: the wide title path)
    {
        uint8_t *th = o + sb.info_len;
        uint32_t carry = 4;   // (the field count in front)
        if (tid == 0) { ByteWriter w(th); w.word(sb.nf); }
        for (uint32_t f0 = 0; f0 < sb.nf; f0 += 256) {
            const uint32_t f = f0 + tid;
            uint32_t size = 0;
            if (f < sb.nf) { BitCounter bc; title_header_field(A, s, f, bc); size = (uint32_t)bc.nbytes(); }
            uint32_t incl = wave_scan_incl_u32(size);
            if ((tid & 63) == 63) s_wsum[tid >> 6] = incl;
            __syncthreads();
            uint32_t before = 0, total = 0;
            for (int w = 0; w < 4; ++w) { if (w < (tid >> 6)) before += s_wsum[w]; total += s_wsum[w]; }
            if (f < sb.nf) {
                DeferWriter w(th + carry + before + incl - size, o, s_jobs, &s_njobs);
                title_header_field(A, s, f, w);
            }
            carry += total;
            __syncthreads();
        }
    }
    __syncthreads();
    {   // the queued byte runs: a wave each, 64 bytes a step
        const uint32_t nj = s_njobs < HDR_JOBS ? s_njobs : HDR_JOBS;
        for (uint32_t k = (uint32_t)tid >> 6; k < nj; k += 4) {
            const HdrCopy j = s_jobs[k];
            for (uint32_t x = (uint32_t)tid & 63u; x < j.len; x += 64) o[j.dst + x] = j.src[x];
        }
    }
    // quality header: alphabet, then one tree per row (tasks.cpp:577-605); threads stride over rows
    uint8_t *q = o + sb.info_len + sb.title_len;
    for (uint32_t k = tid; k < sb.nqual; k += 256) q[k] = sb.qualities[k];
    for (uint32_t row = tid; row <= sb.max_q; row += 256) {
        size_t t = (size_t)s * A.qrows + row;
        ByteWriter w(q + qrow_off[t]);
        w.word(A.qblob_len[t]);
        w.bytes(A.qblob + t * BLOB_SLOT_256, A.qblob_len[t]);
    }
}

// Length table: n_rec values of lt_bits bits each (phyNGSC.cpp:733-742).  One thread per aligned 32-bit
// word of the output buffer that intersects the table.
__global__ __launch_bounds__(256) void k_pack_lentable(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t b = sb.lt_bits;
    if (!b) return;
    const uint64_t t0 = (sb.out_off + 19) * 8, t1 = t0 + (uint64_t)sb.n_rec * b;  // absolute bit range
    const uint64_t w0 = t0 >> 5;
    const uint64_t w = w0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((w << 5) >= t1) return;
    const uint64_t lo = (w << 5) > t0 ? (w << 5) : t0, hi = ((w + 1) << 5) < t1 ? ((w + 1) << 5) : t1;
    uint32_t val = 0;
    uint64_t r = (lo - t0) / b;
    for (;; ++r) {
        uint64_t rb = t0 + r * b;  // first bit of record r
        if (rb >= hi) break;
        uint32_t g = sb.rec_lo + (uint32_t)r;
        uint64_t v = A.se[g] - A.te[g] - 1;  // qua_len
        // place v's b bits at absolute bits [rb, rb+b); keep those inside [w<<5, (w+1)<<5)
        int64_t sh = (int64_t)((w + 1) << 5) - (int64_t)(rb + b);  // left shift of v's LSB inside the word
        uint64_t x = sh >= 0 ? (v << sh) : (v >> (-sh));
        val |= (uint32_t)x;
    }
    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out) + w;
    uint32_t be = __builtin_bswap32(val);
    if (lo == (w << 5) && hi == ((w + 1) << 5)) *outw = be; else if (be) atomicOr(outw, be);
}

// ---------------------------------------------------------------------------------------------------
// Wave-cooperative emit of up to 64 variable-length codes starting at absolute bit `bitpos`.
// Every lane passes (code, len) with len <= 26 (0 for "nothing").  Returns the number of bits emitted.
__device__ __forceinline__ uint32_t wave_emit(uint32_t *outw, uint64_t bitpos, uint32_t code, uint32_t len)
{
    const int lane = threadIdx.x & 63;
    const uint32_t incl = wave_scan_incl_u32(len);
    const uint32_t total = __shfl(incl, 63);
    if (!total) return 0;
    const uint64_t o = bitpos + (incl - len);
    const uint32_t key = (uint32_t)(o >> 5), sh = (uint32_t)(o & 31);
    uint32_t hi = 0, lo = 0;
    if (len) {
        uint64_t x = (uint64_t)code << (64 - len - sh);
        hi = (uint32_t)(x >> 32);
        lo = (uint32_t)x;  // spill into word key+1 (non-zero only when sh + len > 32)
    }
    // a spill always lands in the word the next lane starts in (len <= 32): hand it to that lane
    const uint32_t key_prev = __shfl_up(key, 1), lo_prev = __shfl_up(lo, 1);
    if (lane > 0 && key_prev != key) hi |= lo_prev;
    // segmented inclusive OR-scan keyed by word index (keys are non-decreasing across lanes)
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t ko = __shfl_up(key, d), vo = __shfl_up(hi, d);
        if (lane >= d && ko == key) hi |= vo;
    }
    const uint32_t key_next = __shfl_down(key, 1);
    const bool tail = lane == 63 || key_next != key;
    const uint32_t first_key = __shfl(key, 0), last_key = __shfl(key, 63);
    if (tail) {
        uint32_t be = __builtin_bswap32(hi);
        if (key != first_key && key != last_key) outw[key] = be;     // complete interior word
        else if (be) atomicOr(&outw[key], be);                        // shared with the neighbours
    }
    if (lane == 63 && lo) atomicOr(&outw[key + 1], __builtin_bswap32(lo));
    return total;
}

__global__ __launch_bounds__(PK_THREADS) void k_pack_sq(Arena A, uint32_t lds_entries)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_qtab[];
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t slot_bias = sb.slot_bias;   // (read once: the stores below could alias A.sb)
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t r_begin = blockIdx.x * PK_REC_PER_WG;
    if (r_begin >= n) return;
    const uint32_t r_end = r_begin + PK_REC_PER_WG < n ? r_begin + PK_REC_PER_WG : n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *buf = A.buf;
    SqTables T;
    T.nq = sb.nqual;
    T.qglob = A.qcodes + (size_t)s * A.qrows * 256;
    T.qtab = s_qtab;
    const uint32_t entries = (sb.max_q + 1) * T.nq;
    T.in_lds = entries <= lds_entries;
    if (T.in_lds)
        for (uint32_t k = tid; k < entries; k += PK_THREADS) s_qtab[k] = T.qglob[(size_t)(k / T.nq) * 256 + (k % T.nq)];
    s_dcode[tid] = A.dcodes[(size_t)s * 256 + tid];
    s_qcode[tid] = sb.qua_code[tid];
    __syncthreads();
    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out);
    const uint64_t qbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_hdr) * 8;
    const uint64_t dbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_len + sb.dna_hdr) * 8;
    for (uint32_t r = r_begin + wave; r < r_end; r += PK_WAVES) {
        const uint32_t g = rec_lo + r;
        const uint32_t t = A.te[g], e = A.se[g];
        const uint32_t L = e - t - 1, s0 = t + 1, q0 = e + 3;
        const bool xfer = A.slen[g] >> 31;
        uint64_t qpos = qbase + A.qoff[g + slot_bias], dpos = dbase + A.doff[g + slot_bias];
        for (uint32_t base = 0; base < L; base += 64) {
            uint32_t j = base + lane;
            uint32_t qe = 0, de = 0;
            if (j < L) {
                uint8_t c = buf[s0 + j], q = buf[q0 + j];
                uint32_t a = xfer ? amb_fast2(c) : 1u;
                bool drop = xfer && a > 1;
                uint32_t qv = drop ? transfer_quality(a, q) : q;
                qe = qentry(T, j + 1, s_qcode[qv]);       // tasks.cpp:614-618
                if (!drop) de = s_dcode[c];               // tasks.cpp:549-556
            }
            qpos += wave_emit(outw, qpos, qe >> 5, qe & 31);
            dpos += wave_emit(outw, dpos, de >> 5, de & 31);
        }
    }
}

// Zeroes the part of the output buffer that the sub-blocks will occupy (its size is on the device: no host round trip).
// Shared words - the first / last word of a tile, a record, a stream - are merged with atomicOr by the pack kernels.
// from16: 16-byte units at the front that are zero already (cleared early by k_zero_range, beside the index kernels)
__global__ __launch_bounds__(256) void k_zero_output(uint4 *out, const uint64_t *total_ptr, uint64_t out_cap, uint64_t from16)
{
    uint64_t total = *total_ptr;
    if (total > out_cap) total = out_cap;
    const uint64_t n16 = (total + 15) >> 4;
    for (uint64_t k = from16 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (uint64_t)gridDim.x * blockDim.x)
        out[k] = make_uint4(0, 0, 0, 0);
}

__global__ __launch_bounds__(256) void k_zero_range(uint4 *out, uint64_t n16)
{
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += (uint64_t)gridDim.x * blockDim.x)
        out[k] = make_uint4(0, 0, 0, 0);
}

// SbState of every sub-block (rec_lo, n_rec, rank, slot_bias; the rest is zero already), its first title group and its first tile, from
// the cut list the planner left on the device (engine.cpp built these on the host and copied them over until round 3).
// One workgroup; a thread takes a cut: totals, workgroup scan, then the cuts again.
// It also holds the job against the shape its arrays and grids were sized for (caps): the step may have been enqueued from a
// PREDICTION of the planner's counts and the job's maxima (engine.cpp: JobShape) - another number of sub-blocks, more records,
// slots, groups or tiles than there is room for, or a longer title / read / record than the kernels were chosen for raise
// ERR_SHAPE, every kernel behind this one leaves at once, and the host runs the step again with the job's own numbers.
constexpr uint32_t INIT_THREADS = 256;   // (a job of eight ranks has ~300 cut slots: two rounds.  Not 1024: a workgroup of sixteen waves waits for a CU with that much room while the early clearing runs beside it - the 1 GB step lost 30 us that way)
__global__ __launch_bounds__(INIT_THREADS) void k_init_sb(Arena A, const Cut *cuts, const RankPlan *plans, uint32_t n_ranks, uint32_t max_cuts, uint32_t R,
                                                 ShapeCaps caps, const uint32_t *total_lines, const uint32_t *max_title, const uint32_t *max_q,
                                                 const uint32_t *max_rec_bytes)
{
    if (*A.err) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    __shared__ uint32_t s_bad;
    if (tid == 0) {
        const uint32_t n_rec_total = *total_lines / 4;
        s_bad = (n_rec_total > caps.n_rec_total || (uint64_t)n_rec_total + 2 > caps.rec_bound || *max_title > caps.max_title || *max_q > caps.max_q ||
                 *max_rec_bytes > caps.max_rec_bytes) ? 1u : 0u;
    }
    __syncthreads();
    // (round 4) a thread per CUT SLOT (rank i's k-th cut = cuts[i * max_cuts + k]; slots behind a rank's count are empty) instead of a
    // thread per rank walking its cuts: eight threads with 15 cuts each were two chains of dependent loads, 14 us between the planner
    // and the first statistics kernel of a 1 GB job
    const uint32_t n_slot = n_ranks * max_cuts;
    auto load = [&](uint32_t j, uint32_t v[4]) -> bool {   // v = {1, records, title groups, tiles} of slot j, zeros if it is empty
        v[0] = v[1] = v[2] = v[3] = 0;
        if (j >= n_slot) return false;
        const uint32_t i = j / max_cuts, k = j - i * max_cuts;
        const uint32_t nc = plans[i].n_cuts < max_cuts ? plans[i].n_cuts : max_cuts;
        if (k >= nc) return false;
        const uint32_t n = cuts[j].n_rec;
        v[0] = 1; v[1] = n; v[2] = (n + GROUP - 1) / GROUP; v[3] = R ? (n + R - 1) / R : 0;
        return true;
    };
    {   // the planner's counts, before anything is written
        uint32_t s_cnt = 0, mr = 0;
        uint64_t sl = 0, gr = 0, tl = 0;
        for (uint32_t j = tid; j < n_slot; j += INIT_THREADS) {
            uint32_t v[4];
            load(j, v);
            s_cnt += v[0];
            mr = v[1] > mr ? v[1] : mr;
            sl += v[1]; gr += v[2]; tl += v[3];
        }
        __shared__ uint32_t s_S;
        __shared__ unsigned long long s_sl, s_gr, s_tl;
        if (tid == 0) { s_S = 0; s_sl = s_gr = s_tl = 0; }
        __syncthreads();
        if (s_cnt) atomicAdd(&s_S, s_cnt);
        if (sl) { atomicAdd(&s_sl, (unsigned long long)sl); atomicAdd(&s_gr, (unsigned long long)gr); atomicAdd(&s_tl, (unsigned long long)tl); }
        if (mr > caps.max_rec) atomicOr(&s_bad, 1u);
        __syncthreads();
        if (tid == 0 && (s_S != caps.S || s_sl > caps.slots || s_gr > caps.groups || (R && s_tl > caps.tiles))) s_bad = 1u;
        __syncthreads();
        if (s_bad) { if (tid == 0) atomicMin(A.err, (int32_t)ERR_SHAPE); return; }
    }
    // (the SbState array lies in the arena the engine clears for every call: only the non-zero words are written here)
    __shared__ uint32_t s_w[4][INIT_THREADS / 64];
    uint32_t base[4] = {0, 0, 0, 0};   // sub-blocks, record slots, title groups, tiles of the slots in front (workgroup-uniform)
    for (uint32_t j0 = 0; j0 < n_slot; j0 += INIT_THREADS) {
        const uint32_t j = j0 + tid;
        uint32_t v[4];
        const bool have = load(j, v);
        uint32_t ex[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            uint32_t incl = v[q];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if ((int)lane >= d) incl += o;
            }
            if (lane == 63) s_w[q][wave] = incl;
            ex[q] = incl - v[q];
        }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < 4; ++q) {
            uint32_t tot = 0;
#pragma unroll
            for (uint32_t w = 0; w < INIT_THREADS / 64; ++w) {
                if (w < wave) ex[q] += s_w[q][w];
                tot += s_w[q][w];
            }
            ex[q] += base[q];
            base[q] += tot;
        }
        __syncthreads();   // s_w is written again in the next round
        if (have && ex[0] < A.n_sb) {
            const Cut c = cuts[j];
            SbState &x = A.sb[ex[0]];
            x.rec_lo = c.rec_lo;
            x.n_rec = c.n_rec;
            x.rank = c.rank;
            x.slot_bias = ex[1] - c.rec_lo;   // (mod 2^32)
            A.grp_lo[ex[0]] = ex[2];
            A.tile_lo[ex[0]] = ex[3];
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
void launch_zero_output(hipStream_t st, void *d_out, const uint64_t *total_ptr, uint64_t out_cap, uint64_t cleared)
{
    // (a step whose front is clear already usually has nothing left to clear: a small grid then)
    hipLaunchKernelGGL(k_zero_output, dim3(cleared ? 256 : 4096), dim3(256), 0, st, reinterpret_cast<uint4 *>(d_out), total_ptr, out_cap & ~15ull,
                       cleared >> 4);
}

void launch_zero_range(hipStream_t st, void *d_out, uint64_t bytes)   // bytes: a multiple of 16
{
    const uint64_t n16 = bytes >> 4;
    if (!n16) return;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(2048, (n16 + 255) / 256);
    hipLaunchKernelGGL(k_zero_range, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint4 *>(d_out), n16);
}

void launch_init_sb(hipStream_t st, const Arena &A, const Cut *cuts, const RankPlan *plans, uint32_t n_ranks, uint32_t max_cuts, uint32_t R,
                    const ShapeCaps &caps, const uint32_t *total_lines, const uint32_t *max_title, const uint32_t *max_q, const uint32_t *max_rec_bytes)
{
    hipLaunchKernelGGL(k_init_sb, dim3(1), dim3(INIT_THREADS), 0, st, A, cuts, plans, n_ranks, max_cuts, R, caps, total_lines, max_title, max_q, max_rec_bytes);
}

static uint32_t lds_entries_for(uint32_t qrows)
{
    uint64_t e = (uint64_t)qrows * 96;
    return (uint32_t)(e < PK_LDS_MAX_ENTRIES ? e : PK_LDS_MAX_ENTRIES);
}

void launch_measure_sq(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    uint32_t le = lds_entries_for(A.qrows);
    dim3 grid((max_rec + PK_REC_PER_WG - 1) / PK_REC_PER_WG, A.n_sb);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_measure_sq), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    hipLaunchKernelGGL(k_measure_sq, grid, dim3(PK_THREADS), le * sizeof(uint32_t), st, A, le);
}

void launch_scan_sb(hipStream_t st, const Arena &A)
{
    hipLaunchKernelGGL(k_scan_sb, dim3(A.n_sb), dim3(1024), 0, st, A);
}

void launch_scan_tile(hipStream_t st, const Arena &A, uint32_t R)
{
    hipLaunchKernelGGL(k_scan_tile, dim3(A.n_sb), dim3(1024), 0, st, A, R);
}

void launch_layout(hipStream_t st, const Arena &A, uint32_t R, uint32_t *qrow_off, uint64_t out_cap, uint64_t *total_out, uint32_t *done,
                   void *zero_out, uint64_t cleared)
{
    hipLaunchKernelGGL(k_layout, dim3(A.n_sb), dim3(1024), 0, st, A, R, qrow_off, out_cap, total_out, done, reinterpret_cast<uint4 *>(zero_out),
                       cleared);
}

void launch_sizes_offsets(hipStream_t st, const Arena &A, uint32_t *qrow_off, uint64_t out_cap, uint64_t *total_out)
{
    hipLaunchKernelGGL(k_sizes, dim3(A.n_sb), dim3(64), 0, st, A, qrow_off);
    hipLaunchKernelGGL(k_offsets, dim3(1), dim3(64), 0, st, A, out_cap, total_out);
}

void launch_write_headers(hipStream_t st, const Arena &A, const uint32_t *qrow_off)
{
    hipLaunchKernelGGL(k_write_headers, dim3(A.n_sb), dim3(256), 0, st, A, qrow_off);
}

void launch_pack_lentable(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    uint32_t lt_words = max_rec + 2;  // at most 32 bits per record, plus misalignment
    hipLaunchKernelGGL(k_pack_lentable, dim3((lt_words + 255) / 256, A.n_sb), dim3(256), 0, st, A);
}

void launch_pack_sq(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    uint32_t le = lds_entries_for(A.qrows);
    dim3 grid((max_rec + PK_REC_PER_WG - 1) / PK_REC_PER_WG, A.n_sb);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_pack_sq), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    hipLaunchKernelGGL(k_pack_sq, grid, dim3(PK_THREADS), le * sizeof(uint32_t), st, A, le);
}

}  // namespace ngsc
