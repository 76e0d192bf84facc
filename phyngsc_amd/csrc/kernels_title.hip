// kernels_title.hip - title path with the bytes in registers (no LDS tile): one thread per record reads its
// own title with aligned 16-byte global loads (double buffered) and parses it byte by byte out of VGPRs.
// LDS holds only a few KB of reduction state, so the kernels run at full occupancy (32 waves / CU); the LDS
// tile versions of the same logic were latency bound at 8-12 waves / CU.
//
//   k_rec0_fields        field table of each sub-block's first record            (phyNGSC.cpp:342-379)
//   k_title_stats_reg    AnalyzeTitleFields reductions                           (tasks.cpp:22-193)
//   k_title_measure_reg  caches the non-constant fields of every record, derives the 32-record group flags
//                        and the title bits per record                           (tasks.cpp:393-509)
//   k_title_pack_reg     StoreTitle body from the cached fields; every lane ORs its bits into an LDS staging
//                        area, one coalesced copy-out per 256 records            (tasks.cpp:427-507)
// Algorithmic traffic: the title bytes once per kernel (stats, measure) + 4 B per cached field and record.
#include "ngsc_kernels.h"
#include <cstring>
#include <cstdlib>
#include "ngsc_swar.h"

namespace ngsc {

constexpr int TT_THREADS = 256;

__device__ __forceinline__ uint32_t t_wmax(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t t_g32_or(uint32_t v)
{
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v |= __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ fmask_t t_g32_or(fmask_t v)   // (field masks: the high word only when some title has more than 32 fields)
{
    uint32_t lo = t_g32_or((uint32_t)v), hi = (uint32_t)(v >> 32);
    if (__any(hi != 0)) hi = t_g32_or(hi);
    return ((fmask_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t t_g32_sum(uint32_t v)
{
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ void t_lds_or(uint32_t *p, uint32_t v) { if ((*p & v) != v) atomicOr(p, v); }

// branch-free membership test for the title separators " ._,=:/-#\n" (phyNGSC.cpp:208)
__device__ __forceinline__ uint32_t t_sep(uint32_t c)
{
    const uint32_t m0 = 1u << 10;
    const uint32_t m1 = (1u << 0) | (1u << 3) | (1u << 12) | (1u << 13) | (1u << 14) | (1u << 15) | (1u << 26) | (1u << 29);
    const uint32_t m2 = 1u << 31;
    const uint32_t w = c >> 5;
    const uint32_t m = w == 0 ? m0 : (w == 1 ? m1 : (w == 2 ? m2 : 0u));
    return (m >> (c & 31)) & 1u;
}

// Calls body(c, act, p) for 16 bytes per chunk of the title [ls, te] of this lane's record; every lane of the
// wave makes the same number of calls (act == 0 outside its own title).  The next chunk is loaded while the
// current one is parsed.
__device__ __forceinline__ uint4 load16_guarded(const uint8_t *buf, uint32_t a, uint32_t buf_len)
{
    if (a + 16 <= buf_len) return *reinterpret_cast<const uint4 *>(buf + a);
    uint32_t w[4] = {0, 0, 0, 0};   // last bytes of the job buffer: never read past its end
    for (uint32_t x = 0; x < 16; ++x)
        if (a + x < buf_len) w[x >> 2] |= (uint32_t)buf[a + x] << (8 * (x & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <class F>
__device__ __forceinline__ void for_each_title_byte(const uint8_t *buf, uint32_t buf_len, uint32_t ls, uint32_t te, bool valid,
                                                    F &&body)
{
    const uint32_t a0 = ls & ~15u;
    const uint32_t nch = valid ? ((te - a0) >> 4) + 1 : 0;
    const uint32_t maxch = t_wmax(nch);
    uint4 cur = make_uint4(0, 0, 0, 0), nxt = cur;
    if (nch) cur = load16_guarded(buf, a0, buf_len);
    for (uint32_t ch = 0; ch < maxch; ++ch) {
        if (ch + 1 < nch) nxt = load16_guarded(buf, a0 + 16 * (ch + 1), buf_len);
        const uint32_t w[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (uint32_t b = 0; b < 16; ++b) {
            const uint32_t p = a0 + 16 * ch + b;
            const uint32_t act = (ch < nch && p >= ls && p <= te) ? 1u : 0u;
            const uint32_t c = (w[b >> 2] >> (8 * (b & 3))) & 0xffu;
            body(c, act, p);
        }
        cur = nxt;
    }
}

// =========================================================================================================
// Word-parallel title path for jobs whose titles are all <= 128 bytes (W = 1 or 2 mask words of 64 title bytes).
//
// The byte-serial parsers above cost ~100 VALU instructions per title byte and wave (PMC: 6.4 k VALU per
// wave, VALU-bound).  Here a lane holds its whole title in 16 dwords, classifies four bytes per step with
// v_perm_b32 nibble tables into a 64-bit separator mask and a 64-bit digit mask, and then walks the FIELDS
// (wave-uniform loop over record 0's field count) instead of the bytes: field length = count of trailing
// zeros of the shifted separator mask, field bytes = funnel-shifted dwords of the lane's own LDS column,
// equality / Hamming mask = XOR against record 0's field words, numeric value = SWAR 8-digit parse.
// Same statistics, same s_stat layout and same results as k_title_stats_reg.
template <uint32_t W> constexpr uint32_t TS_ROWS = 16 * W + 1;   // title dwords + one zero row for reads past the title

// Four bytes at once: per byte, bits 0-3 = separator class (non-zero <=> one of " ._,=:/-#\n"), bit 4 = digit.
// class(byte) = LUT_lo[low nibble] & LUT_hi[high nibble]; each LUT is a v_perm_b32 byte table (8 entries per
// instruction, the 16-entry low table is two of them selected by nibble bit 3); bytes >= 0x80 are cleared.
__device__ __forceinline__ uint32_t t_classify4(uint32_t w)
{
    const uint32_t lo3 = w & 0x07070707u;
    const uint32_t pa = __builtin_amdgcn_perm(0x10101010u, 0x12101012u, lo3);   // low nibble 0..7
    const uint32_t pb = __builtin_amdgcn_perm(0x0A020602u, 0x00051010u, lo3);   // low nibble 8..15
    const uint32_t m3 = ((w >> 3) & 0x01010101u) * 0xFFu;
    const uint32_t lo = (pa & ~m3) | (pb & m3);
    const uint32_t hi = __builtin_amdgcn_perm(0x00000800u, 0x14020001u, (w >> 4) & 0x07070707u);
    const uint32_t m7 = ((w >> 7) & 0x01010101u) * 0xFFu;
    return lo & hi & ~m7;
}
// value of four decimal digits (nibbles 0..9 in the bytes, most significant digit in byte 0)
__device__ __forceinline__ uint32_t t_val4(uint32_t x)
{
    const uint32_t p = (x * 10u + (x >> 8)) & 0x00FF00FFu;
    return (p & 0xFFu) * 100u + (p >> 16);
}
// SWAR parse of an all-digit field of len <= 8 whose bytes sit in the low bytes of {w1,w0}
__device__ __forceinline__ uint32_t t_parse8(uint32_t w0, uint32_t w1, uint32_t len)
{
    const uint64_t d = ((uint64_t)(w1 & 0x0F0F0F0Fu) << 32) | (w0 & 0x0F0F0F0Fu);
    const uint64_t t = d << ((8u * (8u - len)) & 63u);    // right-align: leading bytes become zero digits
    return t_val4((uint32_t)t) * 10000u + t_val4((uint32_t)(t >> 32));
}

// ---------------------------------------------------------------------------------------------------------
// One wave per sub-block: the table is built in LDS (lanes fetch the title and clear the table together, lane 0
// walks the ~40 bytes) and copied out with coalesced stores.
__global__ __launch_bounds__(64) void k_rec0_fields(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.x;
    const int lane = threadIdx.x;
    __shared__ Rec0Tab T;
    const SbState &sb = A.sb[s];
    const uint32_t ls0 = A.ls[sb.rec_lo], te0 = A.te[sb.rec_lo];
    const uint32_t tlen = te0 - ls0 + 1;
    if (tlen > REC0_TITLE_CAP) { if (lane == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
    for (uint32_t k = lane; k < sizeof(Rec0Tab) / 4; k += 64) reinterpret_cast<uint32_t *>(&T)[k] = 0;
    __syncthreads();
    for (uint32_t p = lane; p < tlen; p += 64) T.title[p] = A.buf[ls0 + p];
    __syncthreads();
    if (lane == 0) {
        uint32_t nf0 = 0, ncand = 0, start = 0;
        for (uint32_t p = 0; p < tlen; ++p) {
            if (!is_sep(T.title[p])) continue;
            if (nf0 < MAX_FIELDS) {
                uint32_t v;
                bool o;
                T.start[nf0] = (uint16_t)start;
                T.len[nf0] = (uint16_t)(p - start);
                uint8_t km = 0xff;
                if (parse_num(T.title, start, p, v, o)) {   // only a numeric record 0 can leave the field numeric
                    T.val0[nf0] = v;
                    T.ovf0[nf0] = o ? 1 : 0;
                    if (ncand < NUM_CAND_CAP) { km = (uint8_t)ncand; T.cand_f[ncand++] = (uint8_t)nf0; }
                    else { km = 0xfe; A.fstat[(size_t)s * A.fcap + nf0].no_delta = 1; }
                }
                T.kmap[nf0] = km;
            }
            ++nf0;
            start = p + 1;
        }
        // field-aligned copy of the title for the word-parallel path: every field starts on a dword and is zero
        // padded to a multiple of eight bytes
        uint32_t wcur = 0;
        uint8_t *al = reinterpret_cast<uint8_t *>(T.al);
        for (uint32_t f = 0; f < nf0 && f < MAX_FIELDS; ++f) {
            T.woff[f] = (uint16_t)wcur;
            for (uint32_t k = 0; k < T.len[f]; ++k) al[4 * wcur + k] = T.title[T.start[f] + k];
            wcur += 2 * ((T.len[f] + 7u) / 8u);
        }
        // record 0's own separator / digit masks, one bit per title byte (the first 128): what a dword equal to record 0's
        // contributes to another record's masks (t_load_title)
        for (uint32_t p = 0; p < tlen && p < 128; ++p) {
            if (is_sep(T.title[p])) T.sepm[p >> 5] |= 1u << (p & 31);
            if (T.title[p] >= '0' && T.title[p] <= '9') T.digm[p >> 5] |= 1u << (p & 31);
        }
        for (uint32_t f = 0; f < nf0 && f < MAX_FIELDS; ++f) {
            T.fdw[f][0] = (uint32_t)T.start[f] | ((uint32_t)T.len[f] << 16);
            T.fdw[f][1] = (uint32_t)T.kmap[f] | ((uint32_t)T.ovf0[f] << 8) | ((uint32_t)T.woff[f] << 16);
        }
        T.nf0 = nf0;
        T.ncand = ncand;
        A.sb[s].nf = nf0;
        if (nf0 > MAX_FIELDS) {   // more separators than the field masks of these kernels hold: the job runs again on the wide title kernels
            atomicOr(A.wide_needed, 1u);
            atomicMin(A.err, (int32_t)ERR_UNSUPPORTED);
        }
    }
    __syncthreads();
    uint32_t *dst = reinterpret_cast<uint32_t *>(A.rec0 + s);
    for (uint32_t k = lane; k < sizeof(Rec0Tab) / 4; k += 64) dst[k] = reinterpret_cast<const uint32_t *>(&T)[k];
}

// ---------------------------------------------------------------------------------------------------------
// 255 owned records + the previous record as delta halo (thread 0) per workgroup.
__global__ __launch_bounds__(TT_THREADS) void k_title_stats_reg(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    constexpr uint32_t OWNED = TT_THREADS - 1;
    if (blockIdx.x * OWNED >= n) return;
    const int tid = threadIdx.x;

    __shared__ uint32_t s_stat[MAX_FIELDS * FSTAT_WORDS];
    __shared__ uint32_t s_val[NUM_CAND_CAP][TT_THREADS];
    __shared__ fmask_t s_nummask[TT_THREADS];
    __shared__ Rec0Tab s_r0;

    for (uint32_t k = tid; k < sizeof(Rec0Tab) / 4; k += TT_THREADS)
        reinterpret_cast<uint32_t *>(&s_r0)[k] = reinterpret_cast<const uint32_t *>(A.rec0 + s)[k];
    for (uint32_t k = tid; k < MAX_FIELDS * FSTAT_WORDS; k += TT_THREADS) s_stat[k] = 0;
    __syncthreads();
    const uint32_t nf0 = s_r0.nf0;

    const int64_t i = (int64_t)blockIdx.x * OWNED + tid - 1;  // record index inside the sub-block
    const bool valid = i >= 0 && i < (int64_t)n;
    const bool owned = valid && tid >= 1;
    const uint32_t g = valid ? rec_lo + (uint32_t)i : rec_lo;
    const uint32_t ls = A.ls[g], te = A.te[g];

    // parser state in integer registers; st bit0: non-digit seen, bit1: leading zero,
    // bit2: the value exceeds 32 bits, bit3: differs from record 0
    uint32_t f = 0, k = 0, v = 0, stf = 0;
    fmask_t nummask = 0;
    uint32_t f0len = nf0 ? s_r0.len[0] : 0, f0start = nf0 ? s_r0.start[0] : 0;
    for_each_title_byte(A.buf, A.buf_len, ls, te, valid, [&](uint32_t c, uint32_t act, uint32_t) {
        const uint32_t sepb = act & t_sep(c);
        const uint32_t body = act & (sepb ^ 1u);
        const uint32_t d = c - '0';
        const uint32_t dig = d <= 9 ? 1u : 0u;
        const uint32_t ld0 = (k == 0 && d == 0) ? 1u : 0u;
        const uint32_t inr = body & (k < f0len ? 1u : 0u);          // f0len == 0 once f >= nf0
        const uint32_t c0 = s_r0.title[f0start + (inr ? k : 0)];
        const uint32_t nd = inr & (c != c0 ? 1u : 0u);
        stf |= (body & (dig ^ 1u)) | ((body & ld0) << 1) | (nd << 3);
        if (body & dig) {                                              // wraps like utils::to_num (utils.h:117-125); bit 2: it has wrapped
            const uint64_t t10 = (uint64_t)v * 10 + d;
            stf |= (uint32_t)(t10 >> 32) ? 4u : 0u;
            v = (uint32_t)t10;
        }
        if (__any(nd)) {
            if (nd && owned && k < MAX_TEXT_LEN0) t_lds_or(&s_stat[f * FSTAT_WORDS + 10 + (k >> 5)], 1u << (k & 31));
        }
        k += body;
        if (__any(sepb)) {
            if (sepb) {
                if (f < nf0) {
                    const uint32_t len = k, len0 = f0len;
                    const bool isn = !(stf & 1) && len > 0 && !(len > 1 && (stf & 2));  // utils::is_num, utils.h:107-114
                    // A field identical to record 0's cannot change any statistic beyond what record 0's own
                    // thread contributes, so only differing fields (and record 0 itself) touch the counters.
                    const bool same = len == len0 && !(stf & 8);
                    if (owned && (!same || i == 0)) {
                        uint32_t *st = s_stat + f * FSTAT_WORDS;
                        const uint32_t a0 = st[0], a1 = st[1], a2 = st[2], a3 = st[3], a4 = st[4], a5 = st[5], a6 = st[6],
                                       a7 = st[7];
                        const uint32_t bv = bias((int32_t)v);
                        const bool n0 = len > a0, n1 = ~len > a1, n2 = !same && !a2, n3 = len != len0 && !a3,
                                   n4 = !isn && !a4, n5 = isn && ((v >> 31) || (stf & 4)) && !a5, n6 = isn && bv > a6, n7 = isn && ~bv > a7;
                        if (n0 | n1 | n2 | n3 | n4 | n5 | n6 | n7) {
                            if (n0) atomicMax(st + 0, len);
                            if (n1) atomicMax(st + 1, ~len);
                            if (n2) atomicMax(st + 2, 1u);
                            if (n3) atomicMax(st + 3, 1u);
                            if (n4) atomicMax(st + 4, 1u);
                            if (n5) atomicMax(st + 5, 1u);
                            if (n6) atomicMax(st + 6, bv);
                            if (n7) atomicMax(st + 7, ~bv);
                        }
                    }
                    const uint32_t kk = s_r0.kmap[f];
                    if (kk < NUM_CAND_CAP) s_val[kk][tid] = v;
                    nummask |= (fmask_t)isn << f;
                }
                ++f; k = 0; v = 0; stf = 0;
                f0len = f < nf0 ? s_r0.len[f] : 0;
                f0start = f < nf0 ? s_r0.start[f] : 0;
            }
        }
    });
    if (valid && f != nf0) atomicMin(A.err, (int32_t)ERR_FIELDS);  // phyNGSC.cpp:417-421 only warns
    s_nummask[tid] = nummask;
    __syncthreads();
    if (owned && i >= 1) {
        const fmask_t both = nummask & s_nummask[tid - 1];
        const uint32_t ncand = s_r0.ncand;
        for (uint32_t kk = 0; kk < ncand; ++kk) {
            const uint32_t ff = s_r0.cand_f[kk];
            if ((both >> ff) & 1) {
                const int32_t d = (int32_t)(s_val[kk][tid] - s_val[kk][tid - 1]);  // tasks.cpp:149-166
                const uint32_t bd = bias(d);
                if (bd > s_stat[ff * FSTAT_WORDS + 8]) atomicMax(&s_stat[ff * FSTAT_WORDS + 8], bd);
                if (~bd > s_stat[ff * FSTAT_WORDS + 9]) atomicMax(&s_stat[ff * FSTAT_WORDS + 9], ~bd);
            }
        }
    }
    __syncthreads();
    uint32_t *gst = reinterpret_cast<uint32_t *>(A.fstat + (size_t)s * A.fcap);
    for (uint32_t kx = tid; kx < nf0 * FSTAT_WORDS; kx += TT_THREADS) {
        uint32_t x = s_stat[kx];
        if (!x) continue;
        uint32_t w = kx % FSTAT_WORDS;
        if (w >= 10 && w < 26) atomicOr(gst + kx, x);
        else if (w < 10) atomicMax(gst + kx, x);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Cached non-constant title fields of one record: numeric -> value, text -> (offset from the record start) |
// length << 16.
struct CachedTitle {
    uint32_t cur[NCF_CAP];
    uint32_t prev[NCF_CAP];
};

// One record's title bits from its cached fields (tasks.cpp:427-507; same order and rules as
// title_record_emit() in ngsc_common.h, which the fallback path and the CPU harness use).
// rec: first byte of the record (text field offsets are relative to it).
template <class Sink>
__device__ __forceinline__ void title_emit_cached(const uint8_t *rec, const Arena &A, uint32_t s, const FieldDesc *fd,
                                                  const uint8_t *kf_field, uint32_t ncf, bool first,
                                                  const CachedTitle &C, fmask_t gconst, Sink &w)
{
#pragma unroll
    for (uint32_t kf = 0; kf < NCF_CAP; ++kf) {
        if (kf >= ncf) break;
        const uint32_t f = kf_field[kf];
        const FieldDesc &d = fd[f];
        const uint32_t wc = C.cur[kf];
        if (d.kind == KIND_NUM) {
            if (first) {
                w.put(wc - (uint32_t)d.min_v, d.bits_val);                       // tasks.cpp:457
            } else if (!((gconst >> f) & 1)) {
                const uint32_t to_store = d.delta_coding ? (wc - C.prev[kf] - (uint32_t)d.min_d) : (wc - (uint32_t)d.min_v);
                if (d.has_huff) {
                    const uint32_t e = A.ncodes[((size_t)s * A.fcap + f) * MAX_NUM_HUF + to_store];
                    w.put(e >> 5, e & 31);
                } else {
                    w.put(to_store, d.bits_num);
                }
            }
        } else {  // KIND_TEXT
            if (first || !((gconst >> f) & 1)) {                                  // tasks.cpp:487
                const uint32_t start = wc & 0xffffu, len = wc >> 16;
                if (!d.is_len_const) w.put(len - d.min_len, d.bits_len);
                const uint32_t *tc = A.tcodes + (size_t)d.slot * TEXT_TREES * 256;
                // (the field's first eight bytes with one load: short text fields - a filter flag, a barcode - cost a memory
                // round trip per CHARACTER otherwise, in measure and in pack)
                const uint2 f8 = len ? load8_unaligned(A.buf, (uint32_t)(rec - A.buf) + start, A.buf_len) : make_uint2(0, 0);
                for (uint32_t j = 0; j < len; ++j) {
                    if (!ham_bit(d, j)) {
                        const uint32_t c = j < 4 ? (f8.x >> (8 * j)) & 0xffu : (j < 8 ? (f8.y >> (8 * (j - 4))) & 0xffu : rec[start + j]);
                        const uint32_t e = tc[(j < MAX_STAT_LEN ? j : MAX_STAT_LEN) * 256 + c];
                        w.put(e >> 5, e & 31);
                    }
                }
            }
        }
    }
}

// The group's flag bits and their constancy, walked over the sub-block's NON-CONSTANT fields only (kf_field, in field order -
// the order tasks.cpp:401-425 writes them in): same results as title_group_const / title_put_flags, which visit every field.
__device__ __forceinline__ fmask_t t_group_const_kf(const FieldDesc *fd, const uint8_t *kf_field, uint32_t ncf, TitleDiffs g, uint32_t group_size)
{
    fmask_t gconst = 0;
#pragma unroll
    for (uint32_t kf = 0; kf < NCF_CAP; ++kf) {
        if (kf >= ncf) break;
        const uint32_t f = kf_field[kf];
        const FieldDesc &d = fd[f];
        uint32_t flag;
        if (d.kind == KIND_TEXT) flag = !((g.txt >> f) & 1);
        else if (d.delta_coding) flag = group_size == 1 ? (d.min_d == 0) : !((g.dlt >> f) & 1);   // (tasks.cpp:415)
        else flag = !((g.val >> f) & 1);
        gconst |= (fmask_t)flag << f;
    }
    return gconst;
}
template <class Sink>
__device__ __forceinline__ void t_put_flags_kf(const uint8_t *kf_field, uint32_t ncf, fmask_t gconst, Sink &w)
{
#pragma unroll
    for (uint32_t kf = 0; kf < NCF_CAP; ++kf) {
        if (kf >= ncf) break;
        w.put((uint32_t)((gconst >> kf_field[kf]) & 1), 1);
    }
}

// 256 records (8 groups of 32) per workgroup.  SWAR: titles <= 64 * W bytes whose separator masks k_title_stats_swar
// left in A.tsep - the fields are found with count-trailing-zeros steps and only the non-constant numeric ones
// are read (eight bytes, unaligned) and parsed; otherwise the title is parsed byte by byte.
template <bool SWAR, uint32_t W>
__global__ __launch_bounds__(TT_THREADS) void k_title_measure_reg(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t slot_bias = sb.slot_bias;   // (read once: the stores below could alias A.sb)
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo, nf = sb.nf, ncf = sb.ncf;
    const uint32_t t0 = blockIdx.x * TT_THREADS;
    if (t0 >= n) return;
    const uint32_t nt = t0 + TT_THREADS < n ? TT_THREADS : n - t0;
    const uint32_t i0 = rec_lo + t0;
    const int tid = threadIdx.x;

    __shared__ FieldDesc s_fd[MAX_FIELDS];
    __shared__ uint32_t s_w[NCF_CAP][TT_THREADS];
    __shared__ uint8_t s_kf_field[NCF_CAP];
    __shared__ uint32_t s_ntext;   // some non-constant field is a text field

    for (uint32_t k = tid; k < nf * (sizeof(FieldDesc) / 4); k += TT_THREADS)
        reinterpret_cast<uint32_t *>(s_fd)[k] = reinterpret_cast<const uint32_t *>(A.fdesc + (size_t)s * A.fcap)[k];
    if (tid == 0) s_ntext = 0;
    __syncthreads();
    if ((uint32_t)tid < nf && s_fd[tid].kind != KIND_CONST && s_fd[tid].kf < NCF_CAP) {
        s_kf_field[s_fd[tid].kf] = (uint8_t)tid;
        if (s_fd[tid].kind == KIND_TEXT) s_ntext = 1;
    }
    __syncthreads();

    const bool valid = (uint32_t)tid < nt;
    const uint32_t gi = tid & 31;                  // index inside the 32-record group
    const uint32_t gstart = tid - gi;              // first record of the group (relative to t0)
    const uint32_t gsize = valid ? ((gstart + 32 <= nt) ? 32u : nt - gstart) : 0u;
    const uint32_t g = valid ? i0 + tid : i0;
    const uint32_t ls = A.ls[g], te = A.te[g];
    uint32_t bits = 0;
    fmask_t gconst = 0;
    if (SWAR && ncf <= NCF_CAP && !s_ntext) {
        // (round 3) Every non-constant field is numeric - the usual title: a read counter, coordinates - and its value was parsed by
        // k_title_stats_swar: no walk over the title's fields, no separator mask, and the neighbours' values (previous record,
        // first record of the 32-record group: both in this wave) come by lane shuffle instead of through LDS and a barrier.
        // Flags and bits are computed over the list of non-constant fields, not over all fields.
        CachedTitle C;
        TitleDiffs d = {0, 0, 0};
        const int lane = tid & 63;
#pragma unroll
        for (uint32_t kf = 0; kf < NCF_CAP; ++kf) {
            C.cur[kf] = C.prev[kf] = 0;
            if (kf >= ncf) continue;
            const uint32_t ff = s_kf_field[kf];
            const uint32_t wc = valid ? A.tv[(size_t)s_fd[ff].cand * A.tw_stride + g + slot_bias] : 0u;
            const uint32_t wp = (uint32_t)__shfl_up((int)wc, 1), wf = (uint32_t)__shfl((int)wc, lane & 32);
            if (!valid) continue;
            C.cur[kf] = wc;
            A.tw[(size_t)kf * A.tw_stride + g + slot_bias] = wc;
            if (gi) {
                C.prev[kf] = wp;
                if (wc != wf) d.val |= (fmask_t)1 << ff;
                if ((int32_t)(wc - wp) != s_fd[ff].min_d) d.dlt |= (fmask_t)1 << ff;
            }
        }
        TitleDiffs gd = {0, t_g32_or(d.val), t_g32_or(d.dlt)};
        gconst = t_group_const_kf(s_fd, s_kf_field, ncf, gd, gsize);
        if (valid) {
            BitCounter bc;
            if (gi == 0) t_put_flags_kf(s_kf_field, ncf, gconst, bc);
            title_emit_cached(A.buf + ls, A, s, s_fd, s_kf_field, ncf, gi == 0, C, gconst, bc);
            bits = (uint32_t)bc.bits;
        }
    } else if (ncf <= NCF_CAP) {
        // one pass over the title: cache the non-constant fields of this record
        if constexpr (SWAR) {
            uint64_t sepmask[W];
#pragma unroll
            for (uint32_t k = 0; k < W; ++k) sepmask[k] = valid ? A.tsep[(size_t)g * W + k] : 0;
            uint32_t pos = 0;
            for (uint32_t f = 0; f < nf; ++f) {
                const uint32_t len = t_next_set<W>(sepmask, pos);
                const uint32_t kind = s_fd[f].kind;
                if (kind == KIND_NUM) {   // parsed once, by k_title_stats_swar
                    s_w[s_fd[f].kf][tid] = valid ? A.tv[(size_t)s_fd[f].cand * A.tw_stride + g + slot_bias] : 0u;
                } else if (kind == KIND_TEXT) {
                    s_w[s_fd[f].kf][tid] = pos | (len << 16);
                }
                pos += len + 1;
            }
        } else {
            uint32_t f = 0, k = 0, v = 0;
            for_each_title_byte(A.buf, A.buf_len, ls, te, valid, [&](uint32_t c, uint32_t act, uint32_t p) {
                const uint32_t sepb = act & t_sep(c);
                const uint32_t body = act & (sepb ^ 1u);
                v = body ? v * 10 + (c - '0') : v;   // wraps like utils::to_num; only used for all-numeric fields
                k += body;
                if (__any(sepb)) {
                    if (sepb) {
                        if (f < nf && s_fd[f].kind != KIND_CONST)
                            s_w[s_fd[f].kf][tid] = s_fd[f].kind == KIND_NUM ? v : ((p - k - ls) | (k << 16));
                        ++f; k = 0; v = 0;
                    }
                }
            });
        }
        __syncthreads();
        CachedTitle C;
        TitleDiffs d = {0, 0, 0};
#pragma unroll
        for (uint32_t kf = 0; kf < NCF_CAP; ++kf) {
            C.cur[kf] = C.prev[kf] = 0;
            if (kf >= ncf || !valid) continue;
            const uint32_t ff = s_kf_field[kf];
            const uint32_t wc = s_w[kf][tid];
            C.cur[kf] = wc;
            A.tw[(size_t)kf * A.tw_stride + g + slot_bias] = wc;
            if (gi) {
                const uint32_t wp = s_w[kf][tid - 1], wf = s_w[kf][gstart];
                C.prev[kf] = wp;
                if (s_fd[ff].kind == KIND_NUM) {
                    if (wc != wf) d.val |= (fmask_t)1 << ff;
                    if ((int32_t)(wc - wp) != s_fd[ff].min_d) d.dlt |= (fmask_t)1 << ff;
                } else {
                    const uint32_t lc = wc >> 16;
                    const uint8_t *mine = A.buf + ls + (wc & 0xffffu), *frst = A.buf + A.ls[i0 + gstart] + (wf & 0xffffu);
                    bool same = lc == (wf >> 16);
                    if (same && lc) {   // eight bytes at a time (short fields: one step)
                        for (uint32_t j = 0; same && j < lc; j += 8) {
                            const uint2 m8 = load8_unaligned(A.buf, (uint32_t)(mine - A.buf) + j, A.buf_len);
                            const uint2 f8 = load8_unaligned(A.buf, (uint32_t)(frst - A.buf) + j, A.buf_len);
                            const uint64_t km = t_lowbytes(lc - j < 8 ? lc - j : 8);
                            same = (((m8.x ^ f8.x) & (uint32_t)km) | ((m8.y ^ f8.y) & (uint32_t)(km >> 32))) == 0;
                        }
                    }
                    if (!same) d.txt |= (fmask_t)1 << ff;
                }
            }
        }
        TitleDiffs gd = {t_g32_or(d.txt), t_g32_or(d.val), t_g32_or(d.dlt)};
        gconst = t_group_const_kf(s_fd, s_kf_field, ncf, gd, gsize);
        if (valid) {
            BitCounter bc;
            if (gi == 0) t_put_flags_kf(s_kf_field, ncf, gconst, bc);
            title_emit_cached(A.buf + ls, A, s, s_fd, s_kf_field, ncf, gi == 0, C, gconst, bc);
            bits = (uint32_t)bc.bits;
        }
    } else {
        // more non-constant fields than the cache holds: walk (first, previous, current) titles per record
        TitleDiffs d = {0, 0, 0};
        const uint32_t p_prev = (valid && gi) ? A.ls[g - 1] : ls;
        if (valid && gi) d = title_record_diffs(A.buf, s_fd, nf, A.ls[i0 + gstart], p_prev, ls);
        TitleDiffs gd = {t_g32_or(d.txt), t_g32_or(d.val), t_g32_or(d.dlt)};
        gconst = title_group_const(s_fd, nf, gd, gsize);
        if (valid) {
            BitCounter bc;
            if (gi == 0) title_put_flags(s_fd, nf, gconst, bc);
            title_record_emit(A.buf, A, s, s_fd, nf, gi == 0, p_prev, ls, gconst, bc);
            bits = (uint32_t)bc.bits;
        }
    }
    if (valid) A.tbits[g + slot_bias] = bits;
    const uint32_t gbits = t_g32_sum(bits);
    if (valid && gi == 0) {
        const uint32_t grp = A.grp_lo[s] + (t0 + gstart) / GROUP;
        A.goff[grp] = (gbits + 7) >> 3;
        A.gflags[grp] = gconst;
    }
}

// ---------------------------------------------------------------------------------------------------------
// One lane's MSB-first bit writer into a word array (LDS staging, or the zeroed output in the overflow path).
struct TLaneWriter {
    uint32_t *words;
    uint64_t widx;
    uint64_t acc;
    uint32_t nacc;
    __device__ __forceinline__ void init(uint32_t *w, uint64_t word0, uint64_t bitpos)
    {
        words = w;
        widx = (bitpos >> 5) - word0;
        acc = 0;
        nacc = (uint32_t)(bitpos & 31);
    }
    __device__ __forceinline__ void put(uint32_t v, uint32_t nb)
    {
        if (!nb) return;
        if (nb < 32) v &= (1u << nb) - 1u;
        acc = (acc << nb) | v;
        nacc += nb;
        if (nacc >= 32) {
            uint32_t word = (uint32_t)(acc >> (nacc - 32));
            if (word) atomicOr(&words[widx], __builtin_bswap32(word));
            ++widx;
            nacc -= 32;
            acc &= (1ull << nacc) - 1ull;
        }
    }
    __device__ __forceinline__ void finish()
    {
        if (nacc) {
            uint32_t word = (uint32_t)(acc << (32 - nacc));
            if (word) atomicOr(&words[widx], __builtin_bswap32(word));
            nacc = 0;
        }
    }
};

constexpr uint32_t TP_STAGE_WORDS = 2048;  // 8 KB: 32 B of title bits per record on average

__global__ __launch_bounds__(TT_THREADS) void k_title_pack_reg(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t slot_bias = sb.slot_bias;   // (read once: the stores below could alias A.sb)
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo, nf = sb.nf, ncf = sb.ncf;
    const uint32_t t0 = blockIdx.x * TT_THREADS;
    if (t0 >= n) return;
    const uint32_t nt = t0 + TT_THREADS < n ? TT_THREADS : n - t0;
    const uint32_t i0 = rec_lo + t0;
    const bool last_tile = t0 + TT_THREADS >= n;
    const int tid = threadIdx.x;

    __shared__ FieldDesc s_fd[MAX_FIELDS];
    __shared__ uint8_t s_kf_field[NCF_CAP];
    __shared__ uint32_t s_stage[TP_STAGE_WORDS];

    for (uint32_t k = tid; k < TP_STAGE_WORDS; k += TT_THREADS) s_stage[k] = 0;
    for (uint32_t k = tid; k < nf * (sizeof(FieldDesc) / 4); k += TT_THREADS)
        reinterpret_cast<uint32_t *>(s_fd)[k] = reinterpret_cast<const uint32_t *>(A.fdesc + (size_t)s * A.fcap)[k];
    __syncthreads();
    if ((uint32_t)tid < nf && s_fd[tid].kind != KIND_CONST && s_fd[tid].kf < NCF_CAP) s_kf_field[s_fd[tid].kf] = (uint8_t)tid;
    __syncthreads();

    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out);
    const uint32_t grp_first = A.grp_lo[s] + t0 / GROUP;
    const uint32_t ngrp_t = (nt + GROUP - 1) / GROUP;
    const uint64_t title_base = (sb.out_off + sb.info_len + sb.title_hdr) * 8;
    const uint64_t tb0 = title_base + 8ull * A.goff[grp_first];
    const uint64_t tb1 = title_base + 8ull * (last_tile ? sb.title_body : A.goff[grp_first + ngrp_t]);
    const uint64_t tw0 = tb0 >> 5;
    const uint32_t tnw = (uint32_t)(((tb1 + 31) >> 5) - tw0);
    const bool t_lds = tnw + 1 <= TP_STAGE_WORDS;  // otherwise: straight into the zeroed output with atomics

    const bool valid = (uint32_t)tid < nt;
    const uint32_t gi = tid & 31, gstart = tid - gi;
    const uint32_t g = valid ? i0 + tid : i0;
    const uint32_t bits = valid ? A.tbits[g + slot_bias] : 0;
    uint32_t incl = bits;  // inclusive prefix inside the 32-record group
#pragma unroll
    for (int d = 1; d < 32; d <<= 1) {
        uint32_t o = __shfl_up(incl, d);
        if ((int)gi >= d) incl += o;
    }
    if (valid) {
        const uint32_t grp = A.grp_lo[s] + (t0 + gstart) / GROUP;
        const fmask_t gconst = A.gflags[grp];
        TLaneWriter w;
        w.init(t_lds ? s_stage : outw, t_lds ? tw0 : 0, title_base + 8ull * A.goff[grp] + (incl - bits));
        const uint32_t ls = A.ls[g];
        if (ncf <= NCF_CAP) {
            if (gi == 0) t_put_flags_kf(s_kf_field, ncf, gconst, w);
            CachedTitle C;
#pragma unroll
            for (uint32_t kf = 0; kf < NCF_CAP; ++kf) {
                C.cur[kf] = C.prev[kf] = 0;
                if (kf >= ncf) continue;
                C.cur[kf] = A.tw[(size_t)kf * A.tw_stride + g + slot_bias];
                if (gi) C.prev[kf] = A.tw[(size_t)kf * A.tw_stride + g + slot_bias - 1];
            }
            title_emit_cached(A.buf + ls, A, s, s_fd, s_kf_field, ncf, gi == 0, C, gconst, w);
        } else {
            if (gi == 0) title_put_flags(s_fd, nf, gconst, w);
            title_record_emit(A.buf, A, s, s_fd, nf, gi == 0, gi ? A.ls[g - 1] : ls, ls, gconst, w);
        }
        w.finish();
    }
    __syncthreads();
    if (t_lds) {
        for (uint32_t k = tid; k < tnw; k += TT_THREADS) {
            const uint64_t w = tw0 + k;
            const uint32_t x = s_stage[k];
            if ((w << 5) >= tb0 && ((w + 1) << 5) <= tb1) outw[w] = x;   // complete word of this tile's range
            else if (x) atomicOr(&outw[w], x);                           // shared with the neighbours
        }
    }
}

// Title of one record as 16 * W dwords (bytes past the title are other lines' bytes: every use masks by length),
// written to the lane's LDS column, plus its separator / digit masks.
// r0w: record 0's title as dwords (zero padded); diffmask: bit p set where byte p differs from record 0's byte p
// sep0 / dig0: record 0's own separator / digit masks as 32-bit halves (Rec0Tab::sepm / digm).  A dword that equals record
// 0's dword in every lane of the wave - the constant stretches of a title, seven of cfg2's thirteen dwords - is not
// classified at all: its mask bits are record 0's (the vote costs two instructions, the classification and the
// difference bits forty), and dwords behind the wave's longest title are skipped.
// Wave-uniform read-only tables (record 0's field table) are read through the CONSTANT address space: the compiler then
// fetches them with scalar loads into SGPRs - no LDS copy of the table per workgroup, no vector load + readfirstlane per value.
typedef const __attribute__((address_space(4))) uint32_t *cu32p;
typedef const __attribute__((address_space(4))) Rec0Tab *crec0p;

template <uint32_t W>
__device__ __forceinline__ void t_load_title(const Arena &A, uint32_t ls, uint32_t tlen, uint32_t (*s_w)[TT_THREADS], int tid,
                                             uint64_t (&sepmask)[W], uint64_t (&digmask)[W], cu32p r0w, uint64_t (&diffmask)[W],
                                             cu32p sep0, cu32p dig0, uint32_t z, bool t_quad)   // z: rows of s_w that hold title dwords; row z is the zero row
{
    uint32_t sp32[2 * W], dg32[2 * W], df32[2 * W];
#pragma unroll
    for (uint32_t k = 0; k < 2 * W; ++k) sp32[k] = dg32[k] = df32[k] = 0;
    // (round 3) FOUR lanes fetch a record's title, 16 bytes each of 64 consecutive ones: a load instruction of the wave then reaches
    // for 16 stretches of 64 bytes instead of 64 stretches of 16 (a title a lane, a whole record apart) - the texture addresser, at
    // 77 % the busiest unit of this kernel once it ran at full occupancy, prices an instruction by the cache lines it touches.  The
    // dwords go to the record's column of s_w, where the field code below wants them anyway, and every lane reads its own title
    // back from there (one conflict-free LDS read per dword).  NGSC_TS_QUAD=0 (t_quad false): a lane loads its own title.
    const uint32_t lane = (uint32_t)tid & 63u, wave0 = (uint32_t)tid & ~63u;
    if (t_quad) {
        const uint32_t sub = lane & 3u;
#pragma unroll
        for (uint32_t j = 0; j < 4; ++j) {
            const uint32_t rl = 16 * j + (lane >> 2);                       // the record (lane of this wave) this lane fetches for
            const uint32_t ls_r = (uint32_t)__shfl((int)ls, (int)rl), tl_r = (uint32_t)__shfl((int)tlen, (int)rl);
#pragma unroll
            for (uint32_t h = 0; h < W; ++h) {
                const uint32_t c = 4 * h + sub;                             // chunk of the title
                if (!__any(16 * c < tl_r)) continue;
                uint4 q = make_uint4(0, 0, 0, 0);
                if (16 * c < tl_r) q = load16_unaligned(A.buf, ls_r + 16 * c, A.buf_len);
                uint32_t *col = &s_w[0][wave0 + rl];
                if (4 * c + 0 < z) col[(4 * c + 0) * TT_THREADS] = q.x;
                if (4 * c + 1 < z) col[(4 * c + 1) * TT_THREADS] = q.y;
                if (4 * c + 2 < z) col[(4 * c + 2) * TT_THREADS] = q.z;
                if (4 * c + 3 < z) col[(4 * c + 3) * TT_THREADS] = q.w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (uint32_t m = 0; m < 4 * W; ++m) {
        uint4 q = make_uint4(0, 0, 0, 0);
        if (__any(16 * m < tlen)) {
            if (t_quad) {   // (rows behind the title of THIS lane hold zeros or what its quad did not fetch: never used beyond tlen)
                if (16 * m < tlen) {
                    q.x = 4 * m + 0 < z ? s_w[4 * m + 0][tid] : 0u;
                    q.y = 4 * m + 1 < z ? s_w[4 * m + 1][tid] : 0u;
                    q.z = 4 * m + 2 < z ? s_w[4 * m + 2][tid] : 0u;
                    q.w = 4 * m + 3 < z ? s_w[4 * m + 3][tid] : 0u;
                }
            } else if (16 * m < tlen) q = load16_unaligned(A.buf, ls + 16 * m, A.buf_len);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t d = 4 * m + j, sh = 4 * (d & 7);
                if (!__any(4 * d < tlen)) continue;
                const uint32_t x = w[j] ^ r0w[d];
                if (__any(x != 0 && 4 * d < tlen)) {
                    const uint32_t cls = t_classify4(w[j]);
                    const uint32_t sp = t_movemask((((cls & 0x0F0F0F0Fu) + 0x7F7F7F7Fu) >> 7) & 0x01010101u);
                    const uint32_t dg = t_movemask((cls >> 4) & 0x01010101u);
                    sp32[m >> 1] |= sp << sh;
                    dg32[m >> 1] |= dg << sh;
                    df32[m >> 1] |= t_movemask(t_nonzero01(x)) << sh;
                } else {
                    sp32[m >> 1] |= sep0[m >> 1] & (0xFu << sh);
                    dg32[m >> 1] |= dig0[m >> 1] & (0xFu << sh);
                }
            }
        }
        if (!t_quad || !(16 * m < tlen)) {   // (the quad form has the title's rows in place already; rows behind it are cleared here)
            if (4 * m + 0 < z) s_w[4 * m + 0][tid] = q.x;
            if (4 * m + 1 < z) s_w[4 * m + 1][tid] = q.y;
            if (4 * m + 2 < z) s_w[4 * m + 2][tid] = q.z;
            if (4 * m + 3 < z) s_w[4 * m + 3][tid] = q.w;
        }
    }
    s_w[z][tid] = 0;
#pragma unroll
    for (uint32_t k = 0; k < W; ++k) {
        const uint64_t keep = t_lowbits(tlen > 64 * k ? tlen - 64 * k : 0);
        sepmask[k] = (((uint64_t)sp32[2 * k + 1] << 32) | sp32[2 * k]) & keep;
        digmask[k] = (((uint64_t)dg32[2 * k + 1] << 32) | dg32[2 * k]) & keep;
        diffmask[k] = ((uint64_t)df32[2 * k + 1] << 32) | df32[2 * k];
    }
}

// eight bytes of the lane's title starting at byte position pos (rows past the title read the zero row)
template <uint32_t W>
__device__ __forceinline__ void t_field8(uint32_t (*s_w)[TT_THREADS], int tid, uint32_t pos, uint32_t &w0, uint32_t &w1, uint32_t Z)
{
    // Z: the zero row (behind the rows that hold title dwords)
    const uint32_t r = pos >> 2, sh = (pos & 3) * 8;
    const uint32_t x0 = s_w[r < Z ? r : Z][tid], x1 = s_w[r + 1 < Z ? r + 1 : Z][tid], x2 = s_w[r + 2 < Z ? r + 2 : Z][tid];
    w0 = __builtin_amdgcn_alignbit(x1, x0, sh);
    w1 = __builtin_amdgcn_alignbit(x2, x1, sh);
}

template <uint32_t W>
// (round 3) z_rows, nf_cap: the job's longest title in dwords and the most fields a title of that length can have - the title
// rows and the field statistics live in dynamic LDS sized by them: 19.7 KB instead of 24.1 for cfg2's 50-byte titles, i.e. eight
// workgroups per CU (the CU's wave limit) instead of six.  The hardware counters put the kernel at 60 % VALU and 48 % scalar
// issue with LDS and scalar-load waits between them: more resident waves is what it can use.
__global__ __launch_bounds__(TT_THREADS) void k_title_stats_swar(Arena A, uint32_t z_rows, uint32_t nf_cap_flags)
{
    if (*A.err) return;
    const uint32_t nf_cap = nf_cap_flags & 0xffffu;   // (bit 16: quad-cooperative title loads)
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t slot_bias = sb.slot_bias;   // (read once: the stores below could alias A.sb)
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    constexpr uint32_t OWNED = TT_THREADS - 1;
    if (blockIdx.x * OWNED >= n) return;
    const int tid = threadIdx.x;
    const uint32_t lane = (uint32_t)tid & 63u;

    // field statistics in LDS: the 18 words of a FieldStat that this kernel updates (scalars 0..9, Hamming words 10..17), not
    // its 24-word global layout - with 64 fields the difference is what lets a sixth workgroup onto the CU
    constexpr uint32_t SW = 18;
    extern __shared__ __align__(16) uint32_t s_dyn[];   // [nf_cap * SW] field statistics, then [z_rows + 1][TT_THREADS] title dwords
    uint32_t *const s_stat = s_dyn;
    // value deltas between neighbouring records (tasks.cpp:149-166): inside a wave by shuffle, across the three wave
    // boundaries of the workgroup through these (last lane's value / first lane's value, per numeric candidate)
    __shared__ uint32_t s_last[NUM_CAND_CAP][TT_THREADS / 64], s_first[NUM_CAND_CAP][TT_THREADS / 64];
    __shared__ fmask_t s_nm_last[TT_THREADS / 64], s_nm_first[TT_THREADS / 64];
    uint32_t (*const s_w)[TT_THREADS] = reinterpret_cast<uint32_t (*)[TT_THREADS]>(s_dyn + ((nf_cap * SW + 3) & ~3u));
    // record 0's field table: scalar loads straight from global memory (written by k_rec0_fields, read-only here)
    const crec0p R0 = (crec0p)(uintptr_t)(A.rec0 + s);
    const uint32_t r0_nf0 = R0->nf0;
    const uint32_t nf0 = r0_nf0 < nf_cap ? r0_nf0 : nf_cap;   // (more separators than the record 0 table holds is an error already)

    for (uint32_t k = tid; k < nf0 * SW; k += TT_THREADS) s_stat[k] = 0;
    __syncthreads();

    const int64_t i = (int64_t)blockIdx.x * OWNED + tid - 1;  // record index inside the sub-block
    const bool valid = i >= 0 && i < (int64_t)n;
    const bool owned = valid && tid >= 1;
    const uint32_t g = valid ? rec_lo + (uint32_t)i : rec_lo;
    const uint32_t ls = A.ls[g], te = A.te[g];
    const uint32_t tlen = valid ? te - ls + 1 : 0;            // title bytes including the closing newline

    uint64_t sepmask[W], digmask[W], diffmask[W];
    t_load_title<W>(A, ls, tlen, s_w, tid, sepmask, digmask, (cu32p)(uintptr_t)(A.rec0[s].title), diffmask,
                    (cu32p)(uintptr_t)(A.rec0[s].sepm), (cu32p)(uintptr_t)(A.rec0[s].digm), z_rows, (nf_cap_flags >> 16) != 0);
    uint32_t nsep = 0;
#pragma unroll
    for (uint32_t k = 0; k < W; ++k) {
        if (owned) A.tsep[(size_t)g * W + k] = sepmask[k];
        nsep += (uint32_t)__popcll(sepmask[k]);
    }
    if (valid && nsep != r0_nf0) atomicMin(A.err, (int32_t)ERR_FIELDS);  // phyNGSC.cpp:417-421 only warns

    uint32_t pos = 0;
    fmask_t nummask = 0;
    // (round 3) Fields that EVERY lane of the wave holds exactly as record 0 holds them - same place, same bytes, same
    // separator behind them: the constant fields of a title, as long as no field in front of them has changed its length -
    // are recognised from the title-wide difference mask and cost ~15 instructions instead of ~100: nothing about them can
    // change a statistic except through record 0's own lane (whose wave takes the full path) and the zero value delta.
    bool pos_aligned = true;                                  // (wave-uniform) every lane's field starts where record 0's does
    const bool wave_has_rec0 = __any(owned && i == 0);
    for (uint32_t f = 0; f < nf0; ++f) {
        const uint32_t fd0 = R0->fdw[f][0], fd1 = R0->fdw[f][1];
        const uint32_t pos0 = fd0 & 0xffffu, len0 = fd0 >> 16, kk = fd1 & 0xffu, wo = fd1 >> 16;
        if (pos_aligned && !wave_has_rec0 && pos0 + len0 < 64 * W && len0 < 64) {
            const uint64_t fdw = t_window<W>(diffmask, pos0);
            if (!__any(valid && (fdw & t_lowbits(len0 + 1)) != 0)) {
                if (kk != 0xff) {   // record 0's field is numeric, so it is in every lane: value = record 0's
                    nummask |= (fmask_t)1 << f;
                    if (kk < NUM_CAND_CAP) {
                        const uint32_t v0 = R0->val0[f];
                        if (owned) A.tv[(size_t)kk * A.tw_stride + g + slot_bias] = v0;
                        if (lane == 63) s_last[kk][tid >> 6] = v0;
                        if (lane == 0) s_first[kk][tid >> 6] = v0;
                        const bool q = owned && i >= 1 && lane >= 1;   // the delta to the previous record is 0
                        if (__any(q)) {
                            if (q) {
                                const uint32_t bd = bias(0);
                                if (bd > s_stat[f * SW + 8]) atomicMax(&s_stat[f * SW + 8], bd);
                                if (~bd > s_stat[f * SW + 9]) atomicMax(&s_stat[f * SW + 9], ~bd);
                            }
                        }
                    }
                }
                pos += len0 + 1;
                continue;
            }
        }
        const uint32_t len = t_next_set<W>(sepmask, pos);
        const uint32_t minlen = len < len0 ? len : len0;

        // A field that starts where record 0's field starts (every lane of the wave; true for all fields in front of the
        // first one whose length varies) is compared through the title-wide difference mask: its bits [pos, pos + len)
        // are the field's differing bytes.  No funnel-shifted LDS reads, no eight-byte steps.
        uint32_t w0 = 0, w1 = 0, diff = 0, c8 = 0;
        const bool aligned = len0 <= 64 && !__any(valid && (pos != pos0 || len > 64));
        if (aligned) {
            const uint64_t fd = t_window<W>(diffmask, pos0);
            diff = (fd & t_lowbits(len0)) != 0 ? 1u : 0u;
            const uint64_t hm = fd & t_lowbits(minlen);   // Hamming mask: tasks.cpp:187-193
            if (__any(valid && hm != 0)) {
                if (owned && (uint32_t)hm) t_lds_or(&s_stat[f * SW + 10], (uint32_t)hm);
                if (owned && (uint32_t)(hm >> 32)) t_lds_or(&s_stat[f * SW + 11], (uint32_t)(hm >> 32));
            }
            if (kk != 0xff && __any(valid && !(len == len0 && diff == 0))) {   // the value parser below wants the field's first bytes
                t_field8<W>(s_w, tid, pos, w0, w1, z_rows);
                const uint64_t km = t_lowbytes(len);
                w0 &= (uint32_t)km;
                w1 &= (uint32_t)(km >> 32);
            }
        } else
        do {
            uint32_t x0, x1;
            t_field8<W>(s_w, tid, pos + 8 * c8, x0, x1, z_rows);
            const uint32_t left = len > 8 * c8 ? len - 8 * c8 : 0;
            const uint64_t km = t_lowbytes(left);
            x0 &= (uint32_t)km;
            x1 &= (uint32_t)(km >> 32);
            if (c8 == 0) { w0 = x0; w1 = x1; }
            const bool in0 = 8 * c8 < len0;
            const uint32_t a0 = in0 ? R0->al[wo + 2 * c8] : 0, a1 = in0 ? R0->al[wo + 2 * c8 + 1] : 0;
            const uint32_t d0 = x0 ^ a0, d1 = x1 ^ a1;
            diff |= d0 | d1;
            if (__any(d0 | d1)) {
                const uint64_t hm = t_lowbytes(minlen > 8 * c8 ? minlen - 8 * c8 : 0);   // Hamming mask: tasks.cpp:187-193
                const uint32_t hb = t_movemask(t_nonzero01(d0 & (uint32_t)hm)) | (t_movemask(t_nonzero01(d1 & (uint32_t)(hm >> 32))) << 4);
                if (hb && owned && c8 < 32) t_lds_or(&s_stat[f * SW + 10 + (c8 >> 2)], hb << (8 * (c8 & 3)));
            }
            ++c8;
        } while (__any(len > 8 * c8) || 8 * c8 < len0);
        const bool same = len == len0 && diff == 0;

        // numeric test and value (utils::is_num / to_num, utils.h:107-125); only a numeric record 0 can leave the
        // field numeric, and record 0's own lane marks it non-numeric otherwise
        bool isn = false, ovf = false;
        uint32_t v = 0;
        if (kk != 0xff && !__any(valid && !same)) {
            // every lane holds record 0's (numeric) field: nothing to parse
            isn = true;
            v = R0->val0[f];
            ovf = ((fd1 >> 8) & 0xffu) != 0;
            if (kk < NUM_CAND_CAP && owned) A.tv[(size_t)kk * A.tw_stride + g + slot_bias] = v;
            nummask |= (fmask_t)1 << f;
        } else if (kk != 0xff) {
            isn = len > 0 && t_all_set<W>(digmask, pos, len) && !(len > 1 && (w0 & 0xffu) == '0');
            v = t_parse8(w0, w1, len);
            if (__any(isn && len > 8)) {
                if (isn && len > 8) {   // nine digits and more: digit by digit, wrapping like utils::to_num (utils.h:117-125)
                    v = 0;
                    bool big = false;
                    for (uint32_t j = 0; j < len; ++j) {
                        const uint32_t p = pos + j;
                        const uint32_t d = ((s_w[p >> 2][tid] >> (8 * (p & 3))) & 0xffu) - '0';
                        const uint64_t t10 = (uint64_t)v * 10 + d;
                        big = big || (uint32_t)(t10 >> 32) != 0;
                        v = (uint32_t)t10;
                    }
                    ovf = big || (v >> 31) != 0;   // beyond INT32_MAX (parse_num): a negative int32 for the reference, or a value it stores modulo 2^32
                }
            }
            if (kk < NUM_CAND_CAP && owned) A.tv[(size_t)kk * A.tw_stride + g + slot_bias] = v;   // k_title_measure_reg<true> and k_title_hist read it back
            nummask |= (fmask_t)isn << f;
        }
        if (kk < NUM_CAND_CAP) {   // (uniform) delta to the previous record, both numeric
            const uint32_t pv = (uint32_t)__shfl_up((int)v, 1);
            const bool pisn = __shfl_up((int)isn, 1) != 0;
            if (lane == 63) s_last[kk][tid >> 6] = v;
            if (lane == 0) s_first[kk][tid >> 6] = v;
            if (owned && i >= 1 && lane >= 1 && isn && pisn) {
                const uint32_t bd = bias((int32_t)(v - pv));
                if (bd > s_stat[f * SW + 8]) atomicMax(&s_stat[f * SW + 8], bd);
                if (~bd > s_stat[f * SW + 9]) atomicMax(&s_stat[f * SW + 9], ~bd);
            }
        }

        // A field identical to record 0's cannot change any statistic beyond what record 0's own lane
        // contributes, so only differing fields (and record 0 itself) touch the counters.
        const bool upd = owned && (!same || i == 0);
        if (__any(upd)) {
            if (upd) {
                uint32_t *st = s_stat + f * SW;
                const uint32_t a0 = st[0], a1 = st[1], a2 = st[2], a3 = st[3], a4 = st[4], a5 = st[5], a6 = st[6], a7 = st[7];
                const uint32_t bv = bias((int32_t)v);
                const bool n0 = len > a0, n1 = ~len > a1, n2 = !same && !a2, n3 = len != len0 && !a3, n4 = !isn && !a4,
                           n5 = isn && ovf && !a5, n6 = isn && bv > a6, n7 = isn && ~bv > a7;
                if (n0 | n1 | n2 | n3 | n4 | n5 | n6 | n7) {
                    if (n0) atomicMax(st + 0, len);
                    if (n1) atomicMax(st + 1, ~len);
                    if (n2) atomicMax(st + 2, 1u);
                    if (n3) atomicMax(st + 3, 1u);
                    if (n4) atomicMax(st + 4, 1u);
                    if (n5) atomicMax(st + 5, 1u);
                    if (n6) atomicMax(st + 6, bv);
                    if (n7) atomicMax(st + 7, ~bv);
                }
            }
        }
        pos += len + 1;
        pos_aligned = pos_aligned && !__any(valid && len != len0);
    }
    if (lane == 63) s_nm_last[tid >> 6] = nummask;
    if (lane == 0) s_nm_first[tid >> 6] = (owned && i >= 1) ? nummask : (fmask_t)0;
    __syncthreads();
    {
        // the first lanes of waves 1..3 against the last lanes of waves 0..2
        const uint32_t ncand = R0->ncand, kk = (uint32_t)tid >> 2, w = (uint32_t)tid & 3u;
        if (w >= 1 && kk < ncand) {
            const uint32_t ff = A.rec0[s].cand_f[kk];
            if (((s_nm_first[w] & s_nm_last[w - 1]) >> ff) & 1) {
                const uint32_t bd = bias((int32_t)(s_first[kk][w] - s_last[kk][w - 1]));
                if (bd > s_stat[ff * SW + 8]) atomicMax(&s_stat[ff * SW + 8], bd);
                if (~bd > s_stat[ff * SW + 9]) atomicMax(&s_stat[ff * SW + 9], ~bd);
            }
        }
    }
    __syncthreads();
    uint32_t *gst = reinterpret_cast<uint32_t *>(A.fstat + (size_t)s * A.fcap);
    for (uint32_t kx = tid; kx < nf0 * SW; kx += TT_THREADS) {
        uint32_t x = s_stat[kx];
        if (!x) continue;
        const uint32_t w = kx % SW, gx = kx / SW * FSTAT_WORDS + w;   // (a FieldStat in global memory has all its words)
        if (w >= 10) atomicOr(gst + gx, x);
        else atomicMax(gst + gx, x);
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
void launch_rec0_fields(hipStream_t st, const Arena &A)
{
    hipLaunchKernelGGL(k_rec0_fields, dim3(A.n_sb), dim3(64), 0, st, A);
}

void launch_title_stats_reg(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    dim3 grid((max_rec + TT_THREADS - 2) / (TT_THREADS - 1), A.n_sb);
    hipLaunchKernelGGL(k_title_stats_reg, grid, dim3(TT_THREADS), 0, st, A);
}

void launch_title_stats_swar(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t max_title)
{
    dim3 grid((max_rec + TT_THREADS - 2) / (TT_THREADS - 1), A.n_sb);
    // LDS by the job's longest title: its dwords (+ the zero row) and the fields it can have at most (one per separator byte)
    const uint32_t W = A.tsep_words == 2 ? 2u : 1u;
    uint32_t z = (max_title + 3) / 4;
    if (z > 16 * W) z = 16 * W;
    if (z < 1) z = 1;
    uint32_t nfc = max_title < MAX_FIELDS ? (max_title < 1 ? 1u : max_title) : MAX_FIELDS;
    const size_t lds = ((size_t)((nfc * 18 + 3) & ~3u) + (size_t)(z + 1) * TT_THREADS) * 4;
    const uint32_t quad = 1u;
    if (W == 2) hipLaunchKernelGGL(k_title_stats_swar<2>, grid, dim3(TT_THREADS), lds, st, A, z, nfc | (quad << 16));
    else hipLaunchKernelGGL(k_title_stats_swar<1>, grid, dim3(TT_THREADS), lds, st, A, z, nfc | (quad << 16));
}

void launch_title_measure_reg(hipStream_t st, const Arena &A, uint32_t max_rec, bool swar)
{
    dim3 grid((max_rec + TT_THREADS - 1) / TT_THREADS, A.n_sb);
    if (swar && A.tsep_words == 2) hipLaunchKernelGGL((k_title_measure_reg<true, 2>), grid, dim3(TT_THREADS), 0, st, A);
    else if (swar) hipLaunchKernelGGL((k_title_measure_reg<true, 1>), grid, dim3(TT_THREADS), 0, st, A);
    else hipLaunchKernelGGL((k_title_measure_reg<false, 1>), grid, dim3(TT_THREADS), 0, st, A);
}

void launch_title_pack_reg(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    dim3 grid((max_rec + TT_THREADS - 1) / TT_THREADS, A.n_sb);
    hipLaunchKernelGGL(k_title_pack_reg, grid, dim3(TT_THREADS), 0, st, A);
}

}  // namespace ngsc
