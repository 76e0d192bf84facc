// kernels_tile.hip - LDS-tiled per-record kernels (the fast path).
//
// A tile is a run of R consecutive records of one sub-block (R = 256, 128 or 64: a multiple of 32, so the
// 32-record title groups never straddle tiles) whose bytes [ls[i0], ls[i1]) are staged into LDS with
// coalesced 16-byte loads.  Every stage then works out of LDS with one thread per record (titles) or one
// thread per record slice (sequence / quality; T = 256 / R threads share a record), so HBM sees each raw
// byte once per kernel and all random access, histogramming, code lookup and bit merging stays on chip:
//
//   k_title_stats_tile     AnalyzeTitleFields reductions                                (tasks.cpp:22-193)
//   k_seq_qual_stats_tile  N-transfer test, DNA counts, per-position quality histogram   (phyNGSC.cpp:462-653,
//                                                                                         tasks.cpp:260-286)
//   k_measure_tile         parses each title once more, caches its non-constant fields, derives the 32-record
//                          group flags and the title bits per record; quality / DNA bits per slice and tile
//   k_scan_tile            per-sub-block exclusive scans of group bytes and tile bit totals
//   k_pack_tile            StoreTitle body (from the cached fields), StoreQuality and StoreDNA payloads: every
//                          lane accumulates its own bits and ORs whole words into LDS staging; one coalesced
//                          copy-out per stream                                (tasks.cpp:427-507, 544-557, 609-619)
//
// Loops are written with wave-uniform trip counts and predicated bodies: the first version of these kernels
// used nested per-lane while-loops and spent 80 % of its time in scalar exec-mask bookkeeping (SQ_INSTS_SALU).
// Integer / byte work bounded by HBM (tile load) and LDS / VALU issue; no MFMA.
#include "ngsc_kernels.h"
#include <cstdlib>

namespace ngsc {

constexpr int TL_THREADS = 256;
constexpr int TL_WAVES = TL_THREADS / 64;

__device__ __forceinline__ uint32_t wsum(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ uint32_t wmax(uint32_t v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { uint32_t o = __shfl_xor(v, d); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ uint32_t wscan_incl(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// Workgroup (256 threads) exclusive scan; returns this thread's exclusive prefix, *total = sum.  s_part[TL_WAVES].
__device__ __forceinline__ uint32_t wg_scan_excl256(uint32_t v, uint32_t *s_part, uint32_t *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wscan_incl(v);
    __syncthreads();  // s_part may still be read from a previous call
    if (lane == 63) s_part[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < TL_WAVES; ++w) {
        uint32_t p = s_part[w];
        if (w < wave) base += p;
        tot += p;
    }
    *total = tot;
    return base + incl - v;
}

// Stages bytes [b0, ls[i1]) of the job buffer into s_tile, b0 = ls[i0] & ~15; returns b0, so that
// s_tile[pos - b0] == buf[pos] for every pos inside the tile.  (No rebased pointer: a negative offset on an
// LDS pointer that later gets widened to a flat address does not wrap back.)  Ends with a barrier.
__device__ __forceinline__ uint32_t load_tile(const Arena &A, uint32_t i0, uint32_t i1, uint8_t *s_tile)
{
    const uint32_t b0 = A.ls[i0] & ~15u, b1 = A.ls[i1];
    const uint32_t n16 = (b1 - b0 + 15) >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(A.buf + b0);
    uint4 *dst = reinterpret_cast<uint4 *>(s_tile);
    for (uint32_t k = threadIdx.x; k < n16; k += blockDim.x) {
        if (b0 + 16 * k + 16 <= A.buf_len) dst[k] = src[k];
        else {
            for (uint32_t x = 0; x < 16; ++x) {
                uint32_t pos = b0 + 16 * k + x;
                s_tile[16 * k + x] = pos < A.buf_len ? A.buf[pos] : 0;
            }
        }
    }
    __syncthreads();
    return b0;
}

__device__ __forceinline__ void lds_max_t(uint32_t *p, uint32_t v) { if (v > *p) atomicMax(p, v); }
__device__ __forceinline__ void lds_or_t(uint32_t *p, uint32_t v) { if ((*p & v) != v) atomicOr(p, v); }

// A, C, G, T without a table: bit (c - 'A') of a 32-bit mask
__device__ __forceinline__ bool is_acgt(uint32_t c)
{
    const uint32_t x = c - 'A';
    return x < 32u && ((0x00080045u >> x) & 1u);   // bits 0 (A), 2 (C), 6 (G), 19 (T)
}

// branch-free membership test for the title separators " ._,=:/-#\n" (phyNGSC.cpp:208)
__device__ __forceinline__ bool sep_fast(uint32_t c)
{
    const uint32_t m0 = 1u << 10;                                                                    // '\n'
    const uint32_t m1 = (1u << 0) | (1u << 3) | (1u << 12) | (1u << 13) | (1u << 14) | (1u << 15) |  // ' ' # , - . /
                        (1u << 26) | (1u << 29);                                                     // : =
    const uint32_t m2 = 1u << 31;                                                                    // '_'
    const uint32_t w = c >> 5;
    const uint32_t m = w == 0 ? m0 : (w == 1 ? m1 : (w == 2 ? m2 : 0u));
    return (m >> (c & 31)) & 1u;
}

// ====================================================================================================
// Title statistics.  blockDim.x = R threads; a tile is R-1 owned records plus the previous record as delta
// halo (thread 0).  One flat pass over the title bytes per lane: separator test, decimal accumulate and
// comparison with record 0 of the sub-block (whose title and field table sit in LDS) in the same loop body.
// ====================================================================================================
constexpr uint32_t REC0_TITLE_CAP = 512;

__global__ __launch_bounds__(TL_THREADS) void k_title_stats_tile(Arena A, uint32_t dbg)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_dyn[];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t owned_per_tile = blockDim.x - 1;
    if (blockIdx.x * owned_per_tile >= n) return;
    const int tid = threadIdx.x;

    __shared__ uint32_t s_stat[MAX_FIELDS * 24];
    __shared__ uint32_t s_val[NUM_CAND_CAP][TL_THREADS];  // values of the numeric candidate fields
    __shared__ uint32_t s_nummask[TL_THREADS];
    __shared__ uint16_t s_f0_start[MAX_FIELDS], s_f0_len[MAX_FIELDS];
    __shared__ uint8_t s_kmap[MAX_FIELDS], s_cand_f[NUM_CAND_CAP];
    __shared__ uint32_t s_nf0, s_ncand;
    __shared__ uint8_t s_t0[REC0_TITLE_CAP];

    const int64_t ifirst = (int64_t)blockIdx.x * owned_per_tile - 1;
    const uint32_t r0 = ifirst < 0 ? 0u : (uint32_t)ifirst;
    const uint64_t rend = ((uint64_t)blockIdx.x + 1) * owned_per_tile;
    const uint32_t r1 = rend < n ? (uint32_t)rend : n;  // exclusive
    const uint32_t ls0 = A.ls[rec_lo], te0 = A.te[rec_lo];
    const uint32_t t0len = te0 - ls0 + 1;
    if (t0len > REC0_TITLE_CAP) { if (tid == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
    for (uint32_t k = tid; k < t0len; k += blockDim.x) s_t0[k] = A.buf[ls0 + k];
    for (int k = tid; k < (int)(MAX_FIELDS * 24); k += blockDim.x) s_stat[k] = 0;
    const uint32_t b0 = (dbg & 2) ? (A.ls[rec_lo + r0] & ~15u) : load_tile(A, rec_lo + r0, rec_lo + r1, s_dyn);
    const uint8_t *tb = s_dyn;   // tile bytes, indexed by (position - b0)

    if (tid == 0) {
        // field table of record 0 (phyNGSC.cpp:345-379) + numeric candidates (only they can stay numeric)
        uint32_t nf0 = 0, ncand = 0, start = 0;
        for (uint32_t p = 0; p < t0len; ++p) {
            if (!is_sep(s_t0[p])) continue;
            if (nf0 < MAX_FIELDS) {
                uint32_t v; bool o;
                s_f0_start[nf0] = (uint16_t)start;
                s_f0_len[nf0] = (uint16_t)(p - start);
                uint8_t km = 0xff;
                if (parse_num(s_t0, start, p, v, o)) {
                    if (ncand < NUM_CAND_CAP) { km = (uint8_t)ncand; s_cand_f[ncand++] = (uint8_t)nf0; }
                    else {
                        km = 0xfe;
                        if (blockIdx.x == 0) A.fstat[(size_t)s * MAX_FIELDS + nf0].no_delta = 1;
                    }
                }
                s_kmap[nf0] = km;
            }
            ++nf0;
            start = p + 1;
        }
        s_nf0 = nf0;
        s_ncand = ncand;
        if (blockIdx.x == 0) A.sb[s].nf = nf0;
        if (nf0 > MAX_FIELDS) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED);
    }
    __syncthreads();
    const uint32_t nf0 = s_nf0;
    if (nf0 > MAX_FIELDS) return;

    const int64_t i = (int64_t)blockIdx.x * owned_per_tile + tid - 1;  // record index inside the sub-block
    const bool valid = i >= 0 && i < (int64_t)n;
    const bool owned = valid && tid >= 1;
    const uint32_t g = valid ? rec_lo + (uint32_t)i : rec_lo;
    const uint32_t pc = valid ? A.ls[g] - b0 : 0;
    const uint32_t tl = valid ? A.te[g] - A.ls[g] + 1 : 0;  // title bytes including the newline
    const uint32_t maxl = wmax(tl);

    // Per-lane parser state kept in integer registers (flag bits, not bools: the compiler turns bool state
    // into scalar lane masks and spends most of the kernel in s_and/s_or bookkeeping).  The only branches in
    // the byte loop are wave-level: "some lane sits on a separator", "some lane found a byte that differs".
    //   st bit0: non-digit seen, bit1: leading zero, bit2: value > INT32_MAX, bit3: differs from record 0
    uint32_t f = 0, k = 0, nummask = 0, v = 0, stf = 0;
    uint32_t f0len = nf0 ? s_f0_len[0] : 0, f0start = nf0 ? s_f0_start[0] : 0;
    const uint32_t maxl8 = (dbg & 1) ? 0u : ((maxl + 7) & ~7u);
    for (uint32_t base = 0; base < maxl8; base += 8) {
        uint32_t cb[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) cb[u] = tb[(base + u < tl) ? pc + base + u : pc];  // 8 loads in flight
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) {
            const uint32_t it = base + u;
            const uint32_t c = cb[u];
            const uint32_t act = it < tl ? 1u : 0u;
            const uint32_t sepb = act & (sep_fast(c) ? 1u : 0u);
            const uint32_t body = act & (sepb ^ 1u);
            const uint32_t d = c - '0';
            const uint32_t dig = d <= 9 ? 1u : 0u;
            const uint32_t ovn = (v > 214748364u || (v == 214748364u && d > 7)) ? 1u : 0u;
            const uint32_t ld0 = (k == 0 && d == 0) ? 1u : 0u;
            const uint32_t inr = body & (k < f0len ? 1u : 0u);          // f0len == 0 once f >= nf0
            const uint32_t c0 = s_t0[f0start + (inr ? k : 0)];
            const uint32_t nd = inr & (c != c0 ? 1u : 0u);
            stf |= (body & (dig ^ 1u)) | ((body & ld0) << 1) | ((body & dig & ovn) << 2) | (nd << 3);
            v = (body & dig) ? v * 10 + d : v;                             // wraps like utils::to_num
            if (__any(nd)) {
                if (nd && owned && k < 256) lds_or_t(&s_stat[f * 24 + 10 + (k >> 5)], 1u << (k & 31));
            }
            k += body;
            if (__any(sepb)) {
                if (sepb) {
                    if (f < nf0) {
                        const uint32_t len = k, len0 = f0len;
                        const bool isn = !(stf & 1) && len > 0 && !(len > 1 && (stf & 2));  // utils::is_num
                        // A field identical to record 0's cannot change any statistic beyond what record 0's own
                        // thread contributes, so only differing fields (and record 0 itself) touch the counters.
                        const bool same = len == len0 && !(stf & 8);
                        if (owned && (!same || i == 0)) {
                            uint32_t *st = s_stat + f * 24;
                            const uint32_t a0 = st[0], a1 = st[1], a2 = st[2], a3 = st[3], a4 = st[4], a5 = st[5],
                                           a6 = st[6], a7 = st[7];
                            const uint32_t bv = bias((int32_t)v);
                            const bool n0 = len > a0, n1 = ~len > a1, n2 = !same && !a2, n3 = len != len0 && !a3,
                                       n4 = !isn && !a4, n5 = isn && (stf & 4) && !a5, n6 = isn && bv > a6,
                                       n7 = isn && ~bv > a7;
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
                        const uint32_t kk = s_kmap[f];
                        if (kk < NUM_CAND_CAP) s_val[kk][tid] = v;
                        nummask |= (uint32_t)isn << f;
                    }
                    ++f; k = 0; v = 0; stf = 0;
                    f0len = f < nf0 ? s_f0_len[f] : 0;
                    f0start = f < nf0 ? s_f0_start[f] : 0;
                }
            }
        }
    }
    if (valid && f != nf0 && !dbg) atomicMin(A.err, (int32_t)ERR_FIELDS);  // phyNGSC.cpp:417-421 only warns
    s_nummask[tid] = nummask;
    __syncthreads();
    if (owned && i >= 1) {
        const uint32_t both = nummask & s_nummask[tid - 1];
        const uint32_t ncand = s_ncand;
        for (uint32_t kk = 0; kk < ncand; ++kk) {
            const uint32_t ff = s_cand_f[kk];
            if ((both >> ff) & 1) {
                int32_t d = (int32_t)(s_val[kk][tid] - s_val[kk][tid - 1]);  // tasks.cpp:149-166
                lds_max_t(s_stat + ff * 24 + 8, bias(d));
                lds_max_t(s_stat + ff * 24 + 9, ~bias(d));
            }
        }
    }
    __syncthreads();
    uint32_t *gst = reinterpret_cast<uint32_t *>(A.fstat + (size_t)s * MAX_FIELDS);
    for (uint32_t kx = tid; kx < nf0 * 24; kx += blockDim.x) {
        uint32_t x = s_stat[kx];
        if (!x) continue;
        uint32_t w = kx % 24;
        if (w >= 10 && w < 18) atomicOr(gst + kx, x);
        else if (w < 10) atomicMax(gst + kx, x);
    }
}

// ====================================================================================================
// Slices: thread tid of a 256-thread workgroup works on record r = tid / T of the tile, positions
// [j0, j1) = [t*L/T, (t+1)*L/T) with t = tid % T and T = 256 / R threads per record.
// ====================================================================================================
struct Slice {
    bool valid;
    uint32_t r;       // tile-relative record
    uint32_t t;       // slice index inside the record
    uint32_t s0, q0;  // first base / first quality byte (tile-relative offsets)
    uint32_t L;       // read length
    uint32_t j0, j1;
};

__device__ __forceinline__ Slice make_slice(const Arena &A, uint32_t i0, uint32_t nt, uint32_t T, uint32_t b0)
{
    Slice S;
    S.r = threadIdx.x / T;
    S.t = threadIdx.x % T;
    S.valid = S.r < nt;
    S.s0 = S.q0 = S.L = S.j0 = S.j1 = 0;
    if (S.valid) {
        uint32_t te = A.te[i0 + S.r], se = A.se[i0 + S.r];
        S.L = se - te - 1;
        S.s0 = te + 1 - b0;
        S.q0 = se + 3 - b0;
        S.j0 = (uint32_t)(((uint64_t)S.t * S.L) / T);
        S.j1 = (uint32_t)(((uint64_t)(S.t + 1) * S.L) / T);
    }
    return S;
}

// OR / add over the T adjacent lanes that share a record (T in {1, 2, 4})
__device__ __forceinline__ uint32_t rec_or(uint32_t v, uint32_t T)
{
    if (T >= 2) v |= __shfl_xor(v, 1);
    if (T >= 4) v |= __shfl_xor(v, 2);
    return v;
}
__device__ __forceinline__ uint32_t rec_sum(uint32_t v, uint32_t T)
{
    if (T >= 2) v += __shfl_xor(v, 1);
    if (T >= 4) v += __shfl_xor(v, 2);
    return v;
}

// ====================================================================================================
// DNA / quality statistics
// ====================================================================================================
constexpr uint32_t HCOLS = 64, HSTRIDE = 65;

__global__ __launch_bounds__(TL_THREADS) void k_seq_qual_stats_tile(Arena A, uint32_t R, uint32_t tile_bytes,
                                                                    uint32_t lds_rows)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_dyn[];  // [tile_bytes] tile, then [lds_rows][HSTRIDE] histogram
    __shared__ uint32_t s_occ[256];
    __shared__ uint8_t s_amb[256];
    __shared__ uint32_t s_red[3];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t t0 = blockIdx.x * R;
    if (t0 >= n) return;
    const uint32_t nt = t0 + R < n ? R : n - t0;
    const uint32_t i0 = rec_lo + t0;
    const uint32_t T = TL_THREADS / R;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_dyn + tile_bytes);

    for (uint32_t k = tid; k < lds_rows * HSTRIDE; k += TL_THREADS) s_hist[k] = 0;
    s_occ[tid] = 0;
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (tid < 3) s_red[tid] = 0;
    const uint32_t b0 = load_tile(A, i0, i0 + nt, s_dyn);
    const uint8_t *tb = s_dyn;
    const Slice S = make_slice(A, i0, nt, T, b0);

    // line 3 of every record must be a bare '+' (README.md:97-103); the reference assumes it
    if ((uint32_t)tid < nt) {
        uint32_t se = A.se[i0 + tid] - b0;
        if (tb[se + 1] != '+' || tb[se + 2] != '\n') atomicMin(A.err, (int32_t)ERR_FORMAT);
    }
    if (t0 == 0 && tid == 0) {  // colour-space detection on the sub-block's first record, phyNGSC.cpp:478
        uint8_t c1 = tb[S.s0 + 1];
        if (S.L >= 2 && c1 >= '0' && c1 <= '3') atomicMin(A.err, (int32_t)ERR_COLORSPACE);
    }
    const uint32_t len = S.j1 - S.j0;
    const uint32_t maxl = wmax(len);
    // pass 1: N-transfer test (phyNGSC.cpp:549-571).  The serial loop stops at the first failure, so the
    // outcome is "some ambiguous base" && "no failure anywhere": order independent.
    uint32_t fl = 0, namb = 0;  // bit0 any, bit1 bad
    for (uint32_t base = 0; base < maxl; base += 8) {
        uint32_t cc[8];
        bool anyamb = false;
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) cc[u] = tb[S.s0 + ((base + u < len) ? S.j0 + base + u : S.j0)];
#pragma unroll
        for (uint32_t u = 0; u < 8; ++u) anyamb = anyamb || (base + u < len && !is_acgt(cc[u]));
        if (__any(anyamb)) {  // rare: some lane of the wave holds a non-ACGT base in this chunk
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                if (base + u < len && !is_acgt(cc[u])) {
                    const uint32_t a = s_amb[cc[u]];
                    if (a == 0) fl |= 2;
                    else {
                        uint8_t q = tb[S.q0 + S.j0 + base + u];
                        fl |= (q < 33 || q > 40) ? 2u : 1u;
                        ++namb;
                    }
                }
            }
        }
    }
    fl = rec_or(fl, T);
    namb = rec_sum(namb, T);
    const bool xfer = (fl & 1) && !(fl & 2);
    const uint32_t kept = S.L - (xfer ? namb : 0);
    const bool rec_lead = S.valid && S.t == 0;
    if (rec_lead) A.slen[i0 + S.r] = kept | ((uint32_t)xfer << 31);
    {
        uint32_t mq = wmax(rec_lead ? S.L : 0), ms = wmax(rec_lead ? kept : 0), sbs = wsum(rec_lead ? S.L : 0);
        if (lane == 0) { atomicMax(&s_red[0], mq); atomicMax(&s_red[1], ms); atomicAdd(&s_red[2], sbs); }
    }
    // pass 2: histograms.  Lanes start at different positions of their slice (rotation) so that
    // neighbouring lanes update different histogram rows at the same time.
    uint32_t cA = 0, cC = 0, cG = 0, cT = 0;
    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    {
        const uint32_t rot = len ? (uint32_t)lane % len : 0;
        const bool wave_xfer = __any(xfer);
        for (uint32_t base = 0; base < maxl; base += 8) {
            uint32_t cc[8], qq[8], jj[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {   // 16 LDS loads in flight
                uint32_t x = rot + base + u;
                if (x >= len) x -= len;
                jj[u] = S.j0 + ((base + u < len) ? x : 0);
                cc[u] = tb[S.s0 + jj[u]];
                qq[u] = tb[S.q0 + jj[u]];
            }
            uint32_t cnt4 = 0;  // per-chunk A/C/G/T counters, 8 bits each
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const bool act = base + u < len;
                const uint32_t c = cc[u];
                const bool acgt = is_acgt(c);
                bool drop = false;
                uint32_t qv = qq[u];
                if (wave_xfer || __any(act && !acgt)) {   // rare: transfer records or non-ACGT symbols around
                    const uint32_t a = s_amb[c];
                    drop = xfer && a > 1;
                    if (drop) qv = transfer_quality(a, (uint8_t)qv);  // phyNGSC.cpp:579
                    if (act && !drop && !acgt) atomicAdd(&s_occ[c], 1u);
                }
                const uint32_t row = jj[u] + 1, col = qv - 33;
                const bool in_lds = act && col < HCOLS && row < lds_rows;
                if (in_lds) atomicAdd(&s_hist[row * HSTRIDE + col], 1u);
                if (__any(act && !in_lds)) {
                    if (act && !in_lds) atomicAdd(&qh[(size_t)row * 256 + qv], 1u);
                }
                // A=0x41 C=0x43 G=0x47 T=0x54: (c >> 1) & 3 = 0, 1, 3, 2
                cnt4 += (act && !drop && acgt) ? (1u << (((c >> 1) & 3u) * 8u)) : 0u;
            }
            cA += cnt4 & 0xffu; cC += (cnt4 >> 8) & 0xffu; cT += (cnt4 >> 16) & 0xffu; cG += cnt4 >> 24;
        }
    }
    cA = wsum(cA); cC = wsum(cC); cG = wsum(cG); cT = wsum(cT);
    if (lane == 0) {
        atomicAdd(&s_occ['A'], cA); atomicAdd(&s_occ['C'], cC); atomicAdd(&s_occ['G'], cG); atomicAdd(&s_occ['T'], cT);
    }
    __syncthreads();
    for (uint32_t k = tid; k < lds_rows * HCOLS; k += TL_THREADS) {
        uint32_t row = k / HCOLS, col = k % HCOLS;
        uint32_t x = s_hist[row * HSTRIDE + col];
        if (x) atomicAdd(&qh[(size_t)row * 256 + 33 + col], x);
    }
    if (s_occ[tid]) atomicAdd(&A.dna_occ[(size_t)s * 256 + tid], s_occ[tid]);
    if (tid == 0) {
        atomicMax(&A.sb[s].max_q, s_red[0]);
        atomicMax(&A.sb[s].max_s, s_red[1]);
        atomicAdd(&A.sb[s].n_bases, s_red[2]);
    }
}

// ====================================================================================================
// Measure / pack common pieces
// ====================================================================================================
struct SqLut {
    const uint32_t *qtab;  // LDS or global; row stride qstride
    uint32_t qstride;
    const uint32_t *dcode; // LDS [256]
    const uint8_t *qcode;  // LDS [256]
    const uint8_t *amb;    // LDS [256]
};

__device__ __forceinline__ void load_luts(const Arena &A, uint32_t s, const SbState &sb, uint32_t *s_qtab,
                                          uint32_t lds_entries, uint32_t *s_dcode, uint8_t *s_qcode, uint8_t *s_amb,
                                          SqLut &Lt)
{
    const int tid = threadIdx.x;
    const uint32_t nq = sb.nqual, entries = (sb.max_q + 1) * nq;
    const uint32_t *qg = A.qcodes + (size_t)s * A.qrows * 256;
    if (entries <= lds_entries) {
        for (uint32_t k = tid; k < entries; k += TL_THREADS) s_qtab[k] = qg[(size_t)(k / nq) * 256 + (k % nq)];
        Lt.qtab = s_qtab;
        Lt.qstride = nq;
    } else {
        Lt.qtab = qg;
        Lt.qstride = 256;
    }
    s_dcode[tid] = A.dcodes[(size_t)s * 256 + tid];
    s_qcode[tid] = sb.qua_code[tid];
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    Lt.dcode = s_dcode;
    Lt.qcode = s_qcode;
    Lt.amb = s_amb;
}

// group-of-32 OR / add reductions (xor distances < 32 never leave the 32-lane half)
__device__ __forceinline__ uint32_t g32_or(uint32_t v)
{
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v |= __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ uint32_t g32_sum(uint32_t v)
{
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// Cached non-constant title fields of one record: numeric -> value, text -> tile offset | length << 16.
struct CachedTitle {
    uint32_t cur[NCF_CAP];
    uint32_t prev[NCF_CAP];
};

// One record's title bits from its cached fields (tasks.cpp:427-507; same order and rules as
// title_record_emit() in ngsc_common.h, which the untiled kernels and the CPU harness use).
template <class Sink>
__device__ __forceinline__ void title_emit_cached(const uint8_t *tb, const Arena &A, uint32_t s, const FieldDesc *fd,
                                                  const uint8_t *kf_field, uint32_t ncf, bool first,
                                                  const CachedTitle &C, uint32_t gconst, Sink &w)
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
                    const uint32_t e = A.ncodes[((size_t)s * MAX_FIELDS + f) * MAX_NUM_HUF + to_store];
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
                for (uint32_t j = 0; j < len; ++j) {
                    if (!ham_bit(d, j)) {
                        const uint32_t e = tc[(j < MAX_STAT_LEN ? j : MAX_STAT_LEN) * 256 + tb[start + j]];
                        w.put(e >> 5, e & 31);
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(TL_THREADS) void k_measure_tile(Arena A, uint32_t R, uint32_t tile_bytes, uint32_t lds_entries)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_dyn[];  // tile, then quality code table
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    __shared__ FieldDesc s_fd[MAX_FIELDS];
    __shared__ uint32_t s_w[NCF_CAP][TL_THREADS];
    __shared__ uint8_t s_kf_field[NCF_CAP];
    __shared__ uint32_t s_part[TL_WAVES];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo, nf = sb.nf, ncf = sb.ncf;
    const uint32_t t0 = blockIdx.x * R;
    if (t0 >= n) return;
    const uint32_t nt = t0 + R < n ? R : n - t0;
    const uint32_t i0 = rec_lo + t0;
    const uint32_t T = TL_THREADS / R;
    const int tid = threadIdx.x;
    uint32_t *s_qtab = reinterpret_cast<uint32_t *>(s_dyn + tile_bytes);

    for (uint32_t k = tid; k < nf * (sizeof(FieldDesc) / 4); k += TL_THREADS)
        reinterpret_cast<uint32_t *>(s_fd)[k] = reinterpret_cast<const uint32_t *>(A.fdesc + (size_t)s * MAX_FIELDS)[k];
    SqLut Lt;
    load_luts(A, s, sb, s_qtab, lds_entries, s_dcode, s_qcode, s_amb, Lt);
    const uint32_t b0 = load_tile(A, i0, i0 + nt, s_dyn);
    const uint8_t *tb = s_dyn;
    if ((uint32_t)tid < nf && s_fd[tid].kind != KIND_CONST && s_fd[tid].kf < NCF_CAP) s_kf_field[s_fd[tid].kf] = (uint8_t)tid;

    // ---- title (tasks.cpp:393-509) ----
    const bool valid = (uint32_t)tid < nt;
    const uint32_t gi = tid & 31;                  // index inside the 32-record group
    const uint32_t gstart = tid - gi;              // tile-relative first record of the group
    const uint32_t gsize = valid ? ((gstart + 32 <= nt) ? 32u : nt - gstart) : 0u;
    uint32_t bits = 0, gconst = 0;
    if (ncf <= NCF_CAP) {
        // one flat pass over the title: cache the non-constant fields of this record
        const uint32_t pc = valid ? A.ls[i0 + tid] - b0 : 0;
        const uint32_t tl = valid ? A.te[i0 + tid] - A.ls[i0 + tid] + 1 : 0;
        const uint32_t maxl = wmax(tl);
        uint32_t f = 0, k = 0, v = 0;
        for (uint32_t it = 0; it < maxl; ++it) {
            const bool act = it < tl;
            const uint32_t c = tb[act ? pc + it : pc];
            const bool sep = act && sep_fast(c);
            if (act && !sep) { v = v * 10 + (c - '0'); ++k; }   // wraps like utils::to_num (utils.h:117-125)
            if (sep) {
                if (f < nf && s_fd[f].kind != KIND_CONST)
                    s_w[s_fd[f].kf][tid] = s_fd[f].kind == KIND_NUM ? v : ((pc + it - k) | (k << 16));
                ++f; k = 0; v = 0;
            }
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
            A.tw[(size_t)kf * A.tw_stride + i0 + tid] = wc;
            if (gi) {
                const uint32_t wp = s_w[kf][tid - 1], wf = s_w[kf][gstart];
                C.prev[kf] = wp;
                if (s_fd[ff].kind == KIND_NUM) {
                    if (wc != wf) d.val |= 1u << ff;
                    if ((int32_t)(wc - wp) != s_fd[ff].min_d) d.dlt |= 1u << ff;
                } else {
                    const uint32_t lc = wc >> 16, sc = wc & 0xffffu, sf = wf & 0xffffu;
                    bool same = lc == (wf >> 16);
                    for (uint32_t j = 0; same && j < lc; ++j) same = tb[sc + j] == tb[sf + j];
                    if (!same) d.txt |= 1u << ff;
                }
            }
        }
        TitleDiffs gd = {g32_or(d.txt), g32_or(d.val), g32_or(d.dlt)};
        gconst = title_group_const(s_fd, nf, gd, gsize);
        if (valid) {
            BitCounter bc;
            if (gi == 0) title_put_flags(s_fd, nf, gconst, bc);
            title_emit_cached(tb, A, s, s_fd, s_kf_field, ncf, gi == 0, C, gconst, bc);
            bits = (uint32_t)bc.bits;
        }
    } else {
        // many non-constant fields: walk (first, previous, current) titles per record
        TitleDiffs d = {0, 0, 0};
        uint32_t p_cur = 0, p_prev = 0;
        if (valid) {
            p_cur = A.ls[i0 + tid] - b0;
            p_prev = gi ? A.ls[i0 + tid - 1] - b0 : p_cur;
            if (gi) d = title_record_diffs(tb, s_fd, nf, A.ls[i0 + gstart] - b0, p_prev, p_cur);
        }
        TitleDiffs gd = {g32_or(d.txt), g32_or(d.val), g32_or(d.dlt)};
        gconst = title_group_const(s_fd, nf, gd, gsize);
        if (valid) {
            BitCounter bc;
            if (gi == 0) title_put_flags(s_fd, nf, gconst, bc);
            title_record_emit(tb, A, s, s_fd, nf, gi == 0, p_prev, p_cur, gconst, bc);
            bits = (uint32_t)bc.bits;
        }
    }
    if (valid) A.tbits[i0 + tid] = bits;
    {
        uint32_t gbits = g32_sum(bits);
        if (valid && gi == 0) {
            uint32_t grp = A.grp_lo[s] + (t0 + gstart) / GROUP;
            A.goff[grp] = (gbits + 7) >> 3;
            A.gflags[grp] = gconst;
        }
    }
    // ---- quality / DNA: bits of this thread's slice ----
    const Slice S = make_slice(A, i0, nt, T, b0);
    uint32_t qb = 0, db = 0;
    {
        const bool xfer = S.valid ? (A.slen[i0 + S.r] >> 31) : false;
        const uint32_t len = S.j1 - S.j0;
        const uint32_t maxl = wmax(len);
        const bool wave_xfer = __any(xfer);
        for (uint32_t base = 0; base < maxl; base += 8) {
            uint32_t cc[8], qv[8], qc[8], qe[8], de[8];
            uint32_t dropm = 0;
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint32_t j = (base + u < len) ? S.j0 + base + u : S.j0;
                cc[u] = tb[S.s0 + j];
                qv[u] = tb[S.q0 + j];
            }
            if (wave_xfer) {
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) {
                    const uint32_t a = Lt.amb[cc[u]];
                    if (xfer && a > 1) { dropm |= 1u << u; qv[u] = transfer_quality(a, (uint8_t)qv[u]); }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) { qc[u] = Lt.qcode[qv[u]]; de[u] = Lt.dcode[cc[u]]; }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u)
                qe[u] = Lt.qtab[((base + u < len) ? S.j0 + base + u + 1 : 0) * Lt.qstride + qc[u]];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const bool act = base + u < len;
                qb += act ? (qe[u] & 31) : 0;
                db += (act && !((dropm >> u) & 1)) ? (de[u] & 31) : 0;
            }
        }
    }
    const size_t tix = (size_t)A.tile_lo[s] + blockIdx.x;
    A.ubits[tix * TL_THREADS + tid] = qb | (db << 16);
    uint32_t tq, td;
    (void)wg_scan_excl256(qb, s_part, &tq);
    (void)wg_scan_excl256(db, s_part, &td);
    if (tid == 0) { A.wq[tix] = tq; A.wd[tix] = td; }
}

// ====================================================================================================
// Scans: one workgroup (1024 threads) per sub-block
// ====================================================================================================
__device__ uint32_t wg_scan_excl_t(uint32_t *a, uint32_t n, uint32_t *s_wave /*[17]*/)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t run = 0;
    for (uint32_t base = 0; base < n; base += 1024) {
        uint32_t i = base + tid;
        uint32_t v = i < n ? a[i] : 0;
        uint32_t incl = wscan_incl(v);
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        if (tid < 64) {
            uint32_t w = tid < 16 ? s_wave[tid] : 0;
            uint32_t sc = wscan_incl(w);
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

__global__ __launch_bounds__(1024) void k_scan_tile(Arena A, uint32_t R)
{
    if (*A.err) return;
    __shared__ uint32_t s_wave[17];
    const uint32_t s = blockIdx.x;
    SbState &sb = A.sb[s];
    const uint32_t ntiles = (sb.n_rec + R - 1) / R;
    uint32_t tq = wg_scan_excl_t(A.wq + A.tile_lo[s], ntiles, s_wave);
    uint32_t td = wg_scan_excl_t(A.wd + A.tile_lo[s], ntiles, s_wave);
    uint32_t tg = wg_scan_excl_t(A.goff + A.grp_lo[s], (sb.n_rec + GROUP - 1) / GROUP, s_wave);
    if (threadIdx.x == 0) { sb.qbits = tq; sb.dbits = td; sb.title_body = tg; }
}

// ====================================================================================================
// Pack
// ====================================================================================================
// One lane's MSB-first bit writer into a word array (LDS staging, or the zeroed output in the overflow
// path).  Bits are collected in a 64-bit accumulator and leave as whole 32-bit big-endian words, each
// merged with atomicOr: a lane's first and last word are shared with its neighbours.
struct LaneWriter {
    uint32_t *words;
    uint64_t widx;   // index of the word currently being filled (relative to words[0])
    uint64_t acc;
    uint32_t nacc;   // bits pending in acc, including the leading (bitpos & 31) zero bits
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

// Copies staged words [0, nwords) to the output starting at absolute word `word0`.  Words strictly inside
// the stream's bit range [b0, b1) are complete and stored plainly; the boundary words are shared -> atomicOr.
__device__ __forceinline__ void copy_out_words(uint32_t *outw, const uint32_t *stage, uint64_t word0, uint32_t nwords,
                                               uint64_t b0, uint64_t b1)
{
    for (uint32_t k = threadIdx.x; k < nwords; k += blockDim.x) {
        uint64_t w = word0 + k;
        uint32_t v = stage[k];
        if ((w << 5) >= b0 && ((w + 1) << 5) <= b1) outw[w] = v;
        else if (v) atomicOr(&outw[w], v);
    }
}

__global__ __launch_bounds__(TL_THREADS) void k_pack_tile(Arena A, uint32_t R, uint32_t tile_bytes, uint32_t lds_entries,
                                                          uint32_t sw_t, uint32_t sw_q, uint32_t sw_d)
{
    if (*A.err) return;
    // dynamic LDS: tile | quality code table | staging areas (title sw_t, quality sw_q, DNA sw_d words)
    extern __shared__ __align__(16) uint8_t s_dyn[];
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    __shared__ FieldDesc s_fd[MAX_FIELDS];
    __shared__ uint8_t s_kf_field[NCF_CAP];
    __shared__ uint32_t s_part[TL_WAVES];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo, nf = sb.nf, ncf = sb.ncf;
    const uint32_t t0 = blockIdx.x * R;
    if (t0 >= n) return;
    const uint32_t nt = t0 + R < n ? R : n - t0;
    const uint32_t i0 = rec_lo + t0;
    const uint32_t T = TL_THREADS / R;
    const bool last_tile = t0 + R >= n;
    const int tid = threadIdx.x;
    uint32_t *s_qtab = reinterpret_cast<uint32_t *>(s_dyn + tile_bytes);
    uint32_t *s_stage = s_qtab + lds_entries;
    uint32_t *st_t = s_stage, *st_q = s_stage + sw_t, *st_d = s_stage + sw_t + sw_q;

    for (uint32_t k = tid; k < sw_t + sw_q + sw_d; k += TL_THREADS) s_stage[k] = 0;
    for (uint32_t k = tid; k < nf * (sizeof(FieldDesc) / 4); k += TL_THREADS)
        reinterpret_cast<uint32_t *>(s_fd)[k] = reinterpret_cast<const uint32_t *>(A.fdesc + (size_t)s * MAX_FIELDS)[k];
    SqLut Lt;
    load_luts(A, s, sb, s_qtab, lds_entries, s_dcode, s_qcode, s_amb, Lt);
    const uint32_t b0 = load_tile(A, i0, i0 + nt, s_dyn);
    const uint8_t *tb = s_dyn;
    if ((uint32_t)tid < nf && s_fd[tid].kind != KIND_CONST && s_fd[tid].kf < NCF_CAP) s_kf_field[s_fd[tid].kf] = (uint8_t)tid;
    __syncthreads();
    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out);

    // absolute bit ranges of this tile's share of the three payload streams
    const uint32_t grp_first = A.grp_lo[s] + t0 / GROUP;
    const uint32_t ngrp_t = (nt + GROUP - 1) / GROUP;
    const uint64_t title_base = (sb.out_off + sb.info_len + sb.title_hdr) * 8;
    const uint64_t tb0 = title_base + 8ull * A.goff[grp_first];
    const uint64_t tb1 = title_base + 8ull * (last_tile ? sb.title_body : A.goff[grp_first + ngrp_t]);
    const size_t tix = (size_t)A.tile_lo[s] + blockIdx.x;
    const uint64_t qbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_hdr) * 8;
    const uint64_t dbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_len + sb.dna_hdr) * 8;
    const uint64_t qb0 = qbase + A.wq[tix], qb1 = qbase + (last_tile ? sb.qbits : A.wq[tix + 1]);
    const uint64_t db0 = dbase + A.wd[tix], db1 = dbase + (last_tile ? sb.dbits : A.wd[tix + 1]);
    const uint64_t tw0 = tb0 >> 5, qw0 = qb0 >> 5, dw0 = db0 >> 5;
    const uint32_t tnw = (uint32_t)(((tb1 + 31) >> 5) - tw0), qnw = (uint32_t)(((qb1 + 31) >> 5) - qw0),
                   dnw = (uint32_t)(((db1 + 31) >> 5) - dw0);
    // staging overflow (very long titles / codes): write straight into the zeroed output with atomics
    const bool t_lds = tnw + 1 <= sw_t, q_lds = qnw + 1 <= sw_q, d_lds = dnw + 1 <= sw_d;

    // ---- title: every lane ORs its own record's bits ----
    {
        const bool valid = (uint32_t)tid < nt;
        const uint32_t gi = tid & 31, gstart = tid - gi;
        const uint32_t bits = valid ? A.tbits[i0 + tid] : 0;
        uint32_t incl = bits;  // inclusive prefix inside the 32-record group
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            uint32_t o = __shfl_up(incl, d);
            if ((int)gi >= d) incl += o;
        }
        if (valid) {
            const uint32_t grp = A.grp_lo[s] + (t0 + gstart) / GROUP;
            const uint32_t gconst = A.gflags[grp];
            LaneWriter w;
            w.init(t_lds ? st_t : outw, t_lds ? tw0 : 0, title_base + 8ull * A.goff[grp] + (incl - bits));
            if (gi == 0) title_put_flags(s_fd, nf, gconst, w);
            if (ncf <= NCF_CAP) {
                CachedTitle C;
#pragma unroll
                for (uint32_t kf = 0; kf < NCF_CAP; ++kf) {
                    C.cur[kf] = C.prev[kf] = 0;
                    if (kf >= ncf) continue;
                    C.cur[kf] = A.tw[(size_t)kf * A.tw_stride + i0 + tid];
                    if (gi) C.prev[kf] = A.tw[(size_t)kf * A.tw_stride + i0 + tid - 1];
                }
                title_emit_cached(tb, A, s, s_fd, s_kf_field, ncf, gi == 0, C, gconst, w);
            } else {
                const uint32_t p_cur = A.ls[i0 + tid] - b0, p_prev = gi ? A.ls[i0 + tid - 1] - b0 : p_cur;
                title_record_emit(tb, A, s, s_fd, nf, gi == 0, p_prev, p_cur, gconst, w);
            }
            w.finish();
        }
    }

    // ---- quality / DNA: one lane per record slice, offsets from a workgroup scan of the slice bit counts ----
    {
        const Slice S = make_slice(A, i0, nt, T, b0);
        const uint32_t ub = A.ubits[tix * TL_THREADS + tid];
        uint32_t tq, td;
        const uint32_t qo = wg_scan_excl256(ub & 0xffffu, s_part, &tq);
        const uint32_t dofs = wg_scan_excl256(ub >> 16, s_part, &td);
        const bool xfer = S.valid ? (A.slen[i0 + S.r] >> 31) : false;
        const uint32_t len = S.j1 - S.j0;
        const uint32_t maxl = wmax(len);
        LaneWriter wq_, wd_;
        wq_.init(q_lds ? st_q : outw, q_lds ? qw0 : 0, qb0 + qo);
        wd_.init(d_lds ? st_d : outw, d_lds ? dw0 : 0, db0 + dofs);
        const bool wave_xfer = __any(xfer);
        for (uint32_t base = 0; base < maxl; base += 8) {
            uint32_t cc[8], qv[8], qc[8], qe[8], de[8];
            uint32_t dropm = 0;
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const uint32_t j = (base + u < len) ? S.j0 + base + u : S.j0;
                cc[u] = tb[S.s0 + j];
                qv[u] = tb[S.q0 + j];
            }
            if (wave_xfer) {
#pragma unroll
                for (uint32_t u = 0; u < 8; ++u) {
                    const uint32_t a = Lt.amb[cc[u]];
                    if (xfer && a > 1) { dropm |= 1u << u; qv[u] = transfer_quality(a, (uint8_t)qv[u]); }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) { qc[u] = Lt.qcode[qv[u]]; de[u] = Lt.dcode[cc[u]]; }   // tasks.cpp:549-556
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u)                                                          // tasks.cpp:614-618
                qe[u] = Lt.qtab[((base + u < len) ? S.j0 + base + u + 1 : 0) * Lt.qstride + qc[u]];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
                const bool act = base + u < len;
                wq_.put(qe[u] >> 5, act ? (qe[u] & 31) : 0);
                wd_.put(de[u] >> 5, (act && !((dropm >> u) & 1)) ? (de[u] & 31) : 0);
            }
        }
        if (len) { wq_.finish(); wd_.finish(); }
    }
    __syncthreads();
    if (t_lds) copy_out_words(outw, st_t, tw0, tnw, tb0, tb1);
    if (q_lds) copy_out_words(outw, st_q, qw0, qnw, qb0, qb1);
    if (d_lds) copy_out_words(outw, st_d, dw0, dnw, db0, db1);
}

// ---- launchers ---------------------------------------------------------------------------------------
constexpr uint32_t TILE_LDS_BUDGET = 48u << 10;   // bytes of raw records per tile

// Dynamic LDS above 64 KB needs an explicit opt-in per kernel (160 KB per CU on gfx950).
template <class K>
static void allow_big_lds(K kernel)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
}

// Records per tile for a job whose longest record has max_rec_bytes; 0 = use the untiled kernels.
uint32_t tile_records_for(uint32_t max_rec_bytes, uint32_t r_max)
{
    for (uint32_t R : {256u, 128u, 64u})
        if (R <= r_max && (uint64_t)R * max_rec_bytes + 32 <= TILE_LDS_BUDGET) return R;
    return 0;
}

static uint32_t tile_bytes_for(uint32_t R, uint32_t max_rec_bytes) { return ((R * max_rec_bytes + 32 + 15) & ~15u); }

static uint32_t qtab_entries_for(uint32_t qrows)
{
    uint64_t e = (uint64_t)qrows * 64;
    return (uint32_t)(e < 12288 ? e : 12288);  // <= 48 KB
}

void launch_title_stats_tile(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t R, uint32_t max_rec_bytes)
{
    dim3 grid((max_rec + R - 2) / (R - 1), A.n_sb);
    size_t lds = tile_bytes_for(R, max_rec_bytes);
    allow_big_lds(k_title_stats_tile);
    static const uint32_t dbg = getenv("NGSC_DBG_TS") ? (uint32_t)atoi(getenv("NGSC_DBG_TS")) : 0;
    hipLaunchKernelGGL(k_title_stats_tile, grid, dim3(R), lds, st, A, dbg);
}

void launch_seq_qual_stats_tile(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t R, uint32_t max_rec_bytes)
{
    uint32_t tbytes = tile_bytes_for(R, max_rec_bytes);
    uint32_t rows = A.qrows;
    uint32_t max_rows = (96u << 10) > tbytes ? ((96u << 10) - tbytes) / (HSTRIDE * 4) : 0;
    if (rows > max_rows) rows = max_rows;
    dim3 grid((max_rec + R - 1) / R, A.n_sb);
    size_t lds = tbytes + (size_t)rows * HSTRIDE * 4;
    allow_big_lds(k_seq_qual_stats_tile);
    hipLaunchKernelGGL(k_seq_qual_stats_tile, grid, dim3(TL_THREADS), lds, st, A, R, tbytes, rows);
}

void launch_measure_tile(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t R, uint32_t max_rec_bytes)
{
    uint32_t tbytes = tile_bytes_for(R, max_rec_bytes), le = qtab_entries_for(A.qrows);
    dim3 grid((max_rec + R - 1) / R, A.n_sb);
    allow_big_lds(k_measure_tile);
    hipLaunchKernelGGL(k_measure_tile, grid, dim3(TL_THREADS), tbytes + (size_t)le * 4, st, A, R, tbytes, le);
}

void launch_scan_tile(hipStream_t st, const Arena &A, uint32_t R)
{
    hipLaunchKernelGGL(k_scan_tile, dim3(A.n_sb), dim3(1024), 0, st, A, R);
}

void launch_pack_tile(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t R, uint32_t max_rec_bytes)
{
    uint32_t tbytes = tile_bytes_for(R, max_rec_bytes), le = qtab_entries_for(A.qrows);
    // staging sized for typical streams (5 bits per quality symbol, 3 per base, 6 B of title per record);
    // a tile that needs more writes straight to the zeroed output with atomics
    uint32_t maxq = A.qrows - 1;
    uint32_t sw_q = (uint32_t)(((uint64_t)R * maxq * 5) / 32) + 64;
    uint32_t sw_d = (uint32_t)(((uint64_t)R * maxq * 3) / 32) + 64;
    uint32_t sw_t = (R * 6) / 4 + 64;
    if (sw_q > 4096) sw_q = 4096;
    if (sw_d > 3072) sw_d = 3072;
    dim3 grid((max_rec + R - 1) / R, A.n_sb);
    size_t lds = tbytes + (size_t)le * 4 + ((size_t)sw_t + sw_q + sw_d) * 4;
    allow_big_lds(k_pack_tile);
    hipLaunchKernelGGL(k_pack_tile, grid, dim3(TL_THREADS), lds, st, A, R, tbytes, le, sw_t, sw_q, sw_d);
}

}  // namespace ngsc
