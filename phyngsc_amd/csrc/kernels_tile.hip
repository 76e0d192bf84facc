// kernels_tile.hip - LDS-tiled per-record kernels (the fast path).
//
// A tile is a run of R consecutive records of one sub-block (R = 256, 128 or 64: a multiple of 32, so the
// 32-record title groups never straddle tiles) whose bytes [ls[i0], ls[i1]) are staged into LDS with
// coalesced 16-byte loads.  Every stage then works out of LDS with one thread per record (titles) or one
// thread per record slice (sequence / quality; T = 256 / R threads share a record), so HBM sees each raw
// byte once per kernel and all random access, histogramming, code lookup and bit merging stays on chip:
//
//   k_seq_qual_stats_tile  N-transfer test, DNA counts, per-position quality histogram   (phyNGSC.cpp:462-653,
//                                                                                         tasks.cpp:260-286)
//   k_measure_sq_tile      quality / DNA code bits per record slice and per tile
//   k_scan_tile            per-sub-block exclusive scans of group bytes and tile bit totals
//   k_pack_sq_tile         StoreQuality and StoreDNA payloads: every lane accumulates its own bits and ORs whole
//                          words into LDS staging; one coalesced copy-out per stream (tasks.cpp:544-557, 609-619)
// (The title path lives in kernels_title.hip: it parses out of registers and needs no tile.)
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
                                                                    uint32_t lds_rows, uint32_t tpw)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_dyn[];  // [tile_bytes] tile, then [lds_rows][HSTRIDE] histogram
    __shared__ uint32_t s_occ[256];
    __shared__ uint8_t s_amb[256];
    __shared__ uint32_t s_red[3];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + R - 1) / R;
    const uint32_t tile_first = blockIdx.x * tpw;   // this workgroup walks tiles [tile_first, tile_last)
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;
    const uint32_t T = TL_THREADS / R;
    const int tid = threadIdx.x, lane = tid & 63;
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_dyn + tile_bytes);

    for (uint32_t k = tid; k < lds_rows * HSTRIDE; k += TL_THREADS) s_hist[k] = 0;
    s_occ[tid] = 0;
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (tid < 3) s_red[tid] = 0;
    uint32_t cA = 0, cC = 0, cG = 0, cT = 0;
    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    for (uint32_t tile = tile_first; tile < tile_last; ++tile) {
    const uint32_t t0 = tile * R;
    const uint32_t nt = t0 + R < n ? R : n - t0;
    const uint32_t i0 = rec_lo + t0;
    __syncthreads();   // everyone is done with the previous tile
    const uint32_t b0 = load_tile(A, i0, i0 + nt, s_dyn);
    const uint8_t *tb = s_dyn;
    const Slice S = make_slice(A, i0, nt, T, b0);

    // line 3 of every record must be a bare '+' (README.md:97-103); the reference assumes it
    if ((uint32_t)tid < nt) {
        uint32_t se = A.se[i0 + tid] - b0;
        if (tb[se + 1] != '+' || tb[se + 2] != '\n') atomicMin(A.err, (int32_t)ERR_FORMAT);
    }
    if (t0 == 0 && tid == 0 && S.L >= 1 && !A.cs_mode) {  // colour-space detection on the sub-block's first record, phyNGSC.cpp:478
        if (cs_first_read(tb[S.s0], tb[S.s0 + 1])) atomicMin(A.err, (int32_t)ERR_COLORSPACE);
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
    }  // tiles
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

__global__ __launch_bounds__(TL_THREADS) void k_measure_sq_tile(Arena A, uint32_t R, uint32_t tile_bytes, uint32_t lds_entries,
                                                                uint32_t tpw)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_dyn[];  // tile, then quality code table
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    __shared__ uint32_t s_part[TL_WAVES];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + R - 1) / R;
    const uint32_t tile_first = blockIdx.x * tpw;   // this workgroup walks tiles [tile_first, tile_last)
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;
    const uint32_t T = TL_THREADS / R;
    const int tid = threadIdx.x;
    uint32_t *s_qtab = reinterpret_cast<uint32_t *>(s_dyn + tile_bytes);

    SqLut Lt;
    load_luts(A, s, sb, s_qtab, lds_entries, s_dcode, s_qcode, s_amb, Lt);
    for (uint32_t tile = tile_first; tile < tile_last; ++tile) {
    const uint32_t t0 = tile * R;
    const uint32_t nt = t0 + R < n ? R : n - t0;
    const uint32_t i0 = rec_lo + t0;
    __syncthreads();   // everyone is done with the previous tile
    const uint32_t b0 = load_tile(A, i0, i0 + nt, s_dyn);
    const uint8_t *tb = s_dyn;

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
    const size_t tix = (size_t)A.tile_lo[s] + tile;
    A.ubits[tix * TL_THREADS + tid] = qb | (db << 16);
    uint32_t tq, td;
    (void)wg_scan_excl256(qb, s_part, &tq);
    (void)wg_scan_excl256(db, s_part, &td);
    if (tid == 0) { A.wq[tix] = tq; A.wd[tix] = td; }
    }  // tiles
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

__global__ __launch_bounds__(TL_THREADS) void k_pack_sq_tile(Arena A, uint32_t R, uint32_t tile_bytes, uint32_t lds_entries,
                                                             uint32_t sw_q, uint32_t sw_d, uint32_t tpw)
{
    if (*A.err) return;
    // dynamic LDS: tile | quality code table | staging areas (quality sw_q, DNA sw_d words)
    extern __shared__ __align__(16) uint8_t s_dyn[];
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    __shared__ uint32_t s_part[TL_WAVES];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + R - 1) / R;
    const uint32_t tile_first = blockIdx.x * tpw;   // this workgroup walks tiles [tile_first, tile_last)
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;
    const uint32_t T = TL_THREADS / R;
    const int tid = threadIdx.x;
    uint32_t *s_qtab = reinterpret_cast<uint32_t *>(s_dyn + tile_bytes);
    uint32_t *s_stage = s_qtab + lds_entries;
    uint32_t *st_q = s_stage, *st_d = s_stage + sw_q;

    SqLut Lt;
    load_luts(A, s, sb, s_qtab, lds_entries, s_dcode, s_qcode, s_amb, Lt);
    for (uint32_t tile = tile_first; tile < tile_last; ++tile) {
    const uint32_t t0 = tile * R;
    const uint32_t nt = t0 + R < n ? R : n - t0;
    const uint32_t i0 = rec_lo + t0;
    __syncthreads();   // everyone is done with the previous tile
    const bool last_tile = t0 + R >= n;
    for (uint32_t k = tid; k < sw_q + sw_d; k += TL_THREADS) s_stage[k] = 0;
    const uint32_t b0 = load_tile(A, i0, i0 + nt, s_dyn);
    const uint8_t *tb = s_dyn;
    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out);

    // absolute bit ranges of this tile's share of the two payload streams
    const size_t tix = (size_t)A.tile_lo[s] + tile;
    const uint64_t qbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_hdr) * 8;
    const uint64_t dbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_len + sb.dna_hdr) * 8;
    const uint64_t qb0 = qbase + A.wq[tix], qb1 = qbase + (last_tile ? sb.qbits : A.wq[tix + 1]);
    const uint64_t db0 = dbase + A.wd[tix], db1 = dbase + (last_tile ? sb.dbits : A.wd[tix + 1]);
    const uint64_t qw0 = qb0 >> 5, dw0 = db0 >> 5;
    const uint32_t qnw = (uint32_t)(((qb1 + 31) >> 5) - qw0), dnw = (uint32_t)(((db1 + 31) >> 5) - dw0);
    // staging overflow (very long codes): write straight into the zeroed output with atomics
    const bool q_lds = qnw + 1 <= sw_q, d_lds = dnw + 1 <= sw_d;

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
    if (q_lds) copy_out_words(outw, st_q, qw0, qnw, qb0, qb1);
    if (d_lds) copy_out_words(outw, st_d, dw0, dnw, db0, db1);
    }  // tiles
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

// Tiles walked by one workgroup (amortises the code table loads and the histogram flush).
static uint32_t tiles_per_wg()
{
    static const uint32_t v = getenv("NGSC_TILES_PER_WG") ? (uint32_t)atoi(getenv("NGSC_TILES_PER_WG")) : 4;
    return v ? v : 1;
}

static uint32_t tile_bytes_for(uint32_t R, uint32_t max_rec_bytes) { return ((R * max_rec_bytes + 32 + 15) & ~15u); }

static uint32_t qtab_entries_for(uint32_t qrows)
{
    uint64_t e = (uint64_t)qrows * 64;
    return (uint32_t)(e < 12288 ? e : 12288);  // <= 48 KB
}

void launch_seq_qual_stats_tile(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t R, uint32_t max_rec_bytes)
{
    uint32_t tbytes = tile_bytes_for(R, max_rec_bytes);
    uint32_t rows = A.qrows;
    uint32_t max_rows = (96u << 10) > tbytes ? ((96u << 10) - tbytes) / (HSTRIDE * 4) : 0;
    if (rows > max_rows) rows = max_rows;
    const uint32_t tpw = tiles_per_wg();
    dim3 grid(((max_rec + R - 1) / R + tpw - 1) / tpw, A.n_sb);
    size_t lds = tbytes + (size_t)rows * HSTRIDE * 4;
    allow_big_lds(k_seq_qual_stats_tile);
    hipLaunchKernelGGL(k_seq_qual_stats_tile, grid, dim3(TL_THREADS), lds, st, A, R, tbytes, rows, tpw);
}

void launch_measure_sq_tile(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t R, uint32_t max_rec_bytes)
{
    uint32_t tbytes = tile_bytes_for(R, max_rec_bytes), le = qtab_entries_for(A.qrows);
    const uint32_t tpw = tiles_per_wg();
    dim3 grid(((max_rec + R - 1) / R + tpw - 1) / tpw, A.n_sb);
    allow_big_lds(k_measure_sq_tile);
    hipLaunchKernelGGL(k_measure_sq_tile, grid, dim3(TL_THREADS), tbytes + (size_t)le * 4, st, A, R, tbytes, le, tpw);
}

void launch_scan_tile(hipStream_t st, const Arena &A, uint32_t R)
{
    hipLaunchKernelGGL(k_scan_tile, dim3(A.n_sb), dim3(1024), 0, st, A, R);
}

void launch_pack_sq_tile(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t R, uint32_t max_rec_bytes)
{
    uint32_t tbytes = tile_bytes_for(R, max_rec_bytes), le = qtab_entries_for(A.qrows);
    // staging sized for typical streams (5 bits per quality symbol, 3 per base); a tile that needs more
    // writes straight to the zeroed output with atomics
    uint32_t maxq = A.qrows - 1;
    uint32_t sw_q = (uint32_t)(((uint64_t)R * maxq * 5) / 32) + 64;
    uint32_t sw_d = (uint32_t)(((uint64_t)R * maxq * 3) / 32) + 64;
    if (sw_q > 4096) sw_q = 4096;
    if (sw_d > 3072) sw_d = 3072;
    const uint32_t tpw = tiles_per_wg();
    dim3 grid(((max_rec + R - 1) / R + tpw - 1) / tpw, A.n_sb);
    size_t lds = tbytes + (size_t)le * 4 + ((size_t)sw_q + sw_d) * 4;
    allow_big_lds(k_pack_sq_tile);
    hipLaunchKernelGGL(k_pack_sq_tile, grid, dim3(TL_THREADS), lds, st, A, R, tbytes, le, sw_q, sw_d, tpw);
}

}  // namespace ngsc
