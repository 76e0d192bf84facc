// kernels_tile.hip - LDS-tiled versions of the per-record kernels (the fast path).
//
// A tile is a run of R consecutive records of one sub-block (R = 64..256, a multiple of 32 so that the
// 32-record title groups never straddle tiles) whose bytes [ls[i0], ls[i1]) are staged into LDS with
// coalesced 16-byte loads.  Every record-oriented stage then reads bytes from LDS instead of walking
// HBM/L2 with per-lane byte loads:
//
//   k_title_stats_tile     AnalyzeTitleFields reductions, thread per record            (tasks.cpp:22-193)
//   k_seq_qual_stats_tile  N-transfer + DNA counts + per-position quality histogram,    (phyNGSC.cpp:462-653,
//                          lane per symbol over the tile's concatenated bases            tasks.cpp:260-286)
//   k_measure_tile         title group flags + per-record title bits; quality / DNA bit totals per wave chunk
//   k_scan_tile            per-sub-block exclusive scans of group bytes and wave-chunk bit totals
//   k_pack_tile            StoreTitle body, StoreQuality and StoreDNA payloads into LDS staging, then one
//                          coalesced copy-out per stream                               (tasks.cpp:427-507,
//                                                                                       544-557, 609-619)
// HBM traffic per launch = the raw bytes once (+ the stream bytes written by k_pack_tile); all shuffles,
// histograms, code lookups and bit merging happen in LDS / registers.  Integer work, no MFMA.
#include "ngsc_kernels.h"

namespace ngsc {

constexpr int TL_THREADS = 256;
constexpr int TL_WAVES = TL_THREADS / 64;

__device__ __forceinline__ uint32_t amb_fast_t(uint8_t c)
{
    return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? 1u : amb_code(c);
}

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

// Stages bytes [ls[i0] & ~15, ls[i1]) of the job buffer into s_tile; returns the rebased pointer p with
// p[pos] == buf[pos] for every pos inside the tile.  Ends with a barrier.
__device__ __forceinline__ const uint8_t *load_tile(const Arena &A, uint32_t i0, uint32_t i1, uint8_t *s_tile)
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
    return s_tile - b0;
}

__device__ __forceinline__ void lds_max_t(uint32_t *p, uint32_t v) { if (v > *p) atomicMax(p, v); }
__device__ __forceinline__ void lds_or_t(uint32_t *p, uint32_t v) { if ((*p & v) != v) atomicOr(p, v); }

// ====================================================================================================
// Title statistics (the reductions of k_title_stats, bytes from LDS).  blockDim.x = R threads; a tile is
// R-1 owned records plus the previous record as delta halo (thread 0).  All lanes walk the fields in
// lockstep, so the previous record's value comes from the neighbouring lane (__shfl_up); the first lane of
// each wave gets it from a small LDS edge buffer.
// ====================================================================================================
__global__ __launch_bounds__(TL_THREADS) void k_title_stats_tile(Arena A)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_dyn[];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t owned_per_tile = blockDim.x - 1;
    if (blockIdx.x * owned_per_tile >= n) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    __shared__ uint32_t s_stat[MAX_FIELDS * 24];
    __shared__ uint32_t s_edge_v[MAX_FIELDS][TL_WAVES];   // value of each wave's last lane
    __shared__ uint32_t s_own_v[MAX_FIELDS][TL_WAVES];    // value of each wave's first lane
    __shared__ uint32_t s_edge_m[TL_WAVES], s_own_m[TL_WAVES];
    __shared__ uint32_t s_nf0;

    const int64_t ifirst = (int64_t)blockIdx.x * owned_per_tile - 1;
    const uint32_t r0 = ifirst < 0 ? 0u : (uint32_t)ifirst;
    const uint64_t rend = ((uint64_t)blockIdx.x + 1) * owned_per_tile;
    const uint32_t r1 = rend < n ? (uint32_t)rend : n;  // exclusive
    const uint8_t *tb = load_tile(A, rec_lo + r0, rec_lo + r1, s_dyn);
    const uint8_t *gbuf = A.buf;

    for (int k = tid; k < (int)(MAX_FIELDS * 24); k += blockDim.x) s_stat[k] = 0;
    if (tid == 0) {
        uint32_t nf0 = count_fields(gbuf, A.ls[rec_lo], A.te[rec_lo]);
        s_nf0 = nf0;
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
    uint32_t pc = valid ? A.ls[g] : 0, p0 = A.ls[rec_lo];
    const uint32_t tec = valid ? A.te[g] : 0;
    uint32_t nummask = 0;
    bool short_title = false;
    for (uint32_t f = 0; f < nf0; ++f) {
        uint32_t v = 0;
        bool isnum = false;
        const uint32_t e0 = field_end(gbuf, p0);  // record 0 of the sub-block: same addresses in every lane
        if (valid && pc <= tec) {
            const uint32_t ec = field_end(tb, pc);
            const uint32_t len = ec - pc, len0 = e0 - p0;
            bool ovf = false;
            isnum = parse_num(tb, pc, ec, v, ovf);
            if (owned) {
                uint32_t *st = s_stat + f * 24;
                lds_max_t(st + 0, len);
                lds_max_t(st + 1, ~len);
                uint32_t m = len < len0 ? len : len0;
                bool diff = false;
                for (uint32_t p = 0; p < m; ++p) {
                    if (tb[pc + p] != gbuf[p0 + p]) {
                        diff = true;
                        if (p < 256) lds_or_t(st + 10 + (p >> 5), 1u << (p & 31));
                    }
                }
                if (len != len0) { lds_max_t(st + 3, 1); lds_max_t(st + 2, 1); }
                else if (diff) lds_max_t(st + 2, 1);
                if (!isnum) lds_max_t(st + 4, 1);
                else {
                    if (ovf) lds_max_t(st + 5, 1);
                    lds_max_t(st + 6, bias((int32_t)v));
                    lds_max_t(st + 7, ~bias((int32_t)v));
                }
            }
            pc = ec + 1;
        } else if (valid) {
            short_title = true;
        }
        nummask |= (uint32_t)isnum << f;
        // delta against the previous record (tasks.cpp:149-166): neighbour lane, or LDS edge across waves
        const uint32_t pv = __shfl_up(v, 1);
        const uint32_t pnum = __shfl_up((uint32_t)isnum, 1);
        if (lane == 63) s_edge_v[f][wave] = v;
        if (lane == 0) s_own_v[f][wave] = v;
        if (lane > 0 && owned && i >= 1 && isnum && pnum) {
            int32_t d = (int32_t)(v - pv);
            lds_max_t(s_stat + f * 24 + 8, bias(d));
            lds_max_t(s_stat + f * 24 + 9, ~bias(d));
        }
        p0 = e0 + 1;
    }
    if (valid && (short_title || pc != tec + 1)) atomicMin(A.err, (int32_t)ERR_FIELDS);  // phyNGSC.cpp:417-421 only warns
    if (lane == 63) s_edge_m[wave] = nummask;
    if (lane == 0) s_own_m[wave] = nummask;
    __syncthreads();
    if (lane == 0 && wave > 0 && owned && i >= 1) {
        uint32_t both = s_own_m[wave] & s_edge_m[wave - 1];
        for (uint32_t f = 0; f < nf0; ++f) {
            if ((both >> f) & 1) {
                int32_t d = (int32_t)(s_own_v[f][wave] - s_edge_v[f][wave - 1]);
                lds_max_t(s_stat + f * 24 + 8, bias(d));
                lds_max_t(s_stat + f * 24 + 9, ~bias(d));
            }
        }
    }
    __syncthreads();
    uint32_t *gst = reinterpret_cast<uint32_t *>(A.fstat + (size_t)s * MAX_FIELDS);
    for (uint32_t k = tid; k < nf0 * 24; k += blockDim.x) {
        uint32_t v = s_stat[k];
        if (!v) continue;
        uint32_t w = k % 24;
        if (w >= 10 && w < 18) atomicOr(gst + k, v);
        else if (w < 10) atomicMax(gst + k, v);
    }
}

// ====================================================================================================
// Per-tile record bookkeeping shared by the symbol-parallel kernels
// ====================================================================================================
struct TileRecs {
    uint32_t te[TL_THREADS];      // title end   (offset in the job buffer)
    uint32_t se[TL_THREADS];      // sequence end
    uint32_t cum[TL_THREADS + 1]; // exclusive prefix of read lengths inside the tile
    uint32_t flag[TL_THREADS];    // per record scratch (transfer flags)
    uint32_t wpart[TL_WAVES + 1];
};

// Fills te/se/cum for records [i0, i0+nt); all threads must call.  Ends with a barrier.
__device__ __forceinline__ void tile_records(const Arena &A, uint32_t i0, uint32_t nt, TileRecs &T)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t L = 0;
    if ((uint32_t)tid < nt) {
        uint32_t t = A.te[i0 + tid], e = A.se[i0 + tid];
        T.te[tid] = t;
        T.se[tid] = e;
        L = e - t - 1;
    }
    T.flag[tid] = 0;
    uint32_t incl = wscan_incl(L);
    if (lane == 63) T.wpart[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += T.wpart[w];
    T.cum[tid] = base + incl - L;
    if (tid == TL_THREADS - 1) T.cum[TL_THREADS] = base + incl;
    __syncthreads();
}

// Symbol cursor: maps the wave's current 64 symbols to (record, position).  `rs` is a wave-uniform hint.
__device__ __forceinline__ bool locate(const TileRecs &T, uint32_t nt, uint32_t e, uint32_t total, uint32_t &rs,
                                       uint32_t &r, uint32_t &j)
{
    bool act = e < total;
    r = rs;
    if (act) {
        while (r + 1 < nt && e >= T.cum[r + 1]) ++r;
        j = e - T.cum[r];
    } else {
        j = 0;
    }
    // next step starts at or after the record of the last active lane
    uint32_t rl = act ? r : 0;
    rs = wmax(rl);
    return act;
}

__device__ __forceinline__ uint32_t first_record_of(const TileRecs &T, uint32_t nt, uint32_t e)
{
    uint32_t lo = 0, hi = nt;  // largest r with cum[r] <= e
    while (lo + 1 < hi) {
        uint32_t mid = (lo + hi) >> 1;
        if (T.cum[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// ====================================================================================================
// DNA / quality statistics, lane per symbol
// ====================================================================================================
constexpr uint32_t HCOLS = 64, HSTRIDE = 65;

__global__ __launch_bounds__(TL_THREADS) void k_seq_qual_stats_tile(Arena A, uint32_t R, uint32_t tile_bytes,
                                                                    uint32_t lds_rows)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_dyn[];  // [tile_bytes] tile, then [lds_rows][HSTRIDE] histogram
    __shared__ TileRecs T;
    __shared__ uint32_t s_occ[256];
    __shared__ uint32_t s_red[3];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t t0 = blockIdx.x * R;
    if (t0 >= n) return;
    const uint32_t nt = t0 + R < n ? R : n - t0;
    const uint32_t i0 = rec_lo + t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_dyn + tile_bytes);

    for (uint32_t k = tid; k < lds_rows * HSTRIDE; k += TL_THREADS) s_hist[k] = 0;
    s_occ[tid] = 0;
    if (tid < 3) s_red[tid] = 0;
    const uint8_t *tb = load_tile(A, i0, i0 + nt, s_dyn);
    tile_records(A, i0, nt, T);
    const uint32_t total = T.cum[TL_THREADS];
    const uint32_t chunk = ((total + TL_WAVES * 64 - 1) / (TL_WAVES * 64)) * 64;  // symbols per wave, multiple of 64
    const uint32_t c0 = wave * chunk, c1 = c0 + chunk < total ? c0 + chunk : total;

    if (t0 == 0 && tid == 0) {  // colour-space detection on the sub-block's first record, phyNGSC.cpp:478
        uint32_t L = T.se[0] - T.te[0] - 1;
        uint8_t c1b = tb[T.te[0] + 2];
        if (L >= 2 && c1b >= '0' && c1b <= '3') atomicMin(A.err, (int32_t)ERR_COLORSPACE);
    }
    // pass 1: N-transfer test (phyNGSC.cpp:549-571): flag = any | bad << 1 | (#ambiguous) << 8
    if (c0 < total) {
        uint32_t rs = first_record_of(T, nt, c0);
        for (uint32_t e0 = c0; e0 < c1; e0 += 64) {
            uint32_t r, j;
            bool act = locate(T, nt, e0 + lane, c1, rs, r, j);
            if (act) {
                uint8_t c = tb[T.te[r] + 1 + j];
                if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) {
                    uint32_t a = amb_code(c);
                    if (a == 0) atomicOr(&T.flag[r], 2u);
                    else {
                        uint8_t q = tb[T.se[r] + 3 + j];
                        atomicAdd(&T.flag[r], 256u);
                        atomicOr(&T.flag[r], (q < 33 || q > 40) ? 2u : 1u);
                    }
                }
            }
        }
    }
    __syncthreads();
    {
        uint32_t L = 0, kept = 0;
        if ((uint32_t)tid < nt) {
            uint32_t f = T.flag[tid];
            bool xfer = (f & 1) && !(f & 2);
            L = T.se[tid] - T.te[tid] - 1;
            kept = L - (xfer ? (f >> 8) : 0);
            A.slen[i0 + tid] = kept | ((uint32_t)xfer << 31);
            T.flag[tid] = xfer;
        } else {
            T.flag[tid] = 0;
        }
        uint32_t mq = wmax(L), ms = wmax(kept), sb_ = wsum(L);
        if (lane == 0) { atomicMax(&s_red[0], mq); atomicMax(&s_red[1], ms); atomicAdd(&s_red[2], sb_); }
    }
    __syncthreads();
    // pass 2: histograms
    uint32_t cA = 0, cC = 0, cG = 0, cT = 0;
    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    if (c0 < total) {
        uint32_t rs = first_record_of(T, nt, c0);
        for (uint32_t e0 = c0; e0 < c1; e0 += 64) {
            uint32_t r, j;
            bool act = locate(T, nt, e0 + lane, c1, rs, r, j);
            uint8_t c = 0;
            bool keep = false;
            uint32_t a = 1;
            if (act) {
                c = tb[T.te[r] + 1 + j];
                uint8_t q = tb[T.se[r] + 3 + j];
                a = amb_fast_t(c);
                bool drop = T.flag[r] && a > 1;
                keep = !drop;
                uint32_t qv = drop ? transfer_quality(a, q) : q;
                uint32_t row = j + 1, col = qv - 33;
                if (col < HCOLS && row < lds_rows) atomicAdd(&s_hist[row * HSTRIDE + col], 1u);
                else atomicAdd(&qh[(size_t)row * 256 + qv], 1u);
            }
            cA += __popcll(__ballot(keep && c == 'A'));
            cC += __popcll(__ballot(keep && c == 'C'));
            cG += __popcll(__ballot(keep && c == 'G'));
            cT += __popcll(__ballot(keep && c == 'T'));
            if (keep && a != 1) atomicAdd(&s_occ[c], 1u);
        }
    }
    if (lane == 0) {
        atomicAdd(&s_occ['A'], cA); atomicAdd(&s_occ['C'], cC); atomicAdd(&s_occ['G'], cG); atomicAdd(&s_occ['T'], cT);
    }
    __syncthreads();
    for (uint32_t k = tid; k < lds_rows * HCOLS; k += TL_THREADS) {
        uint32_t row = k / HCOLS, col = k % HCOLS;
        uint32_t v = s_hist[row * HSTRIDE + col];
        if (v) atomicAdd(&qh[(size_t)row * 256 + 33 + col], v);
    }
    if (s_occ[tid]) atomicAdd(&A.dna_occ[(size_t)s * 256 + tid], s_occ[tid]);
    if (tid == 0) {
        atomicMax(&A.sb[s].max_q, s_red[0]);
        atomicMax(&A.sb[s].max_s, s_red[1]);
        atomicAdd(&A.sb[s].n_bases, s_red[2]);
    }
}

// ====================================================================================================
// Measure: title flags/bits per record, quality/DNA bit totals per wave chunk
// ====================================================================================================
struct SqLut {
    const uint32_t *qtab;  // LDS or global; row stride qstride
    uint32_t qstride;
    const uint32_t *dcode; // LDS [256]
    const uint8_t *qcode;  // LDS [256]
};

__device__ __forceinline__ void load_luts(const Arena &A, uint32_t s, const SbState &sb, uint32_t *s_qtab,
                                          uint32_t lds_entries, uint32_t *s_dcode, uint8_t *s_qcode, SqLut &Lt)
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
    Lt.dcode = s_dcode;
    Lt.qcode = s_qcode;
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

__global__ __launch_bounds__(TL_THREADS) void k_measure_tile(Arena A, uint32_t R, uint32_t tile_bytes, uint32_t lds_entries)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_dyn[];  // tile, then quality code table
    __shared__ TileRecs T;
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256];
    __shared__ FieldDesc s_fd[MAX_FIELDS];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo, nf = sb.nf;
    const uint32_t t0 = blockIdx.x * R;
    if (t0 >= n) return;
    const uint32_t nt = t0 + R < n ? R : n - t0;
    const uint32_t i0 = rec_lo + t0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *s_qtab = reinterpret_cast<uint32_t *>(s_dyn + tile_bytes);

    for (uint32_t k = tid; k < nf * (sizeof(FieldDesc) / 4); k += TL_THREADS)
        reinterpret_cast<uint32_t *>(s_fd)[k] = reinterpret_cast<const uint32_t *>(A.fdesc + (size_t)s * MAX_FIELDS)[k];
    SqLut Lt;
    load_luts(A, s, sb, s_qtab, lds_entries, s_dcode, s_qcode, Lt);
    const uint8_t *tb = load_tile(A, i0, i0 + nt, s_dyn);
    tile_records(A, i0, nt, T);
    if ((uint32_t)tid < nt) T.flag[tid] = A.slen[i0 + tid] >> 31;  // N-transfer flag of the record
    __syncthreads();

    // ---- title: thread per record (tasks.cpp:393-509) ----
    {
        const bool valid = (uint32_t)tid < nt;
        const uint32_t gi = tid & 31;                  // index inside the 32-record group
        const uint32_t gstart = tid - gi;              // tile-relative first record of the group
        const uint32_t gsize = valid ? ((gstart + 32 <= nt) ? 32u : nt - gstart) : 0u;
        TitleDiffs d = {0, 0, 0};
        uint32_t p_cur = 0, p_prev = 0;
        if (valid) {
            p_cur = A.ls[i0 + tid];
            p_prev = gi ? A.ls[i0 + tid - 1] : p_cur;
            if (gi) d = title_record_diffs(tb, s_fd, nf, A.ls[i0 + gstart], p_prev, p_cur);
        }
        TitleDiffs gd = {g32_or(d.txt), g32_or(d.val), g32_or(d.dlt)};
        const uint32_t gconst = title_group_const(s_fd, nf, gd, gsize);
        uint32_t bits = 0;
        if (valid) {
            BitCounter bc;
            if (gi == 0) title_put_flags(s_fd, nf, gconst, bc);
            title_record_emit(tb, A, s, s_fd, nf, gi == 0, p_prev, p_cur, gconst, bc);
            bits = (uint32_t)bc.bits;
            A.tbits[i0 + tid] = bits;
        }
        uint32_t gbits = g32_sum(bits);
        if (valid && gi == 0) {
            uint32_t grp = A.grp_lo[s] + (t0 + gstart) / GROUP;
            A.goff[grp] = (gbits + 7) >> 3;
            A.gflags[grp] = gconst;
        }
    }
    // ---- quality / DNA: bit totals of this wave's chunk of symbols ----
    const uint32_t total = T.cum[TL_THREADS];
    const uint32_t chunk = ((total + TL_WAVES * 64 - 1) / (TL_WAVES * 64)) * 64;
    const uint32_t c0 = wave * chunk, c1 = c0 + chunk < total ? c0 + chunk : total;
    uint32_t qb = 0, db = 0;
    if (c0 < total) {
        uint32_t rs = first_record_of(T, nt, c0);
        for (uint32_t e0 = c0; e0 < c1; e0 += 64) {
            uint32_t r, j;
            bool act = locate(T, nt, e0 + lane, c1, rs, r, j);
            if (act) {
                uint8_t c = tb[T.te[r] + 1 + j], q = tb[T.se[r] + 3 + j];
                bool xfer = T.flag[r];
                uint32_t a = xfer ? amb_fast_t(c) : 1u;
                bool drop = xfer && a > 1;
                uint32_t qv = drop ? transfer_quality(a, q) : q;
                qb += Lt.qtab[(j + 1) * Lt.qstride + Lt.qcode[qv]] & 31;
                if (!drop) db += Lt.dcode[c] & 31;
            }
        }
    }
    qb = wsum(qb);
    db = wsum(db);
    if (lane == 0) {
        size_t w = ((size_t)A.tile_lo[s] + blockIdx.x) * TL_WAVES + wave;
        A.wq[w] = qb;
        A.wd[w] = db;
    }
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
    uint32_t tq = wg_scan_excl_t(A.wq + (size_t)A.tile_lo[s] * TL_WAVES, ntiles * TL_WAVES, s_wave);
    uint32_t td = wg_scan_excl_t(A.wd + (size_t)A.tile_lo[s] * TL_WAVES, ntiles * TL_WAVES, s_wave);
    uint32_t tg = wg_scan_excl_t(A.goff + A.grp_lo[s], (sb.n_rec + GROUP - 1) / GROUP, s_wave);
    if (threadIdx.x == 0) { sb.qbits = tq; sb.dbits = td; sb.title_body = tg; }
}

// ====================================================================================================
// Pack
// ====================================================================================================
// OR-writer for one lane's bits into a word array (LDS staging or the global output), MSB-first.
struct OrWriter {
    uint32_t *words;   // word 0 = absolute output word `word0`
    uint64_t bitpos;   // absolute bit position in the output buffer
    uint64_t word0;
    __device__ __forceinline__ void put(uint32_t v, uint32_t nb)
    {
        if (!nb) return;
        if (nb < 32) v &= (1u << nb) - 1u;
        const uint32_t sh = (uint32_t)(bitpos & 31);
        const uint64_t x = (uint64_t)v << (64 - nb - sh);  // nb <= 32, sh <= 31
        const uint64_t w = (bitpos >> 5) - word0;
        atomicOr(&words[w], __builtin_bswap32((uint32_t)(x >> 32)));
        if ((uint32_t)x) atomicOr(&words[w + 1], __builtin_bswap32((uint32_t)x));
        bitpos += nb;
    }
};

// Wave-cooperative emit of up to 64 codes into a word array (see k_pack_sq for the derivation).  All tails
// use atomicOr: the target is zero-initialised LDS staging (or, in the overflow path, the zeroed output).
__device__ __forceinline__ uint32_t wave_emit_or(uint32_t *words, uint64_t word0, uint64_t bitpos, uint32_t code, uint32_t len)
{
    const int lane = threadIdx.x & 63;
    const uint32_t incl = wscan_incl(len);
    const uint32_t total = __shfl(incl, 63);
    if (!total) return 0;
    const uint64_t o = bitpos + (incl - len);
    const uint32_t key = (uint32_t)((o >> 5) - word0), sh = (uint32_t)(o & 31);
    uint32_t hi = 0, lo = 0;
    if (len) {
        uint64_t x = (uint64_t)code << (64 - len - sh);
        hi = (uint32_t)(x >> 32);
        lo = (uint32_t)x;
    }
    const uint32_t key_prev = __shfl_up(key, 1), lo_prev = __shfl_up(lo, 1);
    if (lane > 0 && key_prev != key) hi |= lo_prev;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t ko = __shfl_up(key, d), vo = __shfl_up(hi, d);
        if (lane >= d && ko == key) hi |= vo;
    }
    const uint32_t key_next = __shfl_down(key, 1);
    const bool tail = lane == 63 || key_next != key;
    if (tail && hi) atomicOr(&words[key], __builtin_bswap32(hi));
    if (lane == 63 && lo) atomicOr(&words[key + 1], __builtin_bswap32(lo));
    return total;
}

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
    __shared__ TileRecs T;
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256];
    __shared__ FieldDesc s_fd[MAX_FIELDS];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo, nf = sb.nf;
    const uint32_t t0 = blockIdx.x * R;
    if (t0 >= n) return;
    const uint32_t nt = t0 + R < n ? R : n - t0;
    const uint32_t i0 = rec_lo + t0;
    const bool last_tile = t0 + R >= n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *s_qtab = reinterpret_cast<uint32_t *>(s_dyn + tile_bytes);
    uint32_t *s_stage = s_qtab + lds_entries;
    uint32_t *st_t = s_stage, *st_q = s_stage + sw_t, *st_d = s_stage + sw_t + sw_q;

    for (uint32_t k = tid; k < sw_t + sw_q + sw_d; k += TL_THREADS) s_stage[k] = 0;
    for (uint32_t k = tid; k < nf * (sizeof(FieldDesc) / 4); k += TL_THREADS)
        reinterpret_cast<uint32_t *>(s_fd)[k] = reinterpret_cast<const uint32_t *>(A.fdesc + (size_t)s * MAX_FIELDS)[k];
    SqLut Lt;
    load_luts(A, s, sb, s_qtab, lds_entries, s_dcode, s_qcode, Lt);
    const uint8_t *tb = load_tile(A, i0, i0 + nt, s_dyn);
    tile_records(A, i0, nt, T);
    if ((uint32_t)tid < nt) T.flag[tid] = A.slen[i0 + tid] >> 31;
    __syncthreads();
    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out);

    // absolute bit ranges of this tile's share of the three payload streams
    const uint32_t grp_first = A.grp_lo[s] + t0 / GROUP;
    const uint32_t ngrp_t = (nt + GROUP - 1) / GROUP;
    const uint64_t title_base = (sb.out_off + sb.info_len + sb.title_hdr) * 8;
    const uint64_t tb0 = title_base + 8ull * A.goff[grp_first];
    const uint64_t tb1 = title_base + 8ull * (last_tile ? sb.title_body : A.goff[grp_first + ngrp_t]);
    const size_t wix = ((size_t)A.tile_lo[s] + blockIdx.x) * TL_WAVES;
    const uint64_t qbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_hdr) * 8;
    const uint64_t dbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_len + sb.dna_hdr) * 8;
    const uint64_t qb0 = qbase + A.wq[wix], qb1 = qbase + (last_tile ? sb.qbits : A.wq[wix + TL_WAVES]);
    const uint64_t db0 = dbase + A.wd[wix], db1 = dbase + (last_tile ? sb.dbits : A.wd[wix + TL_WAVES]);
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
            OrWriter w;
            w.words = t_lds ? st_t : outw;
            w.word0 = t_lds ? tw0 : 0;
            w.bitpos = title_base + 8ull * A.goff[grp] + (incl - bits);
            const uint32_t p_cur = A.ls[i0 + tid], p_prev = gi ? A.ls[i0 + tid - 1] : p_cur;
            if (gi == 0) title_put_flags(s_fd, nf, gconst, w);
            title_record_emit(tb, A, s, s_fd, nf, gi == 0, p_prev, p_cur, gconst, w);
        }
    }

    // ---- quality / DNA: lane per symbol ----
    const uint32_t total = T.cum[TL_THREADS];
    const uint32_t chunk = ((total + TL_WAVES * 64 - 1) / (TL_WAVES * 64)) * 64;
    const uint32_t c0 = wave * chunk, c1 = c0 + chunk < total ? c0 + chunk : total;
    if (c0 < total) {
        uint64_t qpos = qbase + A.wq[wix + wave], dpos = dbase + A.wd[wix + wave];
        uint32_t *qdst = q_lds ? st_q : outw, *ddst = d_lds ? st_d : outw;
        const uint64_t qz = q_lds ? qw0 : 0, dz = d_lds ? dw0 : 0;
        uint32_t rs = first_record_of(T, nt, c0);
        for (uint32_t e0 = c0; e0 < c1; e0 += 64) {
            uint32_t r, j;
            bool act = locate(T, nt, e0 + lane, c1, rs, r, j);
            uint32_t qe = 0, de = 0;
            if (act) {
                uint8_t c = tb[T.te[r] + 1 + j], q = tb[T.se[r] + 3 + j];
                bool xfer = T.flag[r];
                uint32_t a = xfer ? amb_fast_t(c) : 1u;
                bool drop = xfer && a > 1;
                uint32_t qv = drop ? transfer_quality(a, q) : q;
                qe = Lt.qtab[(j + 1) * Lt.qstride + Lt.qcode[qv]];
                if (!drop) de = Lt.dcode[c];
            }
            qpos += wave_emit_or(qdst, qz, qpos, qe >> 5, qe & 31);
            dpos += wave_emit_or(ddst, dz, dpos, de >> 5, de & 31);
        }
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
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 << 10);
}

// Records per tile for a job whose longest record has max_rec_bytes; 0 = use the untiled kernels.
uint32_t tile_records_for(uint32_t max_rec_bytes, uint32_t r_max)
{
    for (uint32_t R : {256u, 192u, 128u, 64u})
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
    hipLaunchKernelGGL(k_title_stats_tile, grid, dim3(R), lds, st, A);
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
