// kernels_chunk.hip - sequence / quality kernels with one lane per 16-BYTE CHUNK of a read, out of an LDS-staged tile.
//
// A tile is a run of TR (<= 64) consecutive records of one sub-block, owned by ONE WAVE: the four waves of a workgroup
// share the code / histogram tables but stage, index and walk their own tiles, so the tile loop has no workgroup barrier
// and a wave that waits for its loads leaves the SIMD to the others.  The tile's bytes [ls[i0], ls[i0 + nt]) come into
// LDS with coalesced, aligned 16-byte loads (a wave instruction covers 1 KiB of the file; the lane-per-record kernels of
// kernels_sq.hip touch 64 different cache lines per instruction and are bound by the texture addresser).  The work
// list of a tile is its 16-byte chunks - chunk k of record r = bases / qualities [16k, 16k + 16) - and every lane
// takes one chunk per trip: all lanes carry a full chunk whatever the read lengths are (a lane-per-record wave
// runs as long as its longest read), neighbouring lanes sit at different read positions (histogram rows and code
// table rows spread over the LDS banks without tricks), and the unaligned 16 bytes of a chunk are five LDS dwords
// and four v_alignbit.
//
//   k_ck_stats    N-transfer test, DNA symbol counts, per-position quality histogram      (phyNGSC.cpp:462-653,
//                                                                                         tasks.cpp:260-286)
//   k_ck_measure  quality / DNA payload bits per tile
//   k_ck_pack     StoreQuality and StoreDNA payloads (tasks.cpp:544-557, 609-619): the bits of a chunk are measured,
//                 placed by a workgroup scan over the chunks of the trip and ORed into LDS staging; one coalesced
//                 copy-out per tile and stream
// Integer / byte work, bounded by the LDS queue (histogram atomics, table lookups) and VALU issue; no MFMA.
#include "ngsc_kernels.h"
#include "ngsc_sq.h"

namespace ngsc {

constexpr int CK_THREADS = 256;
constexpr int CK_WAVES = CK_THREADS / 64;
constexpr uint32_t CK_PAD_WORDS = 8;        // LDS words behind the span: a chunk gather reads up to 19 bytes past its start

constexpr uint32_t CK_NV = 8;               // 16-byte pieces of a tile per lane: a tile is at most 8 KB

struct CkLds {
    uint32_t *in;    // the tile's bytes
    uint32_t *rec;   // [TR]  seq start (offset in the span) | read length << 16 | transfer flag << 31
    uint2 *map;      // [map_cap]  work list: the record's rec word, record | chunk << 16
};
struct CkTile {
    uint32_t b0;        // job-buffer offset of s_in[0]
    uint32_t nchunks;   // work list length
    uint32_t nfull;     // FULL_FIRST: the first nfull entries are complete 16-byte chunks, the rest are read tails
};
// A tile on its way from global memory: issued one tile ahead, so that the loads fly while the wave works
struct CkPre {
    uint32_t b0, b1;    // span [b0, b1) of the job buffer, b0 16-byte aligned
    uint32_t te, se, sl;
    uint4 v[CK_NV];
};

// LDS traffic between the lanes of one wave: the hardware runs a wave's LDS instructions in order; this keeps the
// compiler from moving reads over writes
__device__ __forceinline__ void ck_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// inclusive scan over the 64 lanes with six DPP adds (four inside the rows of 16, two across them); every lane active
__device__ __forceinline__ uint32_t ck_wscan_incl(uint32_t v)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}
// two exclusive scans over the 64 lanes at once: a and b packed as a | b << 16 (every prefix of either below 2^16)
__device__ __forceinline__ void ck_scan2(uint32_t a, uint32_t b, uint32_t &ea, uint32_t &eb, uint32_t &ta, uint32_t &tb)
{
    const uint32_t v = a | (b << 16);
    const uint32_t incl = ck_wscan_incl(v);
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t ex = incl - v;
    ea = ex & 0xffffu;
    eb = ex >> 16;
    ta = tot & 0xffffu;
    tb = tot >> 16;
}

// Issues the loads of a tile: record table entries of lane < nt and the tile's bytes in 16-byte pieces (all of them
// back to back: one round trip).  The piece that would reach past the job buffer is left to ck_commit().
typedef unsigned int ck_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ck_prefetch(const Arena &A, uint32_t i0, uint32_t nt, bool want_xfer, CkPre &P)
{
    const uint32_t lane = threadIdx.x & 63;
    // buffer loads: 32-bit offsets, and the hardware returns zero for every dword past buf_len (a file whose length is
    // not a multiple of four loses the bytes of its last dword here: ck_commit() fetches those)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(A.buf), (short)0, (int)A.buf_len, 0x00020000);
    P.b0 = A.ls[i0] & ~15u;
    P.b1 = A.ls[i0 + nt];
    P.te = P.se = P.sl = 0;
    if (lane < nt) {
        P.te = A.te[i0 + lane];
        P.se = A.se[i0 + lane];
        if (want_xfer) P.sl = A.slen[i0 + lane];
    }
    const uint32_t n16 = (P.b1 - P.b0 + 15) >> 4;
    const uint32_t voff = P.b0 + 16 * lane;
#pragma unroll
    for (uint32_t j = 0; j < CK_NV; ++j) {
        P.v[j] = make_uint4(0, 0, 0, 0);
        if (64 * j < n16) {   // (uniform)
            const ck_u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)(1024 * j), 0);
            P.v[j] = make_uint4(x.x, x.y, x.z, x.w);
        }
    }
}

// Stages the prefetched tile in LDS, builds its record table and work list.  FULL_FIRST: complete chunks first (record
// order), read tails behind them, so that whole waves take the paths without per-byte masks; otherwise stream order
// (record, chunk), which is the order of the payload bits.  One wave, no workgroup barrier.  Returns
// nchunks == 0xffffffff when the tile does not fit.
template <bool FULL_FIRST>
__device__ __forceinline__ CkTile ck_commit(const Arena &A, uint32_t nt, const CkLds &M, uint32_t in_words, uint32_t map_cap, const CkPre &P)
{
    const uint32_t lane = threadIdx.x & 63;
    CkTile T;
    T.b0 = P.b0;
    uint32_t L = 0, nf = 0, tail = 0, recw = 0;
    if (lane < nt) {
        L = P.se - P.te - 1;
        recw = (P.te + 1 - T.b0) | (L << 16) | (P.sl & 0x80000000u);
        M.rec[lane] = recw;
        nf = L >> 4;
        tail = (L & 15u) ? 1u : 0u;
    }
    uint32_t e0, e1, t0, t1;
    if (FULL_FIRST) ck_scan2(nf, tail, e0, e1, t0, t1);
    else ck_scan2(nf + tail, 0u, e0, e1, t0, t1);
    T.nfull = FULL_FIRST ? t0 : 0u;
    T.nchunks = t0 + t1;
    const uint32_t n16 = (P.b1 - T.b0 + 15) >> 4;
    if (T.nchunks > map_cap || 4 * n16 + CK_PAD_WORDS > in_words || n16 > 64 * CK_NV || (P.b1 - T.b0) > 0xffffu) {   // (uniform)
        T.nchunks = 0xffffffffu;
        return T;
    }
    if (lane < nt) {
        for (uint32_t k = 0; k < nf; ++k) M.map[e0 + k] = make_uint2(recw, lane | (k << 16));
        if (tail) M.map[FULL_FIRST ? t0 + e1 : e0 + nf] = make_uint2(recw, lane | (nf << 16));
    }
    uint4 *dst = reinterpret_cast<uint4 *>(M.in);
#pragma unroll
    for (uint32_t j = 0; j < CK_NV; ++j) {
        const uint32_t k = lane + 64 * j;
        if (k < n16) dst[k] = P.v[j];
    }
    if (lane < CK_PAD_WORDS) M.in[4 * n16 + lane] = 0;
    const uint32_t whole = A.buf_len & ~3u;   // the last, incomplete dword of the job buffer came back as zero
    if (P.b1 > whole && whole >= T.b0) {      // (uniform)
        ck_wave_sync();
        if (lane == 0) {
            uint32_t w = 0;
            for (uint32_t x = 0; whole + x < A.buf_len; ++x) w |= (uint32_t)A.buf[whole + x] << (8 * x);
            M.in[(whole - T.b0) >> 2] = w;
        }
    }
    ck_wave_sync();
    return T;
}

// 16 bytes at byte offset a of the span
__device__ __forceinline__ uint4 ck_gather16(const uint32_t *s_in, uint32_t a)
{
    const uint32_t w = a >> 2, sh = (a & 3u) * 8u;
    const uint32_t x0 = s_in[w], x1 = s_in[w + 1], x2 = s_in[w + 2], x3 = s_in[w + 3], x4 = s_in[w + 4];
    return make_uint4(__builtin_amdgcn_alignbit(x1, x0, sh), __builtin_amdgcn_alignbit(x2, x1, sh),
                      __builtin_amdgcn_alignbit(x3, x2, sh), __builtin_amdgcn_alignbit(x4, x3, sh));
}
__device__ __forceinline__ uint32_t ck_byte(const uint32_t *s_in, uint32_t a) { return (s_in[a >> 2] >> (8 * (a & 3u))) & 0xffu; }

struct CkChunk {
    bool act;
    uint32_t r, k, L, s0, nv;
    bool xfer;
};
__device__ __forceinline__ CkChunk ck_chunk(const CkLds &M, const CkTile &T, uint32_t c)
{
    CkChunk C;
    C.act = c < T.nchunks;
    const uint2 m = M.map[C.act ? c : 0];
    C.r = m.y & 0xffffu;
    C.k = m.y >> 16;
    const uint32_t rec = m.x;
    C.L = (rec >> 16) & 0x7fffu;
    C.s0 = rec & 0xffffu;
    C.xfer = (rec >> 31) != 0;
    const uint32_t left = C.L - 16 * C.k;
    C.nv = C.act ? (left < 16 ? left : 16) : 0;
    return C;
}

// ====================================================================================================
// Statistics
// ====================================================================================================
__global__ __launch_bounds__(CK_THREADS) void k_ck_stats(Arena A, uint32_t TR, uint32_t in_words, uint32_t map_cap, uint32_t lds_rows,
                                                         uint32_t tpw)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_dyn[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per_wave = in_words + TR + 2 * map_cap + 2 * TR;
    CkLds M;                             // this wave's tile
    M.in = s_dyn + wave * per_wave;
    M.rec = M.in + in_words;
    M.map = reinterpret_cast<uint2 *>(M.rec + TR);
    uint32_t *s_fl = M.rec + TR + 2 * map_cap;    // [TR] N-transfer test per record: bit0 some transferable code, bit1 a failure
    uint32_t *s_na = s_fl + TR;          // [TR] number of IUPAC codes of the record
    uint32_t *s_hist = s_dyn + CK_WAVES * per_wave;   // [lds_rows][SR_HSTRIDE], then 64 trash words (shared by the waves)
    __shared__ uint32_t s_occ[256];
    __shared__ uint8_t s_amb[256];
    __shared__ uint32_t s_red[4];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + TR - 1) / TR;
    const uint32_t tile_first = blockIdx.x * tpw;
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;
    const uint32_t hwords = lds_rows * SR_HSTRIDE;

    for (uint32_t k = tid; k < hwords; k += CK_THREADS) s_hist[k] = 0;
    s_occ[tid] = 0;
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (tid < 4) s_red[tid] = 0;

    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    uint8_t *hb = reinterpret_cast<uint8_t *>(s_hist);
    const uint32_t trash = (hwords + lane) * 4;   // 64 words after the histogram
    uint32_t t0 = 0, t1 = 0, t01 = 0, nok = 0, mq = 0, ms = 0, sum_l = 0;
    // every lane walks the 16 positions of its chunk in its own rotated order (word lj first, byte order bsh): lanes that
    // sit at the same read positions would otherwise serialise the LDS atomics on the popular quality values
    const uint32_t lj = lane & 3;
    uint32_t bsh[4], boff[4];
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t bb = (b + (lane >> 2)) & 3;
        bsh[b] = 8 * bb;
        boff[b] = bb * SR_HSTRIDE * 4;
    }

    __syncthreads();   // the tables are cleared
    CkPre P;
    if (tile_first + wave < tile_last) {
        const uint32_t t0r = (tile_first + wave) * TR;
        ck_prefetch(A, rec_lo + t0r, t0r + TR < n ? TR : n - t0r, false, P);
    }
    for (uint32_t tile = tile_first + wave; tile < tile_last; tile += CK_WAVES) {
        const uint32_t t0r = tile * TR;
        const uint32_t nt = t0r + TR < n ? TR : n - t0r;
        const uint32_t i0 = rec_lo + t0r;
        if (lane < TR) { s_fl[lane] = 0; s_na[lane] = 0; }
        bool amb_seen = false;
        const CkTile T = ck_commit<true>(A, nt, M, in_words, map_cap, P);
        if (tile + CK_WAVES < tile_last) {   // the next tile's loads fly while this one is worked on
            const uint32_t t1r = (tile + CK_WAVES) * TR;
            ck_prefetch(A, rec_lo + t1r, t1r + TR < n ? TR : n - t1r, false, P);
        }
        if (T.nchunks == 0xffffffffu) { if (lane == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); break; }
        if (lane < nt) {
            // line 3 of every record must be a bare '+' (README.md:97-103); the reference assumes it
            const uint32_t rec = M.rec[lane], L = (rec >> 16) & 0x7fffu, sq0 = rec & 0xffffu;
            if (ck_byte(M.in, sq0 + L + 1) != '+' || ck_byte(M.in, sq0 + L + 2) != '\n') atomicMin(A.err, (int32_t)ERR_FORMAT);
            if (t0r + lane == 0 && L >= 1 && !A.cs_mode) {   // colour-space detection on the sub-block's first record, phyNGSC.cpp:478
                if (cs_first_read(ck_byte(M.in, sq0), ck_byte(M.in, sq0 + 1))) atomicMin(A.err, (int32_t)ERR_COLORSPACE);
            }
            mq = L > mq ? L : mq;
            sum_l += L;
        }

        // ---- pass 1: everything except the ambiguous (non-ACGT) positions, whose fate depends on the whole record ----
        for (uint32_t base = 0; base < T.nchunks; base += 64) {
            const CkChunk C = ck_chunk(M, T, base + lane);
            const uint32_t nv = C.nv;
            const uint4 S = ck_gather16(M.in, C.s0 + 16 * C.k), Q = ck_gather16(M.in, C.s0 + C.L + 3 + 16 * C.k);
            const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
            uint32_t d[4] = {0, 0, 0, 0}, idx[4] = {0, 0, 0, 0}, dacc = 0, qbad = 0, c0 = 0, c1 = 0, c01 = 0;
            const bool wany[4] = {true, (bool)__any(nv > 4), (bool)__any(nv > 8), (bool)__any(nv > 12)};
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (!wany[k]) continue;
                const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                const uint32_t w = (sw[k] & vm) | (0x41414141u & ~vm);   // bytes past the read count as 'A': no bits below
                idx[k] = sr_idx(w);
                d[k] = sr_not_acgt(w, idx[k]);
                dacc |= d[k];
                qbad |= sr_q_out(qw[k]) & vm;
                const uint32_t p0 = idx[k] & 0x01010101u, p1 = (idx[k] >> 1) & 0x01010101u;
                c0 += __builtin_popcount(p0);
                c1 += __builtin_popcount(p1);
                c01 += __builtin_popcount(p0 & p1);
            }
            const uint32_t row0 = 16 * C.k + 1;                           // histogram row of the chunk's first position
            const uint32_t rb = row0 * SR_HSTRIDE * 4 - 33 * 4;           // byte offset of (row0, quality 0)
            const bool clean = dacc == 0 && qbad == 0;
            const unsigned long long live = __ballot(nv != 0);
            const uint32_t nv_first = live ? (uint32_t)__builtin_amdgcn_readlane((int)nv, __builtin_ctzll(live)) : 0u;
            if (!__any(nv != 0 && !(clean && nv == 16 && row0 + 16 <= lds_rows))) {
                if (nv) {
                    // tasks.cpp:268-283: position histogram rows 1..max_q
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t j = (k + lj) & 3;
                        const uint32_t wsel = j == 0 ? qw[0] : (j == 1 ? qw[1] : (j == 2 ? qw[2] : qw[3]));
                        const uint32_t joff = rb + j * (4 * SR_HSTRIDE * 4);
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t q = (wsel >> bsh[b]) & 0xffu;
                            atomicAdd(reinterpret_cast<uint32_t *>(hb + joff + boff[b] + 4 * q), 1u);
                        }
                    }
                    t0 += c0; t1 += c1; t01 += c01; nok += 16;
                }
            } else if (!__any(nv != 0 && !(clean && nv == nv_first && row0 + nv <= lds_rows))) {
                // read tails of equally long reads: the same nv < 16 bytes in every lane, nothing odd about them
                if (nv) {
                    uint32_t u = lane % nv_first;   // (rotated start: the lanes sit at the same positions)
                    for (uint32_t i = 0; i < nv_first; ++i) {
                        const uint32_t w = u < 4 ? qw[0] : (u < 8 ? qw[1] : (u < 12 ? qw[2] : qw[3]));
                        const uint32_t q = (w >> (8 * (u & 3))) & 0xffu;
                        atomicAdd(reinterpret_cast<uint32_t *>(hb + rb + u * SR_HSTRIDE * 4 + 4 * q), 1u);
                        u = u + 1 == nv_first ? 0 : u + 1;
                    }
                    t0 += c0; t1 += c1; t01 += c01; nok += nv;
                }
            } else {
                uint32_t ambm = 0, qbm = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (!wany[k]) continue;
                    ambm |= t_movemask(t_nonzero01(d[k])) << (4 * k);
                    qbm |= t_movemask((sr_q_out(qw[k]) >> 7) & 0x01010101u) << (4 * k);
                }
                const uint32_t validm = (1u << nv) - 1u;
                const uint32_t rows_left = lds_rows > row0 ? lds_rows - row0 : 0;   // rows row0.. inside LDS
                const uint32_t rowm = rows_left >= 16 ? 0xffffu : ((1u << rows_left) - 1u);
                ambm &= validm;
                const uint32_t okm = validm & ~ambm & ~qbm & rowm;    // added to the LDS histogram now
                const uint32_t glm = validm & ~ambm & ~okm;           // unambiguous but outside the LDS histogram
                if (ambm) {   // recount the bases without the ambiguous ones
                    c0 = c1 = c01 = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t ok01 = t_nonzero01(d[k]) ^ 0x01010101u;
                        const uint32_t p0 = idx[k] & ok01, p1 = (idx[k] >> 1) & ok01;
                        c0 += __builtin_popcount(p0);
                        c1 += __builtin_popcount(p1);
                        c01 += __builtin_popcount(p0 & p1);
                    }
                }
                t0 += c0; t1 += c1; t01 += c01; nok += nv - __builtin_popcount(ambm);
                if (__any(okm)) {   // branch-free: bytes that do not count go to the lane's own trash word
                    const uint32_t rbs = rows_left ? rb : 0u;   // (a lane without rows in LDS has okm == 0: any valid address)
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t j = (k + lj) & 3;
                        const uint32_t wsel = j == 0 ? qw[0] : (j == 1 ? qw[1] : (j == 2 ? qw[2] : qw[3]));
                        const uint32_t joff = rbs + j * (4 * SR_HSTRIDE * 4);
                        const uint32_t okw = okm >> (4 * j);
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t q = (wsel >> bsh[b]) & 0xffu;
                            const uint32_t a = ((okw >> (bsh[b] >> 3)) & 1u) ? joff + boff[b] + 4 * q : trash;
                            atomicAdd(reinterpret_cast<uint32_t *>(hb + a), 1u);
                        }
                    }
                }
                if (__any(glm)) {
#pragma unroll
                    for (uint32_t u = 0; u < 16; ++u)
                        if ((glm >> u) & 1u) atomicAdd(&qh[(size_t)(row0 + u) * 256 + sr_byte(qw, u)], 1u);
                }
                if (__any(ambm)) {
                    amb_seen = true;
                    // N-transfer test of the record (phyNGSC.cpp:549-571): every non-ACGT base must be an IUPAC code whose
                    // quality lies in '!'..'('.  The serial loop breaks at the first failure, so the outcome is "some code"
                    // and "no failure anywhere": order independent, collected per record with LDS atomics.
                    uint32_t cfl = 0, cnt = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        if (!__any((ambm >> (4 * k)) & 15u)) continue;
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t u = 4 * k + b;
                            if ((ambm >> u) & 1u) {
                                const uint32_t a = s_amb[sr_byte(sw, u)];
                                if (a == 0) cfl |= 2;
                                else {
                                    const uint32_t q = sr_byte(qw, u);
                                    cfl |= (q < 33 || q > 40) ? 2u : 1u;
                                    ++cnt;
                                }
                            }
                        }
                    }
                    if (ambm) {
                        atomicOr(&s_fl[C.r], cfl);
                        if (cnt) atomicAdd(&s_na[C.r], cnt);
                    }
                }
            }
        }
        ck_wave_sync();

        // ---- pass 2: the ambiguous positions, now that every record's transfer decision is known (phyNGSC.cpp:573-599) ----
        if (amb_seen) {
            for (uint32_t base = 0; base < T.nchunks; base += 64) {
                const CkChunk C = ck_chunk(M, T, base + lane);
                const uint32_t fl = C.act ? s_fl[C.r] : 0u;
                if (!__any(fl != 0)) continue;
                if (fl) {
                    const uint4 S = ck_gather16(M.in, C.s0 + 16 * C.k), Q = ck_gather16(M.in, C.s0 + C.L + 3 + 16 * C.k);
                    const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
                    const uint32_t ambm = sr_amb_mask(sw, C.nv);
                    const bool xfer = (fl & 1) && !(fl & 2);
#pragma unroll
                    for (uint32_t u = 0; u < 16; ++u) {
                        if ((ambm >> u) & 1u) {
                            const uint32_t c = sr_byte(sw, u), q = sr_byte(qw, u), row = 16 * C.k + u + 1;
                            const uint32_t a = s_amb[c];
                            if (xfer && a > 1) {
                                atomicAdd(&qh[(size_t)row * 256 + transfer_quality(a, (uint8_t)q)], 1u);   // 128..247: no LDS column
                            } else {
                                const uint32_t col = q - 33;
                                if (col < SR_HCOLS && row < lds_rows) atomicAdd(&s_hist[row * SR_HSTRIDE + col], 1u);
                                else atomicAdd(&qh[(size_t)row * 256 + q], 1u);
                                atomicAdd(&s_occ[c], 1u);
                            }
                        }
                    }
                }
            }
        }
        if (lane < nt) {
            const uint32_t L = (M.rec[lane] >> 16) & 0x7fffu, fl = s_fl[lane];
            const bool xfer = (fl & 1) && !(fl & 2);
            const uint32_t kept = L - (xfer ? s_na[lane] : 0u);
            A.slen[i0 + lane] = kept | ((uint32_t)xfer << 31);
            ms = kept > ms ? kept : ms;
        }
        ck_wave_sync();   // (the record table is rewritten by the next tile)
    }

    // ---- workgroup totals ----
    {
        const uint32_t nG = t01, nC = t0 - t01, nT = t1 - t01, nA = nok - t0 - t1 + t01;
        const uint32_t wa = sr_wsum(nA), wc = sr_wsum(nC), wg = sr_wsum(nG), wt = sr_wsum(nT);
        const uint32_t wq = sr_wmax(mq), wms = sr_wmax(ms), wl = sr_wsum(sum_l);
        if (lane == 0) {
            atomicAdd(&s_occ['A'], wa); atomicAdd(&s_occ['C'], wc); atomicAdd(&s_occ['G'], wg); atomicAdd(&s_occ['T'], wt);
            atomicMax(&s_red[0], wq); atomicMax(&s_red[1], wms); atomicAdd(&s_red[2], wl);
        }
    }
    __syncthreads();
    for (uint32_t k = tid; k < lds_rows * SR_HCOLS; k += CK_THREADS) {
        const uint32_t row = k / SR_HCOLS, col = k % SR_HCOLS;
        const uint32_t x = s_hist[row * SR_HSTRIDE + col];
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
// Measure / pack
// ====================================================================================================
// Quality code tables in LDS, indexed by the raw quality byte: row = read position + 1 (tasks.cpp:609-619 uses tree j + 1
// for position j), CK_QS columns per row: column q - 33 for the bytes 33..80, column 48 always zero.  A byte that has no
// entry - outside 33..80 (sr_q_out), a transferred quality 128..247, a row past the LDS budget, a byte behind the end of
// the read - is replaced by CK_QPAD before the lookup, reads the zero entry, and the ones that do carry a code are
// patched from the global table through qua_code afterwards.  So the 16 lookups of a chunk are straight-line code: 16 LDS
// reads in flight together, one wait (lookups interleaved with the exception tests cost one LDS round trip per dword).
constexpr uint32_t CK_QS = SR_HCOLS + 1;
constexpr uint32_t CK_QPAD = 0x51515151u;        // 33 + 48 in every byte
constexpr uint32_t CK_QROWS_EXTRA = 16;          // zero rows behind the table: the pad bytes of a chunk at the table's end

template <class TE>
__device__ __forceinline__ void ck_fill_qtab(const Arena &A, uint32_t s, const SbState &sb, TE *s_qtab, uint32_t lds_rows)
{
    const uint32_t nq = sb.nqual;
    const uint32_t *qg = A.qcodes + (size_t)s * A.qrows * 256;
    for (uint32_t k = threadIdx.x; k < (lds_rows + CK_QROWS_EXTRA) * CK_QS; k += CK_THREADS) s_qtab[k] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < lds_rows * nq; k += CK_THREADS) {
        const uint32_t row = k / nq, ci = k % nq;
        const uint32_t col = (uint32_t)sb.qualities[ci] - 33u;
        if (col < SR_HCOLS) {
            const uint32_t e = qg[(size_t)row * 256 + ci];
            s_qtab[row * CK_QS + col] = sizeof(TE) == 1 ? (TE)(e & 31u) : (TE)e;
        }
    }
}

struct CkTabs {
    const uint32_t *qg;     // global [qrows][256] by code index
    const uint8_t *qcode;   // LDS [256] quality byte -> code index
    const uint8_t *amb;     // LDS [256]
    const uint32_t *dcode;  // LDS [256] base byte -> code << 5 | len
};

// A chunk with its bytes: fetched one trip ahead, so that the LDS round trips (work list entry, then the gathers) run
// beside the previous chunk's arithmetic.
struct CkWork {
    CkChunk C;
    uint4 S, Q;
};
__device__ __forceinline__ CkWork ck_fetch(const CkLds &M, const CkTile &T, uint32_t c, bool need_s = true)
{
    CkWork W;
    W.C = ck_chunk(M, T, c);
    W.S = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
    if (need_s) W.S = ck_gather16(M.in, W.C.s0 + 16 * W.C.k);   // (uniform)
    W.Q = ck_gather16(M.in, W.C.s0 + W.C.L + 3 + 16 * W.C.k);
    return W;
}

// Classification of a chunk: per dword the A/C/T/G index bytes, then 16-bit masks (one bit per byte) of the ambiguous
// bases and of the quality bytes without a table column, and the quality dwords with every byte that has no table entry
// replaced by the pad byte.
struct CkCls {
    uint32_t idx[4];
    uint32_t ambm, qbm, validm, qpatch;
    uint32_t qwm[4];
    uint32_t row0, rowc;    // first table row of the chunk, and the same clamped into the table (+ zero rows)
};
// FULL: every lane holds a complete chunk (no validity masks).  LEAN: the caller knows that every valid quality byte
// below 128 has a table column and every row is in the table (a property of the sub-block's alphabet / the launch) and
// that no byte of this trip has its high bit set: the range test per byte and the row arithmetic fall away.
template <bool FULL, bool LEAN>
__device__ __forceinline__ CkCls ck_classify(const uint32_t (&sw)[4], const uint32_t (&qw)[4], const CkChunk &C, uint32_t lds_rows)
{
    CkCls K;
    const uint32_t nv = FULL ? 16u : C.nv;
    uint32_t qbad = 0, dacc = 0, bad[4], vmk[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t vm = FULL ? ~0u : t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
        vmk[k] = vm;
        bad[k] = LEAN ? 0u : (sr_q_out(qw[k]) & vm);
        qbad |= bad[k];
        const uint32_t w = FULL ? sw[k] : ((sw[k] & vm) | (0x41414141u & ~vm));
        K.idx[k] = sr_idx(w);
        dacc |= sr_not_acgt(w, K.idx[k]);
    }
    K.ambm = K.qbm = 0;
    K.validm = FULL ? 0xffffu : ((1u << nv) - 1u);
    K.row0 = 16 * C.k + 1;
    const uint32_t rows_left = LEAN ? 16u : (lds_rows > K.row0 ? lds_rows - K.row0 : 0);
    const uint32_t rowm = rows_left >= 16 ? 0xffffu : ((1u << rows_left) - 1u);
    K.rowc = LEAN ? K.row0 : (K.row0 < lds_rows ? K.row0 : lds_rows);
    uint32_t keep[4] = {vmk[0], vmk[1], vmk[2], vmk[3]};   // bytes that keep their value for the lookup
    if (__any(dacc != 0 || qbad != 0 || (C.act && rows_left < nv))) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t w = (sw[k] & vmk[k]) | (0x41414141u & ~vmk[k]);
            const uint32_t a01 = t_nonzero01(sr_not_acgt(w, K.idx[k]));
            K.ambm |= t_movemask(a01) << (4 * k);
            K.qbm |= t_movemask((bad[k] >> 7) & 0x01010101u) << (4 * k);
            uint32_t drop = (bad[k] >> 7) * 0xffu;                    // bytes without a column
            if (C.xfer) drop |= a01 * 0xffu;                          // transferred qualities: phyNGSC.cpp:579
            const uint32_t rl = rows_left > 4 * k ? rows_left - 4 * k : 0;
            drop |= ~t_lowbytes32((int32_t)rl);                        // rows past the table
            keep[k] &= ~drop;
        }
        K.ambm &= K.validm;
    }
    K.qpatch = (K.qbm | (C.xfer ? K.ambm : 0u) | (K.validm & ~rowm));
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) K.qwm[k] = (FULL && LEAN && false) ? qw[k] : ((qw[k] & keep[k]) | (CK_QPAD & ~keep[k]));
    return K;
}

// does the sub-block's quality alphabet allow the LEAN forms?  (all symbols below 128 inside the table's columns, all rows
// in the table; thread = byte value, called by the whole workgroup)
__device__ __forceinline__ bool ck_lean_alphabet(const SbState &sb, uint32_t lds_rows, uint32_t *s_flag)
{
    if (threadIdx.x == 0) *s_flag = 0;
    __syncthreads();
    if (threadIdx.x < sb.nqual) {
        const uint32_t q = sb.qualities[threadIdx.x];
        if (q < 128 && (q < 33 || q > 32 + SR_HCOLS)) atomicOr(s_flag, 1u);
    }
    __syncthreads();
    return *s_flag == 0 && lds_rows > sb.max_q;
}
// no byte of the chunk has its high bit set (bytes past the read included: the test errs on the safe side)
__device__ __forceinline__ bool ck_low_bytes(const uint32_t (&qw)[4]) { return ((qw[0] | qw[1] | qw[2] | qw[3]) & 0x80808080u) == 0; }

__global__ __launch_bounds__(CK_THREADS) void k_ck_measure(Arena A, uint32_t TR, uint32_t in_words, uint32_t map_cap, uint32_t lds_rows,
                                                           uint32_t tpw)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_dyn[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per_wave = in_words + TR + 2 * map_cap;
    CkLds M;                             // this wave's tile
    M.in = s_dyn + wave * per_wave;
    M.rec = M.in + in_words;
    M.map = reinterpret_cast<uint2 *>(M.rec + TR);
    uint8_t *s_qlen = reinterpret_cast<uint8_t *>(s_dyn + CK_WAVES * per_wave);   // [lds_rows + 16][CK_QS] code lengths, shared
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + TR - 1) / TR;
    const uint32_t tile_first = blockIdx.x * tpw;
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;

    s_dcode[tid] = A.dcodes[(size_t)s * 256 + tid];
    s_qcode[tid] = sb.qua_code[tid];
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    ck_fill_qtab(A, s, sb, s_qlen, lds_rows);
    __shared__ uint32_t s_flag;
    const bool lean_wg = ck_lean_alphabet(sb, lds_rows, &s_flag);
    const CkTabs Tb = {A.qcodes + (size_t)s * A.qrows * 256, s_qcode, s_amb, s_dcode};
    const bool plain = (sb.flags & F_DNA_PLAIN) != 0;
    // code lengths of A, C, T, G (the order of (c >> 1) & 3) as a v_perm table
    const uint32_t ltab = (s_dcode['A'] & 31u) | ((s_dcode['C'] & 31u) << 8) | ((s_dcode['T'] & 31u) << 16) | ((s_dcode['G'] & 31u) << 24);

    CkPre P;
    if (tile_first + wave < tile_last) {
        const uint32_t t0r = (tile_first + wave) * TR;
        ck_prefetch(A, rec_lo + t0r, t0r + TR < n ? TR : n - t0r, true, P);
    }
    for (uint32_t tile = tile_first + wave; tile < tile_last; tile += CK_WAVES) {
        const uint32_t t0r = tile * TR;
        const uint32_t nt = t0r + TR < n ? TR : n - t0r;
        const uint32_t my_slen = P.sl;
        const CkTile T = ck_commit<true>(A, nt, M, in_words, map_cap, P);   // (any order: complete chunks first)
        if (tile + CK_WAVES < tile_last) {   // the next tile's loads fly while this one is worked on
            const uint32_t t1r = (tile + CK_WAVES) * TR;
            ck_prefetch(A, rec_lo + t1r, t1r + TR < n ? TR : n - t1r, true, P);
        }
        if (T.nchunks == 0xffffffffu) { if (lane == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); break; }
        uint32_t qb = 0;
        uint32_t db = (plain && lane < nt) ? 2u * (my_slen & 0x0fffffffu) : 0u;   // two bits per base that stays (tasks.cpp:544-548)
        // the bases matter for Huffman DNA lengths and for the transferred positions of this tile's records
        const bool need_s = !plain || __any(lane < nt && (my_slen >> 31));
        CkWork W = ck_fetch(M, T, lane, need_s);
        for (uint32_t base = 0; base < T.nchunks; base += 64) {
            const CkChunk C = W.C;
            const uint32_t sw[4] = {W.S.x, W.S.y, W.S.z, W.S.w}, qw[4] = {W.Q.x, W.Q.y, W.Q.z, W.Q.w};
            if (base + 64 < T.nchunks) W = ck_fetch(M, T, base + 64 + lane, need_s);
            const uint32_t nv = C.nv;
            const bool full = base + 64 <= T.nfull;                 // (complete chunks come first in the work list)
            const bool lean = lean_wg && !__any(!ck_low_bytes(qw));
            CkCls K;
            if (full && lean) K = ck_classify<true, true>(sw, qw, C, lds_rows);
            else if (lean) K = ck_classify<false, true>(sw, qw, C, lds_rows);
            else K = ck_classify<false, false>(sw, qw, C, lds_rows);
            // quality bits: 16 table lengths at once, then the exceptions
            const uint32_t abase = K.rowc * CK_QS - 33u;
            uint32_t l[16];
#pragma unroll
            for (uint32_t u = 0; u < 16; ++u) l[u] = s_qlen[abase + u * CK_QS + sr_byte(K.qwm, u)];
            if (__any(K.qpatch)) {
#pragma unroll
                for (uint32_t u = 0; u < 16; ++u) {
                    if ((K.qpatch >> u) & 1u) {
                        uint32_t q = sr_byte(qw, u);
                        if (C.xfer && ((K.ambm >> u) & 1u)) q = transfer_quality(s_amb[sr_byte(sw, u)], (uint8_t)q);   // phyNGSC.cpp:579
                        l[u] = Tb.qg[(size_t)(K.row0 + u) * 256 + Tb.qcode[q]] & 31u;
                    }
                }
            }
#pragma unroll
            for (uint32_t u = 0; u < 16; u += 2) qb += l[u] + l[u + 1];
            // DNA bits of a Huffman-coded alphabet (more than four symbols, tasks.cpp:525-557)
            if (!plain) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (!__any(nv > 4 * k)) continue;
                    const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                    uint32_t l4 = __builtin_amdgcn_perm(0u, ltab, K.idx[k]) & vm;
                    if (__any((K.ambm >> (4 * k)) & 15u)) {
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            if ((K.ambm >> (4 * k + b)) & 1u) {
                                const uint32_t ll = C.xfer ? 0u : (s_dcode[sr_byte(sw, 4 * k + b)] & 31u);
                                l4 = (l4 & ~(0xffu << (8 * b))) | (ll << (8 * b));
                            }
                        }
                    }
                    db = __builtin_amdgcn_sad_u8(l4, 0u, db);
                }
            }
        }
        const uint32_t wq = sr_wsum(qb), wd = sr_wsum(db);
        if (lane == 0) {
            const size_t tix = (size_t)A.tile_lo[s] + tile;
            A.wq[tix] = wq; A.wd[tix] = wd;
        }
        ck_wave_sync();   // (the record table is rewritten by the next tile)
    }
}

// ---- pack ----
__global__ __launch_bounds__(CK_THREADS) void k_ck_pack(Arena A, uint32_t TR, uint32_t in_words, uint32_t map_cap, uint32_t lds_rows,
                                                        uint32_t sw_q, uint32_t sw_d, uint32_t tpw)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_dyn[];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per_wave = in_words + TR + 2 * map_cap + sw_q + sw_d;
    CkLds M;                             // this wave's tile
    M.in = s_dyn + wave * per_wave;
    M.rec = M.in + in_words;
    M.map = reinterpret_cast<uint2 *>(M.rec + TR);
    uint32_t *st_q = M.rec + TR + 2 * map_cap, *st_d = st_q + sw_q;    // staging of the tile's payload words
    uint32_t *s_qtab = s_dyn + CK_WAVES * per_wave;           // [lds_rows + 16][CK_QS] entries, shared
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    __shared__ uint32_t s_longcode;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + TR - 1) / TR;
    const uint32_t tile_first = blockIdx.x * tpw;
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;

    s_dcode[tid] = A.dcodes[(size_t)s * 256 + tid];
    s_qcode[tid] = sb.qua_code[tid];
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (tid == 0) s_longcode = 0;
    ck_fill_qtab(A, s, sb, s_qtab, lds_rows);
    __syncthreads();
    if ((s_dcode[tid] & 31u) > 8) atomicOr(&s_longcode, 1u);
    __syncthreads();
    const bool short_codes = !s_longcode;   // four base codes merge bytewise when each is at most 8 bits long
    __shared__ uint32_t s_flag;
    const bool lean_wg = ck_lean_alphabet(sb, lds_rows, &s_flag);
    const CkTabs Tb = {A.qcodes + (size_t)s * A.qrows * 256, s_qcode, s_amb, s_dcode};
    const bool plain = (sb.flags & F_DNA_PLAIN) != 0;
    const uint32_t eA = s_dcode['A'], eC = s_dcode['C'], eT = s_dcode['T'], eG = s_dcode['G'];
    const uint32_t ltab = (eA & 31u) | ((eC & 31u) << 8) | ((eT & 31u) << 16) | ((eG & 31u) << 24);
    const uint32_t ctab = ((eA >> 5) & 0xffu) | (((eC >> 5) & 0xffu) << 8) | (((eT >> 5) & 0xffu) << 16) | (((eG >> 5) & 0xffu) << 24);
    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out);
    const uint64_t qbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_hdr) * 8;
    const uint64_t dbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_len + sb.dna_hdr) * 8;

    CkPre P;
    if (tile_first + wave < tile_last) {
        const uint32_t t0r = (tile_first + wave) * TR;
        ck_prefetch(A, rec_lo + t0r, t0r + TR < n ? TR : n - t0r, true, P);
    }
    for (uint32_t tile = tile_first + wave; tile < tile_last; tile += CK_WAVES) {
        const uint32_t t0r = tile * TR;
        const uint32_t nt = t0r + TR < n ? TR : n - t0r;
        for (uint32_t k = lane; k < sw_q + sw_d; k += 64) st_q[k] = 0;
        const CkTile T = ck_commit<false>(A, nt, M, in_words, map_cap, P);
        if (tile + CK_WAVES < tile_last) {   // the next tile's loads fly while this one is worked on
            const uint32_t t1r = (tile + CK_WAVES) * TR;
            ck_prefetch(A, rec_lo + t1r, t1r + TR < n ? TR : n - t1r, true, P);
        }
        if (T.nchunks == 0xffffffffu) { if (lane == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); break; }
        const bool last_tile = tile + 1 == ntiles;
        const size_t tix = (size_t)A.tile_lo[s] + tile;
        const uint64_t qb0 = qbase + A.wq[tix], qb1 = qbase + (last_tile ? sb.qbits : A.wq[tix + 1]);
        const uint64_t db0 = dbase + A.wd[tix], db1 = dbase + (last_tile ? sb.dbits : A.wd[tix + 1]);
        const uint64_t qw0 = qb0 >> 5, dw0 = db0 >> 5;
        const uint32_t qnw = (uint32_t)(((qb1 + 31) >> 5) - qw0), dnw = (uint32_t)(((db1 + 31) >> 5) - dw0);
        const bool q_lds = qnw + 1 <= sw_q, d_lds = dnw + 1 <= sw_d;   // else: straight into the zeroed output
        uint64_t qrun = qb0, drun = db0;

        CkWork W = ck_fetch(M, T, lane);
        for (uint32_t base = 0; base < T.nchunks; base += 64) {
            const CkChunk C = W.C;
            const uint32_t sw[4] = {W.S.x, W.S.y, W.S.z, W.S.w}, qw[4] = {W.Q.x, W.Q.y, W.Q.z, W.Q.w};
            if (base + 64 < T.nchunks) W = ck_fetch(M, T, base + 64 + lane);
            const uint32_t nv = C.nv;
            const bool lean = lean_wg && !__any(!ck_low_bytes(qw));
            CkCls K;
            if (lean) K = ck_classify<false, true>(sw, qw, C, lds_rows);
            else K = ck_classify<false, false>(sw, qw, C, lds_rows);

            // ---- quality (tasks.cpp:609-619): 16 table entries at once, the exceptions patched in, then the four symbols
            // of a dword merged into one code word where they fit ----
            const uint32_t abase = K.rowc * CK_QS - 33u;
            uint32_t e[16];
#pragma unroll
            for (uint32_t u = 0; u < 16; ++u) e[u] = s_qtab[abase + u * CK_QS + sr_byte(K.qwm, u)];
            if (__any(K.qpatch)) {
#pragma unroll
                for (uint32_t u = 0; u < 16; ++u) {
                    if ((K.qpatch >> u) & 1u) {
                        uint32_t q = sr_byte(qw, u);
                        if (C.xfer && ((K.ambm >> u) & 1u)) q = transfer_quality(s_amb[sr_byte(sw, u)], (uint8_t)q);   // phyNGSC.cpp:579
                        e[u] = Tb.qg[(size_t)(K.row0 + u) * 256 + Tb.qcode[q]];
                    }
                }
            }
            uint32_t qc[4], ql[4];          // merged code and total length per dword
            bool qsplit[4];                 // (uniform) some lane's dword needs more than 32 bits: its symbols go one by one
            uint32_t qbits = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t e0 = e[4 * k], e1 = e[4 * k + 1], e2 = e[4 * k + 2], e3 = e[4 * k + 3];
                const uint32_t l1 = e1 & 31u, l2 = e2 & 31u, l3 = e3 & 31u;
                const uint32_t tl = (e0 & 31u) + l1 + l2 + l3;
                ql[k] = tl;
                qbits += tl;
                qsplit[k] = __any(tl > 32);
                qc[k] = ((((((e0 >> 5) << l1) | (e1 >> 5)) << l2) | (e2 >> 5)) << l3) | (e3 >> 5);
            }

            // ---- DNA (tasks.cpp:544-557) ----
            // mode 0: two bits per base, no exceptions in the wave (one put of 2 * nv bits)
            // mode 1: code / length bytes of a dword merged (every code <= 8 bits), exceptions patched in
            // mode 2: symbol by symbol
            const int dmode = (plain && !__any(K.ambm != 0)) ? 0 : (short_codes ? 1 : 2);
            uint32_t dc[4] = {0, 0, 0, 0}, dl[4] = {0, 0, 0, 0}, dbits = 0;
            if (dmode == 0) {
                uint32_t x = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    x = (x << 8) | ((__builtin_amdgcn_perm(0u, ctab, K.idx[k]) * 0x40100401u) >> 24);
                dc[0] = nv == 16 ? x : (nv ? x >> (32 - 2 * nv) : 0u);
                dl[0] = 2 * nv;
                dbits = 2 * nv;
            } else if (dmode == 1) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (!__any(nv > 4 * k)) continue;
                    const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                    uint32_t l4 = __builtin_amdgcn_perm(0u, ltab, K.idx[k]) & vm, c4 = __builtin_amdgcn_perm(0u, ctab, K.idx[k]) & vm;
                    if (__any((K.ambm >> (4 * k)) & 15u)) {
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            if ((K.ambm >> (4 * k + b)) & 1u) {
                                const uint32_t ee = C.xfer ? 0u : s_dcode[sr_byte(sw, 4 * k + b)];
                                l4 = (l4 & ~(0xffu << (8 * b))) | ((ee & 31u) << (8 * b));
                                c4 = (c4 & ~(0xffu << (8 * b))) | ((ee >> 5) << (8 * b));
                            }
                        }
                    }
                    const uint32_t l1 = (l4 >> 8) & 0xffu, l2 = (l4 >> 16) & 0xffu, l3 = l4 >> 24;
                    dc[k] = (((((c4 & 0xffu) << l1) | ((c4 >> 8) & 0xffu)) << l2 | ((c4 >> 16) & 0xffu)) << l3) | (c4 >> 24);
                    dl[k] = (l4 & 0xffu) + l1 + l2 + l3;
                    dbits += dl[k];
                }
            } else {
#pragma unroll
                for (uint32_t u = 0; u < 16; ++u) {
                    if (u < nv) {
                        const uint32_t c = sr_byte(sw, u);
                        if (!(C.xfer && s_amb[c] > 1)) dbits += s_dcode[c] & 31u;
                    }
                }
            }

            // ---- place the chunks of this trip: exclusive wave scan of their bits in stream order ----
            uint32_t qo, dofs, tq, td;
            ck_scan2(qbits, dbits, qo, dofs, tq, td);
            SqWriter wq_, wd_;
            wq_.init(q_lds ? st_q : outw, q_lds ? qw0 : 0, qrun + qo);
            wd_.init(d_lds ? st_d : outw, d_lds ? dw0 : 0, drun + dofs);
            qrun += tq;
            drun += td;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (qsplit[k]) {
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) wq_.put(e[4 * k + b] >> 5, e[4 * k + b] & 31u);
                } else {
                    wq_.put(qc[k], ql[k]);
                }
            }
            if (dmode == 0) {
                if (nv) wd_.put(dc[0], dl[0]);
            } else if (dmode == 1) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (!__any(nv > 4 * k)) continue;
                    wd_.put(dc[k], dl[k]);
                }
            } else {
#pragma unroll
                for (uint32_t u = 0; u < 16; ++u) {
                    if (u < nv) {
                        const uint32_t c = sr_byte(sw, u);
                        if (!(C.xfer && s_amb[c] > 1)) { const uint32_t ee = s_dcode[c]; wd_.put(ee >> 5, ee & 31u); }
                    }
                }
            }
            if (nv) { wq_.finish(); wd_.finish(); }
        }
        ck_wave_sync();
        if (q_lds) {
            for (uint32_t k = lane; k < qnw; k += 64) {
                const uint64_t w = qw0 + k;
                const uint32_t v = st_q[k];
                if ((w << 5) >= qb0 && ((w + 1) << 5) <= qb1) outw[w] = v;
                else if (v) atomicOr(&outw[w], v);
            }
        }
        if (d_lds) {
            for (uint32_t k = lane; k < dnw; k += 64) {
                const uint64_t w = dw0 + k;
                const uint32_t v = st_d[k];
                if ((w << 5) >= db0 && ((w + 1) << 5) <= db1) outw[w] = v;
                else if (v) atomicOr(&outw[w], v);
            }
        }
        ck_wave_sync();   // (staging and record table are rewritten by the next tile)
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
// Records per (wave) tile: as many as keep a tile's bytes within ~9 KB of LDS (0: records too long for this path).
uint32_t chunk_tile_records(uint32_t max_rec_bytes)
{
    const uint32_t forced = getenv("NGSC_CHUNK_TILE_RECORDS") ? (uint32_t)atoi(getenv("NGSC_CHUNK_TILE_RECORDS")) : 0;
    if (max_rec_bytes == 0 || max_rec_bytes > 2048) return 0;
    const uint32_t cap = 1024 * CK_NV - 32;      // CK_NV 16-byte pieces per lane
    uint32_t tr = 64;
    while (tr > 2 && (uint64_t)tr * max_rec_bytes > cap) tr /= 2;
    if (forced >= 2 && forced <= 64 && (forced & (forced - 1)) == 0 && (uint64_t)forced * max_rec_bytes <= cap) tr = forced;
    return tr;
}

struct CkCfg { uint32_t in_words, map_cap, tpw; };
static CkCfg ck_cfg(const Arena &A, uint32_t TR, uint32_t max_rec_bytes, uint32_t max_rec)
{
    CkCfg c;
    const uint32_t span = TR * max_rec_bytes + 32;                  // + alignment of the first byte, + rounding
    c.in_words = ((span + 15) / 16) * 4 + CK_PAD_WORDS;
    c.map_cap = span / 32 + TR + 4;                                 // sum of ceil(L / 16), 2 L < record bytes
    const uint32_t forced = getenv("NGSC_SQ_RECORDS_PER_WG") ? (uint32_t)atoi(getenv("NGSC_SQ_RECORDS_PER_WG")) : 0;
    uint32_t rpw = 1024;                                            // records per workgroup: amortises table fill / histogram flush
    const uint64_t records = (uint64_t)max_rec * A.n_sb;
    while (rpw > TR && records / rpw < 2048) rpw /= 2;              // ... but never fewer than ~8 workgroups per CU
    if (forced) rpw = forced;
    if (rpw < CK_WAVES * TR) rpw = CK_WAVES * TR;
    c.tpw = rpw / TR;
    return c;
}

static void ck_lds_attr(const void *kern, size_t lds)
{
    // (the limit applies to static + dynamic LDS: asking for all 160 KB of dynamic LDS is refused)
    if (lds > (48u << 10)) (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((lds + 1023) & ~(size_t)1023));
}

void launch_ck_stats(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t TR, uint32_t max_rec_bytes)
{
    const CkCfg c = ck_cfg(A, TR, max_rec_bytes, max_rec);
    const uint32_t fixed = CK_WAVES * (c.in_words + TR + 2 * c.map_cap + 2 * TR) + 64;
    uint32_t rows = A.qrows;
    const uint32_t budget_words = (100u << 10) / 4 - fixed;         // rows past the budget count straight into global memory
    if (rows * SR_HSTRIDE > budget_words) rows = budget_words / SR_HSTRIDE;
    const size_t lds = ((size_t)fixed + (size_t)rows * SR_HSTRIDE) * 4;
    const uint32_t ntile = (max_rec + TR - 1) / TR;
    dim3 grid((ntile + c.tpw - 1) / c.tpw, A.n_sb);
    ck_lds_attr(reinterpret_cast<const void *>(k_ck_stats), lds);
    hipLaunchKernelGGL(k_ck_stats, grid, dim3(CK_THREADS), lds, st, A, TR, c.in_words, c.map_cap, rows, c.tpw);
}

void launch_ck_measure(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t TR, uint32_t max_rec_bytes)
{
    const CkCfg c = ck_cfg(A, TR, max_rec_bytes, max_rec);
    const uint32_t fixed = CK_WAVES * (c.in_words + TR + 2 * c.map_cap);
    uint32_t rows = A.qrows;
    const uint32_t budget = (32u << 10) / CK_QS;
    if (rows > budget) rows = budget;
    const size_t lds = (size_t)fixed * 4 + (((size_t)(rows + CK_QROWS_EXTRA) * CK_QS + 15) & ~(size_t)15);
    const uint32_t ntile = (max_rec + TR - 1) / TR;
    dim3 grid((ntile + c.tpw - 1) / c.tpw, A.n_sb);
    ck_lds_attr(reinterpret_cast<const void *>(k_ck_measure), lds);
    hipLaunchKernelGGL(k_ck_measure, grid, dim3(CK_THREADS), lds, st, A, TR, c.in_words, c.map_cap, rows, c.tpw);
}

void launch_ck_pack(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t TR, uint32_t max_rec_bytes, uint32_t sw_q, uint32_t sw_d)
{
    const CkCfg c = ck_cfg(A, TR, max_rec_bytes, max_rec);
    if (sw_q > 2048) sw_q = 2048;
    if (sw_d > 1536) sw_d = 1536;
    const uint32_t fixed = CK_WAVES * (c.in_words + TR + 2 * c.map_cap + sw_q + sw_d);
    uint32_t rows = A.qrows;
    const uint32_t budget = (48u << 10) / (4 * CK_QS);
    if (rows > budget) rows = budget;
    const size_t lds = ((size_t)fixed + (size_t)(rows + CK_QROWS_EXTRA) * CK_QS) * 4;
    const uint32_t ntile = (max_rec + TR - 1) / TR;
    dim3 grid((ntile + c.tpw - 1) / c.tpw, A.n_sb);
    ck_lds_attr(reinterpret_cast<const void *>(k_ck_pack), lds);
    hipLaunchKernelGGL(k_ck_pack, grid, dim3(CK_THREADS), lds, st, A, TR, c.in_words, c.map_cap, rows, sw_q, sw_d, c.tpw);
}

}  // namespace ngsc
