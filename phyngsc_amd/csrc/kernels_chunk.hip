// kernels_chunk.hip - sequence / quality kernels with one lane per 16-BYTE CHUNK of a read, out of an LDS-staged tile.
//
// A tile is a run of TR consecutive records of one sub-block.  Its bytes [ls[i0], ls[i0 + nt]) come into LDS with
// coalesced, aligned 16-byte loads (a wave instruction covers 1 KiB of the file; the lane-per-record kernels of
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

struct CkLds {
    uint32_t *in;    // the tile's bytes
    uint32_t *rec;   // [TR]  seq start (offset in the span) | read length << 16 | transfer flag << 31
    uint32_t *map;   // [map_cap]  work list: record | chunk << 16
};
struct CkTile {
    uint32_t b0;        // job-buffer offset of s_in[0]
    uint32_t nchunks;   // work list length
    uint32_t nfull;     // FULL_FIRST: the first nfull entries are complete 16-byte chunks, the rest are read tails
};

// two exclusive scans over the 256 threads at once; s_part[2 * CK_WAVES]
__device__ __forceinline__ void ck_scan2(uint32_t a, uint32_t b, uint32_t *s_part, uint32_t &ea, uint32_t &eb, uint32_t &ta, uint32_t &tb)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t oa = __shfl_up(ia, d), ob = __shfl_up(ib, d);
        if (lane >= d) { ia += oa; ib += ob; }
    }
    __syncthreads();   // s_part may still be read from the previous call
    if (lane == 63) { s_part[wave] = ia; s_part[CK_WAVES + wave] = ib; }
    __syncthreads();
    uint32_t ba = 0, bb = 0;
    ta = tb = 0;
#pragma unroll
    for (int w = 0; w < CK_WAVES; ++w) {
        const uint32_t pa = s_part[w], pb = s_part[CK_WAVES + w];
        if (w < wave) { ba += pa; bb += pb; }
        ta += pa; tb += pb;
    }
    ea = ba + ia - a;
    eb = bb + ib - b;
}

// Stages the tile, builds its record table and work list.  FULL_FIRST: complete chunks first (record order), read tails
// behind them, so that whole waves take the paths without per-byte masks; otherwise stream order (record, chunk), which
// is the order of the payload bits.  Ends with a barrier.  Returns nchunks == 0xffffffff when the tile does not fit.
template <bool FULL_FIRST>
__device__ __forceinline__ CkTile ck_setup(const Arena &A, uint32_t i0, uint32_t nt, const CkLds &M, uint32_t in_words, uint32_t map_cap,
                                           uint32_t *s_part, bool want_xfer, uint32_t &my_slen)
{
    const uint32_t tid = threadIdx.x;
    CkTile T;
    T.b0 = A.ls[i0] & ~15u;
    const uint32_t b1 = A.ls[i0 + nt];
    uint32_t L = 0, nf = 0, tail = 0;
    my_slen = 0;
    if (tid < nt) {
        const uint32_t te = A.te[i0 + tid], se = A.se[i0 + tid];
        L = se - te - 1;
        if (want_xfer) my_slen = A.slen[i0 + tid];
        M.rec[tid] = (te + 1 - T.b0) | (L << 16) | (my_slen & 0x80000000u);
        nf = L >> 4;
        tail = (L & 15u) ? 1u : 0u;
    }
    uint32_t e0, e1, t0, t1;
    if (FULL_FIRST) ck_scan2(nf, tail, s_part, e0, e1, t0, t1);
    else ck_scan2(nf + tail, 0u, s_part, e0, e1, t0, t1);
    T.nfull = FULL_FIRST ? t0 : 0u;
    T.nchunks = t0 + t1;
    const uint32_t n16 = (b1 - T.b0 + 15) >> 4;
    if (T.nchunks > map_cap || 4 * n16 + CK_PAD_WORDS > in_words || (b1 - T.b0) > 0xffffu) { T.nchunks = 0xffffffffu; return T; }   // (uniform)
    if (tid < nt) {
        for (uint32_t k = 0; k < nf; ++k) M.map[e0 + k] = tid | (k << 16);
        if (tail) M.map[FULL_FIRST ? t0 + e1 : e0 + nf] = tid | (nf << 16);
    }
    const uint4 *src = reinterpret_cast<const uint4 *>(A.buf + T.b0);
    uint4 *dst = reinterpret_cast<uint4 *>(M.in);
    for (uint32_t k = tid; k < n16; k += CK_THREADS) {
        if (T.b0 + 16 * k + 16 <= A.buf_len) dst[k] = src[k];
        else {
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t x = 0; x < 16; ++x) {
                const uint32_t pos = T.b0 + 16 * k + x;
                if (pos < A.buf_len) w[x >> 2] |= (uint32_t)A.buf[pos] << (8 * (x & 3));
            }
            dst[k] = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    if (tid < CK_PAD_WORDS) M.in[4 * n16 + tid] = 0;
    __syncthreads();
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
    const uint32_t m = M.map[C.act ? c : 0];
    C.r = m & 0xffffu;
    C.k = m >> 16;
    const uint32_t rec = M.rec[C.r];
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
    CkLds M;
    M.in = s_dyn;
    M.rec = M.in + in_words;
    M.map = M.rec + TR;
    uint32_t *s_fl = M.map + map_cap;    // [TR] N-transfer test per record: bit0 some transferable code, bit1 a failure
    uint32_t *s_na = s_fl + TR;          // [TR] number of IUPAC codes of the record
    uint32_t *s_hist = s_na + TR;        // [lds_rows][SR_HSTRIDE], then 64 trash words
    __shared__ uint32_t s_occ[256];
    __shared__ uint8_t s_amb[256];
    __shared__ uint32_t s_red[4];
    __shared__ uint32_t s_part[2 * CK_WAVES];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + TR - 1) / TR;
    const uint32_t tile_first = blockIdx.x * tpw;
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;
    const uint32_t tid = threadIdx.x, lane = tid & 63;
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

    for (uint32_t tile = tile_first; tile < tile_last; ++tile) {
        const uint32_t t0r = tile * TR;
        const uint32_t nt = t0r + TR < n ? TR : n - t0r;
        const uint32_t i0 = rec_lo + t0r;
        __syncthreads();   // everyone is done with the previous tile (and, the first time, the tables are cleared)
        if (tid < TR) { s_fl[tid] = 0; s_na[tid] = 0; }
        if (tid == 0) s_red[3] = 0;
        uint32_t dummy;
        const CkTile T = ck_setup<true>(A, i0, nt, M, in_words, map_cap, s_part, false, dummy);
        if (T.nchunks == 0xffffffffu) { if (tid == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
        if (tid < nt) {
            // line 3 of every record must be a bare '+' (README.md:97-103); the reference assumes it
            const uint32_t rec = M.rec[tid], L = (rec >> 16) & 0x7fffu, sq0 = rec & 0xffffu;
            if (ck_byte(M.in, sq0 + L + 1) != '+' || ck_byte(M.in, sq0 + L + 2) != '\n') atomicMin(A.err, (int32_t)ERR_FORMAT);
            if (t0r + tid == 0 && L >= 2) {   // colour-space detection on the sub-block's first record, phyNGSC.cpp:478
                const uint32_t c1 = ck_byte(M.in, sq0 + 1);
                if (c1 >= '0' && c1 <= '3') atomicMin(A.err, (int32_t)ERR_COLORSPACE);
            }
            mq = L > mq ? L : mq;
            sum_l += L;
        }

        // ---- pass 1: everything except the ambiguous (non-ACGT) positions, whose fate depends on the whole record ----
        for (uint32_t base = 0; base < T.nchunks; base += CK_THREADS) {
            const CkChunk C = ck_chunk(M, T, base + tid);
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
                        if (!s_red[3]) s_red[3] = 1;
                    }
                }
            }
        }
        __syncthreads();

        // ---- pass 2: the ambiguous positions, now that every record's transfer decision is known (phyNGSC.cpp:573-599) ----
        if (s_red[3]) {
            for (uint32_t base = 0; base < T.nchunks; base += CK_THREADS) {
                const CkChunk C = ck_chunk(M, T, base + tid);
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
        if (tid < nt) {
            const uint32_t L = (M.rec[tid] >> 16) & 0x7fffu, fl = s_fl[tid];
            const bool xfer = (fl & 1) && !(fl & 2);
            const uint32_t kept = L - (xfer ? s_na[tid] : 0u);
            A.slen[i0 + tid] = kept | ((uint32_t)xfer << 31);
            ms = kept > ms ? kept : ms;
        }
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
// Quality code tables in LDS, indexed by the raw quality byte: column = (q - 33) & 63 of SR_HCOLS columns per row
// (bytes outside 33..80 are caught by sr_q_out and looked up in the global table through qua_code, as are rows past
// the LDS budget and the transferred qualities 128..247).
template <class TE>
__device__ __forceinline__ void ck_fill_qtab(const Arena &A, uint32_t s, const SbState &sb, TE *s_qtab, uint32_t lds_rows)
{
    const uint32_t nq = sb.nqual;
    const uint32_t *qg = A.qcodes + (size_t)s * A.qrows * 256;
    for (uint32_t k = threadIdx.x; k < lds_rows * SR_HCOLS + 64; k += CK_THREADS) s_qtab[k] = 0;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < lds_rows * nq; k += CK_THREADS) {
        const uint32_t row = k / nq, ci = k % nq;
        const uint32_t col = (uint32_t)sb.qualities[ci] - 33u;
        if (col < SR_HCOLS) {
            const uint32_t e = qg[(size_t)row * 256 + ci];
            s_qtab[row * SR_HCOLS + col] = sizeof(TE) == 1 ? (TE)(e & 31u) : (TE)e;
        }
    }
}

struct CkTabs {
    const uint32_t *qg;     // global [qrows][256] by code index
    const uint8_t *qcode;   // LDS [256] quality byte -> code index
    const uint8_t *amb;     // LDS [256]
    const uint32_t *dcode;  // LDS [256] base byte -> code << 5 | len
};

// Exception masks of a chunk (16 bits, one per byte): ambiguous bases, quality bytes outside the LDS columns.
__device__ __forceinline__ void ck_masks(const uint32_t (&sw)[4], const uint32_t (&qw)[4], uint32_t nv, bool need_s, uint32_t (&idx)[4],
                                         uint32_t &ambm, uint32_t &qbm)
{
    uint32_t qbad = 0, dacc = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
        qbad |= sr_q_out(qw[k]) & vm;
        const uint32_t w = (sw[k] & vm) | (0x41414141u & ~vm);
        idx[k] = sr_idx(w);
        dacc |= sr_not_acgt(w, idx[k]);
    }
    ambm = qbm = 0;
    if (__any((need_s && dacc != 0) || qbad != 0)) {
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
            const uint32_t w = (sw[k] & vm) | (0x41414141u & ~vm);
            if (need_s) ambm |= t_movemask(t_nonzero01(sr_not_acgt(w, idx[k]))) << (4 * k);
            qbm |= t_movemask((sr_q_out(qw[k]) >> 7) & 0x01010101u) << (4 * k);
        }
        qbm &= (1u << nv) - 1u;
    }
}

__global__ __launch_bounds__(CK_THREADS) void k_ck_measure(Arena A, uint32_t TR, uint32_t in_words, uint32_t map_cap, uint32_t lds_rows,
                                                           uint32_t tpw)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_dyn[];
    CkLds M;
    M.in = s_dyn;
    M.rec = M.in + in_words;
    M.map = M.rec + TR;
    uint8_t *s_qlen = reinterpret_cast<uint8_t *>(M.map + map_cap);   // [lds_rows][SR_HCOLS] code lengths (+ 64)
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    __shared__ uint32_t s_tot[2];
    __shared__ uint32_t s_part[2 * CK_WAVES];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + TR - 1) / TR;
    const uint32_t tile_first = blockIdx.x * tpw;
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;
    const uint32_t tid = threadIdx.x, lane = tid & 63;

    s_dcode[tid] = A.dcodes[(size_t)s * 256 + tid];
    s_qcode[tid] = sb.qua_code[tid];
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (tid < 2) s_tot[tid] = 0;
    ck_fill_qtab(A, s, sb, s_qlen, lds_rows);
    __syncthreads();
    const CkTabs Tb = {A.qcodes + (size_t)s * A.qrows * 256, s_qcode, s_amb, s_dcode};
    const bool plain = (sb.flags & F_DNA_PLAIN) != 0;
    // code lengths of A, C, T, G (the order of (c >> 1) & 3) as a v_perm table
    const uint32_t ltab = (s_dcode['A'] & 31u) | ((s_dcode['C'] & 31u) << 8) | ((s_dcode['T'] & 31u) << 16) | ((s_dcode['G'] & 31u) << 24);

    for (uint32_t tile = tile_first; tile < tile_last; ++tile) {
        const uint32_t t0r = tile * TR;
        const uint32_t nt = t0r + TR < n ? TR : n - t0r;
        const uint32_t i0 = rec_lo + t0r;
        __syncthreads();
        uint32_t my_slen;
        const CkTile T = ck_setup<false>(A, i0, nt, M, in_words, map_cap, s_part, true, my_slen);
        if (T.nchunks == 0xffffffffu) { if (tid == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
        uint32_t qb = 0;
        uint32_t db = (plain && tid < nt) ? 2u * (my_slen & 0x0fffffffu) : 0u;   // two bits per base that stays (tasks.cpp:544-548)
        for (uint32_t base = 0; base < T.nchunks; base += CK_THREADS) {
            const CkChunk C = ck_chunk(M, T, base + tid);
            const uint32_t nv = C.nv;
            const bool need_s = !plain || __any(C.xfer);   // bases matter for Huffman DNA lengths and for the transferred positions
            const uint4 Q = ck_gather16(M.in, C.s0 + C.L + 3 + 16 * C.k);
            uint4 S = make_uint4(0x41414141u, 0x41414141u, 0x41414141u, 0x41414141u);
            if (need_s) S = ck_gather16(M.in, C.s0 + 16 * C.k);
            const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
            uint32_t idx[4], ambm, qbm;
            ck_masks(sw, qw, nv, need_s, idx, ambm, qbm);
            const uint32_t validm = (1u << nv) - 1u;
            const uint32_t row0 = 16 * C.k + 1;
            const uint32_t rows_left = lds_rows > row0 ? lds_rows - row0 : 0;
            const uint32_t rowm = rows_left >= 16 ? 0xffffu : ((1u << rows_left) - 1u);
            const uint32_t qpatch = (qbm | (C.xfer ? ambm : 0u) | (validm & ~rowm));
            const uint32_t rbase = (rows_left ? row0 : 0u) * SR_HCOLS;   // a lane without table rows reads row 0 and patches everything
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (!__any(nv > 4 * k)) continue;
                uint32_t l[4];
#pragma unroll
                for (uint32_t b = 0; b < 4; ++b) {
                    const uint32_t u = 4 * k + b;
                    const uint32_t ro = u < rows_left ? u : 0u;
                    l[b] = s_qlen[rbase + ro * SR_HCOLS + ((sr_byte(qw, u) - 33u) & 63u)];
                }
                if (__any((qpatch >> (4 * k)) & 15u)) {
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) {
                        const uint32_t u = 4 * k + b;
                        if ((qpatch >> u) & 1u) {
                            uint32_t q = sr_byte(qw, u);
                            if (C.xfer && ((ambm >> u) & 1u)) q = transfer_quality(s_amb[sr_byte(sw, u)], (uint8_t)q);   // phyNGSC.cpp:579
                            l[b] = Tb.qg[(size_t)(row0 + u) * 256 + Tb.qcode[q]] & 31u;
                        }
                    }
                }
#pragma unroll
                for (uint32_t b = 0; b < 4; ++b) qb += ((validm >> (4 * k + b)) & 1u) ? l[b] : 0u;
            }
            // DNA bits of a Huffman-coded alphabet (more than four symbols, tasks.cpp:525-557)
            if (!plain) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (!__any(nv > 4 * k)) continue;
                    const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                    uint32_t l4 = __builtin_amdgcn_perm(0u, ltab, idx[k]) & vm;
                    if (__any((ambm >> (4 * k)) & 15u)) {
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            if ((ambm >> (4 * k + b)) & 1u) {
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
        if (lane == 0) { atomicAdd(&s_tot[0], wq); atomicAdd(&s_tot[1], wd); }
        __syncthreads();
        if (tid == 0) {
            const size_t tix = (size_t)A.tile_lo[s] + tile;
            A.wq[tix] = s_tot[0]; A.wd[tix] = s_tot[1];
            s_tot[0] = 0; s_tot[1] = 0;
        }
    }
}

// ---- pack ----
__global__ __launch_bounds__(CK_THREADS) void k_ck_pack(Arena A, uint32_t TR, uint32_t in_words, uint32_t map_cap, uint32_t lds_rows,
                                                        uint32_t sw_q, uint32_t sw_d, uint32_t tpw)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_dyn[];
    CkLds M;
    M.in = s_dyn;
    M.rec = M.in + in_words;
    M.map = M.rec + TR;
    uint32_t *s_qtab = M.map + map_cap;                       // [lds_rows][SR_HCOLS] entries (+ 64)
    uint32_t *st_q = s_qtab + lds_rows * SR_HCOLS + 64, *st_d = st_q + sw_q;   // staging of the tile's payload words
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    __shared__ uint32_t s_part[2 * CK_WAVES];
    __shared__ uint32_t s_longcode;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + TR - 1) / TR;
    const uint32_t tile_first = blockIdx.x * tpw;
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;
    const uint32_t tid = threadIdx.x;

    s_dcode[tid] = A.dcodes[(size_t)s * 256 + tid];
    s_qcode[tid] = sb.qua_code[tid];
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (tid == 0) s_longcode = 0;
    ck_fill_qtab(A, s, sb, s_qtab, lds_rows);
    __syncthreads();
    if ((s_dcode[tid] & 31u) > 8) atomicOr(&s_longcode, 1u);
    __syncthreads();
    const bool short_codes = !s_longcode;   // four base codes merge bytewise when each is at most 8 bits long
    const CkTabs Tb = {A.qcodes + (size_t)s * A.qrows * 256, s_qcode, s_amb, s_dcode};
    const bool plain = (sb.flags & F_DNA_PLAIN) != 0;
    const uint32_t eA = s_dcode['A'], eC = s_dcode['C'], eT = s_dcode['T'], eG = s_dcode['G'];
    const uint32_t ltab = (eA & 31u) | ((eC & 31u) << 8) | ((eT & 31u) << 16) | ((eG & 31u) << 24);
    const uint32_t ctab = ((eA >> 5) & 0xffu) | (((eC >> 5) & 0xffu) << 8) | (((eT >> 5) & 0xffu) << 16) | (((eG >> 5) & 0xffu) << 24);
    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out);
    const uint64_t qbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_hdr) * 8;
    const uint64_t dbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_len + sb.dna_hdr) * 8;

    for (uint32_t tile = tile_first; tile < tile_last; ++tile) {
        const uint32_t t0r = tile * TR;
        const uint32_t nt = t0r + TR < n ? TR : n - t0r;
        const uint32_t i0 = rec_lo + t0r;
        __syncthreads();   // the previous tile's copy-out is done
        for (uint32_t k = tid; k < sw_q + sw_d; k += CK_THREADS) st_q[k] = 0;
        uint32_t my_slen;
        const CkTile T = ck_setup<false>(A, i0, nt, M, in_words, map_cap, s_part, true, my_slen);
        if (T.nchunks == 0xffffffffu) { if (tid == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
        const bool last_tile = tile + 1 == ntiles;
        const size_t tix = (size_t)A.tile_lo[s] + tile;
        const uint64_t qb0 = qbase + A.wq[tix], qb1 = qbase + (last_tile ? sb.qbits : A.wq[tix + 1]);
        const uint64_t db0 = dbase + A.wd[tix], db1 = dbase + (last_tile ? sb.dbits : A.wd[tix + 1]);
        const uint64_t qw0 = qb0 >> 5, dw0 = db0 >> 5;
        const uint32_t qnw = (uint32_t)(((qb1 + 31) >> 5) - qw0), dnw = (uint32_t)(((db1 + 31) >> 5) - dw0);
        const bool q_lds = qnw + 1 <= sw_q, d_lds = dnw + 1 <= sw_d;   // else: straight into the zeroed output
        uint64_t qrun = qb0, drun = db0;

        for (uint32_t base = 0; base < T.nchunks; base += CK_THREADS) {
            const CkChunk C = ck_chunk(M, T, base + tid);
            const uint32_t nv = C.nv;
            const uint4 S = ck_gather16(M.in, C.s0 + 16 * C.k), Q = ck_gather16(M.in, C.s0 + C.L + 3 + 16 * C.k);
            const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
            uint32_t idx[4], ambm, qbm;
            ck_masks(sw, qw, nv, true, idx, ambm, qbm);
            const uint32_t validm = (1u << nv) - 1u;
            ambm &= validm;
            const uint32_t row0 = 16 * C.k + 1;
            const uint32_t rows_left = lds_rows > row0 ? lds_rows - row0 : 0;
            const uint32_t rowm = rows_left >= 16 ? 0xffffu : ((1u << rows_left) - 1u);
            // quality bytes whose entry is not in the LDS table: odd bytes, transferred ones, rows past the table
            const uint32_t qpatch = (qbm | (C.xfer ? ambm : 0u) | (validm & ~rowm));
            const uint32_t rbase = (rows_left ? row0 : 0u) * SR_HCOLS;

            // ---- quality (tasks.cpp:609-619): the four symbols of a dword merged into one code word where they fit ----
            uint32_t qc[4], ql[4];          // merged code and total length per dword (<= 32 bits), or ...
            uint32_t qe[16];                // ... the single entries where some lane's dword needs more than 32 bits
            bool qsplit[4];
            uint32_t qbits = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                qc[k] = ql[k] = 0;
                qsplit[k] = false;
#pragma unroll
                for (uint32_t b = 0; b < 4; ++b) qe[4 * k + b] = 0;
                if (!__any(nv > 4 * k)) continue;
                uint32_t e[4];
#pragma unroll
                for (uint32_t b = 0; b < 4; ++b) {
                    const uint32_t u = 4 * k + b;
                    const uint32_t ro = u < rows_left ? u : 0u;
                    e[b] = s_qtab[rbase + ro * SR_HCOLS + ((sr_byte(qw, u) - 33u) & 63u)];
                }
                if (__any((qpatch >> (4 * k)) & 15u)) {
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) {
                        const uint32_t u = 4 * k + b;
                        if ((qpatch >> u) & 1u) {
                            uint32_t q = sr_byte(qw, u);
                            if (C.xfer && ((ambm >> u) & 1u)) q = transfer_quality(s_amb[sr_byte(sw, u)], (uint8_t)q);   // phyNGSC.cpp:579
                            e[b] = Tb.qg[(size_t)(row0 + u) * 256 + Tb.qcode[q]];
                        }
                    }
                }
                if (__any(nv < 4 * k + 4)) {   // read tails: bytes past the read contribute no bits
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) e[b] = ((validm >> (4 * k + b)) & 1u) ? e[b] : 0u;
                }
                const uint32_t l1 = e[1] & 31u, l2 = e[2] & 31u, l3 = e[3] & 31u;
                const uint32_t tl = (e[0] & 31u) + l1 + l2 + l3;
                ql[k] = tl;
                qbits += tl;
                if (__any(tl > 32)) {
                    qsplit[k] = true;
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) qe[4 * k + b] = e[b];
                } else {
                    qc[k] = ((((((e[0] >> 5) << l1) | (e[1] >> 5)) << l2) | (e[2] >> 5)) << l3) | (e[3] >> 5);
                }
            }

            // ---- DNA (tasks.cpp:544-557) ----
            // mode 0: two bits per base, no exceptions in the wave (one put of 2 * nv bits)
            // mode 1: code / length bytes of a dword merged (every code <= 8 bits), exceptions patched in
            // mode 2: symbol by symbol
            const int dmode = (plain && !__any(ambm != 0)) ? 0 : (short_codes ? 1 : 2);
            uint32_t dc[4] = {0, 0, 0, 0}, dl[4] = {0, 0, 0, 0}, dbits = 0;
            if (dmode == 0) {
                uint32_t x = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    x = (x << 8) | ((__builtin_amdgcn_perm(0u, ctab, idx[k]) * 0x40100401u) >> 24);
                dc[0] = nv == 16 ? x : (nv ? x >> (32 - 2 * nv) : 0u);
                dl[0] = 2 * nv;
                dbits = 2 * nv;
            } else if (dmode == 1) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (!__any(nv > 4 * k)) continue;
                    const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                    uint32_t l4 = __builtin_amdgcn_perm(0u, ltab, idx[k]) & vm, c4 = __builtin_amdgcn_perm(0u, ctab, idx[k]) & vm;
                    if (__any((ambm >> (4 * k)) & 15u)) {
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            if ((ambm >> (4 * k + b)) & 1u) {
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

            // ---- place the chunks of this trip: exclusive scan of their bits in stream order ----
            uint32_t qo, dofs, tq, td;
            ck_scan2(qbits, dbits, s_part, qo, dofs, tq, td);
            SqWriter wq_, wd_;
            wq_.init(q_lds ? st_q : outw, q_lds ? qw0 : 0, qrun + qo);
            wd_.init(d_lds ? st_d : outw, d_lds ? dw0 : 0, drun + dofs);
            qrun += tq;
            drun += td;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                if (!__any(nv > 4 * k)) continue;
                if (qsplit[k]) {
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) wq_.put(qe[4 * k + b] >> 5, qe[4 * k + b] & 31u);
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
                        if (!(C.xfer && s_amb[c] > 1)) { const uint32_t e = s_dcode[c]; wd_.put(e >> 5, e & 31u); }
                    }
                }
            }
            if (nv) { wq_.finish(); wd_.finish(); }
        }
        __syncthreads();
        if (q_lds) {
            for (uint32_t k = tid; k < qnw; k += CK_THREADS) {
                const uint64_t w = qw0 + k;
                const uint32_t v = st_q[k];
                if ((w << 5) >= qb0 && ((w + 1) << 5) <= qb1) outw[w] = v;
                else if (v) atomicOr(&outw[w], v);
            }
        }
        if (d_lds) {
            for (uint32_t k = tid; k < dnw; k += CK_THREADS) {
                const uint64_t w = dw0 + k;
                const uint32_t v = st_d[k];
                if ((w << 5) >= db0 && ((w + 1) << 5) <= db1) outw[w] = v;
                else if (v) atomicOr(&outw[w], v);
            }
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
// Records per tile: as many as keep the tile's bytes within ~40 KB of LDS (0: records too long for this path).
uint32_t chunk_tile_records(uint32_t max_rec_bytes)
{
    const uint32_t forced = getenv("NGSC_CHUNK_TILE_RECORDS") ? (uint32_t)atoi(getenv("NGSC_CHUNK_TILE_RECORDS")) : 0;
    if (max_rec_bytes == 0 || max_rec_bytes > 2048) return 0;
    uint32_t tr = 256;
    while (tr > 16 && (uint64_t)tr * max_rec_bytes > (40u << 10)) tr /= 2;
    if (forced >= 16 && forced <= 256 && (forced & (forced - 1)) == 0 && (uint64_t)forced * max_rec_bytes <= (60u << 10)) tr = forced;
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
    if (rpw < TR) rpw = TR;
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
    const uint32_t fixed = c.in_words + TR + c.map_cap + 2 * TR + 64;
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
    const uint32_t fixed = c.in_words + TR + c.map_cap;
    uint32_t rows = A.qrows;
    const uint32_t budget = (32u << 10) / SR_HCOLS;
    if (rows > budget) rows = budget;
    const size_t lds = (size_t)fixed * 4 + (size_t)rows * SR_HCOLS + 64;
    const uint32_t ntile = (max_rec + TR - 1) / TR;
    dim3 grid((ntile + c.tpw - 1) / c.tpw, A.n_sb);
    ck_lds_attr(reinterpret_cast<const void *>(k_ck_measure), lds);
    hipLaunchKernelGGL(k_ck_measure, grid, dim3(CK_THREADS), lds, st, A, TR, c.in_words, c.map_cap, rows, c.tpw);
}

void launch_ck_pack(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t TR, uint32_t max_rec_bytes, uint32_t sw_q, uint32_t sw_d)
{
    const CkCfg c = ck_cfg(A, TR, max_rec_bytes, max_rec);
    if (sw_q > 8192) sw_q = 8192;
    if (sw_d > 6144) sw_d = 6144;
    const uint32_t fixed = c.in_words + TR + c.map_cap + 64 + sw_q + sw_d;
    uint32_t rows = A.qrows;
    const uint32_t budget = (48u << 10) / (4 * SR_HCOLS);
    if (rows > budget) rows = budget;
    const size_t lds = ((size_t)fixed + (size_t)rows * SR_HCOLS) * 4;
    const uint32_t ntile = (max_rec + TR - 1) / TR;
    dim3 grid((ntile + c.tpw - 1) / c.tpw, A.n_sb);
    ck_lds_attr(reinterpret_cast<const void *>(k_ck_pack), lds);
    hipLaunchKernelGGL(k_ck_pack, grid, dim3(CK_THREADS), lds, st, A, TR, c.in_words, c.map_cap, rows, sw_q, sw_d, c.tpw);
}

}  // namespace ngsc
