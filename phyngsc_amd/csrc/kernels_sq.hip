// kernels_sq.hip - register-streaming sequence / quality kernels: one lane per record, the record's bases and
// qualities read straight from global memory in unaligned 16-byte chunks (neighbouring lanes read neighbouring
// records, so every cache line that comes in is consumed by the wave that asked for it) and processed four
// bytes per instruction where the format allows it.  No LDS tile, no barriers inside the record loop.
//
//   k_sq_stats_reg    N-transfer test, DNA symbol counts, per-position quality histogram   (phyNGSC.cpp:462-653,
//                                                                                          tasks.cpp:260-286)
//   k_sq_measure_reg  quality / DNA payload bits per record and per 256-record tile
//   k_sq_pack_reg     StoreQuality and StoreDNA payloads (tasks.cpp:544-557, 609-619): every lane merges up to
//                     four codes per put, ORs whole words into LDS staging; one coalesced copy-out per stream
//
// The LDS-tiled kernels of round 1 (removed in round 3) spent ~23 VALU instructions per input byte and wave on per-byte
// index arithmetic; here the common case (all lanes inside their read, only A/C/G/T, qualities '!'..'`') costs
// ~3: bases are classified with one v_perm_b32 per dword and counted with popcounts of two bit planes, a
// quality byte costs a bit-field extract, a shift-add and one ds_add_u32 with an immediate row offset.
// Everything else (read tails, ambiguity codes, odd quality bytes, rows past the LDS histogram) takes a
// masked general path chosen per 16-byte chunk with a wave-uniform branch.
#include "ngsc_kernels.h"
#include <cstring>
#include "ngsc_swar.h"
#include "ngsc_sq.h"

namespace ngsc {

// one 16-byte chunk of a record's bases or qualities: from the chunk-major copy the statistics kernel leaves (Arena::dstride) or the file
struct __attribute__((packed, aligned(4))) U3a { uint32_t x, y, z; };
template <bool NT = false>
__device__ __forceinline__ uint4 sq_chunk_load(const Arena &A, const uint8_t *dense, uint32_t g, uint32_t off, uint32_t file_pos)
{
    if (A.dstride) {
        const uint32_t c = off >> 4;
        if (c == A.dtail_c) {   // the short last chunk (Arena::dtail_b bytes per record; the bytes behind them read as zero)
            const uint8_t *p = dense + (size_t)c * A.dstride * 16 + (size_t)g * A.dtail_b;
            if (A.dtail_b == 4) return make_uint4(*reinterpret_cast<const uint32_t *>(p), 0, 0, 0);
            if (A.dtail_b == 8) { const uint2 v = *reinterpret_cast<const uint2 *>(p); return make_uint4(v.x, v.y, 0, 0); }
            const U3a v = *reinterpret_cast<const U3a *>(p);
            return make_uint4(v.x, v.y, v.z, 0);
        }
        if (NT) return ld16_nt(dense + ((size_t)c * A.dstride + g) * 16);   // (the copies' last reader: ngsc_swar.h)
        return *reinterpret_cast<const uint4 *>(dense + ((size_t)c * A.dstride + g) * 16);
    }
    return load16_unaligned(A.buf, file_pos + off, A.buf_len);
}
// aligned 16 bytes at a (a multiple of 16), zero behind the end of the job buffer
__device__ __forceinline__ uint4 load16_guarded_sq(const uint8_t *buf, uint32_t a, uint32_t buf_len)
{
    if (a + 16 <= buf_len) return *reinterpret_cast<const uint4 *>(buf + a);
    uint32_t w[4] = {0, 0, 0, 0};
    for (uint32_t x = 0; x < 16; ++x)
        if (a + x < buf_len) w[x >> 2] |= (uint32_t)buf[a + x] << (8 * (x & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}
__device__ __forceinline__ void sq_chunk_keep(const Arena &A, uint32_t g, uint32_t off, const uint4 &S, const uint4 &Q)
{
    if (A.dstride) {
        const uint32_t c = off >> 4;
        if (c == A.dtail_c) {
            uint8_t *ps = A.dense_s + (size_t)c * A.dstride * 16 + (size_t)g * A.dtail_b;
            uint8_t *pq = A.dense_q + (size_t)c * A.dstride * 16 + (size_t)g * A.dtail_b;
            if (A.dtail_b == 4) { *reinterpret_cast<uint32_t *>(ps) = S.x; *reinterpret_cast<uint32_t *>(pq) = Q.x; }
            else if (A.dtail_b == 8) { *reinterpret_cast<uint2 *>(ps) = make_uint2(S.x, S.y); *reinterpret_cast<uint2 *>(pq) = make_uint2(Q.x, Q.y); }
            else { *reinterpret_cast<U3a *>(ps) = U3a{S.x, S.y, S.z}; *reinterpret_cast<U3a *>(pq) = U3a{Q.x, Q.y, Q.z}; }
            return;
        }
        // (plain stores: with the non-temporal hint the kernel is no faster and the readers lose what the caches kept)
        *reinterpret_cast<uint4 *>(A.dense_s + ((size_t)c * A.dstride + g) * 16) = S;
        *reinterpret_cast<uint4 *>(A.dense_q + ((size_t)c * A.dstride + g) * 16) = Q;
    }
}

// STAGED (short records, round 3): the 64 records of a wave are one contiguous span of the file (~8 KB at 36 bases).  The wave
// fetches that span with coalesced 16-byte loads - eight rows of 1 KB instead of six loads per lane at addresses a whole
// record apart, which the texture path prices at ~160 cycles each (tools/ubench_loads.hip: unaligned scattered loads cost twice
// the aligned ones and keep the kernel off the HBM roofline) - parks it in LDS, and every lane reads its own record's chunks
// back with unaligned 16-byte LDS reads.  Same bytes, same logic behind the loads.  span_cap: bytes of LDS per wave,
// >= 64 x the job's longest record + 48.
// (THREADS: 256; 512 for the staged form - eight spans beside one histogram, see launch_sq_stats_reg)
template <bool BURST, bool STAGED, uint32_t THREADS = SR_THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(4))) void k_sq_stats_reg(Arena A, uint32_t lds_rows, uint32_t rpw, uint32_t span_cap)
{
    static_assert(!(BURST && STAGED), "the staged form is for short records");
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_hist[];   // [lds_rows][SR_HSTRIDE], then 64 trash words, then (STAGED) the waves' spans
    __shared__ uint32_t s_occ[256];
    __shared__ uint8_t s_amb[256];
    __shared__ uint32_t s_red[3];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t first = blockIdx.x * rpw;
    if (first >= n) return;
    const uint32_t last = first + rpw < n ? first + rpw : n;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t hwords = lds_rows * SR_HSTRIDE;
    uint8_t *const span = reinterpret_cast<uint8_t *>(s_hist + ((hwords + 64 + 3) & ~3u)) + (size_t)(tid >> 6) * span_cap;
    uint32_t span_lo = 0;   // file offset of span[0] (wave-uniform)

    for (uint32_t k = tid; k < hwords; k += THREADS) s_hist[k] = 0;
    if (tid < 256) { s_occ[tid] = 0; s_amb[tid] = (uint8_t)amb_code((uint8_t)tid); }
    if (tid < 3) s_red[tid] = 0;
    __syncthreads();

    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    uint8_t *hb = reinterpret_cast<uint8_t *>(s_hist);
    const uint32_t trash = (hwords + lane) * 4;   // 64 words after the histogram
    uint32_t t0 = 0, t1 = 0, t01 = 0, nok = 0, mq = 0, ms = 0, sum_l = 0;   // per-lane totals over this workgroup's records
    const uint32_t lj = lane & 3;                                            // rotation of the quality bytes: word ...
    uint32_t bsh[4], boff[4];                                                // ... and byte inside the word
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t bb = (b + (lane >> 2)) & 3;
        bsh[b] = 8 * bb;
        boff[b] = bb * SR_HSTRIDE * 4;
    }

    uint32_t te_n = 0, se_n = 0;   // the next batch's record table entries are fetched one batch ahead
    if (first + tid < last) { te_n = A.te[rec_lo + first + tid]; se_n = A.se[rec_lo + first + tid]; }
    for (uint32_t base = first; base < last; base += THREADS) {
        const uint32_t i = base + tid;
        const bool valid = i < last;
        const uint32_t g = rec_lo + (valid ? i : first);
        const uint32_t te = te_n, se = se_n;
        if (i + THREADS < last) { te_n = A.te[g + THREADS]; se_n = A.se[g + THREADS]; }
        const uint32_t L = valid ? se - te - 1 : 0;
        const uint32_t s0 = te + 1, q0 = se + 3;
        if (valid) {
            // line 3 of every record must be a bare '+' (README.md:97-103); the reference assumes it.  Checked below from
            // the bytes that follow the read inside its last 16-byte chunk; an empty read has no chunk:
            if (L == 0 && (load8_unaligned(A.buf, se, A.buf_len).x & 0x00ffff00u) != 0x000a2b00u) atomicMin(A.err, (int32_t)ERR_FORMAT);
            if (i == 0 && L >= 1 && !A.cs_mode) {   // colour-space detection on the sub-block's first record, phyNGSC.cpp:478
                if (cs_first_read(A.buf[s0], A.buf[s0 + 1])) atomicMin(A.err, (int32_t)ERR_COLORSPACE);   // (the job runs again, decoded)
            }
        }

        if constexpr (STAGED) {
            // the wave's records [s0 of its first lane, end of its last valid lane's qualities) -> LDS, 1 KB per row
            const unsigned long long vm = __ballot(valid);
            if (vm) {   // (valid lanes are a prefix of the wave)
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)s0) & ~15u;
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(q0 + L), (int)__builtin_popcountll(vm) - 1);
                const uint32_t rows = (hi - lo + 1023u) >> 10;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the previous batch's reads are done
                for (uint32_t r0 = 0; r0 < rows; r0 += 4) {
                    uint4 v[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k)
                        if (r0 + k < rows) v[k] = load16_guarded_sq(A.buf, lo + 1024u * (r0 + k) + 16u * lane, A.buf_len);
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k)
                        if (r0 + k < rows) *reinterpret_cast<uint4 *>(span + 1024u * (r0 + k) + 16u * lane) = v[k];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                span_lo = lo;
            }
        }
        // one 16-byte chunk at file offset pos: from the staged span or from the file
        auto ld16 = [&](uint32_t pos) __attribute__((always_inline)) -> uint4 {
            if constexpr (STAGED) {
                // five ALIGNED dwords and four funnel shifts: a 16-byte LDS read off its natural alignment - a lane's chunk starts
                // at any byte - is replayed at 64 cycles per wave instruction (MI355X_MICROARCH.md, LDS); dword reads a record
                // (an odd number of dwords, usually) apart are conflict-free
                const uint32_t o = pos - span_lo, sh = (o & 3u) * 8u;
                const uint32_t *q = reinterpret_cast<const uint32_t *>(span + (o & ~3u));
                const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                return make_uint4(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh), __builtin_amdgcn_alignbit(d3, d2, sh),
                                  __builtin_amdgcn_alignbit(d4, d3, sh));
            } else {
                return load16_unaligned(A.buf, pos, A.buf_len);
            }
        };

        // ---- one pass over the 16-byte chunks: everything except the ambiguous positions ----
        const uint32_t b_t0 = t0, b_t1 = t1, b_t01 = t01, b_nok = nok;   // (the record's own base counts = what it adds to these)
        uint32_t fl = 0, namb = 0;   // N-transfer test (phyNGSC.cpp:549-571): bit0 some ambiguity code, bit1 a failure
        uint32_t nall = 0, apos0 = 0, apos1 = 0;   // non-ACGT bytes of the record and the positions of the first four
        // NBX chunks of a record per trip.  BURST: NB = 4 chunks (64 bytes) are fetched back to back and each is used as it
        // arrives - lanes are a whole record apart, so a cache line serves one lane and has to stay in L1 until that lane
        // has read all of it (with one chunk per trip the lines of 150-base records were fetched again and again).
        // Otherwise one chunk per trip, the next one loaded while the current one is processed (better for short reads).
        constexpr uint32_t NBX = BURST ? NB : 1u;
        uint4 Sn = make_uint4(0, 0, 0, 0), Qn = Sn;
        if (!BURST && L) {
            Sn = ld16(s0);
            Qn = ld16(q0);
        }
        for (uint32_t grp = 0; __any(16 * NBX * grp < L); ++grp) {
            uint4 Sb[NBX], Qb[NBX];
            if constexpr (BURST) {
#pragma unroll
                for (uint32_t k = 0; k < NBX; ++k) {
                    const uint32_t off = 16 * NBX * grp + 16 * k;
                    Sb[k] = Qb[k] = make_uint4(0, 0, 0, 0);
                    if (L > off) {
                        Sb[k] = load16_unaligned(A.buf, s0 + off, A.buf_len);
                        Qb[k] = load16_unaligned(A.buf, q0 + off, A.buf_len);
                    }
                }
                // (the copies for measure / pack: behind all four loads, so that the loads stay in flight together)
#pragma unroll
                for (uint32_t k = 0; k < NBX; ++k) {
                    const uint32_t off = 16 * NBX * grp + 16 * k;
                    if (L > off) sq_chunk_keep(A, g, off, Sb[k], Qb[k]);
                }
            } else {
                Sb[0] = Sn; Qb[0] = Qn;
                Sn = Qn = make_uint4(0, 0, 0, 0);
                if (L > 16 * grp + 16) {
                    Sn = ld16(s0 + 16 * grp + 16);
                    Qn = ld16(q0 + 16 * grp + 16);
                }
                if (L > 16 * grp) sq_chunk_keep(A, g, 16 * grp, Sb[0], Qb[0]);   // (the chunk that has arrived, not the one in flight)
            }
#pragma unroll
            for (uint32_t kb = 0; kb < NBX; ++kb) {
                if (kb == 0 || __any(16 * (NBX * grp + kb) < L)) {
                    const uint32_t ch = NBX * grp + kb;
                    const uint4 S = Sb[kb], Q = Qb[kb];
                    const uint32_t nv = L > 16 * ch ? (L - 16 * ch < 16 ? L - 16 * ch : 16) : 0;
                    const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
                    if (__any(nv != 0 && L <= 16 * ch + 16)) {   // some lane is at the end of its read: "\n+\n" follows
                        if (nv != 0 && L <= 16 * ch + 16) {
                            uint32_t two;   // the two bytes after the read's newline
                            if (nv <= 13) {
                                const uint32_t o = nv + 1, wi = o >> 2;
                                const uint32_t lo = wi == 0 ? sw[0] : (wi == 1 ? sw[1] : (wi == 2 ? sw[2] : sw[3]));
                                const uint32_t hi = wi == 0 ? sw[1] : (wi == 1 ? sw[2] : sw[3]);
                                two = __builtin_amdgcn_alignbit(hi, lo, (o & 3) * 8) & 0xffffu;
                            } else {
                                two = (load4_unaligned(A.buf, se) >> 8) & 0xffffu;
                            }
                            if (two != 0x0a2bu) atomicMin(A.err, (int32_t)ERR_FORMAT);
                        }
                    }
                    uint32_t d[4] = {0, 0, 0, 0}, idx[4] = {0, 0, 0, 0}, dacc = 0, qbad = 0, c0 = 0, c1 = 0, c01 = 0;
                    const bool wany[4] = {true, (bool)__any(nv > 4), (bool)__any(nv > 8), (bool)__any(nv > 12)};   // words any lane still needs
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
                    const bool rows_ok = 16 * ch + 17 <= lds_rows;
                    const bool odd = dacc != 0 || qbad != 0 || (nv != 0 && nv != 16);
                    const unsigned long long live = __ballot(nv != 0);
                    const uint32_t nv_first = live ? (uint32_t)__builtin_amdgcn_readlane((int)nv, __builtin_ctzll(live)) : 0u;
                    if (rows_ok && !__any(odd)) {
                        if (nv) {
                            // tasks.cpp:268-283: position histogram rows 1..max_q.  All lanes are at the same 16 positions, so
                            // each lane walks them in its own rotated order (word lj, byte lb): 16 different rows per
                            // instruction instead of one, or the LDS atomics serialise on the popular quality values.
                            const uint32_t rb = (16 * ch + 1) * SR_HSTRIDE * 4 - 33 * 4;
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
                    } else if (16 * ch + nv_first + 1 <= lds_rows && !__any(dacc != 0 || qbad != 0) && !__any(nv != 0 && nv != nv_first)) {
                        // the read tails of equally long reads: the same nv < 16 bytes in every lane, nothing odd about them
                        if (nv) {
                            const uint32_t rb = (16 * ch + 1) * SR_HSTRIDE * 4 - 33 * 4;
                            for (uint32_t u = 0; u < nv_first; ++u) {
                                const uint32_t w = u < 4 ? qw[0] : (u < 8 ? qw[1] : (u < 12 ? qw[2] : qw[3]));
                                const uint32_t q = (w >> (8 * (u & 3))) & 0xffu;
                                atomicAdd(reinterpret_cast<uint32_t *>(hb + rb + u * SR_HSTRIDE * 4 + 4 * q), 1u);
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
                        const uint32_t rows_left = lds_rows > 16 * ch + 1 ? lds_rows - (16 * ch + 1) : 0;   // rows 16ch+1.. inside LDS
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
                        const uint32_t rb = (16 * ch + 1) * SR_HSTRIDE * 4 - 33 * 4;
                        if (__any(okm)) {   // branch-free: bytes that do not count go to the lane's own trash word
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k) {
                                const uint32_t j = (k + lj) & 3;
                                const uint32_t wsel = j == 0 ? qw[0] : (j == 1 ? qw[1] : (j == 2 ? qw[2] : qw[3]));
                                const uint32_t joff = rb + j * (4 * SR_HSTRIDE * 4);
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
                                if ((glm >> u) & 1u) atomicAdd(&qh[(size_t)(16 * ch + u + 1) * 256 + sr_byte(qw, u)], 1u);
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {   // only the dwords in which some lane holds an ambiguity code
                            if (!__any((ambm >> (4 * k)) & 15u)) continue;
#pragma unroll
                            for (uint32_t b = 0; b < 4; ++b) {
                                const uint32_t u = 4 * k + b;
                                if ((ambm >> u) & 1u) {
                                    const uint32_t a = s_amb[sr_byte(sw, u)];
                                    if (a == 0) fl |= 2;
                                    else {
                                        const uint32_t q = sr_byte(qw, u);
                                        fl |= (q < 33 || q > 40) ? 2u : 1u;
                                        ++namb;
                                    }
                                    // remember where it was (up to four per record) for the pass below
                                    const uint32_t p16 = 16 * ch + u;
                                    const uint32_t pv = p16 << (16 * (nall & 1u));   // (no choice between two variables: inside the
                                    apos0 |= nall < 2 ? pv : 0u;                      //  lambda that would put both into scratch)
                                    apos1 |= (nall >= 2 && nall < 4) ? pv : 0u;
                                    ++nall;
                                }
                            }
                        }
                    }
                }
            }
        }
        const bool xfer = (fl & 1) && !(fl & 2);
        const uint32_t kept = L - (xfer ? namb : 0);
        if (valid) {
            A.slen[g] = kept | ((nall < 7 ? nall : 7u) << 28) | ((uint32_t)xfer << 31);
            if (nall >= 1 && nall <= 4) A.apos[g] = AmbPos{apos0, apos1};
            {   // A / C / T / G among the record's unambiguous bases, 16 bits each: with them the measure kernel prices a
                // Huffman-coded base alphabet without reading the bases again
                const uint32_t r0 = t0 - b_t0, r1 = t1 - b_t1, r01 = t01 - b_t01, rok = nok - b_nok;
                A.bcnt[g] = (uint64_t)(rok - r0 - r1 + r01) | ((uint64_t)(r0 - r01) << 16) | ((uint64_t)(r1 - r01) << 32) | ((uint64_t)r01 << 48);
            }
        }
        mq = L > mq ? L : mq;
        ms = kept > ms ? kept : ms;
        sum_l += L;

        // ---- the ambiguous positions, now that the record's transfer decision is known (phyNGSC.cpp:573-599) ----
        for (uint32_t k = 0; k < 4 && __any(nall > k && nall <= 4); ++k) {
            if (nall > k && nall <= 4) {
                const uint32_t pos = ((k < 2 ? apos0 : apos1) >> (16 * (k & 1))) & 0xffffu, row = pos + 1;
                const uint32_t c = A.buf[s0 + pos], q = A.buf[q0 + pos];
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
        if (__any(nall > 4)) {   // records with many ambiguity codes: rescan
            for (uint32_t ch = 0; __any(nall > 4 && 16 * ch < L); ++ch) {
                if (nall > 4 && 16 * ch < L) {
                    const uint32_t nv = L - 16 * ch < 16 ? L - 16 * ch : 16;
                    const uint4 S = load16_unaligned(A.buf, s0 + 16 * ch, A.buf_len);
                    const uint4 Q = load16_unaligned(A.buf, q0 + 16 * ch, A.buf_len);
                    const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
                    const uint32_t ambm = sr_amb_mask(sw, nv);
#pragma unroll
                    for (uint32_t u = 0; u < 16; ++u) {
                        if ((ambm >> u) & 1u) {
                            const uint32_t c = sr_byte(sw, u), q = sr_byte(qw, u), row = 16 * ch + u + 1;
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
    for (uint32_t k = tid; k < lds_rows * SR_HCOLS; k += THREADS) {
        const uint32_t row = k / SR_HCOLS, col = k % SR_HCOLS;
        const uint32_t x = s_hist[row * SR_HSTRIDE + col];
        if (x) atomicAdd(&qh[(size_t)row * 256 + 33 + col], x);
    }
    if (tid < 256 && s_occ[tid]) atomicAdd(&A.dna_occ[(size_t)s * 256 + tid], s_occ[tid]);
    if (tid == 0) {
        atomicMax(&A.sb[s].max_q, s_red[0]);
        atomicMax(&A.sb[s].max_s, s_red[1]);
        atomicAdd(&A.sb[s].n_bases, s_red[2]);
    }
}

// ====================================================================================================
// (round 3) LONG reads, staged: LPR lanes per record.  With one lane per record the 64 records of a wave span 22 KB of the
// file at 150 bases - too much LDS to stage them as the short-read form does - and every lane pulls its twenty 16-byte chunks
// in with scattered loads (the texture path prices a wave instruction of those at ~200 cycles; the BURST form above hides
// some of it).  Here a wave takes 64 / LPR records - a contiguous 5.6 KB at 150 bases and LPR = 4 - fetches them with
// coalesced 1 KB rows into LDS like the short-read form, and lane j of a record's LPR lanes walks its chunks j, j + LPR, ...
// Same statistics, same copies; what a record's lanes found (base counts, ambiguity codes and their positions) is put
// together with shuffles inside the aligned lane group.  The body of a chunk is the masked general path of
// k_sq_stats_reg (the lanes of a wave are at different positions of their reads, so no row offset is a constant).
template <uint32_t LPR>
__global__ __launch_bounds__(SR_THREADS) __attribute__((amdgpu_waves_per_eu(3))) void k_sq_stats_multi(Arena A, uint32_t lds_rows, uint32_t rpw, uint32_t span_cap)
{
    static_assert(LPR == 2 || LPR == 4 || LPR == 8, "lanes per record");
    if (*A.err) return;
    // Histogram: 16-bit counters, two positions per word - positions 16c + u and 16c + u + 8 (u < 8) share slot 8c + u, low and
    // high half (a workgroup sees at most 1024 records, so no half can overflow): 15 KB instead of 30 for 150-base reads, which
    // with the spans is the difference between two and four workgroups per CU.  lds_rows: rows of the sub-block's table (positions
    // + 1) - the launcher only takes this kernel when all of them fit.
    extern __shared__ __align__(16) uint32_t s_hist[];   // [slots][SR_HSTRIDE], 64 trash words, the waves' spans
    __shared__ uint32_t s_occ[256];
    __shared__ uint8_t s_amb[256];
    __shared__ uint32_t s_red[3];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t first = blockIdx.x * rpw;
    if (first >= n) return;
    const uint32_t last = first + rpw < n ? first + rpw : n;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t sub = (uint32_t)tid % LPR;          // this lane's place among its record's lanes
    const uint32_t nslots = 8 * ((lds_rows + 14) / 16);   // rows 1 .. lds_rows - 1 = positions 0 .. lds_rows - 2
    const uint32_t hwords = nslots * SR_HSTRIDE;
    uint8_t *const span = reinterpret_cast<uint8_t *>(s_hist + ((hwords + 64 + 3) & ~3u)) + (size_t)(tid >> 6) * span_cap;
    uint32_t span_lo = 0;

    for (uint32_t k = tid; k < hwords; k += SR_THREADS) s_hist[k] = 0;
    s_occ[tid] = 0;
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (tid < 3) s_red[tid] = 0;
    __syncthreads();

    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    uint8_t *hb = reinterpret_cast<uint8_t *>(s_hist);
    const uint32_t trash = (hwords + lane) * 4;
    uint32_t t0 = 0, t1 = 0, t01 = 0, nok = 0, mq = 0, ms = 0, sum_l = 0;
    // rotation of the quality bytes among the lanes that are at the same positions (the lanes sub, sub + LPR, ...)
    const uint32_t lj = ((uint32_t)lane / LPR) & 3;
    uint32_t bsh[4], boff[4];
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t bb = (b + (((uint32_t)lane / LPR) >> 2)) & 3;
        bsh[b] = 8 * bb;
        boff[b] = bb * SR_HSTRIDE * 4;
    }
    auto gsum = [](uint32_t v) __attribute__((always_inline)) {   // sum over the record's lanes
#pragma unroll
        for (uint32_t d = 1; d < LPR; d <<= 1) v += (uint32_t)__shfl_xor((int)v, (int)d);
        return v;
    };
    auto gor = [](uint32_t v) __attribute__((always_inline)) {
#pragma unroll
        for (uint32_t d = 1; d < LPR; d <<= 1) v |= (uint32_t)__shfl_xor((int)v, (int)d);
        return v;
    };

    constexpr uint32_t RPB = SR_THREADS / LPR;   // records per workgroup batch
    for (uint32_t base = first; base < last; base += RPB) {
        const uint32_t i = base + (uint32_t)tid / LPR;
        const bool valid = i < last;
        const uint32_t g = rec_lo + (valid ? i : first);
        const uint32_t te = A.te[g], se = A.se[g];
        const uint32_t L = valid ? se - te - 1 : 0;
        const uint32_t s0 = te + 1, q0 = se + 3;
        if (valid && sub == 0) {
            if (L == 0 && (load8_unaligned(A.buf, se, A.buf_len).x & 0x00ffff00u) != 0x000a2b00u) atomicMin(A.err, (int32_t)ERR_FORMAT);
            if (i == 0 && L >= 1 && !A.cs_mode) {
                if (cs_first_read(A.buf[s0], A.buf[s0 + 1])) atomicMin(A.err, (int32_t)ERR_COLORSPACE);
            }
        }
        {   // the wave's records -> LDS, 1 KB per row
            const unsigned long long vm = __ballot(valid);
            if (vm) {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)s0) & ~15u;
                const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(q0 + L), (int)__builtin_popcountll(vm) - 1);
                const uint32_t rows = (hi - lo + 1023u) >> 10;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                for (uint32_t r0 = 0; r0 < rows; r0 += 4) {
                    uint4 v[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k)
                        if (r0 + k < rows) v[k] = load16_guarded_sq(A.buf, lo + 1024u * (r0 + k) + 16u * lane, A.buf_len);
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k)
                        if (r0 + k < rows) *reinterpret_cast<uint4 *>(span + 1024u * (r0 + k) + 16u * lane) = v[k];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                span_lo = lo;
            }
        }
        auto ld16 = [&](uint32_t pos) __attribute__((always_inline)) -> uint4 {   // (aligned dwords + funnel shifts: see k_sq_stats_reg)
            const uint32_t o = pos - span_lo, sh = (o & 3u) * 8u;
            const uint32_t *q = reinterpret_cast<const uint32_t *>(span + (o & ~3u));
            const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
            return make_uint4(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh), __builtin_amdgcn_alignbit(d3, d2, sh),
                              __builtin_amdgcn_alignbit(d4, d3, sh));
        };

        const uint32_t b_t0 = t0, b_t1 = t1, b_t01 = t01, b_nok = nok;
        uint32_t fl = 0, namb = 0;
        uint32_t nall = 0, apos0 = 0, apos1 = 0;   // this LANE's non-ACGT bytes and the positions of its first four
        for (uint32_t grp = 0; __any(16 * (LPR * grp + sub) < L); ++grp) {
            const uint32_t ch = LPR * grp + sub;
            const uint32_t nv = L > 16 * ch ? (L - 16 * ch < 16 ? L - 16 * ch : 16) : 0;
            uint4 S = make_uint4(0, 0, 0, 0), Q = S;
            if (nv) {
                S = ld16(s0 + 16 * ch);
                Q = ld16(q0 + 16 * ch);
                sq_chunk_keep(A, g, 16 * ch, S, Q);
            }
            const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
            if (nv != 0 && L <= 16 * ch + 16) {   // the end of the read: "\n+\n" follows
                uint32_t two;
                if (nv <= 13) {
                    const uint32_t o = nv + 1, wi = o >> 2;
                    const uint32_t lo = wi == 0 ? sw[0] : (wi == 1 ? sw[1] : (wi == 2 ? sw[2] : sw[3]));
                    const uint32_t hi = wi == 0 ? sw[1] : (wi == 1 ? sw[2] : sw[3]);
                    two = __builtin_amdgcn_alignbit(hi, lo, (o & 3) * 8) & 0xffffu;
                } else {
                    two = (load4_unaligned(A.buf, se) >> 8) & 0xffffu;
                }
                if (two != 0x0a2bu) atomicMin(A.err, (int32_t)ERR_FORMAT);
            }
            uint32_t d[4] = {0, 0, 0, 0}, idx[4] = {0, 0, 0, 0}, dacc = 0, qbad = 0, c0 = 0, c1 = 0, c01 = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                const uint32_t w = (sw[k] & vm) | (0x41414141u & ~vm);
                idx[k] = sr_idx(w);
                d[k] = sr_not_acgt(w, idx[k]);
                dacc |= d[k];
                qbad |= sr_q_out(qw[k]) & vm;
                const uint32_t p0 = idx[k] & 0x01010101u, p1 = (idx[k] >> 1) & 0x01010101u;
                c0 += __builtin_popcount(p0);
                c1 += __builtin_popcount(p1);
                c01 += __builtin_popcount(p0 & p1);
            }
            const uint32_t rb = 8 * ch * SR_HSTRIDE * 4 - 33 * 4;   // (per lane: the lanes of a wave are at different positions)
            if (!__any(nv != 0 && (nv != 16 || dacc != 0 || qbad != 0))) {
                // every lane that still has bytes: sixteen plain qualities, only A/C/G/T
                if (nv) {
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t j = (k + lj) & 3;
                        const uint32_t wsel = j == 0 ? qw[0] : (j == 1 ? qw[1] : (j == 2 ? qw[2] : qw[3]));
                        const uint32_t joff = rb + (j & 1) * (4 * SR_HSTRIDE * 4), inc = 1u << (16 * (j >> 1));
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t q = (wsel >> bsh[b]) & 0xffu;
                            atomicAdd(reinterpret_cast<uint32_t *>(hb + joff + boff[b] + 4 * q), inc);
                        }
                    }
                    t0 += c0; t1 += c1; t01 += c01; nok += 16;
                }
            } else {
                uint32_t ambm = 0, qbm = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    ambm |= t_movemask(t_nonzero01(d[k])) << (4 * k);
                    qbm |= t_movemask((sr_q_out(qw[k]) >> 7) & 0x01010101u) << (4 * k);
                }
                const uint32_t validm = (1u << nv) - 1u;
                ambm &= validm;
                const uint32_t okm = validm & ~ambm & ~qbm;          // (every row of the table has its slot)
                const uint32_t glm = validm & ~ambm & ~okm;
                if (ambm) {
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
                // (bytes behind the read's end were counted as 'A' with both planes clear: nothing to take back)
                t0 += c0; t1 += c1; t01 += c01; nok += nv - __builtin_popcount(ambm);
                if (__any(okm)) {   // branch-free: bytes that do not count go to the lane's own trash word
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t j = (k + lj) & 3;
                        const uint32_t wsel = j == 0 ? qw[0] : (j == 1 ? qw[1] : (j == 2 ? qw[2] : qw[3]));
                        const uint32_t joff = rb + (j & 1) * (4 * SR_HSTRIDE * 4), inc = 1u << (16 * (j >> 1));
                        const uint32_t okw = okm >> (4 * j);
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t q = (wsel >> bsh[b]) & 0xffu;
                            const uint32_t a = ((okw >> (bsh[b] >> 3)) & 1u) ? joff + boff[b] + 4 * q : trash;
                            atomicAdd(reinterpret_cast<uint32_t *>(hb + a), inc);
                        }
                    }
                }
                if (__any(glm)) {
#pragma unroll
                    for (uint32_t u = 0; u < 16; ++u)
                        if ((glm >> u) & 1u) atomicAdd(&qh[(size_t)(16 * ch + u + 1) * 256 + sr_byte(qw, u)], 1u);
                }
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (!__any((ambm >> (4 * k)) & 15u)) continue;
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) {
                        const uint32_t u = 4 * k + b;
                        if ((ambm >> u) & 1u) {
                            const uint32_t a = s_amb[sr_byte(sw, u)];
                            if (a == 0) fl |= 2;
                            else {
                                const uint32_t q = sr_byte(qw, u);
                                fl |= (q < 33 || q > 40) ? 2u : 1u;
                                ++namb;
                            }
                            const uint32_t p16 = 16 * ch + u;
                            const uint32_t pv = p16 << (16 * (nall & 1u));
                            apos0 |= nall < 2 ? pv : 0u;
                            apos1 |= (nall >= 2 && nall < 4) ? pv : 0u;
                            ++nall;
                        }
                    }
                }
            }
        }
        // ---- the record's lanes put together what they found ----
        const uint32_t fl_r = gor(fl), namb_r = gsum(namb), nall_r = gsum(nall < 8 ? nall : 8u);
        const bool xfer = (fl_r & 1) && !(fl_r & 2);
        const uint32_t kept = L - (xfer ? namb_r : 0);
        {
            // positions of up to four ambiguity codes of the record: every lane's list moves behind those of the lanes in front
            uint32_t before = 0;
#pragma unroll
            for (uint32_t d = 1; d < LPR; ++d) {
                const uint32_t o = (uint32_t)__shfl((int)(nall < 8 ? nall : 8u), (int)(((uint32_t)lane & ~(LPR - 1)) + ((sub + LPR - d) % LPR)));
                before += (sub >= d) ? o : 0u;   // lanes sub - d (those in front of this one)
            }
            uint64_t mine = ((uint64_t)apos1 << 32) | apos0;
            mine = before < 4 ? mine << (16 * before) : 0ull;
            uint32_t m0 = gor((uint32_t)mine), m1 = gor((uint32_t)(mine >> 32));
            const uint32_t r0 = gsum(t0 - b_t0), r1 = gsum(t1 - b_t1), r01 = gsum(t01 - b_t01), rok = gsum(nok - b_nok);
            if (valid && sub == 0) {
                A.slen[g] = kept | ((nall_r < 7 ? nall_r : 7u) << 28) | ((uint32_t)xfer << 31);
                if (nall_r >= 1 && nall_r <= 4) A.apos[g] = AmbPos{m0, m1};
                A.bcnt[g] = (uint64_t)(rok - r0 - r1 + r01) | ((uint64_t)(r0 - r01) << 16) | ((uint64_t)(r1 - r01) << 32) | ((uint64_t)r01 << 48);
                mq = L > mq ? L : mq;
                ms = kept > ms ? kept : ms;
                sum_l += L;
            }
        }
        // ---- the ambiguous positions, now that the record's transfer decision is known (phyNGSC.cpp:573-599): each lane its own ----
        for (uint32_t k = 0; k < 4 && __any(nall > k && nall_r <= 4); ++k) {
            if (nall > k && nall_r <= 4) {
                const uint32_t pos = ((k < 2 ? apos0 : apos1) >> (16 * (k & 1))) & 0xffffu, row = pos + 1;
                const uint32_t c = A.buf[s0 + pos], q = A.buf[q0 + pos];
                const uint32_t a = s_amb[c];
                if (xfer && a > 1) {
                    atomicAdd(&qh[(size_t)row * 256 + transfer_quality(a, (uint8_t)q)], 1u);
                } else {
                    const uint32_t col = q - 33;
                    if (col < SR_HCOLS && row < lds_rows) atomicAdd(&s_hist[(8 * (pos >> 4) + (pos & 7)) * SR_HSTRIDE + col], 1u << (2 * (pos & 8)));
                    else atomicAdd(&qh[(size_t)row * 256 + q], 1u);
                    atomicAdd(&s_occ[c], 1u);
                }
            }
        }
        if (__any(nall_r > 4)) {   // records with many ambiguity codes: rescanned by their first lane
            for (uint32_t ch = 0; __any(nall_r > 4 && sub == 0 && 16 * ch < L); ++ch) {
                if (nall_r > 4 && sub == 0 && 16 * ch < L) {
                    const uint32_t nv = L - 16 * ch < 16 ? L - 16 * ch : 16;
                    const uint4 S = load16_unaligned(A.buf, s0 + 16 * ch, A.buf_len);
                    const uint4 Q = load16_unaligned(A.buf, q0 + 16 * ch, A.buf_len);
                    const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
                    const uint32_t ambm = sr_amb_mask(sw, nv);
#pragma unroll
                    for (uint32_t u = 0; u < 16; ++u) {
                        if ((ambm >> u) & 1u) {
                            const uint32_t c = sr_byte(sw, u), q = sr_byte(qw, u), row = 16 * ch + u + 1;
                            const uint32_t a = s_amb[c];
                            if (xfer && a > 1) {
                                atomicAdd(&qh[(size_t)row * 256 + transfer_quality(a, (uint8_t)q)], 1u);
                            } else {
                                const uint32_t col = q - 33;
                                const uint32_t pos = row - 1;
                                if (col < SR_HCOLS && row < lds_rows) atomicAdd(&s_hist[(8 * (pos >> 4) + (pos & 7)) * SR_HSTRIDE + col], 1u << (2 * (pos & 8)));
                                else atomicAdd(&qh[(size_t)row * 256 + q], 1u);
                                atomicAdd(&s_occ[c], 1u);
                            }
                        }
                    }
                }
            }
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
    for (uint32_t k = tid; k < nslots * SR_HCOLS; k += SR_THREADS) {
        const uint32_t slot = k / SR_HCOLS, col = k % SR_HCOLS;
        const uint32_t x = s_hist[slot * SR_HSTRIDE + col];
        const uint32_t row = 16 * (slot >> 3) + (slot & 7) + 1;   // the low half's row; the high half's is eight further on
        if (x & 0xffffu) atomicAdd(&qh[(size_t)row * 256 + 33 + col], x & 0xffffu);
        if (x >> 16) atomicAdd(&qh[(size_t)(row + 8) * 256 + 33 + col], x >> 16);
    }
    if (s_occ[tid]) atomicAdd(&A.dna_occ[(size_t)s * 256 + tid], s_occ[tid]);
    if (tid == 0) {
        atomicMax(&A.sb[s].max_q, s_red[0]);
        atomicMax(&A.sb[s].max_s, s_red[1]);
        atomicAdd(&A.sb[s].n_bases, s_red[2]);
    }
}

// ====================================================================================================
// (round 4) UNEVENLY long reads, staged: the CHUNKS of a batch of records dealt to the lanes.  With a lane (or four) per record
// the wave walks as many trips as its longest read needs - on 36..300-base reads half of the lane-trips are idle - and the
// one-lane-per-record form pulls every chunk in with scattered loads besides (k_sq_stats_reg<BURST>: 0.63 ms per 400 MB, the
// longest kernel of that configuration).  Here a wave takes FR consecutive records (a contiguous span of the file, staged in LDS
// with coalesced rows like the other staged forms) and lists their 16-base chunks chunk-index-major - all first chunks, all second
// chunks, ... of the records that have one - so that lane x of a trip takes entry 64 t + x: no idle lanes but in the last trip,
// and neighbouring lanes hold the same chunk index of neighbouring records, which is what the chunk-major copies want
// (coalesced stores).  What a record's chunks find (base counts, ambiguity codes and their positions) meets in per-record LDS
// words; the record's own lane closes it (transfer decision, lengths, the second look at its ambiguous positions).
constexpr uint32_t FK = 32;          // chunk indices per record at most (reads of up to 512 bases)
template <uint32_t FR>               // FR: records per wave and batch (16, or 8 where the spans of 16 would leave a CU with two waves per SIMD)
struct FlatWave {                    // a wave's tables for its batch
    uint32_t s0[FR], q0[FR], L[FR];  // offsets in the span, read length
    uint32_t mask[FK];               // bit r: record r has a chunk of this index
    uint32_t first[FK + 1];          // entries in front of this chunk index
    uint32_t fl[FR], namb[FR], nall[FR], apos[FR][4];
    unsigned long long cnt[FR];      // c0 | c1 << 16 | c01 << 32 | nok << 48 over the record's chunks
};

template <uint32_t THREADS, uint32_t FR>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(3))) void k_sq_stats_flat(Arena A, uint32_t lds_rows, uint32_t rpw, uint32_t span_cap)
{
    static_assert(FR == 8 || FR == 16, "the record select below walks a 16-bit mask");
    if (*A.err) return;
    // histogram: 16-bit counters, two positions per word, as in k_sq_stats_multi (a workgroup sees at most 1024 records)
    extern __shared__ __align__(16) uint32_t s_hist[];   // [slots][SR_HSTRIDE], 64 trash words, the waves' spans
    __shared__ uint32_t s_occ[256];
    __shared__ uint8_t s_amb[256];
    __shared__ uint32_t s_red[3];
    __shared__ FlatWave<FR> s_fw[THREADS / 64];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t first = blockIdx.x * rpw;
    if (first >= n) return;
    const uint32_t last = first + rpw < n ? first + rpw : n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t nslots = 8 * ((lds_rows + 14) / 16);
    const uint32_t hwords = nslots * SR_HSTRIDE;
    uint8_t *const span = reinterpret_cast<uint8_t *>(s_hist + ((hwords + 64 + 3) & ~3u)) + (size_t)wave * span_cap;
    FlatWave<FR> &W = s_fw[wave];

    for (uint32_t k = tid; k < hwords; k += THREADS) s_hist[k] = 0;
    if (tid < 256) { s_occ[tid] = 0; s_amb[tid] = (uint8_t)amb_code((uint8_t)tid); }
    if (tid < 3) s_red[tid] = 0;
    __syncthreads();

    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    uint8_t *hb = reinterpret_cast<uint8_t *>(s_hist);
    const uint32_t trash = (hwords + lane) * 4;
    uint32_t t0 = 0, t1 = 0, t01 = 0, nok = 0, mq = 0, ms = 0, sum_l = 0;   // (lanes 0 .. FR - 1: over their records)
    const uint32_t lj = (uint32_t)lane & 3;
    uint32_t bsh[4], boff[4];
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t bb = (b + ((uint32_t)lane >> 2)) & 3;
        bsh[b] = 8 * bb;
        boff[b] = bb * SR_HSTRIDE * 4;
    }
    auto sync_wave = []() __attribute__((always_inline)) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };

    for (uint32_t base = first + wave * FR; base < last; base += (THREADS / 64) * FR) {
        // ---- the batch's records: lanes 0 .. FR - 1 ----
        const uint32_t i = base + (uint32_t)lane;
        const bool valid = (uint32_t)lane < FR && i < last;
        const uint32_t g = rec_lo + (valid ? i : base);
        const uint32_t te = A.te[g], se = A.se[g];
        const uint32_t L = valid ? se - te - 1 : 0;
        const uint32_t s0 = te + 1, q0 = se + 3;
        if (valid) {
            if (L == 0 && (load8_unaligned(A.buf, se, A.buf_len).x & 0x00ffff00u) != 0x000a2b00u) atomicMin(A.err, (int32_t)ERR_FORMAT);
            if (i == 0 && L >= 1 && !A.cs_mode) {
                if (cs_first_read(A.buf[s0], A.buf[s0 + 1])) atomicMin(A.err, (int32_t)ERR_COLORSPACE);
            }
        }
        const unsigned long long vm = __ballot(valid);   // (a prefix of the lanes; not empty: base < last)
        const uint32_t nrec = (uint32_t)__builtin_popcountll(vm);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)s0) & ~15u;
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(q0 + L), (int)nrec - 1);
        {   // the batch's bytes -> LDS, 1 KB per row
            const uint32_t rows = (hi - lo + 1023u) >> 10;
            for (uint32_t r0 = 0; r0 < rows; r0 += 4) {
                uint4 v[4];
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)
                    if (r0 + k < rows) v[k] = load16_guarded_sq(A.buf, lo + 1024u * (r0 + k) + 16u * (uint32_t)lane, A.buf_len);
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k)   // (span_cap is a multiple of 16, not of a row: the last row stops at the span's end)
                    if (r0 + k < rows && 1024u * (r0 + k) + 16u * (uint32_t)lane < span_cap) *reinterpret_cast<uint4 *>(span + 1024u * (r0 + k) + 16u * (uint32_t)lane) = v[k];
            }
        }
        const uint32_t nch = (L + 15) / 16;
        if ((uint32_t)lane < FR) {
            W.s0[lane] = s0 - lo; W.q0[lane] = q0 - lo; W.L[lane] = L;
            W.fl[lane] = 0; W.namb[lane] = 0; W.nall[lane] = 0; W.cnt[lane] = 0ull;
            W.apos[lane][0] = W.apos[lane][1] = W.apos[lane][2] = W.apos[lane][3] = 0;
        }
        // chunk-index-major list of the batch's chunks: mask[k] = the records that have chunk k, first[k] = entries in front
        uint32_t total = 0, kmax = 0;
        for (uint32_t k = 0; k < FK; ++k) {
            const uint32_t m = (uint32_t)__ballot(valid && nch > k);
            if (lane == 0) { W.mask[k] = m; W.first[k] = total; }
            if (!m) break;
            total += (uint32_t)__builtin_popcount(m);
            kmax = k + 1;
        }
        if (lane == 0) W.first[kmax] = total;
        if (__any(valid && nch > FK)) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED);   // (the launcher takes this form for reads of up to 16 FK bases only)
        sync_wave();

        // ---- the chunks: entry 64 t + lane ----
        for (uint32_t x0 = 0; x0 < total; x0 += 64) {
            const uint32_t x = x0 + (uint32_t)lane;
            const bool act = x < total;
            // chunk index: the level whose entries hold x (a trip's 64 entries span few levels: from the first lane's level on)
            uint32_t k = 0;
            {
                uint32_t kf = 0;
                while (kf + 1 < kmax && W.first[kf + 1] <= x0) ++kf;   // (uniform)
                k = kf;
                for (uint32_t j = kf + 1; j < kmax && W.first[j] < x0 + 64; ++j) k += (W.first[j] <= x) ? 1u : 0u;   // (uniform trip count)
            }
            // the record: the (x - first[k])-th set bit of mask[k]
            uint32_t r = 0;
            {
                uint32_t m = act ? W.mask[k] : 1u, j = act ? x - W.first[k] : 0u, c;
                c = (uint32_t)__builtin_popcount(m & 0xffu); if (j >= c) { j -= c; r += 8; m >>= 8; }
                c = (uint32_t)__builtin_popcount(m & 0xfu);  if (j >= c) { j -= c; r += 4; m >>= 4; }
                c = (uint32_t)__builtin_popcount(m & 0x3u);  if (j >= c) { j -= c; r += 2; m >>= 2; }
                c = m & 1u;                                  if (j >= c) { r += 1; }
            }
            const uint32_t Lr = act ? W.L[r] : 0u, ch = k;
            const uint32_t nv = Lr > 16 * ch ? (Lr - 16 * ch < 16 ? Lr - 16 * ch : 16) : 0;
            const uint32_t gr = rec_lo + base + r;
            uint4 S = make_uint4(0, 0, 0, 0), Q = S;
            if (nv) {
                auto ld16 = [&](uint32_t o) __attribute__((always_inline)) -> uint4 {   // (aligned dwords + funnel shifts: see k_sq_stats_reg)
                    const uint32_t sh = (o & 3u) * 8u;
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(span + (o & ~3u));
                    const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                    return make_uint4(__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh), __builtin_amdgcn_alignbit(d3, d2, sh),
                                      __builtin_amdgcn_alignbit(d4, d3, sh));
                };
                S = ld16(W.s0[r] + 16 * ch);
                Q = ld16(W.q0[r] + 16 * ch);
                sq_chunk_keep(A, gr, 16 * ch, S, Q);
            }
            const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
            if (nv != 0 && Lr <= 16 * ch + 16) {   // the end of the read: "\n+\n" follows (in the span: it holds the records whole)
                const uint32_t o = W.s0[r] + Lr;
                const uint32_t two = (uint32_t)span[o + 1] | ((uint32_t)span[o + 2] << 8);
                if (two != 0x0a2bu) atomicMin(A.err, (int32_t)ERR_FORMAT);
            }
            uint32_t d[4] = {0, 0, 0, 0}, idx[4] = {0, 0, 0, 0}, dacc = 0, qbad = 0, c0 = 0, c1 = 0, c01 = 0;
#pragma unroll
            for (uint32_t kk = 0; kk < 4; ++kk) {
                const uint32_t vmk = t_lowbytes32((int32_t)nv - 4 * (int32_t)kk);
                const uint32_t w = (sw[kk] & vmk) | (0x41414141u & ~vmk);
                idx[kk] = sr_idx(w);
                d[kk] = sr_not_acgt(w, idx[kk]);
                dacc |= d[kk];
                qbad |= sr_q_out(qw[kk]) & vmk;
                const uint32_t p0 = idx[kk] & 0x01010101u, p1 = (idx[kk] >> 1) & 0x01010101u;
                c0 += __builtin_popcount(p0);
                c1 += __builtin_popcount(p1);
                c01 += __builtin_popcount(p0 & p1);
            }
            const uint32_t rb = 8 * ch * SR_HSTRIDE * 4 - 33 * 4;   // (per lane: the lanes of a trip are at different chunk indices)
            uint32_t cnok = nv;
            if (!__any(nv != 0 && (nv != 16 || dacc != 0 || qbad != 0))) {
                if (nv) {   // sixteen plain qualities, only A / C / G / T
#pragma unroll
                    for (uint32_t kk = 0; kk < 4; ++kk) {
                        const uint32_t j = (kk + lj) & 3;
                        const uint32_t wsel = j == 0 ? qw[0] : (j == 1 ? qw[1] : (j == 2 ? qw[2] : qw[3]));
                        const uint32_t joff = rb + (j & 1) * (4 * SR_HSTRIDE * 4), inc = 1u << (16 * (j >> 1));
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t q = (wsel >> bsh[b]) & 0xffu;
                            atomicAdd(reinterpret_cast<uint32_t *>(hb + joff + boff[b] + 4 * q), inc);
                        }
                    }
                }
            } else {
                uint32_t ambm = 0, qbm = 0;
#pragma unroll
                for (uint32_t kk = 0; kk < 4; ++kk) {
                    ambm |= t_movemask(t_nonzero01(d[kk])) << (4 * kk);
                    qbm |= t_movemask((sr_q_out(qw[kk]) >> 7) & 0x01010101u) << (4 * kk);
                }
                const uint32_t validm = (1u << nv) - 1u;
                ambm &= validm;
                const uint32_t okm = validm & ~ambm & ~qbm;
                const uint32_t glm = validm & ~ambm & ~okm;
                if (ambm) {
                    c0 = c1 = c01 = 0;
#pragma unroll
                    for (uint32_t kk = 0; kk < 4; ++kk) {
                        const uint32_t ok01 = t_nonzero01(d[kk]) ^ 0x01010101u;
                        const uint32_t p0 = idx[kk] & ok01, p1 = (idx[kk] >> 1) & ok01;
                        c0 += __builtin_popcount(p0);
                        c1 += __builtin_popcount(p1);
                        c01 += __builtin_popcount(p0 & p1);
                    }
                }
                cnok = nv - __builtin_popcount(ambm);
                if (__any(okm)) {
#pragma unroll
                    for (uint32_t kk = 0; kk < 4; ++kk) {
                        const uint32_t j = (kk + lj) & 3;
                        const uint32_t wsel = j == 0 ? qw[0] : (j == 1 ? qw[1] : (j == 2 ? qw[2] : qw[3]));
                        const uint32_t joff = rb + (j & 1) * (4 * SR_HSTRIDE * 4), inc = 1u << (16 * (j >> 1));
                        const uint32_t okw = okm >> (4 * j);
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t q = (wsel >> bsh[b]) & 0xffu;
                            const uint32_t a = ((okw >> (bsh[b] >> 3)) & 1u) ? joff + boff[b] + 4 * q : trash;
                            atomicAdd(reinterpret_cast<uint32_t *>(hb + a), inc);
                        }
                    }
                }
                if (__any(glm)) {
#pragma unroll
                    for (uint32_t u = 0; u < 16; ++u)
                        if ((glm >> u) & 1u) atomicAdd(&qh[(size_t)(16 * ch + u + 1) * 256 + sr_byte(qw, u)], 1u);
                }
                if (__any(ambm)) {
#pragma unroll
                    for (uint32_t u = 0; u < 16; ++u) {
                        if ((ambm >> u) & 1u) {
                            const uint32_t a = s_amb[sr_byte(sw, u)];
                            uint32_t f = 2u;
                            if (a != 0) {
                                const uint32_t q = sr_byte(qw, u);
                                f = (q < 33 || q > 40) ? 2u : 1u;
                                atomicAdd(&W.namb[r], 1u);
                            }
                            atomicOr(&W.fl[r], f);
                            const uint32_t slot = atomicAdd(&W.nall[r], 1u);
                            if (slot < 4) W.apos[r][slot] = 16 * ch + u;
                        }
                    }
                }
            }
            if (nv) atomicAdd(&W.cnt[r], (unsigned long long)c0 | ((unsigned long long)c1 << 16) | ((unsigned long long)c01 << 32) | ((unsigned long long)cnok << 48));
        }
        sync_wave();

        // ---- the records again: what their chunks found ----
        const uint32_t fl_r = valid ? W.fl[lane] : 0u, namb_r = valid ? W.namb[lane] : 0u, nall_r = valid ? W.nall[lane] : 0u;
        const bool xfer = (fl_r & 1) && !(fl_r & 2);
        const uint32_t kept = L - (xfer ? namb_r : 0);
        if (valid) {
            const unsigned long long cn = W.cnt[lane];
            const uint32_t r0 = (uint32_t)(cn & 0xffffu), r1 = (uint32_t)((cn >> 16) & 0xffffu), r01 = (uint32_t)((cn >> 32) & 0xffffu), rok = (uint32_t)(cn >> 48);
            A.slen[g] = kept | ((nall_r < 7 ? nall_r : 7u) << 28) | ((uint32_t)xfer << 31);
            if (nall_r >= 1 && nall_r <= 4) A.apos[g] = AmbPos{W.apos[lane][0] | (W.apos[lane][1] << 16), W.apos[lane][2] | (W.apos[lane][3] << 16)};
            A.bcnt[g] = (uint64_t)(rok - r0 - r1 + r01) | ((uint64_t)(r0 - r01) << 16) | ((uint64_t)(r1 - r01) << 32) | ((uint64_t)r01 << 48);
            t0 += r0; t1 += r1; t01 += r01; nok += rok;
            mq = L > mq ? L : mq;
            ms = kept > ms ? kept : ms;
            sum_l += L;
        }
        // the ambiguous positions, now that the record's transfer decision is known (phyNGSC.cpp:573-599)
        if (__any(valid && nall_r != 0)) {
            if (valid && nall_r != 0) {
                auto one = [&](uint32_t pos, uint32_t c, uint32_t q) __attribute__((always_inline)) {
                    const uint32_t a = s_amb[c], row = pos + 1;
                    if (xfer && a > 1) {
                        atomicAdd(&qh[(size_t)row * 256 + transfer_quality(a, (uint8_t)q)], 1u);   // 128..247: no LDS column
                    } else {
                        const uint32_t col = q - 33;
                        if (col < SR_HCOLS && row < lds_rows) atomicAdd(&s_hist[(8 * (pos >> 4) + (pos & 7)) * SR_HSTRIDE + col], 1u << (2 * (pos & 8)));
                        else atomicAdd(&qh[(size_t)row * 256 + q], 1u);
                        atomicAdd(&s_occ[c], 1u);
                    }
                };
                const uint8_t *sp = span + (s0 - lo), *qp = span + (q0 - lo);
                if (nall_r <= 4) {
                    for (uint32_t k = 0; k < nall_r; ++k) { const uint32_t pos = W.apos[lane][k]; one(pos, sp[pos], qp[pos]); }
                } else {
                    for (uint32_t pos = 0; pos < L; ++pos) {
                        const uint32_t c = sp[pos];
                        if (!(c == 'A' || c == 'C' || c == 'G' || c == 'T')) one(pos, c, qp[pos]);
                    }
                }
            }
        }
        sync_wave();   // (the span and the tables are rewritten by the next batch)
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
    for (uint32_t k = tid; k < nslots * SR_HCOLS; k += THREADS) {
        const uint32_t slot = k / SR_HCOLS, col = k % SR_HCOLS;
        const uint32_t x = s_hist[slot * SR_HSTRIDE + col];
        const uint32_t row = 16 * (slot >> 3) + (slot & 7) + 1;   // the low half's row; the high half's is eight further on
        if (x & 0xffffu) atomicAdd(&qh[(size_t)row * 256 + 33 + col], x & 0xffffu);
        if (x >> 16) atomicAdd(&qh[(size_t)(row + 8) * 256 + 33 + col], x >> 16);
    }
    if (tid < 256 && s_occ[tid]) atomicAdd(&A.dna_occ[(size_t)s * 256 + tid], s_occ[tid]);
    if (tid == 0) {
        atomicMax(&A.sb[s].max_q, s_red[0]);
        atomicMax(&A.sb[s].max_s, s_red[1]);
        atomicAdd(&A.sb[s].n_bases, s_red[2]);
    }
}

// ====================================================================================================
// Measure / pack
// ====================================================================================================
// Quality code tables in LDS, indexed by the raw quality byte: column = q - 33 (64 columns), row = position
// (tasks.cpp:609-619 uses Huffman tree j+1 for position j).  Bytes outside 33..96 - the transferred qualities
// 128..247 among them - and rows past the LDS budget are looked up in the global table through qua_code.
struct SqTabs {
    const uint32_t *qg;     // global [qrows][256] by code index
    const uint8_t *qcode;   // LDS [256] quality byte -> code index
    const uint8_t *amb;     // LDS [256]
    const uint32_t *dcode;  // LDS [256] base byte -> code << 5 | len
    uint32_t lds_rows;
};

// Rows of SQ_QS entries: column q - 33 for the bytes 33..80, column 48 always zero, SQ_QROWS_EXTRA zero rows behind the
// table.  The lean paths of measure and pack replace every byte that has no entry (behind the end of the read, a
// transferred quality) by SQ_QPAD: it reads the zero entry, and the 16 lookups of a chunk are straight-line code.
// (round 3) Eight more columns, 49..56, hold the entries of the TRANSFERRED quality of an N (phyNGSC.cpp:579: 208 + (q - 33),
// q = '!'..'('): the pack kernel turns the quality byte of a transferred N into q + 49 and looks it up like any other - N is
// the ambiguity code sequencers write, and with 0.2 % of them nearly every wave-chunk of 150-base reads holds one.
constexpr uint32_t SQ_NCOL0 = SR_HCOLS + 1;      // first of the eight transferred-N columns
constexpr uint32_t SQ_QS = SR_HCOLS + 1 + 8;
constexpr uint32_t SQ_QPAD = 0x51515151u;        // 33 + 48 in every byte
constexpr uint32_t SQ_QROWS_EXTRA = 17;          // pad bytes of the last chunk, and the (q - 33) & 63 reach of the general path
// The tables come from their ready-made copies in global memory (Arena::qfmt, k_sq_format_qtab): rows row_lo .. of the
// sub-block's table as they lie there - a straight 8-byte copy instead of a clearing pass, a barrier and one division per
// entry per workgroup (and, in the windowed pack kernel, per window and tile: five times per tile of 150-base reads).  The rows
// behind lds_rows hold the table's next rows instead of zeros: nothing reads their values (see SQ_QROWS_EXTRA).  The pack kernel
// walks long reads window by window with a table of one window's rows (row_lo), see k_sq_pack_reg.
// (the byte tables lie behind the dword tables: 16-byte multiples per sub-block, the pad bytes never written nor used)
__device__ __host__ __forceinline__ uint32_t sq_qfmt8_stride(uint32_t qrows) { return ((qrows + SQ_QROWS_EXTRA) * SQ_QS + 15u) & ~15u; }
__device__ __forceinline__ uint8_t *sq_qfmt8(const Arena &A, uint32_t s)
{
    return reinterpret_cast<uint8_t *>(A.qfmt + (size_t)A.n_sb * A.qfmt_stride) + (size_t)s * sq_qfmt8_stride(A.qrows);
}
// the measure kernel's table: the first lds_rows + SQ_QROWS_EXTRA rows of the byte table, 16 bytes per step (the LDS area is
// a 16-byte multiple, see launch_sq_measure_reg)
__device__ __forceinline__ void sq_copy_qtab8(const Arena &A, uint32_t s, uint8_t *s_qlen, uint32_t lds_rows)
{
    const uint32_t bytes = ((lds_rows + SQ_QROWS_EXTRA) * SQ_QS + 15u) & ~15u;   // (<= the sub-block's stride: lds_rows <= qrows)
    const uint4 *src = reinterpret_cast<const uint4 *>(sq_qfmt8(A, s));
    for (uint32_t k = threadIdx.x; k < bytes / 16; k += SR_THREADS) reinterpret_cast<uint4 *>(s_qlen)[k] = src[k];
}

__device__ __forceinline__ void sq_copy_qtab(const Arena &A, uint32_t s, uint32_t *s_qtab, uint32_t lds_rows, uint32_t row_lo)
{
    const uint32_t words = (lds_rows + SQ_QROWS_EXTRA) * SQ_QS, lim = (A.qrows + SQ_QROWS_EXTRA) * SQ_QS, src0 = row_lo * SQ_QS;
    const uint32_t *src = A.qfmt + (size_t)s * A.qfmt_stride;
    for (uint32_t k = 2 * threadIdx.x; k < words; k += 2 * SR_THREADS) {   // (row_lo is even - 0 or a multiple of the window: so is src0)
        uint2 v = make_uint2(0, 0);
        if (src0 + k + 1 < lim) v = *reinterpret_cast<const uint2 *>(src + src0 + k);
        else if (src0 + k < lim) v.x = src[src0 + k];
        if (k + 1 < words) *reinterpret_cast<uint2 *>(s_qtab + k) = v;
        else s_qtab[k] = v.x;
    }
}

__global__ __launch_bounds__(SR_THREADS) void k_sq_format_qtab(Arena A)
{
    if (*A.err) return;
    __shared__ uint16_t s_inv[SQ_QS];   // column -> alphabet index + 1
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    if (threadIdx.x < SQ_QS) s_inv[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < sb.nqual) {
        const uint32_t qv = sb.qualities[threadIdx.x];
        // (208..215: a transferred N.  The raw bytes 82..89, which would land on those columns, have none: the measure kernel
        // reads the N columns whatever else the alphabet holds)
        const uint32_t col = (qv >= 208u && qv < 216u) ? SQ_NCOL0 + (qv - 208u) : (qv - 33u < SR_HCOLS ? qv - 33u : ~0u);
        if (col < SQ_QS) s_inv[col] = (uint16_t)(threadIdx.x + 1);
    }
    __syncthreads();
    const uint32_t words = (A.qrows + SQ_QROWS_EXTRA) * SQ_QS;
    const uint32_t *qg = A.qcodes + (size_t)s * A.qrows * 256;
    uint32_t *dst = A.qfmt + (size_t)s * A.qfmt_stride;
    uint8_t *dst8 = sq_qfmt8(A, s);   // (the measure kernel's table: the code lengths alone, a byte each)
    for (uint32_t k = blockIdx.x * SR_THREADS + threadIdx.x; k < words; k += gridDim.x * SR_THREADS) {
        const uint32_t row = k / SQ_QS, inv = s_inv[k % SQ_QS];
        const uint32_t e = (inv && row < A.qrows) ? qg[(size_t)row * 256 + inv - 1] : 0u;
        dst[k] = e;
        dst8[k] = (uint8_t)(e & 31u);
    }
}

// May the lean paths run on this sub-block?  Every alphabet symbol below 128 has a table column (the symbols from 128 on are
// transferred qualities, or raw bytes with the high bit set, which the lean paths test for per chunk) and every row is in
// the table.  Whole workgroup; thread = alphabet index.
__device__ __forceinline__ bool sq_lean_alphabet(const SbState &sb, uint32_t lds_rows, uint32_t *s_flag, bool windowed = false)
{
    if (threadIdx.x == 0) *s_flag = 0;
    __syncthreads();
    if (threadIdx.x < sb.nqual) {
        const uint32_t q = sb.qualities[threadIdx.x];
        if (q < 128 && (q < 33 || q > 32 + SR_HCOLS)) atomicOr(s_flag, 1u);
    }
    __syncthreads();
    return *s_flag == 0 && (windowed || lds_rows > sb.max_q);
}

// entry (code << 5 | len) of quality byte q at position row-1, any q, any row
__device__ __forceinline__ uint32_t sq_qentry_slow(const SqTabs &T, uint32_t row, uint32_t q)
{
    return T.qg[(size_t)row * 256 + T.qcode[q]];
}

// ---- measure: payload bits per record and per 256-record tile ----
template <bool BURST>
__global__ __launch_bounds__(SR_THREADS) void k_sq_measure_reg(Arena A, uint32_t lds_rows, uint32_t tpw)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_qlen[];   // [lds_rows][64] code lengths
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    __shared__ uint32_t s_tot[2];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + SR_THREADS - 1) / SR_THREADS;
    const uint32_t tile_first = blockIdx.x * tpw;
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;
    const int tid = threadIdx.x, lane = tid & 63;

    s_dcode[tid] = A.dcodes[(size_t)s * 256 + tid];
    s_qcode[tid] = sb.qua_code[tid];
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (tid < 2) s_tot[tid] = 0;
    sq_copy_qtab8(A, s, s_qlen, lds_rows);
    __syncthreads();
    SqTabs T = {A.qcodes + (size_t)s * A.qrows * 256, s_qcode, s_amb, s_dcode, lds_rows};
    const bool plain = (sb.flags & F_DNA_PLAIN) != 0;
    __shared__ uint32_t s_leanflag;
    const bool lean_wg = sq_lean_alphabet(sb, lds_rows, &s_leanflag);
    // code lengths of A, C, T, G (the order of (c >> 1) & 3) as a v_perm table
    const uint32_t ltab = (s_dcode['A'] & 31u) | ((s_dcode['C'] & 31u) << 8) | ((s_dcode['T'] & 31u) << 16) | ((s_dcode['G'] & 31u) << 24);

    uint32_t te_n = 0, se_n = 0, sl_n = 0;   // the next tile's record table entries are fetched one tile ahead
    if (tile_first * SR_THREADS + tid < n) {
        const uint32_t g0 = rec_lo + tile_first * SR_THREADS + tid;
        te_n = A.te[g0]; se_n = A.se[g0]; sl_n = A.slen[g0];
    }
    for (uint32_t tile = tile_first; tile < tile_last; ++tile) {
        const uint32_t i = tile * SR_THREADS + tid;
        const bool valid = i < n;
        const uint32_t g = rec_lo + (valid ? i : 0);
        const uint32_t te = te_n, se = se_n;
        const uint32_t L = valid ? se - te - 1 : 0;
        const uint32_t s0 = te + 1, q0 = se + 3;
        const uint32_t sl = valid ? sl_n : 0;
        if (tile + 1 < tile_last && i + SR_THREADS < n) { te_n = A.te[g + SR_THREADS]; se_n = A.se[g + SR_THREADS]; sl_n = A.slen[g + SR_THREADS]; }
        // A transferred record's quality codes differ from the table only at its (few) ambiguity codes.  With at most
        // four of them their positions come from k_sq_stats_reg and the lengths are corrected after the loop (with 2-bit
        // DNA the bases are then not read at all); otherwise the bases are scanned for them chunk by chunk.
        const uint32_t nall = (sl >> 28) & 7u;
        const bool xfer_d = sl >> 31;                 // for the DNA lengths: a transferred record drops its ambiguity codes
        const bool by_list = xfer_d && nall <= 4;
        const bool xfer = xfer_d && !by_list;         // for the quality lengths inside the chunk loop
        const bool need_s = __any(xfer);   // bases matter only where a record's transferred positions are not listed (more than four)
        uint32_t qb = 0, db = plain ? 2u * (sl & 0x0fffffffu) : 0u;
        // NB chunks are fetched back to back, see k_sq_stats_reg
        auto chunk = [&](const uint32_t ch, const uint4 S, const uint4 Q) __attribute__((always_inline)) {
            const uint32_t nv = L > 16 * ch ? (L - 16 * ch < 16 ? L - 16 * ch : 16) : 0;
            const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
            // ---- lean path (as in k_sq_pack_reg): the alphabet lies inside the table, two-bit bases (their bits are known from
            // the record's length), no record of the wave scans its bases for transferred positions, no byte with the high bit
            // set: sixteen straight-line lookups; bytes behind the read's end become the pad byte, which reads the zero column ----
            if (lean_wg && !__any(xfer) && !__any(((qw[0] | qw[1] | qw[2] | qw[3]) & 0x80808080u) != 0 && nv != 0)) {
                const bool tails = __any(nv != 16);
                const uint32_t rbase = (16 * ch + 1) * SQ_QS - 33u;
                uint32_t acc = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    uint32_t qm = qw[k];
                    if (tails) {
                        if (k && !__any(nv > 4 * k)) continue;   // a dword no lane of the wave needs (36 bases: 16 + 16 + 4)
                        const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                        qm = (qw[k] & vm) | (SQ_QPAD & ~vm);
                    }
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) acc += s_qlen[rbase + (4 * k + b) * SQ_QS + ((qm >> (8 * b)) & 0xffu)];
                }
                qb += acc;
                return;
            }
            uint32_t qbad = 0, dacc = 0, idx[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                qbad |= sr_q_out(qw[k]) & vm;
                const uint32_t w = (sw[k] & vm) | (0x41414141u & ~vm);
                idx[k] = sr_idx(w);
                dacc |= sr_not_acgt(w, idx[k]);
            }
            // per-byte exceptions as 16-bit masks (see k_sq_pack_reg, which makes the same decisions)
            const uint32_t validm = (1u << nv) - 1u;
            uint32_t ambm = 0, qbm = 0;
            if (__any((need_s && dacc != 0) || qbad != 0)) {
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                    const uint32_t w = (sw[k] & vm) | (0x41414141u & ~vm);
                    if (need_s) ambm |= t_movemask(t_nonzero01(sr_not_acgt(w, idx[k]))) << (4 * k);
                    qbm |= t_movemask((sr_q_out(qw[k]) >> 7) & 0x01010101u) << (4 * k);
                }
                qbm &= validm;
            }
            const uint32_t rows_left = lds_rows > 16 * ch + 1 ? lds_rows - (16 * ch + 1) : 0;
            const uint32_t rowm = rows_left >= 16 ? 0xffffu : ((1u << rows_left) - 1u);
            const uint32_t qpatch = (qbm | (xfer ? ambm : 0u) | (validm & ~rowm));
            // quality bits: table lengths of the (clamped) bytes, then corrections for the exceptions
            {
                const uint32_t row0 = 16 * ch + 1;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    if (!__any(nv > 4 * k)) continue;
                    uint32_t l[4];
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) {
                        const uint32_t u = 4 * k + b;
                        const uint32_t row = row0 + u < lds_rows ? row0 + u : lds_rows - 1;
                        l[b] = s_qlen[row * SQ_QS + ((sr_byte(qw, u) - 33u) & 63u)];
                    }
                    if (__any((qpatch >> (4 * k)) & 15u)) {
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t u = 4 * k + b;
                            if ((qpatch >> u) & 1u) {
                                uint32_t q = sr_byte(qw, u);
                                if (xfer && ((ambm >> u) & 1u)) q = transfer_quality(s_amb[sr_byte(sw, u)], (uint8_t)q);   // phyNGSC.cpp:579
                                l[b] = sq_qentry_slow(T, row0 + u, q) & 31u;
                            }
                        }
                    }
#pragma unroll
                    for (uint32_t b = 0; b < 4; ++b) qb += ((validm >> (4 * k + b)) & 1u) ? l[b] : 0u;
                }
            }
        };
        if constexpr (BURST) {
            for (uint32_t c4 = 0; __any(16 * NB * c4 < L); ++c4) {
                uint4 Sb[NB], Qb[NB];
#pragma unroll
                for (uint32_t k = 0; k < NB; ++k) {
                    const uint32_t off = 16 * NB * c4 + 16 * k;
                    Sb[k] = Qb[k] = make_uint4(0, 0, 0, 0);
                    if (L > off) {
                        Qb[k] = sq_chunk_load<BURST>(A, A.dense_q, g, off, q0);
                        if (need_s) Sb[k] = sq_chunk_load<BURST>(A, A.dense_s, g, off, s0);
                    }
                }
#pragma unroll
                for (uint32_t k = 0; k < NB; ++k)
                    if (k == 0 || __any(16 * NB * c4 + 16 * k < L)) chunk(NB * c4 + k, Sb[k], Qb[k]);
            }
        } else {
            uint4 Sn = make_uint4(0, 0, 0, 0), Qn = Sn;   // the next chunk is loaded while the current one is processed
            if (L) {
                if (need_s) Sn = sq_chunk_load<BURST>(A, A.dense_s, g, 0, s0);
                Qn = sq_chunk_load<BURST>(A, A.dense_q, g, 0, q0);
            }
            for (uint32_t ch = 0; __any(16 * ch < L); ++ch) {
                const uint4 S = Sn, Q = Qn;
                Sn = Qn = make_uint4(0, 0, 0, 0);
                if (L > 16 * ch + 16) {
                    if (need_s) Sn = sq_chunk_load<BURST>(A, A.dense_s, g, 16 * ch + 16, s0);
                    Qn = sq_chunk_load<BURST>(A, A.dense_q, g, 16 * ch + 16, q0);
                }
                chunk(ch, S, Q);
            }
        }
        if (__any(by_list && nall)) {
            AmbPos ap = {0, 0};
            if (by_list && nall) ap = A.apos[g];
            for (uint32_t k = 0; k < 4 && __any(by_list && nall > k); ++k) {
                if (by_list && nall > k) {
                    const uint32_t pos = ((k < 2 ? ap.x : ap.y) >> (16 * (k & 1))) & 0xffffu, row = pos + 1;
                    const uint32_t q = A.buf[q0 + pos], col = q - 33;
                    const uint32_t l_tab = (col < SR_HCOLS && row < lds_rows) ? s_qlen[row * SQ_QS + col] : (sq_qentry_slow(T, row, q) & 31u);
                    const uint32_t qt = transfer_quality(s_amb[A.buf[s0 + pos]], (uint8_t)q);   // phyNGSC.cpp:579
                    // (round 3) a transferred N - by far the usual ambiguity code - has its columns in the LDS table (SQ_NCOL0 ..): no
                    // two dependent global lookups per N, which 26 % of 150-base records carry at 0.2 % N
                    const uint32_t ncol = qt - 208u;
                    const uint32_t l_new = (ncol < 8u && row < lds_rows) ? s_qlen[row * SQ_QS + SQ_NCOL0 + ncol] : (sq_qentry_slow(T, row, qt) & 31u);
                    qb += l_new - l_tab;
                }
            }
        }
        // DNA bits of a Huffman-coded alphabet (more than four symbols, tasks.cpp:525-557): the counts of A / C / T / G the
        // statistics kernel left, times their code lengths; a record that keeps its ambiguity codes adds theirs (their
        // positions are listed, or - more than four - the bases are read after all)
        if (!plain) {
            const uint64_t bc = valid ? A.bcnt[g] : 0ull;
            db = (uint32_t)(bc & 0xffffu) * (ltab & 0xffu) + (uint32_t)((bc >> 16) & 0xffffu) * ((ltab >> 8) & 0xffu) +
                 (uint32_t)((bc >> 32) & 0xffffu) * ((ltab >> 16) & 0xffu) + (uint32_t)(bc >> 48) * (ltab >> 24);
            const bool keeps = valid && !xfer_d && nall != 0;
            if (__any(keeps)) {
                if (keeps && nall <= 4) {
                    const AmbPos ap = A.apos[g];
                    for (uint32_t k = 0; k < nall; ++k) {
                        const uint32_t pos = ((k < 2 ? ap.x : ap.y) >> (16 * (k & 1))) & 0xffffu;
                        db += s_dcode[A.buf[s0 + pos]] & 31u;
                    }
                } else if (keeps) {
                    for (uint32_t ch = 0; 16 * ch < L; ++ch) {
                        const uint32_t nv = L - 16 * ch < 16 ? L - 16 * ch : 16;
                        const uint4 S = sq_chunk_load<BURST>(A, A.dense_s, g, 16 * ch, s0);
                        const uint32_t sw[4] = {S.x, S.y, S.z, S.w};
                        const uint32_t ambm = sr_amb_mask(sw, nv);
#pragma unroll
                        for (uint32_t u = 0; u < 16; ++u)
                            if ((ambm >> u) & 1u) db += s_dcode[sr_byte(sw, u)] & 31u;
                    }
                }
            }
        }
        const size_t tix = (size_t)A.tile_lo[s] + tile;
        A.ubits[tix * SR_THREADS + tid] = qb | (db << 16);
        const uint32_t wq = sr_wsum(qb), wd = sr_wsum(db);
        if (lane == 0) { atomicAdd(&s_tot[0], wq); atomicAdd(&s_tot[1], wd); }
        __syncthreads();
        if (tid == 0) { A.wq[tix] = s_tot[0]; A.wd[tix] = s_tot[1]; s_tot[0] = 0; s_tot[1] = 0; }
        __syncthreads();
    }
}

// ---- pack ----
// NT: the chunk-major copies are read with non-temporal loads (long reads: ngsc_swar.h)
template <bool BURST, bool NT>
__global__ __launch_bounds__(SR_THREADS) __attribute__((amdgpu_waves_per_eu(4))) void k_sq_pack_reg(Arena A, uint32_t lds_rows, uint32_t sw_q, uint32_t sw_d, uint32_t tpw, uint32_t win)
{
    // win != 0 (long reads): the table holds one WINDOW of win positions (rows r0 .. r0 + win, lds_rows = win + 1), refilled
    // window after window for every tile - a [301][49] table plus staging left room for two workgroups per CU, a window
    // of 48 rows leaves room for four, and the lean path no longer needs the whole table to fit.
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_dynw[];   // [lds_rows][64] quality entries | staging (quality sw_q, DNA sw_d words)
    __shared__ uint32_t s_dcode[256];
    __shared__ uint8_t s_qcode[256], s_amb[256];
    __shared__ uint32_t s_part[SR_THREADS / 64];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t ntiles = (n + SR_THREADS - 1) / SR_THREADS;
    const uint32_t tile_first = blockIdx.x * tpw;
    if (tile_first >= ntiles) return;
    const uint32_t tile_last = tile_first + tpw < ntiles ? tile_first + tpw : ntiles;
    const int tid = threadIdx.x;
    uint32_t *s_qtab = s_dynw;
    uint32_t *st_q = s_dynw + (lds_rows + SQ_QROWS_EXTRA) * SQ_QS, *st_d = st_q + sw_q;

    s_dcode[tid] = A.dcodes[(size_t)s * 256 + tid];
    s_qcode[tid] = sb.qua_code[tid];
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (!win) sq_copy_qtab(A, s, s_qtab, lds_rows, 0);
    __syncthreads();
    SqTabs T = {A.qcodes + (size_t)s * A.qrows * 256, s_qcode, s_amb, s_dcode, lds_rows};
    const uint32_t n_win = win ? (sb.max_q + win - 1) / win : 1u, win_chunks = win ? win / 16 : ~0u;   // (uniform over the workgroup)
    const bool plain = (sb.flags & F_DNA_PLAIN) != 0;
    const uint32_t eA = s_dcode['A'], eC = s_dcode['C'], eT = s_dcode['T'], eG = s_dcode['G'], eN = s_dcode['N'];
    const uint32_t ltab = (eA & 31u) | ((eC & 31u) << 8) | ((eT & 31u) << 16) | ((eG & 31u) << 24);
    const uint32_t ctab = ((eA >> 5) & 0xffu) | (((eC >> 5) & 0xffu) << 8) | (((eT >> 5) & 0xffu) << 16) | (((eG >> 5) & 0xffu) << 24);
    // four Huffman base codes can be merged bytewise when each is at most 8 bits long
    // the byte-vector DNA path needs every code of the alphabet to fit a byte (length <= 8)
    __shared__ uint32_t s_longcode;
    if (tid == 0) s_longcode = 0;
    __syncthreads();
    if ((s_dcode[tid] & 31u) > 8) atomicOr(&s_longcode, 1u);
    __syncthreads();
    const bool short_codes = !s_longcode;
    __shared__ uint32_t s_leanflag;
    const bool lean_wg = sq_lean_alphabet(sb, lds_rows, &s_leanflag, win != 0);
    uint32_t *outw = reinterpret_cast<uint32_t *>(A.out);
    const uint64_t qbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_hdr) * 8;
    const uint64_t dbase = (sb.out_off + sb.info_len + sb.title_len + sb.qual_len + sb.dna_hdr) * 8;

    uint32_t te_n = 0, se_n = 0, sl_n = 0, ub_n = 0;   // the next tile's per-record entries are fetched one tile ahead
    if (tile_first * SR_THREADS + tid < n) {
        const uint32_t g0 = rec_lo + tile_first * SR_THREADS + tid;
        te_n = A.te[g0]; se_n = A.se[g0]; sl_n = A.slen[g0];
    }
    ub_n = A.ubits[((size_t)A.tile_lo[s] + tile_first) * SR_THREADS + tid];
    for (uint32_t tile = tile_first; tile < tile_last; ++tile) {
        __syncthreads();   // the previous tile's copy-out is done
        for (uint32_t k = tid; k < sw_q + sw_d; k += SR_THREADS) st_q[k] = 0;
        const bool last_tile = tile + 1 == ntiles;
        const size_t tix = (size_t)A.tile_lo[s] + tile;
        const uint64_t qb0 = qbase + A.wq[tix], qb1 = qbase + (last_tile ? sb.qbits : A.wq[tix + 1]);
        const uint64_t db0 = dbase + A.wd[tix], db1 = dbase + (last_tile ? sb.dbits : A.wd[tix + 1]);
        const uint64_t qw0 = qb0 >> 5, dw0 = db0 >> 5;
        const uint32_t qnw = (uint32_t)(((qb1 + 31) >> 5) - qw0), dnw = (uint32_t)(((db1 + 31) >> 5) - dw0);
        const bool q_lds = qnw + 1 <= sw_q, d_lds = dnw + 1 <= sw_d;   // else: straight into the zeroed output

        const uint32_t i = tile * SR_THREADS + tid;
        const bool valid = i < n;
        const uint32_t g = rec_lo + (valid ? i : 0);
        const uint32_t te = te_n, se = se_n;
        const uint32_t L = valid ? se - te - 1 : 0;
        const uint32_t s0 = te + 1, q0 = se + 3;
        const bool xfer = valid ? (sl_n >> 31) : false;
        const uint32_t ub = ub_n;
        if (tile + 1 < tile_last) {
            if (i + SR_THREADS < n) { te_n = A.te[g + SR_THREADS]; se_n = A.se[g + SR_THREADS]; sl_n = A.slen[g + SR_THREADS]; }
            ub_n = A.ubits[(tix + 1) * SR_THREADS + tid];
        }
        const uint32_t qo = sq_scan_excl256(ub & 0xffffu, s_part);
        const uint32_t dofs = sq_scan_excl256(ub >> 16, s_part);   // (its first barrier also orders the staging clear)
        SqWriter wq_, wd_;
        wq_.init(q_lds ? st_q : outw, q_lds ? qw0 : 0, qb0 + qo);
        wd_.init(d_lds ? st_d : outw, d_lds ? dw0 : 0, db0 + dofs);

        constexpr uint32_t NBX = BURST ? NB : 1u;   // chunks per trip, see k_sq_stats_reg
        uint4 Sn = make_uint4(0, 0, 0, 0), Qn = Sn;
        if (!BURST && L) {
            Sn = sq_chunk_load<NT>(A, A.dense_s, g, 0, s0);
            Qn = sq_chunk_load<NT>(A, A.dense_q, g, 0, q0);
        }
        for (uint32_t w = 0; w < n_win; ++w) {
        const uint32_t r0 = w * win;
        if (win) {   // (whole workgroup: n_win comes from the sub-block's longest read)
            __syncthreads();
            sq_copy_qtab(A, s, s_qtab, lds_rows, r0);
            __syncthreads();
        }
        for (uint32_t grp = w * (win_chunks / NBX); (!win || grp < (w + 1) * (win_chunks / NBX)) && __any(16 * NBX * grp < L); ++grp) {
            uint4 Sb[NBX], Qb[NBX];
            if constexpr (BURST) {
#pragma unroll
                for (uint32_t k = 0; k < NBX; ++k) {
                    const uint32_t off = 16 * NBX * grp + 16 * k;
                    Sb[k] = Qb[k] = make_uint4(0, 0, 0, 0);
                    if (L > off) {
                        Sb[k] = sq_chunk_load<NT>(A, A.dense_s, g, off, s0);
                        Qb[k] = sq_chunk_load<NT>(A, A.dense_q, g, off, q0);
                    }
                }
            } else {
                Sb[0] = Sn; Qb[0] = Qn;
                Sn = Qn = make_uint4(0, 0, 0, 0);
                if (L > 16 * grp + 16) {
                    Sn = sq_chunk_load<NT>(A, A.dense_s, g, 16 * grp + 16, s0);
                    Qn = sq_chunk_load<NT>(A, A.dense_q, g, 16 * grp + 16, q0);
                }
            }
#pragma unroll
            for (uint32_t kb = 0; kb < NBX; ++kb) {
                if (kb == 0 || __any(16 * (NBX * grp + kb) < L)) {
                    const uint32_t ch = NBX * grp + kb;
                    const uint32_t lrow0 = 16 * ch + 1 - r0;   // row of the chunk's first position inside the table
                    const uint4 S = Sb[kb], Q = Qb[kb];
                    const uint32_t nv = L > 16 * ch ? (L - 16 * ch < 16 ? L - 16 * ch : 16) : 0;
                    const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
                    // ---- lean path: alphabet inside the table (sq_lean_alphabet) and no high bit in this chunk's bytes ----
                    if (lean_wg && !__any(((qw[0] | qw[1] | qw[2] | qw[3]) & 0x80808080u) != 0 && nv != 0)) {
                        const bool tails = __any(nv != 16);
                        // dwords that some lane of the wave still needs (36 bases = 16 + 16 + 4: the last chunk is one dword)
                        const bool wany[4] = {true, !tails || __any(nv > 4), !tails || __any(nv > 8), !tails || __any(nv > 12)};
                        uint32_t vmk[4] = {~0u, ~0u, ~0u, ~0u}, idx[4] = {0, 0, 0, 0}, dacc = 0;
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            uint32_t w = sw[k];
                            if (tails) {
                                vmk[k] = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                                if (!wany[k]) continue;   // (idx = 0, 'A': no lane has a byte here)
                                w = (sw[k] & vmk[k]) | (0x41414141u & ~vmk[k]);
                            }
                            idx[k] = sr_idx(w);
                            dacc |= sr_not_acgt(w, idx[k]);
                        }
                        // Are the wave's non-ACGT bytes of this chunk all N?  Then nothing is an exception: a transferred N is a
                        // quality byte moved to the transferred-N columns and a base of length 0, a kept N a base with its own code
                        // (per-lane table halves), both straight-line.  Other ambiguity codes take the masked path below.
                        // ((w >> 1) & 7 maps A, C, T, G, N to 0, 1, 2, 3, 7: an eight-entry v_perm table holds all five)
                        bool n_only = false;
                        if (__any(dacc != 0)) {
                            uint32_t d5 = 0;
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k) {
                                if (!wany[k]) continue;
                                const uint32_t w = (sw[k] & vmk[k]) | (0x41414141u & ~vmk[k]);
                                d5 |= __builtin_amdgcn_perm(0x4E000000u, 0x47544341u, (w >> 1) & 0x07070707u) ^ w;
                            }
                            n_only = !__any(d5 != 0);
                        }
                        uint32_t ambm = 0, keep[4] = {vmk[0], vmk[1], vmk[2], vmk[3]};
                        if (__any(dacc != 0) && !n_only) {
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k) {
                                if (!wany[k]) continue;
                                const uint32_t w = (sw[k] & vmk[k]) | (0x41414141u & ~vmk[k]);
                                const uint32_t a01 = t_nonzero01(sr_not_acgt(w, idx[k]));
                                ambm |= t_movemask(a01) << (4 * k);
                                if (xfer) keep[k] &= ~(a01 * 0xffu);   // transferred qualities: patched below (phyNGSC.cpp:579)
                            }
                        }
                        const uint32_t qpatch = xfer ? ambm : 0u;
                        const bool masked = tails || __any(qpatch != 0);
                        // quality: 16 entries in flight together (bytes without an entry read the zero column)
                        const uint32_t rbase = lrow0 * SQ_QS - 33u;
                        uint32_t e[16];
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            if (!wany[k]) { e[4 * k] = e[4 * k + 1] = e[4 * k + 2] = e[4 * k + 3] = 0; continue; }
                            uint32_t qm = masked ? ((qw[k] & keep[k]) | (SQ_QPAD & ~keep[k])) : qw[k];
                            if (n_only) {   // q -> q + 49 where a transferred record holds an N: column 49 + (q - 33)
                                const uint32_t w = (sw[k] & vmk[k]) | (0x41414141u & ~vmk[k]);
                                qm += xfer ? t_nonzero01(sr_not_acgt(w, idx[k])) * 49u : 0u;
                            }
#pragma unroll
                            for (uint32_t b = 0; b < 4; ++b) e[4 * k + b] = s_qtab[rbase + (4 * k + b) * SQ_QS + ((qm >> (8 * b)) & 0xffu)];
                        }
                        if (__any(qpatch)) {
                            // all loads first, one wait (a load consumed inside its own predicated block is a memory round
                            // trip per patched position of the wave)
                            uint32_t t[16];
#pragma unroll
                            for (uint32_t u = 0; u < 16; ++u) {
                                t[u] = 0;
                                if ((qpatch >> u) & 1u) {
                                    const uint32_t q = transfer_quality(s_amb[sr_byte(sw, u)], (uint8_t)sr_byte(qw, u));
                                    t[u] = sq_qentry_slow(T, 16 * ch + 1 + u, q);
                                }
                            }
#pragma unroll
                            for (uint32_t u = 0; u < 16; ++u) e[u] = ((qpatch >> u) & 1u) ? t[u] : e[u];
                        }
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            if (!wany[k]) continue;
                            const uint32_t e0 = e[4 * k], e1 = e[4 * k + 1], e2 = e[4 * k + 2], e3 = e[4 * k + 3];
                            const uint32_t l1 = e1 & 31u, l2 = e2 & 31u, l3 = e3 & 31u;
                            const uint32_t tl = (e0 & 31u) + l1 + l2 + l3;
                            if (__any(tl > 32)) {
                                wq_.put(e0 >> 5, e0 & 31u); wq_.put(e1 >> 5, l1); wq_.put(e2 >> 5, l2); wq_.put(e3 >> 5, l3);
                            } else {
                                wq_.put(((((((e0 >> 5) << l1) | (e1 >> 5)) << l2) | (e2 >> 5)) << l3) | (e3 >> 5), tl);
                            }
                        }
                        // DNA (tasks.cpp:544-557)
                        if (plain && !__any(dacc != 0)) {
                            if (nv) {
                                uint32_t x = 0;
#pragma unroll
                                for (uint32_t k = 0; k < 4; ++k)
                                    if (wany[k]) x |= ((__builtin_amdgcn_perm(0u, ctab, idx[k]) * 0x40100401u) >> 24) << (24 - 8 * k);
                                wd_.put(nv == 16 ? x : (x >> (32 - 2 * nv)), 2 * nv);
                            }
                        } else if (short_codes) {
                            // (entry 7 of the eight-entry tables: N - nothing for a transferred record, which drops it)
                            const uint32_t lhi = xfer ? 0u : (eN & 31u) << 24, chi = xfer ? 0u : ((eN >> 5) & 0xffu) << 24;
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k) {
                                if (!__any(nv > 4 * k)) continue;
                                const uint32_t i7 = ((((sw[k] & vmk[k]) | (0x41414141u & ~vmk[k])) >> 1) & 0x07070707u);
                                uint32_t l4 = __builtin_amdgcn_perm(lhi, ltab, i7) & vmk[k], c4 = __builtin_amdgcn_perm(chi, ctab, i7) & vmk[k];
                                if (__any((ambm >> (4 * k)) & 15u)) {
#pragma unroll
                                    for (uint32_t b = 0; b < 4; ++b) {
                                        if ((ambm >> (4 * k + b)) & 1u) {
                                            const uint32_t ee = xfer ? 0u : s_dcode[sr_byte(sw, 4 * k + b)];
                                            l4 = (l4 & ~(0xffu << (8 * b))) | ((ee & 31u) << (8 * b));
                                            c4 = (c4 & ~(0xffu << (8 * b))) | ((ee >> 5) << (8 * b));
                                        }
                                    }
                                }
                                const uint32_t l1 = (l4 >> 8) & 0xffu, l2 = (l4 >> 16) & 0xffu, l3 = l4 >> 24;
                                const uint32_t c = (((((c4 & 0xffu) << l1) | ((c4 >> 8) & 0xffu)) << l2 | ((c4 >> 16) & 0xffu)) << l3) | (c4 >> 24);
                                wd_.put(c, (l4 & 0xffu) + l1 + l2 + l3);
                            }
                        } else {
#pragma unroll
                            for (uint32_t u = 0; u < 16; ++u) {
                                if (u < nv) {
                                    const uint32_t c = sr_byte(sw, u);
                                    if (!(xfer && s_amb[c] > 1)) { const uint32_t ee = s_dcode[c]; wd_.put(ee >> 5, ee & 31u); }
                                }
                            }
                        }
                    } else {
                    uint32_t qbad = 0, dacc = 0, idx[4];
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                        qbad |= sr_q_out(qw[k]) & vm;
                        const uint32_t w = (sw[k] & vm) | (0x41414141u & ~vm);
                        idx[k] = sr_idx(w);
                        dacc |= sr_not_acgt(w, idx[k]);
                    }
                    // Per-byte exceptions as 16-bit masks.  The common bytes go through the table / v_perm paths four at a time;
                    // a word (4 bytes) is patched byte by byte only if some lane of the wave has an exception in that word.
                    const uint32_t validm = (1u << nv) - 1u;
                    uint32_t ambm = 0, qbm = 0;
                    if (__any(dacc != 0 || qbad != 0)) {
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                            const uint32_t w = (sw[k] & vm) | (0x41414141u & ~vm);
                            ambm |= t_movemask(t_nonzero01(sr_not_acgt(w, idx[k]))) << (4 * k);
                            qbm |= t_movemask((sr_q_out(qw[k]) >> 7) & 0x01010101u) << (4 * k);
                        }
                        qbm &= validm;
                    }
                    const uint32_t rows_left = lds_rows > lrow0 ? lds_rows - lrow0 : 0;   // table rows for this chunk's positions
                    const uint32_t rowm = rows_left >= 16 ? 0xffffu : ((1u << rows_left) - 1u);
                    // quality bytes whose entry is not in the LDS table: odd bytes, transferred ones, rows past the table
                    const uint32_t qpatch = (qbm | (xfer ? ambm : 0u) | (validm & ~rowm));

                    // ---- quality (tasks.cpp:609-619): four symbols per put when their codes fit 32 bits together ----
                    {
                        const uint32_t row0 = 16 * ch + 1;
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            if (!__any(nv > 4 * k)) continue;
                            uint32_t e[4];
#pragma unroll
                            for (uint32_t b = 0; b < 4; ++b) {
                                const uint32_t u = 4 * k + b;
                                const uint32_t row = lrow0 + u < lds_rows ? lrow0 + u : lds_rows - 1;
                                e[b] = s_qtab[row * SQ_QS + ((sr_byte(qw, u) - 33u) & 63u)];
                            }
                            if (__any((qpatch >> (4 * k)) & 15u)) {
#pragma unroll
                                for (uint32_t b = 0; b < 4; ++b) {
                                    const uint32_t u = 4 * k + b;
                                    if ((qpatch >> u) & 1u) {
                                        uint32_t q = sr_byte(qw, u);
                                        if (xfer && ((ambm >> u) & 1u)) q = transfer_quality(s_amb[sr_byte(sw, u)], (uint8_t)q);   // phyNGSC.cpp:579
                                        e[b] = sq_qentry_slow(T, row0 + u, q);
                                    }
                                }
                            }
                            if (__any(nv < 4 * k + 4)) {   // read tails: bytes past the read contribute no bits
#pragma unroll
                                for (uint32_t b = 0; b < 4; ++b) e[b] = ((validm >> (4 * k + b)) & 1u) ? e[b] : 0u;
                            }
                            const uint32_t l1 = e[1] & 31u, l2 = e[2] & 31u, l3 = e[3] & 31u;
                            const uint32_t tl = (e[0] & 31u) + l1 + l2 + l3;
                            if (__any(tl > 32)) {
                                wq_.put(e[0] >> 5, e[0] & 31u); wq_.put(e[1] >> 5, l1); wq_.put(e[2] >> 5, l2); wq_.put(e[3] >> 5, l3);
                            } else {
                                wq_.put(((((((e[0] >> 5) << l1) | (e[1] >> 5)) << l2) | (e[2] >> 5)) << l3) | (e[3] >> 5), tl);
                            }
                        }
                    }

                    // ---- DNA (tasks.cpp:544-557) ----
                    if (plain && !__any(ambm != 0)) {
                        if (nv) {
                            // two bits per base: the four code bytes of a dword collapse with one multiply
                            uint32_t x = 0;
#pragma unroll
                            for (uint32_t k = 0; k < 4; ++k)
                                x = (x << 8) | ((__builtin_amdgcn_perm(0u, ctab, idx[k]) * 0x40100401u) >> 24);
                            wd_.put(nv == 16 ? x : (x >> (32 - 2 * nv)), 2 * nv);
                        }
                    } else if (short_codes) {
                        // code and length bytes of the four bases of a dword from v_perm tables; ambiguity codes are patched in
                        // from the byte table (or dropped: transferred records carry them in the quality stream)
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            if (!__any(nv > 4 * k)) continue;
                            const uint32_t vm = t_lowbytes32((int32_t)nv - 4 * (int32_t)k);
                            uint32_t l4 = __builtin_amdgcn_perm(0u, ltab, idx[k]) & vm, c4 = __builtin_amdgcn_perm(0u, ctab, idx[k]) & vm;
                            if (__any((ambm >> (4 * k)) & 15u)) {
#pragma unroll
                                for (uint32_t b = 0; b < 4; ++b) {
                                    if ((ambm >> (4 * k + b)) & 1u) {
                                        const uint32_t ee = xfer ? 0u : s_dcode[sr_byte(sw, 4 * k + b)];
                                        l4 = (l4 & ~(0xffu << (8 * b))) | ((ee & 31u) << (8 * b));
                                        c4 = (c4 & ~(0xffu << (8 * b))) | ((ee >> 5) << (8 * b));
                                    }
                                }
                            }
                            const uint32_t l1 = (l4 >> 8) & 0xffu, l2 = (l4 >> 16) & 0xffu, l3 = l4 >> 24;
                            const uint32_t c = (((((c4 & 0xffu) << l1) | ((c4 >> 8) & 0xffu)) << l2 | ((c4 >> 16) & 0xffu)) << l3) | (c4 >> 24);
                            wd_.put(c, (l4 & 0xffu) + l1 + l2 + l3);
                        }
                    } else {
#pragma unroll
                        for (uint32_t u = 0; u < 16; ++u) {
                            if (u < nv) {
                                const uint32_t c = sr_byte(sw, u);
                                if (!(xfer && s_amb[c] > 1)) { const uint32_t e = s_dcode[c]; wd_.put(e >> 5, e & 31u); }
                            }
                        }
                    }
                    }   // (general path)
                }
            }
        }
        }   // (windows)
        if (L) { wq_.finish(); wd_.finish(); }
        __syncthreads();
        if (q_lds) {
            for (uint32_t k = tid; k < qnw; k += SR_THREADS) {
                const uint64_t w = qw0 + k;
                const uint32_t v = st_q[k];
                if ((w << 5) >= qb0 && ((w + 1) << 5) <= qb1) outw[w] = v;
                else if (v) atomicOr(&outw[w], v);
            }
        }
        if (d_lds) {
            for (uint32_t k = tid; k < dnw; k += SR_THREADS) {
                const uint64_t w = dw0 + k;
                const uint32_t v = st_d[k];
                if ((w << 5) >= db0 && ((w + 1) << 5) <= db1) outw[w] = v;
                else if (v) atomicOr(&outw[w], v);
            }
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
// Records per workgroup: enough to amortise the table fill / histogram flush (1024), but never so many that the job
// has fewer than ~8 workgroups per CU - long reads mean few records per byte, and a 1024-record workgroup of
// 150-bp reads left the 256 CUs with four workgroups each.  (NGSC_SQ_RECORDS_PER_WG overrides.)
static uint32_t sr_records_per_wg(const Arena &A, uint32_t max_rec, uint32_t most = 1024, uint32_t min_wgs = 2048)
{
    // (NGSC_SQ_RECORDS_PER_WG: the tests make small inputs walk several tiles per workgroup with it)
    const char *e = getenv("NGSC_SQ_RECORDS_PER_WG");
    const uint32_t forced = e ? (uint32_t)atoi(e) : 0;
    if (forced) return forced < SR_THREADS ? SR_THREADS : forced / SR_THREADS * SR_THREADS;
    const uint64_t records = (uint64_t)max_rec * A.n_sb;   // upper estimate of the job's records
    // (most: measure and pack, which fill a code table per workgroup, gain another 3-6 % from 2048 / 4096 records per
    // workgroup on jobs large enough to keep 2048 workgroups: 0.388 -> 0.365 ms and 0.154 -> 0.143 ms on the 1 GB job)
    uint32_t rpw = most;
    while (rpw > SR_THREADS && records / rpw < min_wgs) rpw /= 2;
    return rpw;
}

// NB chunks fetched back to back (BURST) - measured per 400 MB: 150-base reads statistics 0.41 -> 0.25 ms and measure
// 0.39 -> 0.27 ms, 36..300-base reads 0.34 -> 0.31 and 0.33 -> 0.31 ms, 36-base reads unchanged; the pack kernel gets
// slower with it and keeps one chunk per trip.  Used when the longest read exceeds 64 bases.
static bool sr_burst(const Arena &A)
{
    return A.qrows - 1 > 16 * NB;   // reads that one trip of NB chunks does not cover anyway
}

// The staged form (see k_sq_stats_reg): for jobs whose 64-record spans fit 10 KB of LDS
// per wave (records of up to ~160 bytes: reads of up to ~50 bases), where three workgroups still share a CU.
static bool sr_staged(const Arena &A, uint32_t max_rec_bytes, uint32_t &span_cap)
{
    span_cap = (64u * max_rec_bytes + 48u + 1023u) & ~1023u;
    if (sr_burst(A) || !max_rec_bytes) return false;
    return span_cap <= (10u << 10);
}

void launch_sq_stats_reg(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t max_rec_bytes, uint32_t n_rec_total)
{
    uint32_t rows = A.qrows;
    const uint32_t max_rows = (60u << 10) / (SR_HSTRIDE * 4);
    if (rows > max_rows) rows = max_rows;
    uint32_t rpw = sr_records_per_wg(A, max_rec, 1024);
    dim3 grid((max_rec + rpw - 1) / rpw, A.n_sb);
    uint32_t span_cap = 0;
    const bool staged = sr_staged(A, max_rec_bytes, span_cap);
    // at least 24 KB, i.e. at most six workgroups per CU: with the eight that a short-read histogram (7.5 KB) allows, the
    // kernel is 3-4 % SLOWER (0.494 against 0.476 ms; measured with 8 / 16 / 24 KB of padding - the lanes' scattered loads
    // share the CU's 32 KB L1, and four to six workgroups' worth of them fit it better); below four it loses again
    size_t lds = ((size_t)rows * SR_HSTRIDE + 64) * 4;
    if (staged) lds = ((lds + 15) & ~(size_t)15) + (size_t)(SR_THREADS / 64) * span_cap;
    else if (lds < (24u << 10)) lds = 24u << 10;
    // (round 3) long reads whose whole table fits the packed LDS histogram: four lanes per record on staged spans.
    // Only where the reads are about equally long (the job's average record is at least 3/4 of its longest): a record's lanes
    // walk ceil(chunks / 4) trips whatever the other records of the wave need, and on 36..150-base reads the form lost 25 %
    // (0.39 -> 0.53 ms per 400 MB) where it gains 8 % on 150-base reads (0.324 -> 0.298; two and eight lanes per record: slower)
    const bool even = (uint64_t)A.buf_len / (n_rec_total ? n_rec_total : 1u) * 4 >= (uint64_t)max_rec_bytes * 3;
    if (!staged && sr_burst(A) && max_rec_bytes && even) {
        constexpr uint32_t lpr = 4;
        const uint32_t mcap = ((64u / lpr) * max_rec_bytes + 48u + 1023u) & ~1023u;
        const uint32_t nslots = 8 * ((A.qrows + 14) / 16);
        const size_t mlds = ((((size_t)nslots * SR_HSTRIDE + 64) * 4 + 15) & ~(size_t)15) + (size_t)(SR_THREADS / 64) * mcap;
        if (mlds <= (64u << 10)) {
            // records per workgroup: batches of SR_THREADS / lpr
            const uint32_t rpb = SR_THREADS / lpr;
            uint32_t mrpw = rpw < rpb ? rpb : rpw / rpb * rpb;
            dim3 mgrid((max_rec + mrpw - 1) / mrpw, A.n_sb);
            auto mk = k_sq_stats_multi<lpr>;
            if (mlds > (48u << 10)) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(mk), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
            hipLaunchKernelGGL(mk, mgrid, dim3(SR_THREADS), mlds, st, A, A.qrows, mrpw, mcap);
            return;
        }
    }
    // (round 4) unevenly long reads of up to 16 FK bases: the chunks of 16-record batches dealt to the lanes on staged spans
    // (NGSC_SQ_FLAT=0: the burst form, which reads longer than that and jobs whose table does not fit LDS take anyway)
    if (!staged && sr_burst(A) && max_rec_bytes && !even && A.qrows - 1 <= 16 * FK &&
        !(getenv("NGSC_SQ_FLAT") && !strcmp(getenv("NGSC_SQ_FLAT"), "0"))) {
        // 256 threads with 16 records per wave and batch, or - where that would leave a CU with two workgroups = two waves per SIMD
        // (a [301]-row histogram beside four 11 KB spans) - 512 threads with 8: eight smaller spans beside one histogram, 16 waves per CU
        const uint32_t nslots = 8 * ((A.qrows + 14) / 16);
        const size_t hbytes = (((size_t)nslots * SR_HSTRIDE + 64) * 4 + 15) & ~(size_t)15;
        const uint32_t cap16 = (16u * max_rec_bytes + 64u + 15u) & ~15u, cap8 = (8u * max_rec_bytes + 64u + 15u) & ~15u;   // (+ the alignment of the span's start and the reach of the last chunk's dword reads)
        const size_t lds16 = hbytes + 4 * (size_t)cap16, lds8 = hbytes + 8 * (size_t)cap8;
        const bool wide_wg = lds16 > (50u << 10) && lds8 <= (76u << 10);
        const size_t flds = wide_wg ? lds8 : lds16;
        if (flds <= (76u << 10)) {
            const uint32_t rpb = wide_wg ? 8u * 8u : 4u * 16u;   // records per workgroup round
            uint32_t frpw = rpw < rpb ? rpb : rpw / rpb * rpb;
            dim3 fgrid((max_rec + frpw - 1) / frpw, A.n_sb);
            auto fk = wide_wg ? k_sq_stats_flat<512, 8> : k_sq_stats_flat<256, 16>;
            if (flds > (48u << 10)) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
            hipLaunchKernelGGL(fk, fgrid, dim3(wide_wg ? 512 : 256), flds, st, A, A.qrows, frpw, wide_wg ? cap8 : cap16);
            return;
        }
    }
    // A histogram of 40 KB and more (reads beyond 200 bases) leaves room for two or three workgroups per CU, and every workgroup
    // clears and flushes all of it: one and a half rounds of resident workgroups are enough then (36..300-base reads, 400 MB:
    // 256 / 512 / 1024 / 2048 records per workgroup 0.620 / 0.566 / 0.537 / 0.635 ms - the rule above chose 256)
    if (!staged && lds >= (40u << 10)) {
        const uint32_t resident = 256u * (uint32_t)((160u << 10) / lds);
        rpw = sr_records_per_wg(A, max_rec, 1024, resident + resident / 2);
        grid = dim3((max_rec + rpw - 1) / rpw, A.n_sb);
    }
    // 512 threads per workgroup for the staged form: eight spans beside one histogram, 80 KB, two workgroups = four waves per SIMD
    // instead of three (44 KB x 3): +1 % on the step (496-500 against 482-492 GB/s)
    if (staged) {
        const size_t lds5 = ((((size_t)rows * SR_HSTRIDE + 64) * 4 + 15) & ~(size_t)15) + (size_t)8 * span_cap;
        const uint32_t rpw5 = rpw < 512 ? 512 : rpw / 512 * 512;
        dim3 grid5((max_rec + rpw5 - 1) / rpw5, A.n_sb);
        auto k5 = k_sq_stats_reg<false, true, 512>;
        if (lds5 > (48u << 10)) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k5), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
        hipLaunchKernelGGL(k5, grid5, dim3(512), lds5, st, A, rows, rpw5, span_cap);
        return;
    }
    auto kern = sr_burst(A) ? k_sq_stats_reg<true, false> : k_sq_stats_reg<false, false>;
    if (lds > (48u << 10)) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
    hipLaunchKernelGGL(kern, grid, dim3(SR_THREADS), lds, st, A, rows, rpw, span_cap);
}

// Arena::qfmt: per sub-block a dword table (a multiple of 4 dwords) and, behind all of those, a byte table (16-byte multiples)
size_t sq_qfmt_words(uint32_t qrows) { return (((size_t)qrows + SQ_QROWS_EXTRA) * SQ_QS + 3) & ~(size_t)3; }
size_t sq_qfmt_bytes(uint32_t qrows, uint32_t n_sb) { return (size_t)n_sb * (4 * sq_qfmt_words(qrows) + sq_qfmt8_stride(qrows)); }

static uint32_t sr_rows_for(uint32_t qrows, uint32_t entry_bytes, uint32_t budget)
{
    const uint32_t max_rows = budget / (SQ_QS * entry_bytes) - SQ_QROWS_EXTRA;
    return qrows < max_rows ? qrows : max_rows;
}

void launch_sq_measure_reg(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    const uint32_t rows = sr_rows_for(A.qrows, 1, 32u << 10);
    const uint32_t tpw = sr_records_per_wg(A, max_rec, 2048) / SR_THREADS;
    dim3 grid(((max_rec + SR_THREADS - 1) / SR_THREADS + tpw - 1) / tpw, A.n_sb);
    auto kern = sr_burst(A) ? k_sq_measure_reg<true> : k_sq_measure_reg<false>;
    hipLaunchKernelGGL(kern, grid, dim3(SR_THREADS), (((size_t)(rows + SQ_QROWS_EXTRA) * SQ_QS + 15) & ~(size_t)15), st, A, rows, tpw);
}

// the quality code tables in their LDS layouts, once per sub-block: behind the Huffman kernel, ahead of both readers (measure, pack)
void launch_sq_format_qtab(hipStream_t st, const Arena &A)
{
    const uint32_t words = (uint32_t)(A.qrows + SQ_QROWS_EXTRA) * SQ_QS;
    hipLaunchKernelGGL(k_sq_format_qtab, dim3((words + 4 * SR_THREADS - 1) / (4 * SR_THREADS), A.n_sb), dim3(SR_THREADS), 0, st, A);
}

// Long reads: a table of one window of 32 positions once the whole table would be well beyond a window (reads of 49 bases
// and more; 32: 0.524 -> 0.453 ms per 400 MB of 36..300-base reads against 48, with the staging above)
static uint32_t sr_pack_window(const Arena &A)
{
    const uint32_t win = 32;
    return A.qrows > win + 1 + 16 ? win : 0;
}

void launch_sq_pack_reg(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t sw_q, uint32_t sw_d)
{
    const uint32_t win = sr_pack_window(A);
    const uint32_t rows = win ? win + 1 : sr_rows_for(A.qrows, 4, 48u << 10);
    if (sw_q > 8192) sw_q = 8192;
    if (sw_d > 6144) sw_d = 6144;
    const uint32_t tpw = sr_records_per_wg(A, max_rec, 4096) / SR_THREADS;
    dim3 grid(((max_rec + SR_THREADS - 1) / SR_THREADS + tpw - 1) / tpw, A.n_sb);
    const size_t lds = ((size_t)(rows + SQ_QROWS_EXTRA) * SQ_QS + sw_q + sw_d) * 4;
    // one chunk per trip: with NB chunks fetched back to back this kernel is 8-10 % slower on every configuration (it
    // is bound by its bit writers, not by the loads), so only that variant is built
    // (measured again with the chunks coming from the chunk-major copies: NB chunks per trip change nothing on 150-base and
    // mixed reads - 0.340 against 0.334 ms, 0.530 / 0.530 - and cost 16 % at 36 bases)
    // (non-temporal loads of the copies where the reads are long: 150-base reads 0.282 -> 0.262 ms per 400 MB, 36..300-base reads
    // 0.520 -> 0.478; at 36 bases the kernel gains nothing and the step loses 0.5 %)
    auto kern = sr_burst(A) ? k_sq_pack_reg<false, true> : k_sq_pack_reg<false, false>;
    if (lds > (48u << 10)) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10);
    // (the tables: written by launch_sq_measure_reg's first kernel)
    hipLaunchKernelGGL(kern, grid, dim3(SR_THREADS), lds, st, A, rows, sw_q, sw_d, tpw, win);
}

}  // namespace ngsc
