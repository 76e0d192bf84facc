// kernels_sq.hip - register-streaming sequence / quality kernels: one lane per record, the record's bases and
// qualities read straight from global memory in unaligned 16-byte chunks (neighbouring lanes read neighbouring
// records, so every cache line that comes in is consumed by the wave that asked for it) and processed four
// bytes per instruction where the format allows it.  No LDS tile, no barriers inside the record loop.
//
//   k_sq_stats_reg   N-transfer test, DNA symbol counts, per-position quality histogram   (phyNGSC.cpp:462-653,
//                                                                                         tasks.cpp:260-286)
//
// The LDS-tiled kernels of kernels_tile.hip spent ~23 VALU instructions per input byte and wave on per-byte
// index arithmetic; here the common case (all lanes inside their read, only A/C/G/T, qualities '!'..'`') costs
// ~3: bases are classified with one v_perm_b32 per dword and counted with popcounts of two bit planes, a
// quality byte costs a bit-field extract, a shift-add and one ds_add_u32 with an immediate row offset.
// Everything else (read tails, ambiguity codes, odd quality bytes, rows past the LDS histogram) takes a
// masked general path chosen per 16-byte chunk with a wave-uniform branch.
#include "ngsc_kernels.h"
#include "ngsc_swar.h"

namespace ngsc {

constexpr int SR_THREADS = 256;
constexpr uint32_t SR_HSTRIDE = 65;   // dwords per LDS histogram row: quality bytes 33..96, one pad column
constexpr uint32_t SR_HCOLS = 64;

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
// 0x80 per byte where the quality byte is outside 33..96 (the columns of the LDS histogram)
__device__ __forceinline__ uint32_t sr_q_out(uint32_t w)
{
    const uint32_t t1 = (w | 0x80808080u) - 0x21212121u;   // bit 7 <=> (b & 0x7f) >= 33
    const uint32_t t2 = (w & 0x7F7F7F7Fu) + 0x1F1F1F1Fu;   // bit 7 <=> (b & 0x7f) >= 97
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

__global__ __launch_bounds__(SR_THREADS) void k_sq_stats_reg(Arena A, uint32_t lds_rows, uint32_t rpw)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_hist[];   // [lds_rows][SR_HSTRIDE]
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

    for (uint32_t k = tid; k < hwords; k += SR_THREADS) s_hist[k] = 0;
    s_occ[tid] = 0;
    s_amb[tid] = (uint8_t)amb_code((uint8_t)tid);
    if (tid < 3) s_red[tid] = 0;
    __syncthreads();

    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    uint8_t *hb = reinterpret_cast<uint8_t *>(s_hist);
    uint32_t t0 = 0, t1 = 0, t01 = 0, nok = 0, mq = 0, ms = 0, sum_l = 0;   // per-lane totals over this workgroup's records
    const uint32_t lj = lane & 3;                                            // rotation of the quality bytes: word ...
    uint32_t bsh[4], boff[4];                                                // ... and byte inside the word
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const uint32_t bb = (b + (lane >> 2)) & 3;
        bsh[b] = 8 * bb;
        boff[b] = bb * SR_HSTRIDE * 4;
    }

    for (uint32_t base = first; base < last; base += SR_THREADS) {
        const uint32_t i = base + tid;
        const bool valid = i < last;
        const uint32_t g = rec_lo + (valid ? i : first);
        const uint32_t te = A.te[g], se = A.se[g];
        const uint32_t L = valid ? se - te - 1 : 0;
        const uint32_t s0 = te + 1, q0 = se + 3;
        if (valid) {
            // line 3 of every record must be a bare '+' (README.md:97-103); the reference assumes it
            if (A.buf[se + 1] != '+' || A.buf[se + 2] != '\n') atomicMin(A.err, (int32_t)ERR_FORMAT);
            if (i == 0 && L >= 2) {   // colour-space detection on the sub-block's first record, phyNGSC.cpp:478
                const uint8_t c1 = A.buf[s0 + 1];
                if (c1 >= '0' && c1 <= '3') atomicMin(A.err, (int32_t)ERR_COLORSPACE);
            }
        }

        // ---- one pass over the 16-byte chunks: everything except the ambiguous positions ----
        uint32_t fl = 0, namb = 0;   // N-transfer test (phyNGSC.cpp:549-571): bit0 some ambiguity code, bit1 a failure
        for (uint32_t ch = 0; __any(16 * ch < L); ++ch) {
            const uint32_t nv = L > 16 * ch ? (L - 16 * ch < 16 ? L - 16 * ch : 16) : 0;
            uint4 S = make_uint4(0, 0, 0, 0), Q = S;
            if (nv) {
                S = load16_unaligned(A.buf, s0 + 16 * ch, A.buf_len);
                Q = load16_unaligned(A.buf, q0 + 16 * ch, A.buf_len);
            }
            const uint32_t sw[4] = {S.x, S.y, S.z, S.w}, qw[4] = {Q.x, Q.y, Q.z, Q.w};
            uint32_t d[4], idx[4], dacc = 0, qbad = 0, c0 = 0, c1 = 0, c01 = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
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
            } else {
                uint32_t ambm = 0, qbm = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
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
                if (__any(okm)) {
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const uint32_t j = (k + lj) & 3;
                        const uint32_t wsel = j == 0 ? qw[0] : (j == 1 ? qw[1] : (j == 2 ? qw[2] : qw[3]));
                        const uint32_t joff = rb + j * (4 * SR_HSTRIDE * 4);
#pragma unroll
                        for (uint32_t b = 0; b < 4; ++b) {
                            const uint32_t q = (wsel >> bsh[b]) & 0xffu;
                            if ((okm >> (4 * j + (bsh[b] >> 3))) & 1u) atomicAdd(reinterpret_cast<uint32_t *>(hb + joff + boff[b] + 4 * q), 1u);
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
                            if (a == 0) fl |= 2;
                            else {
                                const uint32_t q = sr_byte(qw, u);
                                fl |= (q < 33 || q > 40) ? 2u : 1u;
                                ++namb;
                            }
                        }
                    }
                }
            }
        }
        const bool xfer = (fl & 1) && !(fl & 2);
        const uint32_t kept = L - (xfer ? namb : 0);
        if (valid) A.slen[g] = kept | ((uint32_t)xfer << 31);
        mq = L > mq ? L : mq;
        ms = kept > ms ? kept : ms;
        sum_l += L;

        // ---- the ambiguous positions, now that the record's transfer decision is known (phyNGSC.cpp:573-599) ----
        if (__any(fl != 0)) {
            for (uint32_t ch = 0; __any(fl != 0 && 16 * ch < L); ++ch) {
                if (fl != 0 && 16 * ch < L) {
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
    for (uint32_t k = tid; k < lds_rows * SR_HCOLS; k += SR_THREADS) {
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

// ---- launchers ---------------------------------------------------------------------------------------
static uint32_t sr_records_per_wg()
{
    static const uint32_t v = getenv("NGSC_SQ_RECORDS_PER_WG") ? (uint32_t)atoi(getenv("NGSC_SQ_RECORDS_PER_WG")) : 1024;
    return v < SR_THREADS ? SR_THREADS : v;
}

void launch_sq_stats_reg(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    uint32_t rows = A.qrows;
    const uint32_t max_rows = (60u << 10) / (SR_HSTRIDE * 4);
    if (rows > max_rows) rows = max_rows;
    const uint32_t rpw = sr_records_per_wg();
    dim3 grid((max_rec + rpw - 1) / rpw, A.n_sb);
    const size_t lds = (size_t)rows * SR_HSTRIDE * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_sq_stats_reg), hipFuncAttributeMaxDynamicSharedMemorySize, 64 << 10);
    hipLaunchKernelGGL(k_sq_stats_reg, grid, dim3(SR_THREADS), lds, st, A, rows, rpw);
}

}  // namespace ngsc
