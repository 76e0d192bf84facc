// kernels_stats.hip - per-sub-block statistics of the FASTQ compressor (batched over sub-blocks:
// blockIdx.y = sub-block).  The title reductions live in kernels_title.hip; k_seq_qual_stats below is the
// untiled fallback of k_seq_qual_stats_tile for records longer than the LDS tile budget.
//
//   k_seq_qual_stats   DNA/quality prepass (phyNGSC.cpp:462-653) + AnalyzeQuality (tasks.cpp:260-286):
//                      N-transfer decision, DNA symbol counts, per-position quality histograms staged
//                      in LDS (row stride 65 words so the 64 lanes of one record hit distinct banks)
//   k_sq_finalize      alphabets, codes, flags (phyNGSC.cpp:659-687, tasks.cpp:226-257)
//   k_field_finalize   per-field coding decisions (tasks.cpp:196-222, 324-338)
//   k_title_hist       value / delta / per-position character histograms for the title Huffman trees
//
// All of it is integer, branchy, HBM/LDS-bound work: no MFMA.
#include "ngsc_kernels.h"
#include "ngsc_swar.h"

namespace ngsc {

constexpr int TS_THREADS = 256;
constexpr int TS_OWNED = TS_THREADS - 1;  // thread 0 of a workgroup re-parses the previous record (delta halo)

__device__ __forceinline__ void lds_max(uint32_t *p, uint32_t v) { if (v > *p) atomicMax(p, v); }
__device__ __forceinline__ void lds_or(uint32_t *p, uint32_t v) { if ((*p & v) != v) atomicOr(p, v); }

// ---------------------------------------------------------------------------------------------------
constexpr int SQ_THREADS = 256;
constexpr int SQ_WAVES = SQ_THREADS / 64;
constexpr uint32_t SQ_REC_PER_WG = 1024;
constexpr uint32_t HIST_COLS = 64;           // quality bytes '!' (33) .. '`' (96) are staged in LDS
constexpr uint32_t HIST_STRIDE = 65;         // odd stride: lanes of one record (consecutive rows) spread over banks
constexpr uint32_t HIST_MAX_ROWS = 360;      // 360*65*4 = 93.6 KB

__device__ __forceinline__ uint32_t amb_fast(uint8_t c)
{
    return (c == 'A' || c == 'C' || c == 'G' || c == 'T') ? 1u : amb_code(c);
}

__global__ __launch_bounds__(SQ_THREADS) void k_seq_qual_stats(Arena A, uint32_t lds_rows)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint32_t s_hist[];  // [lds_rows][HIST_STRIDE]
    __shared__ uint32_t s_occ[256];
    __shared__ uint32_t s_max[3];
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    const uint32_t r_begin = blockIdx.x * SQ_REC_PER_WG;
    if (r_begin >= n) return;
    const uint32_t r_end = r_begin + SQ_REC_PER_WG < n ? r_begin + SQ_REC_PER_WG : n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint8_t *buf = A.buf;

    for (uint32_t k = tid; k < lds_rows * HIST_STRIDE; k += SQ_THREADS) s_hist[k] = 0;
    s_occ[tid] = 0;
    if (tid < 3) s_max[tid] = 0;
    __syncthreads();

    uint32_t cA = 0, cC = 0, cG = 0, cT = 0, wmax_q = 0, wmax_s = 0, wbases = 0;
    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    for (uint32_t r = r_begin + wave; r < r_end; r += SQ_WAVES) {
        const uint32_t g = rec_lo + r;
        const uint32_t t = A.te[g], e = A.se[g];
        const uint32_t L = e - t - 1, s0 = t + 1, q0 = e + 3;
        if (lane == 0 && (buf[e + 1] != '+' || buf[e + 2] != '\n')) atomicMin(A.err, (int32_t)ERR_FORMAT);  // bare '+' line
        if (r == 0 && lane == 0 && L >= 1 && !A.cs_mode) {   // colour-space detection, phyNGSC.cpp:478
            if (cs_first_read(buf[s0], buf[s0 + 1])) atomicMin(A.err, (int32_t)ERR_COLORSPACE);
        }
        // pass 1: transfer test (phyNGSC.cpp:549-571).  The serial loop breaks at the first failure, so
        // the outcome is "some ambiguous base" && "no failure anywhere" - order independent.
        bool any = false, bad = false;
        for (uint32_t base = 0; base < L; base += 64) {
            uint32_t j = base + lane;
            if (j < L) {
                uint8_t c = buf[s0 + j];
                uint32_t a = amb_fast(c);
                if (a == 0) bad = true;
                else if (a > 1) {
                    uint8_t q = buf[q0 + j];
                    if (q < 33 || q > 40) bad = true; else any = true;
                }
            }
        }
        const bool xfer = __any(any) && !__any(bad);
        // pass 2: counts
        uint32_t kept = 0;
        for (uint32_t base = 0; base < L; base += 64) {
            uint32_t j = base + lane;
            bool act = j < L;
            uint8_t c = 0, q = 0;
            uint32_t a = 1;
            if (act) { c = buf[s0 + j]; q = buf[q0 + j]; a = amb_fast(c); }
            bool drop = xfer && a > 1;
            bool keep = act && !drop;
            if (act) {
                uint32_t qv = drop ? transfer_quality(a, q) : q;  // phyNGSC.cpp:579
                uint32_t row = j + 1, col = qv - 33;
                if (col < HIST_COLS && row < lds_rows) atomicAdd(&s_hist[row * HIST_STRIDE + col], 1u);
                else atomicAdd(&qh[(size_t)row * 256 + qv], 1u);
            }
            cA += __popcll(__ballot(keep && c == 'A'));
            cC += __popcll(__ballot(keep && c == 'C'));
            cG += __popcll(__ballot(keep && c == 'G'));
            cT += __popcll(__ballot(keep && c == 'T'));
            if (keep && a != 1) atomicAdd(&s_occ[c], 1u);
            kept += __popcll(__ballot(keep));
        }
        if (lane == 0) A.slen[g] = kept | ((uint32_t)xfer << 31);
        wbases += L;
        wmax_q = L > wmax_q ? L : wmax_q;
        wmax_s = kept > wmax_s ? kept : wmax_s;
    }
    if (lane == 0) {
        atomicAdd(&s_occ['A'], cA); atomicAdd(&s_occ['C'], cC); atomicAdd(&s_occ['G'], cG); atomicAdd(&s_occ['T'], cT);
        atomicMax(&s_max[0], wmax_q); atomicMax(&s_max[1], wmax_s); atomicAdd(&s_max[2], wbases);
    }
    __syncthreads();
    for (uint32_t k = tid; k < lds_rows * HIST_COLS; k += SQ_THREADS) {
        uint32_t row = k / HIST_COLS, col = k % HIST_COLS;
        uint32_t v = s_hist[row * HIST_STRIDE + col];
        if (v) atomicAdd(&qh[(size_t)row * 256 + 33 + col], v);
    }
    if (s_occ[tid]) atomicAdd(&A.dna_occ[(size_t)s * 256 + tid], s_occ[tid]);
    if (tid == 0) { atomicMax(&A.sb[s].max_q, s_max[0]); atomicMax(&A.sb[s].max_s, s_max[1]); atomicAdd(&A.sb[s].n_bases, s_max[2]); }
}

// One workgroup of 256 threads per sub-block; thread = byte value.
// The column sums over the sub-block's rows with eight loads in flight per thread (one load at a time walked the 300 rows of
// 36..300-base reads in 95 us, on the chain statistics -> trees -> measure with nothing beside it).  256 threads: a 1024-thread
// form with the rows in four parts was faster alone and SLOWER in the step - beside the title statistics kernel a workgroup of
// sixteen waves waited 200-300 us for a CU with room (kernel trace, profiles/r03_final_kernel_stats.csv history).
// (round 4, measured and not kept: eight workgroups per sub-block adding up every eighth row each - atomics into row 0 - with the
// last one to finish going on.  The release fence in front of the "I am done" counter writes the XCD's whole L2 back on this chip:
// the kernel went from 0.011 to 0.092 ms and the 36-base step from 1.870 to 2.109; profiles/r04_ab_finalize_split_not_kept.txt)
// small_only (a predicted step whose context's last job had no alphabet of more than 64 symbols): the 512-symbol tree instance is not
// launched at all; a sub-block that needs it after all is a missed prediction (ERR_SHAPE: the step runs again, with it).
__global__ __launch_bounds__(256) void k_sq_finalize(Arena A, uint32_t small_only)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SbState &sb = A.sb[s];
    __shared__ uint32_t s_cnt[2][4];
    uint32_t *qh = A.qhist + (size_t)s * A.qrows * 256;
    const uint32_t max_q = sb.max_q;
    uint32_t tot = 0;
    {
        // (round 4: thirty-two loads in flight per thread instead of eight - the kernel is nothing but their latency, 19 rounds of
        // it for 150-base reads: 63 us at the head of the side chain)
        constexpr uint32_t NF = 32;
        uint32_t t[NF], row = 1;
#pragma unroll
        for (uint32_t k = 0; k < NF; ++k) t[k] = 0;
        for (; row + NF - 1 <= max_q; row += NF) {
            uint32_t v[NF];
#pragma unroll
            for (uint32_t k = 0; k < NF; ++k) v[k] = qh[(size_t)(row + k) * 256 + tid];
#pragma unroll
            for (uint32_t k = 0; k < NF; ++k) t[k] += v[k];
        }
        for (; row + 7 <= max_q; row += 8) {
#pragma unroll
            for (uint32_t k = 0; k < 8; ++k) t[k] += qh[(size_t)(row + k) * 256 + tid];
        }
        for (; row <= max_q; ++row) t[0] += qh[(size_t)row * 256 + tid];
#pragma unroll
        for (uint32_t k = 0; k < NF; ++k) tot += t[k];
    }
    qh[tid] = tot;  // row 0: position-independent histogram (quality_stats[0], tasks.cpp:283)
    const bool hasq = tot != 0;
    const bool hass = A.dna_occ[(size_t)s * 256 + tid] != 0;
    unsigned long long mq = __ballot(hasq), ms = __ballot(hass);
    if (lane == 0) { s_cnt[0][wave] = __popcll(mq); s_cnt[1][wave] = __popcll(ms); }
    __syncthreads();
    uint32_t bq = 0, bs = 0;
    for (int w = 0; w < wave; ++w) { bq += s_cnt[0][w]; bs += s_cnt[1][w]; }
    unsigned long long below = (1ull << lane) - 1ull;
    uint32_t iq = bq + __popcll(mq & below), is = bs + __popcll(ms & below);
    // alphabets in ascending byte order, codes = rank (phyNGSC.cpp:674-686)
    sb.qua_code[tid] = hasq ? (uint8_t)iq : 0;
    sb.sym_code[tid] = hass ? (uint8_t)is : 0;
    if (hasq) sb.qualities[iq] = (uint8_t)tid;
    if (hass) sb.symbols[is] = (uint8_t)tid;
    const uint32_t nqual = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
    const uint32_t nsym = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
    // flags: phyNGSC.cpp:433-435, 645-667 (VARIABLE_LENGTH and DELTA_CONSTANT always end up set, Q1);
    // AnalyzeDNA (tasks.cpp:226-257) keeps DNA_PLAIN iff <= 4 symbols (its skew test compares zeros, Q4)
    uint32_t flags = F_PLUS_ONLY | F_CONST_NUM_FIELDS | F_VARIABLE_LENGTH;
    if (!(sb.cs & 4)) flags |= F_DELTA_CONSTANT;
    if (sb.cs & 1) flags |= F_USE_DELTA;
    if (sb.cs & 2) flags |= F_DELTA_NO_BEGIN_NUC;
    if (nsym <= 4) flags |= F_DNA_PLAIN;
    // DNA code LUT by raw byte; the Huffman case is overwritten by the tree kernel
    A.dcodes[(size_t)s * 256 + tid] = hass ? ((is << 5) | 2u) : 0u;
    if (tid == 0) {
        sb.nqual = nqual;
        sb.nsym = nsym;
        sb.flags = flags;
        sb.lt_bits = bit_length(max_q);
        if (!nqual || !nsym) atomicMin(A.err, (int32_t)ERR_FORMAT);
        if (nqual > SMALL_TREE || nsym > SMALL_TREE) {
            atomicOr(A.big_trees, 1u);
            if (small_only) atomicMin(A.err, (int32_t)ERR_SHAPE);
        }
    }
}

// One thread per (sub-block, field).
// quiet (a predicted step whose context's last job had no title field with a histogram or a tree to build): the title's histogram and
// tree kernels are not launched at all; a field that needs them after all is a missed prediction (ERR_SHAPE: the step runs again).
__global__ __launch_bounds__(64) void k_field_finalize(Arena A, uint32_t quiet)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.x, f = threadIdx.x;
    const SbState &sb = A.sb[s];
    if (f >= sb.nf || f >= MAX_FIELDS) return;
    const uint8_t *buf = A.buf;
    // record 0's field f: from the table k_rec0_fields left (no byte-serial walk over the title: up to fifty dependent loads
    // per thread, 14 us alone and 75 us beside a throughput kernel, on the title chain's critical path)
    const uint32_t p = A.ls[sb.rec_lo] + A.rec0[s].start[f], e = p + A.rec0[s].len[f];
    FieldDesc d;
    for (int k = 0; k < 3; ++k) d.pad[k] = 0;
    d.cand = A.rec0[s].kmap[f];
    int rc = finalize_field(A.fstat[(size_t)s * A.fcap + f], p, e - p, buf[e], sb.n_rec, f, sb.nf, d);
    if (rc == OK && d.kind == KIND_TEXT) {
        uint32_t slot = atomicAdd(A.slot_counter, 1u);
        if (slot >= A.n_slots) rc = ERR_UNSUPPORTED;
        else { d.slot = slot; A.slot_owner[slot] = s * A.fcap + f; }
    }
    if (rc) atomicMin(A.err, (int32_t)rc);
    {
        // (round 4) which of the title's histogram / tree launches have work (Arena::num_huff); the tree of a Huffman-coded number
        // that never changes its value - a running index's delta - is written here, by this thread
        const bool huff = !rc && d.kind == KIND_NUM && d.has_huff;
        if (huff && d.n_sym <= 1) {
            const size_t tix = (size_t)s * A.fcap + f;
            uint8_t blob[16];
            uint32_t cl = 0;
            const uint32_t bl = huffman_single_symbol(&cl, blob);
            A.ncodes[tix * MAX_NUM_HUF] = cl;
            for (uint32_t k = 0; k < bl; ++k) A.nblob[tix * BLOB_SLOT_512 + k] = blob[k];
            A.nblob_len[tix] = bl;
        }
        const uint32_t want = (huff && d.n_sym > 1 ? 1u : 0u) | (!rc && d.kind == KIND_TEXT ? 2u : 0u);
        if (want & ~*A.num_huff) atomicOr(A.num_huff, want);
        if (want && quiet) atomicMin(A.err, (int32_t)ERR_SHAPE);
    }
    // compact index among the non-constant fields (all lanes with f < nf are still here)
    unsigned long long ncm = __ballot(d.kind != KIND_CONST);
    d.kf = (uint32_t)__popcll(ncm & ((1ull << f) - 1ull));
    if (f == 0) A.sb[s].ncf = (uint32_t)__popcll(ncm);
    A.fdesc[(size_t)s * A.fcap + f] = d;
}

// ---------------------------------------------------------------------------------------------------
// Title histograms (tasks.cpp:83-93 chars[][], :149-184 num_values / delta_values).  A workgroup walks
// TH_BATCHES batches of 255 records (+ the previous record as delta halo) and keeps its counts in LDS until
// the end: numeric Huffman fields in [8][512] counters, the first 16 positions of up to two text fields in
// packed 16-bit counters.  Everything else (more fields, later positions) goes to global memory with one
// atomic per wave and distinct character.  Few fat workgroups matter here: the per-(position, character)
// counters of a sub-block are a handful of addresses, and same-address global atomics serialise.
constexpr uint32_t NH_LDS_FIELDS = 8;
constexpr uint32_t TH_BATCHES = 16;
constexpr uint32_t TH_TEXT = 2, TH_ROWS = 16;

// The counters' 40 KB of LDS are dynamic, and a predicted step whose context's last job had nothing to count does not launch the
// kernel (k_field_finalize raises ERR_SHAPE if the job has something after all: the step runs again) - 2 450 workgroups that each
// wait for 40 KB of LDS to find out that they have nothing to do took 20 us alone and 100 beside the payload measure kernel.
__global__ __launch_bounds__(TS_THREADS) void k_title_hist(Arena A, uint32_t use_tv, uint32_t batches)
{
    if (*A.err) return;
    if (!(*A.num_huff & 3u)) return;   // (no Huffman-coded number with more than one value, no text field in any sub-block - the usual title)
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t slot_bias = sb.slot_bias;   // (read once: the stores below could alias A.sb)
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo, nf = sb.nf;
    const uint32_t wg_first = blockIdx.x * TS_OWNED * batches;
    if (wg_first >= n) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint8_t *buf = A.buf;
    const FieldDesc *fd = A.fdesc + (size_t)s * A.fcap;

    extern __shared__ __align__(16) uint32_t s_hist_dyn[];
    uint32_t (*const s_nh)[MAX_NUM_HUF] = reinterpret_cast<uint32_t (*)[MAX_NUM_HUF]>(s_hist_dyn);
    uint32_t (*const s_val)[TS_THREADS] = reinterpret_cast<uint32_t (*)[TS_THREADS]>(s_hist_dyn + NH_LDS_FIELDS * MAX_NUM_HUF);
    // two 16-bit counters per word: character c -> word c >> 1, half c & 1
    uint32_t (*const s_th)[TH_ROWS][128] = reinterpret_cast<uint32_t (*)[TH_ROWS][128]>(s_hist_dyn + NH_LDS_FIELDS * MAX_NUM_HUF + NH_LDS_FIELDS * TS_THREADS);
    __shared__ fmask_t s_need;           // bit f: field needs a histogram
    __shared__ uint32_t s_walk;          // some needed field is text or has no LDS slot: field boundaries are needed
    __shared__ uint8_t s_k[MAX_FIELDS];  // compact LDS index of numeric Huffman fields / text fields (0xff: global atomics)

    if (tid == 0) {
        uint32_t kk = 0, kt = 0, walk = 0;
        fmask_t need = 0;
        for (uint32_t f = 0; f < nf; ++f) {
            s_k[f] = 0xff;
            if (fd[f].kind == KIND_NUM && fd[f].has_huff && fd[f].n_sym > 1) {
                need |= (fmask_t)1 << f;
                if (kk < NH_LDS_FIELDS) s_k[f] = (uint8_t)kk++;
                else walk = 1;
            } else if (fd[f].kind == KIND_TEXT) {
                need |= (fmask_t)1 << f;
                walk = 1;
                if (kt < TH_TEXT) s_k[f] = (uint8_t)kt++;
            }
        }
        s_need = need;
        s_walk = walk;
    }
    __syncthreads();
    const fmask_t need = s_need;
    if (!need) return;   // (the usual title of plain numbers: nothing is Huffman-coded - leave before the 24 KB of counters are cleared)
    const bool walk = s_walk != 0;
    for (uint32_t k = tid; k < NH_LDS_FIELDS * MAX_NUM_HUF; k += TS_THREADS) (&s_nh[0][0])[k] = 0;
    for (uint32_t k = tid; k < TH_TEXT * TH_ROWS * 128; k += TS_THREADS) (&s_th[0][0][0])[k] = 0;
    __syncthreads();

    for (uint32_t batch = 0; batch < batches; ++batch) {
        const uint32_t b_first = wg_first + batch * TS_OWNED;
        if (b_first >= n) break;
        const int64_t i = (int64_t)b_first + tid - 1;
        const bool valid = i >= 0 && i < (int64_t)n;
        const bool owned = valid && tid >= 1;
        const uint32_t g = valid ? rec_lo + (uint32_t)i : rec_lo;
        const uint32_t ls = A.ls[g];
        if (!walk) {
            // every needed field is numeric and k_title_stats_swar kept its values: no need to touch the titles
            for (uint32_t f = 0; f < nf; ++f)
                if ((need >> f) & 1) s_val[s_k[f]][tid] = (valid && use_tv) ? A.tv[(size_t)fd[f].cand * A.tw_stride + g + slot_bias] : 0u;
        }
        if (walk || !use_tv) {
            uint64_t sepmask[2] = {0, 0};   // (two-word form for both title widths: this kernel is not the bottleneck)
            if (use_tv && valid) {
                sepmask[0] = A.tsep[(size_t)g * A.tsep_words];
                if (A.tsep_words == 2) sepmask[1] = A.tsep[(size_t)g * 2 + 1];
            }
            uint32_t pc = ls;
            // lockstep over fields so that the lanes of a wave look at the same field together
            for (uint32_t f = 0; f < nf; ++f) {
                uint32_t ec;
                if (use_tv) {   // field boundaries from the separator mask of k_title_stats_swar
                    const uint32_t rel = pc - ls;
                    ec = pc + t_next_set<2>(sepmask, rel);
                } else {
                    ec = field_end(buf, pc);
                }
                const FieldDesc &d = fd[f];
                if ((need >> f) & 1) {
                    if (d.kind == KIND_NUM) {
                        uint32_t v = 0;
                        if (use_tv) v = valid ? A.tv[(size_t)d.cand * A.tw_stride + g + slot_bias] : 0u;
                        else { bool o; parse_num(buf, pc, ec, v, o); }
                        const uint32_t k = s_k[f];
                        if (k != 0xff) {
                            s_val[k][tid] = v;
                        } else if (owned) {  // more than 8 numeric Huffman fields: slow path, previous value re-parsed
                            uint32_t *nh = A.nhist + ((size_t)s * A.fcap + f) * MAX_NUM_HUF;
                            if (!d.delta_coding) {
                                uint32_t idx = v - (uint32_t)d.min_v;
                                if (idx < d.n_sym) atomicAdd(&nh[idx], i == 0 ? 2u : 1u);
                            } else if (i >= 1) {
                                uint32_t pp = A.ls[g - 1], ep = 0;
                                for (uint32_t k2 = 0; k2 <= f; ++k2) { ep = field_end(buf, pp); if (k2 < f) pp = ep + 1; }
                                uint32_t pv = 0;
                                bool o;
                                parse_num(buf, pp, ep, pv, o);
                                uint32_t idx = v - pv - (uint32_t)d.min_d;
                                if (idx < d.n_sym) atomicAdd(&nh[idx], 1u);
                            }
                        }
                    } else {
                        // text: chars[min(x,128)][c]++ (tasks.cpp:83-93)
                        const uint32_t len = owned ? ec - pc : 0;
                        const uint32_t kt = s_k[f];
                        uint32_t wmax = len;
#pragma unroll
                        for (int dd = 32; dd >= 1; dd >>= 1) { uint32_t o2 = __shfl_xor(wmax, dd); wmax = o2 > wmax ? o2 : wmax; }
                        uint32_t *ch = A.chist + (size_t)d.slot * TEXT_TREES * 256;
                        const uint2 f8 = len ? load8_unaligned(buf, pc, A.buf_len) : make_uint2(0, 0);   // the first eight characters with one load
                        for (uint32_t x = 0; x < wmax; ++x) {
                            bool act = x < len;
                            const uint32_t c = !act ? 0x100u : (x < 4 ? (f8.x >> (8 * x)) & 0xffu : (x < 8 ? (f8.y >> (8 * (x - 4))) & 0xffu : buf[pc + x]));
                            if (kt != 0xff && x < TH_ROWS) {   // LDS counters (wave-uniform condition)
                                // (round 4) the characters of a text field repeat from record to record (a flag, a barcode): one
                                // atomic per distinct character for the first three of them - sixty-four lanes adding to one word
                                // are served one after the other - and a lane each for whatever is left (free text)
                                unsigned long long m = __ballot(act);
                                for (int round = 0; round < 3 && m; ++round) {
                                    const int leader = __ffsll((long long)m) - 1;
                                    const uint32_t c0 = __shfl(c, leader);
                                    const unsigned long long same = __ballot(act && c == c0);
                                    if (lane == leader) atomicAdd(&s_th[kt][x][c0 >> 1], (uint32_t)__popcll(same) << (16 * (c0 & 1)));
                                    if (c == c0) act = false;
                                    m &= ~same;
                                }
                                if (m && act) atomicAdd(&s_th[kt][x][c >> 1], 1u << (16 * (c & 1)));
                                continue;
                            }
                            const uint32_t row = x < MAX_STAT_LEN ? x : MAX_STAT_LEN;
                            unsigned long long m = __ballot(act);
                            while (m) {   // one atomic per wave and distinct character
                                const int leader = __ffsll((long long)m) - 1;
                                const uint32_t c0 = __shfl(c, leader);
                                const unsigned long long same = __ballot(act && c == c0);
                                if (lane == leader) atomicAdd(&ch[row * 256 + c0], (uint32_t)__popcll(same));
                                if (c == c0) act = false;
                                m &= ~same;
                            }
                        }
                    }
                }
                pc = ec + 1;
            }
        }
        __syncthreads();
        // numeric histograms staged in LDS: value coding counts record 0 twice (phyNGSC.cpp:368 + tasks.cpp:175-177)
        for (uint32_t f = 0; f < nf; ++f) {
            const uint32_t k = s_k[f];
            if (k == 0xff || !((need >> f) & 1) || fd[f].kind != KIND_NUM) continue;
            const FieldDesc &d = fd[f];
            uint32_t idx = 0xffffffffu, wgt = 0;
            if (owned) {
                const uint32_t v = s_val[k][tid];
                if (!d.delta_coding) { idx = v - (uint32_t)d.min_v; wgt = i == 0 ? 2u : 1u; }
                else if (i >= 1) { idx = v - s_val[k][tid - 1] - (uint32_t)d.min_d; wgt = 1u; }
            }
            // (round 4) a counter that every lane of the wave hits - the delta of a running index, a constant lane number -
            // takes ONE atomic with the lanes' sum: sixty-four lanes adding to one LDS word are served one after the other
            const bool act = wgt != 0 && idx < d.n_sym;
            const unsigned long long m = __ballot(act);
            if (m) {
                const int leader = __ffsll((long long)m) - 1;
                const uint32_t i0 = (uint32_t)__shfl((int)idx, leader);
                if (__ballot(act && idx == i0) == m) {
                    const uint32_t sum = (uint32_t)__popcll(m) + (uint32_t)__popcll(__ballot(act && wgt == 2u));
                    if (lane == leader) atomicAdd(&s_nh[k][i0], sum);
                } else if (act) atomicAdd(&s_nh[k][idx], wgt);
            }
        }
        __syncthreads();   // s_val is rewritten by the next batch
    }
    // flush
    for (uint32_t f = 0; f < nf; ++f) {
        const uint32_t k = s_k[f];
        if (k == 0xff || !((need >> f) & 1)) continue;
        if (fd[f].kind == KIND_NUM) {
            uint32_t *nh = A.nhist + ((size_t)s * A.fcap + f) * MAX_NUM_HUF;
            for (uint32_t x = tid; x < fd[f].n_sym; x += TS_THREADS)
                if (s_nh[k][x]) atomicAdd(&nh[x], s_nh[k][x]);
        } else {
            uint32_t *ch = A.chist + (size_t)fd[f].slot * TEXT_TREES * 256;
            for (uint32_t x = tid; x < TH_ROWS * 128; x += TS_THREADS) {
                const uint32_t w = s_th[k][x / 128][x % 128];
                if (w & 0xffffu) atomicAdd(&ch[(x / 128) * 256 + 2 * (x % 128)], w & 0xffffu);
                if (w >> 16) atomicAdd(&ch[(x / 128) * 256 + 2 * (x % 128) + 1], w >> 16);
            }
        }
    }
}

// ---- launchers ---------------------------------------------------------------------------------------
void launch_seq_qual_stats(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    uint32_t rows = A.qrows < HIST_MAX_ROWS ? A.qrows : HIST_MAX_ROWS;
    dim3 grid((max_rec + SQ_REC_PER_WG - 1) / SQ_REC_PER_WG, A.n_sb);
    size_t lds = (size_t)rows * HIST_STRIDE * sizeof(uint32_t);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_seq_qual_stats), hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    hipLaunchKernelGGL(k_seq_qual_stats, grid, dim3(SQ_THREADS), lds, st, A, rows);
}

void launch_sq_finalize(hipStream_t st, const Arena &A, bool small_only)
{
    hipLaunchKernelGGL(k_sq_finalize, dim3(A.n_sb), dim3(256), 0, st, A, small_only ? 1u : 0u);
}

void launch_field_finalize(hipStream_t st, const Arena &A, bool quiet)
{
    hipLaunchKernelGGL(k_field_finalize, dim3(A.n_sb), dim3(64), 0, st, A, quiet ? 1u : 0u);
}

void launch_title_hist(hipStream_t st, const Arena &A, uint32_t max_rec, bool use_tv)
{
    // batches of 255 records per workgroup: as many as leave the job ~2000 workgroups (a workgroup's batches are a serial chain of
    // loads, and its LDS histograms are flushed once), at most TH_BATCHES - the 16-bit text counters hold 255 * 16 records
    uint32_t batches = (uint32_t)(((uint64_t)max_rec * A.n_sb) / ((uint64_t)TS_OWNED * 2048));
    batches = batches < 1 ? 1 : (batches > TH_BATCHES ? TH_BATCHES : batches);
    dim3 grid((max_rec + TS_OWNED * batches - 1) / (TS_OWNED * batches), A.n_sb);
    const size_t lds_bytes = 4 * ((size_t)NH_LDS_FIELDS * MAX_NUM_HUF + (size_t)NH_LDS_FIELDS * TS_THREADS + (size_t)TH_TEXT * TH_ROWS * 128);
    hipLaunchKernelGGL(k_title_hist, grid, dim3(TS_THREADS), lds_bytes, st, A, use_tv ? 1u : 0u, batches);
}

}  // namespace ngsc
