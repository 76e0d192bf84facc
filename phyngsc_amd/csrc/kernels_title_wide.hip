// kernels_title_wide.hip - the title path for jobs in which some sub-block's first title has more than MAX_FIELDS (64)
// separators.  The reference has no field limit (phyNGSC.cpp:342-423 pushes a Field per separator; AnalyzeTitleFields,
// tasks.cpp:22-223, and StoreTitle, tasks.cpp:289-510, loop over them); the word-parallel and cached title kernels of
// kernels_title.hip keep a bit per field in one 64-bit mask and LDS tables sized for 64 fields.  Such titles do not occur in
// sequencer output, so this path is written for exactness, not speed: the field arrays of the job are WIDE_FIELDS (512)
// wide (Arena::fcap; a first title holds at most 512 bytes, so it cannot have more separators), statistics go straight to
// global memory, and the 32-record title groups are measured and written by one thread each through the same sequential
// helpers of ngsc_common.h that the CPU harness checks against the oracle (title_group<WideMask>).
//
//   k_rec0_wide             field count of every sub-block's first record              (phyNGSC.cpp:342-379)
//   k_title_stats_wide      AnalyzeTitleFields reductions, a thread per record          (tasks.cpp:22-193)
//   k_field_finalize_wide   per-field coding decisions                                  (tasks.cpp:196-222, 324-338)
//   k_title_hist_wide       value / delta / per-position character histograms           (tasks.cpp:83-93, 149-184)
//   k_title_measure_wide    bytes of every 32-record group                              (tasks.cpp:393-509)
//   k_title_pack_wide       StoreTitle body, a thread per group (groups are byte aligned, tasks.cpp:508)
// The numeric / text Huffman trees are built by the kernels of kernels_huffman.hip, the title header by title_header().
#include "ngsc_kernels.h"

namespace ngsc {

constexpr int TW_THREADS = 256;

// unsigned maximum / OR into a statistic that other threads raise too: the atomic only when it would change the value
__device__ __forceinline__ void w_max(uint32_t *p, uint32_t v)
{
    if (v > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, v);
}
__device__ __forceinline__ void w_or(uint32_t *p, uint32_t v)
{
    if ((__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & v) != v) atomicOr(p, v);
}

// One thread per sub-block: separators of its first title = its field count (phyNGSC.cpp:345-379).
__global__ __launch_bounds__(64) void k_rec0_wide(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.x * 64 + threadIdx.x;
    if (s >= A.n_sb) return;
    const SbState &sb = A.sb[s];
    const uint32_t ls0 = A.ls[sb.rec_lo], te0 = A.te[sb.rec_lo];
    if (te0 - ls0 + 1 > REC0_TITLE_CAP) { atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
    A.sb[s].nf = count_fields(A.buf, ls0, te0);   // (<= 512 = WIDE_FIELDS: one per title byte at most)
}

// A thread per record walks three titles field by field: the sub-block's first record, the record in front of its own, its own.
__global__ __launch_bounds__(TW_THREADS) void k_title_stats_wide(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo, nf = sb.nf;
    const uint32_t i = blockIdx.x * TW_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint8_t *buf = A.buf;
    const uint32_t g = rec_lo + i;
    const uint32_t ls = A.ls[g], te = A.te[g];
    if (count_fields(buf, ls, te) != nf) { atomicMin(A.err, (int32_t)ERR_FIELDS); return; }   // phyNGSC.cpp:417-421 only warns
    uint32_t p0 = A.ls[rec_lo], pp = i ? A.ls[g - 1] : ls, pc = ls;
    FieldStat *fst = A.fstat + (size_t)s * A.fcap;
    for (uint32_t f = 0; f < nf; ++f) {
        const uint32_t e0 = field_end(buf, p0), ep = field_end(buf, pp), ec = field_end(buf, pc);
        const uint32_t len = ec - pc, len0 = e0 - p0;
        uint32_t v = 0, vp = 0;
        bool ovf = false, ovp = false;
        const bool isn = parse_num(buf, pc, ec, v, ovf);
        uint32_t *st = reinterpret_cast<uint32_t *>(fst + f);
        // Hamming mask (tasks.cpp:187-193): positions inside both fields where the bytes differ
        const uint32_t m = len < len0 ? len : len0;
        bool diff = false;
        for (uint32_t k = 0; k < m; ++k) {
            if (buf[pc + k] != buf[p0 + k]) {
                diff = true;
                if (k < MAX_TEXT_LEN0) w_or(st + 10 + (k >> 5), 1u << (k & 31));
            }
        }
        const bool same = len == len0 && !diff;
        // a field identical to record 0's changes nothing beyond what record 0's own thread contributes
        if (!same || i == 0) {
            w_max(st + 0, len);
            w_max(st + 1, ~len);
            if (!same) w_max(st + 2, 1u);
            if (len != len0) w_max(st + 3, 1u);
            if (!isn) w_max(st + 4, 1u);
            else {
                if (ovf) w_max(st + 5, 1u);
                w_max(st + 6, bias((int32_t)v));
                w_max(st + 7, ~bias((int32_t)v));
            }
        }
        if (i >= 1 && isn && parse_num(buf, pp, ep, vp, ovp)) {   // tasks.cpp:149-166: both neighbours numeric
            const uint32_t bd = bias((int32_t)(v - vp));
            w_max(st + 8, bd);
            w_max(st + 9, ~bd);
        }
        p0 = e0 + 1; pp = ep + 1; pc = ec + 1;
    }
}

// One workgroup of WIDE_FIELDS threads per sub-block, thread = field.
__global__ __launch_bounds__(WIDE_FIELDS) void k_field_finalize_wide(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.x, f = threadIdx.x;
    const SbState &sb = A.sb[s];
    const uint32_t nf = sb.nf;
    const uint8_t *buf = A.buf;
    __shared__ uint32_t s_cnt[WIDE_FIELDS / 64];
    FieldDesc d;
    d.kind = KIND_CONST;
    int rc = OK;
    if (f < nf) {
        uint32_t p = A.ls[sb.rec_lo], e = 0;
        for (uint32_t k = 0; k <= f; ++k) { e = field_end(buf, p); if (k < f) p = e + 1; }   // record 0's field f
        for (int k = 0; k < 3; ++k) d.pad[k] = 0;
        d.cand = 0xff;   // (the candidate value table of the word-parallel kernels is not used on this path)
        rc = finalize_field(A.fstat[(size_t)s * A.fcap + f], p, e - p, buf[e], sb.n_rec, f, nf, d);
        if (rc == OK && d.kind == KIND_TEXT) {
            const uint32_t slot = atomicAdd(A.slot_counter, 1u);
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
        }
    }
    // compact index among the non-constant fields, in field order
    const bool nc = f < nf && d.kind != KIND_CONST;
    const unsigned long long m = __ballot(nc);
    const uint32_t lane = f & 63u, wave = f >> 6;
    if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < WIDE_FIELDS / 64; ++w) { before += w < wave ? s_cnt[w] : 0u; total += s_cnt[w]; }
    if (f < nf) {
        d.kf = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        A.fdesc[(size_t)s * A.fcap + f] = d;
    }
    if (f == 0) A.sb[s].ncf = total;
}

// A thread per record: every numeric field with a tree and every text field of its title into the global histograms.
__global__ __launch_bounds__(TW_THREADS) void k_title_hist_wide(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo, nf = sb.nf;
    const uint32_t i = blockIdx.x * TW_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint8_t *buf = A.buf;
    const FieldDesc *fd = A.fdesc + (size_t)s * A.fcap;
    const uint32_t g = rec_lo + i;
    uint32_t pc = A.ls[g], pp = i ? A.ls[g - 1] : pc;
    for (uint32_t f = 0; f < nf; ++f) {
        const uint32_t ec = field_end(buf, pc), ep = field_end(buf, pp);
        const FieldDesc &d = fd[f];
        if (d.kind == KIND_NUM && d.has_huff && d.n_sym > 1) {
            uint32_t v = 0, pv = 0;
            bool o;
            parse_num(buf, pc, ec, v, o);
            uint32_t *nh = A.nhist + ((size_t)s * A.fcap + f) * MAX_NUM_HUF;
            if (!d.delta_coding) {   // value coding counts record 0 twice (phyNGSC.cpp:368 + tasks.cpp:175-177)
                const uint32_t idx = v - (uint32_t)d.min_v;
                if (idx < d.n_sym) atomicAdd(&nh[idx], i == 0 ? 2u : 1u);
            } else if (i >= 1) {
                parse_num(buf, pp, ep, pv, o);
                const uint32_t idx = v - pv - (uint32_t)d.min_d;
                if (idx < d.n_sym) atomicAdd(&nh[idx], 1u);
            }
        } else if (d.kind == KIND_TEXT) {   // chars[min(x, 128)][c]++ (tasks.cpp:83-93)
            uint32_t *ch = A.chist + (size_t)d.slot * TEXT_TREES * 256;
            for (uint32_t x = 0; x < ec - pc; ++x) atomicAdd(&ch[(x < MAX_STAT_LEN ? x : MAX_STAT_LEN) * 256 + buf[pc + x]], 1u);
        }
        pc = ec + 1; pp = ep + 1;
    }
}

// A thread per 32-record group (tasks.cpp:393-509): its bytes.
__global__ __launch_bounds__(64) void k_title_measure_wide(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t ngrp = (sb.n_rec + GROUP - 1) / GROUP;
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k >= ngrp) return;
    const uint32_t g0 = sb.rec_lo + k * GROUP, g1 = g0 + (k * GROUP + GROUP <= sb.n_rec ? GROUP : sb.n_rec - k * GROUP);
    BitCounter bc;
    title_group<WideMask>(A, s, g0, g1, bc);
    A.goff[A.grp_lo[s] + k] = (uint32_t)bc.nbytes();
}

// ... and its bits, at the group's byte offset inside the title stream (the output is zeroed; a group owns whole bytes).
__global__ __launch_bounds__(64) void k_title_pack_wide(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    const uint32_t ngrp = (sb.n_rec + GROUP - 1) / GROUP;
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k >= ngrp) return;
    const uint32_t g0 = sb.rec_lo + k * GROUP, g1 = g0 + (k * GROUP + GROUP <= sb.n_rec ? GROUP : sb.n_rec - k * GROUP);
    ByteWriter w(A.out + sb.out_off + sb.info_len + sb.title_hdr + A.goff[A.grp_lo[s] + k]);
    title_group<WideMask>(A, s, g0, g1, w);
}

// ---- launchers ---------------------------------------------------------------------------------------
void launch_title_wide_stats(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    hipLaunchKernelGGL(k_rec0_wide, dim3((A.n_sb + 63) / 64), dim3(64), 0, st, A);
    hipLaunchKernelGGL(k_title_stats_wide, dim3((max_rec + TW_THREADS - 1) / TW_THREADS, A.n_sb), dim3(TW_THREADS), 0, st, A);
    hipLaunchKernelGGL(k_field_finalize_wide, dim3(A.n_sb), dim3(WIDE_FIELDS), 0, st, A);
    hipLaunchKernelGGL(k_title_hist_wide, dim3((max_rec + TW_THREADS - 1) / TW_THREADS, A.n_sb), dim3(TW_THREADS), 0, st, A);
}

void launch_title_wide_measure(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    const uint32_t ngrp = (max_rec + GROUP - 1) / GROUP;
    hipLaunchKernelGGL(k_title_measure_wide, dim3((ngrp + 63) / 64, A.n_sb), dim3(64), 0, st, A);
}

void launch_title_wide_pack(hipStream_t st, const Arena &A, uint32_t max_rec)
{
    const uint32_t ngrp = (max_rec + GROUP - 1) / GROUP;
    hipLaunchKernelGGL(k_title_pack_wide, dim3((ngrp + 63) / 64, A.n_sb), dim3(64), 0, st, A);
}

}  // namespace ngsc
