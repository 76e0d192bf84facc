// kernels_cs.hip - colour-space (SOLiD) reads, phyNGSC.cpp:473-547: a primer base followed by transitions '0'..'3'
// ('.' = unknown).  The reference decodes them to bases in its read buffer before anything else looks at the record; here
// the job is run on a COPY of its input in which the colour-space sub-blocks are rewritten the same way, record by
// record at its own place, and every other kernel stays what it is:
//
//   title \n  decoded bases (L')  \n + \n  "qualities" (L')  \n  padding
//
// L' = L - translation: while every read of the sub-block starts with the first read's primer base and first quality
// they are dropped (FLAG_DELTA_CONSTANT); the first read that starts differently ends that for the rest of the sub-block
// (the reference's loop is sequential: k_cs_detect finds that read with a minimum over the records).  The "quality"
// line holds what the reference leaves there (phyNGSC.cpp:531 copies from the SEQUENCE line): the colours when the read
// is shifted, else the decoded bases behind the kept first quality.  A transition c turns base b into b ^ c on the scale
// A, C, G, T = 0..3; '.' and '/' give 'N' and leave the running base alone - a serial dependence along the read, so
// one lane walks one read.  Rare input: correctness first, byte loads and stores.
#include "ngsc_kernels.h"

namespace ngsc {

__global__ __launch_bounds__(256) void k_cs_detect(Arena A)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.x, tid = threadIdx.x;
    SbState &sb = A.sb[s];
    const uint32_t n = sb.n_rec, rec_lo = sb.rec_lo;
    __shared__ uint32_t s_v;
    const uint32_t te0 = A.te[rec_lo], se0 = A.se[rec_lo];
    const uint32_t L0 = se0 - te0 - 1;
    const uint32_t c0 = L0 >= 1 ? A.buf[te0 + 1] : 0u, c1 = L0 >= 2 ? A.buf[te0 + 2] : 0u, q0 = L0 >= 1 ? A.buf[se0 + 3] : 0u;
    const bool is_delta = c1 >= '0' && c1 <= '3', no_begin = c0 >= '0' && c0 <= '3';
    if (tid == 0) s_v = 0xffffffffu;
    __syncthreads();
    if (is_delta && !no_begin) {   // phyNGSC.cpp:489-493: every read starts like the first one?
        for (uint32_t r = tid; r < n; r += blockDim.x) {
            const uint32_t te = A.te[rec_lo + r], se = A.se[rec_lo + r];
            const uint32_t L = se - te - 1;
            const bool same = L >= 1 && A.buf[te + 1] == c0 && A.buf[se + 3] == q0;
            if (!same) atomicMin(&s_v, r);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t v = s_v;
        sb.cs = (is_delta ? 1u : 0u) | (no_begin ? 2u : 0u) | ((no_begin || v != 0xffffffffu) ? 4u : 0u);
        sb.cs_v = v;
        // the dropped primer bases still count as DNA symbols (phyNGSC.cpp:508-509 counts before the shift)
        if (is_delta && !no_begin) atomicAdd(&A.dna_occ[(size_t)s * 256 + c0], v < n ? v : n);
    }
}

// orig: the job's bytes as they came; copy: the buffer the other kernels will read (starts as an exact copy)
__global__ __launch_bounds__(256) void k_cs_transform(Arena A, const uint8_t *orig, uint8_t *copy, uint32_t *se_w)
{
    if (*A.err) return;
    const uint32_t s = blockIdx.y;
    const SbState &sb = A.sb[s];
    if (!(sb.cs & 1)) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sb.n_rec) return;
    const uint32_t g = sb.rec_lo + i;
    const uint32_t te = A.te[g], se = A.se[g];
    const uint32_t L = se - te - 1, s0 = te + 1, q0 = se + 3;
    const bool no_begin = (sb.cs & 2) != 0;
    const uint32_t tr = (!no_begin && i < sb.cs_v) ? 1u : 0u;
    if (L < tr) { atomicMin(A.err, (int32_t)ERR_FORMAT); return; }   // (the reference would write in front of the read)
    const uint32_t Lp = L - tr, qp = s0 + Lp + 3;
    uint32_t cur = 0, k = 0;
    uint8_t sym = 'A';
    if (!no_begin && L) {
        sym = orig[s0];
        k = 1;
        if (!tr) {
            copy[qp] = orig[q0];   // (copy[s0] already holds the primer base)
            // A kept primer base that is an ambiguity code with a quality in '!'..'(' could make the record a transfer
            // candidate; the reference then counts its DNA symbols before dropping them.  Not reproduced.
            if (amb_code(sym) > 1 && orig[q0] >= 33 && orig[q0] <= 40) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED);
        }
    }
    bool bad = false;
    for (; k < L; ++k) {
        if (sym == 'A') cur = 0; else if (sym == 'C') cur = 1; else if (sym == 'G') cur = 2; else if (sym == 'T') cur = 3;
        const uint8_t colour = orig[s0 + k];
        const uint32_t c = (uint32_t)colour - '.';
        bad = bad || c > 5;   // not a colour: the reference indexes past its table
        const uint32_t b = cur ^ ((c - 2) & 3u);
        sym = c < 2 ? (uint8_t)'N' : (uint8_t)(b == 0 ? 'A' : (b == 1 ? 'C' : (b == 2 ? 'G' : 'T')));
        copy[s0 + k - tr] = sym;
        copy[qp + k - tr] = tr ? colour : sym;
    }
    if (bad) atomicMin(A.err, (int32_t)ERR_FORMAT);
    copy[s0 + Lp] = '\n'; copy[s0 + Lp + 1] = '+'; copy[s0 + Lp + 2] = '\n';
    copy[qp + Lp] = '\n';
    for (uint32_t x = 0; x < 2 * tr; ++x) copy[qp + Lp + 1 + x] = '\n';
    se_w[g] = s0 + Lp;
}

void launch_cs(hipStream_t st, const Arena &A, uint32_t max_rec, const uint8_t *orig, uint8_t *copy, uint32_t *se_w)
{
    hipLaunchKernelGGL(k_cs_detect, dim3(A.n_sb), dim3(256), 0, st, A);
    hipLaunchKernelGGL(k_cs_transform, dim3((max_rec + 255) / 256, A.n_sb), dim3(256), 0, st, A, orig, copy, se_w);
}

}  // namespace ngsc
