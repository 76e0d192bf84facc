// kernels_huffman.hip - Huffman code build + tree serialisation, one wavefront per tree.
//
// Replaces HuffmanEncoder::Insert/Complete/StoreTree (huffman.h:176-186, huffman.cpp:18-118) for every
// tree of every sub-block in one launch: max_quality_length+1 quality trees (tasks.cpp:590-605), the
// optional DNA tree (tasks.cpp:525-542), numeric title-field trees (tasks.cpp:338-347) and the
// per-position text-field trees (tasks.cpp:364-387).
//
// The reference's heap pops the smallest (frequency, symbol) pair, a strict total order, so the result
// is reproduced by: rank-sort of the (freq<<32|symbol) keys by the 64 lanes in LDS, then a two-queue
// merge + code walk + pre-order serialisation by lane 0 (n <= 512, a few hundred LDS steps).
// Latency-bound, negligible bytes: no roofline claim attached to this kernel.
#include "ngsc_kernels.h"
#include <type_traits>

namespace ngsc {

// Working set of one tree of up to CAP symbols.  The 512-symbol version (numeric title fields) is 26 KB, i.e.
// six single-wave workgroups per CU; quality and DNA alphabets almost always fit 64 symbols (2.7 KB), which lets
// a CU build 5x more trees at a time - the lane-0 merge is latency, not throughput.
template <uint32_t CAP, uint32_t BLOB>
struct TreeLdsT {
    uint64_t keys[CAP];
    uint64_t sorted[CAP];
    uint32_t ifreq[CAP];
    uint32_t code[2 * CAP];
    uint16_t lc[2 * CAP];
    uint16_t rc[2 * CAP];
    uint16_t stack[2 * CAP + 2];
    uint8_t len[2 * CAP];
    uint8_t blob[BLOB];
    uint32_t blob_len;
    uint32_t max_len;
};
using TreeLds = TreeLdsT<MAX_NUM_HUF, BLOB_SLOT_512>;
constexpr uint32_t SMALL_TREE = 64;
using TreeLdsSmall = TreeLdsT<SMALL_TREE, 160>;

// keys[0..n) must be filled (by all lanes) before the call.  Leaves T.code/T.len/T.blob/T.blob_len.
template <class Tree>
__device__ void build_tree_wave(Tree &T, uint32_t n)
{
    const int lane = threadIdx.x & 63;
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64) {
        uint64_t k = T.keys[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += T.keys[j] < k;
        T.sorted[rank] = k;
    }
    __syncthreads();
    if (lane == 0) {
        uint32_t mx = 0;
        T.blob_len = huffman_sequential(T.sorted, n, T.ifreq, T.lc, T.rc, T.code, T.len, T.stack, T.blob, &mx);
        T.max_len = mx;
    }
    __syncthreads();
}

// grid: (qrows + 1 [+ MAX_FIELDS], n_sb); block: 64 threads (one wave).  SMALL: the quality and DNA trees of the
// sub-blocks whose alphabets fit 64 symbols; the other instance does the rest (numeric title fields, big alphabets).
template <bool SMALL, class Tree>
__device__ __forceinline__ void huffman_sb_tree(const Arena &A, Tree &T, uint32_t s, uint32_t t)
{
    const int lane = threadIdx.x;
    const SbState &sb = A.sb[s];
    const bool fits_small = sb.nqual <= SMALL_TREE && sb.nsym <= SMALL_TREE;
    if (t <= A.qrows && fits_small != SMALL) return;
    if (t < A.qrows) {
        // quality tree of row t (row 0: global statistics; row k+1: position k), tasks.cpp:590-605
        if (t > sb.max_q) return;
        const uint32_t n = sb.nqual;
        const uint32_t *row = A.qhist + ((size_t)s * A.qrows + t) * 256;
        for (uint32_t j = lane; j < n; j += 64) T.keys[j] = ((uint64_t)row[sb.qualities[j]] << 32) | j;
        build_tree_wave(T, n);
        if (T.max_len > MAX_CODE_LEN) { if (lane == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
        uint32_t *dst = A.qcodes + ((size_t)s * A.qrows + t) * 256;
        for (uint32_t j = lane; j < n; j += 64) dst[j] = (T.code[j] << 5) | T.len[j];
        uint8_t *bd = A.qblob + ((size_t)s * A.qrows + t) * BLOB_SLOT_256;
        for (uint32_t k = lane; k < T.blob_len; k += 64) bd[k] = T.blob[k];
        if (lane == 0) A.qblob_len[(size_t)s * A.qrows + t] = T.blob_len;
    } else if (t == A.qrows) {
        // DNA tree, only when the alphabet has more than 4 symbols (tasks.cpp:525-542)
        if (sb.flags & F_DNA_PLAIN) return;
        const uint32_t n = sb.nsym;
        const uint32_t *occ = A.dna_occ + (size_t)s * 256;
        for (uint32_t j = lane; j < n; j += 64) T.keys[j] = ((uint64_t)occ[sb.symbols[j]] << 32) | j;
        build_tree_wave(T, n);
        if (T.max_len > MAX_CODE_LEN) { if (lane == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
        uint32_t *dst = A.dcodes + (size_t)s * 256;
        for (uint32_t j = lane; j < n; j += 64) dst[sb.symbols[j]] = (T.code[j] << 5) | T.len[j];
        uint8_t *bd = A.dblob + (size_t)s * BLOB_SLOT_256;
        for (uint32_t k = lane; k < T.blob_len; k += 64) bd[k] = T.blob[k];
        if (lane == 0) A.dblob_len[s] = T.blob_len;
    } else {
        // numeric title field tree (tasks.cpp:338-347)
        if (SMALL) return;   // (never launched for these)
        const uint32_t f = t - A.qrows - 1;
        if (f >= sb.nf) return;
        const FieldDesc &d = A.fdesc[(size_t)s * MAX_FIELDS + f];
        if (d.kind != KIND_NUM || !d.has_huff) return;
        const uint32_t n = d.n_sym;
        const size_t tix = (size_t)s * MAX_FIELDS + f;
        const uint32_t *h = A.nhist + tix * MAX_NUM_HUF;
        for (uint32_t j = lane; j < n; j += 64) T.keys[j] = ((uint64_t)h[j] << 32) | j;
        build_tree_wave(T, n);
        if (T.max_len > MAX_CODE_LEN) { if (lane == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
        uint32_t *dst = A.ncodes + tix * MAX_NUM_HUF;
        for (uint32_t j = lane; j < n; j += 64) dst[j] = (T.code[j] << 5) | T.len[j];
        uint8_t *bd = A.nblob + tix * BLOB_SLOT_512;
        for (uint32_t k = lane; k < T.blob_len; k += 64) bd[k] = T.blob[k];
        if (lane == 0) A.nblob_len[tix] = T.blob_len;
    }
}

// Trees t0 .. t0 + nt - 1 of every sub-block, dealt to the workgroups in turn.  The 512-symbol instance is launched with a
// small grid: its 26 KB of LDS per (single-wave) workgroup must be free on a CU before a workgroup can even find out that
// its sub-block's alphabets fit the small instance and it has nothing to do - 4560 such workgroups queued behind the title
// statistics kernel's LDS for 260 us of the staggered schedule, and the measure kernel behind them (round 3).
template <bool SMALL>
__global__ __launch_bounds__(64) void k_huffman_sb(Arena A, uint32_t t0, uint32_t nt)
{
    if (*A.err) return;
    using Tree = typename std::conditional<SMALL, TreeLdsSmall, TreeLds>::type;
    __shared__ Tree T;
    const uint32_t pairs = nt * A.n_sb;
    for (uint32_t p = blockIdx.x; p < pairs; p += gridDim.x) {
        huffman_sb_tree<SMALL>(A, T, p / nt, t0 + p % nt);
        __syncthreads();   // (T is used again)
    }
}

// Text-field trees: grid (129, TEXT_GRID_Y); each workgroup row walks the used slots with stride
// gridDim.y (the slot count lives on the device).  Tree j of a field exists iff text_tree_needed().
constexpr uint32_t TEXT_GRID_Y = 64;

__global__ __launch_bounds__(64) void k_huffman_text(Arena A)
{
    if (*A.err) return;
    __shared__ TreeLds T;
    const uint32_t j = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t used = *A.slot_counter < A.n_slots ? *A.slot_counter : A.n_slots;
    for (uint32_t slot = blockIdx.y; slot < used; slot += gridDim.y) {
        const FieldDesc &d = A.fdesc[A.slot_owner[slot]];
        if (d.kind != KIND_TEXT || d.slot != slot || !text_tree_needed(d, j)) continue;
        const size_t tix = (size_t)slot * TEXT_TREES + j;
        const uint32_t *h = A.chist + tix * 256;
        for (uint32_t c = lane; c < 256; c += 64) T.keys[c] = ((uint64_t)h[c] << 32) | c;
        build_tree_wave(T, 256);
        if (T.max_len > MAX_CODE_LEN) { if (lane == 0) atomicMin(A.err, (int32_t)ERR_UNSUPPORTED); return; }
        uint32_t *dst = A.tcodes + tix * 256;
        for (uint32_t c = lane; c < 256; c += 64) dst[c] = (T.code[c] << 5) | T.len[c];
        uint8_t *bd = A.tblob + tix * BLOB_SLOT_256;
        for (uint32_t k = lane; k < T.blob_len; k += 64) bd[k] = T.blob[k];
        if (lane == 0) A.tblob_len[tix] = T.blob_len;
        __syncthreads();
    }
}

// big_grid: workgroups of the 512-symbol instance
static void launch_huffman_range(hipStream_t st, const Arena &A, bool small, uint32_t t0, uint32_t nt, uint32_t big_grid = 1024)
{
    const uint32_t pairs = nt * A.n_sb;
    if (!pairs) return;
    if (small) hipLaunchKernelGGL(k_huffman_sb<true>, dim3(pairs), dim3(64), 0, st, A, t0, nt);
    else hipLaunchKernelGGL(k_huffman_sb<false>, dim3(pairs < big_grid ? pairs : big_grid), dim3(64), 0, st, A, t0, nt);
}

void launch_huffman_sb(hipStream_t st, const Arena &A)
{
    launch_huffman_range(st, A, true, 0, A.qrows + 1);
    launch_huffman_range(st, A, false, 0, A.qrows + 1 + MAX_FIELDS);
}

// The same trees in two launches for the two chains of the staggered schedule (engine.cpp): quality / DNA trees need the
// sequence/quality statistics only, the numeric title field trees the title statistics only.
void launch_huffman_sq(hipStream_t st, const Arena &A, bool big_expected)
{
    launch_huffman_range(st, A, true, 0, A.qrows + 1);
    // (quality alphabets of more than 64 symbols are rare: a workgroup per CU - each needs 26 KB of LDS that the title statistics
    // kernel beside it is holding, and a thousand of them took 166 us to find out that they had nothing to do.  A context whose
    // last call met such an alphabet launches six per CU: the trees are real work then)
    launch_huffman_range(st, A, false, 0, A.qrows + 1, big_expected ? 1536 : 256);
}
void launch_huffman_fields(hipStream_t st, const Arena &A)
{
    launch_huffman_range(st, A, false, A.qrows + 1, MAX_FIELDS);
}

void launch_huffman_text(hipStream_t st, const Arena &A)
{
    hipLaunchKernelGGL(k_huffman_text, dim3(TEXT_TREES, TEXT_GRID_Y), dim3(64), 0, st, A);
}

}  // namespace ngsc
