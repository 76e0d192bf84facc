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
#include <cstdlib>

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
using TreeLdsSmall = TreeLdsT<SMALL_TREE, 160>;

// A tree is one wave's work, whatever else its workgroup holds: LDS traffic of a wave is processed in order, so a tree's
// phases need the compiler's order and the wave's lanes together again, not a workgroup barrier.
__device__ __forceinline__ void tree_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// keys[0..n) must be filled (by all lanes) before the call.  Leaves T.code/T.len/T.blob/T.blob_len.
template <class Tree>
__device__ void build_tree_wave(Tree &T, uint32_t n)
{
    const int lane = threadIdx.x & 63;
    tree_sync();
    // (round 4) The rank sort - n compares per key, n^2 / 64 LDS reads per lane: 1024 for a 256-symbol text tree, 4096 for a 512-value
    // number - only over the keys that have a frequency.  The others (frequency 0: key = symbol) are smaller than every one of those
    // and in symbol order among themselves: a key's place is the count of zero keys in front of it, a wave vote per 64 symbols.  A
    // text position holds two or three different characters, and its tree took 25 us for it: 97 us of a 36..300-base step's title chain.
    {
        uint64_t *const nz = reinterpret_cast<uint64_t *>(T.code);   // (n keys fit: code has 2 n words; it is cleared below)
        uint32_t Z = 0, M = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            const uint64_t k = i < n ? T.keys[i] : 0ull;
            const bool isz = i < n && (k >> 32) == 0, isn = i < n && (k >> 32) != 0;
            const unsigned long long mz = __ballot(isz), mn = __ballot(isn);
            if (isz) T.sorted[Z + (uint32_t)__popcll(mz & below)] = k;
            if (isn) nz[M + (uint32_t)__popcll(mn & below)] = k;
            Z += (uint32_t)__popcll(mz);
            M += (uint32_t)__popcll(mn);
        }
        tree_sync();
        for (uint32_t i = lane; i < M; i += 64) {
            const uint64_t k = nz[i];
            uint32_t rank = 0;
            for (uint32_t j = 0; j < M; ++j) rank += nz[j] < k;
            T.sorted[Z + rank] = k;
        }
        tree_sync();
    }
    // the passes over all n symbols with the whole wave (ngsc_common.h: huffman_zeros / _clear / _min_len), the merge, the code walk
    // and the blob - over the symbols that are present - by lane 0
    uint32_t zeros = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {   // sorted ascending: the zero frequencies are a prefix
        const uint32_t i = i0 + lane;
        const unsigned long long m = __ballot(i < n && (T.sorted[i < n ? i : 0] >> 32) == 0);
        zeros += (uint32_t)__popcll(m);
        if (m != ~0ull) break;
    }
    for (uint32_t i = lane; i < 2 * n - 1; i += 64) { T.code[i] = 0; T.len[i] = 0; }
    tree_sync();
    int root = 0;
    if (lane == 0) {
        uint32_t mx = 0;
        root = huffman_core(T.sorted, n, zeros, T.ifreq, T.lc, T.rc, T.code, T.len, &mx);
        T.max_len = mx;
    }
    tree_sync();
    uint32_t min_len = n;
    for (uint32_t i = lane; i < n; i += 64) {
        const uint32_t l = T.len[i];
        if (l > 0 && l < min_len) min_len = l;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)min_len, d); min_len = o < min_len ? o : min_len; }
    if (lane == 0) T.blob_len = huffman_blob(n, root, min_len, T.lc, T.rc, T.stack, T.blob);
    tree_sync();
}

// (round 3) Trees of at most 64 symbols - every quality / base tree of an ordinary file - are built in REGISTERS: the sorted leaves,
// the internal nodes' frequencies and children and the traversal stack are arrays with one element per lane, read and written with
// v_readlane / v_writelane at scalar indices, the loop state lives in scalar registers (every lane runs the same uniform code).
// The serial part of a tree - 40 merge steps, 81 nodes written out in pre-order - was ~500 dependent LDS round trips of one lane
// (~30 us, the whole launch's time since all trees of a job are in flight at once); code lengths and codes come from a parallel
// walk of every leaf up its parent chain instead of the top-down pass over the internal nodes.  Same tree, same blob
// (huffman.cpp:18-118): ties go to the smaller node id as in huffman_sequential(), which stays the path for larger alphabets.
__device__ __forceinline__ uint32_t hw_rl(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
// (an element written at a scalar index: a compare against the lane id and a select - two VALU instructions, no hazards; this
// compiler has no v_writelane builtin and the instruction takes a single SGPR operand on gfx9)
#define hw_wl(val, idx, old) ((lane == (uint32_t)(idx)) ? (uint32_t)(val) : (uint32_t)(old))

// (round 3, last step) The pre-order blob is written in PARALLEL as well: a node's bit offset in the pre-order string is the sum, over
// the edges from the root down to it, of 1 (the parent's own bit) plus - on a right edge - the bits of the left sibling's
// subtree; the merge loop keeps every internal node's subtree bits, the leaves add the sum up on the walk that already
// gives them their codes, and each leaf ORs its "1 | id" field into a zeroed bit string (internal nodes are its 0 bits).
// (A tree of a single symbol has no internal node: its blob is the one leaf.)
__device__ void build_tree_regs(TreeLdsSmall &T, uint32_t n_in)
{
    const uint32_t lane = threadIdx.x & 63;
    auto U = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    const uint32_t n = U(n_in);
    tree_sync();
    // rank sort (as in build_tree_wave): lane r ends up with the r-th smallest (frequency, symbol) key
    for (uint32_t i = lane; i < n; i += 64) {
        const uint64_t k = T.keys[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; ++j) rank += T.keys[j] < k;
        T.sorted[rank] = k;
    }
    tree_sync();
    const uint64_t key = lane < n ? T.sorted[lane] : ~0ull;
    const uint32_t vf = (uint32_t)(key >> 32), vs = (uint32_t)(key & 0xffffu);
    // compact: zero frequencies go while more than two symbols remain (huffman.cpp:44-50); sorted, so they are the first lanes
    const uint32_t zeros = (uint32_t)__popcll(__ballot(lane < n && vf == 0));
    const uint32_t drop = n > 2 ? (zeros < n - 2 ? zeros : n - 2) : 0;
    const uint32_t present = n - drop;
    // ---- merge: two queues, leaves (lane order) and internal nodes (creation order) ----
    uint32_t vif = 0, vlc = 0, vrc = 0;           // per internal node k (lane k): frequency, children (node ids)
    uint32_t par_l = 0xffu, par_i = 0xffu;        // parent (creation index) of leaf `lane` (by SYMBOL) / of internal node `lane`; 0xff: none
    uint32_t right_l = 0, right_i = 0;            // ... and whether the node is its parent's right child
    uint32_t bits_per_id = 0;
    while ((2u << bits_per_id) <= n) ++bits_per_id;
    if (n & (n - 1)) ++bits_per_id;
    const uint32_t leafbits = 1 + bits_per_id;    // a leaf in the blob: a 1 and its id
    uint32_t vsub = 0, vlsub = 0;                 // per internal node k (lane k): blob bits of its subtree / of its left child's
    uint32_t lq = drop, iq = 0, made = 0;
    for (uint32_t k = 0; k + 1 < present; ++k) {
        uint32_t f2[2], id2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t lf = lq < n ? hw_rl(vf, lq) : 0u, inf = iq < made ? hw_rl(vif, iq) : 0u;
            const bool take_leaf = (lq < n && iq < made) ? lf <= inf : lq < n;
            if (take_leaf) { f2[t] = lf; id2[t] = hw_rl(vs, lq); ++lq; }
            else { f2[t] = inf; id2[t] = n + iq; ++iq; }
        }
        vif = hw_wl(f2[0] + f2[1], made, vif);
        vlc = hw_wl(id2[0], made, vlc);
        vrc = hw_wl(id2[1], made, vrc);
        {
            const uint32_t sb0 = id2[0] < n ? leafbits : hw_rl(vsub, id2[0] - n), sb1 = id2[1] < n ? leafbits : hw_rl(vsub, id2[1] - n);
            vsub = hw_wl(1 + sb0 + sb1, made, vsub);
            vlsub = hw_wl(sb0, made, vlsub);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (id2[t] < n) { par_l = hw_wl(made, id2[t], par_l); right_l = hw_wl((uint32_t)t, id2[t], right_l); }
            else { par_i = hw_wl(made, id2[t] - n, par_i); right_i = hw_wl((uint32_t)t, id2[t] - n, right_i); }
        }
        ++made;
    }
    const uint32_t root = n + present - 2;        // (node id; for a single symbol: leaf 0)
    // ---- code of every leaf: up its parent chain (huffman.cpp:73-79 gives a child its parent's code << 1 | right) ----
    uint32_t code = 0, len = 0, off = 0;          // (off: the leaf's bit offset in the pre-order string)
    {
        uint32_t p = lane < n ? par_l : 0xffu, bit = right_l;
        while (__any(p != 0xffu)) {
            const uint32_t np = (uint32_t)__shfl((int)par_i, (int)(p & 63u)), nb = (uint32_t)__shfl((int)right_i, (int)(p & 63u));
            const uint32_t ls = (uint32_t)__shfl((int)vlsub, (int)(p & 63u));
            if (p != 0xffu) { code |= bit << len; ++len; off += 1u + (bit ? ls : 0u); p = np; bit = nb; }
        }
    }
    uint32_t mx = len, mn = (lane < n && len > 0) ? len : 0xffffffffu;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t a = (uint32_t)__shfl_xor((int)mx, d), b = (uint32_t)__shfl_xor((int)mn, d);
        mx = a > mx ? a : mx;
        mn = b < mn ? b : mn;
    }
    if (lane < n) { T.code[lane] = code; T.len[lane] = (uint8_t)len; }
    // ---- blob (huffman.cpp:88-118): root id, symbol count, shortest code, the tree in pre-order ----
    const uint32_t min_len = mn == 0xffffffffu ? n : U(mn);
    uint8_t *blob = T.blob;
    if (lane == 0) {
        blob[0] = (uint8_t)(root >> 24); blob[1] = (uint8_t)(root >> 16); blob[2] = (uint8_t)(root >> 8); blob[3] = (uint8_t)root;
        blob[4] = (uint8_t)(n >> 24); blob[5] = (uint8_t)(n >> 16); blob[6] = (uint8_t)(n >> 8); blob[7] = (uint8_t)n;
        blob[8] = (uint8_t)min_len;
    }
    if (present >= 2) {
        uint32_t *words = T.ifreq;                // (not used by this form otherwise) 64 words = 2048 bits; a tree of 64 symbols has 511
        words[lane] = 0;
        tree_sync();
        if (lane < n && len > 0) {                // a leaf of the tree (a dropped symbol has no parent)
            const uint64_t val = (uint64_t)((1u << bits_per_id) | lane) << (64u - (off & 31u) - leafbits);   // (big-endian bit order)
            atomicOr(&words[off >> 5], (uint32_t)(val >> 32));
            if ((uint32_t)val) atomicOr(&words[(off >> 5) + 1], (uint32_t)val);
        }
        tree_sync();
        const uint32_t nbytes = (hw_rl(vsub, present - 2) + 7u) >> 3;   // (the root: the last internal node made)
        for (uint32_t k = lane; k < nbytes; k += 64) blob[9 + k] = (uint8_t)(words[k >> 2] >> (24u - 8u * (k & 3u)));
        if (lane == 0) { T.blob_len = 9 + nbytes; T.max_len = U(mx); }
        tree_sync();
        return;
    }
    // one symbol: the root is leaf 0, "1 | id" and padding
    {
        const uint32_t v = (1u << bits_per_id) | root, nb = 1 + bits_per_id;   // nb <= 7 for n <= 64
        if (lane == 0) { blob[9] = (uint8_t)(v << (8 - nb)); T.blob_len = 10; T.max_len = U(mx); }
    }
    tree_sync();
}

// grid: (qrows + 1 [+ fields], n_sb); block: 64 threads (one wave).  SMALL: the quality and DNA trees of the
// sub-blocks whose alphabets fit 64 symbols; the other instance does the rest (numeric title fields, big alphabets).
template <bool SMALL, class Tree>
__device__ __forceinline__ void huffman_sb_tree(const Arena &A, Tree &T, uint32_t s, uint32_t t)
{
    const int lane = threadIdx.x & 63;
    const SbState &sb = A.sb[s];
    const bool fits_small = sb.nqual <= SMALL_TREE && sb.nsym <= SMALL_TREE;
    if (t <= A.qrows && fits_small != SMALL) return;
    if (t < A.qrows) {
        // quality tree of row t (row 0: global statistics; row k+1: position k), tasks.cpp:590-605
        if (t > sb.max_q) return;
        const uint32_t n = sb.nqual;
        const uint32_t *row = A.qhist + ((size_t)s * A.qrows + t) * 256;
        for (uint32_t j = lane; j < n; j += 64) T.keys[j] = ((uint64_t)row[sb.qualities[j]] << 32) | j;
        if constexpr (SMALL) build_tree_regs(T, n); else build_tree_wave(T, n);
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
        if constexpr (SMALL) build_tree_regs(T, n); else build_tree_wave(T, n);
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
        const FieldDesc &d = A.fdesc[(size_t)s * A.fcap + f];
        if (d.kind != KIND_NUM || !d.has_huff || d.n_sym <= 1) return;   // (a one-symbol tree: written by the field finalisation)
        const uint32_t n = d.n_sym;
        const size_t tix = (size_t)s * A.fcap + f;
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
// (round 3) The small instance: FOUR trees per workgroup, a wave each.  One single-wave workgroup per tree was bound by the rate at
// which workgroups are dispatched - 10 000 to 37 000 of them per launch left the SIMDs with less than one resident wave on
// average (SQ_WAVE_CYCLES / kernel time, profiles/r03_pmc_final_cfg2_a.json), and behind 150- and 300-base statistics nothing
// runs beside this kernel.
constexpr uint32_t SMALL_TREES_PER_WG = 4;
template <bool SMALL> struct SmallTrees {   // the small instance's static LDS (the other instance must not carry it)
    __device__ static TreeLdsSmall *get() { __shared__ TreeLdsSmall Ts[SMALL_TREES_PER_WG]; return Ts; }
};
template <> struct SmallTrees<false> { __device__ static TreeLdsSmall *get() { return nullptr; } };
// (round 4) The 512-symbol instance's tree is DYNAMIC LDS, and a predicted step (engine.cpp) does not launch the instance at all when
// the context's last job had no alphabet of more than 64 symbols (k_sq_finalize raises ERR_SHAPE if the job has one after all): with
// static LDS its workgroups waited for a CU with 26 KB to spare just to find out that they had nothing to do, 14-150 us on the side
// chain beside the title statistics kernel, which fills every CU's LDS.  (An LDS-less one-workgroup launch that only looked did that
// job for most of the round; folding the look into the finalisation kernels takes three launches off a step: 125 MB jobs
// 0.448 -> 0.438 ms, profiles/r04_ab_no_probe_launches.txt)
template <bool SMALL>
// (measured and not kept: amdgpu_waves_per_eu(5) / (6) for the register-resident tree build, which takes 97 VGPRs = four workgroups per
// CU - five fit 90 registers: steps within +-1.5 %; six spill two registers and the scratch segment that brings makes the kernel 3.5x
// slower alone; profiles/r04_ab_tree_kernel_occupancy.txt)
__global__ __launch_bounds__(SMALL ? 64 * SMALL_TREES_PER_WG : 64) void k_huffman_sb(Arena A, uint32_t t0, uint32_t nt)
{
    if (*A.err) return;
    // (the 512-symbol instance over quality / base trees only: nothing to do unless some alphabet is that big - its workgroups
    // used to find that out pair by pair, 40-60 dependent loads each: 35-64 us behind the small instance on the same stream)
    if (!SMALL && t0 + nt <= A.qrows + 1 && !*A.big_trees) return;
    if (!SMALL && t0 > A.qrows && !(*A.num_huff & 1u)) return;   // (numeric title fields only, none of them Huffman-coded)
    using Tree = typename std::conditional<SMALL, TreeLdsSmall, TreeLds>::type;
    constexpr uint32_t TPW = SMALL ? SMALL_TREES_PER_WG : 1;
    extern __shared__ __align__(16) uint8_t s_tree_dyn[];   // (the 512-symbol instance's tree: dynamic, so that a launch can do without)
    Tree &T = *(SMALL ? reinterpret_cast<Tree *>(SmallTrees<SMALL>::get() + (threadIdx.x >> 6)) : reinterpret_cast<Tree *>(s_tree_dyn));
    const uint32_t pairs = nt * A.n_sb;
    for (uint32_t p = blockIdx.x * TPW + (threadIdx.x >> 6); p < pairs; p += gridDim.x * TPW) {
        huffman_sb_tree<SMALL>(A, T, p / nt, t0 + p % nt);
        tree_sync();   // (T is used again)
    }
}

// Text-field trees: grid (129, TEXT_GRID_Y); each workgroup row walks the used slots with stride
// gridDim.y (the slot count lives on the device).  Tree j of a field exists iff text_tree_needed().
constexpr uint32_t TEXT_GRID_Y = 64;

// (round 4) ... and the numeric title fields' trees in the same launch (workgroups behind the text grid: t0 = first field tree, nt =
// fields per sub-block, as in k_huffman_sb<false>): one kernel boundary less on the title chain.
// (not launched by a predicted step whose context's last job had no title field that needs a tree: k_field_finalize)
__global__ __launch_bounds__(64) void k_huffman_text(Arena A, uint32_t text_wgs, uint32_t t0, uint32_t nt)
{
    if (*A.err) return;
    extern __shared__ __align__(16) uint8_t s_tree_dyn[];
    TreeLds &T = *reinterpret_cast<TreeLds *>(s_tree_dyn);
    if (blockIdx.x >= text_wgs) {
        if (!(*A.num_huff & 1u)) return;   // (none of the numeric fields is Huffman-coded)
        const uint32_t pairs = nt * A.n_sb, nwg = gridDim.x - text_wgs;
        for (uint32_t p = blockIdx.x - text_wgs; p < pairs; p += nwg) {
            huffman_sb_tree<false>(A, T, p / nt, t0 + p % nt);
            tree_sync();   // (T is used again)
        }
        return;
    }
    if (!(*A.num_huff & 2u)) return;   // (no text field in any sub-block)
    const uint32_t j = blockIdx.x % TEXT_TREES, by = blockIdx.x / TEXT_TREES, ny = text_wgs / TEXT_TREES;
    const int lane = threadIdx.x;
    const uint32_t used = *A.slot_counter < A.n_slots ? *A.slot_counter : A.n_slots;
    for (uint32_t slot = by; slot < used; slot += ny) {
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
        tree_sync();
    }
}

// big_grid: workgroups of the 512-symbol instance
static void launch_huffman_range(hipStream_t st, const Arena &A, bool small, uint32_t t0, uint32_t nt, uint32_t big_grid = 1024)
{
    const uint32_t pairs = nt * A.n_sb;
    if (!pairs) return;
    if (small) hipLaunchKernelGGL(k_huffman_sb<true>, dim3((pairs + SMALL_TREES_PER_WG - 1) / SMALL_TREES_PER_WG), dim3(64 * SMALL_TREES_PER_WG), 0, st, A, t0, nt);
    else hipLaunchKernelGGL(k_huffman_sb<false>, dim3(pairs < big_grid ? pairs : big_grid), dim3(64), sizeof(TreeLds), st, A, t0, nt);
}

// Two launches for the two chains of the staggered schedule (engine.cpp): quality / DNA trees need the
// sequence/quality statistics only, the numeric title field trees the title statistics only.
// with_big false (a predicted step whose context has not met a big alphabet): without the 512-symbol instance, see k_huffman_sb
void launch_huffman_sq(hipStream_t st, const Arena &A, bool big_expected, bool with_big)
{
    launch_huffman_range(st, A, true, 0, A.qrows + 1);
    // (quality alphabets of more than 64 symbols are rare: a workgroup per CU - each needs 26 KB of LDS that the title statistics
    // kernel beside it is holding, and a thousand of them took 166 us to find out that they had nothing to do.  A context whose
    // last call met such an alphabet launches six per CU: the trees are real work then)
    if (with_big) launch_huffman_range(st, A, false, 0, A.qrows + 1, big_expected ? 1536 : 256);
}
// text-field trees and numeric-field trees of the title, one launch
void launch_huffman_title(hipStream_t st, const Arena &A)
{
    const uint32_t text_wgs = TEXT_TREES * TEXT_GRID_Y, pairs = A.fcap * A.n_sb;
    const uint32_t field_wgs = pairs < 1024 ? pairs : 1024;
    hipLaunchKernelGGL(k_huffman_text, dim3(text_wgs + field_wgs), dim3(64), sizeof(TreeLds), st, A, text_wgs, A.qrows + 1, A.fcap);
}

}  // namespace ngsc
