// kernels_decode.hip - the inverse path on the GPU: sub-block bytes (info | title | quality | DNA) -> FASTQ text.
//
// Replaces the reference's uncalled decoder fragments FetchTitleHeader / FetchTitleBody (tasks.cpp:625-954), FetchDNA
// (:957-1033), FetchQuality (:1036-1101) and HuffmanEncoder::LoadTree / Decode (huffman.cpp:122-219); written from the
// bit layout of our own encoder (kernels_pack / kernels_title / kernels_huffman), which is the reference's.
//
// The format decides the shape: a sub-block is ONE bit string - the title stream has no stored length, the quality
// stream starts where it ends, the DNA stream where that ends, and every Huffman code's position depends on all codes
// before it - so the unit of parallel work is the sub-block (120 per GB of FASTQ, ~6000 in a 50 GB file), not the
// record.  One wave per sub-block: lane 0 walks the serial parts (title fields, per-position quality codes, Huffman-coded
// DNA) into scratch buffers; then all 64 lanes do what is parallel - 2-bit DNA at offsets from a prefix sum of the read
// lengths, and the assembly of the four lines of every record at offsets from a prefix sum of the record sizes.
//   k_dec_head      headers, length table, title walk per sub-block (one wave each)
//   k_dec_qspec_*   the quality stream of equally long reads, speculatively: every bit offset, then the chain, then the records
//   k_dec_tail      what is left of the serial streams: qualities of unequal reads, Huffman-coded bases
//   k_dec_assemble  per record: title \n bases \n + \n qualities \n
// Integer / bit work; bounded by dependent-load latency of one lane per wave (no MFMA, no bandwidth to speak of).
#include <cstring>
#include "ngsc_kernels.h"
#include "ngsc_swar.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace ngsc {

constexpr uint32_t DEC_MAX_FIELDS = 64;      // fields whose tables the head kernel keeps in LDS; a title with more (up to WIDE_FIELDS) has them in the sub-block's scratch pool

// ---- MSB-first bit reader over global memory with a 64-bit window (bit_stream.h:92-147) ----
// win holds the stream's bits [pos, pos + have) left aligned; pos + have is a multiple of 8 once the window has been
// opened, so it grows by whole bytes (four at a time while they last).
struct BitIn {
    const uint8_t *p;
    uint64_t nbits, pos;   // total bits, cursor
    uint64_t win;
    uint32_t have;
    uint32_t bad;          // 1 malformed stream, 2 scratch pool exhausted, 3 text capacity
    __device__ void init(const uint8_t *ptr, uint64_t nbytes) { p = ptr; nbits = nbytes * 8; pos = 0; win = 0; have = 0; bad = 0; }
    __device__ void seek(uint64_t bitpos) { pos = bitpos; win = 0; have = 0; }
    __device__ __noinline__ void fill()
    {
        if (have == 0 && (pos & 7) && pos < nbits) {   // open the window in the middle of a byte
            win = ((uint64_t)p[pos >> 3] << 56) << (pos & 7);
            have = 8 - (uint32_t)(pos & 7);
        }
        const uint64_t nbytes = nbits >> 3;
        while (have <= 32) {
            const uint64_t byte = (pos + have) >> 3;
            if (byte + 4 <= nbytes) {
                win |= (uint64_t)__builtin_bswap32(reinterpret_cast<const U1u *>(p + byte)->x) << (32 - have);
                have += 32;
            } else if (byte < nbytes) {
                win |= (uint64_t)p[byte] << (56 - have);
                have += 8;
            } else break;
        }
    }
    __device__ __forceinline__ uint32_t bits(uint32_t n)   // n <= 32
    {
        if (!n) return 0;
        if (pos + n > nbits) { bad = bad ? bad : 1; pos = nbits; have = 0; win = 0; return 0; }
        if (have < n) fill();
        const uint32_t v = (uint32_t)(win >> (64 - n));
        win <<= n;
        have -= n;
        pos += n;
        return v;
    }
    __device__ __forceinline__ uint32_t peek(uint32_t n)   // the next n <= 16 bits without moving (zeros behind the end of the stream)
    {
        if (have < n) fill();
        return (uint32_t)(win >> (64 - n));
    }
    __device__ __forceinline__ void skip(uint32_t n)       // n bits that peek() has shown
    {
        if (!n) return;
        if (pos + n > nbits) { bad = bad ? bad : 1; pos = nbits; have = 0; win = 0; return; }
        win <<= n;
        have -= n;
        pos += n;
    }
    __device__ __forceinline__ uint32_t bit() { return bits(1); }
    __device__ void align()
    {
        const uint32_t r = (uint32_t)(pos & 7);
        if (r) (void)bits(8 - r);
    }
    __device__ uint32_t u8() { align(); return bits(8); }
    __device__ uint32_t u32() { align(); return bits(32); }
};

// ---- Huffman tree from its pre-order blob (huffman.cpp:88-118 / 122-188): child tables, 16-bit node ids ----
struct DTree {
    uint32_t n, root;       // symbols; root node id (leaf ids 0..n-1, internal ids from n in visit order)
    uint16_t *lc, *rc;      // [n] indexed by node id - n
    uint16_t *tab;          // [1 << DEC_K] by the next DEC_K bits: length << 12 | symbol; 0 = code longer than DEC_K bits (or
                            // symbol 0 of a one-symbol tree): walk the tree.  nullptr: no table
};
constexpr uint32_t DEC_K = 9;

// Decode table of a loaded tree: a dependent load per SYMBOL instead of one per code bit (a lone lane pays the full
// latency of each).  Depth-first over the codes of at most DEC_K bits; entry = node | len << 10 | code << 14.
// (map: entries carry map[symbol] instead of the symbol - the symbol streams' tables, whose consumers want the byte)
__device__ void tree_table(DTree &t, uint16_t *tab, uint32_t K = DEC_K, const uint8_t *map = nullptr)
{
    t.tab = K == DEC_K ? tab : nullptr;   // (tree_get() reads DEC_K-bit tables; the quality loop addresses its own)
    if (!tab) return;
    for (uint32_t x = 0; x < (1u << K); ++x) tab[x] = 0;
    uint32_t stack[2 * DEC_K + 4];
    uint32_t sp = 0;
    stack[sp++] = t.root;
    while (sp) {
        const uint32_t e = stack[--sp];
        const uint32_t node = e & 1023u, len = (e >> 10) & 15u, code = e >> 14;
        if (node < t.n) {
            const uint32_t span = 1u << (K - len), first = code << (K - len);
            const uint16_t v = (uint16_t)((len << 12) | (map ? map[node & 0xffu] : node));
            for (uint32_t x = 0; x < span; ++x) tab[first + x] = v;
        } else if (len < K) {
            const uint32_t idx = node - t.n;
            stack[sp++] = t.rc[idx] | ((len + 1) << 10) | (((code << 1) | 1u) << 14);
            stack[sp++] = t.lc[idx] | ((len + 1) << 10) | ((code << 1) << 14);
        }
    }
}

struct Pool {   // bump allocator over the sub-block's scratch
    uint8_t *base;
    uint64_t off, cap;
    __device__ void *take(uint64_t bytes)
    {
        const uint64_t o = (off + 7) & ~7ull;
        if (o + bytes > cap) return nullptr;
        off = o + bytes;
        return base + o;
    }
};

// lc_pre / rc_pre: child tables of n + 1 entries each that the caller has set aside (else they come from the pool)
__device__ bool tree_load(BitIn &r, uint32_t expect_n, Pool &pool, DTree &t, uint16_t *stack /* 2n + 2 */, uint16_t *lc_pre = nullptr,
                          uint16_t *rc_pre = nullptr)
{
    const uint32_t blen = r.u32();
    const uint64_t end = r.pos + (uint64_t)blen * 8;
    const uint32_t root = r.u32(), n = r.u32();
    (void)r.u8();
    if (r.bad || n != expect_n || n == 0 || n > 512 || root >= 2 * n || end > r.nbits) { r.bad = 1; return false; }
    t.n = n;
    t.lc = lc_pre ? lc_pre : (uint16_t *)pool.take(2 * (n + 1));
    t.rc = rc_pre ? rc_pre : (uint16_t *)pool.take(2 * (n + 1));
    if (!t.lc || !t.rc) { r.bad = 2; return false; }
    uint32_t bits_per_id = 0;
    while ((2u << bits_per_id) <= n) ++bits_per_id;
    if (n & (n - 1)) ++bits_per_id;
    // slots waiting for a child: idx * 2 = left of internal node idx, idx * 2 + 1 = right
    uint32_t made = 0, sp = 0, rootref = 0;
    bool first = true;
    for (;;) {
        uint32_t node;
        if (r.bit()) node = r.bits(bits_per_id);
        else {
            if (made >= n) { r.bad = 1; break; }
            node = n + made++;
        }
        if (r.bad) break;
        if (first) { rootref = node; first = false; }
        else {
            const uint32_t slot = stack[--sp];
            if (slot & 1) t.rc[slot >> 1] = (uint16_t)node; else t.lc[slot >> 1] = (uint16_t)node;
        }
        if (node >= n) {
            const uint32_t idx = node - n;
            stack[sp++] = (uint16_t)(idx * 2 + 1);
            stack[sp++] = (uint16_t)(idx * 2);
        }
        if (!sp) break;
    }
    t.root = rootref;
    t.tab = nullptr;
    if (r.bad) return false;
    r.seek(end);
    return true;
}

__device__ __forceinline__ uint32_t tree_get(const DTree &t, BitIn &r)
{
    if (t.tab) {
        const uint32_t e = t.tab[r.peek(DEC_K)];
        if (e) { r.skip(e >> 12); return e & 0xfffu; }
    }
    uint32_t node = t.root;
    while (node >= t.n) {
        const uint32_t idx = node - t.n;
        node = r.bit() ? t.rc[idx] : t.lc[idx];
        if (r.bad) return 0;
    }
    return node;
}

struct DField {
    uint8_t sep, is_const, is_num, is_len_const, delta_coding, gflag, has_g, pad;
    uint32_t len0, min_len, max_len, bits_len, bits_num, bits_val;
    int32_t min_v, max_v, min_d, max_d, prev;
    uint8_t *data0, *ham, *gval;
    uint32_t gval_len;
    DTree g;
    DTree *l;   // [MAX_STAT_LEN + 1] (n == 0: no tree at that position)
};

// What the assembly kernel needs of a decoded sub-block (device memory, one per sub-block)
struct DecSub {
    uint32_t n_rec, max_q, flags, nsym;
    uint32_t plain;
    int32_t err;
    uint64_t dna_bitpos;        // where the DNA payload starts in the sub-block (plain DNA is decoded by the assembly kernel)
    uint64_t text_bytes;
    uint64_t t_head, t_qual, t_dna;   // wall_clock64 ticks (100 MHz) of the three phases, for NGSC_VERBOSE
    // titles are rendered by the assembly kernel, all threads at once, from what the serial walk over the title body left:
    uint32_t nf, nvw;           // fields; value words per record
    uint32_t spec, pad_;        // the quality stream went through the speculative decoder
    uint64_t t_ph[6];           // head kernel, ticks: info + length table, title header, title body, title lengths, quality header
    uint64_t af_off;            // AField[nf] at pool + af_off
    uint64_t tchar_off;         // decoded characters of the varying text fields, in the text area
    uint8_t dsym[256];
};

// One title field as the renderer needs it.  kind 0: constant (len0 bytes at pool + data_off); 1: numeric, value word vidx of the
// record; 2: text, words vidx (offset of its characters in the character area) and vidx + 1 (length)
struct AField { uint32_t kind, sep, len0, vidx; uint64_t data_off; };

// A varying field as the serial walk over the title body needs it (LDS)
struct VField {
    uint32_t f, kind, vidx, delta, has_g, bits_val, bits_num, bits_len, is_len_const, len0, min_len, max_len;
    int32_t min_v, min_d;
};

__device__ __forceinline__ uint32_t dec_ndigits(uint32_t v)
{
    return 1u + (v >= 10u) + (v >= 100u) + (v >= 1000u) + (v >= 10000u) + (v >= 100000u) + (v >= 1000000u) + (v >= 10000000u) +
           (v >= 100000000u) + (v >= 1000000000u);
}

// decimal digits of v straight into dst (no local array: a lone lane's scratch memory costs a memory round trip per byte)
__device__ __forceinline__ uint32_t dec_put_decimal(uint8_t *dst, uint32_t v)
{
    uint32_t nd = 1;
    for (uint32_t t = v; t >= 10; t /= 10) ++nd;
    for (uint32_t k = nd; k-- > 0;) { dst[k] = (uint8_t)('0' + v % 10); v /= 10; }
    return nd;
}

// scratch layout of one sub-block: [qlen n][toff n+1][slen n][lay 2][text tcap][pool].  The text area holds the titles,
// then - where they end, which is known before the quality stream is opened - the qualities, then the bases of a
// Huffman-coded alphabet: together never more than the sub-block's FASTQ text.
struct DecScratch {
    uint32_t *qlen, *toff, *slen, *rs;   // rs: start bit of every record's qualities, relative to the stream's start (+ its end)
    uint64_t *lay;          // [0] offset of the qualities in the text area, [1] of the bases
    uint8_t *titles;
    uint8_t *pool;
    uint64_t tcap, pool_cap;
};

__device__ DecScratch dec_scratch(uint8_t *base, uint64_t stride, uint32_t s, uint32_t rec_cap, uint64_t tcap, uint64_t)
{
    DecScratch D;
    uint8_t *b = base + (uint64_t)s * stride;
    D.qlen = (uint32_t *)b; b += 4ull * rec_cap;
    D.toff = (uint32_t *)b; b += 4ull * (rec_cap + 1);
    D.slen = (uint32_t *)b; b += 4ull * rec_cap;
    D.rs = (uint32_t *)b; b += 4ull * (rec_cap + 1);
    b = (uint8_t *)(((uintptr_t)b + 15) & ~(uintptr_t)15);
    D.lay = (uint64_t *)b; b += 16;
    D.titles = b; b += tcap;
    D.pool = b;
    D.tcap = tcap;
    D.pool_cap = stride - (uint64_t)(b - (base + (uint64_t)s * stride));
    return D;
}

// ---- the long symbol streams (qualities; Huffman-coded bases): lane 0 decodes, the wave feeds it ------------------------
// A lone lane issues one dependent instruction every few cycles and waits a full memory latency for every byte of the
// stream it fetches itself and for every record length - the first version spent ~2000 cycles per symbol that way.  Here
// the WAVE stages the next DEC_IN_DW dwords of the bit string in LDS (byte-swapped: a dword = 32 stream bits, MSB first),
// lane 0 decodes from that with a 64-bit window - per symbol: index, one LDS table read, shift - into an LDS output buffer,
// and the wave writes the buffer out, coalesced.  Rounds end when the staged input or the output buffer is used up.
constexpr uint32_t DEC_IN_DW = 512;      // 2 KB of bit string per round (+ 2 dwords of look-ahead)
constexpr uint32_t DEC_OUT = 6144;       // symbols per round at most

struct DecCtl {   // state of one sub-block's decode: lane 0 -> wave inside a kernel (an LDS copy), kernel -> kernel (global)
    int32_t err;
    uint32_t n, max_q, Kq, q_lds, nqual, nf, lt_bits;
    uint64_t pos;            // bit position behind what has been consumed
    uint64_t qual_total, text, qual_total_bases;
    const uint16_t *qtab;    // per-position tables of Kq bits in global memory (the pool)
    const DTree *qt;
    uint32_t spec;           // every read has max_q qualities: the speculative decoder (k_dec_qspec_*) takes the quality stream
    uint32_t spec_bad;       // ... and found something it does not handle: the tail kernel decodes the stream serially
    uint64_t cov_lo, cov_hi; // hops computed so far / being computed: bit offsets relative to the stream's start
    uint64_t chain_cur;      // where the chain of record starts stands (relative), and at which record
    uint32_t chain_i, pad1_;
    uint64_t pool_off;       // bump offset of the pool behind the head kernel
    uint64_t t_head;
    // Huffman-coded bases, decoded in parallel (k_dec_dna_*): 0 not taken, 1 waiting for the kernels, 2 decoded, 3 they gave up
    uint32_t dna_spec, dna_maxlen, dna_nseg, dna_bad;
    uint64_t dna_p0, dna_total;      // first bit of the code stream; symbols in it
    const uint16_t *dna_tab;         // its DEC_K-bit table in global memory (entries carry the output byte)
    uint32_t dna_n, dna_root;        // the tree, for the serial fallback
    uint16_t *dna_lc, *dna_rc;
    uint64_t t_qual;
    uint8_t qsym[256];       // quality alphabet: symbol index -> byte
};

// vdst's lane `lane_imm` := the scalar s (v_writelane_b32; this compiler has no builtin for it)
__device__ __forceinline__ uint32_t dec_writelane(uint32_t v, uint32_t s, int lane_imm)
{
    asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(lane_imm));
    return v;
}

// lens[i] symbols for record i = 0..n-1; PER_POS: symbol j of a record uses table row j + 1 (tasks.cpp:609-619) and its
// tree trees[j + 1]; else one table / trees[0].  map: symbol index -> output byte (LDS).  kept (nullable): per record, the
// number of output bytes below 128.  Returns through C.pos / C.err; all lanes call it.
template <bool PER_POS, uint32_t TS = 12>   // TS: symbols per trip at most
__device__ __forceinline__ void dec_symbols_wave(const uint8_t *p, uint64_t nbits, DecCtl &C, const uint16_t *tab, uint32_t K, const DTree *trees,
                                                 const uint32_t *lens, uint32_t n, uint8_t *dst, uint32_t *kept, const uint8_t *map, uint32_t *s_in,
                                                 uint8_t *s_out, uint32_t nrows, const uint8_t *t_lc = nullptr, const uint8_t *t_rc = nullptr,
                                                 const uint8_t *t_root = nullptr, uint32_t t_nq = 0, bool tab_lds = false)
{
    // tab_lds: the per-position table lies in LDS (rows behind a record's last symbol may then be read and thrown away)
    // t_lc / t_rc / t_root (LDS, optional): the rows' trees as byte child tables [row][t_nq] (alphabets of up to 128 symbols) - a code
    // longer than the table's K bits is then walked inside the chain, out of LDS, instead of ending the trip for a walk in global memory
    const uint32_t lane = threadIdx.x;
    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(p);   // (sub-blocks start 16-byte aligned and are padded to 16)
    const uint32_t ndw = (uint32_t)((nbits + 31) >> 5);
    // the decoder's state: a window of four dwords of the bit string (w0 first), sh bits of w0 already consumed
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0, sh = 0;
    // (values that come from memory go through readfirstlane: that is what tells the compiler they are wave-uniform)
    auto U = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    const uint32_t pos_lo = U((uint32_t)C.pos), pos_hi = U((uint32_t)(C.pos >> 32));
    const uint64_t pos0 = ((uint64_t)pos_hi << 32) | pos_lo;
    uint32_t wi = (uint32_t)(pos0 >> 5), i = 0, j = 0, L = 0, kc = 0, bad = 0, Lnext = 0;
    uint64_t kbase = 0;
    const uint32_t skip0 = (uint32_t)(pos0 & 31);   // the stream starts inside dword wi
    n = U(n); K = U(K);
    bool opened = false;
    // Every lane runs the decoder, on the same data: nothing in the loop depends on the lane but the table lookups below, so its
    // values are wave-uniform, the compiler keeps them in scalar registers and branches without exec-mask bookkeeping (a loop
    // under `if (lane == 0)` was ~190 instructions per symbol, most of them mask handling); stores of identical values to one
    // address are harmless.
    // record lengths: the one after next is fetched (into a vector register: no wait) when a record begins, and has that record's
    // whole decode to arrive - a length read where it was needed cost the chain a trip to memory per record
    uint32_t Lpre = 0;
    if (n) { L = U(lens[0]); Lnext = n > 1 ? U(lens[1]) : 0; Lpre = n > 2 ? lens[2] : 0; }
    for (;;) {
        const uint32_t base = wi;
        for (uint32_t t = lane; t < DEC_IN_DW + 4; t += 64) s_in[t] = base + t < ndw ? __builtin_bswap32(p32[base + t]) : 0u;
        __syncthreads();
        uint32_t ko = 0;
        {
            const uint32_t limit = base + DEC_IN_DW;
            if (!opened) {   // the first dwords; the bits in front of the stream count as consumed
                w0 = U(s_in[0]); w1 = U(s_in[1]); w2 = U(s_in[2]); w3 = U(s_in[3]);
                sh = skip0;
                wi += 4;
                opened = true;
            }
            // nb bits leave the window (whole dwords come in from the staged bit string)
            // (the dword that comes in next is fetched - into a vector register, no wait - as soon as its place is known)
            uint32_t wpre = s_in[wi - base];
            auto consume = [&](uint32_t nb) __attribute__((always_inline)) {
                sh += nb;
                while (sh >= 32) { w0 = w1; w1 = w2; w2 = w3; w3 = U(wpre); ++wi; sh -= 32; wpre = s_in[wi - base]; }
            };
            while (i < n && wi < limit && ko + TS <= DEC_OUT && !bad) {
                if (j == L) {   // next record (also the empty ones)
                    if (kept) kept[i] = kc;
                    kc = 0; j = 0; ++i;
                    L = Lnext;
                    Lnext = U(Lpre);
                    Lpre = i + 2 < n ? lens[i + 2] : 0;
                    continue;
                }
                // (round 4) Up to TS symbols per trip.  Lane o stands for bit offset o of the window: it looks up, for each of the next
                // T symbols, the entry that symbol has IF it starts at offset o - T independent table reads per lane, in flight
                // together.  The chain is then resolved with scalar reads: the entry of symbol t sits in the lane its predecessor's
                // code ends at.  (Round 3 looked four symbols ahead, with the lanes split among them: a trip - a table latency, the
                // scalar chain, the refill - cost ~1100 cycles whatever it decoded, 275 per symbol; the reads of unequal length that the
                // speculative kernels do not take decoded at 0.4 GB/s.)
                const uint32_t T = L - j < TS ? L - j : TS;
                uint32_t ev[TS];
                {
                    const uint32_t so = sh + lane, d = so >> 5, r = so & 31u;           // (so <= 94: dwords d, d + 1 of the four)
                    const uint32_t hi = d == 0 ? w0 : (d == 1 ? w1 : w2), lo = d == 0 ? w1 : (d == 1 ? w2 : w3);
                    // the K bits at offset so: the top of (hi:lo) << r, with 32-bit funnel shifts
                    const uint32_t top = r ? __builtin_amdgcn_alignbit(hi, lo, 32u - r) : hi;
                    const uint32_t field = top >> (32u - K);
                    if (!PER_POS || (tab_lds && j + TS < nrows)) {
                        // rows j + 1 .. j + TS, all of them inside the table: one address and a stride (what lies behind the
                        // record's last symbol is read and never looked at)
                        const uint16_t *pl = tab + (PER_POS ? ((size_t)(j + 1) << K) : (size_t)0) + field;
                        const uint32_t rs = PER_POS ? (1u << K) : 0u;
#pragma unroll
                        for (uint32_t t = 0; t < TS; ++t) ev[t] = pl[(size_t)t * rs];
                    } else {
#pragma unroll
                        for (uint32_t t = 0; t < TS; ++t) {
                            ev[t] = 0;
                            if (t < T) {
                                const uint32_t row = PER_POS ? (j + 1 + t < nrows ? j + 1 + t : nrows - 1) : 0;
                                ev[t] = tab[((size_t)row << K) + field];
                            }
                        }
                    }
                }
                // (round 4, second step) The chain is branch-free: lane o of nx[t] holds the offset behind symbol t if it starts at o
                // (o itself where the entry is an escape: the chain then stays put), and symbol t + 1's start is ONE v_readlane of
                // nx[t] at symbol t's start - twelve dependent instructions.  Which symbols were really there is decided afterwards,
                // in vector form: lane t gets symbol t's start and entry (written with v_writelane), a symbol counts while every one
                // before it had an entry and started inside the window.  (The scalar loop this replaces tested, branched and stored
                // per symbol: ~30 scalar instructions and four taken branches each, 400 cycles per symbol with one wave on the SIMD.)
                uint32_t so[TS + 1];
                so[0] = 0;
                uint32_t ovec = 0, evec = 0;
#pragma unroll
                for (uint32_t t = 0; t < TS; ++t) {
                    const uint32_t nx = lane + (ev[t] >> 12);
                    const uint32_t at = so[t] & 63u;
                    so[t + 1] = (uint32_t)__builtin_amdgcn_readlane((int)nx, (int)at);   // (behind a symbol that starts beyond the window: anything - it does not count, nor do its successors)
                    const uint32_t e_t = (uint32_t)__builtin_amdgcn_readlane((int)ev[t], (int)at);
                    ovec = dec_writelane(ovec, so[t], t);
                    evec = dec_writelane(evec, e_t, t);
                }
                const bool there = lane < T && ovec < 64u && (evec >> 12) != 0;
                const unsigned long long gone = ~__ballot(there);
                uint32_t cnt = (uint32_t)__builtin_ctzll(gone);   // (lanes >= T are never there: gone != 0)
                uint32_t o_end = 0;
                if (cnt) {
                    const uint32_t sym = evec & 0xffu;
                    if (lane < cnt) s_out[ko + lane] = (uint8_t)sym;
                    kc += (uint32_t)__popcll(__ballot(lane < cnt && sym < 128u));
                    o_end = (uint32_t)__builtin_amdgcn_readlane((int)(ovec + (evec >> 12)), (int)(cnt - 1));
                }
                if (t_lc && cnt < T && o_end < 64u) {
                    // the symbol the chain stopped at has a code longer than K bits (or its tree one symbol): the row's tree out of LDS,
                    // bit by bit out of the window - from the node its first K bits lead to where the table knows it (entry 0x800 |
                    // internal node, dec_mark_long_codes), else from the root
                    const uint32_t e_x = (uint32_t)__builtin_amdgcn_readlane((int)evec, (int)cnt);
                    const uint32_t row = PER_POS ? (j + 1 + cnt < nrows ? j + 1 + cnt : nrows - 1) : 0, rowo = row * t_nq;
                    const bool from_node = (e_x & 0x800u) != 0;
                    uint32_t node = from_node ? t_nq + (e_x & 0x7ffu) : U(t_root[row]), len = from_node ? K : 0u;
                    bool ok = true;
                    while (node >= t_nq) {
                        const uint32_t bp = sh + o_end + len;
                        if (bp >= 128 || len >= 32) { ok = false; break; }
                        const uint32_t wsel = bp < 32 ? w0 : (bp < 64 ? w1 : (bp < 96 ? w2 : w3));
                        node = U(((wsel << (bp & 31u)) >> 31) ? t_rc[rowo + node - t_nq] : t_lc[rowo + node - t_nq]);
                        ++len;
                    }
                    if (ok) {
                        const uint32_t sym = U(map[node & 0xffu]);
                        s_out[ko + cnt] = (uint8_t)sym;
                        kc += sym < 128;
                        o_end += len;
                        ++cnt;
                    } else if (!cnt) bad = 1;   // (with symbols in hand the window moves on and the next trip looks again)
                }
                if (cnt) consume(o_end);
                else if (t_lc) { if (!bad) bad = 1; }   // (cannot happen: the first symbol of a trip starts at offset 0, inside the window)
                else {   // ... or in global memory
                    const DTree &Tr = trees[PER_POS ? j + 1 : 0];
                    const uint32_t tn = U(Tr.n);
                    uint32_t node = U(Tr.root), steps = 0;
                    while (node >= tn) {
                        if (wi > limit + 2 || ++steps > 64) { bad = 1; break; }
                        const uint32_t idx = node - tn;
                        node = U(((w0 << sh) >> 31) ? Tr.rc[idx] : Tr.lc[idx]);
                        consume(1);
                    }
                    const uint32_t sym = U(map[node & 0xffu]);
                    s_out[ko] = (uint8_t)sym;
                    kc += sym < 128;
                    cnt = 1;
                }
                ko += cnt; j += cnt;
            }
            if (i < n && j == L && i + 1 == n) { if (kept) kept[i] = kc; ++i; }   // the last record closes without a successor
            const uint64_t pos = (uint64_t)(wi - 4) * 32 + sh;
            if (pos > nbits) bad = 1;
            if (lane == 0) {
                C.pos = pos;
                if (bad) C.err = ERR_FORMAT;
            }
        }
        __syncthreads();
        for (uint32_t t = lane; t < ko; t += 64) dst[kbase + t] = s_out[t];
        kbase += ko;
        const bool done = i >= n || bad;   // (computed by every lane alike: a value read back from LDS would count as divergent)
        __syncthreads();
        if (done) break;   // (else: lane 0 keeps its window; wi points behind it, and the next round stages from wi)
    }
}

// Lane 0: the info header (phyNGSC.cpp:719-742).  The length table behind it is read by the whole wave (fixed-width entries).
__device__ int dec_info(BitIn &r, DecScratch &D, DecSub &O, uint32_t rec_cap, DecCtl &C)
{
    const uint32_t n = r.u32(), max_q = r.u32();
    (void)r.u32();
    const uint32_t nsym = r.u8();
    (void)r.u8();
    const uint32_t nqual = r.u8(), flags = r.u32();
    if (r.bad || n == 0 || n > rec_cap || !nsym || !nqual || max_q > (1u << 24)) return ERR_FORMAT;
    if (flags & F_USE_DELTA) return ERR_COLORSPACE;   // colour space is stored lossily (phyNGSC.cpp:531): nothing to give back
    O.n_rec = n; O.max_q = max_q; O.flags = flags; O.nsym = nsym; O.plain = (flags & F_DNA_PLAIN) ? 1u : 0u;
    const uint32_t b = bit_length(max_q);
    C.n = n; C.max_q = max_q; C.nqual = nqual; C.lt_bits = b > 32 ? 32 : b; C.pos = r.pos;
    if (r.pos + (uint64_t)n * C.lt_bits > r.nbits) return ERR_FORMAT;
    (void)D;
    return OK;
}

// w <= 32 bits at bit position bitpos of the stream (any lane, any position)
__device__ __forceinline__ uint32_t dec_bits_at(const uint8_t *p, uint64_t nbytes, uint64_t bitpos, uint32_t w)
{
    if (!w) return 0;
    const uint64_t byte = bitpos >> 3;
    uint64_t x = 0;
    if (byte + 8 <= nbytes) {
        const U2u v = *reinterpret_cast<const U2u *>(p + byte);
        x = ((uint64_t)__builtin_bswap32(v.x) << 32) | __builtin_bswap32(v.y);
    } else {
        for (uint32_t k = 0; k < 8; ++k) x = (x << 8) | (byte + k < nbytes ? p[byte + k] : 0u);
    }
    return (uint32_t)((x << (bitpos & 7)) >> (64 - w));
}

// Lane 0: title header.  Returns an error code (0: go on).
__device__ int dec_head(BitIn &r, DecScratch &D, Pool &pool, DecSub &O, DField *F, DecCtl &C, uint16_t *stack)
{
    const uint32_t n = C.n;
    // ---- title header (cf. FetchTitleHeader tasks.cpp:625-735; value-coding trees as the encoder stores them, Q2b) ----
    const uint32_t nf = r.u32();
    if (r.bad || nf == 0 || nf > WIDE_FIELDS) return r.bad ? ERR_FORMAT : ERR_UNSUPPORTED;
    const uint32_t keep_from = num_values_kept_from(nf);
    for (uint32_t f = 0; f < nf && !r.bad; ++f) {
        DField &fd = F[f];
        fd.l = nullptr; fd.has_g = 0; fd.gval = nullptr; fd.data0 = fd.ham = nullptr; fd.gval_len = 0; fd.prev = 0; fd.gflag = 0;
        fd.is_num = fd.is_len_const = fd.delta_coding = 0;
        fd.len0 = fd.min_len = fd.max_len = fd.bits_len = fd.bits_num = fd.bits_val = 0;
        fd.min_v = fd.max_v = fd.min_d = fd.max_d = 0;
        fd.sep = (uint8_t)r.u8();
        fd.is_const = r.u8() != 0;
        if (fd.is_const) {
            fd.len0 = r.u32();
            if (r.bad || (uint64_t)fd.len0 * 8 > r.nbits) { r.bad = 1; break; }
            fd.data0 = (uint8_t *)pool.take(fd.len0 + 1);
            if (!fd.data0) { r.bad = 2; break; }
            r.align();
            for (uint32_t j = 0; j < fd.len0; ++j) fd.data0[j] = (uint8_t)r.bits(8);
            continue;
        }
        fd.is_num = r.u8() != 0;
        if (fd.is_num) {
            fd.min_v = (int32_t)r.u32(); fd.max_v = (int32_t)r.u32(); fd.min_d = (int32_t)r.u32(); fd.max_d = (int32_t)r.u32();
            const int64_t vr = (int64_t)fd.max_v - fd.min_v, dr = (int64_t)fd.max_d - fd.min_d;
            fd.delta_coding = !(vr < dr);
            const int64_t diff = fd.delta_coding ? dr : vr;
            fd.bits_num = bit_length((uint64_t)diff);
            fd.bits_val = bit_length((uint64_t)vr);
            const bool have_map = fd.delta_coding ? (n >= 2u) : (f >= keep_from);
            if (diff + 1 <= (int64_t)MAX_NUM_HUF && have_map) {
                fd.has_g = tree_load(r, (uint32_t)(diff + 1), pool, fd.g, stack) ? 1 : 0;
                if (fd.has_g) tree_table(fd.g, (uint16_t *)pool.take(2u << DEC_K));
            }
            continue;
        }
        fd.is_len_const = r.u8() != 0;
        fd.len0 = r.u32(); fd.max_len = r.u32(); fd.min_len = r.u32();
        if (r.bad || (uint64_t)fd.len0 * 8 > r.nbits || fd.max_len > (1u << 24) || fd.min_len > fd.max_len) { r.bad = 1; break; }
        fd.bits_len = bit_length((uint64_t)(fd.max_len - fd.min_len));
        fd.data0 = (uint8_t *)pool.take(fd.len0 + 1);
        fd.ham = (uint8_t *)pool.take(fd.len0 + 1);
        fd.gval = (uint8_t *)pool.take(fd.max_len + 1);
        fd.l = (DTree *)pool.take(sizeof(DTree) * (MAX_STAT_LEN + 1));
        if (!fd.data0 || !fd.ham || !fd.gval || !fd.l) { r.bad = 2; break; }
        for (uint32_t j = 0; j <= MAX_STAT_LEN; ++j) fd.l[j].n = 0;
        r.align();
        for (uint32_t j = 0; j < fd.len0; ++j) fd.data0[j] = (uint8_t)r.bits(8);
        for (uint32_t j = 0; j < fd.len0; ++j) fd.ham[j] = (uint8_t)r.bit();
        const uint32_t npos = fd.max_len < MAX_STAT_LEN ? fd.max_len : MAX_STAT_LEN;
        for (uint32_t j = 0; j < npos && !r.bad; ++j)
            if (j >= fd.len0 || !fd.ham[j]) { if (tree_load(r, 256, pool, fd.l[j], stack)) tree_table(fd.l[j], (uint16_t *)pool.take(2u << DEC_K)); }
        if (fd.max_len >= MAX_STAT_LEN && !r.bad && tree_load(r, 256, pool, fd.l[MAX_STAT_LEN], stack))
            tree_table(fd.l[MAX_STAT_LEN], (uint16_t *)pool.take(2u << DEC_K));
        r.align();
    }
    if (r.bad) return r.bad == 2 ? ERR_NOMEM : ERR_FORMAT;

    // (the title body is walked by the whole wave, scalarised: dec_title_body_wave)
    C.pos = r.pos; C.nf = nf;
    (void)D; (void)O;
    return OK;
}

// The whole wave, behind the title body: quality alphabet and trees.
// (round 4) The max_q + 1 trees are loaded by the LANES, a tree each: lane 0 reads the alphabet and hops from blob to blob over their
// length words, then every lane parses its trees' pre-order strings and fills their tables.  One lane doing all of it took 120 us
// per tree: 37 ms of a 36..300-base sub-block's decode.
__device__ int dec_head_quality(BitIn &r, DecScratch &D, Pool &pool, uint8_t *qsym, DecCtl &C, uint32_t *s_x /* 8 words of LDS */)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t max_q = C.max_q, nqual = C.nqual;
    const uint64_t qual_total = C.qual_total;
    const uint64_t rows = (uint64_t)max_q + 1;
    // s_x: [0] error, [1] Kq, [2..3] qt, [4..5] qtab, [6..7] the lanes' work area
    if (lane == 0) {
        int err = OK;
        // ---- quality (cf. FetchQuality tasks.cpp:1036-1101): alphabet, max_q + 1 trees; the codes are decoded by the wave ----
        const uint64_t text = C.text;   // bytes of the text area the value table and the text fields' characters take
        const uint64_t q_at = (text + 15) & ~15ull, s_at = q_at + ((qual_total + 15) & ~15ull);
        if (s_at + qual_total > D.tcap) err = ERR_CAPACITY;
        D.lay[0] = q_at; D.lay[1] = s_at;
        r.align();
        for (uint32_t i = 0; i < 256; ++i) qsym[i] = 0;
        for (uint32_t i = 0; i < nqual; ++i) qsym[i] = (uint8_t)r.bits(8);
        DTree *qt = (DTree *)pool.take(sizeof(DTree) * rows);
        // per tree: two child tables of nqual + 1 entries; per lane: a traversal stack of 2 nqual + 2 entries; a start position per tree
        const uint64_t per_tree = (4ull * (nqual + 1) + 7) & ~7ull, per_lane = (2ull * (2 * nqual + 2) + 7) & ~7ull;
        uint8_t *work = (uint8_t *)pool.take(rows * per_tree + 64 * per_lane + 8 * rows);
        // the per-position tables, DEC_K bits wide, in the pool: the kernels that decode the symbols copy what fits into LDS
        // (narrower if need be: dec_fill_lds_table)
        // (as wide as the pool allows next to what the base stream needs later: reads of 3 000 bases and more have 3 001 trees,
        // whose 9-bit tables alone are 3.4 MB of the 4 MB pool; found by the round-3 fuzz run, seeds 400796 / 400922:
        // NGSC_ERR_NOMEM for such files)
        uint32_t Kq = DEC_K;
        {
            const uint64_t keep = 96u << 10;
            const uint64_t room = pool.cap > pool.off + keep ? pool.cap - pool.off - keep : 0;
            while (Kq > 5 && ((rows << Kq) * 2) > room) --Kq;
        }
        uint16_t *qtab = (uint16_t *)pool.take((rows << Kq) * 2);
        if (!err && (!qt || !work || !qtab)) err = ERR_NOMEM;
        if (!err) {
            uint64_t *start = (uint64_t *)(work + rows * per_tree + 64 * per_lane);
            for (uint32_t p = 0; p <= max_q; ++p) {   // blob p: a length word, then that many bytes
                r.align();
                start[p] = r.pos;
                const uint32_t blen = r.u32();
                if (r.bad || r.pos + (uint64_t)blen * 8 > r.nbits) { err = ERR_FORMAT; break; }
                r.seek(r.pos + (uint64_t)blen * 8);
            }
        }
        s_x[0] = (uint32_t)err; s_x[1] = Kq;
        s_x[2] = (uint32_t)(uintptr_t)qt; s_x[3] = (uint32_t)((uintptr_t)qt >> 32);
        s_x[4] = (uint32_t)(uintptr_t)qtab; s_x[5] = (uint32_t)((uintptr_t)qtab >> 32);
        s_x[6] = (uint32_t)(uintptr_t)work; s_x[7] = (uint32_t)((uintptr_t)work >> 32);
    }
    __syncthreads();
    if (s_x[0]) return (int)s_x[0];
    const uint32_t Kq = s_x[1];
    DTree *qt = (DTree *)(uintptr_t)(((uint64_t)s_x[3] << 32) | s_x[2]);
    uint16_t *qtab = (uint16_t *)(uintptr_t)(((uint64_t)s_x[5] << 32) | s_x[4]);
    uint8_t *work = (uint8_t *)(uintptr_t)(((uint64_t)s_x[7] << 32) | s_x[6]);
    {
        const uint64_t per_tree = (4ull * (nqual + 1) + 7) & ~7ull, per_lane = (2ull * (2 * nqual + 2) + 7) & ~7ull;
        const uint64_t *start = (const uint64_t *)(work + rows * per_tree + 64 * per_lane);
        uint16_t *stack = (uint16_t *)(work + rows * per_tree + (uint64_t)lane * per_lane);
        uint32_t bad = 0;
        for (uint32_t p = lane; p <= max_q; p += 64) {
            BitIn rr = r;
            rr.bad = 0;
            rr.seek(start[p]);
            qt[p].n = 0;
            uint16_t *lc = (uint16_t *)(work + (uint64_t)p * per_tree);
            if (tree_load(rr, nqual, pool, qt[p], stack, lc, lc + nqual + 1)) tree_table(qt[p], qtab + ((size_t)p << Kq), Kq, qsym);
            bad |= rr.bad;
        }
        if (__any(bad != 0)) return ERR_FORMAT;
    }
    __syncthreads();
    if (lane == 0) {
        r.align();
        C.Kq = Kq; C.q_lds = 0u; C.qtab = qtab; C.qt = qt;
        C.pos = r.pos;
    }
    return OK;
}


// ---- title body (cf. FetchTitleBody tasks.cpp:738-954): groups of 32 records, byte aligned -----------------------------------
// Round 2 had lane 0 write every title byte by byte (constant fields copied, numbers printed in decimal: ~7 us per record,
// more than half of a short-read sub-block's decode time).  Here the serial part is only what IS serial: the walk over the
// bit string, which leaves one value word per varying numeric field and record (text fields: offset and length of their
// decoded characters) - a few dozen scalar instructions per record - and the assembly kernel renders the titles from those
// words with all its threads.  Every lane runs the walk on the same values (readfirstlane on what comes from memory), so
// it compiles to scalar code as in dec_symbols_wave; the bit string comes through the same LDS staging.
constexpr uint32_t DEC_TPTR_FIELDS = 4, DEC_TPTR_POS = 32;   // text-field table pointers cached in LDS: fields x positions
constexpr uint32_t DEC_TSLOTS = 24;   // ... and as many of the tables themselves (1 KB each)
constexpr uint32_t DEC_THAM = 64;     // ... and the first bytes of those fields' Hamming flags / first values

__device__ __forceinline__ void dec_title_body_wave(const uint8_t *p, uint64_t nbits, DecCtl &C, const VField *VF, uint32_t nvf, uint32_t nvw,
                                                    const DField *F, uint32_t *vtab, uint8_t *tchars, uint64_t tchar_cap, uint32_t *s_in,
                                                    int32_t *s_prev, uint32_t *s_gv, const uint16_t *const *s_tptr, uint8_t *s_gfl,
                                                    const uint8_t *s_tslot, const uint16_t *s_ttab, const uint8_t *s_ham, const uint8_t *s_d0,
                                                    const uint32_t *s_tslot8)
{
    // s_tslot8: per varying field (the first 64), two words: the LDS table slots of its positions 0..7 (a byte each, 0xff: none)
    // s_tslot / s_ttab / s_ham / s_d0 (round 4): LDS copies of the first text fields' per-position tables, Hamming flags and first
    // values - a character read them from global memory one after the other, two or three round trips each: 4 us per record of a
    // title with a flag and a six-letter barcode, 100 ms per sub-block
    // s_gfl: the group flags of the varying fields from the 65th on (a byte each; the first 64 live in a register pair)
    const uint32_t lane = threadIdx.x;
    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(p);
    const uint32_t ndw = (uint32_t)((nbits + 31) >> 5);
    auto U = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    const uint64_t pos0 = ((uint64_t)U((uint32_t)(C.pos >> 32)) << 32) | U((uint32_t)C.pos);
    const uint32_t n = U(C.n);
    nvf = U(nvf); nvw = U(nvw);
    uint64_t win = 0, tused = 0;
    uint32_t have = 0, wi = (uint32_t)(pos0 >> 5), base = wi, bad = 0;
    auto stage = [&]() {
        base = wi;
        __syncthreads();
        for (uint32_t t = lane; t < DEC_IN_DW + 4; t += 64) s_in[t] = base + t < ndw ? __builtin_bswap32(p32[base + t]) : 0u;
        __syncthreads();
    };
    auto refill = [&]() {
        if (wi - base >= DEC_IN_DW + 2) stage();
        win |= (uint64_t)U(s_in[wi - base]) << (32 - have);
        have += 32;
        ++wi;
    };
    auto get = [&](uint32_t nb) -> uint32_t {   // nb <= 32; more than 32 bits are in the window between calls
        if (!nb) return 0u;
        const uint32_t v = (uint32_t)(win >> (64 - nb));
        win <<= nb;
        have -= nb;
        if (have <= 32) refill();
        return v;
    };
    stage();
    {
        const uint32_t skip0 = (uint32_t)(pos0 & 31);
        win = (uint64_t)U(s_in[0]) << 32 << skip0;
        have = 32 - skip0;
        ++wi;
        refill();
        if (have <= 32) refill();
    }
    // a Huffman symbol through a tree's DEC_K-bit table (codes longer than that: walk the tree in global memory)
    auto sym = [&](const DTree *t, const uint16_t *tab) -> uint32_t {
        if (tab) {
            const uint32_t e = U(tab[(uint32_t)(win >> (64 - DEC_K))]);
            if (e) { (void)get(e >> 12); return e & 0xfffu; }
        }
        const uint32_t tn = U(t->n);
        uint32_t node = U(t->root), steps = 0;
        while (node >= tn) {
            if (++steps > 64) { bad = 1; return 0u; }
            const uint32_t idx = node - tn;
            node = U(get(1) ? t->rc[idx] : t->lc[idx]);
        }
        return node;
    };
    // The first four varying fields - nearly always all of them - live in scalar registers: a field's constants read from LDS
    // one by one, each a round trip that the next branch waits for, made a record cost 0.75 us (50 ms per sub-block).
    // (round 4) The others, and every field's running state, live in the LANES: lane k of a vector register holds field k's value and
    // a v_readlane fetches it in a few cycles where an LDS read costs a lone wave its whole latency.  A text field's first eight
    // positions are there too - which of them hold the first record's character, those characters, the LDS slots of the others'
    // tables - so that a character costs one table read instead of three or four dependent ones (4 us per record of a title with
    // a flag and a barcode: 100 ms per sub-block).
    struct VFc { uint32_t f, kind, vidx, delta, has_g, bits_val, bits_num, bits_len, is_len_const, len0, min_len, max_len; int32_t min_v, min_d;
                 uint32_t ham8, d0a, d0b, sla, slb, fast8; };   // fast8: the five words in front hold the field's first eight positions
    const bool lk = lane < nvf && lane < 64u;
    const VField &vl = VF[lk ? lane : 0];
    const uint32_t vc_f = lk ? vl.f : 0u, vc_kind = lk ? vl.kind : 0u, vc_vidx = lk ? vl.vidx : 0u, vc_delta = lk ? vl.delta : 0u,
                   vc_has_g = lk ? vl.has_g : 0u, vc_bits_val = lk ? vl.bits_val : 0u, vc_bits_num = lk ? vl.bits_num : 0u,
                   vc_bits_len = lk ? vl.bits_len : 0u, vc_ilc = lk ? vl.is_len_const : 0u, vc_len0 = lk ? vl.len0 : 0u,
                   vc_min_len = lk ? vl.min_len : 0u, vc_max_len = lk ? vl.max_len : 0u, vc_min_v = lk ? (uint32_t)vl.min_v : 0u,
                   vc_min_d = lk ? (uint32_t)vl.min_d : 0u;
    uint32_t vc_ham8 = 0, vc_d0a = 0, vc_d0b = 0, vc_sla = 0xffffffffu, vc_slb = 0xffffffffu;
    if (lk && vc_kind == 2) {
        const DField &fd = F[vc_f];
        for (uint32_t j = 0; j < 8 && j < vc_len0; ++j) {
            vc_ham8 |= (uint32_t)(fd.ham[j] != 0) << j;
            if (j < 4) vc_d0a |= (uint32_t)fd.data0[j] << (8 * j); else vc_d0b |= (uint32_t)fd.data0[j] << (8 * (j - 4));
        }
        vc_sla = s_tslot8[2 * lane]; vc_slb = s_tslot8[2 * lane + 1];
    }
    auto RL = [](uint32_t v, uint32_t k) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)k); };
    auto ldc = [&](uint32_t k) {
        VFc c;
        c.f = RL(vc_f, k); c.kind = RL(vc_kind, k); c.vidx = RL(vc_vidx, k); c.delta = RL(vc_delta, k); c.has_g = RL(vc_has_g, k);
        c.bits_val = RL(vc_bits_val, k); c.bits_num = RL(vc_bits_num, k); c.bits_len = RL(vc_bits_len, k); c.is_len_const = RL(vc_ilc, k);
        c.len0 = RL(vc_len0, k); c.min_len = RL(vc_min_len, k); c.max_len = RL(vc_max_len, k); c.min_v = (int32_t)RL(vc_min_v, k);
        c.min_d = (int32_t)RL(vc_min_d, k);
        c.ham8 = RL(vc_ham8, k); c.d0a = RL(vc_d0a, k); c.d0b = RL(vc_d0b, k); c.sla = RL(vc_sla, k); c.slb = RL(vc_slb, k);
        c.fast8 = 1;
        return c;
    };
    auto ldc_lds = [&](uint32_t k) {   // (fields from the 65th on: out of LDS / the scratch pool, as before)
        VFc c;
        const VField &v = VF[k];
        c.f = U(v.f); c.kind = U(v.kind); c.vidx = U(v.vidx); c.delta = U(v.delta); c.has_g = U(v.has_g); c.bits_val = U(v.bits_val);
        c.bits_num = U(v.bits_num); c.bits_len = U(v.bits_len); c.is_len_const = U(v.is_len_const); c.len0 = U(v.len0);
        c.min_len = U(v.min_len); c.max_len = U(v.max_len); c.min_v = (int32_t)U((uint32_t)v.min_v); c.min_d = (int32_t)U((uint32_t)v.min_d);
        c.ham8 = 0; c.d0a = c.d0b = 0; c.sla = c.slb = 0xffffffffu; c.fast8 = 0;
        return c;
    };
    auto ptr64 = [&](const void *q) { const uint64_t a = (uint64_t)(uintptr_t)q; return ((uint64_t)U((uint32_t)(a >> 32)) << 32) | U((uint32_t)a); };
    // one field of one record; prev / gv0 / gv1: the field's running value and its group's first text value
    auto do_field = [&](uint32_t k, const VFc &c, int32_t &prev, uint32_t &gv0, uint32_t &gv1, uint32_t *vrow, bool first, bool gfl) {
        if (c.kind == 1) {
            int32_t v;
            if (first) v = c.min_v + (int32_t)get(c.bits_val);
            else if (gfl) v = c.delta ? prev + c.min_d : prev;
            else {
                int32_t x;
                if (c.has_g) { const DTree *t = &F[c.f].g; x = (int32_t)sym(t, (const uint16_t *)(uintptr_t)ptr64(t->tab)); }
                else x = (int32_t)get(c.bits_num);
                v = c.delta ? prev + c.min_d + x : c.min_v + x;
            }
            prev = v;
            vrow[c.vidx] = (uint32_t)v;
        } else if (!first && gfl) {   // text: the group's first record's characters
            vrow[c.vidx] = gv0;
            vrow[c.vidx + 1] = gv1;
        } else {
            const uint32_t L = c.is_len_const ? c.len0 : c.min_len + get(c.bits_len);
            if (L > c.max_len || tused + L > tchar_cap) { bad = L > c.max_len ? 1u : 3u; return; }
            bool done = false;
            if (c.fast8 && L >= 1 && L <= 8) {
                // (round 4) A short text field - a barcode, a flag with a lane number - in one go, the way dec_symbols_wave takes a trip:
                // lane o stands for bit offset o of the window and looks up, for every position t of the field, the entry that
                // position's code has if it starts at o (a position that holds the first record's character takes no bits); the
                // chain of starts is then one v_readlane per position.  A character used to cost a table read the walk waited for,
                // its bits taken out of the window and a byte store of its own: ~770 cycles each, most of a title's time.
                const uint32_t hi = (uint32_t)(win >> 32), lo = (uint32_t)win;
                const uint32_t top = lane == 0 ? hi : (lane < 32 ? __builtin_amdgcn_alignbit(hi, lo, 32u - lane) : (lo << (lane - 32u)));
                const uint32_t fieldx = top >> (32u - DEC_K);
                uint32_t evv[8];
                bool tabs = true;
#pragma unroll
                for (uint32_t t = 0; t < 8; ++t) {
                    evv[t] = 0;
                    if (t >= L) continue;
                    if (t < c.len0 && ((c.ham8 >> t) & 1u)) { evv[t] = 0x10000u | ((t < 4 ? c.d0a >> (8 * t) : c.d0b >> (8 * (t - 4))) & 0xffu); continue; }
                    const uint32_t slot = (t < 4 ? c.sla >> (8 * t) : c.slb >> (8 * (t - 4))) & 0xffu;
                    if (slot == 0xffu) { tabs = false; continue; }
                    evv[t] = s_ttab[(slot << DEC_K) + fieldx];
                }
                if (tabs) {
                    uint32_t so = 0, cvec = 0, okv = 0;
#pragma unroll
                    for (uint32_t t = 0; t < 8; ++t) {
                        const uint32_t nx = lane + ((evv[t] >> 12) & 15u);
                        const uint32_t e_t = (uint32_t)__builtin_amdgcn_readlane((int)evv[t], (int)(so & 63u));
                        const uint32_t son = (uint32_t)__builtin_amdgcn_readlane((int)nx, (int)(so & 63u));
                        cvec = dec_writelane(cvec, e_t, t);
                        okv = dec_writelane(okv, so, t);
                        so = t < L ? son : so;
                    }
                    // every position had an entry (or the first record's character) and started inside the window's 64 offsets; the
                    // field's bits are all there
                    const bool good = lane >= L || (okv < 64u && ((cvec & 0x10000u) || ((cvec >> 12) & 15u) != 0));
                    if (!__any(!good) && so <= have && so <= 64u) {
                        if (lane < L) tchars[tused + lane] = (uint8_t)cvec;
                        if (so > 32u) { (void)get(32u); (void)get(so - 32u); } else (void)get(so);
                        done = true;
                    }
                }
            }
            const DField &fd = F[c.f];
            const uint8_t *ham = nullptr, *d0 = nullptr;
            const DTree *tl = nullptr;
            if (!done) {
                ham = (const uint8_t *)(uintptr_t)ptr64(fd.ham); d0 = (const uint8_t *)(uintptr_t)ptr64(fd.data0);
                tl = (const DTree *)(uintptr_t)ptr64(fd.l);
            }
            for (uint32_t j = 0; j < L && !bad && !done; ++j) {
                uint32_t ch;
                if (c.fast8 && j < 8) {   // the field's first eight positions: everything but the table entry is in registers
                    if (j < c.len0 && ((c.ham8 >> j) & 1u)) ch = ((j < 4 ? c.d0a >> (8 * j) : c.d0b >> (8 * (j - 4))) & 0xffu);
                    else {
                        const uint32_t slot = (j < 4 ? c.sla >> (8 * j) : c.slb >> (8 * (j - 4))) & 0xffu;
                        if (slot != 0xffu) {
                            const uint32_t e = U(s_ttab[(slot << DEC_K) + (uint32_t)(win >> (64 - DEC_K))]);
                            if (e) { (void)get(e >> 12); ch = e & 0xfffu; }
                            else { const DTree *t = tl + j; ch = sym(t, nullptr); }
                        } else {
                            const DTree *t = tl + j;
                            if (!U(t->n)) { bad = 1; break; }
                            ch = sym(t, (const uint16_t *)(uintptr_t)ptr64(t->tab));
                        }
                    }
                    tchars[tused + j] = (uint8_t)ch;
                    continue;
                }
                const bool cached = k < DEC_TPTR_FIELDS && j < DEC_THAM;
                if (j < c.len0 && U(cached ? s_ham[k * DEC_THAM + j] : ham[j])) ch = U(cached ? s_d0[k * DEC_THAM + j] : d0[j]);
                else {
                    const uint32_t tj = j < MAX_STAT_LEN ? j : MAX_STAT_LEN;
                    const DTree *t = tl + tj;
                    const uint16_t *tab;
                    if (k < DEC_TPTR_FIELDS && tj < DEC_TPTR_POS) {
                        const uint32_t slot = U(s_tslot[k * DEC_TPTR_POS + tj]);
                        if (slot != 0xffu) tab = s_ttab + ((size_t)slot << DEC_K);
                        else {
                            tab = (const uint16_t *)(uintptr_t)ptr64(s_tptr[k * DEC_TPTR_POS + tj]);
                            if (!tab && !U(t->n)) { bad = 1; break; }
                        }
                    } else {
                        if (!U(t->n)) { bad = 1; break; }
                        tab = (const uint16_t *)(uintptr_t)ptr64(t->tab);
                    }
                    ch = sym(t, tab);
                }
                tchars[tused + j] = (uint8_t)ch;
            }
            vrow[c.vidx] = (uint32_t)tused;
            vrow[c.vidx + 1] = L;
            if (first) { gv0 = (uint32_t)tused; gv1 = L; }
            tused += L;
        }
    };
    VFc c0 = {}, c1 = {}, c2 = {}, c3 = {};
    if (nvf > 0) c0 = ldc(0);
    if (nvf > 1) c1 = ldc(1);
    if (nvf > 2) c2 = ldc(2);
    if (nvf > 3) c3 = ldc(3);
    int32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
    uint32_t ga0 = 0, ga1 = 0, gb0 = 0, gb1 = 0, gc0 = 0, gc1 = 0, gd0 = 0, gd1 = 0;
    uint32_t vs_prev = 0, vs_g0 = 0, vs_g1 = 0;
    for (uint32_t g0 = 0; g0 < n && !bad; g0 += GROUP) {
        const uint32_t g1 = g0 + GROUP < n ? g0 + GROUP : n;
        uint64_t gmask = 0;
        for (uint32_t k = 0; k < nvf && k < 64; ++k) gmask |= (uint64_t)get(1) << k;
        for (uint32_t k = 64; k < nvf; ++k) { const uint32_t b = get(1); if (lane == 0) s_gfl[k] = (uint8_t)b; }
        if (nvf > 64) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (uint32_t i = g0; i < g1 && !bad; ++i) {
            uint32_t *vrow = vtab + (size_t)i * nvw;
            const bool first = i == g0;
            if (nvf > 0) do_field(0, c0, p0, ga0, ga1, vrow, first, gmask & 1);
            if (nvf > 1 && !bad) do_field(1, c1, p1, gb0, gb1, vrow, first, (gmask >> 1) & 1);
            if (nvf > 2 && !bad) do_field(2, c2, p2, gc0, gc1, vrow, first, (gmask >> 2) & 1);
            if (nvf > 3 && !bad) do_field(3, c3, p3, gd0, gd1, vrow, first, (gmask >> 3) & 1);
            for (uint32_t k = 4; k < nvf && k < 64 && !bad; ++k) {   // state in lane k of vs_prev / vs_g0 / vs_g1
                const VFc c = ldc(k);
                int32_t pv = (int32_t)RL(vs_prev, k);
                uint32_t x0 = RL(vs_g0, k), x1 = RL(vs_g1, k);
                do_field(k, c, pv, x0, x1, vrow, first, (gmask >> k) & 1);
                vs_prev = lane == k ? (uint32_t)pv : vs_prev;
                vs_g0 = lane == k ? x0 : vs_g0;
                vs_g1 = lane == k ? x1 : vs_g1;
            }
            for (uint32_t k = 64; k < nvf && !bad; ++k) {
                const VFc c = ldc_lds(k);
                int32_t pv = (int32_t)U((uint32_t)s_prev[k]);
                uint32_t x0 = U(s_gv[2 * k]), x1 = U(s_gv[2 * k + 1]);
                do_field(k, c, pv, x0, x1, vrow, first, U(s_gfl[k]) != 0);
                s_prev[k] = pv; s_gv[2 * k] = x0; s_gv[2 * k + 1] = x1;
            }
        }
        // byte-align behind the group (tasks.cpp:508): pos = wi * 32 - have
        const uint32_t d = have & 7u;
        if (d) (void)get(d);
    }
    const uint64_t pos = (uint64_t)wi * 32 - have;
    if (pos > nbits && !bad) bad = 1;
    if (lane == 0) {
        C.pos = pos;
        C.text = tused;
        if (bad) C.err = bad == 3 ? ERR_CAPACITY : ERR_FORMAT;
    }
    __syncthreads();
}

// Narrower copy of a sub-block's per-position tables for LDS: Kl <= Kq bits (an entry whose code is longer than Kl bits
// becomes 0 = "walk the tree").  Whole workgroup; returns Kl, or 0 when not even 6-bit tables fit (use the global ones).
__device__ __forceinline__ uint32_t dec_fill_lds_table(uint16_t *s_qtab, uint32_t lds_entries, const uint16_t *qtab, uint32_t Kq, uint32_t rows)
{
    uint32_t Kl = Kq;
    while (Kl > 5 && ((uint64_t)rows << Kl) > lds_entries) --Kl;
    if (((uint64_t)rows << Kl) > lds_entries) return 0;
    const uint32_t total = rows << Kl, sh = Kq - Kl;
    for (uint32_t k = threadIdx.x; k < total; k += blockDim.x) {
        const uint32_t row = k >> Kl, x = k & ((1u << Kl) - 1u);
        const uint16_t e = qtab[((size_t)row << Kq) + (x << sh)];
        s_qtab[k] = (e >> 12) <= Kl ? e : (uint16_t)0;
    }
    __syncthreads();
    return Kl;
}

// One wave per sub-block: lane 0 walks the headers, the wave the length table and the title body.  Leaves the state of
// the decode in ctl[s]: the quality stream is taken up by the speculative kernels (k_dec_qspec_*) or by the tail kernel.
__global__ __launch_bounds__(64) void k_dec_head(const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb,
                                                 uint8_t *scratch, uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap,
                                                 DecSub *out, DecCtl *ctl, uint32_t allow_spec, uint32_t *summary)
{
    const uint32_t s = blockIdx.x;
    if (s >= n_sb) return;
    const uint32_t lane = threadIdx.x;
    __shared__ DField s_F0[DEC_MAX_FIELDS];   // (a lone lane pays a full memory latency for every field it touches in global memory)
    __shared__ VField s_VF0[DEC_MAX_FIELDS];
    __shared__ int32_t s_prev0[DEC_MAX_FIELDS];
    __shared__ uint32_t s_gv0[2 * DEC_MAX_FIELDS];
    __shared__ uint8_t s_gfl[WIDE_FIELDS];
    // (a title of more than DEC_MAX_FIELDS fields - up to WIDE_FIELDS, as the encoder takes them - has these four in the scratch pool)
    __shared__ DField *s_Fp;
    __shared__ VField *s_VFp;
    __shared__ int32_t *s_prevp;
    __shared__ uint32_t *s_gvp;
    __shared__ const uint16_t *s_tptr[DEC_TPTR_FIELDS * DEC_TPTR_POS];
    __shared__ uint8_t s_tslot[DEC_TPTR_FIELDS * DEC_TPTR_POS];
    __shared__ uint32_t s_tslot8[128], s_nslot;
    __shared__ const uint16_t *s_tsrc[DEC_TSLOTS];
    __shared__ __align__(16) uint16_t s_ttab[DEC_TSLOTS << DEC_K];
    __shared__ uint8_t s_ham[DEC_TPTR_FIELDS * DEC_THAM], s_d0[DEC_TPTR_FIELDS * DEC_THAM];
    __shared__ uint32_t s_in[DEC_IN_DW + 4];
    __shared__ DecCtl C;
    __shared__ uint32_t s_nv[2];
    DecSub &O = out[s];
    BitIn r;
    r.init(sub + sb_off[s], sb_len[s]);
    DecScratch D = dec_scratch(scratch, stride, s, rec_cap, tcap, qcap);
    Pool pool = {D.pool, 0, D.pool_cap};
    uint16_t *stack = nullptr;
    const uint64_t t0 = wall_clock64();
    if (lane == 0) {
        O.err = OK; O.text_bytes = 0; O.n_rec = 0;
        ctl[s].err = ERR_FORMAT;   // (overwritten with the whole state when the head has gone through)
        C.spec = 0; C.spec_bad = 0; C.dna_spec = 0; C.dna_bad = 0;
        stack = (uint16_t *)pool.take(2 * (2 * 512 + 2));
        C.err = dec_info(r, D, O, rec_cap, C);
        if (C.err) O.err = C.err;
    }
    __syncthreads();
    if (C.err) return;
    const uint32_t n = C.n;
    // ---- the length table (phyNGSC.cpp:730-742): n entries of lt_bits bits, every lane its own entries ----
    {
        const uint32_t b = C.lt_bits, mq = C.max_q;
        const uint64_t lt0 = C.pos, nbytes = sb_len[s];
        uint64_t tot = 0;
        uint32_t badl = 0, uneq = 0;
        for (uint32_t i = lane; i < n; i += 64) {
            const uint32_t L = dec_bits_at(r.p, nbytes, lt0 + (uint64_t)i * b, b);
            D.qlen[i] = L;
            tot += L;
            badl |= L > mq;
            uneq |= L != mq;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            tot += __shfl_xor(tot, d); badl |= (uint32_t)__shfl_xor((int)badl, d); uneq |= (uint32_t)__shfl_xor((int)uneq, d);
        }
        if (lane == 0) {
            // every read has max_q qualities (and a record's codes fit the 16-bit hop of the speculative decoder)
            C.spec = (allow_spec && !uneq && mq >= 1 && mq <= 2000) ? 1u : 0u;
            C.qual_total = tot;
            if (badl) C.err = ERR_FORMAT;
            else if (2 * tot > D.tcap) C.err = ERR_CAPACITY;
            if (C.err) O.err = C.err;
        }
    }
    __syncthreads();
    if (C.err) return;
    const uint64_t tp1 = wall_clock64();
    // ---- title header (lane 0), then the tables the body walk and the renderer work from ----
    if (lane == 0) {
        r.seek(C.pos + (uint64_t)n * C.lt_bits);
        r.align();
        {
            const uint64_t at = r.pos;
            const uint32_t nf_peek = r.u32();
            r.seek(at);
            s_Fp = s_F0; s_VFp = s_VF0; s_prevp = s_prev0; s_gvp = s_gv0;
            if (!r.bad && nf_peek > DEC_MAX_FIELDS && nf_peek <= WIDE_FIELDS) {
                s_Fp = (DField *)pool.take(sizeof(DField) * nf_peek);
                s_VFp = (VField *)pool.take(sizeof(VField) * nf_peek);
                s_prevp = (int32_t *)pool.take(4 * nf_peek);
                s_gvp = (uint32_t *)pool.take(8 * nf_peek);
            }
        }
        DField *const s_F = s_Fp;
        VField *const s_VF = s_VFp;
        C.err = (s_Fp && s_VFp && s_prevp && s_gvp) ? dec_head(r, D, pool, O, s_F, C, stack) : (int)ERR_NOMEM;
        if (!C.err) {
            const uint32_t nf = C.nf;
            AField *af = (AField *)pool.take(sizeof(AField) * nf);
            uint32_t nvf = 0, nvw = 0;
            if (!af) C.err = ERR_NOMEM;
            for (uint32_t k = 0; k < DEC_TPTR_FIELDS * DEC_TPTR_POS; ++k) s_tptr[k] = nullptr;
            for (uint32_t f = 0; f < nf && !C.err; ++f) {
                const DField &fd = s_F[f];
                AField &a = af[f];
                a.sep = fd.sep; a.len0 = fd.len0; a.vidx = 0; a.data_off = 0;
                if (fd.is_const) { a.kind = 0; a.data_off = (uint64_t)(fd.data0 - D.pool); continue; }
                VField &v = s_VF[nvf];
                v.f = f; v.kind = fd.is_num ? 1u : 2u; v.vidx = nvw; v.delta = fd.delta_coding; v.has_g = fd.has_g;
                v.bits_val = fd.bits_val > 32 ? 32 : fd.bits_val; v.bits_num = fd.bits_num > 32 ? 32 : fd.bits_num;
                v.bits_len = fd.bits_len > 32 ? 32 : fd.bits_len; v.is_len_const = fd.is_len_const; v.len0 = fd.len0;
                v.min_len = fd.min_len; v.max_len = fd.max_len; v.min_v = fd.min_v; v.min_d = fd.min_d;
                a.kind = v.kind; a.vidx = nvw;
                if (!fd.is_num && nvf < DEC_TPTR_FIELDS)
                    for (uint32_t j = 0; j < DEC_TPTR_POS && j <= MAX_STAT_LEN; ++j) s_tptr[nvf * DEC_TPTR_POS + j] = fd.l[j].n ? fd.l[j].tab : nullptr;
                nvw += fd.is_num ? 1u : 2u;
                ++nvf;
            }
            s_nv[0] = nvf; s_nv[1] = nvw;
            O.nf = nf; O.nvw = nvw; O.af_off = af ? (uint64_t)((uint8_t *)af - D.pool) : 0;
            // text area: [value table n x nvw words][characters of the text fields][qualities][bases of a Huffman-coded alphabet]
            const uint64_t vbytes = (4ull * n * nvw + 15) & ~15ull;
            if (!C.err && vbytes + 2 * C.qual_total > D.tcap) C.err = ERR_CAPACITY;
            O.tchar_off = vbytes;
        }
        if (C.err) O.err = C.err;
    }
    __syncthreads();
    if (C.err) return;
    {   // the text fields' tables (positions 0..7 of every field first, then positions 8..31 of the first four), Hamming flags and
        // first values into LDS (all lanes)
        if (lane == 0) {
            uint32_t ns = 0;
            const uint32_t nvf0 = s_nv[0];
            for (uint32_t k = 0; k < 128; ++k) s_tslot8[k] = 0xffffffffu;
            for (uint32_t k = 0; k < DEC_TPTR_FIELDS * DEC_TPTR_POS; ++k) s_tslot[k] = 0xff;
            for (uint32_t k = 0; k < nvf0 && k < 64; ++k) {
                const VField &v = s_VFp[k];
                if (v.kind != 2) continue;
                const DField &fd = s_Fp[v.f];
                for (uint32_t j = 0; j < 8 && j <= MAX_STAT_LEN && ns < DEC_TSLOTS; ++j) {
                    if (!fd.l[j].n || !fd.l[j].tab) continue;
                    s_tsrc[ns] = fd.l[j].tab;
                    s_tslot8[2 * k + (j >> 2)] = (s_tslot8[2 * k + (j >> 2)] & ~(0xffu << (8 * (j & 3)))) | (ns << (8 * (j & 3)));
                    if (k < DEC_TPTR_FIELDS) s_tslot[k * DEC_TPTR_POS + j] = (uint8_t)ns;
                    ++ns;
                }
            }
            for (uint32_t k = 0; k < DEC_TPTR_FIELDS; ++k)
                for (uint32_t j = 8; j < DEC_TPTR_POS && ns < DEC_TSLOTS; ++j)
                    if (s_tptr[k * DEC_TPTR_POS + j]) { s_tsrc[ns] = s_tptr[k * DEC_TPTR_POS + j]; s_tslot[k * DEC_TPTR_POS + j] = (uint8_t)ns; ++ns; }
            s_nslot = ns;
        }
        __syncthreads();
        for (uint32_t slot = 0; slot < s_nslot; ++slot) {
            const uint16_t *src = s_tsrc[slot];
            for (uint32_t x = lane; x < (1u << DEC_K); x += 64) s_ttab[(slot << DEC_K) + x] = src[x];
        }
        const uint32_t nvf = s_nv[0];
        for (uint32_t k = 0; k < DEC_TPTR_FIELDS && k < nvf; ++k) {
            const VField &v = s_VFp[k];
            uint8_t hm = 0, dv = 0;
            if (v.kind == 2 && lane < v.len0 && lane < DEC_THAM) { const DField &fd = s_Fp[v.f]; hm = fd.ham[lane]; dv = fd.data0[lane]; }
            if (lane < DEC_THAM) { s_ham[k * DEC_THAM + lane] = hm; s_d0[k * DEC_THAM + lane] = dv; }
        }
        __syncthreads();
    }
    const uint64_t tp2 = wall_clock64();
    {
        const uint64_t vbytes = O.tchar_off;
        const uint64_t room = D.tcap - vbytes - 2 * C.qual_total;   // what is left for the text fields' characters (32 bytes kept for alignment)
        dec_title_body_wave(r.p, r.nbits, C, s_VFp, s_nv[0], s_nv[1], s_Fp, (uint32_t *)D.titles, D.titles + vbytes, room > 64 ? room - 64 : 0, s_in,
                            s_prevp, s_gvp, s_tptr, s_gfl, s_tslot, s_ttab, s_ham, s_d0, s_tslot8);
    }
    if (C.err) { if (lane == 0) O.err = C.err; return; }
    const uint64_t tp3 = wall_clock64();
    // ---- title length of every record (its place in the text needs it): constant fields and separators + the digits of its
    // numbers + its text fields' characters; every lane its share of the records ----
    {
        const uint32_t nf = O.nf, nvw = O.nvw;
        const AField *af = (const AField *)(D.pool + O.af_off);
        uint32_t fixed = nf;   // one separator per field
        for (uint32_t f = 0; f < nf; ++f) fixed += af[f].kind == 0 ? af[f].len0 : 0u;
        const uint32_t *vtab = (const uint32_t *)D.titles;
        uint64_t tot = 0;
        for (uint32_t i = lane; i < n; i += 64) {
            uint32_t tl = fixed;
            const uint32_t *vrow = vtab + (size_t)i * nvw;
            for (uint32_t f = 0; f < nf; ++f) {
                const uint32_t kind = af[f].kind;
                if (kind == 1) tl += dec_ndigits(vrow[af[f].vidx]);
                else if (kind == 2) tl += vrow[af[f].vidx + 1];
            }
            D.toff[i] = tl;
            tot += tl;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) tot += __shfl_xor(tot, d);
        if (lane == 0) O.text_bytes = tot + 2ull * C.qual_total + 4ull * n;
    }
    const uint64_t tp4 = wall_clock64();
    // ---- quality alphabet and trees (lane 0), then the state for the kernels behind this one ----
    if (lane == 0) {
        O.t_ph[0] = tp1 - t0; O.t_ph[1] = tp2 - tp1; O.t_ph[2] = tp3 - tp2; O.t_ph[3] = tp4 - tp3;
        C.text += O.tchar_off;   // the value table and the characters
        r.seek(C.pos);
    }
    __syncthreads();
    {
        __shared__ uint32_t s_x[8];
        const int qerr = dec_head_quality(r, D, pool, C.qsym, C, s_x);
        __syncthreads();
        if (lane == 0 && qerr) C.err = qerr;
    }
    if (lane == 0) {
        if (C.err) O.err = C.err;
        C.pool_off = pool.off;
        C.t_head = wall_clock64() - t0;
        O.t_ph[4] = wall_clock64() - tp4;
        if (!C.err) {
            // first phase of the speculative decoder: with two-bit bases the quality stream ends at or behind
            // (end of the sub-block - alphabet - 2 bits per base of all reads); else nothing is known about its end
            const uint64_t span = r.nbits - C.pos;
            const uint64_t dna_max = 8ull * O.nsym + 2ull * C.n * C.max_q;
            C.cov_lo = 0; C.chain_i = 0; C.chain_cur = 0;
            C.cov_hi = (O.plain && dna_max + 64 < span) ? span - dna_max + 64 : span;
            if (r.nbits >= (1ull << 31)) C.spec = 0;   // (the chain keeps 32-bit offsets)
        }
    }
    __syncthreads();
    if (C.err) return;
    for (uint32_t k = lane; k < sizeof(DecCtl) / 4; k += 64) reinterpret_cast<uint32_t *>(&ctl[s])[k] = reinterpret_cast<const uint32_t *>(&C)[k];
    if (lane == 0 && C.spec) { atomicMax(&summary[0], C.max_q); atomicMax(&summary[1], C.nqual); }   // the longest read / largest alphabet the speculative kernels will meet: their LDS
}

// ---- the quality stream of equally long reads, decoded speculatively ------------------------------------------------------------
// A sub-block's quality stream is one bit string of per-position Huffman codes: where a code starts depends on every code in
// front of it, so one wave per sub-block decoded it serially (round 2: 115 ns per symbol, 0.3 s per 8 MiB window - the whole
// decoder's speed at file sizes that do not fill the chip with sub-blocks).  When every read has the same number L of
// qualities (the length table says so) the stream is a chain of RECORDS, each L codes from the trees 1..L, and the only
// unknown of a record is the bit it starts at.  So:
//   A  every bit offset b of the stream gets hop[b] = the bits a record would take if it started at b - L table lookups per
//      offset, ~60 times more decoding than the stream holds, all of it independent (a lane per offset, the tables in LDS):
//      milliseconds for the whole job, where the serial walk took hundreds;
//   B  one wave per sub-block follows the chain of true record starts through hop[] (staged through LDS in order: a hop
//      costs an LDS read, not a memory round trip) and writes them down;
//   C  a lane per record decodes its L symbols from its start bit.
// Anything the lookups cannot settle (a code longer than the LDS tables hold walks its tree; a record longer than the
// 16-bit hop, a chain that leaves the stream) marks the sub-block spec_bad and the tail kernel decodes it serially.
constexpr uint32_t QS_THREADS = 512;
constexpr uint32_t QS_PER_THREAD = 16;                     // candidate offsets per thread and workgroup
constexpr uint32_t QS_TILE = QS_THREADS * QS_PER_THREAD;   // offsets per workgroup
constexpr uint32_t QS_INVALID = 0xFFFFu;

// symbol lookups of one record from bit offset b of the stream p (global memory; bits behind the end read as zeros).
// EMIT: store the symbols (bytes) to dst and count those below 128.  Returns the record's bits, or ~0u when it cannot tell.
template <bool EMIT>
__device__ __forceinline__ uint32_t qs_record(const uint32_t *p32, uint32_t ndw, uint64_t b, uint32_t L, const uint16_t *s_tab, uint32_t Kl,
                                              const DTree *qt, const uint8_t *qsym, uint8_t *dst, uint32_t *kept)
{
    uint32_t d = (uint32_t)(b >> 5);
    const uint32_t sh = (uint32_t)(b & 31);
    auto dw = [&](uint32_t k) -> uint32_t { return k < ndw ? __builtin_bswap32(p32[k]) : 0u; };
    // the window as two dwords (hi: the next 32 bits of the stream, lo: those behind them; `have` bits are real, the rest
    // zero): the codes the tables hold are at most 9 bits long, so a symbol costs 32-bit operations only - one funnel shift
    // moves the window (64-bit shifts run at a fraction of the rate)
    const uint64_t w0 = (((uint64_t)dw(d) << 32) | dw(d + 1)) << sh;
    uint32_t hi = (uint32_t)(w0 >> 32), lo = (uint32_t)w0, have = 64 - sh, total = 0, kc = 0;
    d += 2;
    const uint32_t ksh = 32 - Kl;
    for (uint32_t j = 0; j < L; ++j) {
        const uint32_t e = s_tab[((j + 1) << Kl) + (hi >> ksh)];
        uint32_t len, symb;
        if (e) { len = e >> 12; symb = e & 0xffu; }
        else {   // a code longer than the LDS table's width (or a one-symbol tree): walk the tree (global memory)
            const DTree &T = qt[j + 1];
            uint32_t node = T.root;
            uint64_t w = ((uint64_t)hi << 32) | lo;
            uint32_t hv = have, dd = d;
            len = 0;
            while (node >= T.n) {
                if (len >= 32 || hv == 0) return ~0u;
                node = (w >> 63) ? T.rc[node - T.n] : T.lc[node - T.n];
                w <<= 1; --hv; ++len;
                if (hv < 32) { w |= (uint64_t)dw(dd) << (32 - hv); hv += 32; ++dd; }
            }
            symb = qsym[node & 0xffu];
        }
        if (EMIT) { dst[j] = (uint8_t)symb; kc += symb < 128u; }
        if (len) {
            hi = len < 32 ? __builtin_amdgcn_alignbit(hi, lo, 32 - len) : lo;
            lo = len < 32 ? lo << len : 0u;
        }
        have -= len;
        total += len;
        while (have <= 32) {   // the real bits are all in hi now
            const uint32_t x = dw(d++);
            hi |= have < 32 ? x >> have : 0u;
            lo = have ? x << (32 - have) : 0u;
            have += 32;
        }
    }
    if (EMIT) *kept = kc;
    return total;
}

// (round 3) Kernel A only needs code LENGTHS: a table of bytes (K bits of look-ahead per row, the code's length, 0 = longer than K)
// takes half the LDS of the entries with symbols - three workgroups per CU instead of two at 150 bases - and leaves room for
// the trees themselves (child tables as bytes: alphabets of up to 128 symbols), so that a code longer than K bits is walked in
// LDS, not in global memory: with K = 8 at 150 bases one lane in four hundred steps meets such a code, i.e. one wave in six at
// every step, and the whole wave waited ~10 dependent memory round trips for it (0.15 s of the 0.6 s a 400 MB file of 150-base
// reads took).  Returns the record's bits from bit offset b, or ~0u.
struct QsLenTabs {
    const uint8_t *len;      // [rows << K]
    const uint8_t *lc, *rc;  // [rows][nq] children of internal node nq + i of the row's tree (byte node ids), or nullptr
    const uint8_t *root;     // [rows]
    uint32_t K, nq;
};
__device__ __forceinline__ uint32_t qs_record_len(const uint32_t *p32, uint32_t ndw, uint64_t b, uint32_t L, const QsLenTabs &T, const DTree *qt)
{
    uint32_t d = (uint32_t)(b >> 5);
    const uint32_t sh = (uint32_t)(b & 31);
    auto dw = [&](uint32_t k) -> uint32_t { return k < ndw ? __builtin_bswap32(p32[k]) : 0u; };
    const uint64_t w0 = (((uint64_t)dw(d) << 32) | dw(d + 1)) << sh;
    uint32_t hi = (uint32_t)(w0 >> 32), lo = (uint32_t)w0, have = 64 - sh, total = 0;
    d += 2;
    const uint32_t ksh = 32 - T.K;
    for (uint32_t j = 0; j < L; ++j) {
        // an entry: the code's length in its low four bits and - where the NEXT position's code also lies inside the K bits - that
        // one's length in the high four: two symbols per dependent lookup (the lookups, not the arithmetic, are this loop's time)
        const uint32_t ent = T.len[((j + 1) << T.K) + (hi >> ksh)];
        uint32_t len = ent & 15u;
        if (ent >> 4) { len += ent >> 4; ++j; }
        if (!len) {   // a code longer than K bits (or a one-symbol tree, whose codes have no bits): walk the tree
            uint64_t w = ((uint64_t)hi << 32) | lo;
            uint32_t hv = have, dd = d;
            if (T.lc) {
                const uint32_t nq = T.nq, rowo = (j + 1) * nq;
                uint32_t node = T.root[j + 1];
                while (node >= nq) {
                    if (len >= 32 || hv == 0) return ~0u;
                    node = (w >> 63) ? T.rc[rowo + node - nq] : T.lc[rowo + node - nq];
                    w <<= 1; --hv; ++len;
                    if (hv < 32) { w |= (uint64_t)dw(dd) << (32 - hv); hv += 32; ++dd; }
                }
            } else {
                const DTree &Q = qt[j + 1];
                uint32_t node = Q.root;
                while (node >= Q.n) {
                    if (len >= 32 || hv == 0) return ~0u;
                    node = (w >> 63) ? Q.rc[node - Q.n] : Q.lc[node - Q.n];
                    w <<= 1; --hv; ++len;
                    if (hv < 32) { w |= (uint64_t)dw(dd) << (32 - hv); hv += 32; ++dd; }
                }
            }
        }
        if (len) {
            hi = len < 32 ? __builtin_amdgcn_alignbit(hi, lo, 32 - len) : lo;
            lo = len < 32 ? lo << len : 0u;
        }
        have -= len;
        total += len;
        while (have <= 32) {
            const uint32_t x = dw(d++);
            hi |= have < 32 ? x >> have : 0u;
            lo = have ? x << (32 - have) : 0u;
            have += 32;
        }
    }
    return total;
}

// A: hop[b - q0] for the bit offsets q0 + [cov_lo, cov_hi) of a sub-block.  The stream's end is not known before it is
// decoded; with two-bit bases it is at least (sub-block end - what the bases can take at most), which is what the first
// phase covers; the chain kernel asks for more (records with transferred bases make the quality stream longer and the
// base stream shorter) and later phases deliver it.
__global__ __launch_bounds__(QS_THREADS) void k_dec_qspec_a(const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, const DecCtl *ctl,
                                                            uint16_t *hop, const uint64_t *hop_off, uint32_t lds_entries, uint32_t with_trees)
{
    const uint32_t s = blockIdx.y;
    const DecCtl &C = ctl[s];
    if (C.err || C.spec != 1 || C.spec_bad) return;
    const uint64_t nbits = (uint64_t)sb_len[s] * 8, q0 = C.pos;
    const uint64_t first = C.cov_lo + (uint64_t)blockIdx.x * QS_TILE;   // (relative to q0)
    if (first >= C.cov_hi) return;
    extern __shared__ __align__(16) uint8_t s_len[];   // [rows << Kl] code lengths, then (alphabets of up to 128 symbols) the trees
    const uint32_t L = C.max_q, rows = L + 1, Kq = C.Kq, nq = C.nqual;
    uint32_t Kl = Kq;
    while (Kl > 5 && ((uint64_t)rows << Kl) > lds_entries) --Kl;
    if (((uint64_t)rows << Kl) > lds_entries) return;   // (the chain kernel finds the hops missing: hop[] is cleared to "invalid" beforehand)
    {
        const uint32_t total = rows << Kl, shq = Kq - Kl;
        const uint16_t *qtab = C.qtab;
        for (uint32_t k = threadIdx.x; k < total; k += QS_THREADS) {
            const uint32_t row = k >> Kl, x = k & ((1u << Kl) - 1u);
            uint32_t ln = qtab[((size_t)row << Kq) + (x << shq)] >> 12;
            if (ln > Kl) ln = 0;
            uint32_t ln2 = 0;
            if (ln && ln < Kl && row >= 1 && row + 1 < rows) {   // (row 0 is the position-independent tree: no record position)
                // the bits behind the first code, zero-padded: the next position's length is what the table says iff it fits them
                const uint32_t x2 = (x << ln) & ((1u << Kl) - 1u);
                ln2 = qtab[((size_t)(row + 1) << Kq) + (x2 << shq)] >> 12;
                if (ln2 > Kl - ln) ln2 = 0;
            }
            s_len[k] = (uint8_t)(ln | (ln2 << 4));
        }
    }
    QsLenTabs T = {s_len, nullptr, nullptr, nullptr, Kl, nq};
    if ((with_trees & 1u) && nq >= 2 && nq <= 128) {
        uint8_t *lc = s_len + (((size_t)rows << Kl) + 15 & ~(size_t)15), *rc = lc + (size_t)rows * nq, *root = rc + (size_t)rows * nq;
        for (uint32_t k = threadIdx.x; k < rows * nq; k += QS_THREADS) {
            const uint32_t row = k / nq, i = k % nq;
            const DTree &Q = C.qt[row];
            const bool in = Q.n == nq && i + 1 < nq;   // internal nodes nq .. 2 nq - 2
            lc[k] = in ? (uint8_t)Q.lc[i] : (uint8_t)0;
            rc[k] = in ? (uint8_t)Q.rc[i] : (uint8_t)0;
        }
        for (uint32_t k = threadIdx.x; k < rows; k += QS_THREADS) root[k] = (uint8_t)C.qt[k].root;
        T.lc = lc; T.rc = rc; T.root = root;
    }
    __syncthreads();
    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(sub + sb_off[s]);
    const uint32_t ndw = (uint32_t)((nbits + 31) >> 5);
    uint16_t *h = hop + hop_off[s];
    const uint64_t cov_hi = C.cov_hi;
    // (round 4, measured and not kept: two offsets per thread walked in lock-step, so that a lane keeps two table lookups in flight -
    // 0.31 s against 0.21 per GB of 150-base reads: the second chain's registers and the lock-step's idle halves cost more than the
    // overlap gains; profiles/r04_decode_two_chains_not_kept.txt)
#pragma unroll 1
    for (uint32_t o = 0; o < QS_PER_THREAD; ++o) {
        const uint64_t rel = first + threadIdx.x + (uint64_t)o * QS_THREADS;
        if (rel >= cov_hi) break;
        const uint64_t b = q0 + rel;
        const uint32_t t = qs_record_len(p32, ndw, b, L, T, C.qt);
        h[rel] = (t == ~0u || t >= QS_INVALID || b + t > nbits) ? (uint16_t)QS_INVALID : (uint16_t)t;
    }
}

// B: the chain of record starts.  One wave per sub-block; hop[] comes through LDS in the order the chain reads it.
// next_tiles: tiles the next phase of A will cover if the chain runs out of hops (0: there is no next phase).
constexpr uint32_t QS_CHUNK = 8192;   // hops staged at a time (16 KB)
__global__ __launch_bounds__(64) void k_dec_qspec_b(const uint32_t *sb_len, DecCtl *ctl, const uint16_t *hop, const uint64_t *hop_off,
                                                    uint8_t *scratch, uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint32_t next_tiles)
{
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    DecCtl &C = ctl[s];
    if (C.err || C.spec != 1 || C.spec_bad) return;
    __shared__ __align__(16) uint16_t s_h[QS_CHUNK];
    DecScratch D = dec_scratch(scratch, stride, s, rec_cap, tcap, 0);
    const uint64_t nbits = (uint64_t)sb_len[s] * 8, q0 = C.pos, span = nbits - q0;
    const uint16_t *h = hop + hop_off[s];
    auto U = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    const uint32_t n = U(C.n);
    const uint64_t cov_hi = ((uint64_t)U((uint32_t)(C.cov_hi >> 32)) << 32) | U((uint32_t)C.cov_hi);
    uint64_t cur = ((uint64_t)U((uint32_t)(C.chain_cur >> 32)) << 32) | U((uint32_t)C.chain_cur), base = 0;   // relative to q0
    uint32_t i = U(C.chain_i), bad = 0, more = 0;
    bool staged = false;
    for (; i < n; ++i) {
        if (cur >= cov_hi) { more = 1; break; }   // hops not computed yet
        if (!staged || cur - base >= QS_CHUNK) {
            base = cur & ~7ull;   // 16-byte aligned loads
            __syncthreads();
            for (uint32_t t = lane; t < QS_CHUNK / 8; t += 64) {
                uint4 v = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                if (base + 8ull * t + 8 <= span) v = *reinterpret_cast<const uint4 *>(h + base + 8ull * t);
                else for (uint32_t k = 0; k < 8; ++k) if (base + 8ull * t + k < span) reinterpret_cast<uint16_t *>(&v)[k] = h[base + 8ull * t + k];
                reinterpret_cast<uint4 *>(s_h)[t] = v;
            }
            __syncthreads();
            staged = true;
        }
        D.rs[i] = (uint32_t)cur;
        const uint32_t hp = U((uint32_t)s_h[cur - base]);
        if (hp == QS_INVALID) { bad = 1; break; }
        cur += hp;
    }
    if (lane == 0) {
        if (bad || q0 + cur > nbits || (more && (!next_tiles || cov_hi >= span))) C.spec_bad = 1;
        else if (more) {
            C.chain_i = i; C.chain_cur = cur;
            C.cov_lo = cov_hi;
            const uint64_t want = cov_hi + (uint64_t)next_tiles * QS_TILE;
            C.cov_hi = want < span ? want : span;
        } else { D.rs[n] = (uint32_t)cur; C.pos = q0 + cur; C.spec = 2; }   // 2: all starts known, C takes over
    }
}

// C: a lane per record
__global__ __launch_bounds__(QS_THREADS) void k_dec_qspec_c(const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, DecCtl *ctl,
                                                            uint8_t *scratch, uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint32_t lds_entries)
{
    const uint32_t s = blockIdx.y;
    DecCtl &C = ctl[s];
    if (C.err || C.spec != 2) return;
    const uint32_t n = C.n, L = C.max_q;
    if (blockIdx.x * QS_THREADS >= n) return;
    extern __shared__ __align__(16) uint16_t s_qtab[];
    const uint32_t Kl = dec_fill_lds_table(s_qtab, lds_entries, C.qtab, C.Kq, L + 1);
    DecScratch D = dec_scratch(scratch, stride, s, rec_cap, tcap, 0);
    if (!Kl) { if (threadIdx.x == 0) atomicOr(&C.spec_bad, 1u); return; }
    const uint32_t i = blockIdx.x * QS_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint64_t nbits = (uint64_t)sb_len[s] * 8;
    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(sub + sb_off[s]);
    // (C.pos is the stream's END by now: its start is end - rs[n])
    const uint64_t q0 = C.pos - D.rs[n];
    uint32_t kept = 0;
    const uint32_t t = qs_record<true>(p32, (uint32_t)((nbits + 31) >> 5), q0 + D.rs[i], L, s_qtab, Kl, C.qt, C.qsym, D.titles + D.lay[0] + (uint64_t)i * L, &kept);
    D.slen[i] = kept;
    if (t != D.rs[i + 1] - D.rs[i]) atomicOr(&C.spec_bad, 1u);   // (cannot happen: A and C run the same lookups)
}

// One wave per sub-block, behind the head kernel (and the speculative quality kernels): the quality stream if it is still
// to be decoded, then the bases.
// mode 0: everything.  mode 1 (the parallel base decoder is available): everything but the code stream of a Huffman-coded base
// alphabet whose codes all fit the table - the kernels k_dec_dna_a/b/c take that; mode 2: behind them - the sub-blocks they
// decoded are closed, those they gave up on are decoded here after all.
__global__ __launch_bounds__(64) void k_dec_tail(const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb,
                                                 uint8_t *scratch, uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap,
                                                 DecSub *out, DecCtl *ctl, uint32_t lds_entries, uint32_t mode)
{
    const uint32_t s = blockIdx.x;
    if (s >= n_sb) return;
    const uint32_t lane = threadIdx.x;
    __shared__ uint8_t s_map[256];
    __shared__ uint32_t s_in[DEC_IN_DW + 4];
    __shared__ uint8_t s_out[DEC_OUT];
    __shared__ DecCtl C;
    __shared__ uint16_t s_dtab[1u << DEC_K];
    extern __shared__ __align__(16) uint16_t s_qtab[];
    DecSub &O = out[s];
    if (O.err) return;
    for (uint32_t k = lane; k < sizeof(DecCtl) / 4; k += 64) reinterpret_cast<uint32_t *>(&C)[k] = reinterpret_cast<const uint32_t *>(&ctl[s])[k];
    __syncthreads();
    if (C.err) { if (lane == 0) O.err = C.err; return; }
    BitIn r;
    r.init(sub + sb_off[s], sb_len[s]);
    DecScratch D = dec_scratch(scratch, stride, s, rec_cap, tcap, qcap);
    Pool pool = {D.pool, C.pool_off, D.pool_cap};
    uint16_t *stack = nullptr;
    if (lane == 0) stack = (uint16_t *)pool.take(2 * (2 * 512 + 2));
    const uint32_t n = C.n;
    const uint64_t t1 = wall_clock64();
    uint8_t *d_qual = D.titles + D.lay[0], *d_seq = D.titles + D.lay[1];
    __shared__ DTree s_dt;
    if (mode == 2) {
        if (C.dna_spec == 0) return;   // (closed by the first call)
        const bool good = C.dna_spec == 2 && !C.dna_bad;
        if (!good) {   // the code stream serially, from what the first call left
            for (uint32_t k = lane; k < (1u << DEC_K); k += 64) s_dtab[k] = C.dna_tab[k];
            if (lane == 0) {
                s_dt.n = C.dna_n; s_dt.root = C.dna_root; s_dt.lc = C.dna_lc; s_dt.rc = C.dna_rc; s_dt.tab = s_dtab;
                C.pos = C.dna_p0;
            }
            for (uint32_t k = lane; k < 256; k += 64) s_map[k] = k < O.nsym ? O.dsym[k] : (uint8_t)0;
            __syncthreads();
            dec_symbols_wave<false>(r.p, r.nbits, C, s_dtab, DEC_K, &s_dt, D.slen, n, d_seq, nullptr, s_map, s_in, s_out, 1);
            __syncthreads();
            if (C.err) { if (lane == 0) O.err = C.err; return; }
        }
        if (lane == 0) {
            O.dna_bitpos = C.pos;
            O.t_head = C.t_head; O.t_qual = C.t_qual; O.t_dna = wall_clock64() - t1;
            O.spec = (C.spec == 2 && !C.spec_bad) ? 1u : 0u;
        }
        return;
    }
    if (!(C.spec == 2 && !C.spec_bad)) {   // the quality stream, serially
        s_map[lane] = C.qsym[lane]; s_map[lane + 64] = C.qsym[lane + 64]; s_map[lane + 128] = C.qsym[lane + 128]; s_map[lane + 192] = C.qsym[lane + 192];
        if (C.spec == 2 && lane == 0) C.pos -= D.rs[n];   // (the chain had moved it to the stream's end)
        __syncthreads();
        // LDS: the per-position tables, as many look-ahead bits as fit, and - for alphabets of up to 128 symbols, when they leave the
        // tables at least half of the room - the rows' trees as byte child tables, for the codes the tables do not hold
        const uint32_t rows = C.max_q + 1, nq = C.nqual;
        uint32_t tree_bytes = (nq >= 1 && nq <= 128) ? ((rows * (2 * nq + 1) + 15u) & ~15u) : 0u;
        if (tree_bytes > lds_entries) tree_bytes = 0;   // (lds_entries 16-bit entries = 2 lds_entries bytes)
        const uint32_t tab_entries = lds_entries - tree_bytes / 2;
        uint8_t *t_lc = nullptr, *t_rc = nullptr, *t_root = nullptr;
        if (tree_bytes) {
            t_lc = reinterpret_cast<uint8_t *>(s_qtab + tab_entries);
            t_rc = t_lc + (size_t)rows * nq;
            t_root = t_rc + (size_t)rows * nq;
            for (uint32_t k = lane; k < rows * nq; k += 64) {
                const uint32_t row = k / nq, i = k % nq;
                const DTree &Q = C.qt[row];
                const bool in = Q.n == nq && i + 1 < nq;   // internal nodes nq .. 2 nq - 2
                t_lc[k] = in ? (uint8_t)Q.lc[i] : (uint8_t)0;
                t_rc[k] = in ? (uint8_t)Q.rc[i] : (uint8_t)0;
            }
            for (uint32_t k = lane; k < rows; k += 64) t_root[k] = C.qt[k].n == nq ? (uint8_t)C.qt[k].root : (uint8_t)0;
        }
        const uint32_t Kl = dec_fill_lds_table(s_qtab, tab_entries, C.qtab, C.Kq, rows);
        if (Kl && tree_bytes) {
            // (dec_mark_long_codes) a prefix whose code is longer than Kl bits: the internal node those Kl bits lead to, so that the
            // decoder's walk starts there instead of at the root
            const uint32_t total = rows << Kl;
            for (uint32_t k = lane; k < total; k += 64) {
                if (s_qtab[k]) continue;
                const uint32_t row = k >> Kl, x = k & ((1u << Kl) - 1u);
                uint32_t node = t_root[row], b = Kl;
                while (b && node >= nq) { --b; node = ((x >> b) & 1u) ? t_rc[row * nq + node - nq] : t_lc[row * nq + node - nq]; }
                if (!b && node >= nq && node - nq < 0x800u) s_qtab[k] = (uint16_t)(0x800u | (node - nq));
            }
            __syncthreads();
        }
        if (Kl) dec_symbols_wave<true, 16>(r.p, r.nbits, C, s_qtab, Kl, C.qt, D.qlen, n, d_qual, D.slen, s_map, s_in, s_out, rows, t_lc, t_rc, t_root, tree_bytes ? nq : 0u, true);
        else dec_symbols_wave<true>(r.p, r.nbits, C, C.qtab, C.Kq, C.qt, D.qlen, n, d_qual, D.slen, s_map, s_in, s_out, rows);
        __syncthreads();
        if (C.err) { if (lane == 0) O.err = C.err; return; }
    }

    // ---- DNA (cf. FetchDNA tasks.cpp:957-1033): alphabet; a Huffman-coded alphabet is decoded here, two-bit codes by the assembly kernel ----
    const uint64_t t2 = wall_clock64();
    {   // bases kept per record -> their total (every lane its share)
        uint64_t bases = 0;
        for (uint32_t i = lane; i < n; i += 64) bases += D.slen[i];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) bases += __shfl_xor(bases, d);
        if (lane == 0) C.qual_total_bases = bases;
    }
    __syncthreads();
    if (lane == 0) {
        r.seek(C.pos);
        r.align();
        const uint32_t nsym = O.nsym;
        for (uint32_t i = 0; i < 256; ++i) s_map[i] = 0;
        for (uint32_t i = 0; i < nsym; ++i) { const uint8_t c = (uint8_t)r.bits(8); O.dsym[i] = c; s_map[i] = c; }
        if (!O.plain) {
            if (!tree_load(r, nsym, pool, s_dt, stack)) C.err = r.bad == 2 ? ERR_NOMEM : ERR_FORMAT;
            else tree_table(s_dt, s_dtab, DEC_K, s_map);
        } else {
            if (r.bad || r.pos + 2 * C.qual_total_bases > r.nbits) C.err = ERR_FORMAT;
        }
        if (r.bad && !C.err) C.err = ERR_FORMAT;
        C.pos = r.pos;
        if (C.err) O.err = C.err;
    }
    __syncthreads();
    if (C.err) return;
    if (!O.plain && mode == 1) {
        // hand the code stream to the parallel kernels when every code fits the table (a zero entry is a longer code)
        __shared__ uint32_t s_ml;
        if (lane == 0) s_ml = 0;
        __syncthreads();
        uint32_t ml = 0, hole = 0;
        for (uint32_t k = lane; k < (1u << DEC_K); k += 64) { const uint32_t e = s_dtab[k]; hole |= e == 0; ml = (e >> 12) > ml ? (e >> 12) : ml; }
        if (hole) ml = 0xffu;
        atomicMax(&s_ml, ml);
        __syncthreads();
        const uint32_t maxlen = s_ml;
        uint16_t *gtab = nullptr;
        if (lane == 0 && maxlen >= 1 && maxlen <= DEC_K && C.qual_total_bases) gtab = (uint16_t *)pool.take(2u << DEC_K);
        gtab = (uint16_t *)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)gtab >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uint64_t)gtab));
        if (gtab) {
            for (uint32_t k = lane; k < (1u << DEC_K); k += 64) gtab[k] = s_dtab[k];
            if (lane == 0) {
                DecCtl &G = ctl[s];
                G.dna_spec = 1; G.dna_maxlen = maxlen; G.dna_nseg = 0; G.dna_bad = 0;
                G.dna_p0 = C.pos; G.dna_total = C.qual_total_bases; G.dna_tab = gtab;
                G.dna_n = s_dt.n; G.dna_root = s_dt.root; G.dna_lc = s_dt.lc; G.dna_rc = s_dt.rc;
                G.t_qual = t2 - t1;
                G.pos = C.pos; G.err = C.err; G.qual_total_bases = C.qual_total_bases; G.spec = C.spec; G.spec_bad = C.spec_bad;
            }
            return;
        }
    }
    if (!O.plain) {
        dec_symbols_wave<false>(r.p, r.nbits, C, s_dtab, DEC_K, &s_dt, D.slen, n, d_seq, nullptr, s_map, s_in, s_out, 1);
        __syncthreads();
        if (C.err) { if (lane == 0) O.err = C.err; return; }
    }
    if (lane == 0) {
        O.dna_bitpos = C.pos;
        O.t_head = C.t_head; O.t_qual = t2 - t1; O.t_dna = wall_clock64() - t2;
        O.spec = (C.spec == 2 && !C.spec_bad) ? 1u : 0u;
        if (mode == 1) ctl[s].dna_spec = 0;
    }
}

// ---- Huffman-coded bases in parallel -------------------------------------------------------------------------------------------
// The base stream of a sub-block with more than four symbols (an N that keeps its quality is enough) is ONE Huffman-coded bit
// string (tasks.cpp:544-557): 3.6 million codes per 8 MiB window of 150-base reads, which one wave decoded in 0.39 s - two
// thirds of the decoder's time on such files.  Unlike the qualities it has one tree for all positions, so where a code starts
// is the only state, and a stretch of the stream can be entered in at most `maxlen` ways (the code in front of it began less
// than a code length before its first bit):
//   A  a lane per (segment of DS_SEG bits, entry offset e < maxlen) walks the segment from e: where the walk leaves it (exit
//      offset into the next segment) and how many codes it met.  ~maxlen times the stream's codes in lookups, all independent;
//   B  one wave per sub-block follows the true chain - segment g is entered where segment g - 1 was left - and writes down each
//      segment's entry offset and the number of bases in front of it;
//   C  a lane per segment decodes its codes to their place in the output.
// A code longer than the table's DEC_K bits never gets here (k_dec_tail hands over only complete tables); a chain that does
// not add up (a damaged stream) marks the sub-block and the serial decoder of k_dec_tail reports what is wrong with it.
constexpr uint32_t DS_SEG = 1024;     // bits per segment
constexpr uint32_t DS_EP = 16;        // entry slots per segment (maxlen <= DEC_K = 9 of them used)
constexpr uint32_t DS_A_SEGS = 256 / DS_EP;   // segments per workgroup of kernel A
constexpr uint16_t DS_INVALID = 0xFFFFu;

// the 32 stream bits starting at bit `a` of a staged stretch (dwords of 32 stream bits each, MSB first)
__device__ __forceinline__ uint32_t ds_peek32(const uint32_t *s_in, uint32_t a)
{
    const uint32_t d = a >> 5, sh = a & 31;
    return sh ? __builtin_amdgcn_alignbit(s_in[d], s_in[d + 1], 32 - sh) : s_in[d];
}

__global__ __launch_bounds__(256) void k_dec_dna_a(const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, const DecCtl *ctl,
                                                   uint16_t *hop, const uint64_t *hop_off)
{
    const uint32_t s = blockIdx.y;
    const DecCtl &C = ctl[s];
    if (C.err || C.dna_spec != 1) return;
    const uint64_t nbits = (uint64_t)sb_len[s] * 8, p0 = C.dna_p0;
    const uint64_t nseg = (nbits - p0 + DS_SEG - 1) / DS_SEG;
    const uint64_t g0 = (uint64_t)blockIdx.x * DS_A_SEGS;
    if (g0 >= nseg) return;
    __shared__ uint32_t s_in[DS_A_SEGS * (DS_SEG / 32) + 4];
    __shared__ uint16_t s_tab[1u << DEC_K];
    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(sub + sb_off[s]);
    const uint32_t ndw = (uint32_t)((nbits + 31) >> 5);
    const uint64_t bit0 = p0 + g0 * DS_SEG;
    const uint32_t d0 = (uint32_t)(bit0 >> 5);
    for (uint32_t k = threadIdx.x; k < DS_A_SEGS * (DS_SEG / 32) + 4; k += 256) s_in[k] = d0 + k < ndw ? __builtin_bswap32(p32[d0 + k]) : 0u;
    for (uint32_t k = threadIdx.x; k < (1u << DEC_K); k += 256) s_tab[k] = C.dna_tab[k];
    __syncthreads();
    const uint32_t seg = threadIdx.x / DS_EP, e = threadIdx.x % DS_EP;
    if (g0 + seg >= nseg || e >= C.dna_maxlen) return;
    const uint32_t rel0 = (uint32_t)(bit0 & 31) + seg * DS_SEG;   // the segment's first bit inside the staged stretch
    uint32_t pos = e, cnt = 0;
    bool bad = false;
    while (pos < DS_SEG) {
        const uint32_t ent = s_tab[ds_peek32(s_in, rel0 + pos) >> (32 - DEC_K)];
        if (!ent) { bad = true; break; }
        pos += ent >> 12;
        ++cnt;
    }
    hop[hop_off[s] + (g0 + seg) * DS_EP + e] = bad ? DS_INVALID : (uint16_t)(((pos - DS_SEG) << 12) | cnt);
}

// B: per segment, (bases in front of it) << 4 | entry offset, written behind the table of A (u32 entries)
__global__ __launch_bounds__(64) void k_dec_dna_b(const uint32_t *sb_len, DecCtl *ctl, uint16_t *hop, const uint64_t *hop_off)
{
    const uint32_t s = blockIdx.x, lane = threadIdx.x;
    DecCtl &C = ctl[s];
    if (C.err || C.dna_spec != 1) return;
    const uint64_t nbits = (uint64_t)sb_len[s] * 8, p0 = C.dna_p0;
    const uint32_t nseg = (uint32_t)((nbits - p0 + DS_SEG - 1) / DS_SEG);
    const uint16_t *T = hop + hop_off[s];
    uint32_t *E = reinterpret_cast<uint32_t *>(hop + hop_off[s] + (((uint64_t)nseg * DS_EP + 7) & ~7ull));
    constexpr uint32_t CH = 256;   // segments staged at a time
    __shared__ __align__(16) uint16_t s_t[CH * DS_EP];
    __shared__ uint32_t s_e[CH];
    auto U = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
    const uint64_t total = ((uint64_t)U((uint32_t)(C.dna_total >> 32)) << 32) | U((uint32_t)C.dna_total);
    uint64_t out = 0;
    uint32_t e = 0, g = 0, bad = 0;
    for (uint32_t base = 0; base < nseg && out < total && !bad; base += CH) {
        const uint32_t cnt = nseg - base < CH ? nseg - base : CH;
        __syncthreads();
        for (uint32_t t = lane; t < cnt * DS_EP / 8; t += 64) reinterpret_cast<uint4 *>(s_t)[t] = reinterpret_cast<const uint4 *>(T + (uint64_t)base * DS_EP)[t];
        __syncthreads();
        uint32_t k = 0;
        for (; k < cnt && out < total; ++k) {
            const uint32_t t = U((uint32_t)s_t[k * DS_EP + e]);
            if (t == DS_INVALID || out >= (1ull << 28)) { bad = 1; break; }
            s_e[k] = ((uint32_t)out << 4) | e;
            out += t & 0xfffu;
            e = t >> 12;
        }
        __syncthreads();
        for (uint32_t t = lane; t < k; t += 64) E[base + t] = s_e[t];
        g = base + k;
    }
    if (lane == 0) {
        if (bad || out < total) { C.dna_spec = 3; }
        else { C.dna_nseg = g; C.dna_spec = 2; }
    }
}

// C: a lane per segment; a workgroup's 64 segments staged in LDS
constexpr uint32_t DS_C_SEGS = 64;
__global__ __launch_bounds__(64) void k_dec_dna_c(const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, DecCtl *ctl, uint8_t *scratch,
                                                  uint64_t stride, uint32_t rec_cap, uint64_t tcap, const uint16_t *hop, const uint64_t *hop_off)
{
    const uint32_t s = blockIdx.y;
    DecCtl &C = ctl[s];
    if (C.err || C.dna_spec != 2) return;
    const uint32_t g0 = blockIdx.x * DS_C_SEGS;
    if (g0 >= C.dna_nseg) return;
    const uint64_t nbits = (uint64_t)sb_len[s] * 8, p0 = C.dna_p0, total = C.dna_total;
    const uint32_t nseg_all = (uint32_t)((nbits - p0 + DS_SEG - 1) / DS_SEG);
    const uint32_t *E = reinterpret_cast<const uint32_t *>(hop + hop_off[s] + (((uint64_t)nseg_all * DS_EP + 7) & ~7ull));
    __shared__ uint32_t s_in[DS_C_SEGS * (DS_SEG / 32) + 4];
    __shared__ uint16_t s_tab[1u << DEC_K];
    const uint32_t *p32 = reinterpret_cast<const uint32_t *>(sub + sb_off[s]);
    const uint32_t ndw = (uint32_t)((nbits + 31) >> 5);
    const uint64_t bit0 = p0 + (uint64_t)g0 * DS_SEG;
    const uint32_t d0 = (uint32_t)(bit0 >> 5);
    for (uint32_t k = threadIdx.x; k < DS_C_SEGS * (DS_SEG / 32) + 4; k += 64) s_in[k] = d0 + k < ndw ? __builtin_bswap32(p32[d0 + k]) : 0u;
    for (uint32_t k = threadIdx.x; k < (1u << DEC_K); k += 64) s_tab[k] = C.dna_tab[k];
    __syncthreads();
    const uint32_t g = g0 + threadIdx.x;
    if (g >= C.dna_nseg) return;
    DecScratch D = dec_scratch(scratch, stride, s, rec_cap, tcap, 0);
    uint8_t *dst = D.titles + D.lay[1];
    const uint32_t ent0 = E[g];
    uint64_t out = ent0 >> 4;
    uint32_t pos = ent0 & 15u;
    const uint32_t rel0 = (uint32_t)(bit0 & 31) + threadIdx.x * DS_SEG;
    const uint64_t seg_bit0 = bit0 + (uint64_t)threadIdx.x * DS_SEG;
    while (pos < DS_SEG && out < total) {
        const uint32_t ent = s_tab[ds_peek32(s_in, rel0 + pos) >> (32 - DEC_K)];
        if (!ent) { atomicOr(&C.dna_bad, 1u); return; }
        pos += ent >> 12;
        if (seg_bit0 + pos > nbits) { atomicOr(&C.dna_bad, 1u); return; }   // a code that runs past the sub-block's end
        dst[out++] = (uint8_t)ent;
    }
    if (out >= total && (pos < DS_SEG || g + 1 == C.dna_nseg)) ctl[s].pos = seg_bit0 + pos;   // (the lane that decodes the last code: where the stream ends)
}


// The four lines of every record, records dealt to the threads.  (round 4) gridDim.y workgroups per sub-block, a stretch of its
// records each (a workgroup first adds up the lengths of the records in front of its stretch: a few hundred KB of coalesced reads
// instead of a table of offsets) - one workgroup per sub-block left 120 of them to write a GB of text: 45 ms, a tenth of the
// decode of 36..300-base reads.
__global__ __launch_bounds__(256) void k_dec_assemble(const uint8_t *sub, const uint64_t *sb_off, uint32_t n_sb, uint8_t *scratch,
                                                      uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap, const DecSub *in,
                                                      const uint64_t *text_off, uint8_t *text)
{
    const uint32_t s = blockIdx.x;
    if (s >= n_sb) return;
    const DecSub &O = in[s];
    if (O.err) return;
    __shared__ uint64_t s_run[2];   // running text offset and base count in front of the current batch
    __shared__ uint64_t s_wt[4], s_wb[4], s_wq[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    DecScratch D = dec_scratch(scratch, stride, s, rec_cap, tcap, qcap);
    const uint8_t *sb = sub + sb_off[s];
    uint8_t *dst = text + text_off[s];
    const uint8_t amb_base[17] = {0, 0, 'Y', 'R', 'W', 'S', 'K', 'M', 'D', 'V', 'H', 'B', 'N', 'X', 'U', '.', '-'};
    const uint32_t nf = O.nf;
    const AField *af = (const AField *)(D.pool + O.af_off);
    // this workgroup's stretch of records, and what lies in front of it
    const uint32_t chunk = ((O.n_rec + gridDim.y - 1) / gridDim.y + 255u) & ~255u;
    const uint32_t r0 = blockIdx.y * chunk, r1 = r0 + chunk < O.n_rec ? r0 + chunk : O.n_rec;
    if (r0 >= O.n_rec) return;
    uint64_t qbase = 0;   // qualities are stored back to back: running offset = sum of the read lengths
    {
        uint64_t st = 0, sq = 0, sbs = 0;
        for (uint32_t i = tid; i < r0; i += 256) {
            const uint32_t L = D.qlen[i];
            st += (uint64_t)D.toff[i] + 2ull * L + 4;
            sq += L;
            sbs += D.slen[i];
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { st += __shfl_xor(st, d); sq += __shfl_xor(sq, d); sbs += __shfl_xor(sbs, d); }
        if (lane == 0) { s_wt[wave] = st; s_wq[wave] = sq; s_wb[wave] = sbs; }
        __syncthreads();
        qbase = s_wq[0] + s_wq[1] + s_wq[2] + s_wq[3];
        if (tid == 0) { s_run[0] = s_wt[0] + s_wt[1] + s_wt[2] + s_wt[3]; s_run[1] = s_wb[0] + s_wb[1] + s_wb[2] + s_wb[3]; }
        __syncthreads();
    }
    for (uint32_t base = r0; base < r1; base += 256) {
        const uint32_t i = base + tid;
        const bool valid = i < r1;
        const uint32_t L = valid ? D.qlen[i] : 0, kept = valid ? D.slen[i] : 0;
        const uint32_t tl = valid ? D.toff[i] : 0;   // title bytes of the record (k_dec_head)
        // exclusive scans over the batch: text bytes, quality bytes, bases
        uint64_t tb = valid ? (uint64_t)tl + 2ull * L + 4 : 0, qb = L, bb = kept;
        uint64_t it = tb, iq = qb, ib = bb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint64_t a = __shfl_up(it, d), b2 = __shfl_up(iq, d), c2 = __shfl_up(ib, d);
            if ((int)lane >= d) { it += a; iq += b2; ib += c2; }
        }
        __syncthreads();
        if (lane == 63) { s_wt[wave] = it; s_wb[wave] = ib; s_wq[wave] = iq; }
        __syncthreads();
        uint64_t bt = 0, bq = 0, bbs = 0, tt = 0, tq = 0, tbs = 0;
        for (uint32_t w = 0; w < 4; ++w) {
            if (w < wave) { bt += s_wt[w]; bq += s_wq[w]; bbs += s_wb[w]; }
            tt += s_wt[w]; tq += s_wq[w]; tbs += s_wb[w];
        }
        const uint64_t o_text = s_run[0] + bt + it - tb, o_q = qbase + bq + iq - qb, o_b = s_run[1] + bbs + ib - bb;
        if (valid) {
            uint8_t *o = dst + o_text;
            {   // the title, field by field, from the record's value words (ends with the last field's separator, '\n')
                const uint32_t *vrow = (const uint32_t *)D.titles + (size_t)i * O.nvw;
                const uint8_t *tch = D.titles + O.tchar_off;
                uint8_t *w = o;
                for (uint32_t f = 0; f < nf; ++f) {
                    const AField a = af[f];
                    if (a.kind == 0) {
                        const uint8_t *c = D.pool + a.data_off;
                        for (uint32_t j = 0; j < a.len0; ++j) w[j] = c[j];
                        w += a.len0;
                    } else if (a.kind == 1) {
                        w += dec_put_decimal(w, vrow[a.vidx]);
                    } else {
                        const uint8_t *c = tch + vrow[a.vidx];
                        const uint32_t L = vrow[a.vidx + 1];
                        for (uint32_t j = 0; j < L; ++j) w[j] = c[j];
                        w += L;
                    }
                    *w++ = (uint8_t)a.sep;
                }
            }
            o += tl;
            const uint8_t *q = D.titles + D.lay[0] + o_q;
            uint64_t bi = o_b;                                   // index of the record's next base in the DNA stream
            uint32_t j = 0;
            // (round 4) eight positions per step where nothing was transferred into the quality stream (no byte >= 128): one 8-byte
            // load of qualities, the sixteen bits of eight two-bit bases (or eight decoded base bytes), two 8-byte stores.  A byte per
            // step - a quality, a base byte and the alphabet entry, each a trip to memory the next one waits for - made this
            // kernel 24 ms per GB of 36..300-base reads: latency, not bytes.
            const uint32_t sym4 = O.plain ? ((uint32_t)O.dsym[0] | ((uint32_t)O.dsym[1] << 8) | ((uint32_t)O.dsym[2] << 16) | ((uint32_t)O.dsym[3] << 24)) : 0u;
            auto one = [&](uint32_t jj) __attribute__((always_inline)) {   // position jj, the general way
                const uint32_t qv = q[jj];
                uint8_t c, qo = (uint8_t)qv;
                if (qv >= 128) {   // inverse of the N -> quality transfer (phyNGSC.cpp:573-588)
                    const uint32_t v = qv - 128u + 16u, a = v >> 3;
                    c = a <= 16 ? amb_base[a] : 0;
                    qo = (uint8_t)(33 + (v & 7));
                } else if (O.plain) {
                    const uint64_t bp = O.dna_bitpos + 2 * bi;
                    c = (uint8_t)(sym4 >> (8 * ((sb[bp >> 3] >> (6 - (bp & 7))) & 3u)));
                    ++bi;
                } else {
                    c = D.titles[D.lay[1] + bi++];
                }
                o[jj] = c;
                o[L + 3 + jj] = qo;
            };
            for (; j + 8 <= L; j += 8) {
                const U2u qq = *reinterpret_cast<const U2u *>(q + j);
                if ((qq.x | qq.y) & 0x80808080u) {               // a transferred position among these eight: one by one
                    for (uint32_t k = 0; k < 8; ++k) one(j + k);
                    continue;
                }
                uint32_t b0, b1;                                 // the eight bases, a byte each
                if (O.plain) {
                    const uint64_t bp = O.dna_bitpos + 2 * bi;
                    const uint8_t *pb = sb + (bp >> 3);
                    const uint32_t w24 = ((uint32_t)pb[0] << 16) | ((uint32_t)pb[1] << 8) | (uint32_t)pb[2];   // (a sub-block is padded to 16 bytes: pb[2] is there)
                    const uint32_t c16 = (w24 >> (8 - (uint32_t)(bp & 7))) & 0xffffu;                            // sixteen bits, the first base on top
                    b0 = b1 = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        b0 |= ((sym4 >> (8 * ((c16 >> (14 - 2 * k)) & 3u))) & 0xffu) << (8 * k);
                        b1 |= ((sym4 >> (8 * ((c16 >> (6 - 2 * k)) & 3u))) & 0xffu) << (8 * k);
                    }
                } else {
                    const U2u bb = *reinterpret_cast<const U2u *>(D.titles + D.lay[1] + bi);
                    b0 = bb.x; b1 = bb.y;
                }
                bi += 8;
                U2u so = {b0, b1};
                *reinterpret_cast<U2u *>(o + j) = so;
                *reinterpret_cast<U2u *>(o + L + 3 + j) = qq;
            }
            for (; j < L; ++j) one(j);
            o[L] = '\n'; o[L + 1] = '+'; o[L + 2] = '\n'; o[2 * L + 3] = '\n';
        }
        __syncthreads();
        if (tid == 0) { s_run[0] += tt; s_run[1] += tbs; }
        qbase += tq;
        __syncthreads();
    }
}

size_t dec_ctl_bytes() { return sizeof(DecCtl); }
uint32_t dec_qspec_tile() { return QS_TILE; }

// head -> (speculative quality kernels) -> tail.  hop: 2 bytes per bit of the batch's sub-blocks (hop_off: each sub-block's
// first entry), or nullptr: every quality stream is decoded serially by the tail kernel.  summary: one device word that
// receives the longest read length among the sub-blocks the speculative kernels will take (0: none) - the host reads it
// between the two calls and sizes their LDS tables with it (spec_max_q).
static double dec_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void dec_lap(hipStream_t st, double &t0, const char *name)
{
    static const bool verbose = getenv("NGSC_VERBOSE") != nullptr;
    if (!verbose) return;
    (void)hipStreamSynchronize(st);
    const double t1 = dec_now();
    if (name) fprintf(stderr, "[ngsc decode] %-20s %8.2f ms\n", name, t1 - t0);
    t0 = t1;
}

void launch_dec_head(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb, uint8_t *scratch,
                     uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap, void *out, void *ctl, bool allow_spec, uint32_t *summary)
{
    double t0 = 0;
    dec_lap(st, t0, nullptr);
    hipLaunchKernelGGL(k_dec_head, dim3(n_sb), dim3(64), 0, st, sub, sb_off, sb_len, n_sb, scratch, stride, rec_cap, tcap, qcap, (DecSub *)out,
                       (DecCtl *)ctl, allow_spec ? 1u : 0u, summary);
    dec_lap(st, t0, "k_dec_head");
}

void launch_dec_streams(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb, uint8_t *scratch,
                        uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap, void *out, void *ctl, uint16_t *hop, const uint64_t *hop_off,
                        uint32_t max_tiles, uint32_t spec_max_q, uint32_t spec_max_nq)
{
    // LDS of k_dec_tail's (single-wave) workgroups: 16-bit table entries and the rows' trees as byte tables.  A batch that cannot fill
    // the chip anyway (up to a wave per CU) gets 120 KB each - seven look-ahead bits for reads of up to 400 bases beside their trees;
    // larger batches 60 KB, so that two workgroups share a CU
    const uint32_t lds_entries = n_sb <= 256 ? (60u << 10) : (30u << 10);
    double t0 = 0;
    dec_lap(st, t0, nullptr);
    if (hop && spec_max_q) {
        // LDS of the speculative kernels: the tables of the batch's longest equal-length reads, as wide as 120 KB allow
        // (what a workgroup does not use costs occupancy: 38 KB at 36 bases lets four workgroups share a CU)
        const uint32_t lds_cap = 60u << 10, rows = spec_max_q + 1;
        uint32_t K = DEC_K;
        while (K > 5 && ((uint64_t)rows << K) > lds_cap) --K;
        const uint32_t lds_spec = (uint32_t)std::min<uint64_t>(lds_cap, (uint64_t)rows << K);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_dec_qspec_a), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds_cap));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_dec_qspec_c), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds_cap));
        // phases: what the first one leaves uncovered is small when few bases were transferred to the quality stream (each
        // moves the stream's end by a few bits), so the next ones are small; the last takes whatever is left
        // kernel A: byte entries (lengths only) - as many look-ahead bits as 60 KB of them allow - and the trees (up to 128
        // symbols: rows x 128 x 2 + rows bytes at most; without room for them long codes walk the trees in global memory)
        uint32_t Ka = DEC_K;
        while (Ka > 5 && ((uint64_t)rows << Ka) > lds_cap) --Ka;
        const uint32_t lds_a_entries = (uint32_t)std::min<uint64_t>(lds_cap, (uint64_t)rows << Ka);
        const uint32_t nq_cap = spec_max_nq < 2 ? 2u : (spec_max_nq > 128 ? 128u : spec_max_nq);
        size_t tree_bytes = (size_t)rows * (2 * nq_cap + 1) + 32;
        if (((size_t)lds_a_entries + 15 & ~(size_t)15) + tree_bytes > (120u << 10)) tree_bytes = 0;   // (long reads: no room for the trees)
        const size_t lds_a = ((size_t)lds_a_entries + 15 & ~(size_t)15) + tree_bytes;
        const uint32_t phase_tiles[4] = {max_tiles, 16, 256, max_tiles};
        for (int ph = 0; ph < 4; ++ph) {
            hipLaunchKernelGGL(k_dec_qspec_a, dim3(phase_tiles[ph], n_sb), dim3(QS_THREADS), lds_a, st, sub, sb_off, sb_len, (const DecCtl *)ctl,
                               hop, hop_off, lds_a_entries, tree_bytes ? 1u : 0u);
            if (ph == 0) dec_lap(st, t0, "k_dec_qspec_a");
            hipLaunchKernelGGL(k_dec_qspec_b, dim3(n_sb), dim3(64), 0, st, sb_len, (DecCtl *)ctl, hop, hop_off, scratch, stride, rec_cap, tcap,
                               ph < 3 ? phase_tiles[ph + 1] : 0u);
            if (ph == 0) dec_lap(st, t0, "k_dec_qspec_b");
        }
        dec_lap(st, t0, "k_dec_qspec a+b 1-3");
        hipLaunchKernelGGL(k_dec_qspec_c, dim3((rec_cap + QS_THREADS - 1) / QS_THREADS, n_sb), dim3(QS_THREADS), 2 * lds_spec, st, sub, sb_off, sb_len,
                           (DecCtl *)ctl, scratch, stride, rec_cap, tcap, lds_spec);
        dec_lap(st, t0, "k_dec_qspec_c");
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_dec_tail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds_entries));
    const bool dna_par = !(getenv("NGSC_DEC_DNA_PAR") && !strcmp(getenv("NGSC_DEC_DNA_PAR"), "0"));
    const uint32_t mode = hop && dna_par ? 1u : 0u;   // (the parallel base decoder keeps its tables in the hop area)
    hipLaunchKernelGGL(k_dec_tail, dim3(n_sb), dim3(64), 2 * lds_entries, st, sub, sb_off, sb_len, n_sb, scratch, stride, rec_cap, tcap, qcap,
                       (DecSub *)out, (DecCtl *)ctl, lds_entries, mode);
    dec_lap(st, t0, "k_dec_tail");
    if (mode) {
        // segments of the longest sub-block of the batch: max_tiles counts QS_TILE bit offsets of a whole sub-block
        const uint32_t max_seg = (uint32_t)(((uint64_t)max_tiles * QS_TILE + DS_SEG - 1) / DS_SEG) + 1;
        hipLaunchKernelGGL(k_dec_dna_a, dim3((max_seg + DS_A_SEGS - 1) / DS_A_SEGS, n_sb), dim3(256), 0, st, sub, sb_off, sb_len, (const DecCtl *)ctl, hop, hop_off);
        hipLaunchKernelGGL(k_dec_dna_b, dim3(n_sb), dim3(64), 0, st, sb_len, (DecCtl *)ctl, hop, hop_off);
        hipLaunchKernelGGL(k_dec_dna_c, dim3((max_seg + DS_C_SEGS - 1) / DS_C_SEGS, n_sb), dim3(64), 0, st, sub, sb_off, sb_len, (DecCtl *)ctl, scratch, stride,
                           rec_cap, tcap, hop, hop_off);
        dec_lap(st, t0, "k_dec_dna a+b+c");
        hipLaunchKernelGGL(k_dec_tail, dim3(n_sb), dim3(64), 2 * lds_entries, st, sub, sb_off, sb_len, n_sb, scratch, stride, rec_cap, tcap, qcap,
                           (DecSub *)out, (DecCtl *)ctl, lds_entries, 2u);
        dec_lap(st, t0, "k_dec_tail (close)");
    }
}

void launch_dec_assemble(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, uint32_t n_sb, uint8_t *scratch, uint64_t stride,
                         uint32_t rec_cap, uint64_t tcap, uint64_t qcap, const void *in, const uint64_t *text_off, uint8_t *text)
{
    static const bool verbose = getenv("NGSC_VERBOSE") != nullptr;
    double t0 = 0;
    if (verbose) { (void)hipStreamSynchronize(st); t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    // (stretches of records per sub-block: enough workgroups for the chip from a hundred sub-blocks, a few thousand records each)
    const uint32_t parts = n_sb >= 2048 ? 1u : (n_sb >= 256 ? 4u : 16u);
    hipLaunchKernelGGL(k_dec_assemble, dim3(n_sb, parts), dim3(256), 0, st, sub, sb_off, n_sb, scratch, stride, rec_cap, tcap, qcap, (const DecSub *)in,
                       text_off, text);
    if (verbose) {
        (void)hipStreamSynchronize(st);
        fprintf(stderr, "[ngsc decode] %-16s %8.2f ms\n", "k_dec_assemble",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0);
    }
}

size_t dec_sub_bytes() { return sizeof(DecSub); }
int32_t dec_sub_err(const void *h, uint32_t s) { return ((const DecSub *)h)[s].err; }
uint64_t dec_sub_text(const void *h, uint32_t s) { return ((const DecSub *)h)[s].text_bytes; }
uint32_t dec_sub_records(const void *h, uint32_t s) { return ((const DecSub *)h)[s].n_rec; }
uint32_t dec_sub_spec(const void *h, uint32_t s) { return ((const DecSub *)h)[s].spec; }
void dec_sub_ticks(const void *h, uint32_t s, uint64_t *t) { const DecSub &d = ((const DecSub *)h)[s]; t[0] = d.t_head; t[1] = d.t_qual; t[2] = d.t_dna; for (int k = 0; k < 5; ++k) t[3 + k] = d.t_ph[k]; }

}  // namespace ngsc
