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
//   k_dec_streams   serial decode per sub-block -> titles, qualities (+ sizes)
//   k_dec_assemble  per record: title \n bases \n + \n qualities \n
// Integer / bit work; bounded by dependent-load latency of one lane per wave (no MFMA, no bandwidth to speak of).
#include "ngsc_kernels.h"
#include "ngsc_swar.h"

namespace ngsc {

constexpr uint32_t DEC_MAX_FIELDS = 64;

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
    __device__ void fill()
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
    __device__ uint32_t bits(uint32_t n)   // n <= 32
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
    __device__ uint32_t peek(uint32_t n)   // the next n <= 16 bits without moving (zeros behind the end of the stream)
    {
        if (have < n) fill();
        return (uint32_t)(win >> (64 - n));
    }
    __device__ void skip(uint32_t n)       // n bits that peek() has shown
    {
        if (!n) return;
        if (pos + n > nbits) { bad = bad ? bad : 1; pos = nbits; have = 0; win = 0; return; }
        win <<= n;
        have -= n;
        pos += n;
    }
    __device__ uint32_t bit() { return bits(1); }
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
__device__ void tree_table(DTree &t, uint16_t *tab, uint32_t K = DEC_K)
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
            const uint16_t v = (uint16_t)((len << 12) | node);
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

__device__ bool tree_load(BitIn &r, uint32_t expect_n, Pool &pool, DTree &t, uint16_t *stack /* 2n + 2 */)
{
    const uint32_t blen = r.u32();
    const uint64_t end = r.pos + (uint64_t)blen * 8;
    const uint32_t root = r.u32(), n = r.u32();
    (void)r.u8();
    if (r.bad || n != expect_n || n == 0 || n > 512 || root >= 2 * n || end > r.nbits) { r.bad = 1; return false; }
    t.n = n;
    t.lc = (uint16_t *)pool.take(2 * (n + 1));
    t.rc = (uint16_t *)pool.take(2 * (n + 1));
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
    uint8_t dsym[256];
};

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
    uint32_t *qlen, *toff, *slen;
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
    b = (uint8_t *)(((uintptr_t)b + 15) & ~(uintptr_t)15);
    D.lay = (uint64_t *)b; b += 16;
    D.titles = b; b += tcap;
    D.pool = b;
    D.tcap = tcap;
    D.pool_cap = stride - (uint64_t)(b - (base + (uint64_t)s * stride));
    return D;
}

// One wave per sub-block; lane 0 decodes the serial streams.
__global__ __launch_bounds__(64) void k_dec_streams(const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb,
                                                    uint8_t *scratch, uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap,
                                                    DecSub *out, uint32_t lds_entries)
{
    const uint32_t s = blockIdx.x;
    if (s >= n_sb || threadIdx.x != 0) return;
    DecSub &O = out[s];
    O.err = OK; O.text_bytes = 0; O.n_rec = 0;
    BitIn r;
    r.init(sub + sb_off[s], sb_len[s]);
    DecScratch D = dec_scratch(scratch, stride, s, rec_cap, tcap, qcap);
    Pool pool = {D.pool, 0, D.pool_cap};
    uint16_t *stack = (uint16_t *)pool.take(2 * (2 * 512 + 2));

    // ---- info header phyNGSC.cpp:719-742 ----
    const uint32_t n = r.u32(), max_q = r.u32();
    (void)r.u32();
    const uint32_t nsym = r.u8();
    (void)r.u8();
    const uint32_t nqual = r.u8(), flags = r.u32();
    if (r.bad || n == 0 || n > rec_cap || !nsym || !nqual || max_q > (1u << 24)) { O.err = ERR_FORMAT; return; }
    uint64_t qual_total = 0;
    if (flags & F_USE_DELTA) { O.err = ERR_COLORSPACE; return; }   // colour space is stored lossily (phyNGSC.cpp:531): nothing to give back
    O.n_rec = n; O.max_q = max_q; O.flags = flags; O.nsym = nsym; O.plain = (flags & F_DNA_PLAIN) ? 1u : 0u;
    {
        const uint32_t b = bit_length(max_q);
        uint64_t tot = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t L = r.bits(b > 32 ? 32 : b);
            D.qlen[i] = L;
            tot += L;
            if (L > max_q) r.bad = 1;
        }
        r.align();
        if (r.bad || 2 * tot > D.tcap) { O.err = r.bad ? ERR_FORMAT : ERR_CAPACITY; return; }
        qual_total = tot;
    }

    // ---- title header (cf. FetchTitleHeader tasks.cpp:625-735; value-coding trees as the encoder stores them, Q2b) ----
    const uint32_t nf = r.u32();
    if (r.bad || nf == 0 || nf > DEC_MAX_FIELDS) { O.err = r.bad ? ERR_FORMAT : ERR_UNSUPPORTED; return; }
    __shared__ DField s_F[DEC_MAX_FIELDS];   // (a lone lane pays a full memory latency for every field it touches in global memory)
    DField *F = s_F;
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
    if (r.bad) { O.err = r.bad == 2 ? ERR_NOMEM : ERR_FORMAT; return; }

    // ---- title body (cf. FetchTitleBody tasks.cpp:738-954): groups of 32 records, byte aligned ----
    {
        uint64_t tn = 0;
        for (uint32_t g0 = 0; g0 < n && !r.bad; g0 += GROUP) {
            const uint32_t g1 = g0 + GROUP < n ? g0 + GROUP : n;
            for (uint32_t f = 0; f < nf; ++f)
                if (!F[f].is_const) { F[f].gflag = (uint8_t)r.bit(); F[f].prev = 0; }
            for (uint32_t i = g0; i < g1 && !r.bad; ++i) {
                D.toff[i] = (uint32_t)tn;
                for (uint32_t f = 0; f < nf; ++f) {
                    DField &fd = F[f];
                    if (tn + fd.max_len + fd.len0 + 16 > D.tcap) { r.bad = 3; break; }
                    if (fd.is_const) {
                        for (uint32_t j = 0; j < fd.len0; ++j) D.titles[tn + j] = fd.data0[j];
                        tn += fd.len0;
                    } else if (fd.is_num) {
                        int32_t v;
                        if (i == g0) v = fd.min_v + (int32_t)r.bits(fd.bits_val > 32 ? 32 : fd.bits_val);
                        else if (fd.gflag) v = fd.delta_coding ? fd.prev + fd.min_d : fd.prev;
                        else {
                            const int32_t x = fd.has_g ? (int32_t)tree_get(fd.g, r) : (int32_t)r.bits(fd.bits_num > 32 ? 32 : fd.bits_num);
                            v = fd.delta_coding ? fd.prev + fd.min_d + x : fd.min_v + x;
                        }
                        fd.prev = v;
                        tn += dec_put_decimal(D.titles + tn, (uint32_t)v);
                    } else if (i != g0 && fd.gflag) {
                        for (uint32_t j = 0; j < fd.gval_len; ++j) D.titles[tn + j] = fd.gval[j];
                        tn += fd.gval_len;
                    } else {
                        const uint32_t L = fd.is_len_const ? fd.len0 : fd.min_len + r.bits(fd.bits_len > 32 ? 32 : fd.bits_len);
                        if (L > fd.max_len) { r.bad = 1; break; }
                        for (uint32_t j = 0; j < L; ++j) {
                            uint8_t c;
                            if (j < fd.len0 && fd.ham[j]) c = fd.data0[j];
                            else {
                                const DTree &t = fd.l[j < MAX_STAT_LEN ? j : MAX_STAT_LEN];
                                if (!t.n) { r.bad = 1; break; }
                                c = (uint8_t)tree_get(t, r);
                            }
                            D.titles[tn + j] = c;
                            if (i == g0) fd.gval[j] = c;
                        }
                        if (i == g0) fd.gval_len = L;
                        tn += L;
                    }
                    D.titles[tn++] = fd.sep;
                }
            }
            r.align();
        }
        D.toff[n] = (uint32_t)tn;
    }
    if (r.bad) { O.err = r.bad == 3 ? ERR_CAPACITY : ERR_FORMAT; return; }

    // ---- quality (cf. FetchQuality tasks.cpp:1036-1101): alphabet, max_q + 1 trees, codes by position ----
    uint64_t text = D.toff[n];
    const uint64_t q_at = (text + 15) & ~15ull, s_at = q_at + ((qual_total + 15) & ~15ull);
    if (s_at + qual_total > D.tcap) { O.err = ERR_CAPACITY; return; }
    D.lay[0] = q_at; D.lay[1] = s_at;
    uint8_t *d_qual = D.titles + q_at, *d_seq = D.titles + s_at;
    {
        __shared__ uint8_t qsym[256];
        r.align();
        for (uint32_t i = 0; i < nqual; ++i) qsym[i] = (uint8_t)r.bits(8);
        DTree *qt = (DTree *)pool.take(sizeof(DTree) * ((uint64_t)max_q + 1));
        if (!qt) { O.err = ERR_NOMEM; return; }
        // the per-position tables: in LDS while they fit (the quality codes are most of the sub-block's symbols)
        extern __shared__ __align__(16) uint16_t s_qtab[];
        uint32_t Kq = DEC_K;   // the widest tables that fit (codes longer than Kq bits walk their tree)
        while (Kq > 5 && (((uint64_t)max_q + 1) << Kq) > lds_entries) --Kq;
        const bool q_lds = (((uint64_t)max_q + 1) << Kq) <= lds_entries;
        uint16_t *qtab = q_lds ? s_qtab : (uint16_t *)pool.take((((uint64_t)max_q + 1) << Kq) * 2);
        if (!qtab) { O.err = ERR_NOMEM; return; }
        for (uint32_t p = 0; p <= max_q && !r.bad; ++p)
            if (tree_load(r, nqual, pool, qt[p], stack)) tree_table(qt[p], qtab + ((size_t)p << Kq), Kq);
        r.align();
        if (r.bad) { O.err = r.bad == 2 ? ERR_NOMEM : ERR_FORMAT; return; }
        uint64_t k = 0;
        for (uint32_t i = 0; i < n && !r.bad; ++i) {
            const uint32_t L = D.qlen[i];
            uint32_t kept = 0;
            for (uint32_t j = 0; j < L; ++j) {
                const uint32_t e = qtab[((size_t)(j + 1) << Kq) + r.peek(Kq)];
                uint32_t sym;
                if (e) { r.skip(e >> 12); sym = e & 0xfffu; }
                else sym = tree_get(qt[j + 1], r);
                const uint8_t q = qsym[sym];
                d_qual[k++] = q;
                kept += q < 128;   // (a byte from 128 on is a transferred ambiguity code: no base in the DNA stream)
            }
            D.slen[i] = kept;
            text += 2ull * L + 4;
        }
        r.align();
    }
    if (r.bad) { O.err = ERR_FORMAT; return; }

    // ---- DNA (cf. FetchDNA tasks.cpp:957-1033): alphabet; a Huffman-coded alphabet is decoded here, two-bit codes by all lanes ----
    for (uint32_t i = 0; i < nsym; ++i) O.dsym[i] = (uint8_t)r.bits(8);
    if (!O.plain) {
        DTree dt;
        if (!tree_load(r, nsym, pool, dt, stack)) { O.err = r.bad == 2 ? ERR_NOMEM : ERR_FORMAT; return; }
        tree_table(dt, (uint16_t *)pool.take(2u << DEC_K));
        uint64_t k = 0;
        for (uint32_t i = 0; i < n && !r.bad; ++i)
            for (uint32_t j = 0; j < D.slen[i]; ++j) {
                const uint32_t c = tree_get(dt, r);
                if (c >= nsym) { r.bad = 1; break; }
                d_seq[k++] = O.dsym[c];
            }
        if (r.bad) { O.err = ERR_FORMAT; return; }
    } else {
        uint64_t bases = 0;
        for (uint32_t i = 0; i < n; ++i) bases += D.slen[i];
        if (r.pos + 2 * bases > r.nbits) { O.err = ERR_FORMAT; return; }
    }
    O.dna_bitpos = r.pos;
    O.text_bytes = text;
}

// One workgroup per sub-block: the four lines of every record, records dealt to the threads.
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
    if (tid == 0) { s_run[0] = 0; s_run[1] = 0; }
    __syncthreads();
    uint64_t qbase = 0;   // qualities are stored back to back: running offset = sum of the read lengths
    for (uint32_t base = 0; base < O.n_rec; base += 256) {
        const uint32_t i = base + tid;
        const bool valid = i < O.n_rec;
        const uint32_t L = valid ? D.qlen[i] : 0, kept = valid ? D.slen[i] : 0;
        const uint32_t tl = valid ? D.toff[i + 1] - D.toff[i] : 0;
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
            const uint8_t *t = D.titles + D.toff[i];
            for (uint32_t j = 0; j < tl; ++j) o[j] = t[j];      // (ends with the last field's separator, '\n')
            o += tl;
            const uint8_t *q = D.titles + D.lay[0] + o_q;
            uint64_t bi = o_b;                                   // index of the record's next base in the DNA stream
            for (uint32_t j = 0; j < L; ++j) {
                const uint32_t qv = q[j];
                uint8_t c, qo = (uint8_t)qv;
                if (qv >= 128) {   // inverse of the N -> quality transfer (phyNGSC.cpp:573-588)
                    const uint32_t v = qv - 128u + 16u, a = v >> 3;
                    c = a <= 16 ? amb_base[a] : 0;
                    qo = (uint8_t)(33 + (v & 7));
                } else if (O.plain) {
                    const uint64_t bp = O.dna_bitpos + 2 * bi;
                    c = O.dsym[(sb[bp >> 3] >> (6 - (bp & 7))) & 3u];
                    ++bi;
                } else {
                    c = D.titles[D.lay[1] + bi++];
                }
                o[j] = c;
                o[L + 3 + j] = qo;
            }
            o[L] = '\n'; o[L + 1] = '+'; o[L + 2] = '\n'; o[2 * L + 3] = '\n';
        }
        __syncthreads();
        if (tid == 0) { s_run[0] += tt; s_run[1] += tbs; }
        qbase += tq;
        __syncthreads();
    }
}

void launch_dec_streams(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb, uint8_t *scratch,
                        uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap, void *out)
{
    const uint32_t lds_entries = 30u << 10;   // 60 KB of 16-bit table entries: the per-position tables of reads up to 59 bases
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_dec_streams), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * lds_entries));
    hipLaunchKernelGGL(k_dec_streams, dim3(n_sb), dim3(64), 2 * lds_entries, st, sub, sb_off, sb_len, n_sb, scratch, stride, rec_cap, tcap, qcap,
                       (DecSub *)out, lds_entries);
}

void launch_dec_assemble(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, uint32_t n_sb, uint8_t *scratch, uint64_t stride,
                         uint32_t rec_cap, uint64_t tcap, uint64_t qcap, const void *in, const uint64_t *text_off, uint8_t *text)
{
    hipLaunchKernelGGL(k_dec_assemble, dim3(n_sb), dim3(256), 0, st, sub, sb_off, n_sb, scratch, stride, rec_cap, tcap, qcap, (const DecSub *)in,
                       text_off, text);
}

size_t dec_sub_bytes() { return sizeof(DecSub); }
int32_t dec_sub_err(const void *h, uint32_t s) { return ((const DecSub *)h)[s].err; }
uint64_t dec_sub_text(const void *h, uint32_t s) { return ((const DecSub *)h)[s].text_bytes; }
uint32_t dec_sub_records(const void *h, uint32_t s) { return ((const DecSub *)h)[s].n_rec; }

}  // namespace ngsc
