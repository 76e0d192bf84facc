/*
 * ngsc_decode.c - CPU decoder for one phyNGSC sub-block (info | title | quality | DNA -> FASTQ text).
 *
 * TEST INFRASTRUCTURE ONLY (see ngsc_oracle.h): the round-trip leg of the parity tests.  The product
 * (phyngsc_amd/csrc) is an encoder; nothing there links or calls this file.
 *
 * The reference ships decoder fragments that nothing calls (FetchTitleHeader/FetchTitleBody
 * tasks.cpp:625-954, FetchDNA :957-1033, FetchQuality :1036-1101, HuffmanEncoder::LoadTree
 * huffman.cpp:122-219) and no decompressor.  This file is the inverse of oracle/ngsc_oracle.c's
 * ngsc_o_subblock(), written from the encoder's bit layout; the Fetch* fragments are cited where they
 * agree.  One place where they do NOT agree with the reference's own encoder: FetchTitleHeader
 * (tasks.cpp:689) expects a Huffman tree for every numeric field with a range <= 512, while the
 * encoder omits it for value-coded fields whose statistics map was dropped by the vector reallocation
 * (quirk Q2b, ngsc_oracle.c title_stream()).  The decoder follows the encoder.
 */
#include "ngsc_oracle.h"
#include <stdlib.h>
#include <string.h>

#define MAX_FIELD_STAT_LEN 128u
#define TITLE_GROUP 32u
#define MAX_NUM_VAL_HUF 512

/* ---- MSB-first bit reader (bit_stream.h:92-147 GetBits/GetByte/GetWord, FlushInputWordBuffer) ---- */
typedef struct {
    const uint8_t *p;
    uint64_t nbits, pos; /* total bits, cursor */
    int bad;
} br_t;

static inline uint32_t br_bit(br_t *r)
{
    if (r->pos >= r->nbits) { r->bad = 1; return 0; }
    uint32_t b = (r->p[r->pos >> 3] >> (7 - (r->pos & 7))) & 1u;
    ++r->pos;
    return b;
}
static inline uint32_t br_bits(br_t *r, uint32_t n)
{
    uint32_t v = 0;
    if (r->pos + n > r->nbits) { r->bad = 1; r->pos = r->nbits; return 0; }
    while (n && (r->pos & 7)) { v = (v << 1) | br_bit(r); --n; }
    while (n >= 8) { v = (v << 8) | r->p[r->pos >> 3]; r->pos += 8; n -= 8; }
    while (n) { v = (v << 1) | br_bit(r); --n; }
    return v;
}
static inline void br_align(br_t *r) { r->pos = (r->pos + 7) & ~7ull; }
static inline uint32_t br_u8(br_t *r) { br_align(r); return br_bits(r, 8); }
static inline uint32_t br_u32(br_t *r) { br_align(r); return br_bits(r, 32); }

static uint32_t bit_length(uint64_t x)
{
    for (uint32_t i = 0; i < 32; ++i)
        if (x < (1ull << i)) return i;
    return 64;
}

/* ---- Huffman tree: pre-order blob -> child table (huffman.cpp:122-188) ---- */
typedef struct {
    uint32_t n, root, n_nodes;
    int32_t *lc, *rc; /* indexed by node id - n */
} tree_t;

static void tree_free(tree_t *t)
{
    if (!t) return;
    free(t->lc);
    free(t->rc);
    free(t);
}

/* StoreTree(BitStream&) huffman.cpp:191-205: align, u32 blob size, blob = root u32, n u32, min_len u8,
   pre-order bits (internal 0 | leaf 1 + id), zero padded. */
static tree_t *tree_load(br_t *r, uint32_t expect_n)
{
    uint32_t blen = br_u32(r);
    uint64_t end = r->pos + (uint64_t)blen * 8;
    uint32_t root = br_u32(r), n = br_u32(r);
    (void)br_u8(r);
    if (r->bad || n != expect_n || n == 0 || root >= 2 * n || end > r->nbits) { r->bad = 1; return NULL; }
    tree_t *t = (tree_t *)calloc(1, sizeof(tree_t));
    t->n = n;
    t->root = root;
    t->lc = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    t->rc = (int32_t *)malloc(sizeof(int32_t) * (n + 1));
    uint32_t bits_per_id = 0;
    while ((2u << bits_per_id) <= n) ++bits_per_id;
    if (n & (n - 1)) ++bits_per_id;
    /* Pre-order walk.  Internal ids are assigned by the encoder in merge order with the root being the
       largest; a node's id is not in the blob, only the shape, so ids are rebuilt top-down: the decoder
       needs only the shape, and numbers internal nodes in visit order. */
    int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (2 * n + 2)); /* slots waiting for a child: +idx = left of idx, -idx-1 = right */
    uint32_t made = 0;
    int sp = 0;
    int32_t rootref = -1;
    int first = 1;
    for (;;) {
        int32_t node;
        if (br_bit(r)) node = (int32_t)br_bits(r, bits_per_id);
        else {
            if (made >= n) { r->bad = 1; break; }
            node = (int32_t)(n + made++);
        }
        if (r->bad) break;
        if (first) { rootref = node; first = 0; }
        else {
            int32_t slot = stack[--sp];
            if (slot >= 0) t->lc[slot] = node;
            else t->rc[-slot - 1] = node;
        }
        if (node >= (int32_t)n) {
            int32_t idx = node - (int32_t)n;
            stack[sp++] = -idx - 1;
            stack[sp++] = idx;
        }
        if (!sp) break;
    }
    free(stack);
    t->root = (uint32_t)rootref;
    t->n_nodes = made;
    if (r->bad) { tree_free(t); return NULL; }
    r->pos = end;
    return t;
}

static inline uint32_t tree_get(const tree_t *t, br_t *r)
{
    int32_t node = (int32_t)t->root;
    while (node >= (int32_t)t->n) {
        int32_t idx = node - (int32_t)t->n;
        node = br_bit(r) ? t->rc[idx] : t->lc[idx];
        if (r->bad) return 0;
    }
    return (uint32_t)node;
}

/* ---- growable output ---- */
typedef struct {
    uint8_t *p;
    size_t n, cap;
} ob_t;
static void ob_put(ob_t *o, const uint8_t *d, size_t n)
{
    if (o->n + n > o->cap) {
        size_t nc = o->cap ? o->cap : (1u << 20);
        while (nc < o->n + n) nc *= 2;
        o->p = (uint8_t *)realloc(o->p, nc);
        o->cap = nc;
    }
    memcpy(o->p + o->n, d, n);
    o->n += n;
}

typedef struct {
    uint8_t sep;
    int is_const, is_num, is_len_const, delta_coding;
    uint32_t len0, min_len, max_len, bits_len, bits_num, bits_val;
    int32_t min_v, max_v, min_d, max_d;
    uint8_t *data0, *ham;
    tree_t *g;
    tree_t *l[MAX_FIELD_STAT_LEN + 1];
    /* running state */
    int32_t prev;
    uint8_t gflag;
    uint8_t *gval; /* group's first text value */
    uint32_t gval_len, gval_cap;
} dfield_t;

static uint32_t put_decimal(uint8_t *dst, uint32_t v)
{
    uint8_t tmp[12];
    uint32_t k = 0;
    do { tmp[k++] = (uint8_t)('0' + v % 10); v /= 10; } while (v);
    for (uint32_t i = 0; i < k; ++i) dst[i] = tmp[k - 1 - i];
    return k;
}

static const uint8_t AMB_BASE[17] = {0, 0, 'Y', 'R', 'W', 'S', 'K', 'M', 'D', 'V', 'H', 'B', 'N', 'X', 'U', '.', '-'};

int ngsc_o_decode_subblock(const uint8_t *sb, uint64_t len, uint8_t **fastq, uint64_t *fastq_len, uint64_t *consumed,
                           uint32_t *n_records)
{
    br_t R = {sb, len * 8, 0, 0}, *r = &R;
    int rc = NGSC_O_OK;
    dfield_t *F = NULL;
    uint32_t nf = 0, *qlen = NULL, *toff = NULL;
    tree_t **qt = NULL, *dt = NULL;
    uint8_t *qual = NULL;
    ob_t titles = {0, 0, 0}, out = {0, 0, 0};
    uint32_t max_q = 0;

    /* ---- info header phyNGSC.cpp:719-742 ---- */
    uint32_t n = br_u32(r);
    max_q = br_u32(r);
    (void)br_u32(r); /* max sequence length */
    uint32_t nsym = br_u8(r);
    (void)br_u8(r); /* QUALITY_PLAIN */
    uint32_t nqual = br_u8(r);
    uint32_t flags = br_u32(r);
    if (r->bad || n == 0 || n > NGSC_O_MAX_RECORDS + 1 || !nsym || !nqual || max_q > (1u << 24)) return NGSC_O_EFORMAT;
    qlen = (uint32_t *)malloc(sizeof(uint32_t) * n);
    {
        uint32_t b = bit_length(max_q);
        for (uint32_t i = 0; i < n; ++i) qlen[i] = br_bits(r, b);
        br_align(r);
    }

    /* ---- title header (encoder: ngsc_oracle.c title_stream(); cf. FetchTitleHeader tasks.cpp:625-735) ---- */
    nf = br_u32(r);
    if (r->bad || nf == 0 || nf > 4096) { rc = NGSC_O_EFORMAT; goto done; }
    F = (dfield_t *)calloc(nf, sizeof(dfield_t));
    uint32_t keep_from = 0; /* Q2b: value-coding trees exist only for fields >= the last vector reallocation */
    while (nf >= 2 && (keep_from ? keep_from * 2 : 1) <= nf - 1) keep_from = keep_from ? keep_from * 2 : 1;
    for (uint32_t f = 0; f < nf && !r->bad; ++f) {
        dfield_t *fd = &F[f];
        fd->sep = (uint8_t)br_u8(r);
        fd->is_const = br_u8(r) != 0;
        if (fd->is_const) {
            fd->len0 = br_u32(r);
            if (r->bad || (uint64_t)fd->len0 * 8 > r->nbits) { r->bad = 1; break; }
            fd->data0 = (uint8_t *)malloc(fd->len0 + 1);
            br_align(r);
            for (uint32_t j = 0; j < fd->len0; ++j) fd->data0[j] = (uint8_t)br_bits(r, 8);
            continue;
        }
        fd->is_num = br_u8(r) != 0;
        if (fd->is_num) {
            fd->min_v = (int32_t)br_u32(r);
            fd->max_v = (int32_t)br_u32(r);
            fd->min_d = (int32_t)br_u32(r);
            fd->max_d = (int32_t)br_u32(r);
            int64_t vr = (int64_t)fd->max_v - fd->min_v, dr = (int64_t)fd->max_d - fd->min_d;
            fd->delta_coding = !(vr < dr);
            int64_t diff = fd->delta_coding ? dr : vr;
            fd->bits_num = bit_length((uint64_t)diff);
            fd->bits_val = bit_length((uint64_t)vr);
            int have_map = fd->delta_coding ? (n >= 2u) : (f >= keep_from);
            if (diff + 1 <= MAX_NUM_VAL_HUF && have_map) fd->g = tree_load(r, (uint32_t)(diff + 1));
            continue;
        }
        fd->is_len_const = br_u8(r) != 0;
        fd->len0 = br_u32(r);
        fd->max_len = br_u32(r);
        fd->min_len = br_u32(r);
        if (r->bad || (uint64_t)fd->len0 * 8 > r->nbits || fd->max_len > (1u << 24)) { r->bad = 1; break; }
        fd->bits_len = bit_length((uint64_t)(fd->max_len - fd->min_len));
        fd->data0 = (uint8_t *)malloc(fd->len0 + 1);
        fd->ham = (uint8_t *)malloc(fd->len0 + 1);
        br_align(r);
        for (uint32_t j = 0; j < fd->len0; ++j) fd->data0[j] = (uint8_t)br_bits(r, 8);
        for (uint32_t j = 0; j < fd->len0; ++j) fd->ham[j] = (uint8_t)br_bit(r);
        uint32_t npos = fd->max_len < MAX_FIELD_STAT_LEN ? fd->max_len : MAX_FIELD_STAT_LEN;
        for (uint32_t j = 0; j < npos && !r->bad; ++j)
            if (j >= fd->len0 || !fd->ham[j]) fd->l[j] = tree_load(r, 256);
        if (fd->max_len >= MAX_FIELD_STAT_LEN && !r->bad) fd->l[MAX_FIELD_STAT_LEN] = tree_load(r, 256);
        br_align(r);
        fd->gval_cap = fd->max_len + 1;
        fd->gval = (uint8_t *)malloc(fd->gval_cap);
    }
    if (r->bad) { rc = NGSC_O_EFORMAT; goto done; }

    /* ---- title body (cf. FetchTitleBody tasks.cpp:738-954): groups of 32 records ---- */
    toff = (uint32_t *)malloc(sizeof(uint32_t) * ((size_t)n + 1));
    {
        uint8_t num[16];
        uint8_t *tmp = NULL;
        uint32_t tmp_cap = 0;
        for (uint32_t g0 = 0; g0 < n && !r->bad; g0 += TITLE_GROUP) {
            uint32_t g1 = g0 + TITLE_GROUP < n ? g0 + TITLE_GROUP : n;
            for (uint32_t f = 0; f < nf; ++f)
                if (!F[f].is_const) { F[f].gflag = (uint8_t)br_bit(r); F[f].prev = 0; }
            for (uint32_t i = g0; i < g1 && !r->bad; ++i) {
                toff[i] = (uint32_t)titles.n;
                for (uint32_t f = 0; f < nf; ++f) {
                    dfield_t *fd = &F[f];
                    if (fd->is_const) ob_put(&titles, fd->data0, fd->len0);
                    else if (fd->is_num) {
                        int32_t v;
                        if (i == g0) v = fd->min_v + (int32_t)br_bits(r, fd->bits_val > 32 ? 32 : fd->bits_val);
                        else if (fd->gflag) v = fd->delta_coding ? fd->prev + fd->min_d : fd->prev;
                        else {
                            int32_t x = fd->g ? (int32_t)tree_get(fd->g, r)
                                              : (int32_t)br_bits(r, fd->bits_num > 32 ? 32 : fd->bits_num);
                            v = fd->delta_coding ? fd->prev + fd->min_d + x : fd->min_v + x;
                        }
                        fd->prev = v;
                        ob_put(&titles, num, put_decimal(num, (uint32_t)v));
                    } else if (i != g0 && fd->gflag) ob_put(&titles, fd->gval, fd->gval_len);
                    else {
                        uint32_t L = fd->is_len_const ? fd->len0 : fd->min_len + br_bits(r, fd->bits_len);
                        if (L > fd->max_len) { r->bad = 1; break; }
                        if (L + 1 > tmp_cap) { tmp_cap = 2 * L + 64; tmp = (uint8_t *)realloc(tmp, tmp_cap); }
                        for (uint32_t j = 0; j < L; ++j) {
                            if (j < fd->len0 && fd->ham[j]) tmp[j] = fd->data0[j];
                            else {
                                const tree_t *t = fd->l[j < MAX_FIELD_STAT_LEN ? j : MAX_FIELD_STAT_LEN];
                                if (!t) { r->bad = 1; break; }
                                tmp[j] = (uint8_t)tree_get(t, r);
                            }
                        }
                        ob_put(&titles, tmp, L);
                        if (i == g0) { memcpy(fd->gval, tmp, L); fd->gval_len = L; }
                    }
                    ob_put(&titles, &fd->sep, 1);
                }
            }
            br_align(r);
        }
        free(tmp);
        toff[n] = (uint32_t)titles.n;
    }
    if (r->bad) { rc = NGSC_O_EFORMAT; goto done; }

    /* ---- quality (cf. FetchQuality tasks.cpp:1036-1101): alphabet, max_q+1 trees, per-position codes ---- */
    {
        uint8_t qsym[256];
        br_align(r);
        for (uint32_t i = 0; i < nqual; ++i) qsym[i] = (uint8_t)br_bits(r, 8);
        qt = (tree_t **)calloc((size_t)max_q + 1, sizeof(tree_t *));
        for (uint32_t p = 0; p <= max_q && !r->bad; ++p) qt[p] = tree_load(r, nqual);
        br_align(r);
        if (r->bad) { rc = NGSC_O_EFORMAT; goto done; }
        uint64_t tot = 0;
        for (uint32_t i = 0; i < n; ++i) {
            if (qlen[i] > max_q) { rc = NGSC_O_EFORMAT; goto done; }
            tot += qlen[i];
        }
        qual = (uint8_t *)malloc(tot + 1);
        uint64_t k = 0;
        for (uint32_t i = 0; i < n && !r->bad; ++i)
            for (uint32_t j = 0; j < qlen[i]; ++j) qual[k++] = qsym[tree_get(qt[j + 1], r)];
        br_align(r);
    }
    if (r->bad) { rc = NGSC_O_EFORMAT; goto done; }

    /* ---- DNA (cf. FetchDNA tasks.cpp:957-1033) + inverse of the N->quality transfer phyNGSC.cpp:573-588 ---- */
    {
        uint8_t dsym[256];
        for (uint32_t i = 0; i < nsym; ++i) dsym[i] = (uint8_t)br_bits(r, 8);
        const int plain = (flags & 2u) != 0; /* FLAG_DNA_PLAIN defs.h:71 */
        if (!plain) dt = tree_load(r, nsym);
        if (r->bad) { rc = NGSC_O_EFORMAT; goto done; }
        uint64_t k = 0;
        uint8_t *line = (uint8_t *)malloc((size_t)max_q + 2);
        for (uint32_t i = 0; i < n && !r->bad; ++i) {
            const uint32_t L = qlen[i];
            ob_put(&out, titles.p + toff[i], toff[i + 1] - toff[i]); /* ends with the field separator '\n' */
            uint8_t *q = qual + k;
            for (uint32_t j = 0; j < L; ++j) {
                if (q[j] >= 128) { /* a transferred ambiguity code: base and quality packed into one byte */
                    uint32_t v = (uint32_t)q[j] - 128u + 16u;
                    uint32_t a = v >> 3;
                    line[j] = a <= 16 ? AMB_BASE[a] : 0;
                    q[j] = (uint8_t)(33 + (v & 7));
                    if (!line[j]) r->bad = 1;
                } else {
                    uint32_t c = plain ? br_bits(r, 2) : tree_get(dt, r);
                    if (c >= nsym) { r->bad = 1; break; }
                    line[j] = dsym[c];
                }
            }
            line[L] = '\n';
            ob_put(&out, line, L + 1);
            ob_put(&out, (const uint8_t *)"+\n", 2);
            ob_put(&out, q, L);
            ob_put(&out, (const uint8_t *)"\n", 1);
            k += L;
        }
        free(line);
        br_align(r);
    }
    if (r->bad) { rc = NGSC_O_EFORMAT; goto done; }
    *fastq = out.p;
    *fastq_len = out.n;
    out.p = NULL;
    if (consumed) *consumed = r->pos >> 3;
    if (n_records) *n_records = n;

done:
    if (F)
        for (uint32_t f = 0; f < nf; ++f) {
            free(F[f].data0); free(F[f].ham); free(F[f].gval);
            tree_free(F[f].g);
            for (uint32_t j = 0; j <= MAX_FIELD_STAT_LEN; ++j) tree_free(F[f].l[j]);
        }
    free(F);
    if (qt) for (uint32_t p = 0; p <= max_q; ++p) tree_free(qt[p]);
    free(qt);
    tree_free(dt);
    free(qlen); free(toff); free(qual); free(titles.p); free(out.p);
    return rc;
}
