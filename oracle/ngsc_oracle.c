/*
 * ngsc_oracle.c - CPU restatement of the phyNGSC FASTQ sub-block compressor.
 *
 * TEST INFRASTRUCTURE ONLY (see ngsc_oracle.h).  The product path (phyngsc_amd/csrc, HIP) never
 * links, loads or calls this file.
 *
 * It is a restatement, not a copy: the reference's std::map statistics are replaced by two-pass
 * array statistics, its STL-heap Huffman by a sort + two-queue merge (equivalent because the
 * reference's heap order (frequency, symbol) is a strict total order, huffman.h:57-60), its
 * BitStream/BitMemory classes by one MSB-first bit writer (bit_stream.h:149-200 semantics:
 * big-endian bit string, explicit zero padding to the next byte), and the MPI rank loop by a
 * function that walks "virtual ranks".  Citations are file:line into the reference repository.
 *
 * Build: gcc -O2 -fopenmp -shared -fPIC -o libngsc_oracle.so ngsc_oracle.c -lm
 */
#include "ngsc_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>

/* Varying numeric title fields whose digits exceed 32 bits (and whose wrapped value stays below 2^31, so that the reference
   carries on): the reference stores the value modulo 2^32 - its container does not hold its input.  The restatement follows
   the reference; this count lets the tests tell such inputs apart (the HIP engine refuses them with NGSC_ERR_RANGE). */
static uint64_t g_lossy_values = 0;
uint64_t ngsc_o_lossy_count(int reset)
{
    uint64_t v;
#pragma omp atomic read
    v = g_lossy_values;
    if (reset) {
#pragma omp atomic write
        g_lossy_values = 0;
    }
    return v;
}
#endif

/* ------------------------------------------------------------------------------------------------
 * MSB-first bit writer (bit_stream.h:80-277; bit_memory.h:149-348 has the same semantics)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t *p;
    size_t n, cap;
    uint64_t acc;
    uint32_t nacc;
    int oom;
} bw_t;

static void bw_init(bw_t *w, size_t cap)
{
    w->p = (uint8_t *)malloc(cap ? cap : 64);
    w->n = 0;
    w->cap = cap ? cap : 64;
    w->acc = 0;
    w->nacc = 0;
    w->oom = w->p == NULL;
}

static inline void bw_raw(bw_t *w, uint8_t b)
{
    if (w->n == w->cap) {
        size_t nc = w->cap * 2;
        uint8_t *q = (uint8_t *)realloc(w->p, nc);
        if (!q) { w->oom = 1; return; }
        w->p = q;
        w->cap = nc;
    }
    w->p[w->n++] = b;
}

/* PutBits / Put2Bits / PutBit: append the low n bits of v, most significant first (bit_stream.h:149-169). */
static inline void bw_bits(bw_t *w, uint32_t v, uint32_t n)
{
    if (!n) return;
    if (n < 32) v &= (1u << n) - 1u;
    w->acc = (w->acc << n) | v;
    w->nacc += n;
    while (w->nacc >= 8) {
        bw_raw(w, (uint8_t)(w->acc >> (w->nacc - 8)));
        w->nacc -= 8;
    }
    w->acc &= (1ull << w->nacc) - 1ull;
}

/* FlushPartialWordBuffer: zero-pad to the next byte (bit_stream.h:183-200). */
static inline void bw_align(bw_t *w)
{
    if (w->nacc) {
        bw_raw(w, (uint8_t)(w->acc << (8 - w->nacc)));
        w->nacc = 0;
        w->acc = 0;
    }
}

/* PutByte / PutWord / PutBytes bypass the accumulator in the reference (bit_stream.h:211-265);
   every call site is byte aligned, which we enforce by aligning first. */
static inline void bw_u8(bw_t *w, uint8_t b) { bw_align(w); bw_raw(w, b); }
static inline void bw_u32(bw_t *w, uint32_t v)
{
    bw_align(w);
    bw_raw(w, (uint8_t)(v >> 24));
    bw_raw(w, (uint8_t)(v >> 16));
    bw_raw(w, (uint8_t)(v >> 8));
    bw_raw(w, (uint8_t)v);
}
static inline void bw_bytes(bw_t *w, const uint8_t *d, size_t n)
{
    bw_align(w);
    for (size_t i = 0; i < n; ++i) bw_raw(w, d[i]);
}

/* BitStream::BitLength (bit_stream.h:268-277): smallest i<32 with x < 2^i, else 64. */
static uint32_t bit_length(uint64_t x)
{
    for (uint32_t i = 0; i < 32; ++i)
        if (x < (1ull << i)) return i;
    return 64;
}

/* floor(log2(x)) + 1 for x >= 1, as computed with doubles in phyNGSC.cpp:844, tasks.cpp:1108-1118 */
static uint32_t floor_log2_plus1(uint64_t x)
{
    uint32_t b = 0;
    while (x) { ++b; x >>= 1; }
    return b;
}

/* ceil(log2(x)) for x >= 1 (phyNGSC.cpp:163, tasks.cpp:1156) */
static uint32_t ceil_log2(uint64_t x)
{
    uint32_t b = 0;
    while ((1ull << b) < x) ++b;
    return b;
}

/* ------------------------------------------------------------------------------------------------
 * Huffman (huffman.cpp:18-118, huffman.h:134-147, 176-186)
 * ---------------------------------------------------------------------------------------------- */
static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

int ngsc_o_huffman(const uint32_t *freq, uint32_t n, ngsc_o_code *codes, uint8_t *blob, uint32_t *blob_len)
{
    if (!n) return -1;
    uint64_t *key = (uint64_t *)malloc(sizeof(uint64_t) * n);
    int32_t *lc = (int32_t *)malloc(sizeof(int32_t) * 2 * n);
    int32_t *rc = (int32_t *)malloc(sizeof(int32_t) * 2 * n);
    uint32_t *ifreq = (uint32_t *)malloc(sizeof(uint32_t) * n);
    /* heap order == ascending (frequency, symbol): huffman.h:57-60 (operator< inverted for a max-heap) */
    for (uint32_t i = 0; i < n; ++i) key[i] = ((uint64_t)freq[i] << 32) | i;
    qsort(key, n, sizeof(uint64_t), cmp_u64);
    /* compact: pop zero-frequency symbols while more than 2 remain (huffman.cpp:44-50) */
    uint32_t zeros = 0;
    while (zeros < n && (key[zeros] >> 32) == 0) ++zeros;
    uint32_t drop = n > 2 ? (zeros < n - 2 ? zeros : n - 2) : 0;
    uint32_t present = n - drop;
    for (uint32_t i = 0; i < 2 * n - 1; ++i) { codes[i].code = 0; codes[i].len = 0; }
    /* merge loop huffman.cpp:57-70 as a two-queue merge; ties go to the smaller node id, and a leaf
       id is always smaller than an internal id (n_symbols+i), exactly as the heap would order them */
    uint32_t lq = drop, iq = 0, made = 0;
    for (uint32_t k = 0; k + 1 < present; ++k) {
        uint32_t f[2];
        int32_t id[2];
        for (int s = 0; s < 2; ++s) {
            int take_leaf;
            if (lq < n && iq < made) take_leaf = (uint32_t)(key[lq] >> 32) <= ifreq[iq];
            else take_leaf = lq < n;
            if (take_leaf) { f[s] = (uint32_t)(key[lq] >> 32); id[s] = (int32_t)(key[lq] & 0xffffffffu); ++lq; }
            else { f[s] = ifreq[iq]; id[s] = (int32_t)(n + iq); ++iq; }
        }
        ifreq[made] = f[0] + f[1];
        lc[n + made] = id[0];
        rc[n + made] = id[1];
        ++made;
    }
    int root = (int)(n + present) - 2;
    /* code assignment huffman.cpp:73-79: walk internal nodes from the root downwards */
    for (int i = root; i >= (int)n; --i) {
        codes[lc[i]].len = codes[i].len + 1;
        codes[lc[i]].code = codes[i].code << 1;
        codes[rc[i]].len = codes[i].len + 1;
        codes[rc[i]].code = (codes[i].code << 1) | 1;
    }
    if (blob) {
        /* StoreTree(mem,len) huffman.cpp:88-118 */
        uint32_t bits_per_id = 0;
        while ((2u << bits_per_id) <= n) ++bits_per_id; /* utils::int_log(n,2), utils.h:91-104 */
        if (n & (n - 1)) ++bits_per_id;
        uint32_t min_len = n;
        for (uint32_t i = 0; i < n; ++i)
            if (codes[i].len < min_len && codes[i].len > 0) min_len = codes[i].len;
        bw_t w;
        bw_init(&w, 64);
        bw_u32(&w, (uint32_t)root);
        bw_u32(&w, n);
        bw_u8(&w, (uint8_t)min_len);
        /* EncodeProcess huffman.h:134-147: pre-order, leaf = 1 + id, internal = 0 */
        int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (2 * n + 2));
        int sp = 0;
        stack[sp++] = root;
        while (sp) {
            int32_t node = stack[--sp];
            if (node < (int32_t)n) {
                bw_bits(&w, 1, 1);
                bw_bits(&w, (uint32_t)node, bits_per_id);
            } else {
                bw_bits(&w, 0, 1);
                stack[sp++] = rc[node];
                stack[sp++] = lc[node];
            }
        }
        bw_align(&w);
        free(stack);
        memcpy(blob, w.p, w.n);
        *blob_len = (uint32_t)w.n;
        free(w.p);
    }
    free(key); free(lc); free(rc); free(ifreq);
    return root;
}

/* HuffmanEncoder::StoreTree(BitStream&, ...) huffman.cpp:191-205: align, u32 size, blob */
static void put_tree(bw_t *w, const uint32_t *freq, uint32_t n, ngsc_o_code *codes)
{
    uint8_t *blob = (uint8_t *)malloc(32 + 2 * (size_t)n * 2);
    uint32_t blen = 0;
    ngsc_o_huffman(freq, n, codes, blob, &blen);
    bw_align(w);
    bw_u32(w, blen);
    bw_bytes(w, blob, blen);
    free(blob);
}

/* ------------------------------------------------------------------------------------------------
 * Record index (phyNGSC.cpp:254-331, behaviour with one thread; output is thread-count invariant)
 * ---------------------------------------------------------------------------------------------- */
static int64_t find_nl(const uint8_t *b, uint64_t from, uint64_t limit)
{
    for (uint64_t i = from; i < limit; ++i)
        if (b[i] == '\n') return (int64_t)i;
    return -1;
}

int ngsc_o_record_index(const uint8_t *win, uint64_t avail, uint64_t wsize, uint64_t rec_start, uint32_t overlap,
                        uint32_t *title_end, uint32_t *seq_end, uint32_t *n_records, uint64_t *next_offset)
{
    uint32_t n = 0;
    /* first record: the reference looks for "\n+\n" after the second newline (:276-293) */
    int64_t te = find_nl(win, rec_start, avail);
    if (te < 0) return NGSC_O_EFORMAT;
    int64_t se = find_nl(win, (uint64_t)te + 1, avail);
    if (se < 0 || (uint64_t)se + 2 >= avail) return NGSC_O_EFORMAT;
    if (win[se + 1] != '+' || win[se + 2] != '\n') return NGSC_O_EFORMAT;
    title_end[n] = (uint32_t)te;
    seq_end[n] = (uint32_t)se;
    ++n;
    uint64_t next = (uint64_t)se + (uint64_t)(se - te) + 3; /* :296 */
    /* NOTE: no overlap test after the first record (:295-300 falls through to :303) */
    uint64_t pos = next;
    int64_t stop_at = (int64_t)wsize - (int64_t)overlap;
    for (;;) {
        uint64_t lim = wsize < avail ? wsize : avail;
        te = pos < lim ? find_nl(win, pos, lim) : -1; /* for-loop bound i < th_i_end (:268) */
        if (te < 0) break;
        se = find_nl(win, (uint64_t)te + 1, avail); /* :308 has no bound: bytes past the window */
        if (se < 0) break;
        if (n > NGSC_O_MAX_RECORDS) return NGSC_O_ETOOMANY; /* :321-326 (reference overflows its array here) */
        title_end[n] = (uint32_t)te;
        seq_end[n] = (uint32_t)se;
        ++n;
        next = (uint64_t)se + (uint64_t)(se - te) + 3; /* :312 */
        if ((int64_t)next >= stop_at) break;             /* :315-318 */
        pos = next + 1;                                  /* for-loop ++i */
    }
    if (n > NGSC_O_MAX_RECORDS) return NGSC_O_ETOOMANY;
    *n_records = n;
    *next_offset = (uint64_t)seq_end[n - 1] * 2 - title_end[n - 1] + 3; /* :745 */
    return NGSC_O_OK;
}

/* phyNGSC.cpp:131-156 */
int64_t ngsc_o_first_record(const uint8_t *buf, uint64_t avail)
{
    uint64_t i = 0;
    while (i < avail && buf[i] != '@') ++i;
    if (i >= avail) return -1;
    uint64_t first_at = i;
    while (i < avail && buf[i] != '\n') ++i;
    if (i >= avail) return -1;
    if (i + 1 < avail && buf[i + 1] == '@') return (int64_t)(i + 1);
    return (int64_t)first_at;
}

/* ------------------------------------------------------------------------------------------------
 * Title analysis + store (phyNGSC.cpp:342-423, tasks.cpp:22-223, tasks.cpp:289-510)
 * ---------------------------------------------------------------------------------------------- */
#define MAX_FIELD_STAT_LEN 128u
#define TITLE_GROUP 32u
#define MAX_NUM_VAL_HUF 512u

static inline int is_sep(uint8_t c)
{
    /* " ._,=:/-#\n"  phyNGSC.cpp:208 */
    switch (c) {
    case ' ': case '.': case '_': case ',': case '=': case ':': case '/': case '-': case '#': case '\n':
        return 1;
    default:
        return 0;
    }
}

/* utils::is_num utils.h:107-114 */
static int field_is_num(const uint8_t *s, uint32_t len)
{
    for (uint32_t i = 0; i < len; ++i)
        if (s[i] < '0' || s[i] > '9') return 0;
    return (len == 1) || (len > 1 && s[0] != '0');
}

typedef struct {
    uint32_t len0;
    const uint8_t *data0;
    uint8_t sep;
    uint32_t min_len, max_len;
    int is_const, is_len_const, is_num;
    int32_t min_v, max_v, min_d, max_d;
    int delta_coding;
    uint32_t bits_num, bits_val, bits_len;
    uint8_t *ham; /* Hamming mask, len0 entries */
    ngsc_o_code *gcodes;           /* numeric Huffman (Huffman_global) or NULL */
    ngsc_o_code *lcodes[MAX_FIELD_STAT_LEN + 1]; /* per-position Huffman (Huffman_local) or NULL */
} field_t;

typedef struct {
    const uint8_t *buf;
    uint32_t n, nf;
    const uint32_t *te, *se; /* title_end / seq_end per record */
    uint32_t *ls;            /* line start per record */
    uint32_t *fe;            /* fields_end_pos[r*nf+f] */
} titles_t;

static inline void field_span(const titles_t *t, uint32_t r, uint32_t f, uint32_t *s, uint32_t *e)
{
    *s = f ? t->fe[(size_t)r * t->nf + f - 1] + 1 : t->ls[r];
    *e = t->fe[(size_t)r * t->nf + f];
}

static int title_stream(const titles_t *t, bw_t *w)
{
    const uint32_t n = t->n, nf = t->nf;
    const uint8_t *buf = t->buf;
    field_t *F = (field_t *)calloc(nf, sizeof(field_t));
    int32_t *val = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    uint32_t *hist = NULL; /* [129][256] per text field, allocated on demand */
    int rc = NGSC_O_OK;

    /* ---- analysis: one field at a time, like AnalyzeTitleFields(c_field) tasks.cpp:22-223 ---- */
    for (uint32_t f = 0; f < nf && rc == NGSC_O_OK; ++f) {
        field_t *fd = &F[f];
        uint32_t s0, e0;
        field_span(t, 0, f, &s0, &e0);
        fd->len0 = e0 - s0; /* field init from record 0: phyNGSC.cpp:345-379 */
        fd->data0 = buf + s0;
        fd->sep = buf[e0];
        fd->min_len = fd->max_len = fd->len0;
        fd->is_const = 1;
        fd->is_len_const = 1;
        fd->is_num = 1;
        fd->ham = (uint8_t *)malloc(fd->len0 + 1);
        memset(fd->ham, 1, fd->len0 + 1);
        for (uint32_t r = 0; r < n; ++r) {
            uint32_t s, e;
            field_span(t, r, f, &s, &e);
            uint32_t len = e - s;
            if (len > fd->max_len) fd->max_len = len;       /* tasks.cpp:54-61 */
            if (len < fd->min_len) fd->min_len = len;
            if (len != fd->len0) { fd->is_const = 0; fd->is_len_const = 0; } /* :95-110 */
            else if (fd->is_const && memcmp(fd->data0, buf + s, len) != 0) fd->is_const = 0;
            if (fd->is_num && !field_is_num(buf + s, len)) fd->is_num = 0; /* :112-115, running AND */
            uint32_t m = len < fd->len0 ? len : fd->len0; /* Hamming mask :187-193 */
            for (uint32_t p = 0; p < m; ++p)
                if (fd->data0[p] != buf[s + p]) fd->ham[p] = 0;
        }
        if (fd->is_num) {
            /* numeric statistics tasks.cpp:117-184; utils::to_num is uint32 stored in int32 (Q6) */
            for (uint32_t r = 0; r < n; ++r) {
                uint32_t s, e;
                field_span(t, r, f, &s, &e);
                uint64_t v = 0;
                int wrapped = 0; /* the digits say more than 32 bits hold: the reference goes on with the value modulo 2^32 */
                for (uint32_t p = s; p < e; ++p) {
                    v = v * 10 + (uint64_t)(buf[p] - '0');
                    if (v >> 32) wrapped = 1;
                    v &= 0xffffffffull; /* uint32 arithmetic, utils.h:117-125 */
                }
                const int big = v > 0x7fffffffull; /* the value the reference goes on with is a negative int32 */
                if (wrapped && !big && !fd->is_const) {
                    /* diagnostic only (ngsc_o_lossy_count): the reference writes a container that does not hold this title */
#pragma omp atomic
                    g_lossy_values++;
                }
                /* a constant field is stored once as text and its (wrapped) value is never used; a varying
                   one with a value the int32 cannot hold sends the reference into negative ranges */
                if (big && !fd->is_const) { rc = NGSC_O_ERANGE; break; }
                val[r] = (int32_t)(uint32_t)v;
            }
            if (rc) break;
            fd->min_v = fd->max_v = val[0];
            fd->min_d = 1;  /* Field() defaults structures.h:103-106 */
            fd->max_d = -1;
            for (uint32_t r = 0; r < n; ++r) {
                if (val[r] < fd->min_v) fd->min_v = val[r];
                if (val[r] > fd->max_v) fd->max_v = val[r];
                if (r >= 1) {
                    int32_t d = val[r] - val[r - 1];
                    if (r == 1) { fd->min_d = fd->max_d = d; }
                    else {
                        if (d > fd->max_d) fd->max_d = d;
                        if (d < fd->min_d) fd->min_d = d;
                    }
                }
            }
            /* tasks.cpp:205-222 */
            int64_t vr = (int64_t)fd->max_v - fd->min_v, dr = (int64_t)fd->max_d - fd->min_d;
            if (dr > 0x7fffffffll) { rc = NGSC_O_ERANGE; break; }
            fd->delta_coding = !(vr < dr);
            int64_t diff = fd->delta_coding ? dr : vr;
            /* BitLength takes uint64: a negative int32 sign-extends and yields 64 (only when n==1,
               where every field is constant and the widths are never used) */
            fd->bits_num = bit_length((uint64_t)diff);
            fd->bits_val = bit_length((uint64_t)vr);
        } else if (!fd->is_const) {
            fd->bits_len = bit_length((uint64_t)(fd->max_len - fd->min_len)); /* tasks.cpp:202 */
        }
    }
    if (rc) goto done;

    /* ---- header tasks.cpp:302-390 ---- */
    bw_u32(w, nf);
    for (uint32_t f = 0; f < nf; ++f) {
        field_t *fd = &F[f];
        bw_u8(w, fd->sep);
        bw_u8(w, (uint8_t)fd->is_const);
        if (fd->is_const) {
            bw_u32(w, fd->len0);
            bw_bytes(w, fd->data0, fd->len0);
            continue;
        }
        bw_u8(w, (uint8_t)fd->is_num);
        if (fd->is_num) {
            bw_u32(w, (uint32_t)fd->min_v);
            bw_u32(w, (uint32_t)fd->max_v);
            bw_u32(w, (uint32_t)fd->min_d);
            bw_u32(w, (uint32_t)fd->max_d);
            int64_t diff = fd->delta_coding ? (int64_t)fd->max_d - fd->min_d : (int64_t)fd->max_v - fd->min_v;
            diff += 1;
            /* tasks.cpp:338: Huffman iff diff <= 512 and the value map is non-empty.  The maps
               (num_values / delta_values, Q2/Q3) are complete whenever diff <= 512, so a second
               counting pass reproduces them; record 0's value is counted twice (Q2). */
            /* Q2b: `fields` is a std::vector<Field> filled by push_back (phyNGSC.cpp:350) and
               Field's copy constructor (structures.h:126-202) does not copy num_values, so the
               record-0 entry made at :368 is lost for every field pushed before the vector's last
               reallocation (libstdc++ doubles: at indices 1,2,4,8,...).  Those fields never count
               values (tasks.cpp:175 tests size()), their map stays empty and value coding gets no
               Huffman tree.  delta_values is filled after all pushes and is unaffected. */
            uint32_t keep_from = 0;
            while (nf >= 2 && keep_from * 2 <= nf - 1 && (keep_from ? keep_from * 2 : 1) <= nf - 1)
                keep_from = keep_from ? keep_from * 2 : 1;
            int have_map = fd->delta_coding ? (n >= 2u) : (f >= keep_from);
            if (diff <= (int64_t)MAX_NUM_VAL_HUF && have_map) {
                uint32_t cnt[MAX_NUM_VAL_HUF];
                memset(cnt, 0, sizeof(cnt));
                for (uint32_t r = 0; r < n; ++r) {
                    uint32_t s, e;
                    field_span(t, r, f, &s, &e);
                    uint32_t v = 0;
                    for (uint32_t p = s; p < e; ++p) v = v * 10 + (uint32_t)(buf[p] - '0');
                    val[r] = (int32_t)v;
                }
                if (fd->delta_coding) {
                    for (uint32_t r = 1; r < n; ++r) cnt[(val[r] - val[r - 1]) - fd->min_d]++;
                } else {
                    cnt[val[0] - fd->min_v]++; /* phyNGSC.cpp:368 */
                    for (uint32_t r = 0; r < n; ++r) cnt[val[r] - fd->min_v]++;
                }
                fd->gcodes = (ngsc_o_code *)malloc(sizeof(ngsc_o_code) * 2 * MAX_NUM_VAL_HUF);
                put_tree(w, cnt, (uint32_t)diff, fd->gcodes);
            }
            continue;
        }
        bw_u8(w, (uint8_t)fd->is_len_const);
        bw_u32(w, fd->len0);
        bw_u32(w, fd->max_len);
        bw_u32(w, fd->min_len);
        bw_bytes(w, fd->data0, fd->len0);
        for (uint32_t j = 0; j < fd->len0; ++j) bw_bits(w, fd->ham[j], 1);
        /* per-position character statistics tasks.cpp:83-93 */
        if (!hist) hist = (uint32_t *)malloc(sizeof(uint32_t) * (MAX_FIELD_STAT_LEN + 1) * 256);
        memset(hist, 0, sizeof(uint32_t) * (MAX_FIELD_STAT_LEN + 1) * 256);
        for (uint32_t r = 0; r < n; ++r) {
            uint32_t s, e;
            field_span(t, r, f, &s, &e);
            uint32_t len = e - s;
            for (uint32_t x = 0; x < len; ++x) {
                uint32_t row = x < MAX_FIELD_STAT_LEN ? x : MAX_FIELD_STAT_LEN;
                hist[row * 256 + buf[s + x]]++;
            }
        }
        uint32_t npos = fd->max_len < MAX_FIELD_STAT_LEN ? fd->max_len : MAX_FIELD_STAT_LEN;
        for (uint32_t j = 0; j < npos; ++j) {
            if (j >= fd->len0 || !fd->ham[j]) {
                fd->lcodes[j] = (ngsc_o_code *)malloc(sizeof(ngsc_o_code) * 511);
                put_tree(w, hist + j * 256, 256, fd->lcodes[j]);
            }
        }
        if (fd->max_len >= MAX_FIELD_STAT_LEN) {
            /* tasks.cpp:378-387. For max_len == 128 exactly the reference reads chars[128] of a
               128-entry vector (undefined); we use the all-zero histogram that slot would hold. */
            fd->lcodes[MAX_FIELD_STAT_LEN] = (ngsc_o_code *)malloc(sizeof(ngsc_o_code) * 511);
            put_tree(w, hist + MAX_FIELD_STAT_LEN * 256, 256, fd->lcodes[MAX_FIELD_STAT_LEN]);
        }
        bw_align(w);
    }

    /* ---- body tasks.cpp:393-509: groups of 32 records, byte aligned after each group ---- */
    {
        int32_t *prev = (int32_t *)calloc(nf, sizeof(int32_t));
        uint8_t *gflag = (uint8_t *)malloc(nf);
        for (uint32_t g0 = 0; g0 < n; g0 += TITLE_GROUP) {
            uint32_t g1 = g0 + TITLE_GROUP < n ? g0 + TITLE_GROUP : n;
            /* group flags (BlockDesc, tasks.cpp:63-81, 127-147, 406-424) */
            for (uint32_t f = 0; f < nf; ++f) {
                field_t *fd = &F[f];
                if (fd->is_const) continue;
                prev[f] = 0;
                uint32_t s0, e0;
                field_span(t, g0, f, &s0, &e0);
                int flag = 1;
                if (!fd->is_num) {
                    for (uint32_t r = g0 + 1; r < g1 && flag; ++r) {
                        uint32_t s, e;
                        field_span(t, r, f, &s, &e);
                        if (e - s != e0 - s0 || memcmp(buf + s, buf + s0, e - s) != 0) flag = 0;
                    }
                } else {
                    int32_t pv = 0, first_delta = 0;
                    int have_delta = 0, dconst = 1, vconst = 1;
                    int32_t v0 = 0;
                    for (uint32_t r = g0; r < g1; ++r) {
                        uint32_t s, e;
                        field_span(t, r, f, &s, &e);
                        uint32_t v = 0;
                        for (uint32_t p = s; p < e; ++p) v = v * 10 + (uint32_t)(buf[p] - '0');
                        if (r == g0) v0 = (int32_t)v;
                        else {
                            if ((int32_t)v != v0) vconst = 0;
                            int32_t d = (int32_t)v - pv;
                            if (!have_delta) { first_delta = d; have_delta = 1; }
                            else if (d != first_delta) dconst = 0;
                        }
                        pv = (int32_t)v;
                    }
                    /* :415 block_delta_constant (0 if the group has one record) must equal min_delta */
                    dconst = dconst && ((have_delta ? first_delta : 0) == fd->min_d);
                    flag = fd->delta_coding ? dconst : vconst;
                }
                gflag[f] = (uint8_t)flag;
                bw_bits(w, (uint32_t)flag, 1);
            }
            for (uint32_t r = g0; r < g1; ++r) {
                for (uint32_t f = 0; f < nf; ++f) {
                    field_t *fd = &F[f];
                    if (fd->is_const) continue;
                    uint32_t s, e;
                    field_span(t, r, f, &s, &e);
                    if (fd->is_num) {
                        uint32_t uv = 0;
                        for (uint32_t p = s; p < e; ++p) uv = uv * 10 + (uint32_t)(buf[p] - '0');
                        int32_t v = (int32_t)uv;
                        if (r == g0) {
                            if (fd->bits_val > 32) { rc = NGSC_O_ERANGE; goto body_done; }
                            bw_bits(w, (uint32_t)(v - fd->min_v), fd->bits_val); /* :457 */
                        } else if (!gflag[f]) {
                            int32_t to_store = fd->delta_coding ? v - prev[f] - fd->min_d : v - fd->min_v;
                            if (fd->gcodes) bw_bits(w, fd->gcodes[to_store].code, fd->gcodes[to_store].len);
                            else {
                                if (fd->bits_num > 32) { rc = NGSC_O_ERANGE; goto body_done; }
                                bw_bits(w, (uint32_t)to_store, fd->bits_num);
                            }
                        }
                        prev[f] = v;
                        continue;
                    }
                    if (r != g0 && gflag[f]) continue; /* :487 */
                    if (!fd->is_len_const) bw_bits(w, (e - s) - fd->min_len, fd->bits_len);
                    for (uint32_t j = 0; j < e - s; ++j) {
                        if (j >= fd->len0 || !fd->ham[j]) {
                            const ngsc_o_code *cd = fd->lcodes[j < MAX_FIELD_STAT_LEN ? j : MAX_FIELD_STAT_LEN];
                            uint8_t c = buf[s + j];
                            bw_bits(w, cd[c].code, cd[c].len);
                        }
                    }
                }
            }
            bw_align(w); /* :508 */
        }
    body_done:
        free(prev);
        free(gflag);
    }

done:
    for (uint32_t f = 0; f < nf; ++f) {
        free(F[f].ham);
        free(F[f].gcodes);
        for (uint32_t j = 0; j <= MAX_FIELD_STAT_LEN; ++j) free(F[f].lcodes[j]);
    }
    free(F);
    free(val);
    free(hist);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * One sub-block
 * ---------------------------------------------------------------------------------------------- */
enum { /* defs.h:69-80 */
    F_TRY_LZ = 1, F_DNA_PLAIN = 2, F_CONST_NUM_FIELDS = 4, F_PLUS_ONLY = 8, F_USE_DELTA = 16,
    F_DELTA_CONSTANT = 32, F_DELTA_NO_BEGIN_NUC = 64, F_VARIABLE_LENGTH = 128, F_LINE_BREAKS = 256
};

static uint8_t amb_code(uint8_t c)
{
    /* trans_amb_codes phyNGSC.cpp:184-206 */
    switch (c) {
    case 'A': case 'C': case 'G': case 'T': return 1;
    case 'Y': return 2; case 'R': return 3; case 'W': return 4; case 'S': return 5;
    case 'K': return 6; case 'M': return 7; case 'D': return 8; case 'V': return 9;
    case 'H': return 10; case 'B': return 11; case 'N': return 12; case 'X': return 13;
    case 'U': return 14; case '.': return 15; case '-': return 16;
    default: return 0;
    }
}

int ngsc_o_subblock(const uint8_t *win, uint64_t avail, uint64_t wsize, uint64_t rec_start, uint32_t overlap,
                    uint8_t **out, uint64_t *out_len, ngsc_o_sbinfo *info)
{
    int rc = NGSC_O_OK;
    uint8_t *buf = (uint8_t *)calloc(avail + 64, 1); /* private copy: the prepass rewrites it in place */
    uint32_t *te = (uint32_t *)malloc(sizeof(uint32_t) * (NGSC_O_MAX_RECORDS + 2));
    uint32_t *se = (uint32_t *)malloc(sizeof(uint32_t) * (NGSC_O_MAX_RECORDS + 2));
    uint32_t *ls = NULL, *fe = NULL, *slen = NULL, *qlen = NULL, *qhist = NULL;
    bw_t wi, wt, wq, wd;
    memset(&wi, 0, sizeof wi); memset(&wt, 0, sizeof wt); memset(&wq, 0, sizeof wq); memset(&wd, 0, sizeof wd);
    if (!buf || !te || !se) { rc = NGSC_O_ENOMEM; goto out; }
    memcpy(buf, win, avail);

    uint32_t n = 0;
    uint64_t next = 0;
    rc = ngsc_o_record_index(buf, avail, wsize, rec_start, overlap, te, se, &n, &next);
    if (rc) goto out;

    /* ---- title tokenisation phyNGSC.cpp:342-423 ---- */
    uint32_t nf = 0;
    for (uint64_t i = rec_start; i <= te[0]; ++i) nf += is_sep(buf[i]);
    ls = (uint32_t *)malloc(sizeof(uint32_t) * n);
    fe = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)n * nf);
    for (uint32_t r = 0; r < n; ++r) {
        uint32_t start = r ? se[r - 1] + (se[r - 1] - te[r - 1]) + 3 : (uint32_t)rec_start;
        if (r && (buf[start - 1] != '\n' || start > te[r])) { rc = NGSC_O_EFORMAT; goto out; }
        ls[r] = start;
        uint32_t k = 0;
        for (uint32_t p = start; p <= te[r]; ++p) {
            if (is_sep(buf[p])) {
                if (k == nf) { rc = NGSC_O_EFIELDS; goto out; }
                fe[(size_t)r * nf + k++] = p;
            }
        }
        if (k != nf) { rc = NGSC_O_EFIELDS; goto out; }
    }

    /* ---- DNA / quality prepass phyNGSC.cpp:462-653 ---- */
    /* Colour space (SOLiD reads: a primer base, then transitions '0'..'3', '.' = unknown), phyNGSC.cpp:473-547.  The first
       record of the sub-block decides (the reference decides per OpenMP thread: this is its one-thread behaviour, the
       one the golden files pin): colours when its second character is a digit, "no begin nucleotide" when its first
       one is.  While every read starts with the first read's primer base and first quality, reads are stored without
       them ("delta constant", translation = 1); the first read that differs ends that for the rest of the sub-block. */
    int cs_delta, cs_nobegin, cs_const = 1;
    uint8_t cs_seq0, cs_qua0;
    {
        uint32_t s0 = te[0] + 1;
        cs_seq0 = buf[s0];
        cs_qua0 = buf[se[0] + 3];
        cs_delta = buf[s0 + 1] >= '0' && buf[s0 + 1] <= '3';   /* :478 */
        cs_nobegin = cs_seq0 >= '0' && cs_seq0 <= '3';         /* :480-484 */
        if (cs_nobegin) cs_const = 0;
    }
    uint32_t dna_occ[256], qua_seen[256];
    memset(dna_occ, 0, sizeof dna_occ);
    memset(qua_seen, 0, sizeof qua_seen);
    slen = (uint32_t *)malloc(sizeof(uint32_t) * n);
    qlen = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint32_t max_q = 0, max_s = 0;
    for (uint32_t r = 0; r < n; ++r) {
        uint32_t s = te[r] + 1, e = se[r], L = e - s, q0 = se[r] + 3;
        if ((uint64_t)q0 + L > avail) { rc = NGSC_O_EFORMAT; goto out; } /* quality line cut off by the end of the data */
        const uint32_t q0_orig = q0;
        uint32_t ql = L;   /* quality symbols of the record */
        if (cs_delta && !cs_nobegin && cs_const) cs_const = buf[s] == cs_seq0 && buf[q0] == cs_qua0;   /* :489-493 */
        if (cs_delta) {
            /* :495-546.  A transition c turns base b into b ^ c on the two-bit scale A, C, G, T; '.' and '/' give 'N'
               and leave the running base alone.  The decoded base goes to the read, shifted left by one when the primer
               base is dropped; the "quality" position gets the read's own byte - the colour when shifted, else the base
               just stored (the reference copies from the sequence line here, :531) - so only a kept first quality
               survives. */
            static const char base_of[4] = {'A', 'C', 'G', 'T'};
            const uint32_t tr = (uint32_t)cs_const;
            if (L < tr) { rc = NGSC_O_EFORMAT; goto out; }   /* an empty read: the reference would write in front of it */
            uint32_t cur = 0;   /* 'A' */
            uint8_t sym = 'A';
            uint32_t k = 0;
            if (!cs_nobegin) {
                sym = buf[s];
                dna_occ[sym]++;
                k = 1;
            }
            for (; k < L; ++k) {
                if (sym == 'A') cur = 0; else if (sym == 'C') cur = 1; else if (sym == 'G') cur = 2; else if (sym == 'T') cur = 3;
                const uint32_t c = (uint32_t)buf[s + k] - '.';
                if (c > 5) { rc = NGSC_O_EFORMAT; goto out; }   /* not a colour: the reference indexes past its table */
                sym = c < 2 ? 'N' : (uint8_t)base_of[cur ^ (c - 2)];
                dna_occ[sym]++;
                const uint8_t colour = buf[s + k];
                buf[s + k - tr] = sym;
                buf[q0 + k - tr] = tr ? colour : sym;
            }
            buf[e - tr] = 0;
            buf[q0 + L - tr] = 0;
            e -= tr;
            ql -= tr;
            L = e - s;
        }
        (void)q0_orig;
        /* transfer test :549-571: every non-ACGT IUPAC base must have quality '!'..'(' */
        int any = 0, ok = 1;
        for (uint32_t j = 0; j < L; ++j) {
            uint8_t a = amb_code(buf[s + j]);
            if (a == 1) continue;
            if (a == 0) { ok = 0; break; }
            uint8_t q = buf[q0 + j];
            if (q < 33 || q > 40) { ok = 0; break; }
            any = 1;
        }
        uint32_t newlen = L;
        if (any && ok) { /* :573-588 */
            uint32_t k = s;
            for (uint32_t j = 0; j < L; ++j) {
                uint8_t a = amb_code(buf[s + j]);
                if (a > 1) buf[q0 + j] = (uint8_t)(128 + (a << 3) - 16 + (buf[q0 + j] - 33));
                else buf[k++] = buf[s + j];
            }
            buf[k] = 0;
            newlen = k - s;
        }
        if (!cs_delta)
            for (uint32_t j = 0; j < newlen; ++j) dna_occ[buf[s + j]]++;   /* :593-599 (colour space counted while decoding) */
        for (uint32_t j = 0; j < ql; ++j) qua_seen[buf[q0 + j]] = 1;   /* :614-615 */
        if (ql > max_q) max_q = ql;
        if (newlen > max_s) max_s = newlen;
        slen[r] = newlen;
        qlen[r] = ql;
    }
    /* flags: :433-435, :645-667.  The min-length reduction never lowers its UINT_MAX initial
       value (:633-640), so FLAG_VARIABLE_LENGTH is always set; DELTA_CONSTANT stays set (Q1). */
    uint32_t flags = F_PLUS_ONLY | F_DNA_PLAIN | F_CONST_NUM_FIELDS | F_VARIABLE_LENGTH;
    if (cs_const) flags |= F_DELTA_CONSTANT;
    if (cs_delta) flags |= F_USE_DELTA;
    if (cs_nobegin) flags |= F_DELTA_NO_BEGIN_NUC;
    uint8_t symbols[256], qualities[256], sym_code[256], qua_code[256];
    uint32_t nsym = 0, nqual = 0;
    memset(sym_code, 0, 256);
    memset(qua_code, 0, 256);
    for (uint32_t c = 0; c < 256; ++c) { /* :674-686 */
        if (dna_occ[c]) { symbols[nsym] = (uint8_t)c; sym_code[c] = (uint8_t)nsym++; }
        if (qua_seen[c]) { qualities[nqual] = (uint8_t)c; qua_code[c] = (uint8_t)nqual++; }
    }
    if (!nsym || !nqual) { rc = NGSC_O_EFORMAT; goto out; }
    /* AnalyzeDNA tasks.cpp:226-257: the 4-symbol skew test compares zeros (Q4) => plain iff <= 4 */
    if (nsym > 4) flags &= ~(uint32_t)F_DNA_PLAIN;

    /* ---- title stream ---- */
    bw_init(&wt, 1 << 16);
    {
        titles_t t = {buf, n, nf, te, se, ls, fe};
        rc = title_stream(&t, &wt);
        if (rc) goto out;
    }

    /* ---- DNA stream tasks.cpp:513-569 ---- */
    bw_init(&wd, 1 << 20);
    for (uint32_t i = 0; i < nsym; ++i) bw_u8(&wd, symbols[i]);
    {
        ngsc_o_code dcodes[512];
        if (!(flags & F_DNA_PLAIN)) {
            uint32_t st[256];
            for (uint32_t i = 0; i < nsym; ++i) st[i] = dna_occ[symbols[i]];
            put_tree(&wd, st, nsym, dcodes);
        }
        for (uint32_t r = 0; r < n; ++r) {
            uint32_t s = te[r] + 1;
            for (uint32_t j = 0; j < slen[r]; ++j) {
                uint8_t c = sym_code[buf[s + j]];
                if (flags & F_DNA_PLAIN) bw_bits(&wd, c, 2);
                else bw_bits(&wd, dcodes[c].code, dcodes[c].len);
            }
        }
        bw_align(&wd);
    }

    /* ---- quality stream tasks.cpp:260-286, 572-622 ---- */
    bw_init(&wq, 1 << 20);
    for (uint32_t i = 0; i < nqual; ++i) bw_u8(&wq, qualities[i]);
    {
        qhist = (uint32_t *)calloc((size_t)(max_q + 1) * nqual, sizeof(uint32_t));
        for (uint32_t r = 0; r < n; ++r) {
            uint32_t q0 = se[r] + 3;
            for (uint32_t j = 0; j < qlen[r]; ++j) {
                uint8_t c = qua_code[buf[q0 + j]];
                qhist[(size_t)(j + 1) * nqual + c]++;
                qhist[c]++;
            }
        }
        ngsc_o_code *qc = (ngsc_o_code *)malloc(sizeof(ngsc_o_code) * (size_t)(max_q + 1) * 2 * nqual);
        for (uint32_t p = 0; p <= max_q; ++p)
            put_tree(&wq, qhist + (size_t)p * nqual, nqual, qc + (size_t)p * 2 * nqual);
        bw_align(&wq);
        for (uint32_t r = 0; r < n; ++r) {
            uint32_t q0 = se[r] + 3;
            for (uint32_t j = 0; j < qlen[r]; ++j) {
                const ngsc_o_code *cd = qc + (size_t)(j + 1) * 2 * nqual + qua_code[buf[q0 + j]];
                bw_bits(&wq, cd->code, cd->len);
            }
        }
        bw_align(&wq);
        free(qc);
    }

    /* ---- info header + length table phyNGSC.cpp:719-742 ---- */
    bw_init(&wi, 64 + n * 2);
    bw_u32(&wi, n);
    bw_u32(&wi, max_q);
    bw_u32(&wi, max_s);
    bw_u8(&wi, (uint8_t)nsym);
    bw_u8(&wi, 0); /* QUALITY_PLAIN */
    bw_u8(&wi, (uint8_t)nqual);
    bw_u32(&wi, flags);
    {
        uint32_t b = bit_length(max_q);
        for (uint32_t r = 0; r < n; ++r) bw_bits(&wi, qlen[r], b);
        bw_align(&wi);
    }
    if (wi.oom || wt.oom || wq.oom || wd.oom) { rc = NGSC_O_ENOMEM; goto out; }

    /* ---- concat info | title | quality | DNA phyNGSC.cpp:793-840 ---- */
    {
        uint64_t tot = wi.n + wt.n + wq.n + wd.n;
        uint8_t *o = (uint8_t *)malloc(tot ? tot : 1);
        memcpy(o, wi.p, wi.n);
        memcpy(o + wi.n, wt.p, wt.n);
        memcpy(o + wi.n + wt.n, wq.p, wq.n);
        memcpy(o + wi.n + wt.n + wq.n, wd.p, wd.n);
        *out = o;
        *out_len = tot;
    }
    if (info) {
        info->n_records = n; info->max_qlen = max_q; info->max_slen = max_s;
        info->nsym = nsym; info->nqual = nqual; info->flags = flags; info->n_fields = nf;
        info->info_len = (uint32_t)wi.n; info->title_len = (uint32_t)wt.n;
        info->qual_len = (uint32_t)wq.n; info->dna_len = (uint32_t)wd.n;
        info->next_offset = next;
    }
out:
    free(buf); free(te); free(se); free(ls); free(fe); free(slen); free(qlen); free(qhist);
    free(wi.p); free(wt.p); free(wq.p); free(wd.p);
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Block header / footer
 * ---------------------------------------------------------------------------------------------- */
uint32_t ngsc_o_header_size(uint32_t bewr, uint32_t beso, uint32_t nosb)
{
    /* BlockHeader::HeaderSize structures.h:323-333 */
    uint32_t bits = bewr + 6 + 12 + beso * nosb + 5 + 2;
    return (bits + 7) / 8;
}

uint32_t ngsc_o_make_header(uint8_t *out, uint32_t wrid, uint32_t bewr, uint32_t bhs, uint32_t beso, uint32_t bcss,
                            const uint32_t *sbol, uint32_t nosb)
{
    bw_t w;
    bw_init(&w, 64);
    bw_bits(&w, wrid, bewr); /* tasks.cpp:1186-1196 */
    bw_bits(&w, bhs, 12);
    bw_bits(&w, nosb, 6);
    bw_bits(&w, beso, 5);
    bw_bits(&w, bcss, 2);
    for (uint32_t i = 0; i < nosb; ++i) bw_bits(&w, sbol[i], beso);
    bw_align(&w);
    uint32_t n = (uint32_t)w.n;
    memcpy(out, w.p, n);
    free(w.p);
    return n;
}

uint32_t ngsc_o_make_footer(uint8_t *out, uint32_t P, uint64_t fastq_size, uint32_t n_blocks, uint32_t n_subblocks,
                            const int32_t *overlaps, const uint32_t *cbo, const uint32_t *lb_sizes)
{
    /* tasks.cpp:1104-1176 */
    uint32_t lb_max = 0, lb_min = 0xffffffffu;
    int32_t ov_max = overlaps[0];
    for (uint32_t i = 0; i < P; ++i) {
        if (lb_sizes[i] > lb_max) lb_max = lb_sizes[i];
        if (lb_sizes[i] < lb_min) lb_min = lb_sizes[i];
        if (overlaps[i] > ov_max) ov_max = overlaps[i];
    }
    uint32_t BEPS = floor_log2_plus1(P), BEFS = floor_log2_plus1(fastq_size), BEBS = floor_log2_plus1(n_blocks);
    uint32_t BESS = floor_log2_plus1(n_subblocks), BELB = floor_log2_plus1(lb_max);
    uint32_t BEOV = floor_log2_plus1((uint64_t)(ov_max > 0 ? ov_max : 0));
    uint32_t LBES = lb_max != lb_min ? 0 : 1;
    bw_t w;
    bw_init(&w, 256);
    bw_bits(&w, BEPS, 4); bw_bits(&w, BEFS, 6); bw_bits(&w, BEBS, 4); bw_bits(&w, BESS, 4);
    bw_bits(&w, BELB, 5); bw_bits(&w, BEOV, 4); bw_bits(&w, LBES, 1);
    bw_bits(&w, P, BEPS);
    if (BEFS > 32) {
        bw_bits(&w, (uint32_t)(fastq_size >> 32), BEFS - 32);
        bw_bits(&w, (uint32_t)(fastq_size & 0xffffffffu), 32);
    } else {
        bw_bits(&w, (uint32_t)fastq_size, BEFS);
    }
    bw_bits(&w, n_blocks, BEBS);
    bw_bits(&w, n_subblocks, BESS);
    for (uint32_t i = 1; i < P; ++i) bw_bits(&w, (uint32_t)overlaps[i], BEOV);
    uint32_t cb = ceil_log2(P);
    for (uint32_t i = 0; i < n_blocks; ++i) bw_bits(&w, cbo[i], cb);
    if (!LBES)
        for (uint32_t i = 0; i < P; ++i) bw_bits(&w, lb_sizes[i], BELB);
    bw_align(&w);
    uint32_t len = (uint32_t)w.n;
    bw_raw(&w, (uint8_t)(len >> 8)); /* Put2Bytes(GetIO_Buffer_Pos()) tasks.cpp:1175 */
    bw_raw(&w, (uint8_t)len);
    uint32_t n = (uint32_t)w.n;
    memcpy(out, w.p, n);
    free(w.p);
    return n;
}

/* ------------------------------------------------------------------------------------------------
 * Whole file, P virtual ranks
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint8_t **blocks;
    uint32_t *block_len;
    uint32_t n_blocks, cap_blocks;
    uint32_t n_subblocks;
    uint32_t last_block_size;
    int32_t overlap_used;
    int rc;
} rank_out_t;

static void rank_push_block(rank_out_t *ro, uint8_t *b, uint32_t len)
{
    if (ro->n_blocks == ro->cap_blocks) {
        ro->cap_blocks = ro->cap_blocks ? ro->cap_blocks * 2 : 16;
        ro->blocks = (uint8_t **)realloc(ro->blocks, sizeof(uint8_t *) * ro->cap_blocks);
        ro->block_len = (uint32_t *)realloc(ro->block_len, sizeof(uint32_t) * ro->cap_blocks);
    }
    ro->blocks[ro->n_blocks] = b;
    ro->block_len[ro->n_blocks] = len;
    ro->n_blocks++;
}

static void compress_rank(const uint8_t *fastq, uint64_t FS, uint32_t P, uint32_t rank, rank_out_t *ro)
{
    const uint64_t W = NGSC_O_WINDOW;
    memset(ro, 0, sizeof *ro);
    /* region split phyNGSC.cpp:113-124 */
    int64_t region = (int64_t)(FS / P);
    int64_t wr_start = (int64_t)rank * region;
    uint32_t overlap = NGSC_O_OVERLAP;
    int64_t wr_end = (rank != P - 1) ? wr_start + region + overlap - 1 : (int64_t)FS - 1;
    int64_t bsize = (int64_t)W;
    if (region < bsize) {
        bsize = wr_end - wr_start + 1;
        if (rank == P - 1) overlap = 0;
    }
    /* first-record sync :131-156 */
    int64_t cur = 0;
    if (rank != 0) {
        uint64_t av = (uint64_t)wr_start < FS ? FS - (uint64_t)wr_start : 0;
        if (av > (uint64_t)bsize) av = (uint64_t)bsize;
        cur = ngsc_o_first_record(fastq + wr_start, av);
        if (cur < 0) {
            if (getenv("NGSC_O_VERBOSE")) fprintf(stderr, "oracle: rank %u finds no record start in its region\n", rank);
            ro->rc = NGSC_O_EFORMAT; return;
        }
    }
    ro->overlap_used = (int32_t)cur;
    uint64_t rec_start = (uint64_t)cur;
    int64_t bytes_read = 0;
    int64_t win_pos = wr_start; /* file offset of the current window */

    /* block assembly state phyNGSC.cpp:842-903 */
    const uint32_t bewr = ceil_log2(P);
    uint8_t *payload = (uint8_t *)malloc(W);
    uint64_t written = 0;
    uint32_t sbol[4096];
    uint32_t nsb = 0, beso = 0, bcss = 0;

    while (bytes_read < region) { /* :168 */
        uint64_t avail = (uint64_t)win_pos < FS ? FS - (uint64_t)win_pos : 0;
        /* The reference reads records that end past its window out of whatever lies behind the buffer
           (phyNGSC.cpp:308, :615: undefined once a record is longer than the overlap).  The oracle defines
           that case: the bytes are the file's own.  1 MiB of them is kept; a record that still does not fit
           is rejected in ngsc_o_subblock. */
        uint64_t cap = (uint64_t)bsize + (1u << 20);
        if (avail > cap) avail = cap;
        uint8_t *sb = NULL;
        uint64_t sblen = 0;
        ngsc_o_sbinfo inf;
        int rc = ngsc_o_subblock(fastq + win_pos, avail, (uint64_t)bsize, rec_start, overlap, &sb, &sblen, &inf);
        if (rc) {
            if (getenv("NGSC_O_VERBOSE"))
                fprintf(stderr, "oracle: rank %u window at %lld (size %lld, first record at +%llu): rc %d\n", rank, (long long)win_pos,
                        (long long)bsize, (unsigned long long)rec_start, rc);
            ro->rc = rc; free(payload); return;
        }
        ro->n_subblocks++;
        bytes_read += (int64_t)inf.next_offset; /* :745 */
        win_pos = bytes_read + wr_start;        /* :746 */
        if (win_pos + bsize > wr_end) {         /* :748-755 */
            if (rank == P - 1) overlap = 0;
            bsize = wr_end - win_pos;
        }
        rec_start = 0;

        /* :843-903 */
        if (nsb >= 4095) { ro->rc = NGSC_O_EARG; free(sb); free(payload); return; }
        sbol[nsb++] = (uint32_t)sblen;
        uint32_t mx = 0;
        for (uint32_t i = 0; i < nsb; ++i) if (sbol[i] > mx) mx = sbol[i];
        beso = floor_log2_plus1(mx);
        uint32_t h = ngsc_o_header_size(bewr, beso, nsb);
        if (written + sblen + h > W) {
            bcss |= 1; /* LSBS */
            uint32_t fill = (uint32_t)(W - (written + h));
            sbol[nsb - 1] = fill;
            uint8_t *blk = (uint8_t *)malloc(W);
            uint32_t hn = ngsc_o_make_header(blk, rank, bewr, h, beso, bcss, sbol, nsb);
            (void)hn;
            memcpy(blk + h, payload, written);
            memcpy(blk + h + written, sb, fill);
            rank_push_block(ro, blk, (uint32_t)W);
            memcpy(payload, sb + fill, sblen - fill);
            written = sblen - fill;
            bcss |= 2;            /* FSBS */
            bcss &= ~1u;
            nsb = 0;
            sbol[nsb++] = (uint32_t)written;
        } else {
            memcpy(payload + written, sb, sblen);
            written += sblen;
        }
        free(sb);
    }
    if (written > 0) { /* :910-928 */
        uint32_t h = ngsc_o_header_size(bewr, beso, nsb);
        uint8_t *blk = (uint8_t *)malloc(written + h + 8);
        uint32_t hn = ngsc_o_make_header(blk, rank, bewr, h, beso, bcss, sbol, nsb);
        memcpy(blk + hn, payload, written);
        rank_push_block(ro, blk, (uint32_t)(written + hn));
        ro->last_block_size = (uint32_t)(written + hn);
    } else {
        ro->last_block_size = 0;
    }
    free(payload);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int ngsc_o_compress_file(const uint8_t *fastq, uint64_t size, uint32_t P, int threads, uint8_t **out,
                         uint64_t *out_len, ngsc_o_filestats *stats)
{
    if (P < 2 || !size) return NGSC_O_EARG; /* phyNGSC.cpp:91-97 */
    rank_out_t *ro = (rank_out_t *)calloc(P, sizeof(rank_out_t));
    double t0 = now_s();
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (int r = 0; r < (int)P; ++r) compress_rank(fastq, size, P, (uint32_t)r, &ro[r]);
    double t1 = now_s();
    int rc = NGSC_O_OK;
    for (uint32_t r = 0; r < P; ++r) if (ro[r].rc) { rc = ro[r].rc; break; }
    uint32_t nb = 0, nsb = 0, maxb = 0;
    uint64_t total = 0;
    for (uint32_t r = 0; r < P; ++r) {
        nb += ro[r].n_blocks;
        nsb += ro[r].n_subblocks;
        if (ro[r].n_blocks > maxb) maxb = ro[r].n_blocks;
        for (uint32_t k = 0; k < ro[r].n_blocks; ++k) total += ro[r].block_len[k];
    }
    if (rc == NGSC_O_OK) {
        uint8_t *o = (uint8_t *)malloc(total + 65536 + (size_t)nb * 2);
        uint32_t *cbo = (uint32_t *)malloc(sizeof(uint32_t) * (nb + 1));
        int32_t *ov = (int32_t *)malloc(sizeof(int32_t) * P);
        uint32_t *lbs = (uint32_t *)malloc(sizeof(uint32_t) * P);
        uint64_t off = 0;
        uint32_t bi = 0;
        /* deterministic order (replaces arrival order, phyNGSC.cpp:875/925 MPI_File_write_shared + :997-1009
           timestamp sort): ranks in groups of consecutive ranks whose regions make up at most 512 MiB of input,
           group after group; inside a group round k = k-th block of each of its ranks.  (The order equal-speed ranks
           produce when a node works through the file 512 MiB at a time; one group = the whole file up to 512 MiB.) */
        uint64_t region = size / P;
        uint32_t rpg = region ? (uint32_t)((512ull << 20) / region) : P;
        if (rpg < 1) rpg = 1;
        if (rpg > P) rpg = P;
        (void)maxb;
        for (uint32_t g0 = 0; g0 < P; g0 += rpg) {
            uint32_t g1 = g0 + rpg < P ? g0 + rpg : P, gmax = 0;
            for (uint32_t r = g0; r < g1; ++r) if (ro[r].n_blocks > gmax) gmax = ro[r].n_blocks;
            for (uint32_t k = 0; k < gmax; ++k)
                for (uint32_t r = g0; r < g1; ++r)
                    if (k < ro[r].n_blocks) {
                        memcpy(o + off, ro[r].blocks[k], ro[r].block_len[k]);
                        off += ro[r].block_len[k];
                        cbo[bi++] = r;
                    }
        }
        for (uint32_t r = 0; r < P; ++r) { ov[r] = ro[r].overlap_used; lbs[r] = ro[r].last_block_size; }
        off += ngsc_o_make_footer(o + off, P, size, nb, nsb, ov, cbo, lbs);
        *out = o;
        *out_len = off;
        free(cbo); free(ov); free(lbs);
        if (stats) { stats->n_blocks = nb; stats->n_subblocks = nsb; stats->out_bytes = off; stats->seconds = t1 - t0; }
    }
    for (uint32_t r = 0; r < P; ++r) {
        for (uint32_t k = 0; k < ro[r].n_blocks; ++k) free(ro[r].blocks[k]);
        free(ro[r].blocks);
        free(ro[r].block_len);
    }
    free(ro);
    return rc;
}

#ifdef NGSC_ORACLE_MAIN
/* ngsc_oracle_cli <in.fastq> <out.ngsc> <P> [threads] */
int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s in.fastq out.ngsc P [threads]\n", argv[0]); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    fseek(f, 0, SEEK_END);
    uint64_t sz = (uint64_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *in = (uint8_t *)malloc(sz + 1);
    if (fread(in, 1, sz, f) != sz) { perror("read"); return 2; }
    fclose(f);
    uint8_t *o = NULL;
    uint64_t on = 0;
    ngsc_o_filestats st;
    int rc = ngsc_o_compress_file(in, sz, (uint32_t)atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 1, &o, &on, &st);
    if (rc) { fprintf(stderr, "oracle error %d\n", rc); return 3; }
    f = fopen(argv[2], "wb");
    fwrite(o, 1, on, f);
    fclose(f);
    fprintf(stderr, "blocks=%u subblocks=%u out=%llu B in %.3f s\n", st.n_blocks, st.n_subblocks,
            (unsigned long long)on, st.seconds);
    free(o); free(in);
    return 0;
}
#endif
