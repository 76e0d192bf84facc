/*
 * fastq_gen - deterministic synthetic FASTQ generator (test/bench infrastructure, not product).
 *
 * Every record is a pure function of (seed, record index): a splitmix64 stream keyed by both.
 * That makes generation reproducible across machines (the GPU box regenerates the inputs whose
 * reference manifests were produced in the build container) and, with fixed_width=1 and
 * len_min==len_max, seekable: every record has the same byte size, so any byte range of a
 * multi-GB logical file can be produced independently by the rank that owns it.
 *
 * Record shape follows the reference's README sample (README.md:107-126):
 *   @ERR005195.<i> BGI-FC30BFTAAXX_5_1_000:<n>/2 \n <bases> \n + \n <qualities> \n
 *
 * Build: gcc -O2 -shared -fPIC -o libfastq_gen.so fastq_gen.c   (ctypes)
 *        gcc -O2 -DFASTQ_GEN_MAIN -o fastq_gen fastq_gen.c      (CLI)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t seed;
    uint64_t first_index;  /* number printed in the first record's title                     */
    uint32_t style;        /* 0 ERR005195/36bp-like, 1 SRR/HWI long title, 2 Illumina-1.8     */
    uint32_t len_min;      /* read length range, uniform                                     */
    uint32_t len_max;
    uint32_t qual_model;   /* 0: 70% 'I' else uniform '!'..'I'; 1: geometric below Q38;      */
                           /* 2: skewed alphabet, weight ~ 1/(1+|40-q|)^2                    */
    uint32_t n_rate_ppm;   /* probability (ppm) that a base is N                             */
    uint32_t n_lowq_pct;   /* percent of N bases that get a transferable quality '!'..'('    */
    uint32_t fixed_width;  /* 1: numeric title fields drawn from fixed-digit ranges           */
    uint32_t reserved;
} fq_params;

static inline uint64_t sm64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint32_t put_u64(uint8_t *o, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    for (int i = 0; i < n; ++i) o[i] = (uint8_t)tmp[n - 1 - i];
    return (uint32_t)n;
}

static inline uint32_t put_str(uint8_t *o, const char *s)
{
    uint32_t n = (uint32_t)strlen(s);
    memcpy(o, s, n);
    return n;
}

/* cumulative weights for qual_model 2, built once */
static uint32_t g_skew_cdf[42];
static int g_skew_ready = 0;
static void skew_init(void)
{
    uint32_t acc = 0;
    for (int q = 0; q < 42; ++q) {
        int d = q > 40 ? q - 40 : 40 - q;
        acc += (uint32_t)(1000000.0 / ((1.0 + d) * (1.0 + d)));
        g_skew_cdf[q] = acc;
    }
    g_skew_ready = 1;
}

/* Writes record `i` (0-based) at `o`; returns its byte length. `o` needs >= 2*len_max + 160 bytes. */
uint64_t fq_record(const fq_params *p, uint64_t i, uint8_t *o)
{
    /* key the stream by a full-avalanche hash of (seed, i); a plain additive key would make
       record i+1's splitmix stream a one-step shift of record i's */
    uint64_t k0 = (p->seed + 0x632BE59BD9B4E019ull) * 0xD1342543DE82EF95ull;
    uint64_t k1 = i + 0x2545F4914F6CDD1Dull;
    uint64_t s = sm64(&k0) ^ (sm64(&k1) * 0xA24BAED4963EE407ull);
    s = sm64(&s);
    uint8_t *w = o;
    uint64_t idx = p->first_index + i;
    uint32_t span = p->len_max - p->len_min + 1;
    uint32_t L = p->len_min + (uint32_t)(sm64(&s) % span);
    uint64_t r;

    switch (p->style) {
    default:
    case 0:
        w += put_str(w, "@ERR005195.");
        w += put_u64(w, idx);
        w += put_str(w, " BGI-FC30BFTAAXX_5_1_000:");
        r = sm64(&s);
        w += put_u64(w, p->fixed_width ? 1000 + r % 2000 : 1 + r % 2000);
        w += put_str(w, "/2");
        break;
    case 1:
        w += put_str(w, "@SRR062634.");
        w += put_u64(w, idx);
        w += put_str(w, " HWI-ST:7:");
        r = sm64(&s);
        w += put_u64(w, 1101 + r % 1216);
        *w++ = ':';
        r = sm64(&s);
        w += put_u64(w, p->fixed_width ? 10000 + r % 10000 : 1000 + r % 19000);
        *w++ = ':';
        r = sm64(&s);
        w += put_u64(w, p->fixed_width ? 100000 + r % 100000 : 1000 + r % 199000);
        w += put_str(w, " length=");
        w += put_u64(w, L);
        break;
    case 2:
        w += put_str(w, "@M00001:23:000000000-A1B2C:1:");
        r = sm64(&s);
        w += put_u64(w, 1101 + r % 1000);
        *w++ = ':';
        r = sm64(&s);
        w += put_u64(w, p->fixed_width ? 10000 + r % 10000 : 1000 + r % 29000);
        *w++ = ':';
        r = sm64(&s);
        w += put_u64(w, p->fixed_width ? 10000 + r % 10000 : 1000 + r % 29000);
        r = sm64(&s);
        w += put_str(w, (r % 16) ? " 1:N:0:" : " 1:Y:0:");
        w += put_str(w, (r >> 8) % 4 ? "ATCACG" : "CGATGT");
        break;
    }
    *w++ = '\n';

    uint8_t *seq = w;
    uint8_t *qual = w + L + 3;
    static const char acgt[4] = {'A', 'C', 'G', 'T'};
    if (!g_skew_ready) skew_init();

    uint64_t bits = 0;
    int nb = 0;
    for (uint32_t k = 0; k < L; ++k) {
        if (nb < 2) { bits = sm64(&s); nb = 64; }
        seq[k] = (uint8_t)acgt[bits & 3];
        bits >>= 2; nb -= 2;
    }
    /* qualities */
    for (uint32_t k = 0; k < L; ++k) {
        uint64_t x = sm64(&s);
        uint8_t q;
        switch (p->qual_model) {
        default:
        case 0:
            q = (x % 10) < 7 ? 'I' : (uint8_t)('!' + (x >> 8) % 41);
            break;
        case 1: { /* geometric drop below Q38: P(38-d) ~ 2^-(d+1), floor at Q2 */
            uint32_t d = 0;
            uint64_t y = x;
            while ((y & 1) && d < 36) { y >>= 1; ++d; }
            q = (uint8_t)('!' + 38 - d);
            break;
        }
        case 2: {
            uint32_t t = (uint32_t)(x % g_skew_cdf[41]);
            uint32_t qi = 0;
            while (g_skew_cdf[qi] <= t) ++qi;
            q = (uint8_t)('!' + qi);
            break;
        }
        }
        qual[k] = q;
    }
    /* N injection */
    if (p->n_rate_ppm) {
        for (uint32_t k = 0; k < L; ++k) {
            uint64_t x = sm64(&s);
            if ((x % 1000000u) < p->n_rate_ppm) {
                seq[k] = 'N';
                if (((x >> 32) % 100u) < p->n_lowq_pct) qual[k] = (uint8_t)('!' + (x >> 48) % 8);
            }
        }
    }
    w = seq + L;
    *w++ = '\n';
    *w++ = '+';
    *w++ = '\n';
    w += L;
    *w++ = '\n';
    return (uint64_t)(w - o);
}

/* Byte size of every record when the configuration is fixed-width; 0 otherwise. */
uint64_t fq_fixed_record_size(const fq_params *p)
{
    if (!p->fixed_width || p->len_min != p->len_max) return 0;
    uint8_t tmp[4096];
    if (p->len_max > 1900) return 0;
    uint64_t a = fq_record(p, 0, tmp);
    /* index digits must not change over the range used: caller picks first_index accordingly */
    return a;
}

/*
 * Generates whole records starting at record index `rec0` until either `n_records` records have
 * been written (n_records != 0) or the next record would exceed `target_bytes`.
 * Returns bytes written; *n_out receives the record count.
 */
uint64_t fq_generate(const fq_params *p, uint64_t rec0, uint64_t n_records, uint64_t target_bytes,
                     uint8_t *out, uint64_t cap, uint64_t *n_out)
{
    uint64_t off = 0, n = 0;
    uint64_t need = 2ull * p->len_max + 160;
    uint8_t *tmp = (uint8_t *)malloc(need);
    for (;;) {
        if (n_records && n >= n_records) break;
        uint64_t len = fq_record(p, rec0 + n, tmp);
        if (!n_records && off + len > target_bytes) break;
        if (off + len > cap) break;
        memcpy(out + off, tmp, len);
        off += len;
        ++n;
    }
    free(tmp);
    if (n_out) *n_out = n;
    return off;
}

/*
 * Seekable range generation for fixed-width configurations: fills out[0..len) with bytes
 * [byte_off, byte_off+len) of the infinite logical file made of records 0,1,2,...
 * Returns 0 on success, -1 if the configuration is not fixed-width.
 */
int fq_generate_range(const fq_params *p, uint64_t byte_off, uint64_t len, uint8_t *out)
{
    uint64_t rs = fq_fixed_record_size(p);
    if (!rs) return -1;
    uint8_t *tmp = (uint8_t *)malloc(2ull * p->len_max + 160);
    uint64_t rec = byte_off / rs, skip = byte_off % rs, done = 0;
    while (done < len) {
        uint64_t l = fq_record(p, rec, tmp);
        if (l != rs) { free(tmp); return -2; }
        uint64_t take = l - skip;
        if (take > len - done) take = len - done;
        memcpy(out + done, tmp + skip, take);
        done += take;
        skip = 0;
        ++rec;
    }
    free(tmp);
    return 0;
}

#ifdef FASTQ_GEN_MAIN
/* fastq_gen <out.fastq> <target_bytes> [seed style len_min len_max qual_model n_ppm n_lowq fixed first_index] */
int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s out.fastq target_bytes [seed style len_min len_max qual_model n_ppm n_lowq_pct fixed first_index]\n", argv[0]);
        return 1;
    }
    fq_params p = {1234, 1, 0, 36, 36, 0, 0, 0, 0, 0};
    uint64_t target = strtoull(argv[2], 0, 10);
    if (argc > 3) p.seed = strtoull(argv[3], 0, 10);
    if (argc > 4) p.style = (uint32_t)atoi(argv[4]);
    if (argc > 5) p.len_min = (uint32_t)atoi(argv[5]);
    if (argc > 6) p.len_max = (uint32_t)atoi(argv[6]);
    if (argc > 7) p.qual_model = (uint32_t)atoi(argv[7]);
    if (argc > 8) p.n_rate_ppm = (uint32_t)atoi(argv[8]);
    if (argc > 9) p.n_lowq_pct = (uint32_t)atoi(argv[9]);
    if (argc > 10) p.fixed_width = (uint32_t)atoi(argv[10]);
    if (argc > 11) p.first_index = strtoull(argv[11], 0, 10);
    FILE *f = fopen(argv[1], "wb");
    if (!f) { perror("fopen"); return 2; }
    const uint64_t CH = 64ull << 20;
    uint8_t *buf = (uint8_t *)malloc(CH + 8192);
    uint64_t total = 0, rec = 0;
    while (total < target) {
        uint64_t want = target - total < CH ? target - total : CH;
        uint64_t n = 0;
        uint64_t got = fq_generate(&p, rec, 0, want, buf, CH + 8192, &n);
        if (!n) break;
        fwrite(buf, 1, got, f);
        total += got;
        rec += n;
    }
    fclose(f);
    free(buf);
    fprintf(stderr, "wrote %llu bytes, %llu records\n", (unsigned long long)total, (unsigned long long)rec);
    return 0;
}
#endif
