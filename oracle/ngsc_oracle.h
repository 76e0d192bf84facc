/*
 * ngsc_oracle.h - CPU restatement of the phyNGSC sub-block compressor (TEST INFRASTRUCTURE).
 *
 * This is the parity oracle for the MI355X HIP engine in phyngsc_amd/.  It is NOT part of the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * It restates, in plain C and with a different structure (two-pass array statistics instead of
 * std::map, a two-queue Huffman merge instead of an STL heap), the algorithm of the reference
 * (pcdslab/PHYNGSC).  Every function cites the reference file:line it follows.
 *
 * PARITY STATUS: pinned.  tests/test_oracle_golden.py checks this restatement block-for-block
 * against .ngsc files produced by the unmodified reference binary (oracle/build_ref.sh ->
 * oracle/_ref/phyNGSC, run by tests/golden/make_golden.py; fixtures under tests/golden/).
 */
#ifndef NGSC_ORACLE_H
#define NGSC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    NGSC_O_OK = 0,
    NGSC_O_EFORMAT = -1,     /* not 4-line FASTQ with a bare '+' line (reference: undefined behaviour)   */
    NGSC_O_EFIELDS = -2,     /* a title has a different separator count than record 0 (ref: warning + UB) */
    NGSC_O_ETOOMANY = -3,    /* > 100000 records in one sub-block (ref: warning + heap overflow)          */
    NGSC_O_ECOLORSPACE = -4, /* SOLiD colour-space input (phyNGSC.cpp:473-547): out of scope              */
    NGSC_O_ERANGE = -5,      /* numeric title value >= 2^31 / bit width 64 (ref: PutBits(x,64) UB)        */
    NGSC_O_ENOMEM = -6,
    NGSC_O_EARG = -7
};

#define NGSC_O_WINDOW (1u << 23) /* READ_BUFFER_SIZE / WRITE_BUFFER_SIZE, defs.h:20-21 */
#define NGSC_O_OVERLAP 500u      /* phyNGSC.cpp:48 */
#define NGSC_O_MAX_RECORDS 100000u /* phyNGSC.cpp:51 */

typedef struct {
    uint32_t code;
    uint32_t len;
} ngsc_o_code;

/* Per-sub-block facts, exposed so GPU stage tests can localise a mismatch. */
typedef struct {
    uint32_t n_records;
    uint32_t max_qlen;  /* max_quality_length, phyNGSC.cpp:721            */
    uint32_t max_slen;  /* global_max_sequence_length, phyNGSC.cpp:722    */
    uint32_t nsym;      /* DNA alphabet size                              */
    uint32_t nqual;     /* quality alphabet size                          */
    uint32_t flags;     /* fastq_flags, phyNGSC.cpp:730                   */
    uint32_t n_fields;
    uint32_t info_len, title_len, qual_len, dna_len; /* stream byte lengths, order info|title|quality|DNA */
    uint64_t next_offset; /* offset of the next record inside the window, phyNGSC.cpp:745 */
} ngsc_o_sbinfo;

/*
 * Huffman code build + tree blob (huffman.cpp:18-118, huffman.h:134-147).
 * freq[n]; codes[] must hold 2n-1 entries.  blob (may be NULL) receives the serialised tree
 * (root_id, n_symbols, min_len, pre-order bits) WITHOUT the leading u32 mem_size; *blob_len its size.
 * blob capacity needed: 9 + (n*(1+bits_per_id) + n + 7)/8 + 8.  Returns root_id.
 */
int ngsc_o_huffman(const uint32_t *freq, uint32_t n, ngsc_o_code *codes, uint8_t *blob, uint32_t *blob_len);

/*
 * One sub-block (one iteration of the loop at phyNGSC.cpp:168-840).
 *  win/avail : window bytes actually available (avail may exceed wsize: bytes past the window are
 *              what a longer buffer would hold; it may be smaller at end of file)
 *  wsize     : r_buffer_size;  rec_start: rec_start_pos;  overlap: current overlap value
 * On success *out (malloc'd, caller frees) holds info|title|quality|DNA.
 */
int ngsc_o_subblock(const uint8_t *win, uint64_t avail, uint64_t wsize, uint64_t rec_start, uint32_t overlap,
                    uint8_t **out, uint64_t *out_len, ngsc_o_sbinfo *info);

/* Record index only (phyNGSC.cpp:254-331, single-thread semantics). Arrays hold >= NGSC_O_MAX_RECORDS+1. */
int ngsc_o_record_index(const uint8_t *win, uint64_t avail, uint64_t wsize, uint64_t rec_start, uint32_t overlap,
                        uint32_t *title_end, uint32_t *seq_end, uint32_t *n_records, uint64_t *next_offset);

/* First-record synchronisation of rank>0 (phyNGSC.cpp:131-156). Returns offset inside buf, or -1. */
int64_t ngsc_o_first_record(const uint8_t *buf, uint64_t avail);

/* Block header (tasks.cpp:1179-1200, structures.h:323-333). Returns header byte count written to out. */
uint32_t ngsc_o_make_header(uint8_t *out, uint32_t wrid, uint32_t bewr, uint32_t bhs, uint32_t beso, uint32_t bcss,
                            const uint32_t *sbol, uint32_t nosb);
uint32_t ngsc_o_header_size(uint32_t bewr, uint32_t beso, uint32_t nosb);

/* Footer (tasks.cpp:1104-1176). cbo[n_blocks] = WRID of every block in physical order. Returns bytes. */
uint32_t ngsc_o_make_footer(uint8_t *out, uint32_t P, uint64_t fastq_size, uint32_t n_blocks, uint32_t n_subblocks,
                            const int32_t *overlaps, const uint32_t *cbo, const uint32_t *lb_sizes);

typedef struct {
    uint32_t n_blocks;
    uint32_t n_subblocks;
    uint64_t out_bytes;
    double seconds; /* wall time of the compression proper (excludes output concatenation) */
} ngsc_o_filestats;

/*
 * Whole file with P virtual ranks (phyNGSC.cpp:113-166 region split + sync, 168-906 loop,
 * 842-928 block assembly, 930-1057 footer).  Blocks are emitted in the deterministic order
 * "round k = k-th block of every rank, ranks ascending" (the reference's order is arrival order).
 * threads: OpenMP threads across virtual ranks (1 = scalar).
 */
int ngsc_o_compress_file(const uint8_t *fastq, uint64_t size, uint32_t P, int threads, uint8_t **out,
                         uint64_t *out_len, ngsc_o_filestats *stats);

/*
 * Diagnostic: varying numeric title values met since the last reset whose digits exceed 32 bits while the wrapped value stays
 * below 2^31 - the reference (and this restatement) store them modulo 2^32, so the container does not hold its input
 * (utils.h:117-125).  The HIP engine refuses such input with NGSC_ERR_RANGE; the tests use this count to expect exactly that.
 */
uint64_t ngsc_o_lossy_count(int reset);

/*
 * Decoder (ngsc_decode.c), the round-trip leg of the tests: one sub-block's info|title|quality|DNA
 * bytes -> the FASTQ text of its records.  *fastq is malloc()ed; consumed = bytes of sb read (equals
 * the sub-block's SBOL length).  The reference has no decompressor; its unused Fetch* fragments
 * (tasks.cpp:625-1101) are cited in the source.
 */
int ngsc_o_decode_subblock(const uint8_t *sb, uint64_t len, uint8_t **fastq, uint64_t *fastq_len, uint64_t *consumed,
                           uint32_t *n_records);

#ifdef __cplusplus
}
#endif
#endif
