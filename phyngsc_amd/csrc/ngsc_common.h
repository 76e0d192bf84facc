// ngsc_common.h - constants, device-visible structures and the per-thread format logic shared by
// the HIP kernels of the MI355X FASTQ sub-block compressor.
//
// Everything marked NGSC_HD is plain per-thread logic without wave intrinsics.  The kernels in
// kernels_*.hip call these helpers from device code; tests/harness/ compiles the same helpers
// with g++ to unit-test the format logic against the oracle on machines without a GPU.  There is
// no CPU execution path in the product library: only __global__ kernels call these functions.
//
// Reference citations are file:line into pcdslab/PHYNGSC.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NGSC_HD __host__ __device__ inline
#else
#define NGSC_HD inline
#endif

namespace ngsc {

// ---- format constants ---------------------------------------------------------------------------
constexpr uint32_t WINDOW = 1u << 23;       // READ/WRITE_BUFFER_SIZE, defs.h:20-21
constexpr uint32_t OVERLAP = 500;           // phyNGSC.cpp:48
constexpr uint32_t MAX_RECORDS = 100000;    // records_per_th, phyNGSC.cpp:51
constexpr uint32_t GROUP = 32;              // DEFAULT_B_SIZE, tasks.cpp:26
constexpr uint32_t MAX_STAT_LEN = 128;      // MAX_FIELD_STAT_LEN, tasks.cpp:25
constexpr uint32_t MAX_NUM_HUF = 512;       // MAX_NUM_VAL_HUF, tasks.cpp:27

// ---- engine limits (inputs beyond them get NGSC_ERR_UNSUPPORTED, never a wrong stream) ------------
constexpr uint32_t SMALL_TREE = 64;         // symbols of a tree the small Huffman instance builds (kernels_huffman.hip)
constexpr uint32_t MAX_FIELDS = 64;         // separators per title the word-parallel / cached title kernels take (one bit each in fmask_t)
typedef uint64_t fmask_t;                   // a bit per title field
constexpr uint32_t WIDE_FIELDS = 512;       // ... and what the wide title kernels take (kernels_title_wide.hip): a first title holds at most
                                            // REC0_TITLE_CAP = 512 bytes, so it cannot have more separators
// Field masks of the wide path: one bit per field in WIDE_FIELDS / 64 words
struct WideMask { uint64_t w[WIDE_FIELDS / 64]; };
NGSC_HD void fm_set(fmask_t &m, uint32_t f, uint32_t bit = 1u) { m |= (fmask_t)(bit & 1u) << f; }
NGSC_HD uint32_t fm_get(const fmask_t &m, uint32_t f) { return (uint32_t)((m >> f) & 1u); }
NGSC_HD void fm_or(fmask_t &a, const fmask_t &b) { a |= b; }
NGSC_HD void fm_clear(fmask_t &m) { m = 0; }
NGSC_HD void fm_set(WideMask &m, uint32_t f, uint32_t bit = 1u) { m.w[f >> 6] |= (uint64_t)(bit & 1u) << (f & 63u); }
NGSC_HD uint32_t fm_get(const WideMask &m, uint32_t f) { return (uint32_t)((m.w[f >> 6] >> (f & 63u)) & 1u); }
NGSC_HD void fm_or(WideMask &a, const WideMask &b) { for (uint32_t k = 0; k < WIDE_FIELDS / 64; ++k) a.w[k] |= b.w[k]; }
NGSC_HD void fm_clear(WideMask &m) { for (uint32_t k = 0; k < WIDE_FIELDS / 64; ++k) m.w[k] = 0; }
constexpr uint32_t MAX_TEXT_LEN0 = 512;     // record-0 length of a non-constant text field (Hamming mask bits): a whole first title (REC0_TITLE_CAP)
constexpr uint32_t MAX_CODE_LEN = 26;       // packed code table entry = code << 5 | len
constexpr uint32_t NCF_CAP = 8;             // non-constant title fields cached per record by the tiled path
constexpr uint32_t REC0_TITLE_CAP = 512;      // longest first-record title the title kernels keep in LDS
constexpr uint32_t NUM_CAND_CAP = MAX_FIELDS;  // numeric candidate fields (numeric in a sub-block's first record) whose values and deltas are tracked: all of them

constexpr uint32_t BLOB_SLOT_256 = 336;     // >= 9 + ceil((256*9 + 255)/8) = 329
constexpr uint32_t BLOB_SLOT_512 = 720;     // >= 9 + ceil((512*10 + 511)/8) = 713
constexpr uint32_t TEXT_TREES = MAX_STAT_LEN + 1;

// fastq_flags, defs.h:69-80
enum : uint32_t {
    F_TRY_LZ = 1, F_DNA_PLAIN = 2, F_CONST_NUM_FIELDS = 4, F_PLUS_ONLY = 8, F_USE_DELTA = 16,
    F_DELTA_CONSTANT = 32, F_DELTA_NO_BEGIN_NUC = 64, F_VARIABLE_LENGTH = 128, F_LINE_BREAKS = 256
};

// error codes (mirrored in include/ngsc.h)
enum : int {
    OK = 0,
    ERR_FORMAT = -1,       // not 4-line FASTQ with a bare '+' line / quality length != sequence length
    ERR_FIELDS = -2,       // title with a different separator count than the sub-block's first record
    ERR_TOOMANY = -3,      // > 100000 records in one sub-block
    ERR_COLORSPACE = -4,   // inside the engine: "run the job again on a decoded copy" (kernels_cs.hip); to callers: only from the decoder
    ERR_RANGE = -5,        // numeric title value that the reference's uint32 parse leaves >= 2^31, or delta range >= 2^31
    ERR_NOMEM = -6,
    ERR_ARG = -7,
    ERR_UNSUPPORTED = -8,  // engine limit exceeded (MAX_FIELDS, MAX_TEXT_LEN0, MAX_CODE_LEN, text slots)
    ERR_HIP = -9,
    ERR_CAPACITY = -10,    // output buffer too small
    ERR_SHAPE = -100       // inside the engine: the job is not of the shape its arrays were sized for (engine.cpp: JobShape) - run it again
};

enum : uint32_t { KIND_CONST = 0, KIND_NUM = 1, KIND_TEXT = 2 };

// ---- small helpers --------------------------------------------------------------------------------
// title separators " ._,=:/-#\n" (phyNGSC.cpp:208) as a 128-bit membership bitmap
NGSC_HD bool is_sep(uint8_t c)
{
    const uint64_t lo = (1ull << 10) | (1ull << 32) | (1ull << 35) | (1ull << 44) | (1ull << 45) | (1ull << 46) |
                        (1ull << 47) | (1ull << 58) | (1ull << 61);
    const uint64_t hi = (1ull << (95 - 64));
    return c < 64 ? ((lo >> c) & 1) : (c < 128 ? ((hi >> (c - 64)) & 1) : 0);
}

// trans_amb_codes (phyNGSC.cpp:184-206): 1 = ACGT, 2..16 = transferable IUPAC symbol, 0 = anything else
NGSC_HD uint32_t amb_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'C': case 'G': case 'T': return 1;
    case 'Y': return 2; case 'R': return 3; case 'W': return 4; case 'S': return 5;
    case 'K': return 6; case 'M': return 7; case 'D': return 8; case 'V': return 9;
    case 'H': return 10; case 'B': return 11; case 'N': return 12; case 'X': return 13;
    case 'U': return 14; case '.': return 15; case '-': return 16;
    default: return 0;
    }
}

// Does a sub-block whose first read starts with these two bytes need the colour-space treatment? (phyNGSC.cpp:478-484)
NGSC_HD bool cs_first_read(uint32_t c0, uint32_t c1) { return (c1 >= '0' && c1 <= '3') || (c0 >= '0' && c0 <= '3'); }

// quality byte of a transferred base (phyNGSC.cpp:579)
NGSC_HD uint8_t transfer_quality(uint32_t amb, uint8_t q) { return (uint8_t)(128 + (amb << 3) - 16 + (q - 33)); }

// BitStream::BitLength (bit_stream.h:268-277)
NGSC_HD uint32_t bit_length(uint64_t x)
{
    for (uint32_t i = 0; i < 32; ++i)
        if (x < (1ull << i)) return i;
    return 64;
}
NGSC_HD uint32_t floor_log2_plus1(uint64_t x) { uint32_t b = 0; while (x) { ++b; x >>= 1; } return b; }
NGSC_HD uint32_t ceil_log2(uint64_t x) { uint32_t b = 0; while ((1ull << b) < x) ++b; return b; }

// order-preserving int32 -> uint32 map, so every statistic is an unsigned max over zero-initialised memory
NGSC_HD uint32_t bias(int32_t v) { return (uint32_t)v ^ 0x80000000u; }
NGSC_HD int32_t unbias(uint32_t u) { return (int32_t)(u ^ 0x80000000u); }

// utils::is_num + utils::to_num (utils.h:107-125).  v wraps like uint32 (what the reference goes on with); ovf is set when
// the TRUE value exceeds INT32_MAX: the reference then either holds a negative int32 (wrapped value >= 2^31: it loops over a
// negative count) or carries on with the value modulo 2^32 and writes a container that does not hold its input.  A varying
// numeric field with such a value is refused (finalize_field: ERR_RANGE); a constant one is stored as text and never parsed.
NGSC_HD bool parse_num(const uint8_t *b, uint32_t s, uint32_t e, uint32_t &v, bool &ovf)
{
    uint32_t len = e - s;
    if (len == 0) return false;
    uint32_t acc = 0;   // wraps like utils::to_num (utils.h:117-125)
    bool big = false;
    for (uint32_t i = s; i < e; ++i) {
        uint32_t d = (uint32_t)b[i] - '0';
        if (d > 9) return false;
        const uint64_t t = (uint64_t)acc * 10 + d;
        big = big || (t >> 32) != 0;
        acc = (uint32_t)t;
    }
    if (len > 1 && b[s] == '0') return false;
    v = acc;
    ovf = big || (acc >> 31) != 0;
    return true;
}

// Number of separators in [ls, te] == number of title fields (phyNGSC.cpp:345-379, 409-415)
NGSC_HD uint32_t count_fields(const uint8_t *b, uint32_t ls, uint32_t te)
{
    uint32_t n = 0;
    for (uint32_t i = ls; i <= te; ++i) n += is_sep(b[i]);
    return n;
}

// Advances over one field starting at pos: returns its end (position of the separator).
NGSC_HD uint32_t field_end(const uint8_t *b, uint32_t pos)
{
    while (!is_sep(b[pos])) ++pos;
    return pos;
}

NGSC_HD bool bytes_equal(const uint8_t *b, uint32_t s1, uint32_t s2, uint32_t len)
{
    for (uint32_t i = 0; i < len; ++i)
        if (b[s1 + i] != b[s2 + i]) return false;
    return true;
}

// ---- device-visible structures --------------------------------------------------------------------
struct Cut {            // one sub-block (one iteration of the loop phyNGSC.cpp:168-906)
    uint32_t rank;
    uint32_t rec_lo;    // first record (index into the job's record table)
    uint32_t n_rec;
    uint32_t overlap;   // overlap value in force (informational)
    uint64_t win_pos;   // file offset of the window
    uint64_t wsize;     // r_buffer_size of the window
};

struct RankPlan {
    uint32_t n_cuts;
    int32_t overlap_used;  // wr_ov_used, phyNGSC.cpp:160
    uint32_t first_rec;
    int32_t err;
    // where the rank's loop stands behind the last planned sub-block (a rank walked in several parts, ngsc_compress_rank_part)
    uint64_t next_pos;     // file offset of the next window = of its first record
    uint64_t bytes_read;   // p_bytes_read, phyNGSC.cpp:745-746
    uint64_t bsize;        // size of the next window, phyNGSC.cpp:748-755
    uint32_t overlap;      // overlap in force for it
    uint32_t done;         // the rank is finished
};

// The shape a step's arrays and grids were sized for (engine.cpp: JobShape): k_init_sb holds the job against it.
struct ShapeCaps {
    uint32_t n_rec_total, S, max_rec, rec_bound;
    uint64_t slots, groups, tiles;
    uint32_t max_title, max_q, max_rec_bytes;
};

// A rank taken up in the middle (all zero / null: from its first record).
struct PlanResume {
    uint64_t next_pos, bytes_read, bsize;
    uint32_t overlap;
    uint32_t resume;       // 0: start of the rank
};

// Per (sub-block, field) statistics accumulated with unsigned atomicMax / atomicOr over zeroed memory.
struct FieldStat {
    uint32_t max_len;
    uint32_t inv_min_len;   // max of ~len
    uint32_t not_const;     // some record differs from record 0
    uint32_t not_len_const;
    uint32_t not_num;
    uint32_t ovf;           // some numeric value > INT32_MAX (parse_num)
    uint32_t max_v, inv_min_v;  // biased int32
    uint32_t max_d, inv_min_d;  // biased int32 (deltas between consecutive records)
    uint32_t not_ham[16];   // bit p set: some record differs from record 0 at position p (< 512 = REC0_TITLE_CAP)
    uint32_t no_delta;      // deltas of this field were not tracked (too many numeric candidates)
    uint32_t pad[5];
};
constexpr uint32_t FSTAT_WORDS = 32;   // words per FieldStat: scalars 0..9, Hamming words 10..25, no_delta 26
static_assert(sizeof(FieldStat) == 4 * FSTAT_WORDS, "FieldStat layout");

// Finalised field description (tasks.cpp:196-222 + the header decisions of tasks.cpp:304-390)
struct FieldDesc {
    uint32_t start0;        // record 0's field start (offset in the job buffer)
    uint32_t len0;
    uint32_t sep;
    uint32_t kind;          // KIND_*
    uint32_t is_len_const;
    uint32_t min_len, max_len;
    int32_t min_v, max_v, min_d, max_d;
    uint32_t delta_coding;
    uint32_t bits_num, bits_val, bits_len;
    uint32_t has_huff;      // numeric Huffman tree stored (tasks.cpp:338)
    uint32_t n_sym;         // its symbol count (diff)
    int32_t base;           // min_v or min_d
    uint32_t slot;          // text statistics / code slot, 0xffffffff if none
    uint32_t ham[16];       // Hamming mask bits (1 = every record equals record 0 at that position)
    uint32_t kf;            // index among the sub-block's non-constant fields
    uint32_t cand;          // numeric candidate index of the field (Rec0Tab::kmap), 0xff if record 0's field is not numeric
    uint32_t pad[3];
};
static_assert(sizeof(FieldDesc) == 160, "FieldDesc layout");

struct SbState {        // per sub-block scalars
    uint32_t rec_lo, n_rec, rank, nf;
    uint32_t max_q, max_s;          // atomicMax
    uint32_t nsym, nqual, flags, lt_bits;
    uint32_t qbits, dbits;          // payload bit totals
    uint32_t qual_hdr, dna_hdr;     // header byte counts (alphabet + trees)
    uint32_t title_hdr, title_body; // bytes
    uint32_t info_len, title_len, qual_len, dna_len;
    uint32_t total;
    uint32_t n_bases;               // sum of sequence lengths (= quality bytes) of the sub-block
    uint32_t ncf;                   // number of non-constant title fields
    uint32_t slot_bias;             // record index + slot_bias = the record's slot in the arrays that depend on the sub-block
                                    // (slot of the sub-block's first record = records of all earlier sub-blocks; mod 2^32)
    uint32_t cs;                    // colour space (phyNGSC.cpp:473-547): bit0 reads are colours (FLAG_USE_DELTA), bit1 no primer
                                    // base (FLAG_DELTA_NO_BEGIN_NUC), bit2 NOT every read starts like the first (!FLAG_DELTA_CONSTANT)
    uint32_t cs_v;                  // first record that does not start like the first one (reads in front of it are stored
                                    // without their primer base and first quality)
    uint64_t out_off;               // byte offset of the sub-block in the output buffer
    uint8_t symbols[256];
    uint8_t qualities[256];
    uint8_t sym_code[256];
    uint8_t qua_code[256];
};

// Field table of a sub-block's first record (phyNGSC.cpp:342-379): every other title is compared with it.
struct Rec0Tab {
    uint32_t nf0, ncand;
    uint16_t start[MAX_FIELDS], len[MAX_FIELDS];  // field start (offset in the title) and length
    uint8_t kmap[MAX_FIELDS];                     // numeric candidate index, 0xff = not numeric, 0xfe = untracked
    uint8_t cand_f[NUM_CAND_CAP];                 // candidate index -> field
    uint8_t title[REC0_TITLE_CAP];
    uint32_t val0[MAX_FIELDS];                    // value of record 0's field where it is numeric (kmap != 0xff)
    uint8_t ovf0[MAX_FIELDS];                     // ... and whether it exceeds INT32_MAX
    uint16_t woff[MAX_FIELDS];                    // dword offset of the field inside al
    uint32_t al[(REC0_TITLE_CAP + 8 * MAX_FIELDS) / 4];  // field-aligned copy, each field zero padded to 8 bytes
    uint32_t sepm[4], digm[4];                    // separator / digit bit per byte of the first 128 title bytes
    // per field, as dwords a kernel can fetch with SCALAR loads (gfx950 has no sub-dword scalar loads; the u8 / u16 arrays above
    // cost a vector load and a readfirstlane per value): [0] = start | len << 16, [1] = kmap | ovf0 << 8 | woff << 16
    uint32_t fdw[MAX_FIELDS][2];
};
static_assert(sizeof(Rec0Tab) % 4 == 0, "Rec0Tab is copied word-wise");

struct AmbPos { uint32_t x, y; };   // positions 0,1 in x, 2,3 in y

// Everything a kernel needs to find a sub-block's working memory.
struct Arena {
    // job-wide
    const uint8_t *buf;     // FASTQ bytes of the job
    uint32_t buf_len;
    const uint32_t *te, *se, *ls;   // record table: title end, sequence end, line start (ls has n+1 entries)
    uint32_t *slen;         // per record: post-transfer sequence length | non-ACGT count (0..6, 7 = more) << 28 | transfer flag << 31
    AmbPos *apos;           // per record with 1..4 non-ACGT bases: their positions, 16 bits each (k_sq_stats_reg)
    uint64_t *bcnt;         // per record: A | C << 16 | T << 32 | G << 48 among its unambiguous bases (k_sq_stats_reg)
    uint32_t *qoff;         // per record: quality payload bits, then exclusive offsets
    uint32_t *doff;         // per record: DNA payload bits, then exclusive offsets
    uint32_t *goff;         // per title group: bytes, then exclusive offsets
    uint32_t *grp_lo;       // per sub-block: first title group index
    uint32_t *tbits;        // per record: title bits (tiled path)
    fmask_t *gflags;        // per title group: flag bits (tiled path)
    uint32_t *wq, *wd;      // per tile: quality / DNA payload bits, then exclusive offsets (tiled path)
    uint32_t *ubits;        // per (tile, thread): quality bits | DNA bits << 16 of the thread's record slice
    uint64_t *tsep;         // per record: separator mask of the title, tsep_words words of 64 title bytes (word-parallel path)
    uint32_t tsep_words;    // 1: titles <= 64 B, 2: titles <= 128 B
    uint32_t *tv;           // [NUM_CAND_CAP][tw_stride]: value of every numeric candidate field per record (word-parallel path)
    uint32_t *tw;           // [NCF_CAP][tw_stride]: cached non-constant title fields per record (tiled path)
    uint32_t tw_stride;
    uint32_t *tile_lo;      // per sub-block: first tile index
    int32_t *err;           // job error word (min over error codes)
    // per sub-block arrays (index by sub-block)
    uint32_t fcap;          // fields per sub-block in fstat / fdesc / nhist / ncodes / nblob: MAX_FIELDS, or WIDE_FIELDS for a job
                            // with a title of more than MAX_FIELDS separators (which the wide title kernels take)
    uint32_t *wide_needed;  // set by k_rec0_fields when a first title has more than fcap separators: the job runs again, wide
    SbState *sb;
    FieldStat *fstat;       // [S][fcap]
    FieldDesc *fdesc;       // [S][fcap]
    Rec0Tab *rec0;          // [S]
    uint32_t *dna_occ;      // [S][256]
    uint32_t *qhist;        // [S][qrows][256]  row 0 = totals, row k+1 = position k
    uint32_t *nhist;        // [S][fcap][512]
    uint32_t *qcodes;       // [S][qrows][256]  entry = code<<5|len, indexed by alphabet index
    uint8_t *qblob;         // [S][qrows][BLOB_SLOT_256]
    uint32_t *qblob_len;    // [S][qrows]
    uint32_t *dcodes;       // [S][256] indexed by raw base byte (plain: len 2)
    uint8_t *dblob;         // [S][BLOB_SLOT_256]
    uint32_t *dblob_len;    // [S]
    uint32_t *ncodes;       // [S][fcap][512]
    uint8_t *nblob;         // [S][fcap][BLOB_SLOT_512]
    uint32_t *nblob_len;    // [S][fcap]
    // text-field slot pool
    uint32_t *slot_counter;
    uint32_t *slot_owner;   // [n_slots] -> sub-block * fcap + field
    uint32_t n_slots;
    uint32_t *chist;        // [n_slots][129][256]
    uint32_t *tcodes;       // [n_slots][129][256]
    uint8_t *tblob;         // [n_slots][129][BLOB_SLOT_256]
    uint32_t *tblob_len;    // [n_slots][129]
    uint32_t qrows;         // job max quality length + 1
    uint32_t n_sb;
    uint32_t cs_mode;       // the job runs on a copy of its input in which colour-space reads are decoded (kernels_cs.hip)
    // NGSC_SQ_DENSE: the statistics kernel leaves every record's bases and qualities as 16-byte chunks in CHUNK-MAJOR arrays
    // (chunk c of record g at dense_x + (c * dstride + g) * 16, dstride = records of the job), and measure / pack read them
    // there: the 64 lanes of a wave - 64 consecutive records - store and load 1 KB of consecutive memory per instruction
    // instead of 16 bytes each at addresses a whole record apart.  dstride == 0: off
    uint8_t *dense_s, *dense_q;
    uint32_t dstride;
    // (round 3) the LAST chunk of the job's longest read, when it holds at most 12 bytes (36 bases: 16 + 16 + 4), is kept as
    // dtail_b = 4 / 8 / 12 bytes per record instead of 16: chunk dtail_c of record g at dense_x + dtail_c * dstride * 16 + g * dtail_b
    // (a quarter of the copies' bytes at 36 bases).  dtail_c == 0xffffffff: every chunk is 16 bytes
    uint32_t dtail_c, dtail_b;
    // (round 3) set by k_sq_finalize when some sub-block's quality or base alphabet exceeds SMALL_TREE symbols: the 512-symbol
    // Huffman instance has quality / base trees to build (else its workgroups leave at once)
    uint32_t *big_trees;
    // (round 4) raised by the field finalisation when some numeric title field is Huffman-coded: else the launch that builds those
    // trees has nothing to do and its workgroups leave at once (they used to find that out field by field: 23 us on the title chain)
    uint32_t *num_huff;   // bit 0: a Huffman-coded numeric field of more than one value (a histogram, a tree), bit 1: a text field (histograms,
                          // trees); the histogram and tree kernels leave at once without theirs
    // (round 3) the quality code table of every sub-block in the layout the pack kernel keeps in LDS (k_sq_format_qtab):
    // [qrows + 17][57] dwords, qfmt_stride dwords per sub-block
    uint32_t *qfmt;
    uint32_t qfmt_stride;
    uint8_t *out;           // output buffer
};

// ---- MSB-first bit sinks (BitStream semantics, bit_stream.h:149-200) ---------------------------------
// Counts bits only.
struct BitCounter {
    uint64_t bits = 0;
    NGSC_HD void put(uint32_t, uint32_t n) { bits += n; }
    NGSC_HD void align() { bits = (bits + 7) & ~7ull; }
    NGSC_HD void byte(uint8_t) { align(); bits += 8; }
    NGSC_HD void word(uint32_t) { align(); bits += 32; }
    NGSC_HD void bytes(const uint8_t *, uint32_t n) { align(); bits += 8ull * n; }
    NGSC_HD uint64_t nbytes() const { return (bits + 7) >> 3; }
};

// Writes bytes to memory owned exclusively by the calling thread.
struct ByteWriter {
    uint8_t *p;
    uint64_t acc = 0;
    uint32_t nacc = 0;
    uint64_t n = 0;
    NGSC_HD explicit ByteWriter(uint8_t *dst) : p(dst) {}
    NGSC_HD void put(uint32_t v, uint32_t nb)
    {
        if (!nb) return;
        if (nb < 32) v &= (1u << nb) - 1u;
        acc = (acc << nb) | v;
        nacc += nb;
        while (nacc >= 8) { p[n++] = (uint8_t)(acc >> (nacc - 8)); nacc -= 8; }
        acc &= (1ull << nacc) - 1ull;
    }
    NGSC_HD void align()
    {
        if (nacc) { p[n++] = (uint8_t)(acc << (8 - nacc)); nacc = 0; acc = 0; }
    }
    NGSC_HD void byte(uint8_t b) { align(); p[n++] = b; }
    NGSC_HD void word(uint32_t v)
    {
        align();
        p[n++] = (uint8_t)(v >> 24); p[n++] = (uint8_t)(v >> 16); p[n++] = (uint8_t)(v >> 8); p[n++] = (uint8_t)v;
    }
    NGSC_HD void bytes(const uint8_t *d, uint32_t len) { align(); for (uint32_t i = 0; i < len; ++i) p[n++] = d[i]; }
    NGSC_HD uint64_t nbytes() const { return n; }
};

// =====================================================================================================
// Title logic (phyNGSC.cpp:342-423, tasks.cpp:22-223, tasks.cpp:289-510)
// =====================================================================================================

// libstdc++ doubles std::vector capacity, so while the reference push_back()s its Field objects
// (phyNGSC.cpp:350) every Field pushed before the last reallocation is copy-constructed, and Field's
// copy constructor (structures.h:126-202) drops num_values.  Only fields with index >= this value
// still hold record 0's entry and keep counting values (tasks.cpp:175), i.e. can get a value-coding
// Huffman tree.  Returns the largest power of two <= nf-1 (0 for nf == 1).
NGSC_HD uint32_t num_values_kept_from(uint32_t nf)
{
    if (nf < 2) return 0;
    uint32_t p = 1;
    while (p * 2 <= nf - 1) p *= 2;
    return p;
}

// Finalises one field from its statistics (tasks.cpp:196-222 and the decisions at tasks.cpp:324-338).
// n_rec: records in the sub-block; f: field index; nf: field count.  Returns an error code.
NGSC_HD int finalize_field(const FieldStat &st, uint32_t start0, uint32_t len0, uint8_t sep, uint32_t n_rec,
                           uint32_t f, uint32_t nf, FieldDesc &d)
{
    d.start0 = start0;
    d.len0 = len0;
    d.sep = sep;
    d.max_len = st.max_len;
    d.min_len = ~st.inv_min_len;
    d.is_len_const = !st.not_len_const;
    d.slot = 0xffffffffu;
    d.kf = 0;
    d.has_huff = 0;
    d.n_sym = 0;
    d.base = 0;
    d.delta_coding = 0;
    d.bits_num = d.bits_val = d.bits_len = 0;
    d.min_v = d.max_v = 0;
    d.min_d = 1;   // Field() defaults, structures.h:103-106
    d.max_d = -1;
    for (int i = 0; i < 16; ++i) d.ham[i] = ~st.not_ham[i];
    if (!st.not_const) { d.kind = KIND_CONST; return OK; }
    if (!st.not_num) {
        if (st.ovf) return ERR_RANGE;
        if (st.no_delta) return ERR_UNSUPPORTED;
        d.kind = KIND_NUM;
        d.min_v = unbias(~st.inv_min_v);
        d.max_v = unbias(st.max_v);
        if (n_rec >= 2) { d.min_d = unbias(~st.inv_min_d); d.max_d = unbias(st.max_d); }
        int64_t vr = (int64_t)d.max_v - d.min_v, dr = (int64_t)d.max_d - d.min_d;
        if (dr > 0x7fffffffll) return ERR_RANGE;
        d.delta_coding = !(vr < dr);
        int64_t diff = d.delta_coding ? dr : vr;
        d.bits_num = bit_length((uint64_t)diff);
        d.bits_val = bit_length((uint64_t)vr);
        if (d.bits_num > 32 || d.bits_val > 32) return ERR_RANGE;
        diff += 1;
        bool have_map = d.delta_coding ? (n_rec >= 2) : (f >= num_values_kept_from(nf));
        if (diff <= (int64_t)MAX_NUM_HUF && have_map) {
            d.has_huff = 1;
            d.n_sym = (uint32_t)diff;
            d.base = d.delta_coding ? d.min_d : d.min_v;
        }
        return OK;
    }
    d.kind = KIND_TEXT;
    if (len0 > MAX_TEXT_LEN0) return ERR_UNSUPPORTED;
    d.bits_len = bit_length((uint64_t)(d.max_len - d.min_len));
    return OK;
}

NGSC_HD bool ham_bit(const FieldDesc &d, uint32_t j) { return j < d.len0 && ((d.ham[j >> 5] >> (j & 31)) & 1); }

// Does position j of a text field get a Huffman tree? (tasks.cpp:364-377; tree 128: 378-387)
NGSC_HD bool text_tree_needed(const FieldDesc &d, uint32_t j)
{
    if (j < MAX_STAT_LEN) {
        uint32_t npos = d.max_len < MAX_STAT_LEN ? d.max_len : MAX_STAT_LEN;
        return j < npos && !ham_bit(d, j);
    }
    return j == MAX_STAT_LEN && d.max_len >= MAX_STAT_LEN;
}

// Title header (tasks.cpp:302-390) written or measured through Sink: the field count, then one part per field.  Every part
// starts and ends on a byte boundary (its own bit runs - a text field's Hamming mask - are closed by the align() calls), so the
// parts can be measured independently and written at offsets from a prefix sum (kernels_pack.hip: k_write_headers).
template <class Sink>
NGSC_HD void title_header_field(const Arena &A, uint32_t s, uint32_t f, Sink &w)
{
    const uint8_t *buf = A.buf;
    const FieldDesc &d = A.fdesc[(size_t)s * A.fcap + f];
    w.byte((uint8_t)d.sep);
    w.byte(d.kind == KIND_CONST);
    if (d.kind == KIND_CONST) {
        w.word(d.len0);
        w.bytes(buf + d.start0, d.len0);
        return;
    }
    w.byte(d.kind == KIND_NUM);
    if (d.kind == KIND_NUM) {
        w.word((uint32_t)d.min_v);
        w.word((uint32_t)d.max_v);
        w.word((uint32_t)d.min_d);
        w.word((uint32_t)d.max_d);
        if (d.has_huff) {
            size_t t = (size_t)s * A.fcap + f;
            uint32_t bl = A.nblob_len[t];
            w.align();
            w.word(bl);
            w.bytes(A.nblob + t * BLOB_SLOT_512, bl);
        }
        return;
    }
    w.byte((uint8_t)d.is_len_const);
    w.word(d.len0);
    w.word(d.max_len);
    w.word(d.min_len);
    w.bytes(buf + d.start0, d.len0);
    for (uint32_t j = 0; j < d.len0; ++j) w.put(ham_bit(d, j), 1);
    for (uint32_t j = 0; j <= MAX_STAT_LEN; ++j) {
        if (!text_tree_needed(d, j)) continue;
        size_t t = (size_t)d.slot * TEXT_TREES + j;
        uint32_t bl = A.tblob_len[t];
        w.align();
        w.word(bl);
        w.bytes(A.tblob + t * BLOB_SLOT_256, bl);
    }
    w.align();
}

template <class Sink>
NGSC_HD void title_header(const Arena &A, uint32_t s, Sink &w)
{
    const SbState &sb = A.sb[s];
    w.word(sb.nf);
    for (uint32_t f = 0; f < sb.nf; ++f) title_header_field(A, s, f, w);
}

// ---- 32-record title groups (tasks.cpp:393-509), decomposed per record --------------------------------
// All functions take the byte buffer explicitly (`buf[pos]`, pos = offset inside the job buffer) so that a
// kernel can hand in an LDS-resident tile (pointer rebased by the tile's first offset).

template <class M>
struct TitleDiffsT {
    M txt, val, dlt;   // bit f set: field f of this record breaks the group's constancy
};
typedef TitleDiffsT<fmask_t> TitleDiffs;

// Record at p_cur (not the group's first record) against the group's first record (p_first) and its
// predecessor (p_prev): BlockDesc updates of tasks.cpp:63-81 and 127-147, plus the store-time rule that a
// constant delta only counts when it equals min_delta (tasks.cpp:415).
template <class M = fmask_t>
NGSC_HD TitleDiffsT<M> title_record_diffs(const uint8_t *buf, const FieldDesc *fd, uint32_t nf, uint32_t p_first,
                                          uint32_t p_prev, uint32_t p_cur)
{
    TitleDiffsT<M> r;
    fm_clear(r.txt); fm_clear(r.val); fm_clear(r.dlt);
    uint32_t pf = p_first, pp = p_prev, pc = p_cur;
    for (uint32_t f = 0; f < nf; ++f) {
        uint32_t ef = field_end(buf, pf), ep = field_end(buf, pp), ec = field_end(buf, pc);
        const FieldDesc &d = fd[f];
        if (d.kind == KIND_TEXT) {
            if (ec - pc != ef - pf || !bytes_equal(buf, pc, pf, ec - pc)) fm_set(r.txt, f);
        } else if (d.kind == KIND_NUM) {
            uint32_t vf = 0, vp = 0, vc = 0;
            bool o;
            parse_num(buf, pf, ef, vf, o);
            parse_num(buf, pp, ep, vp, o);
            parse_num(buf, pc, ec, vc, o);
            if (vc != vf) fm_set(r.val, f);
            if ((int32_t)(vc - vp) != d.min_d) fm_set(r.dlt, f);
        }
        pf = ef + 1; pp = ep + 1; pc = ec + 1;
    }
    return r;
}

// Group flag bits from the OR of its records' diffs.  bit f = the flag written for field f (1 = constant).
template <class M>
NGSC_HD M title_group_const(const FieldDesc *fd, uint32_t nf, const TitleDiffsT<M> &g, uint32_t group_size)
{
    M gconst;
    fm_clear(gconst);
    for (uint32_t f = 0; f < nf; ++f) {
        const FieldDesc &d = fd[f];
        if (d.kind == KIND_CONST) continue;
        uint32_t flag;
        if (d.kind == KIND_TEXT) flag = !fm_get(g.txt, f);
        else if (d.delta_coding) {
            // a one-record group keeps block_delta_constant == 0, which must equal min_delta (tasks.cpp:415)
            flag = group_size == 1 ? (d.min_d == 0) : !fm_get(g.dlt, f);
        } else flag = !fm_get(g.val, f);
        fm_set(gconst, f, flag);
    }
    return gconst;
}

// The group's flag bits: one per non-constant field, in field order (tasks.cpp:401-425).
template <class M, class Sink>
NGSC_HD void title_put_flags(const FieldDesc *fd, uint32_t nf, const M &gconst, Sink &w)
{
    for (uint32_t f = 0; f < nf; ++f)
        if (fd[f].kind != KIND_CONST) w.put(fm_get(gconst, f), 1);
}

// One record's title bits (tasks.cpp:427-507).  first: it is the group's first record.
template <class M, class Sink>
NGSC_HD void title_record_emit(const uint8_t *buf, const Arena &A, uint32_t s, const FieldDesc *fd, uint32_t nf,
                               bool first, uint32_t p_prev, uint32_t p_cur, const M &gconst, Sink &w)
{
    uint32_t pp = p_prev, pc = p_cur;
    for (uint32_t f = 0; f < nf; ++f) {
        uint32_t ep = field_end(buf, pp), ec = field_end(buf, pc);
        const FieldDesc &d = fd[f];
        if (d.kind == KIND_NUM) {
            uint32_t vp = 0, vc = 0;
            bool o;
            parse_num(buf, pc, ec, vc, o);
            if (first) {
                w.put(vc - (uint32_t)d.min_v, d.bits_val);               // tasks.cpp:457
            } else if (!fm_get(gconst, f)) {
                parse_num(buf, pp, ep, vp, o);
                uint32_t to_store = d.delta_coding ? (vc - vp - (uint32_t)d.min_d) : (vc - (uint32_t)d.min_v);
                if (d.has_huff) {
                    uint32_t e = A.ncodes[((size_t)s * A.fcap + f) * MAX_NUM_HUF + to_store];
                    w.put(e >> 5, e & 31);
                } else {
                    w.put(to_store, d.bits_num);
                }
            }
        } else if (d.kind == KIND_TEXT) {
            if (first || !fm_get(gconst, f)) {                            // tasks.cpp:487
                uint32_t len = ec - pc;
                if (!d.is_len_const) w.put(len - d.min_len, d.bits_len);
                const uint32_t *tc = A.tcodes + (size_t)d.slot * TEXT_TREES * 256;
                for (uint32_t j = 0; j < len; ++j) {
                    if (!ham_bit(d, j)) {
                        uint32_t e = tc[(j < MAX_STAT_LEN ? j : MAX_STAT_LEN) * 256 + buf[pc + j]];
                        w.put(e >> 5, e & 31);
                    }
                }
            }
        }
        pp = ep + 1; pc = ec + 1;
    }
}

// One whole group, sequentially (the wide title kernels, one thread per group, and the CPU harness).
// g0/g1 are record indices inside the job's record table.  M: the field mask type (fmask_t: up to MAX_FIELDS fields).
template <class M = fmask_t, class Sink>
NGSC_HD void title_group(const Arena &A, uint32_t s, uint32_t g0, uint32_t g1, Sink &w)
{
    const uint8_t *buf = A.buf;
    const FieldDesc *fd = A.fdesc + (size_t)s * A.fcap;
    const uint32_t nf = A.sb[s].nf;
    TitleDiffsT<M> acc;
    fm_clear(acc.txt); fm_clear(acc.val); fm_clear(acc.dlt);
    for (uint32_t r = g0 + 1; r < g1; ++r) {
        TitleDiffsT<M> d = title_record_diffs<M>(buf, fd, nf, A.ls[g0], A.ls[r - 1], A.ls[r]);
        fm_or(acc.txt, d.txt); fm_or(acc.val, d.val); fm_or(acc.dlt, d.dlt);
    }
    const M gconst = title_group_const(fd, nf, acc, g1 - g0);
    title_put_flags(fd, nf, gconst, w);
    for (uint32_t r = g0; r < g1; ++r)
        title_record_emit(buf, A, s, fd, nf, r == g0, r > g0 ? A.ls[r - 1] : A.ls[r], A.ls[r], gconst, w);
    w.align();  // tasks.cpp:508
}

// =====================================================================================================
// Huffman: sequential part executed by one lane per tree (huffman.cpp:18-118)
// =====================================================================================================
// Inputs: sorted[n] = (freq << 32 | symbol) ascending.  Work arrays: ifreq[n], lc[2n], rc[2n], code[2n], len[2n],
// stack[2n+2], blob (>= BLOB_SLOT bytes).  Outputs: code/len per node, blob bytes, returns blob length;
// *max_len_out = longest code.
// The passes over all n symbols - counting the zero frequencies, clearing code / len, the shortest code - are split off
// (huffman_zeros / huffman_clear / huffman_min_len) so that a wave can do them with all its lanes (kernels_huffman.hip: one
// lane walking 256 or 512 LDS entries three times was most of a text or numeric-field tree's 80 us); what is left for ONE
// lane is the merge, the top-down code walk and the pre-order blob over the symbols that are PRESENT (huffman_core).
NGSC_HD uint32_t huffman_zeros(const uint64_t *sorted, uint32_t n)
{
    uint32_t zeros = 0;
    while (zeros < n && (sorted[zeros] >> 32) == 0) ++zeros;
    return zeros;
}
NGSC_HD void huffman_clear(uint32_t *code, uint8_t *len, uint32_t n)
{
    for (uint32_t i = 0; i < 2 * n - 1; ++i) { code[i] = 0; len[i] = 0; }
}
NGSC_HD uint32_t huffman_min_len(const uint8_t *len, uint32_t n)
{
    uint32_t min_len = n;
    for (uint32_t i = 0; i < n; ++i)
        if (len[i] < min_len && len[i] > 0) min_len = len[i];
    return min_len;
}
// merge + code walk: zeros = huffman_zeros(), code / len cleared.  Returns the root's node id.
NGSC_HD int huffman_core(const uint64_t *sorted, uint32_t n, uint32_t zeros, uint32_t *ifreq, uint16_t *lc, uint16_t *rc, uint32_t *code,
                         uint8_t *len, uint32_t *max_len_out)
{
    // compact: drop zero-frequency symbols while more than two remain (huffman.cpp:44-50)
    uint32_t drop = n > 2 ? (zeros < n - 2 ? zeros : n - 2) : 0;
    uint32_t present = n - drop;
    // merge (huffman.cpp:57-70) as a two-queue merge; ties go to the smaller node id and a leaf id is
    // always smaller than an internal id, which is the heap's (frequency, symbol) order
    uint32_t lq = drop, iq = 0, made = 0;
    for (uint32_t k = 0; k + 1 < present; ++k) {
        uint32_t f2[2];
        uint16_t id2[2];
        for (int t = 0; t < 2; ++t) {
            bool take_leaf;
            if (lq < n && iq < made) take_leaf = (uint32_t)(sorted[lq] >> 32) <= ifreq[iq];
            else take_leaf = lq < n;
            if (take_leaf) { f2[t] = (uint32_t)(sorted[lq] >> 32); id2[t] = (uint16_t)(sorted[lq] & 0xffffu); ++lq; }
            else { f2[t] = ifreq[iq]; id2[t] = (uint16_t)(n + iq); ++iq; }
        }
        ifreq[made] = f2[0] + f2[1];
        lc[n + made] = id2[0];
        rc[n + made] = id2[1];
        ++made;
    }
    int root = (int)(n + present) - 2;
    uint32_t mx = 0;
    for (int i = root; i >= (int)n; --i) {  // huffman.cpp:73-79
        uint8_t l = (uint8_t)(len[i] + 1);
        len[lc[i]] = l; code[lc[i]] = code[i] << 1;
        len[rc[i]] = l; code[rc[i]] = (code[i] << 1) | 1;
        if (l > mx) mx = l;
    }
    *max_len_out = mx;
    return root;
}
// blob (huffman.cpp:88-118): root_id, n_symbols, min_len, pre-order bits.  Returns its length.
NGSC_HD uint32_t huffman_blob(uint32_t n, int root, uint32_t min_len, const uint16_t *lc, const uint16_t *rc, uint16_t *stack, uint8_t *blob)
{
    uint32_t bits_per_id = 0;
    while ((2u << bits_per_id) <= n) ++bits_per_id;
    if (n & (n - 1)) ++bits_per_id;
    ByteWriter w(blob);
    w.word((uint32_t)root);
    w.word(n);
    w.byte((uint8_t)min_len);
    int sp = 0;
    stack[sp++] = (uint16_t)root;
    while (sp) {
        uint16_t node = stack[--sp];
        if (node < n) {
            w.put(1, 1);
            w.put(node, bits_per_id);
        } else {
            w.put(0, 1);
            stack[sp++] = rc[node];
            stack[sp++] = lc[node];
        }
    }
    w.align();
    return (uint32_t)w.nbytes();
}
// everything by one thread (the CPU harness)
NGSC_HD uint32_t huffman_sequential(const uint64_t *sorted, uint32_t n, uint32_t *ifreq, uint16_t *lc, uint16_t *rc,
                                    uint32_t *code, uint8_t *len, uint16_t *stack, uint8_t *blob,
                                    uint32_t *max_len_out)
{
    const uint32_t zeros = huffman_zeros(sorted, n);
    huffman_clear(code, len, n);
    const int root = huffman_core(sorted, n, zeros, ifreq, lc, rc, code, len, max_len_out);
    return huffman_blob(n, root, huffman_min_len(len, n), lc, rc, stack, blob);
}

// The tree of a ONE-symbol alphabet (a numeric title field whose delta never changes): what the tree kernels would build for it,
// by one thread - the field finalisation writes it, so that such a field does not make the tree launch carry its LDS (round 4).
NGSC_HD uint32_t huffman_single_symbol(uint32_t *code_len_out, uint8_t *blob /* >= 16 bytes */)
{
    const uint64_t sorted[1] = {0};
    uint32_t ifreq[1] = {0}, code[2] = {0, 0}, max_len = 0;
    uint16_t lc[2] = {0, 0}, rc[2] = {0, 0}, stack[4] = {0, 0, 0, 0};
    uint8_t len[2] = {0, 0};
    const uint32_t n = huffman_sequential(sorted, 1, ifreq, lc, rc, code, len, stack, blob, &max_len);
    *code_len_out = (code[0] << 5) | len[0];
    return n;
}

// =====================================================================================================
// Sub-block cuts for one virtual rank (phyNGSC.cpp:113-124, 168, 303-318, 745-755)
// =====================================================================================================
NGSC_HD uint32_t lower_bound_u32(const uint32_t *a, uint32_t n, uint64_t key)
{
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Two lower bounds, a[ra] >= ka and b[rb] >= kb, searched in lockstep over the index window [lo, lo + span): the
// two loads of a step are independent, which halves the latency chain of the cut planner (its searches are the
// serial part of the index stage).  Results are clamped to the window.
NGSC_HD void lower_bound_pair(const uint32_t *a, uint32_t na, uint64_t ka, const uint32_t *b, uint32_t nb, uint64_t kb,
                              uint32_t lo, uint32_t span, uint32_t &ra, uint32_t &rb)
{
    uint32_t la = lo < na ? lo : na, ha = (uint64_t)lo + span < na ? lo + span : na;
    uint32_t lb = lo < nb ? lo : nb, hb = (uint64_t)lo + span < nb ? lo + span : nb;
    while (la < ha || lb < hb) {
        const uint32_t ma = la + ((ha - la) >> 1), mb = lb + ((hb - lb) >> 1);
        const uint32_t va = la < ha ? a[ma] : 0u, vb = lb < hb ? b[mb] : 0u;
        if (la < ha) { if ((uint64_t)va < ka) la = ma + 1; else ha = ma; }
        if (lb < hb) { if ((uint64_t)vb < kb) lb = mb + 1; else hb = mb; }
    }
    ra = la;
    rb = lb;
}

struct RankGeom {
    int64_t region, wr_start, wr_end, bsize;
    uint32_t overlap;
};

NGSC_HD RankGeom rank_geometry(uint64_t file_size, uint32_t P, uint32_t r)
{
    RankGeom g;
    g.region = (int64_t)(file_size / P);
    g.wr_start = (int64_t)r * g.region;
    g.overlap = OVERLAP;
    g.wr_end = (r != P - 1) ? g.wr_start + g.region + g.overlap - 1 : (int64_t)file_size - 1;
    g.bsize = (int64_t)WINDOW;
    if (g.region < g.bsize) {
        g.bsize = g.wr_end - g.wr_start + 1;
        if (r == P - 1) g.overlap = 0;
    }
    return g;
}

// te/ls are offsets inside the job buffer whose first byte is file offset buf_off; ls has n_rec+1 entries.
// first_abs: file offset of the rank's first record (from the sync heuristic).  Returns an error code.
// Search policy of plan_rank(): one thread (host harness, reference semantics) ...
struct ScalarSearch {
    NGSC_HD uint32_t one(const uint32_t *a, uint32_t n, uint64_t key) const { return lower_bound_u32(a, n, key); }
    NGSC_HD uint32_t at(const uint32_t *a, uint32_t idx) const { return a[idx]; }
    NGSC_HD void pair(const uint32_t *a, uint32_t na, uint64_t ka, const uint32_t *b, uint32_t nb, uint64_t kb, uint32_t lo,
                      uint32_t span, uint32_t &ra, uint32_t &rb, uint32_t /* hint */ = 0) const
    {
        lower_bound_pair(a, na, ka, b, nb, kb, lo, span, ra, rb);
    }
};
// (... or a whole wave probing 64 positions per round, see kernels_index.hip)

// buf_end_abs: file offset behind the job buffer's last byte, for a rank that is walked in several parts: a window is
// planned only while every record that starts inside it is complete in the buffer (or the buffer reaches the end of
// the rank's region); the rest is left to the next part (RankPlan::next_pos ...).  0: the buffer holds the whole rank.
template <class Search>
NGSC_HD int plan_rank_with(const Search &search, uint64_t file_size, uint32_t P, uint32_t r, uint64_t buf_off, const uint32_t *te,
                           const uint32_t *ls, uint32_t n_rec_total, uint64_t first_abs, Cut *cuts, uint32_t max_cuts,
                           RankPlan *plan, const PlanResume *from = nullptr, uint64_t buf_end_abs = 0)
{
    RankGeom g = rank_geometry(file_size, P, r);
    uint32_t n_cuts = 0;
    plan->n_cuts = 0;
    plan->overlap_used = (int32_t)((int64_t)first_abs - g.wr_start);
    plan->done = 0;
    uint32_t cur = search.one(ls, n_rec_total + 1, first_abs - buf_off);
    plan->first_rec = cur;
    if (cur > n_rec_total || (uint64_t)ls[cur] + buf_off != first_abs) return ERR_FORMAT;
    int64_t bytes_read = 0, win_pos = g.wr_start, bsize = g.bsize;
    uint32_t overlap = g.overlap;
    if (from && from->resume) {
        bytes_read = (int64_t)from->bytes_read;
        win_pos = (int64_t)from->next_pos;
        bsize = (int64_t)from->bsize;
        overlap = from->overlap;
    }
    const uint64_t rank_end = (uint64_t)g.wr_end + 1 < file_size ? (uint64_t)g.wr_end + 1 : file_size;
    const bool whole = buf_end_abs == 0 || buf_end_abs >= rank_end;       // the buffer reaches the end of the rank's region
    const uint64_t indexed_end = (uint64_t)ls[n_rec_total] + buf_off;     // behind the last complete record of the buffer
    uint32_t hint = 0;   // records of the previous window: where this window's bounds are expected (a search hint, nothing else)
    while (bytes_read < g.region) {
        if (!whole && (uint64_t)(win_pos + bsize) > indexed_end) {        // this window waits for the next part
            plan->next_pos = (uint64_t)win_pos; plan->bytes_read = (uint64_t)bytes_read; plan->bsize = (uint64_t)bsize;
            plan->overlap = overlap;
            return OK;
        }
        if (cur >= n_rec_total) return ERR_FORMAT;
        int64_t wend = win_pos + bsize, stop_at = win_pos + bsize - (int64_t)overlap;
        uint32_t last = cur;
        // candidates after the first record need their title end inside the window (phyNGSC.cpp:268, 303-306)
        // A window holds at most 2^21 records (8 MiB, >= 4 bytes each), so both bounds lie in [cur, cur + 2^21 + 4);
        // a bound below the window clamps to cur, which the tests below treat like any value <= cur.
        int64_t wend_rel = wend - (int64_t)buf_off, stop_rel = stop_at - (int64_t)buf_off;
        uint32_t m, lb;
        search.pair(te, n_rec_total, wend_rel > 0 ? (uint64_t)wend_rel : 0, ls, n_rec_total + 1,
                    stop_rel > 0 ? (uint64_t)stop_rel : 0, cur, (1u << 21) + 4, m, lb, hint);  // te[m-1] < wend
        if (m > cur + 1) {
            // first record (other than `cur`, which is never tested: :295-300) whose next offset >= stop_at
            uint32_t j = lb > 0 ? lb - 1 : 0;
            if (j < cur + 1) j = cur + 1;
            last = j < m - 1 ? j : m - 1;
        }
        uint32_t cnt = last - cur + 1;
        hint = cnt;
        if (cnt > MAX_RECORDS) return ERR_TOOMANY;
        if (n_cuts >= max_cuts) return ERR_NOMEM;
        Cut &c = cuts[n_cuts++];
        plan->n_cuts = n_cuts;   // (stored, never read back: the chain of windows does not wait for memory here)
        c.rank = r; c.rec_lo = cur; c.n_rec = cnt; c.overlap = overlap;
        c.win_pos = (uint64_t)win_pos; c.wsize = (uint64_t)bsize;
        int64_t next_abs = (int64_t)search.at(ls, last + 1) + (int64_t)buf_off;
        bytes_read = next_abs - g.wr_start;   // :745-746
        win_pos = next_abs;
        if (win_pos + bsize > g.wr_end) {      // :748-755
            if (r == P - 1) overlap = 0;
            bsize = g.wr_end - win_pos;
        }
        cur = last + 1;
    }
    plan->done = 1;
    plan->next_pos = (uint64_t)win_pos; plan->bytes_read = (uint64_t)bytes_read; plan->bsize = (uint64_t)bsize; plan->overlap = overlap;
    return OK;
}

NGSC_HD int plan_rank(uint64_t file_size, uint32_t P, uint32_t r, uint64_t buf_off, const uint32_t *te,
                      const uint32_t *ls, uint32_t n_rec_total, uint64_t first_abs, Cut *cuts, uint32_t max_cuts,
                      RankPlan *plan)
{
    return plan_rank_with(ScalarSearch(), file_size, P, r, buf_off, te, ls, n_rec_total, first_abs, cuts, max_cuts, plan);
}

// =====================================================================================================
// Block header / footer bit layouts (host side of the container: tasks.cpp:1104-1200)
// =====================================================================================================
NGSC_HD uint32_t header_size(uint32_t bewr, uint32_t beso, uint32_t nosb)
{
    return (bewr + 6 + 12 + beso * nosb + 5 + 2 + 7) / 8;  // structures.h:323-333
}

}  // namespace ngsc
