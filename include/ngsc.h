/*
 * ngsc.h - C ABI of libngsc_hip.so, the MI355X (gfx950) FASTQ sub-block compression engine.
 *
 * Drop-in boundary for the hot path of pcdslab/PHYNGSC (file:line below are into that repository).
 * The reference exposes no FFI; what its main() calls per sub-block is the tasks.h function set
 *   AnalyzeTitleFields / AnalyzeDNA / AnalyzeQuality / StoreTitle / StoreDNA / StoreQuality
 *   (tasks.h:20-42) on top of BitStream (bit_stream.h) and HuffmanEncoder (huffman.h),
 * driven by the loop phyNGSC.cpp:168-906, plus MakeHeader / MakeFooter (tasks.h:61-65) for the
 * container.  This library replaces exactly that:
 *
 *   ngsc_compress_device()   <- phyNGSC.cpp:113-166 (region split, first-record sync), :254-331
 *                               (record index), :342-742 (field split, prepass, Analyze*, Store*,
 *                               info header), :745-755 (sub-block cuts), :793-840 (stream concat),
 *                               for all sub-blocks of a range of virtual ranks, on the GPU
 *   ngsc_plan_blocks()       <- phyNGSC.cpp:842-928 (8 MiB block assembly, split sub-blocks)
 *   ngsc_make_header()       <- MakeHeader, tasks.cpp:1179-1200
 *   ngsc_make_footer()       <- MakeFooter, tasks.cpp:1104-1176
 *   ngsc_compress_file()     <- main(), phyNGSC.cpp:30-1076, `<in.fastq> <out.ngsc>` semantics with
 *                               P virtual ranks instead of `mpiexec -np P`
 *
 * Conventions: plain pointers and sizes, no exceptions, every function returns 0 or a negative
 * NGSC_ERR_* code and never exits the process (the reference printf()s and exit()s).  Device
 * pointers must be 16-byte aligned.  Input buffers are const: the in-place N-transfer rewrite of
 * the reference (phyNGSC.cpp:573-588) is recomputed on the fly instead.  A context is bound to one
 * GPU and one stream; use one context per GPU (contexts are independent and thread-compatible).
 * There is no CPU fallback: without a HIP device every compute entry point fails with NGSC_ERR_HIP.
 */
#ifndef NGSC_H
#define NGSC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGSC_VERSION 100

enum {
    NGSC_OK = 0,
    NGSC_ERR_FORMAT = -1,      /* not 4-line FASTQ with bare '+' lines / quality length != sequence length */
    NGSC_ERR_FIELDS = -2,      /* title separator count differs inside a sub-block (reference: warning, then UB) */
    NGSC_ERR_TOOMANY = -3,     /* > 100000 records in one sub-block (reference: warning + overflow) */
    NGSC_ERR_COLORSPACE = -4,  /* decoder only: a colour-space (SOLiD) sub-block, which the format stores lossily (phyNGSC.cpp:531); the compressor handles such reads */
    NGSC_ERR_RANGE = -5,       /* a varying numeric title field with a value beyond INT32_MAX (reference: negative int32 and an endless loop, or
                                * - digits beyond 32 bits - the value stored modulo 2^32), or a delta range >= 2^31 */
    NGSC_ERR_NOMEM = -6,
    NGSC_ERR_ARG = -7,
    NGSC_ERR_UNSUPPORTED = -8, /* engine limit: a first title or a non-constant text field's first value > 512 B, a Huffman code > 26 bits */
    NGSC_ERR_HIP = -9,         /* HIP runtime / device failure (ngsc_last_hip_error()) */
    NGSC_ERR_CAPACITY = -10,   /* output buffer too small: ngsc_result_bytes() tells the size needed */
    NGSC_ERR_IO = -11
};

#define NGSC_BLOCK_SIZE (8u << 20) /* WRITE_BUFFER_SIZE, defs.h:21 */

typedef struct ngsc_ctx ngsc_ctx;

/* One compressed sub-block (one iteration of phyNGSC.cpp:168-906). Bytes live at d_out + out_off. */
typedef struct {
    uint32_t rank;       /* WRID of the virtual rank that owns it */
    uint32_t n_records;
    uint64_t win_pos;    /* file offset of its read window */
    uint64_t wsize;      /* window size (r_buffer_size) */
    uint64_t out_off;
    uint32_t info_len, title_len, qual_len, dna_len; /* stream order info|title|quality|DNA */
    uint32_t max_qlen, max_slen, nsym, nqual, flags, n_fields;
    uint32_t n_bases;    /* sum of read lengths: sequence bytes == quality bytes of the sub-block */
    uint32_t reserved;
} ngsc_subblock_t;

typedef struct {
    uint32_t rank;
    uint32_t n_subblocks;
    uint32_t first_subblock; /* index into ngsc_result_subblocks() */
    int32_t overlap_used;    /* wr_ov_used, phyNGSC.cpp:160 (footer OV) */
    uint64_t stream_off;     /* the rank's sub-blocks are contiguous in d_out from here */
    uint64_t stream_bytes;
} ngsc_rank_t;

/* One output block of a rank: header bytes followed by stream[stream_off, stream_off+payload). */
typedef struct {
    uint32_t wrid;
    uint32_t size;       /* header_len + payload: 8 MiB except the rank's last block */
    uint32_t header_len;
    uint32_t payload;
    uint64_t stream_off; /* offset inside the rank's sub-block stream */
    uint8_t header[288]; /* MakeHeader bytes: NOSB <= 63 entries of <= 32 bits */
} ngsc_block_t;

typedef struct {
    const char *name;
    double ms;        /* accumulated device time of this kernel (HIP events on the context's stream) */
    uint32_t launches;
} ngsc_kernel_time_t;

int ngsc_version(void);
const char *ngsc_strerror(int code);

int ngsc_ctx_create(int device_id, ngsc_ctx **out);
void ngsc_ctx_destroy(ngsc_ctx *ctx);
const char *ngsc_last_hip_error(ngsc_ctx *ctx);
int ngsc_ctx_device(ngsc_ctx *ctx); /* the HIP device the context was created on */
/* Gives the decoder's working memory back to the device (per-sub-block scratch and the speculative decoder's hop table stay with the
 * context between ngsc_decompress_* calls: several GB after a large file).  The compressor does this by itself when it finds no room. */
int ngsc_ctx_trim(ngsc_ctx *ctx);

/* Per-kernel device timing (hipEvent pairs around every launch). Off by default. */
int ngsc_set_profiling(ngsc_ctx *ctx, int enable);
int ngsc_profile_only(ngsc_ctx *ctx, const char *kernel_name); /* time just this kernel (NULL: all); event records
                                                                  * between launches cost ~10 us each */
int ngsc_reset_kernel_times(ngsc_ctx *ctx);
uint32_t ngsc_get_kernel_times(ngsc_ctx *ctx, ngsc_kernel_time_t *out, uint32_t cap);

/*
 * Compresses the virtual ranks [rank_lo, rank_hi) of a P-rank run over a FASTQ file of file_size
 * bytes.  d_buf holds the file bytes [buf_off, buf_off + buf_len) in device memory; it must start
 * at or before the first rank's region and extend past the last rank's region by the reference's
 * overlap (500 B) plus one record, or to the end of the file.  buf_len < 4 GiB.
 * hip_stream: a hipStream_t (e.g. torch's current stream) or NULL for the context's own stream.
 * On return the stream has been synchronised and d_out[0 .. ngsc_result_bytes()) holds, rank after
 * rank, every sub-block's info|title|quality|DNA bytes.
 */
int ngsc_compress_device(ngsc_ctx *ctx, const void *d_buf, uint64_t buf_len, uint64_t buf_off, uint64_t file_size,
                         uint32_t P, uint32_t rank_lo, uint32_t rank_hi, void *d_out, uint64_t out_cap,
                         void *hip_stream);

/*
 * One rank walked in several calls: the reference streams a rank's region through 8 MiB windows whatever its size
 * (phyNGSC.cpp:113-128, 168, 745-755); a region of 4 GiB and more does not fit one ngsc_compress_device() job.
 * *state: zero-initialised for the rank's first part (d_buf then starts at or before the rank's region, as above);
 * afterwards it holds where the rank's loop stands - next_pos is the file offset the next part's buffer has to start at
 * or before - and done == 1 once the rank is finished.  A call compresses every window whose records are complete in the
 * buffer (all of them when the buffer reaches the end of the region or of the file) and leaves the results as
 * ngsc_compress_device() does: one rank entry, the sub-blocks of this part.  NGSC_ERR_UNSUPPORTED: not even one window fits.
 */
typedef struct {
    uint64_t next_pos;     /* file offset of the rank's next window = of its first record */
    uint64_t bytes_read;   /* p_bytes_read so far (phyNGSC.cpp:745-746) */
    uint64_t bsize;        /* size of the next window (phyNGSC.cpp:748-755) */
    uint32_t overlap;      /* overlap in force for it */
    uint32_t done;
    int32_t overlap_used;  /* wr_ov_used of the rank (phyNGSC.cpp:160), set by its first part */
    uint32_t resume;       /* 0 before the first part */
} ngsc_rank_part_t;
int ngsc_compress_rank_part(ngsc_ctx *ctx, const void *d_buf, uint64_t buf_len, uint64_t buf_off, uint64_t file_size,
                            uint32_t P, uint32_t rank, ngsc_rank_part_t *state, void *d_out, uint64_t out_cap,
                            void *hip_stream);

uint64_t ngsc_result_bytes(ngsc_ctx *ctx);
uint32_t ngsc_result_num_subblocks(ngsc_ctx *ctx);
uint32_t ngsc_result_num_ranks(ngsc_ctx *ctx);
const ngsc_subblock_t *ngsc_result_subblocks(ngsc_ctx *ctx);
const ngsc_rank_t *ngsc_result_ranks(ngsc_ctx *ctx);

/*
 * The inverse path: n_sb sub-blocks (each info | title | quality | DNA as ngsc_compress_device() writes them; sb_off /
 * sb_len: their byte offsets and lengths inside d_sub, host arrays) -> the FASTQ text of their records, sub-block
 * after sub-block, in d_text; text_off[0 .. n_sb] (host) receives where each sub-block's text starts and the total.
 * Replaces the reference's uncalled decoder fragments (FetchTitleHeader / FetchTitleBody / FetchDNA / FetchQuality,
 * tasks.cpp:625-1101; LoadTree / Decode, huffman.cpp:122-219).  A sub-block is one bit string whose streams have no
 * stored lengths, so the unit of parallel work is the sub-block: one wave each for the serial streams, then all threads
 * for the record assembly.  Colour-space sub-blocks (FLAG_USE_DELTA) are stored lossily by the format and are refused
 * (NGSC_ERR_COLORSPACE).  n_records: nullable, records per sub-block.  The stream is synchronised on return.
 */
int ngsc_decompress_device(ngsc_ctx *ctx, const void *d_sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb,
                           void *d_text, uint64_t text_cap, uint64_t *text_off, uint32_t *n_records, void *hip_stream);

/* ---- container (host side) ------------------------------------------------------------------- */
/* Splits one rank's sub-block sizes into 8 MiB blocks (phyNGSC.cpp:842-928). */
int ngsc_plan_blocks(uint32_t P, uint32_t wrid, const uint32_t *sb_sizes, uint32_t n_sb, ngsc_block_t *blocks,
                     uint32_t cap, uint32_t *n_blocks);
uint32_t ngsc_make_header(uint8_t *out, uint32_t wrid, uint32_t bewr, uint32_t bhs, uint32_t beso, uint32_t bcss,
                          const uint32_t *sbol, uint32_t nosb);
/* cbo: WRID of every block in physical order; lb_sizes/overlaps: one entry per rank. out >= 64 + bytes(cbo). */
uint32_t ngsc_make_footer(uint8_t *out, uint32_t P, uint64_t fastq_size, uint32_t n_blocks, uint32_t n_subblocks,
                          const int32_t *overlaps, const uint32_t *cbo, const uint32_t *lb_sizes);

/*
 * Block order of the files this library writes (the reference's is arrival order, different from run to run): ranks in
 * groups of consecutive ranks whose regions make up at most NGSC_ORDER_GROUP_BYTES of input (floor(512 MiB / (file_size /
 * P)) ranks, at least one), group after group; inside a group round k = k-th block of each of its ranks, ranks
 * ascending.  A file of up to 512 MiB is one group.  A group's place depends only on the groups in front of it, so the
 * driver writes each group while the next one is compressed.
 */
#define NGSC_ORDER_GROUP_BYTES (512ull << 20)

/*
 * `<in.fastq> <out.ngsc>` (phyNGSC.cpp:61-108) with P virtual ranks on this context's GPU: reads the
 * file job by job, uploads, compresses, and writes each group's blocks at its place (above) while the next
 * job runs; footer at the end.  seconds_out (nullable, 3 doubles): read, H2D+device+D2H+write (overlapped), footer.
 */
int ngsc_compress_file(ngsc_ctx *ctx, const char *in_path, const char *out_path, uint32_t P, double *seconds_out);

/*
 * The same job spread over n_ctx contexts (normally one per GPU of the node) from ONE process: the
 * groups of the block order are dealt to the contexts in turn (group g to context g % n_ctx; P % n_ctx == 0),
 * each context works on its own host threads and device, and every group is written at its place --
 * so the file is byte-identical for every n_ctx.  This replaces the reference's `mpiexec -np P`
 * fan-out (phyNGSC.cpp:45-57, 113-116) when one host drives several GPUs; there is no device-to-
 * device traffic, only the per-rank block lists meet on the host (phyNGSC.cpp:875-1057).
 * Files and rank regions of any size: a context walks its groups in jobs of consecutive ranks, and a rank whose
 * region is larger than a job in parts (ngsc_compress_rank_part), reading each job's byte range from the file
 * itself; host memory holds two jobs' input and output, not the file.
 */
int ngsc_compress_file_multi(ngsc_ctx **ctxs, uint32_t n_ctx, const char *in_path, const char *out_path, uint32_t P,
                             double *seconds_out);

/*
 * ---- one process, several GPUs, one collective ---------------------------------------------------------------------
 * A team = one host thread per context (one context per GPU) + an RCCL communicator created INSIDE this process
 * (ncclCommInitAll over the contexts' devices; ranks talk over xGMI, no launcher).  It replaces the reference's only two
 * exchanges, MPI_Gather of {n_blocks, n_subblocks, last_block_size, wr_overlap} (phyNGSC.cpp:934-964) and MPI_Gatherv
 * of the block timestamps (:994): after compressing its shard of virtual ranks - no data-path collective - every
 * member all-gathers one row of u64 per virtual rank,
 *     row = { n_blocks, n_subblocks, overlap_used (sign-extended), size of block 0, 1, ... }   (row_len words, zero padded)
 * and every member then holds all P rows; file offsets follow from a local prefix sum over the deterministic block
 * order (ngsc_rows_to_offsets).  RCCL is bound at run time (dlopen librccl.so.1; NGSC_RCCL=0 or NGSC_RCCL_LIB=path
 * override).  Without it, or when two contexts share a device, the team still works: ngsc_team_rccl_ranks() == 0,
 * ngsc_team_describe() says why, and the rows meet in host memory.
 */
typedef struct ngsc_team ngsc_team;
typedef struct {
    const void *d_buf;          /* file bytes [buf_off, buf_off + buf_len) on the member's GPU */
    uint64_t buf_len, buf_off;
    uint32_t rank_lo, rank_hi;  /* member i of n owns the virtual ranks [i * P / n, (i + 1) * P / n) */
    void *d_out;
    uint64_t out_cap;
} ngsc_shard_t;
int ngsc_team_create(ngsc_ctx **ctxs, uint32_t n_ctx, ngsc_team **out);
void ngsc_team_destroy(ngsc_team *team);
uint32_t ngsc_team_size(ngsc_team *team);
uint32_t ngsc_team_rccl_ranks(ngsc_team *team);   /* ncclCommCount() of the communicator; 0 = no RCCL, host memory */
const char *ngsc_team_describe(ngsc_team *team);  /* which collective a step uses, and why if it is not RCCL */
/* One step: member i compresses shards[i] on its own thread and GPU (ngsc_compress_device; the results stay in its
 * context), plans its ranks' 8 MiB blocks, then the rows are all-gathered.  rows: P x row_len words, identical on every
 * member (checked).  row_len >= 3 + the largest block count of a rank, else NGSC_ERR_CAPACITY.  shard_ms (nullable):
 * host time of each member's part. */
int ngsc_team_compress(ngsc_team *team, const ngsc_shard_t *shards, uint64_t file_size, uint32_t P, uint64_t *rows,
                       uint32_t row_len, double *shard_ms);
/* Host time each member spent in the exchange of the last ngsc_team_compress (vote + H2D + collective + D2H; it includes waiting
 * for the slowest member's compression): ms[team size]. */
void ngsc_team_exchange_ms(ngsc_team *team, double *ms);
/* The bare collective: member i contributes send[i * count .. ), receives all n * count words; recv: n x (n * count). */
int ngsc_team_allgather_u64(ngsc_team *team, const uint64_t *send, uint32_t count, uint64_t *recv);
/* The same from a thread of the caller's that acts as member `member` (ngsc_compress_file_multi's workers: every member calls, in
 * the same order): send[count] -> recv[n * count].  status: 0, or the member's error - the members vote in host memory in front of
 * the collective, and if any of them reports an error (or ngsc_team_abort() was called) NOBODY enters it and all return that
 * code: a member that gave up cannot leave the others waiting in RCCL.  Replaces MPI_Gather / MPI_Gatherv, phyNGSC.cpp:964, 994. */
int ngsc_team_member_allgather(ngsc_team *team, uint32_t member, int status, const uint64_t *send, uint32_t count, uint64_t *recv);
/* A member that cannot go on (I/O error, allocation failure): everyone waiting in the team's collective, now or later, returns `code`. */
void ngsc_team_abort(ngsc_team *team, int code);
/* Offsets from rows (host): offsets[r * (row_len - 3) + k] = file offset of rank r's k-th block; cbo: WRID of every block
 * in file order (nullable); *end: where the footer goes. */
int ngsc_rows_to_offsets(const uint64_t *rows, uint32_t P, uint32_t row_len, uint64_t file_size, uint64_t *offsets, uint32_t *cbo,
                         uint32_t cbo_cap, uint32_t *n_blocks, uint64_t *end);

/*
 * The reading side of the two bit layouts above (host, no GPU): ReadFooter (tasks.cpp:1203-1292) and the block header as
 * MakeHeader (tasks.cpp:1179-1200) wrote it.  footer: the footer_len bytes in front of the file's last two bytes (which hold
 * footer_len, big endian).  overlaps / lb_sizes: P entries (lb_sizes all zero when the footer stores none: LBES, every last
 * block then has the same size); cbo: n_blocks entries, the WRID of every block in file order.  Arrays may be NULL to fetch
 * the scalars first.  NGSC_ERR_FORMAT: not a footer / header; NGSC_ERR_CAPACITY: an array is too small.
 */
typedef struct {
    uint32_t P, n_blocks, n_subblocks, lbes;
    uint64_t fastq_size;
} ngsc_footer_t;
int ngsc_read_footer(const uint8_t *footer, uint64_t footer_len, ngsc_footer_t *out, int32_t *overlaps, uint32_t *lb_sizes,
                     uint32_t cap_ranks, uint32_t *cbo, uint32_t cap_blocks);
typedef struct {
    uint32_t wrid, header_len, n_subblocks, bcss; /* bcss bit 1: the first sub-block continues the previous block's last; bit 0:
                                                   * the last one continues in the rank's next block */
    uint32_t sb_len[63];                          /* bytes of each sub-block (piece) in this block */
} ngsc_block_header_t;
int ngsc_read_header(const uint8_t *block, uint64_t avail, uint32_t P, ngsc_block_header_t *out);

/*
 * `<in.ngsc> <out.fastq>`: the container read back (ReadFooter tasks.cpp:1203-1292; block headers as MakeHeader
 * tasks.cpp:1179-1200 wrote them; the reference has the fragments but no caller).  Ranks ascending, each rank's sub-blocks
 * through ngsc_decompress_device() in batches; rank r's text is written at r * (FS / P) + OV[r], so a record that two
 * ranks stored appears once and the file gets its exact size back.  Accepts any block order, the
 * reference's arrival order included.  seconds_out (nullable, 3 doubles): file reads, H2D + decode + D2H, writes.
 */
int ngsc_decompress_file(ngsc_ctx *ctx, const char *in_path, const char *out_path, double *seconds_out);

#ifdef __cplusplus
}
#endif
#endif
