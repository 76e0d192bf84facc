// ngsc_kernels.h - launch entry points of the HIP kernels (internal to libngsc_hip.so).
#pragma once
#include "ngsc_common.h"
#include <hip/hip_runtime.h>

namespace ngsc {

// ---- kernels_index.hip ----
uint32_t index_num_tiles(uint32_t len);
void launch_sync_first(hipStream_t st, const uint8_t *buf, uint32_t len, uint64_t buf_off, uint64_t file_size, uint32_t P,
                       uint32_t rank_lo, uint32_t n_ranks, uint64_t *first_abs, uint32_t *start_rel, int32_t *err, uint64_t resume_abs,
                       void *zero, size_t zero_bytes);   // (+ clears the step's small state)
void launch_nl_count(hipStream_t st, const uint8_t *buf, uint32_t len, const uint32_t *start_rel, uint32_t *tile_cnt, uint32_t *nl_mask,
                     bool at_file_end);
size_t index_mask_words(uint32_t len);
void launch_scan_tiles(hipStream_t st, uint32_t *tile_cnt, uint32_t n_tiles, uint32_t *total_lines);
void launch_nl_write(hipStream_t st, const uint32_t *nl_mask, uint32_t len, const uint32_t *start_rel, const uint32_t *tile_off,
                     uint32_t *te, uint32_t *se, uint32_t *ls, uint32_t rec_cap);
void launch_cuts_validate(hipStream_t st, uint64_t file_size, uint32_t P, uint32_t rank_lo, uint32_t n_ranks, uint64_t buf_off,
                          const uint32_t *te, const uint32_t *se, const uint32_t *ls, uint32_t n_rec, const uint32_t *total_lines,
                          const uint64_t *first_abs, Cut *cuts, uint32_t max_cuts, RankPlan *plans, uint32_t *max_q,
                          uint32_t *max_rec, uint32_t *max_title, int32_t *err, const uint8_t *tail_buf, const PlanResume &from,
                          uint64_t buf_end_abs, const uint32_t *tile_off, uint32_t n_tiles);

// ---- kernels_sq.hip ----
void launch_sq_stats_reg(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t max_rec_bytes, uint32_t n_rec_total);
void launch_sq_format_qtab(hipStream_t st, const Arena &A);   // the quality code tables as measure and pack keep them in LDS
void launch_sq_measure_reg(hipStream_t st, const Arena &A, uint32_t max_rec);
size_t sq_qfmt_words(uint32_t qrows);   // dwords per sub-block of Arena::qfmt (a multiple of 4)
size_t sq_qfmt_bytes(uint32_t qrows, uint32_t n_sb);   // the whole of Arena::qfmt: dword tables, then byte tables
void launch_sq_pack_reg(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t stage_q_words, uint32_t stage_d_words);

// ---- kernels_cs.hip (colour-space reads decoded into a copy of the job's input) ----
void launch_cs(hipStream_t st, const Arena &A, uint32_t max_rec, const uint8_t *orig, uint8_t *copy, uint32_t *se_w);

// ---- kernels_stats.hip ----
void launch_seq_qual_stats(hipStream_t st, const Arena &A, uint32_t max_rec);
void launch_sq_finalize(hipStream_t st, const Arena &A, bool small_only = false);
void launch_field_finalize(hipStream_t st, const Arena &A, bool quiet = false);
void launch_title_hist(hipStream_t st, const Arena &A, uint32_t max_rec, bool use_tv);

// ---- kernels_huffman.hip ----
void launch_huffman_sq(hipStream_t st, const Arena &A, bool big_expected = false, bool with_big = true);   // quality / DNA trees (with_big: both instances)
void launch_huffman_title(hipStream_t st, const Arena &A);   // text-field and numeric-field trees of the title

// ---- kernels_pack.hip ----
void launch_measure_sq(hipStream_t st, const Arena &A, uint32_t max_rec);
void launch_scan_sb(hipStream_t st, const Arena &A);
void launch_scan_tile(hipStream_t st, const Arena &A, uint32_t R);   // (lane-per-record path: per-tile bit counts)
void launch_sizes_offsets(hipStream_t st, const Arena &A, uint32_t *qrow_off, uint64_t out_cap, uint64_t *total_out);
void launch_layout(hipStream_t st, const Arena &A, uint32_t R, uint32_t *qrow_off, uint64_t out_cap, uint64_t *total_out, uint32_t *done,
                   void *zero_out = nullptr, uint64_t cleared = 0);   // scan_tile + sizes + offsets
void launch_zero_output(hipStream_t st, void *d_out, const uint64_t *total_ptr, uint64_t out_cap, uint64_t cleared = 0);   // cleared: leading bytes that are zero already
void launch_zero_range(hipStream_t st, void *d_out, uint64_t bytes);
void launch_init_sb(hipStream_t st, const Arena &A, const Cut *cuts, const RankPlan *plans, uint32_t n_ranks, uint32_t max_cuts, uint32_t R,
                    const ShapeCaps &caps, const uint32_t *total_lines, const uint32_t *max_title, const uint32_t *max_q, const uint32_t *max_rec_bytes);
void launch_write_headers(hipStream_t st, const Arena &A, const uint32_t *qrow_off);
void launch_pack_lentable(hipStream_t st, const Arena &A, uint32_t max_rec);
void launch_pack_sq(hipStream_t st, const Arena &A, uint32_t max_rec);

// ---- kernels_title.hip (register-resident title path) ----
void launch_rec0_fields(hipStream_t st, const Arena &A);
void launch_title_stats_reg(hipStream_t st, const Arena &A, uint32_t max_rec);
void launch_title_stats_swar(hipStream_t st, const Arena &A, uint32_t max_rec, uint32_t max_title);
void launch_title_measure_reg(hipStream_t st, const Arena &A, uint32_t max_rec, bool swar);
void launch_title_pack_reg(hipStream_t st, const Arena &A, uint32_t max_rec);

// ---- kernels_title_wide.hip (titles of more than MAX_FIELDS separators) ----
void launch_title_wide_stats(hipStream_t st, const Arena &A, uint32_t max_rec);     // field count, statistics, field decisions, histograms
void launch_title_wide_measure(hipStream_t st, const Arena &A, uint32_t max_rec);
void launch_title_wide_pack(hipStream_t st, const Arena &A, uint32_t max_rec);

}  // namespace ngsc
