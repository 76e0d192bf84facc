// decode.cpp - host side of the GPU decoder: ngsc_decompress_device() (include/ngsc.h), batches of sub-blocks through
// k_dec_streams / k_dec_assemble (kernels_decode.hip), and the container reader for `ngsc -d` (ReadFooter tasks.cpp:1203-1292,
// block headers tasks.cpp:1179-1200 read back).
#include "../../include/ngsc.h"
#include "ngsc_kernels.h"

#include <algorithm>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

using namespace ngsc;

namespace ngsc {
void launch_dec_head(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb, uint8_t *scratch,
                     uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap, void *out, void *ctl, bool allow_spec, uint32_t *summary);
void launch_dec_streams(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb, uint8_t *scratch,
                        uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap, void *out, void *ctl, uint16_t *hop, const uint64_t *hop_off,
                        uint32_t max_tiles, uint32_t spec_max_q, uint32_t spec_max_nq);
size_t dec_ctl_bytes();
uint32_t dec_qspec_tile();
uint32_t dec_sub_spec(const void *h, uint32_t s);
void launch_dec_assemble(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, uint32_t n_sb, uint8_t *scratch, uint64_t stride,
                         uint32_t rec_cap, uint64_t tcap, uint64_t qcap, const void *in, const uint64_t *text_off, uint8_t *text);
size_t dec_sub_bytes();
}
extern "C" void *ngsc_ctx_scratch(ngsc_ctx *c, uint32_t slot, size_t bytes);   // engine.cpp
namespace ngsc {
int32_t dec_sub_err(const void *h, uint32_t s);
uint64_t dec_sub_text(const void *h, uint32_t s);
uint32_t dec_sub_records(const void *h, uint32_t s);
void dec_sub_ticks(const void *h, uint32_t s, uint64_t *t);
}

extern "C" {

// (the context's device; working memory is the context's and stays between calls)
int ngsc_decompress_device(ngsc_ctx *c, const void *d_sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb, void *d_text,
                           uint64_t text_cap, uint64_t *text_off, uint32_t *n_records, void *hip_stream)
{
    if (!c || !d_sub || !sb_off || !sb_len || !n_sb || !d_text || !text_off) return NGSC_ERR_ARG;
    if (hipSetDevice(ngsc_ctx_device(c)) != hipSuccess) return NGSC_ERR_HIP;
    hipStream_t st = (hipStream_t)hip_stream;   // nullptr: the default stream
    // scratch of one sub-block: a sub-block holds the records of one 8 MiB window (+ the record that straddles its end)
    const uint32_t rec_cap = MAX_RECORDS + 2;
    const uint64_t tcap = 10ull << 20, qcap = 0, pool = 4ull << 20;   // text area: value table, qualities, bases <= the window's text
    const uint64_t stride = ((4ull * rec_cap * 4 + 128 + tcap + pool) + 255) & ~255ull;
    // batches: at most 2048 sub-blocks (~15 MiB of scratch each) and - for the speculative quality decoder's hop table, 2 bytes
    // per BIT of a sub-block - at most 512 MiB of compressed bytes
    const bool spec = !(getenv("NGSC_DEC_SPEC") && !strcmp(getenv("NGSC_DEC_SPEC"), "0"));
    const uint64_t hop_budget = 512ull << 20;
    uint32_t batch = std::min<uint32_t>(n_sb, 2048);
    {   // ... and what the device has room for: at most half of its free memory goes into the per-sub-block scratch
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) {
            // (the hop table of the speculative quality decoder - 16 bytes per compressed byte of a batch - is budgeted with the scratch)
            uint64_t all_len = 0;
            for (uint32_t k = 0; k < n_sb; ++k) all_len += sb_len[k];
            const uint64_t per_sb = stride + (spec ? 16 * (all_len / n_sb + 1) : 0);
            const uint64_t fit = (uint64_t)free_b / 2 / per_sb;
            batch = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(batch, fit));
        }
    }
    uint64_t max_batch_bytes = 0;
    {   // the largest compressed size a batch of `batch` consecutive sub-blocks can have under the budget
        uint64_t run = 0;
        uint32_t cnt = 0, worst = batch;
        for (uint32_t k = 0, k0 = 0; k < n_sb; ++k) {
            run += sb_len[k]; ++cnt;
            while (cnt > 1 && (run > hop_budget || cnt > batch)) { run -= sb_len[k0++]; --cnt; }
            max_batch_bytes = std::max(max_batch_bytes, run);
            (void)worst;
        }
    }
    // working memory: kept in the context between calls (one carve for the small tables, one for the per-sub-block scratch,
    // one for the hop table); an allocation that fails - a shared or fragmented device - is tried again with half the batch
    const size_t rb = dec_sub_bytes();
    int rc = NGSC_OK;
    uint8_t *scratch = nullptr;
    for (;;) {
        scratch = (uint8_t *)ngsc_ctx_scratch(c, 0, stride * batch);
        if (scratch || batch == 1) break;
        batch = (batch + 1) / 2;
    }
    std::vector<uint8_t> h_res(rb * batch);
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_off = 0, o_toff = o_off + up(8ull * batch), o_hoff = o_toff + up(8ull * batch), o_len = o_hoff + up(8ull * batch);
    const size_t o_res = o_len + up(4ull * batch), o_ctl = o_res + up(rb * batch), o_sum = o_ctl + up(dec_ctl_bytes() * batch);
    uint8_t *tabs = scratch ? (uint8_t *)ngsc_ctx_scratch(c, 1, o_sum + 256) : nullptr;
    if (!scratch || !tabs) rc = NGSC_ERR_NOMEM;
    uint64_t *d_off = (uint64_t *)(tabs + o_off), *d_toff = (uint64_t *)(tabs + o_toff), *d_hoff = (uint64_t *)(tabs + o_hoff);
    uint32_t *d_len = (uint32_t *)(tabs + o_len), *d_sum = (uint32_t *)(tabs + o_sum);
    void *d_res = tabs + o_res, *d_ctl = tabs + o_ctl;
    uint16_t *d_hop = nullptr;
    if (!rc && spec) d_hop = (uint16_t *)ngsc_ctx_scratch(c, 2, 16ull * max_batch_bytes + 64);   // (no room: serial decode)
    uint64_t text_pos = 0;
    uint32_t n_spec = 0;
    for (uint32_t b0 = 0; b0 < n_sb && !rc;) {
        uint32_t nb = 0;
        uint64_t bbytes = 0;
        while (b0 + nb < n_sb && nb < batch && (nb == 0 || bbytes + sb_len[b0 + nb] <= hop_budget)) { bbytes += sb_len[b0 + nb]; ++nb; }
        std::vector<uint64_t> hoff(nb);
        uint64_t hpos = 0;
        uint32_t max_tiles = 1;
        for (uint32_t k = 0; k < nb; ++k) {
            hoff[k] = hpos;
            hpos += 8ull * sb_len[b0 + k];
            max_tiles = std::max<uint32_t>(max_tiles, (uint32_t)(8ull * sb_len[b0 + k] / dec_qspec_tile()) + 1);
        }
        const bool use_hop = d_hop && hpos <= 8ull * max_batch_bytes;
        if (hipMemcpyAsync(d_off, sb_off + b0, 8ull * nb, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_len, sb_len + b0, 4ull * nb, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_hoff, hoff.data(), 8ull * nb, hipMemcpyHostToDevice, st) != hipSuccess) { rc = NGSC_ERR_HIP; break; }
        uint32_t spec_sum[2] = {0, 0};   // longest read, largest quality alphabet among the sub-blocks the speculative kernels take
        if (hipMemsetAsync(d_sum, 0, 8, st) != hipSuccess) { rc = NGSC_ERR_HIP; break; }
        launch_dec_head(st, (const uint8_t *)d_sub, d_off, d_len, nb, scratch, stride, rec_cap, tcap, qcap, d_res, d_ctl, use_hop, d_sum);
        if (use_hop && (hipMemcpyAsync(spec_sum, d_sum, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)) {
            rc = NGSC_ERR_HIP;
            break;
        }
        // the hop table is cleared ("no hop known") only when some sub-block of the batch will be decoded speculatively: 16 bytes per
        // compressed byte - 4 GB for a batch of 36..300-base reads, none of which takes that path
        if (use_hop && spec_sum[0] && hipMemsetAsync(d_hop, 0xFF, 2ull * hpos, st) != hipSuccess) { rc = NGSC_ERR_HIP; break; }
        launch_dec_streams(st, (const uint8_t *)d_sub, d_off, d_len, nb, scratch, stride, rec_cap, tcap, qcap, d_res, d_ctl, use_hop ? d_hop : nullptr,
                           d_hoff, max_tiles, spec_sum[0], spec_sum[1]);
        if (hipMemcpyAsync(h_res.data(), d_res, rb * nb, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = NGSC_ERR_HIP;
            break;
        }
        std::vector<uint64_t> toff(nb);
        for (uint32_t k = 0; k < nb; ++k) n_spec += dec_sub_err(h_res.data(), k) ? 0u : dec_sub_spec(h_res.data(), k);
        if (getenv("NGSC_VERBOSE")) {   // the slowest sub-block of the batch, phase by phase (100 MHz ticks)
            uint64_t mx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (uint32_t k = 0; k < nb; ++k) { uint64_t t[8]; dec_sub_ticks(h_res.data(), k, t); for (int q = 0; q < 8; ++q) mx[q] = std::max(mx[q], t[q]); }
            fprintf(stderr, "[ngsc decode] head kernel: info + length table %.2f ms, title header %.2f ms, title body %.2f ms, title lengths %.2f ms, "
                            "quality header %.2f ms\n", mx[3] / 1e5, mx[4] / 1e5, mx[5] / 1e5, mx[6] / 1e5, mx[7] / 1e5);
            fprintf(stderr, "[ngsc decode] %u sub-blocks (%u with the speculative quality decoder so far): header + titles %.1f ms, tail kernel: "
                            "qualities %.1f ms, bases %.1f ms (slowest of the batch)\n", nb, n_spec, mx[0] / 1e5, mx[1] / 1e5, mx[2] / 1e5);
        }
        for (uint32_t k = 0; k < nb && !rc; ++k) {
            if (dec_sub_err(h_res.data(), k)) { rc = dec_sub_err(h_res.data(), k); break; }
            toff[k] = text_pos;
            text_off[b0 + k] = text_pos;
            if (n_records) n_records[b0 + k] = dec_sub_records(h_res.data(), k);
            text_pos += dec_sub_text(h_res.data(), k);
        }
        if (rc) break;
        if (text_pos > text_cap) { rc = NGSC_ERR_CAPACITY; break; }
        if (hipMemcpyAsync(d_toff, toff.data(), 8ull * nb, hipMemcpyHostToDevice, st) != hipSuccess) { rc = NGSC_ERR_HIP; break; }
        launch_dec_assemble(st, (const uint8_t *)d_sub, d_off, nb, scratch, stride, rec_cap, tcap, qcap, d_res, d_toff, (uint8_t *)d_text);
        if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) { rc = NGSC_ERR_HIP; break; }
        b0 += nb;
    }
    text_off[n_sb] = text_pos;
    return rc;
}

// ---- container reader --------------------------------------------------------------------------------------------------
namespace {

struct BitRd {   // MSB-first reads (the reading side of BitStream, bit_stream.h:280-433); never walks past the buffer
    const uint8_t *p; uint64_t n, bit = 0; bool ok = true;
    BitRd(const uint8_t *d, uint64_t len) : p(d), n(len) {}
    uint32_t get(uint32_t w)
    {
        uint32_t v = 0;
        for (; w; --w, ++bit) {
            if ((bit >> 3) >= n) { ok = false; return 0; }
            v = (v << 1) | ((p[bit >> 3] >> (7 - (bit & 7))) & 1u);
        }
        return v;
    }
};

uint32_t ceil_log2_u32(uint32_t x) { uint32_t b = 0; while ((1ull << b) < x) ++b; return b; }

struct FooterIn {   // ReadFooter, tasks.cpp:1203-1292 (written by MakeFooter :1104-1176)
    uint32_t PS = 0, BS = 0, SS = 0, LBES = 0;
    uint64_t FS = 0;
    std::vector<uint32_t> OV, CBO, LBS;
};

bool parse_footer(const uint8_t *d, uint64_t len, FooterIn &f)
{
    BitRd r(d, len);
    const uint32_t BEPS = r.get(4), BEFS = r.get(6), BEBS = r.get(4), BESS = r.get(4), BELB = r.get(5), BEOV = r.get(4);
    f.LBES = r.get(1);
    f.PS = r.get(BEPS);
    if (BEFS > 32) { const uint64_t hi = r.get(BEFS - 32); f.FS = (hi << 32) | r.get(32); }
    else f.FS = r.get(BEFS);
    f.BS = r.get(BEBS);
    f.SS = r.get(BESS);
    if (!r.ok || f.PS < 1 || (uint64_t)f.BS * ceil_log2_u32(f.PS) > len * 8) return false;
    f.OV.assign(f.PS, 0);
    for (uint32_t i = 1; i < f.PS; ++i) f.OV[i] = r.get(BEOV);
    const uint32_t cb = ceil_log2_u32(f.PS);
    f.CBO.resize(f.BS);
    for (uint32_t j = 0; j < f.BS; ++j) { f.CBO[j] = r.get(cb); if (f.CBO[j] >= f.PS) return false; }
    if (!f.LBES) { f.LBS.resize(f.PS); for (uint32_t i = 0; i < f.PS; ++i) f.LBS[i] = r.get(BELB); }
    return r.ok;
}

struct Piece { uint64_t off; uint32_t len; };
struct SubIn { uint32_t rank; uint64_t bytes = 0; std::vector<Piece> pieces; };
// A sub-block is one 8 MiB window of FASTQ text; its streams cannot be longer than a few times that (the worst case is a
// bit per symbol more than the input).  A container that claims more - e.g. through a chain of continuation bits - is refused
// before any buffer is sized from the sum.
constexpr uint64_t MAX_SUB_BYTES = 64ull << 20;

bool pread_all(int fd, void *dst, uint64_t n, uint64_t off)
{
    uint8_t *p = (uint8_t *)dst;
    while (n) {
        const ssize_t k = pread(fd, p, n, (off_t)off);
        if (k <= 0) return false;
        p += k; n -= (uint64_t)k; off += (uint64_t)k;
    }
    return true;
}
bool pwrite_all_at(int fd, const void *src, uint64_t n, uint64_t off)
{
    const uint8_t *p = (const uint8_t *)src;
    while (n) {
        const ssize_t k = pwrite(fd, p, n, (off_t)off);
        if (k <= 0) return false;
        p += k; n -= (uint64_t)k; off += (uint64_t)k;
    }
    return true;
}
double dsec() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int ngsc_read_footer(const uint8_t *footer, uint64_t footer_len, ngsc_footer_t *out, int32_t *overlaps, uint32_t *lb_sizes,
                     uint32_t cap_ranks, uint32_t *cbo, uint32_t cap_blocks)
{
    if (!footer || !out) return NGSC_ERR_ARG;
    FooterIn F;
    if (!parse_footer(footer, footer_len, F)) return NGSC_ERR_FORMAT;
    out->P = F.PS; out->n_blocks = F.BS; out->n_subblocks = F.SS; out->lbes = F.LBES; out->fastq_size = F.FS;
    if ((overlaps || lb_sizes) && cap_ranks < F.PS) return NGSC_ERR_CAPACITY;
    if (cbo && cap_blocks < F.BS) return NGSC_ERR_CAPACITY;
    for (uint32_t i = 0; i < F.PS; ++i) {
        if (overlaps) overlaps[i] = (int32_t)F.OV[i];
        if (lb_sizes) lb_sizes[i] = F.LBES ? 0u : F.LBS[i];
    }
    if (cbo) for (uint32_t j = 0; j < F.BS; ++j) cbo[j] = F.CBO[j];
    return NGSC_OK;
}

int ngsc_read_header(const uint8_t *block, uint64_t avail, uint32_t P, ngsc_block_header_t *out)
{
    if (!block || !out || P < 1) return NGSC_ERR_ARG;
    BitRd r(block, avail);
    out->wrid = r.get(ceil_log2_u32(P));
    out->header_len = r.get(12);
    out->n_subblocks = r.get(6);
    const uint32_t beso = r.get(5);
    out->bcss = r.get(2);
    if (!r.ok || out->wrid >= P || out->n_subblocks > 63) return NGSC_ERR_FORMAT;
    for (uint32_t i = 0; i < 63; ++i) out->sb_len[i] = i < out->n_subblocks ? r.get(beso) : 0u;
    return r.ok ? NGSC_OK : NGSC_ERR_FORMAT;
}

// `.ngsc` -> FASTQ.  Footer, then every block's header (blocks lie in the footer's CBO order, 8 MiB each except a rank's
// last), which gives each rank's sub-blocks as pieces of the file (a sub-block may continue over block borders: BCSS bits,
// tasks.cpp:1179-1200); ranks ascending, their sub-blocks go through ngsc_decompress_device() in batches and the text of
// rank r lands at its place in the output: s_r = r * (FS / P) + OV[r] (the first-record sync distance the footer keeps,
// phyNGSC.cpp:131-160) up to s_(r+1).  A record that two ranks both stored (phyNGSC.cpp:295-300) is
// thereby written once, and a file whose last line had no newline gets its exact size back.
int ngsc_decompress_file(ngsc_ctx *c, const char *in_path, const char *out_path, double *seconds_out)
{
    if (!c || !in_path || !out_path) return NGSC_ERR_ARG;
    if (hipSetDevice(ngsc_ctx_device(c)) != hipSuccess) return NGSC_ERR_HIP;
    const double t0 = dsec();
    double t_read = 0, t_dev = 0, t_write = 0;
    const int fd = open(in_path, O_RDONLY);
    if (fd < 0) return NGSC_ERR_IO;
    struct stat sb;
    if (fstat(fd, &sb) != 0) { close(fd); return NGSC_ERR_IO; }
    if (sb.st_size < 4) { close(fd); return NGSC_ERR_FORMAT; }
    const uint64_t size = (uint64_t)sb.st_size;
    uint8_t tail[2];
    if (!pread_all(fd, tail, 2, size - 2)) { close(fd); return NGSC_ERR_IO; }
    const uint64_t flen = ((uint64_t)tail[0] << 8) | tail[1];
    if (flen + 2 > size) { close(fd); return NGSC_ERR_FORMAT; }
    std::vector<uint8_t> fbytes(flen);
    FooterIn F;
    if (!pread_all(fd, fbytes.data(), flen, size - 2 - flen)) { close(fd); return NGSC_ERR_IO; }
    if (!parse_footer(fbytes.data(), flen, F)) { close(fd); return NGSC_ERR_FORMAT; }
    const uint64_t body = size - 2 - flen;
    // block sizes (tasks.cpp:1280-1292): 8 MiB except the last block of each rank; with LBES all last blocks share a size
    std::vector<int64_t> last_of(F.PS, -1);
    for (uint32_t j = 0; j < F.BS; ++j) last_of[F.CBO[j]] = j;
    std::vector<uint64_t> bsize(F.BS, NGSC_BLOCK_SIZE), boff(F.BS + 1, 0);
    uint32_t n_last = 0;
    for (uint32_t w = 0; w < F.PS; ++w) n_last += last_of[w] >= 0;
    if (F.LBES) {
        const uint64_t full = (uint64_t)(F.BS - n_last) * NGSC_BLOCK_SIZE;
        if (!n_last || body < full || (body - full) % n_last) { close(fd); return NGSC_ERR_FORMAT; }
        for (uint32_t w = 0; w < F.PS; ++w) if (last_of[w] >= 0) bsize[last_of[w]] = (body - full) / n_last;
    } else {
        for (uint32_t w = 0; w < F.PS; ++w) if (last_of[w] >= 0) bsize[last_of[w]] = F.LBS[w];
    }
    for (uint32_t j = 0; j < F.BS; ++j) boff[j + 1] = boff[j] + bsize[j];
    if (boff[F.BS] != body) { close(fd); return NGSC_ERR_FORMAT; }
    // headers -> per-rank sub-block piece lists
    std::vector<std::vector<SubIn>> subs(F.PS);
    std::vector<SubIn> open_sub(F.PS);
    std::vector<uint8_t> open_has(F.PS, 0);
    const uint32_t bewr = ceil_log2_u32(F.PS);
    std::vector<uint8_t> hb(4096 + 8);
    double tr = dsec();
    for (uint32_t j = 0; j < F.BS; ++j) {
        const uint64_t n = std::min<uint64_t>(hb.size(), bsize[j]);
        if (!pread_all(fd, hb.data(), n, boff[j])) { close(fd); return NGSC_ERR_IO; }
        BitRd r(hb.data(), n);
        const uint32_t wrid = r.get(bewr), bhs = r.get(12), nosb = r.get(6), beso = r.get(5), bcss = r.get(2);
        if (!r.ok || wrid != F.CBO[j] || bhs > bsize[j]) { close(fd); return NGSC_ERR_FORMAT; }
        uint64_t pos = boff[j] + bhs;
        for (uint32_t i = 0; i < nosb; ++i) {
            const uint32_t ln = r.get(beso);
            if (!r.ok || pos + ln > boff[j + 1]) { close(fd); return NGSC_ERR_FORMAT; }
            const bool first_cont = i == 0 && (bcss & 2), last_cont = i == nosb - 1 && (bcss & 1);
            if (first_cont != (open_has[wrid] != 0)) { close(fd); return NGSC_ERR_FORMAT; }   // a continuation needs its beginning
            if (!first_cont) { open_sub[wrid] = SubIn(); open_sub[wrid].rank = wrid; }
            open_sub[wrid].pieces.push_back({pos, ln});
            open_sub[wrid].bytes += ln;
            if (open_sub[wrid].bytes > MAX_SUB_BYTES) { close(fd); return NGSC_ERR_FORMAT; }
            pos += ln;
            open_has[wrid] = last_cont;
            if (!last_cont) subs[wrid].push_back(std::move(open_sub[wrid]));
        }
    }
    for (uint32_t w = 0; w < F.PS; ++w) if (open_has[w]) { close(fd); return NGSC_ERR_FORMAT; }
    t_read += dsec() - tr;
    // where each rank's text belongs
    std::vector<uint64_t> s_at(F.PS + 1);
    for (uint32_t w = 0; w < F.PS; ++w) s_at[w] = std::min<uint64_t>(F.FS, (uint64_t)w * (F.FS / F.PS) + F.OV[w]);
    s_at[F.PS] = F.FS;
    for (uint32_t w = 0; w < F.PS; ++w) if (s_at[w + 1] < s_at[w]) { close(fd); return NGSC_ERR_FORMAT; }
    // a footer that claims more text than any stream of this size can hold (a 1-base read under a 512-byte constant title is
    // about 1 KiB of text for a few bits of streams) is refused before a byte of the output is reserved
    if (F.FS > (size << 12) + (1ull << 20)) { close(fd); return NGSC_ERR_FORMAT; }
    const int ofd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (ofd < 0) { close(fd); return NGSC_ERR_IO; }
    // the file's size is known: its pages are reserved up front (as the compress driver does: preallocated page-cache pages take
    // writes 1.5x faster on the GPU box's file system, profiles/r03_ubench_write.txt).  NGSC_FALLOCATE=0: off
    // (a file system that has no room says so here and not after the whole file has been decoded; one without fallocate - EOPNOTSUPP -
    // is written as before)
    if (!(getenv("NGSC_FALLOCATE") && !strcmp(getenv("NGSC_FALLOCATE"), "0")) && ::fallocate(ofd, 0, 0, (off_t)F.FS) != 0 && errno == ENOSPC) {
        fprintf(stderr, "[ngsc -d] no room for %llu bytes of output in %s\n", (unsigned long long)F.FS, out_path);
        close(fd);
        close(ofd);
        (void)unlink(out_path);
        return NGSC_ERR_IO;
    }
    // batches: at most 512 sub-blocks (two decoder waves per CU) or 2 GiB of compressed bytes
    std::vector<const SubIn *> order;
    for (uint32_t w = 0; w < F.PS; ++w) for (const SubIn &x : subs[w]) order.push_back(&x);
    const uint64_t text_per_sb = 10ull << 20;
    std::vector<uint64_t> written(F.PS, 0);
    int rc = NGSC_OK;
    struct Batch { size_t b0, b1; uint64_t bytes; };
    std::vector<Batch> batches;
    uint64_t max_in = 0;
    uint32_t max_nb = 0;
    {
        const uint32_t per = getenv("NGSC_DEC_BATCH") ? (uint32_t)std::max(1, atoi(getenv("NGSC_DEC_BATCH"))) : 128u;   // (tests: small batches.  10 GB of 36-base reads, 64 / 96 / 128 / 192 / 256 sub-blocks per batch:
                                                                                                                             //  2.16 / 1.85 / 1.90 / 2.16 / 3.1 s; 150-base reads 4.33 / - / 4.39 / - / 4.74 s - the buffers of a batch are pinned
                                                                                                                             //  and unpinned at ~6 GB/s, which larger batches' faster kernels do not win back; profiles/r03_decode_batch_sweep.txt)
        for (size_t b0 = 0; b0 < order.size();) {
            size_t b1 = b0; uint64_t bytes = 0;
            while (b1 < order.size() && b1 - b0 < per && bytes + order[b1]->bytes + 32 <= (2ull << 30)) { bytes += (order[b1]->bytes + 15u) & ~15ull; ++b1; }
            if (b1 == b0) { bytes = (order[b1]->bytes + 15u) & ~15ull; ++b1; }
            batches.push_back({b0, b1, bytes});
            max_in = std::max(max_in, bytes + 64);
            max_nb = std::max<uint32_t>(max_nb, (uint32_t)(b1 - b0));
            b0 = b1;
        }
    }
    // (round 3) Three stages, two buffers between each pair: a reader thread (pread of a batch's pieces into pinned memory, upload),
    // this thread (the decoder kernels), a writer thread (download, pwrite at each rank's place) - the file used to go through
    // read, upload, decode, download and write one after the other with blocking copies (the compress driver of container.cpp
    // has overlapped them since round 2).
    const uint64_t cap_text = (uint64_t)max_nb * text_per_sb;
    uint8_t *h_in[2] = {nullptr, nullptr}, *h_text[2] = {nullptr, nullptr}, *d_in[2] = {nullptr, nullptr}, *d_text[2] = {nullptr, nullptr};
    hipStream_t s_up = nullptr, s_dec = nullptr, s_down = nullptr;
    const uint32_t nbuf = batches.size() > 1 ? 2u : 1u;
    for (uint32_t k = 0; k < nbuf && !rc; ++k)
        if (hipHostMalloc((void **)&h_in[k], max_in, hipHostMallocDefault) != hipSuccess || hipMalloc((void **)&d_in[k], max_in) != hipSuccess ||
            hipMalloc((void **)&d_text[k], cap_text) != hipSuccess)
            rc = NGSC_ERR_NOMEM;
    // (the pinned text buffers - 2.5 GB each at 256 short-read sub-blocks, ~0.4 s of page pinning apiece - come from a thread of
    // their own beside the first batches' read, upload and decode; inside the writer thread, which is the slowest stage, they
    // cost the pipeline their full time: 3.0 -> 3.4 s on 10 GB)
    const double t_setup = dsec() - t0;
    if (!rc && (hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s_dec, hipStreamNonBlocking) != hipSuccess ||
                hipStreamCreateWithFlags(&s_down, hipStreamNonBlocking) != hipSuccess))
        rc = NGSC_ERR_HIP;
    struct Meta { std::vector<uint64_t> off, toff; std::vector<uint32_t> len; };
    std::vector<Meta> meta(batches.size());
    std::mutex mu;
    std::condition_variable cv;
    size_t uploaded = 0, decoded = 0, drained = 0;   // batches that have passed each stage (drained: its device text buffer is free again)
    int fail = NGSC_OK;
    const int dev = ngsc_ctx_device(c);
    auto set_fail = [&](int code) { std::lock_guard<std::mutex> lk(mu); if (!fail) fail = code; cv.notify_all(); };
    std::thread reader, downloader, writer, pinner;
    size_t text_ready = 0;   // pinned text buffers allocated so far
    size_t downloaded = 0, wrote = 0;
    double t_down = 0;
    if (!rc) {
        pinner = std::thread([&]() {
            (void)hipSetDevice(dev);
            for (uint32_t k = 0; k < nbuf; ++k) {
                if (hipHostMalloc((void **)&h_text[k], cap_text, hipHostMallocDefault) != hipSuccess) { h_text[k] = nullptr; set_fail(NGSC_ERR_NOMEM); return; }
                std::lock_guard<std::mutex> lk(mu);
                ++text_ready;
                cv.notify_all();
            }
        });
        reader = std::thread([&]() {
            (void)hipSetDevice(dev);
            const unsigned nt = std::max(1u, std::min(8u, getenv("NGSC_IO_THREADS") ? (unsigned)atoi(getenv("NGSC_IO_THREADS")) : 8u));
            for (size_t b = 0; b < batches.size(); ++b) {
                {   // the buffer pair of batch b - 2 must have been decoded
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return fail || b < decoded + nbuf; });
                    if (fail) return;
                }
                const double t = dsec();
                const Batch &B = batches[b];
                const uint32_t nb = (uint32_t)(B.b1 - B.b0);
                Meta &M = meta[b];
                M.off.resize(nb); M.len.resize(nb); M.toff.assign(nb + 1, 0);
                uint8_t *dst = h_in[b % nbuf];
                uint64_t pos = 0;
                struct Job { uint64_t at, off; uint32_t len; };
                std::vector<Job> jobs;
                for (uint32_t k = 0; k < nb; ++k) {
                    const SubIn &x = *order[B.b0 + k];
                    M.off[k] = pos; M.len[k] = (uint32_t)x.bytes;
                    uint64_t q = pos;
                    for (const Piece &pc : x.pieces) { jobs.push_back({q, pc.off, pc.len}); q += pc.len; }
                    const uint64_t end = pos + ((x.bytes + 15u) & ~15ull);
                    memset(dst + q, 0, end - q);   // (the decoder reads whole 16-byte units)
                    pos = end;
                }
                memset(dst + pos, 0, 64);
                std::vector<char> okv(nt, 1);
                std::vector<std::thread> th;
                for (unsigned w = 0; w < nt; ++w) {
                    auto work = [&, w]() { for (size_t j = w; j < jobs.size(); j += nt) if (!pread_all(fd, dst + jobs[j].at, jobs[j].len, jobs[j].off)) okv[w] = 0; };
                    if (w + 1 == nt) work(); else th.emplace_back(work);
                }
                for (auto &x : th) x.join();
                for (char o : okv) if (!o) { set_fail(NGSC_ERR_IO); return; }
                if (hipMemcpyAsync(d_in[b % nbuf], dst, B.bytes + 64, hipMemcpyHostToDevice, s_up) != hipSuccess || hipStreamSynchronize(s_up) != hipSuccess) {
                    set_fail(NGSC_ERR_HIP);
                    return;
                }
                std::lock_guard<std::mutex> lk(mu);
                t_read += dsec() - t;
                ++uploaded;
                cv.notify_all();
            }
        });
        // download and write are stages of their own: batch b + 1 comes down while batch b is being written (the writes are the
        // slowest stage of a short-read file, and the download was a sixth of their thread's time)
        downloader = std::thread([&]() {
            (void)hipSetDevice(dev);
            for (size_t b = 0; b < batches.size(); ++b) {
                {   // decoded, a pinned buffer exists, and the batch that used it last has been written
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return fail || (decoded > b && text_ready > b % nbuf && b < wrote + nbuf); });
                    if (fail) return;
                }
                const double t = dsec();
                const uint32_t nb = (uint32_t)(batches[b].b1 - batches[b].b0);
                if (hipMemcpyAsync(h_text[b % nbuf], d_text[b % nbuf], meta[b].toff[nb], hipMemcpyDeviceToHost, s_down) != hipSuccess ||
                    hipStreamSynchronize(s_down) != hipSuccess) {
                    set_fail(NGSC_ERR_HIP);
                    return;
                }
                std::lock_guard<std::mutex> lk(mu);
                t_down += dsec() - t;
                ++drained;   // the decoder may fill this device buffer again
                ++downloaded;
                cv.notify_all();
            }
        });
        writer = std::thread([&]() {
            for (size_t b = 0; b < batches.size(); ++b) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return fail || downloaded > b; });
                    if (fail) return;
                }
                const double t = dsec();
                const Batch &B = batches[b];
                const uint32_t nb = (uint32_t)(B.b1 - B.b0);
                const Meta &M = meta[b];
                uint8_t *src = h_text[b % nbuf];
                struct Run { uint64_t at, from, n; };
                std::vector<Run> runs;
                for (uint32_t k = 0; k < nb; ++k) {   // (consecutive sub-blocks of a rank are consecutive in the text: one write per run)
                    const uint32_t w = order[B.b0 + k]->rank;
                    uint32_t k2 = k;
                    while (k2 + 1 < nb && order[B.b0 + k2 + 1]->rank == w) ++k2;
                    const uint64_t have = M.toff[k2 + 1] - M.toff[k], room = s_at[w + 1] - s_at[w] - std::min(written[w], s_at[w + 1] - s_at[w]);
                    const uint64_t n = std::min(have, room);
                    if (n) runs.push_back({s_at[w] + written[w], M.toff[k], n});
                    written[w] += n;
                    k = k2;
                }
                // big runs are cut into pieces for the I/O threads
                std::vector<Run> parts;
                for (const Run &r : runs)
                    for (uint64_t o = 0; o < r.n; o += 64ull << 20) parts.push_back({r.at + o, r.from + o, std::min<uint64_t>(64ull << 20, r.n - o)});
                const unsigned nt = std::max(1u, std::min(16u, getenv("NGSC_IO_THREADS") ? (unsigned)atoi(getenv("NGSC_IO_THREADS")) : 12u));   // (as container.cpp's io_threads())
                std::vector<char> okv(nt, 1);
                std::vector<std::thread> th;
                for (unsigned w = 0; w < nt; ++w) {
                    auto work = [&, w]() { for (size_t j = w; j < parts.size(); j += nt) if (!pwrite_all_at(ofd, src + parts[j].from, parts[j].n, parts[j].at)) okv[w] = 0; };
                    if (w + 1 == nt) work(); else th.emplace_back(work);
                }
                for (auto &x : th) x.join();
                for (char o : okv) if (!o) {
                    fprintf(stderr, "[ngsc -d] write to %s failed: %s\n", out_path, strerror(errno));
                    set_fail(NGSC_ERR_IO);
                    return;
                }
                std::lock_guard<std::mutex> lk(mu);
                t_write += dsec() - t;
                ++wrote;
                cv.notify_all();
            }
        });
        for (size_t b = 0; b < batches.size(); ++b) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return fail || (uploaded > b && b < drained + nbuf); });
                if (fail) break;
            }
            const double t = dsec();
            Meta &M = meta[b];
            const uint32_t nb = (uint32_t)M.off.size();
            const int r = ngsc_decompress_device(c, d_in[b % nbuf], M.off.data(), M.len.data(), nb, d_text[b % nbuf], cap_text, M.toff.data(), nullptr, s_dec);
            if (r) { set_fail(r); break; }
            std::lock_guard<std::mutex> lk(mu);
            t_dev += dsec() - t;
            ++decoded;
            cv.notify_all();
        }
        reader.join();
        downloader.join();
        writer.join();
        pinner.join();
        rc = fail;
    }
    const double t_pipe = dsec() - t0;
    if (s_up) (void)hipStreamDestroy(s_up);
    if (s_dec) (void)hipStreamDestroy(s_dec);
    if (s_down) (void)hipStreamDestroy(s_down);
    for (uint32_t k = 0; k < 2; ++k) {
        if (h_in[k]) (void)hipHostFree(h_in[k]);
        if (h_text[k]) (void)hipHostFree(h_text[k]);
        if (d_in[k]) (void)hipFree(d_in[k]);
        if (d_text[k]) (void)hipFree(d_text[k]);
    }
    if (!rc)   // every rank must have filled its stretch of the file
        for (uint32_t w = 0; w < F.PS; ++w) if (written[w] != s_at[w + 1] - s_at[w]) { rc = NGSC_ERR_FORMAT; break; }
    if (!rc && ftruncate(ofd, (off_t)F.FS) != 0) rc = NGSC_ERR_IO;
    close(fd);
    if (close(ofd) != 0 && !rc) rc = NGSC_ERR_IO;
    // a failed job leaves no file behind: the size came from the (untrusted) footer and the space was reserved up front - a
    // zero-filled file of that size would look like a result
    if (rc) (void)unlink(out_path);
    if (seconds_out) { seconds_out[0] = t_read; seconds_out[1] = t_dev; seconds_out[2] = t_down + t_write; }
    if (getenv("NGSC_VERBOSE"))
        fprintf(stderr, "[ngsc -d] %u ranks, %u blocks, %zu sub-blocks in %zu batches: read + upload %.3f s, decode %.3f s, download %.3f s, write %.3f s "
                        "(overlapped), buffers and streams %.3f s, pipeline done at %.3f s, total %.3f s\n", F.PS, F.BS, order.size(), batches.size(), t_read, t_dev, t_down, t_write, t_setup, t_pipe, dsec() - t0);
    return rc;
}

}  // extern "C"
