// decode.cpp - host side of the GPU decoder: ngsc_decompress_device() (include/ngsc.h), batches of sub-blocks through
// k_dec_streams / k_dec_assemble (kernels_decode.hip), and the container reader for `ngsc -d` (ReadFooter tasks.cpp:1203-1292,
// block headers tasks.cpp:1179-1200 read back).
#include "../../include/ngsc.h"
#include "ngsc_kernels.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace ngsc;

namespace ngsc {
void launch_dec_streams(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb, uint8_t *scratch,
                        uint64_t stride, uint32_t rec_cap, uint64_t tcap, uint64_t qcap, void *out);
void launch_dec_assemble(hipStream_t st, const uint8_t *sub, const uint64_t *sb_off, uint32_t n_sb, uint8_t *scratch, uint64_t stride,
                         uint32_t rec_cap, uint64_t tcap, uint64_t qcap, const void *in, const uint64_t *text_off, uint8_t *text);
size_t dec_sub_bytes();
int32_t dec_sub_err(const void *h, uint32_t s);
uint64_t dec_sub_text(const void *h, uint32_t s);
uint32_t dec_sub_records(const void *h, uint32_t s);
}

extern "C" {

// (the context's own stream and device; scratch is allocated per call and freed again: decoding is not the hot path)
int ngsc_decompress_device(ngsc_ctx *c, const void *d_sub, const uint64_t *sb_off, const uint32_t *sb_len, uint32_t n_sb, void *d_text,
                           uint64_t text_cap, uint64_t *text_off, uint32_t *n_records, void *hip_stream)
{
    if (!c || !d_sub || !sb_off || !sb_len || !n_sb || !d_text || !text_off) return NGSC_ERR_ARG;
    if (hipSetDevice(ngsc_ctx_device(c)) != hipSuccess) return NGSC_ERR_HIP;
    hipStream_t st = (hipStream_t)hip_stream;   // nullptr: the default stream
    // scratch of one sub-block: a sub-block holds the records of one 8 MiB window (+ the record that straddles its end)
    const uint32_t rec_cap = MAX_RECORDS + 2;
    const uint64_t tcap = 10ull << 20, qcap = 0, pool = 4ull << 20;   // text area: titles, qualities, bases <= the window's text
    const uint64_t stride = ((4ull * rec_cap * 3 + 96 + tcap + pool) + 255) & ~255ull;
    const uint32_t batch = std::min<uint32_t>(n_sb, 2048);              // ~15 MiB of scratch per sub-block in flight
    uint8_t *scratch = nullptr;
    uint64_t *d_off = nullptr, *d_toff = nullptr;
    uint32_t *d_len = nullptr;
    void *d_res = nullptr;
    const size_t rb = dec_sub_bytes();
    std::vector<uint8_t> h_res(rb * batch);
    int rc = NGSC_OK;
    if (hipMalloc(&scratch, stride * batch) != hipSuccess || hipMalloc(&d_off, 8ull * batch) != hipSuccess ||
        hipMalloc(&d_toff, 8ull * batch) != hipSuccess || hipMalloc(&d_len, 4ull * batch) != hipSuccess || hipMalloc(&d_res, rb * batch) != hipSuccess)
        rc = NGSC_ERR_NOMEM;
    uint64_t text_pos = 0;
    for (uint32_t b0 = 0; b0 < n_sb && !rc; b0 += batch) {
        const uint32_t nb = std::min(batch, n_sb - b0);
        if (hipMemcpyAsync(d_off, sb_off + b0, 8ull * nb, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_len, sb_len + b0, 4ull * nb, hipMemcpyHostToDevice, st) != hipSuccess) { rc = NGSC_ERR_HIP; break; }
        launch_dec_streams(st, (const uint8_t *)d_sub, d_off, d_len, nb, scratch, stride, rec_cap, tcap, qcap, d_res);
        if (hipMemcpyAsync(h_res.data(), d_res, rb * nb, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            rc = NGSC_ERR_HIP;
            break;
        }
        std::vector<uint64_t> toff(nb);
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
    }
    text_off[n_sb] = text_pos;
    if (scratch) (void)hipFree(scratch);
    if (d_off) (void)hipFree(d_off);
    if (d_toff) (void)hipFree(d_toff);
    if (d_len) (void)hipFree(d_len);
    if (d_res) (void)hipFree(d_res);
    return rc;
}

}  // extern "C"
