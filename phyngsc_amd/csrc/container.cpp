// container.cpp - host side of the .ngsc container: 8 MiB block assembly (phyNGSC.cpp:842-928),
// MakeHeader (tasks.cpp:1179-1200), MakeFooter (tasks.cpp:1104-1176) and the file-to-file driver that
// stands in for main() (phyNGSC.cpp:30-1076) with P virtual ranks on one GPU.
//
// Block order: the reference appends blocks with MPI_File_write_shared in arrival order and records
// it in the footer's CBO list; any order that keeps each rank's blocks in sequence is a valid file
// (tasks.cpp:1282-1292).  We use the deterministic order "round k = k-th block of every rank".
#include "../../include/ngsc.h"
#include "ngsc_common.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace ngsc;

extern "C" {

uint32_t ngsc_make_header(uint8_t *out, uint32_t wrid, uint32_t bewr, uint32_t bhs, uint32_t beso, uint32_t bcss,
                          const uint32_t *sbol, uint32_t nosb)
{
    ByteWriter w(out);
    w.put(wrid, bewr);
    w.put(bhs, 12);
    w.put(nosb, 6);
    w.put(beso, 5);
    w.put(bcss, 2);
    for (uint32_t i = 0; i < nosb; ++i) w.put(sbol[i], beso);
    w.align();
    return (uint32_t)w.nbytes();
}

uint32_t ngsc_make_footer(uint8_t *out, uint32_t P, uint64_t fastq_size, uint32_t n_blocks, uint32_t n_subblocks,
                          const int32_t *overlaps, const uint32_t *cbo, const uint32_t *lb_sizes)
{
    uint32_t lb_max = 0, lb_min = 0xffffffffu;
    int32_t ov_max = overlaps[0];
    for (uint32_t i = 0; i < P; ++i) {
        lb_max = std::max(lb_max, lb_sizes[i]);
        lb_min = std::min(lb_min, lb_sizes[i]);
        ov_max = std::max(ov_max, overlaps[i]);
    }
    // floor(log2(x)) + 1 (tasks.cpp:1108-1118)
    uint32_t BEPS = floor_log2_plus1(P), BEFS = floor_log2_plus1(fastq_size), BEBS = floor_log2_plus1(n_blocks);
    uint32_t BESS = floor_log2_plus1(n_subblocks), BELB = floor_log2_plus1(lb_max);
    uint32_t BEOV = floor_log2_plus1((uint64_t)(ov_max > 0 ? ov_max : 0));
    uint32_t LBES = lb_max != lb_min ? 0 : 1;
    ByteWriter w(out);
    w.put(BEPS, 4); w.put(BEFS, 6); w.put(BEBS, 4); w.put(BESS, 4); w.put(BELB, 5); w.put(BEOV, 4); w.put(LBES, 1);
    w.put(P, BEPS);
    if (BEFS > 32) {
        w.put((uint32_t)(fastq_size >> 32), BEFS - 32);
        w.put((uint32_t)(fastq_size & 0xffffffffu), 32);
    } else {
        w.put((uint32_t)fastq_size, BEFS);
    }
    w.put(n_blocks, BEBS);
    w.put(n_subblocks, BESS);
    for (uint32_t i = 1; i < P; ++i) w.put((uint32_t)overlaps[i], BEOV);
    uint32_t cb = ceil_log2(P);  // tasks.cpp:1156
    for (uint32_t i = 0; i < n_blocks; ++i) w.put(cbo[i], cb);
    if (!LBES)
        for (uint32_t i = 0; i < P; ++i) w.put(lb_sizes[i], BELB);
    w.align();
    uint32_t len = (uint32_t)w.nbytes();
    out[len] = (uint8_t)(len >> 8);  // Put2Bytes(footer length), tasks.cpp:1175
    out[len + 1] = (uint8_t)len;
    return len + 2;
}

int ngsc_plan_blocks(uint32_t P, uint32_t wrid, const uint32_t *sb_sizes, uint32_t n_sb, ngsc_block_t *blocks,
                     uint32_t cap, uint32_t *n_blocks)
{
    if (!sb_sizes || !blocks || !n_blocks || P < 2) return NGSC_ERR_ARG;
    const uint64_t W = NGSC_BLOCK_SIZE;
    const uint32_t bewr = ceil_log2(P);  // phyNGSC.cpp:163
    uint32_t sbol[64];
    uint32_t nsb = 0, beso = 0, bcss = 0, nb = 0;
    uint64_t written = 0;     // payload bytes in the open block
    uint64_t stream_pos = 0;  // bytes of the rank's stream already placed in closed blocks + open block
    uint64_t open_start = 0;  // stream offset where the open block's payload starts
    for (uint32_t i = 0; i < n_sb; ++i) {
        uint64_t S = sb_sizes[i];
        if (nsb >= 63) return NGSC_ERR_UNSUPPORTED;  // NOSB is a 6-bit field
        sbol[nsb++] = (uint32_t)S;                   // phyNGSC.cpp:843
        uint32_t mx = 0;
        for (uint32_t k = 0; k < nsb; ++k) mx = std::max(mx, sbol[k]);
        beso = floor_log2_plus1(mx);                 // :844
        uint32_t h = header_size(bewr, beso, nsb);
        if (written + S + h > W) {                   // :852
            if (written + h > W) return NGSC_ERR_UNSUPPORTED;  // reference would underflow bytes_fill_buffer
            bcss |= 1;                               // LSBS
            uint32_t fill = (uint32_t)(W - (written + h));
            sbol[nsb - 1] = fill;
            if (nb >= cap) return NGSC_ERR_CAPACITY;
            ngsc_block_t &b = blocks[nb++];
            b.wrid = wrid;
            b.header_len = ngsc_make_header(b.header, wrid, bewr, h, beso, bcss, sbol, nsb);
            b.payload = (uint32_t)(written + fill);
            b.size = b.header_len + b.payload;
            b.stream_off = open_start;
            stream_pos += fill;
            open_start = stream_pos;
            written = S - fill;
            stream_pos += written;
            bcss |= 2;                               // FSBS
            bcss &= ~1u;
            nsb = 0;
            sbol[nsb++] = (uint32_t)written;
        } else {
            written += S;
            stream_pos += S;
        }
    }
    if (written > 0) {                               // :910-928
        uint32_t h = header_size(bewr, beso, nsb);
        if (nb >= cap) return NGSC_ERR_CAPACITY;
        ngsc_block_t &b = blocks[nb++];
        b.wrid = wrid;
        b.header_len = ngsc_make_header(b.header, wrid, bewr, h, beso, bcss, sbol, nsb);
        b.payload = (uint32_t)written;
        b.size = b.header_len + b.payload;
        b.stream_off = open_start;
    }
    *n_blocks = nb;
    return NGSC_OK;
}

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool write_all(int fd, const void *p, size_t n)
{
    const uint8_t *b = (const uint8_t *)p;
    while (n) {
        ssize_t w = ::write(fd, b, n);
        if (w <= 0) return false;
        b += w;
        n -= (size_t)w;
    }
    return true;
}

// One worker = one context (= one GPU).  It walks its virtual ranks in JOBS of consecutive ranks of about 512 MiB
// (NGSC_JOB_BYTES sets the size; the engine's own limit is 4 GiB per job): a job's byte range is read from the file
// into pinned memory, uploaded, compressed, and its sub-block streams brought back.  Workers share nothing but the file.
struct Worker {
    ngsc_ctx *ctx = nullptr;
    uint32_t rank_lo = 0, rank_hi = 0;
    int rc = NGSC_OK;
    double io_s = 0;                        // time spent reading the file
    std::vector<uint8_t *> chunks;          // one pinned buffer per job: its sub-block bytes, rank after rank
    std::vector<const uint8_t *> rank_ptr;  // per entry of ranks: where that rank's stream starts
    std::vector<ngsc_rank_t> ranks;         // first_subblock into sb_sizes
    std::vector<uint32_t> sb_sizes;         // per sub-block, in result order
};

static uint64_t job_limit()
{
    const char *e = getenv("NGSC_JOB_BYTES");
    const uint64_t v = e ? strtoull(e, nullptr, 10) : 0;
    return v ? v : (512ull << 20);   // 512 MiB jobs: big enough for the kernels, small enough to overlap IO with compute
}

// byte range of ranks [a, b) (phyNGSC.cpp:113-116) + slack for the record that straddles the end
static void job_range(uint64_t FS, uint32_t P, uint32_t a, uint32_t b, uint64_t &lo, uint64_t &hi)
{
    const RankGeom g0 = rank_geometry(FS, P, a), g1 = rank_geometry(FS, P, b - 1);
    lo = (uint64_t)g0.wr_start & ~15ull;
    hi = std::min<uint64_t>(FS, (uint64_t)g1.wr_end + 1 + 65536);
}

// Two-stage pipeline per worker: a reader thread fills one of two pinned buffers from the file and uploads it
// while the worker thread compresses the previous job and brings its streams back.
struct JobPipe {
    std::mutex m;
    std::condition_variable cv;
    int uploaded = 0;    // jobs whose input is in device memory
    int consumed = 0;    // jobs whose compression is finished (their input buffers are free again)
    int reader_rc = NGSC_OK;
    bool abort = false;
};

static void run_worker(Worker *w, const char *in_path, uint64_t FS, uint32_t P)
{
    const int dev = ngsc_ctx_device(w->ctx);
    if (hipSetDevice(dev) != hipSuccess) { w->rc = NGSC_ERR_HIP; return; }
    const uint64_t limit = job_limit();
    // the jobs, and the largest one (buffers are allocated once)
    std::vector<std::pair<uint32_t, uint32_t>> jobs;
    uint64_t max_len = 0;
    for (uint32_t a = w->rank_lo; a < w->rank_hi;) {
        uint32_t b = a + 1;
        uint64_t lo, hi;
        job_range(FS, P, a, b, lo, hi);
        if (hi - lo >= 0xFFFFFF00ull) { w->rc = NGSC_ERR_UNSUPPORTED; return; }   // one rank alone is too big: use more ranks
        while (b < w->rank_hi) {
            uint64_t lo2, hi2;
            job_range(FS, P, a, b + 1, lo2, hi2);
            if (hi2 - lo2 > limit) break;
            ++b;
            hi = hi2;
        }
        max_len = std::max(max_len, hi - lo);
        jobs.emplace_back(a, b);
        a = b;
    }
    const double t_alloc0 = now_s();
    const int nbuf = jobs.size() > 1 ? 2 : 1;
    void *d_in[2] = {nullptr, nullptr}, *h_in[2] = {nullptr, nullptr}, *d_out = nullptr;
    hipStream_t copy_stream = nullptr;
    const uint64_t out_cap = max_len + (max_len >> 2) + (64u << 20);
    for (int i = 0; i < nbuf && !w->rc; ++i)
        if (hipMalloc(&d_in[i], max_len + 64) != hipSuccess || hipHostMalloc(&h_in[i], max_len + 64, hipHostMallocDefault) != hipSuccess)
            w->rc = NGSC_ERR_HIP;
    if (!w->rc && (hipMalloc(&d_out, out_cap) != hipSuccess || hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking) != hipSuccess))
        w->rc = NGSC_ERR_HIP;

    const bool verbose = getenv("NGSC_VERBOSE") != nullptr;
    double t_comp = 0, t_d2h = 0, t_wait = 0;
    const double t_alloc = now_s() - t_alloc0;
    JobPipe pipe;
    std::thread reader;
    if (!w->rc) {
        reader = std::thread([&]() {
            int rc = NGSC_OK;
            int fd = -1;
            if (hipSetDevice(dev) != hipSuccess) rc = NGSC_ERR_HIP;
            else if ((fd = ::open(in_path, O_RDONLY)) < 0) rc = NGSC_ERR_IO;
            for (size_t j = 0; j < jobs.size() && !rc; ++j) {
                {   // buffer j % nbuf is free once job j - nbuf has been compressed
                    std::unique_lock<std::mutex> lk(pipe.m);
                    pipe.cv.wait(lk, [&] { return pipe.abort || pipe.consumed + nbuf > (int)j; });
                    if (pipe.abort) break;
                }
                uint64_t lo, hi;
                job_range(FS, P, jobs[j].first, jobs[j].second, lo, hi);
                const uint64_t len = hi - lo;
                uint8_t *hb = (uint8_t *)h_in[j % nbuf];
                const double t0 = now_s();
                uint64_t got = 0;
                while (got < len) {
                    const ssize_t r = ::pread(fd, hb + got, (size_t)std::min<uint64_t>(len - got, 1u << 30), (off_t)(lo + got));
                    if (r <= 0) break;
                    got += (uint64_t)r;
                }
                w->io_s += now_s() - t0;
                if (got != len) { rc = NGSC_ERR_IO; break; }
                if (hipMemcpyAsync(d_in[j % nbuf], hb, len, hipMemcpyHostToDevice, copy_stream) != hipSuccess ||
                    hipStreamSynchronize(copy_stream) != hipSuccess) { rc = NGSC_ERR_HIP; break; }
                {
                    std::lock_guard<std::mutex> lk(pipe.m);
                    pipe.uploaded = (int)j + 1;
                }
                pipe.cv.notify_all();
            }
            if (fd >= 0) ::close(fd);
            {
                std::lock_guard<std::mutex> lk(pipe.m);
                pipe.reader_rc = rc;
                if (rc) pipe.abort = true;
            }
            pipe.cv.notify_all();
        });
    }
    for (size_t j = 0; j < jobs.size() && !w->rc; ++j) {
        double t0 = now_s();
        {
            std::unique_lock<std::mutex> lk(pipe.m);
            pipe.cv.wait(lk, [&] { return pipe.abort || pipe.uploaded > (int)j; });
            if (pipe.abort) { w->rc = pipe.reader_rc ? pipe.reader_rc : NGSC_ERR_IO; break; }
        }
        t_wait += now_s() - t0;
        t0 = now_s();
        uint64_t lo, hi;
        job_range(FS, P, jobs[j].first, jobs[j].second, lo, hi);
        w->rc = ngsc_compress_device(w->ctx, d_in[j % nbuf], hi - lo, lo, FS, P, jobs[j].first, jobs[j].second, d_out, out_cap, nullptr);
        if (w->rc) break;
        t_comp += now_s() - t0;
        t0 = now_s();
        const uint64_t total = ngsc_result_bytes(w->ctx);
        void *chunk = nullptr;   // pinned: a pageable destination makes the copy run at ~2 GB/s
        if (hipHostMalloc(&chunk, total + 64, hipHostMallocDefault) != hipSuccess) { w->rc = NGSC_ERR_HIP; break; }
        w->chunks.push_back((uint8_t *)chunk);
        if (total && hipMemcpy(chunk, d_out, total, hipMemcpyDeviceToHost) != hipSuccess) { w->rc = NGSC_ERR_HIP; break; }
        t_d2h += now_s() - t0;
        const ngsc_subblock_t *sbs = ngsc_result_subblocks(w->ctx);
        const ngsc_rank_t *rks = ngsc_result_ranks(w->ctx);
        const uint32_t nr = ngsc_result_num_ranks(w->ctx), ns = ngsc_result_num_subblocks(w->ctx);
        const uint32_t sb_base = (uint32_t)w->sb_sizes.size();
        for (uint32_t k = 0; k < nr; ++k) {
            ngsc_rank_t rk = rks[k];
            w->rank_ptr.push_back((const uint8_t *)chunk + rk.stream_off);
            rk.first_subblock += sb_base;
            w->ranks.push_back(rk);
        }
        for (uint32_t k = 0; k < ns; ++k) w->sb_sizes.push_back(sbs[k].info_len + sbs[k].title_len + sbs[k].qual_len + sbs[k].dna_len);
        {
            std::lock_guard<std::mutex> lk(pipe.m);
            pipe.consumed = (int)j + 1;
        }
        pipe.cv.notify_all();
    }
    if (reader.joinable()) {
        {
            std::lock_guard<std::mutex> lk(pipe.m);
            if (w->rc) pipe.abort = true;
        }
        pipe.cv.notify_all();
        reader.join();
        if (!w->rc && pipe.reader_rc) w->rc = pipe.reader_rc;
    }
    const double t_free = now_s();
    for (int i = 0; i < 2; ++i) {
        if (d_in[i]) (void)hipFree(d_in[i]);
        if (h_in[i]) (void)hipHostFree(h_in[i]);
    }
    if (d_out) (void)hipFree(d_out);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (verbose)
        fprintf(stderr, "[ngsc] device %d: %zu jobs, buffers %.3f s (alloc, before the first job), wait-for-input %.3f s, compress %.3f s, "
                        "D2H %.3f s, file read %.3f s (overlapped), free %.3f s\n", dev, jobs.size(), t_alloc, t_wait, t_comp, t_d2h, w->io_s, now_s() - t_free);
}

int ngsc_compress_file_multi(ngsc_ctx **ctxs, uint32_t n_ctx, const char *in_path, const char *out_path, uint32_t P,
                             double *seconds_out)
{
    if (!ctxs || !n_ctx || !in_path || !out_path || P < 2 || P % n_ctx) return NGSC_ERR_ARG;
    for (uint32_t i = 0; i < n_ctx; ++i)
        if (!ctxs[i]) return NGSC_ERR_ARG;
    double t0 = now_s();
    struct stat sbuf;
    if (::stat(in_path, &sbuf) != 0 || sbuf.st_size <= 0) return NGSC_ERR_IO;
    const uint64_t FS = (uint64_t)sbuf.st_size;
    double t1 = now_s();

    // virtual ranks are dealt to the contexts in contiguous runs (P is a multiple of the context count)
    std::vector<Worker> workers(n_ctx);
    const uint32_t per = P / n_ctx;
    for (uint32_t i = 0; i < n_ctx; ++i) { workers[i].ctx = ctxs[i]; workers[i].rank_lo = i * per; workers[i].rank_hi = (i + 1) * per; }
    if (n_ctx == 1) run_worker(&workers[0], in_path, FS, P);
    else {
        std::vector<std::thread> th;
        for (uint32_t i = 0; i < n_ctx; ++i) th.emplace_back(run_worker, &workers[i], in_path, FS, P);
        for (auto &t : th) t.join();
    }
    double t2 = now_s();
    auto free_chunks = [&]() {
        for (auto &w : workers) {
            for (uint8_t *c : w.chunks) (void)hipHostFree(c);
            w.chunks.clear();
        }
    };
    for (auto &w : workers)
        if (w.rc) { const int rc = w.rc; free_chunks(); return rc; }

    // container: blocks of every rank, deterministic order, footer (phyNGSC.cpp:842-1057)
    std::vector<std::vector<ngsc_block_t>> rb(P);
    std::vector<int32_t> ov(P);
    std::vector<uint32_t> lbs(P);
    std::vector<const uint8_t *> stream(P);
    uint32_t n_sub_total = 0, max_blocks = 0, n_blocks_total = 0;
    for (auto &w : workers) {
        for (size_t ri = 0; ri < w.ranks.size(); ++ri) {
            const ngsc_rank_t &rk = w.ranks[ri];
            const uint32_t r = rk.rank;
            uint32_t capb = (uint32_t)(rk.stream_bytes / (NGSC_BLOCK_SIZE - 4096)) + 4, nb = 0;
            rb[r].resize(capb);
            int rc = ngsc_plan_blocks(P, r, w.sb_sizes.data() + rk.first_subblock, rk.n_subblocks, rb[r].data(), capb, &nb);
            if (rc) { free_chunks(); return rc; }
            rb[r].resize(nb);
            ov[r] = rk.overlap_used;
            lbs[r] = nb ? rb[r][nb - 1].size : 0;
            stream[r] = w.rank_ptr[ri];
            n_sub_total += rk.n_subblocks;
            n_blocks_total += nb;
            max_blocks = std::max(max_blocks, nb);
        }
    }
    int ofd = ::open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (ofd < 0) { free_chunks(); return NGSC_ERR_IO; }
    std::vector<uint32_t> cbo;
    bool okw = true;
    for (uint32_t k = 0; k < max_blocks && okw; ++k)
        for (uint32_t r = 0; r < P && okw; ++r)
            if (k < rb[r].size()) {
                const ngsc_block_t &b = rb[r][k];
                okw = write_all(ofd, b.header, b.header_len) && write_all(ofd, stream[r] + b.stream_off, b.payload);
                cbo.push_back(r);
            }
    std::vector<uint8_t> footer(256 + 4 * (size_t)n_blocks_total + 8 * (size_t)P);
    uint32_t fl = ngsc_make_footer(footer.data(), P, FS, n_blocks_total, n_sub_total, ov.data(), cbo.data(), lbs.data());
    okw = okw && write_all(ofd, footer.data(), fl);
    ::close(ofd);
    double t3 = now_s();
    free_chunks();
    double io = 0;
    for (auto &w : workers) io = std::max(io, w.io_s);
    if (seconds_out) { seconds_out[0] = (t1 - t0) + io; seconds_out[1] = (t2 - t1) - io; seconds_out[2] = t3 - t2; }
    return okw ? NGSC_OK : NGSC_ERR_IO;
}

int ngsc_compress_file(ngsc_ctx *ctx, const char *in_path, const char *out_path, uint32_t P, double *seconds_out)
{
    return ngsc_compress_file_multi(&ctx, 1, in_path, out_path, P, seconds_out);
}

}  // extern "C"
