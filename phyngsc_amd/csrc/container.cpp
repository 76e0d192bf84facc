// container.cpp - host side of the .ngsc container: 8 MiB block assembly (phyNGSC.cpp:842-928),
// MakeHeader (tasks.cpp:1179-1200), MakeFooter (tasks.cpp:1104-1176) and the file-to-file driver that
// stands in for main() (phyNGSC.cpp:30-1076) with P virtual ranks on one GPU.
//
// Block order: the reference appends blocks with MPI_File_write_shared in arrival order and records
// it in the footer's CBO list; any order that keeps each rank's blocks in sequence is a valid file
// (tasks.cpp:1282-1292).  We use a deterministic order: groups of consecutive ranks (at most 512 MiB of input each), and
// inside a group "round k = k-th block of each of its ranks" (see ranks_per_group() below).
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
            // what is left of the sub-block must fit the next block: BCSS says "continues" for one block border only (the
            // reference cannot meet such a sub-block - 8 MiB of FASTQ text never compress to more than a block)
            if (written + header_size(bewr, floor_log2_plus1(written), 1) > W) return NGSC_ERR_UNSUPPORTED;
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

// number of threads a big read or write of the file driver is split over (page-cache copies run at ~10 GB/s per thread)
static unsigned io_threads()
{
    const char *e = getenv("NGSC_IO_THREADS");
    const int v = e ? atoi(e) : 12;   // (round 3 sweep on the 50 GB file, 64 ranks: 4 / 6 / 8 / 12 threads = 3.3 / 3.0 / 3.0 / 2.7 s)
    return (unsigned)std::max(1, std::min(v, 16));
}

// pread of [off, off + len) into dst, split over io_threads() threads
static bool pread_parallel(int fd, uint8_t *dst, uint64_t len, uint64_t off)
{
    const unsigned nt = len < (64u << 20) ? 1u : io_threads();
    std::vector<std::thread> th;
    std::vector<char> ok(nt, 1);
    const uint64_t per = ((len + nt - 1) / nt + 4095) & ~4095ull;
    for (unsigned t = 0; t < nt; ++t) {
        const uint64_t a = std::min<uint64_t>(len, (uint64_t)t * per), b = std::min<uint64_t>(len, a + per);
        auto work = [=, &ok]() {
            uint64_t got = a;
            while (got < b) {
                const ssize_t r = ::pread(fd, dst + got, (size_t)std::min<uint64_t>(b - got, 1u << 30), (off_t)(off + got));
                if (r <= 0) { ok[t] = 0; return; }
                got += (uint64_t)r;
            }
        };
        if (t + 1 == nt) work(); else th.emplace_back(work);
    }
    for (auto &x : th) x.join();
    for (char c : ok) if (!c) return false;
    return true;
}

static bool pwrite_all(int fd, const void *p, size_t n, uint64_t off)
{
    const uint8_t *b = (const uint8_t *)p;
    while (n) {
        ssize_t w = ::pwrite(fd, b, n, (off_t)off);
        if (w <= 0) return false;
        b += w;
        n -= (size_t)w;
        off += (uint64_t)w;
    }
    return true;
}

// independent writes shared by io_threads() threads
struct WriteItem { uint64_t off; const uint8_t *hdr; uint32_t hdr_len; const uint8_t *payload; uint32_t len; };
static bool write_items(int fd, const std::vector<WriteItem> &items)
{
    const unsigned nt = items.size() < 4 ? 1u : std::min<unsigned>(io_threads(), (unsigned)items.size());
    std::vector<char> okv(nt, 1);
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) {
        auto work = [&, t]() {
            for (size_t x = t; x < items.size(); x += nt) {
                const WriteItem &w = items[x];
                if (!pwrite_all(fd, w.hdr, w.hdr_len, w.off) || !pwrite_all(fd, w.payload, w.len, w.off + w.hdr_len)) okv[t] = 0;
            }
        };
        if (t + 1 == nt) work(); else th.emplace_back(work);
    }
    for (auto &x : th) x.join();
    for (char c : okv) if (!c) return false;
    return true;
}

// ---- block order --------------------------------------------------------------------------------------------------
// Ranks in groups of consecutive ranks whose regions make up at most NGSC_ORDER_GROUP_BYTES of input, group after
// group; inside a group round k = k-th block of each of its ranks.  A group's place in the file depends only on the
// groups in front of it, so a group is written as soon as it is compressed: host memory holds a group, not the file.
static uint32_t ranks_per_group(uint64_t FS, uint32_t P)
{
    const uint64_t region = FS / P;
    uint64_t rpg = region ? NGSC_ORDER_GROUP_BYTES / region : P;
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(P, rpg));
}

// What the workers share: the output file and the place of the next group in it.
struct Sink {
    int fd = -1;
    std::mutex m;
    std::condition_variable cv;
    uint32_t next_group = 0;   // the group whose blocks go to `cursor`
    uint64_t cursor = 0;
    bool fail = false;
    std::vector<uint32_t> cbo;             // WRID of every block in file order
    std::vector<int32_t> ov;               // per rank
    std::vector<uint32_t> lbs, nblocks;    // per rank: last block size, block count
    uint32_t n_sub_total = 0;
    // waits for the group's turn; returns false when another worker failed
    bool begin_group(uint32_t g)
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return fail || next_group == g; });
        return !fail;
    }
    void end_group()
    {
        { std::lock_guard<std::mutex> lk(m); ++next_group; }
        cv.notify_all();
    }
    void abort()
    {
        { std::lock_guard<std::mutex> lk(m); fail = true; }
        cv.notify_all();
    }
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

// One unit of device work: consecutive whole ranks [a, b) of one group, or one PART of a rank whose region is larger
// than a job (ngsc_compress_rank_part; the parts' buffers overlap by a window and a record, so that every part starts at
// or before the window its predecessor left).
struct Job {
    uint32_t group, a, b;
    uint64_t lo, hi;       // byte range of the file the job's buffer holds
    bool part, first_part, last_of_group;
    bool stream;           // part of a rank that is a group of its own: its blocks are written as they fill
    bool last_part;        // the part that reaches the end of the rank's region
};
constexpr uint64_t PART_OVERLAP = (uint64_t)WINDOW + 65536;   // a window + the longest record a part boundary allows

// One worker = one context (= one GPU) and every n-th group.  A reader thread fills one of two pinned buffers from the
// file and uploads it while the worker thread compresses the previous job; a writer thread brings the sub-block streams
// back into one of two pinned buffers, plans the blocks and writes them at the group's place while the next job runs.
struct Worker {
    ngsc_ctx *ctx = nullptr;
    uint32_t index = 0, n_workers = 1;
    int rc = NGSC_OK;
    double io_s = 0, t_comp = 0, t_wait_in = 0, t_wait_out = 0, t_d2h = 0, t_write = 0, t_xchg = 0;
    // Several GPUs: no shared cursor.  The workers' groups of one ROUND (groups round * n .. round * n + n - 1, one per worker)
    // meet in the team's collective - an RCCL all-gather of the block-size rows over the GPUs' communicator, the reference's
    // MPI_Gather / MPI_Gatherv (phyNGSC.cpp:964, 994) - and every worker derives the same offsets from the same rows by a
    // local prefix sum, writes its own blocks, and keeps the footer's tables (worker 0's copy is the one written).
    ngsc_team *team = nullptr;
    uint64_t cursor = 0;
    std::vector<uint32_t> cbo, lbs, nblocks;
    std::vector<int32_t> ov;
    uint32_t n_sub_total = 0;
};

// words per rank row of the exchange: n_blocks, n_subblocks, overlap_used, block sizes (a rank's stream cannot be larger than
// what the output buffers of its jobs hold: 1.25 x its region + 64 MiB a job)
static uint32_t exchange_row_len(uint64_t FS, uint32_t P)
{
    const uint64_t region = FS / P + OVERLAP + 65536;
    const uint64_t jobs = region / (64u << 20) + 2;
    return 3 + (uint32_t)std::min<uint64_t>(1u << 20, (region + (region >> 2) + jobs * (64ull << 20)) / (NGSC_BLOCK_SIZE - 4096) + 4);
}

// One round of the exchange from worker w's writer thread.  mine: rpg rows of this worker's group (absent ranks: all zero), rc_in:
// what the worker has to report.  On success the offsets of this worker's blocks are in my_off[rank slot][block], and the footer
// tables / cursor have moved past the round's groups.
static int exchange_round(Worker *w, uint32_t round, uint32_t rpg, uint32_t row_len, uint32_t P, uint32_t n_groups, int rc_in,
                          const std::vector<uint64_t> &mine, std::vector<std::vector<uint64_t>> &my_off)
{
    const uint32_t n = w->n_workers, count = rpg * row_len;
    std::vector<uint64_t> all((size_t)count * n);
    const double t0 = now_s();
    const int rc = ngsc_team_member_allgather(w->team, w->index, rc_in, mine.data(), count, all.data());
    w->t_xchg += now_s() - t0;
    if (rc) return rc;
    my_off.assign(rpg, std::vector<uint64_t>());
    for (uint32_t v = 0; v < n; ++v) {
        const uint32_t g = round * n + v;
        if (g >= n_groups) break;
        const uint32_t r0 = g * rpg, nr = std::min(rpg, P - r0);
        const uint64_t *rows = all.data() + (size_t)v * count;
        uint64_t maxb = 0;
        for (uint32_t k = 0; k < nr; ++k) {
            const uint64_t nb = rows[(size_t)k * row_len];
            if (nb > (uint64_t)row_len - 3) return NGSC_ERR_FORMAT;
            maxb = std::max(maxb, nb);
            if (v == w->index) my_off[k].resize(nb);
        }
        for (uint64_t b = 0; b < maxb; ++b)
            for (uint32_t k = 0; k < nr; ++k) {
                const uint64_t *row = rows + (size_t)k * row_len;
                if (b >= row[0]) continue;
                if (v == w->index) my_off[k][b] = w->cursor;
                w->cursor += row[3 + b];
                w->cbo.push_back(r0 + k);
            }
        for (uint32_t k = 0; k < nr; ++k) {
            const uint64_t *row = rows + (size_t)k * row_len;
            w->ov[r0 + k] = (int32_t)(int64_t)row[2];
            w->lbs[r0 + k] = row[0] ? (uint32_t)row[3 + row[0] - 1] : 0;
            w->nblocks[r0 + k] = (uint32_t)row[0];
            w->n_sub_total += (uint32_t)row[1];
        }
    }
    return NGSC_OK;
}

struct Pipe {
    std::mutex m;
    std::condition_variable cv;
    int uploaded = 0;      // jobs whose input is in device memory
    int compressed = 0;    // jobs whose compression is finished (input buffers free again, output waits for the writer)
    int written = 0;       // jobs whose output has left its device / pinned buffers
    int rc = NGSC_OK;
    bool abort = false;
    void fail(int code)
    {
        { std::lock_guard<std::mutex> lk(m); if (!rc) rc = code; abort = true; }
        cv.notify_all();
    }
};

// results of one job, copied out of the context before the next job overwrites them
struct JobResult {
    uint64_t total = 0;
    std::vector<ngsc_rank_t> ranks;
    std::vector<uint32_t> sb_sizes;
    ngsc_rank_part_t part;
};

static void run_worker(Worker *w, Sink *sink, const char *in_path, uint64_t FS, uint32_t P)
{
    const int dev = ngsc_ctx_device(w->ctx);
    if (hipSetDevice(dev) != hipSuccess) { w->rc = NGSC_ERR_HIP; sink->abort(); return; }
    const uint64_t limit = std::min<uint64_t>(job_limit(), 0xE0000000ull);
    const uint32_t rpg = ranks_per_group(FS, P), n_groups = (P + rpg - 1) / rpg;
    // ---- the jobs of this worker's groups ----
    std::vector<Job> jobs;
    uint64_t max_len = 0;
    for (uint32_t g = w->index; g < n_groups; g += w->n_workers) {
        const uint32_t g0 = g * rpg, g1 = std::min(P, g0 + rpg);
        for (uint32_t a = g0; a < g1;) {
            uint64_t lo, hi;
            job_range(FS, P, a, a + 1, lo, hi);
            // (a part has to hold a window and a record beyond its predecessor's: job sizes below three such overlaps - only
            // test settings ask for them - still cut whole ranks, but parts no smaller than that)
            const uint64_t part_limit = std::max<uint64_t>(limit, 3 * PART_OVERLAP);
            if (hi - lo > part_limit) {   // one rank alone is more than a job: walk it in parts
                const uint64_t rank_hi = hi;
                const uint64_t rank_end = std::min<uint64_t>(FS, (uint64_t)rank_geometry(FS, P, a).wr_end + 1);
                bool first = true;
                for (uint64_t p = lo;; p = (p + part_limit - PART_OVERLAP) & ~15ull) {
                    uint64_t e = std::min<uint64_t>(rank_hi, p + part_limit);
                    if (e >= rank_end) e = rank_hi;   // the part that reaches the end of the region finishes the rank
                    jobs.push_back({g, a, a + 1, p, e, true, first, false, g1 - g0 == 1 && !w->team, e == rank_hi});   // (several GPUs: a group's blocks
                                                                                                               // are placed by the round's exchange, when the group is complete)
                    max_len = std::max(max_len, e - p);
                    first = false;
                    if (e == rank_hi) break;
                }
                ++a;
            } else {
                uint32_t b = a + 1;
                while (b < g1) {
                    uint64_t lo2, hi2;
                    job_range(FS, P, a, b + 1, lo2, hi2);
                    if (hi2 - lo2 > limit) break;
                    ++b;
                    hi = hi2;
                }
                jobs.push_back({g, a, b, lo, hi, false, false, false, false, false});
                max_len = std::max(max_len, hi - lo);
                a = b;
            }
        }
        jobs.back().last_of_group = true;
    }
    const uint32_t n_rounds = (n_groups + w->n_workers - 1) / w->n_workers, row_len = w->team ? exchange_row_len(FS, P) : 0;
    uint32_t rounds_done = 0;   // (by this worker's writer thread)
    // a worker without a group in some round (the last one, or a file with fewer groups than GPUs) still takes part in the exchange
    auto idle_rounds = [&](int status) {
        if (!w->team) return;
        std::vector<std::vector<uint64_t>> none;
        const std::vector<uint64_t> zero((size_t)rpg * row_len, 0);
        for (; rounds_done < n_rounds; ++rounds_done) {
            const int r = exchange_round(w, rounds_done, rpg, row_len, P, n_groups, status, zero, none);
            if (r) { if (!w->rc) w->rc = r; break; }
        }
    };
    if (jobs.empty()) { idle_rounds(NGSC_OK); return; }
    const int nbuf = jobs.size() > 1 ? 2 : 1;
    void *d_in[2] = {nullptr, nullptr}, *h_in[2] = {nullptr, nullptr}, *d_out[2] = {nullptr, nullptr}, *h_out[2] = {nullptr, nullptr};
    hipStream_t in_stream = nullptr, out_stream = nullptr;
    const uint64_t out_cap = max_len + (max_len >> 2) + (64u << 20);
    const double t_alloc0 = now_s();
    for (int i = 0; i < nbuf && !w->rc; ++i)
        if (hipMalloc(&d_in[i], max_len + 64) != hipSuccess || hipHostMalloc(&h_in[i], max_len + 64, hipHostMallocDefault) != hipSuccess ||
            hipMalloc(&d_out[i], out_cap) != hipSuccess || hipHostMalloc(&h_out[i], out_cap, hipHostMallocDefault) != hipSuccess)
            w->rc = NGSC_ERR_HIP;
    if (!w->rc && (hipStreamCreateWithFlags(&in_stream, hipStreamNonBlocking) != hipSuccess ||
                   hipStreamCreateWithFlags(&out_stream, hipStreamNonBlocking) != hipSuccess))
        w->rc = NGSC_ERR_HIP;
    // (test hook: NGSC_TEST_FAIL_ALLOC=<device index> makes that worker's buffer allocation fail)
    if (const char *fa = getenv("NGSC_TEST_FAIL_ALLOC"))
        if (atoi(fa) == (int)w->index && !w->rc) w->rc = NGSC_ERR_NOMEM;
    const double t_alloc = now_s() - t_alloc0;
    if (w->rc) {
        // no reader or writer thread will ever run: nobody would end the wait below, and the other workers would wait for
        // this worker's groups in begin_group() - tell them, release what was allocated, and return the error
        sink->abort();
        if (w->team) ngsc_team_abort(w->team, w->rc);
        for (int i = 0; i < 2; ++i) {
            if (d_in[i]) (void)hipFree(d_in[i]);
            if (h_in[i]) (void)hipHostFree(h_in[i]);
            if (d_out[i]) (void)hipFree(d_out[i]);
            if (h_out[i]) (void)hipHostFree(h_out[i]);
        }
        if (in_stream) (void)hipStreamDestroy(in_stream);
        if (out_stream) (void)hipStreamDestroy(out_stream);
        return;
    }
    Pipe pipe;
    std::vector<JobResult> results(jobs.size());
    std::thread reader, writer;
    if (!w->rc) {
        // ---- reader: file -> pinned -> device, one job ahead ----
        reader = std::thread([&]() {
            int rc = NGSC_OK, fd = -1;
            if (hipSetDevice(dev) != hipSuccess) rc = NGSC_ERR_HIP;
            else if ((fd = ::open(in_path, O_RDONLY)) < 0) rc = NGSC_ERR_IO;
            for (size_t j = 0; j < jobs.size() && !rc; ++j) {
                {   // buffer j % nbuf is free once job j - nbuf has been compressed
                    std::unique_lock<std::mutex> lk(pipe.m);
                    pipe.cv.wait(lk, [&] { return pipe.abort || pipe.compressed + nbuf > (int)j; });
                    if (pipe.abort) break;
                }
                const uint64_t lo = jobs[j].lo, len = jobs[j].hi - jobs[j].lo;
                uint8_t *hb = (uint8_t *)h_in[j % nbuf];
                const double t0 = now_s();
                const bool okr = pread_parallel(fd, hb, len, lo);
                w->io_s += now_s() - t0;
                if (!okr) { rc = NGSC_ERR_IO; break; }
                if (hipMemcpyAsync(d_in[j % nbuf], hb, len, hipMemcpyHostToDevice, in_stream) != hipSuccess ||
                    hipStreamSynchronize(in_stream) != hipSuccess) { rc = NGSC_ERR_HIP; break; }
                { std::lock_guard<std::mutex> lk(pipe.m); pipe.uploaded = (int)j + 1; }
                pipe.cv.notify_all();
            }
            if (fd >= 0) ::close(fd);
            if (rc) pipe.fail(rc);
        });
        // ---- writer: device -> pinned, block plan, pwrite at the group's place ----
        writer = std::thread([&]() {
            int rc = NGSC_OK;
            if (hipSetDevice(dev) != hipSuccess) rc = NGSC_ERR_HIP;
            // the group being collected
            struct RankAcc { uint32_t rank; int32_t ov; std::vector<uint32_t> sb; std::vector<uint8_t> carry; uint64_t carry_base = 0;
                             uint32_t emitted = 0; uint32_t last_size = 0; };
            std::vector<RankAcc> acc;         // ranks of the current group, in rank order (carry: a rank's stream when the
                                              // group came in more than one job)
            bool holding_turn = false;        // a single rank walked in parts writes its blocks as they fill
            std::vector<ngsc_block_t> blocks;
            auto plan = [&](RankAcc &ra) -> int {
                uint64_t bytes = 0;
                for (uint32_t x : ra.sb) bytes += x;
                const uint32_t capb = (uint32_t)(bytes / (NGSC_BLOCK_SIZE - 4096)) + 4;
                blocks.resize(capb);
                uint32_t nb = 0;
                int r = ngsc_plan_blocks(P, ra.rank, ra.sb.data(), (uint32_t)ra.sb.size(), blocks.data(), capb, &nb);
                blocks.resize(r ? 0 : nb);
                return r;
            };
            for (size_t j = 0; j < jobs.size() && !rc; ++j) {
                {
                    std::unique_lock<std::mutex> lk(pipe.m);
                    pipe.cv.wait(lk, [&] { return pipe.abort || pipe.compressed > (int)j; });
                    if (pipe.abort) break;
                }
                const Job &jb = jobs[j];
                const JobResult &res = results[j];
                uint8_t *hb = (uint8_t *)h_out[j % nbuf];
                double t0 = now_s();
                if (res.total && (hipMemcpyAsync(hb, d_out[j % nbuf], res.total, hipMemcpyDeviceToHost, out_stream) != hipSuccess ||
                                  hipStreamSynchronize(out_stream) != hipSuccess)) { rc = NGSC_ERR_HIP; break; }
                w->t_d2h += now_s() - t0;
                t0 = now_s();
                const bool single_job_group = !jb.part && jb.last_of_group && acc.empty();
                // collect the job's ranks
                const size_t acc0 = acc.size();
                for (size_t k = 0; k < res.ranks.size(); ++k) {
                    const ngsc_rank_t &rk = res.ranks[k];
                    // (round 3) a rank walked in parts whose blocks are written as they fill (jb.stream) keeps only what is left over
                    // behind the last written block (less than a block): its new bytes are written straight from the pinned buffer.
                    // Until round 3 every byte of such a rank went through a std::vector first - a single-threaded copy of the whole
                    // output, which is what the compress thread was waiting for ("output buffer" in the log).
                    const bool direct = jb.part && jb.stream;
                    if (jb.part && !jb.first_part) {   // the same rank goes on
                        RankAcc &ra = acc.back();
                        ra.sb.insert(ra.sb.end(), res.sb_sizes.begin(), res.sb_sizes.end());
                        if (!direct) ra.carry.insert(ra.carry.end(), hb + rk.stream_off, hb + rk.stream_off + rk.stream_bytes);
                    } else {
                        RankAcc ra;
                        ra.rank = rk.rank;
                        ra.ov = jb.part ? res.part.overlap_used : rk.overlap_used;
                        ra.sb.assign(res.sb_sizes.begin() + rk.first_subblock, res.sb_sizes.begin() + rk.first_subblock + rk.n_subblocks);
                        if (!single_job_group && !direct) ra.carry.assign(hb + rk.stream_off, hb + rk.stream_off + rk.stream_bytes);
                        acc.push_back(std::move(ra));
                    }
                }
                (void)acc0;
                if (jb.part && jb.stream) {
                    // ---- one rank, its blocks written as they fill (the group is this rank: its blocks follow each other) ----
                    RankAcc &ra = acc.back();
                    if (!holding_turn) {
                        if (!sink->begin_group(jb.group)) { rc = NGSC_ERR_IO; break; }
                        holding_turn = true;
                    }
                    if ((rc = plan(ra))) break;
                    const bool done = res.part.done != 0;
                    const uint32_t upto = done ? (uint32_t)blocks.size() : (blocks.empty() ? 0 : (uint32_t)blocks.size() - 1);
                    // the rank's stream so far: [carry_base, new_base) in ra.carry, [new_base, new_base + new_len) in the pinned buffer
                    const uint8_t *newp = hb + res.ranks[0].stream_off;
                    const uint64_t new_len = res.ranks[0].stream_bytes, new_base = ra.carry_base + ra.carry.size();
                    uint64_t drop_to = ra.carry_base;
                    std::vector<WriteItem> items;
                    std::vector<std::vector<uint8_t>> joined;   // a block that begins in the carry: its bytes put together (one per job at most)
                    joined.reserve(upto > ra.emitted ? upto - ra.emitted : 0);
                    for (uint32_t k = ra.emitted; k < upto; ++k) {
                        const ngsc_block_t &b = blocks[k];
                        const uint8_t *src;
                        if (b.stream_off >= new_base) src = newp + (b.stream_off - new_base);
                        else {
                            const uint64_t from_carry = std::min<uint64_t>(b.payload, new_base - b.stream_off);
                            joined.emplace_back();
                            std::vector<uint8_t> &t = joined.back();
                            t.resize(b.payload);
                            memcpy(t.data(), ra.carry.data() + (b.stream_off - ra.carry_base), from_carry);
                            if (b.payload > from_carry) memcpy(t.data() + from_carry, newp, b.payload - from_carry);
                            src = t.data();
                        }
                        items.push_back({sink->cursor, b.header, b.header_len, src, b.payload});
                        sink->cursor += b.size;            // (this worker holds the turn: nobody else moves the cursor)
                        sink->cbo.push_back(ra.rank);
                        ra.last_size = b.size;
                        drop_to = b.stream_off + b.payload;
                    }
                    if (!write_items(sink->fd, items)) { rc = NGSC_ERR_IO; break; }
                    ra.emitted = upto;
                    // what is left for the next part: the stream behind the last written block
                    if (drop_to >= new_base) ra.carry.assign(newp + (drop_to - new_base), newp + new_len);
                    else {
                        ra.carry.erase(ra.carry.begin(), ra.carry.begin() + (size_t)(drop_to - ra.carry_base));
                        ra.carry.insert(ra.carry.end(), newp, newp + new_len);
                    }
                    ra.carry_base = drop_to;
                    if (done) {
                        sink->ov[ra.rank] = ra.ov; sink->lbs[ra.rank] = ra.last_size; sink->nblocks[ra.rank] = ra.emitted;
                        sink->n_sub_total += (uint32_t)ra.sb.size();
                    }
                }
                if (jb.last_of_group) {
                    if (holding_turn) {
                        holding_turn = false;
                        sink->end_group();
                    } else {
                        // ---- whole ranks: plan every rank's blocks, take the group's place, write round by round ----
                        std::vector<std::vector<ngsc_block_t>> rb(acc.size());
                        uint32_t maxb = 0;
                        uint64_t gbytes = 0;
                        for (size_t k = 0; k < acc.size() && !rc; ++k) {
                            if ((rc = plan(acc[k]))) break;
                            rb[k] = blocks;
                            maxb = std::max<uint32_t>(maxb, (uint32_t)rb[k].size());
                            for (auto &b : rb[k]) gbytes += b.size;
                        }
                        if (rc && !w->team) break;
                        if (w->team) {
                            // ---- several GPUs: this group's rows into the round's exchange, offsets out of everybody's rows ----
                            std::vector<uint64_t> mine((size_t)rpg * row_len, 0);
                            for (size_t k = 0; k < acc.size() && !rc; ++k) {
                                if (rb[k].size() + 3 > row_len || k >= rpg) { rc = NGSC_ERR_CAPACITY; break; }
                                uint64_t *row = mine.data() + k * row_len;
                                row[0] = rb[k].size();
                                row[1] = acc[k].sb.size();
                                row[2] = (uint64_t)(int64_t)acc[k].ov;
                                for (size_t b = 0; b < rb[k].size(); ++b) row[3 + b] = rb[k][b].size;
                            }
                            std::vector<std::vector<uint64_t>> my_off;
                            rc = exchange_round(w, jb.group / w->n_workers, rpg, row_len, P, n_groups, rc, mine, my_off);
                            if (rc) break;
                            rounds_done = jb.group / w->n_workers + 1;
                            std::vector<WriteItem> items;
                            for (size_t r = 0; r < acc.size(); ++r) {
                                const uint8_t *stream = single_job_group ? hb + res.ranks[r].stream_off : acc[r].carry.data();
                                for (size_t b = 0; b < rb[r].size(); ++b)
                                    items.push_back({my_off[r][b], rb[r][b].header, rb[r][b].header_len, stream + rb[r][b].stream_off, rb[r][b].payload});
                            }
                            if (!write_items(sink->fd, items)) { rc = NGSC_ERR_IO; break; }
                            acc.clear();
                            w->t_write += now_s() - t0;
                            { std::lock_guard<std::mutex> lk(pipe.m); pipe.written = (int)j + 1; }
                            pipe.cv.notify_all();
                            continue;
                        }
                        if (!sink->begin_group(jb.group)) { rc = NGSC_ERR_IO; break; }
                        uint64_t off = sink->cursor;
                        sink->cursor += gbytes;
                        std::vector<std::pair<uint64_t, std::pair<size_t, uint32_t>>> todo;   // offset, (rank slot, block)
                        for (uint32_t k = 0; k < maxb; ++k)
                            for (size_t r = 0; r < acc.size(); ++r)
                                if (k < rb[r].size()) {
                                    todo.push_back({off, {r, k}});
                                    off += rb[r][k].size;
                                    sink->cbo.push_back(acc[r].rank);
                                }
                        for (size_t r = 0; r < acc.size(); ++r) {
                            sink->ov[acc[r].rank] = acc[r].ov;
                            sink->lbs[acc[r].rank] = rb[r].empty() ? 0 : rb[r].back().size;
                            sink->nblocks[acc[r].rank] = (uint32_t)rb[r].size();
                            sink->n_sub_total += (uint32_t)acc[r].sb.size();
                        }
                        sink->end_group();   // the place is taken: later groups may go on while these bytes are written
                        // the blocks are independent writes: a few threads share them
                        std::vector<WriteItem> items;
                        for (auto &t : todo) {
                            const size_t r = t.second.first;
                            const ngsc_block_t &b = rb[r][t.second.second];
                            const uint8_t *stream = single_job_group ? hb + res.ranks[r].stream_off : acc[r].carry.data();
                            items.push_back({t.first, b.header, b.header_len, stream + b.stream_off, b.payload});
                        }
                        if (!write_items(sink->fd, items)) { rc = NGSC_ERR_IO; break; }
                    }
                    acc.clear();
                }
                w->t_write += now_s() - t0;
                { std::lock_guard<std::mutex> lk(pipe.m); pipe.written = (int)j + 1; }
                pipe.cv.notify_all();
            }
            bool stopped;
            { std::lock_guard<std::mutex> lk(pipe.m); stopped = pipe.abort; }
            if (rc) { pipe.fail(rc); sink->abort(); if (w->team) ngsc_team_abort(w->team, rc); }
            else if (stopped) { if (w->team) ngsc_team_abort(w->team, pipe.rc ? pipe.rc : (int)NGSC_ERR_IO); }   // (the reader or the compress thread gave up)
            else idle_rounds(NGSC_OK);
        });
    }
    // ---- the worker thread itself: compress job after job ----
    ngsc_rank_part_t part;
    memset(&part, 0, sizeof part);
    for (size_t j = 0; j < jobs.size() && !w->rc; ++j) {
        double t0 = now_s();
        {
            std::unique_lock<std::mutex> lk(pipe.m);
            pipe.cv.wait(lk, [&] { return pipe.abort || pipe.uploaded > (int)j; });
            if (pipe.abort) break;
        }
        w->t_wait_in += now_s() - t0;
        t0 = now_s();
        {   // output buffer j % nbuf is free once job j - nbuf has been written
            std::unique_lock<std::mutex> lk(pipe.m);
            pipe.cv.wait(lk, [&] { return pipe.abort || pipe.written + nbuf > (int)j; });
            if (pipe.abort) break;
        }
        w->t_wait_out += now_s() - t0;
        t0 = now_s();
        const Job &jb = jobs[j];
        int rc;
        if (jb.part) {
            if (jb.first_part) memset(&part, 0, sizeof part);
            else if (part.done || part.next_pos < jb.lo || part.next_pos >= jb.hi) { pipe.fail(NGSC_ERR_UNSUPPORTED); break; }   // a record longer than the parts' overlap
            rc = ngsc_compress_rank_part(w->ctx, d_in[j % nbuf], jb.hi - jb.lo, jb.lo, FS, P, jb.a, &part, d_out[j % nbuf], out_cap, nullptr);
            if (!rc && jb.last_part != (part.done != 0)) rc = NGSC_ERR_UNSUPPORTED;
        } else {
            rc = ngsc_compress_device(w->ctx, d_in[j % nbuf], jb.hi - jb.lo, jb.lo, FS, P, jb.a, jb.b, d_out[j % nbuf], out_cap, nullptr);
        }
        if (rc) { pipe.fail(rc); break; }
        w->t_comp += now_s() - t0;
        JobResult &res = results[j];
        res.total = ngsc_result_bytes(w->ctx);
        res.part = part;
        const ngsc_subblock_t *sbs = ngsc_result_subblocks(w->ctx);
        const ngsc_rank_t *rks = ngsc_result_ranks(w->ctx);
        const uint32_t nr = ngsc_result_num_ranks(w->ctx), ns = ngsc_result_num_subblocks(w->ctx);
        res.ranks.assign(rks, rks + nr);
        res.sb_sizes.resize(ns);
        for (uint32_t k = 0; k < ns; ++k) res.sb_sizes[k] = sbs[k].info_len + sbs[k].title_len + sbs[k].qual_len + sbs[k].dna_len;
        { std::lock_guard<std::mutex> lk(pipe.m); pipe.compressed = (int)j + 1; }
        pipe.cv.notify_all();
    }
    {   // let the threads run out (or stop them)
        std::unique_lock<std::mutex> lk(pipe.m);
        pipe.cv.wait(lk, [&] { return pipe.abort || pipe.written >= (int)jobs.size(); });
    }
    if (reader.joinable()) reader.join();
    if (writer.joinable()) writer.join();
    if (!w->rc && pipe.rc) w->rc = pipe.rc;
    if (w->rc) { sink->abort(); if (w->team) ngsc_team_abort(w->team, w->rc); }
    const double t_free = now_s();
    for (int i = 0; i < 2; ++i) {
        if (d_in[i]) (void)hipFree(d_in[i]);
        if (h_in[i]) (void)hipHostFree(h_in[i]);
        if (d_out[i]) (void)hipFree(d_out[i]);
        if (h_out[i]) (void)hipHostFree(h_out[i]);
    }
    if (in_stream) (void)hipStreamDestroy(in_stream);
    if (out_stream) (void)hipStreamDestroy(out_stream);
    if (getenv("NGSC_VERBOSE"))
        fprintf(stderr, "[ngsc] device %d: %zu jobs, buffers %.3f s (alloc, before the first job), waits: input %.3f s, output buffer %.3f s; "
                        "compress %.3f s; beside it: file read %.3f s, D2H %.3f s, plan + write %.3f s (of it in the exchange of block rows: %.3f s); free %.3f s\n", dev, jobs.size(), t_alloc,
                w->t_wait_in, w->t_wait_out, w->t_comp, w->io_s, w->t_d2h, w->t_write, w->t_xchg, now_s() - t_free);
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
    Sink sink;
    sink.fd = ::open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (sink.fd < 0) return NGSC_ERR_IO;
    // (round 3) room for the output is reserved up front - a guess of 40 % of the input, cut back to the real size at the end:
    // on the GPU box's file system eight threads write preallocated page-cache pages at 16 GB/s and fresh ones at 10.5
    // (tools/ubench_write.c, profiles/r03_ubench_write.txt).  fallocate(2), not posix_fallocate: a file system without it says
    // so instead of being filled with zeros.  NGSC_FALLOCATE=0: off
    bool reserved = false;
    if (!(getenv("NGSC_FALLOCATE") && !strcmp(getenv("NGSC_FALLOCATE"), "0")))
        reserved = ::fallocate(sink.fd, 0, 0, (off_t)(FS / 5 * 2 + (16u << 20))) == 0;
    sink.ov.assign(P, 0);
    sink.lbs.assign(P, 0);
    sink.nblocks.assign(P, 0);
    double t1 = now_s();

    // the groups of the block order are dealt to the contexts in turn: group g to context g % n_ctx
    std::vector<Worker> workers(n_ctx);
    for (uint32_t i = 0; i < n_ctx; ++i) { workers[i].ctx = ctxs[i]; workers[i].index = i; workers[i].n_workers = n_ctx; }
    ngsc_team *team = nullptr;
    if (n_ctx == 1) run_worker(&workers[0], &sink, in_path, FS, P);
    else {
        // several GPUs: where a group's blocks land is agreed through the team's collective (RCCL over the GPUs' communicator;
        // host memory when RCCL is not to be had - two contexts on one device, no library), not through a shared cursor
        if (int trc = ngsc_team_create(ctxs, n_ctx, &team)) { ::close(sink.fd); return trc; }
        if (getenv("NGSC_VERBOSE")) fprintf(stderr, "[ngsc] %u contexts, exchange of block rows: %s\n", n_ctx, ngsc_team_describe(team));
        for (auto &w : workers) { w.team = team; w.ov.assign(P, 0); w.lbs.assign(P, 0); w.nblocks.assign(P, 0); }
        std::vector<std::thread> th;
        for (uint32_t i = 0; i < n_ctx; ++i) th.emplace_back(run_worker, &workers[i], &sink, in_path, FS, P);
        for (auto &t : th) t.join();
        // (every worker holds the same tables: worker 0's go into the footer)
        sink.cbo = workers[0].cbo; sink.ov = workers[0].ov; sink.lbs = workers[0].lbs; sink.nblocks = workers[0].nblocks;
        sink.n_sub_total = workers[0].n_sub_total; sink.cursor = workers[0].cursor;
        ngsc_team_destroy(team);
    }
    double t2 = now_s();
    int rc = NGSC_OK;
    for (auto &w : workers)
        if (w.rc) { rc = w.rc; break; }
    bool okw = !sink.fail;
    if (!rc && okw) {
        // footer (phyNGSC.cpp:930-1057)
        const uint32_t n_blocks_total = (uint32_t)sink.cbo.size();
        std::vector<uint8_t> footer(256 + 4 * (size_t)n_blocks_total + 8 * (size_t)P);
        uint32_t fl = ngsc_make_footer(footer.data(), P, FS, n_blocks_total, sink.n_sub_total, sink.ov.data(), sink.cbo.data(), sink.lbs.data());
        okw = pwrite_all(sink.fd, footer.data(), fl, sink.cursor);
        if (okw && reserved && ::ftruncate(sink.fd, (off_t)(sink.cursor + fl)) != 0) okw = false;
    }
    ::close(sink.fd);
    double t3 = now_s();
    double io = 0;
    for (auto &w : workers) io = std::max(io, w.io_s);
    if (seconds_out) { seconds_out[0] = (t1 - t0) + io; seconds_out[1] = (t2 - t1) - io; seconds_out[2] = t3 - t2; }
    if (rc || !okw) (void)::unlink(out_path);   // (no half-written, preallocated file left behind)
    if (rc) return rc;
    return okw ? NGSC_OK : NGSC_ERR_IO;
}

int ngsc_compress_file(ngsc_ctx *ctx, const char *in_path, const char *out_path, uint32_t P, double *seconds_out)
{
    return ngsc_compress_file_multi(&ctx, 1, in_path, out_path, P, seconds_out);
}

}  // extern "C"
