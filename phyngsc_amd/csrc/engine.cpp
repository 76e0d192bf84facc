// engine.cpp - host orchestration of the HIP kernels + the C ABI of libngsc_hip.so (include/ngsc.h).
//
// One context = one GPU + one stream + a grow-only device workspace.  A call to
// ngsc_compress_device() runs, for a range of virtual ranks:
//   sync_first -> nl_count/scan -> [sync: line count] -> nl_write, validate, cuts -> [sync: cut list]
//   -> title_stats, seq_qual_stats, finalize, title_hist, huffman, measure, scan/layout
//   -> zero(out), headers, length table, title pack, quality/DNA pack -> [sync: error word, sizes, result tables]
// There is no CPU implementation of any of these stages in this library.
#include "../../include/ngsc.h"
#include "ngsc_kernels.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

using namespace ngsc;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct KTime {
    const char *name;
    double ms = 0;
    uint32_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace

// What the host has to know about a job before it can size the per-record arrays and choose kernels: the planner's counts and
// the job-wide maxima of the validation pass.  A first call reads them back behind the index kernels (one host sync in the
// middle of the step).  Later calls of the same shape PREDICT them from the call before - jobs of one file look alike, the bench
// repeats one job - enqueue the whole step at once, and let k_init_sb compare the prediction with what the index kernels found:
// on a miss every kernel behind it leaves at once (the error word), and the step runs again the synchronous way.
struct JobShape {
    uint32_t valid = 0;
    // the calls it applies to
    uint64_t file_size = 0;
    uint32_t P = 0, n_ranks = 0, part = 0;
    uint64_t buf_len = 0;
    // counts (capacities when predicted: the actual numbers may be smaller) ...
    uint32_t n_rec_total = 0, S = 0, max_rec = 0;
    uint64_t slots = 0, groups = 0, tiles = 0;
    // ... and maxima (a predicted maximum must not be exceeded)
    uint32_t max_title = 0, max_q = 0, max_rec_bytes = 0;
};

struct ngsc_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t side_stream = nullptr;          // the latency-bound kernels of the staggered schedule (high priority)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_zero = nullptr;
    std::vector<Cut> cuts_tmp;                  // a call's cut list while its result tables are made
    size_t zero_hint = 0;         // bytes of the zero-initialised arena the last call used: cleared while the host waits for the cut list
    uint64_t out_hint = 0;        // output bytes of the last call: that much of the output buffer is cleared beside the index kernels
    std::string hip_err;
    bool profiling = false;
    bool debug_sync = getenv("NGSC_DEBUG_SYNC") != nullptr;
    std::string profile_only;   // when set, only this kernel is timed (two event records per call instead of ~60)
    uint32_t min_text_slots = 0;  // text-field table slots the last call turned out to need (kept for the next call)
    bool cs_retry = false;        // the job is run again on a copy in which colour-space reads are decoded
    uint32_t qbits_x16 = 0, dbits_x16 = 0;   // payload bits per base (x 16) of the densest sub-block of the last call: staging hint
    bool slot_retry = false;
    bool wide = false;            // titles of more than MAX_FIELDS separators: the job runs on the wide title kernels (kernels_title_wide.hip);
                                  // found out by a first run, kept for the context's later jobs while they need it
    bool no_spec = getenv("NGSC_NO_SPEC") != nullptr;   // every call reads the job's shape back (the step's first-call form)
    JobShape shape;               // the last job's counts and maxima: the next call's prediction
    bool big_alphabets = false;   // the last call met a quality / base alphabet of more than 64 symbols (grid of the 512-symbol Huffman instance)
    uint32_t title_need = 3;      // the last call's Arena::num_huff word: which of the title's histogram / tree launches had work (all, until a call has been through)
    std::vector<KTime> ktimes;
    std::vector<hipEvent_t> event_pool;

    // workspace
    DevBuf small, tiles, recs, recs2, persb, zero, slots, misc, cscopy, dense;
    DevBuf dec[4];                // the decoder's working memory (ngsc_ctx_scratch), kept between calls
    // pinned host staging
    void *h_pin = nullptr;
    size_t h_pin_cap = 0;

    // results
    std::vector<ngsc_subblock_t> res_sb;
    std::vector<ngsc_rank_t> res_rank;
    uint64_t res_bytes = 0;
};

#define HIPCHK(ctx, expr)                                                                       \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (ctx)->hip_err = std::string(#expr) + ": " + hipGetErrorString(_e);                 \
            return NGSC_ERR_HIP;                                                                \
        }                                                                                       \
    } while (0)

static int ensure(ngsc_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return NGSC_OK;
    if (b.p) HIPCHK(c, hipFree(b.p));
    b.p = nullptr;
    b.cap = 0;
    size_t want = bytes + bytes / 8 + 4096;
    if (hipMalloc(&b.p, want) != hipSuccess) {
        // the decoder's working memory stays with the context between calls (several GB after a large file): a compress job
        // that finds no room gives it back and tries once more
        (void)hipGetLastError();
        b.p = nullptr;
        bool freed = false;
        for (DevBuf &d : c->dec)
            if (d.p) { (void)hipFree(d.p); d.p = nullptr; d.cap = 0; freed = true; }
        if (!freed || hipMalloc(&b.p, want) != hipSuccess) {
            (void)hipGetLastError();
            b.p = nullptr;
            c->hip_err = "hipMalloc: out of device memory";
            return NGSC_ERR_NOMEM;
        }
    }
    b.cap = want;
    return NGSC_OK;
}

static int ensure_pinned(ngsc_ctx *c, size_t bytes)
{
    if (bytes <= c->h_pin_cap) return NGSC_OK;
    if (c->h_pin) HIPCHK(c, hipHostFree(c->h_pin));
    c->h_pin = nullptr;
    c->h_pin_cap = 0;
    HIPCHK(c, hipHostMalloc(&c->h_pin, bytes + 4096, hipHostMallocDefault));
    c->h_pin_cap = bytes + 4096;
    return NGSC_OK;
}

// carve helper: 256-byte aligned sub-allocations of one DevBuf
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes)
    {
        size_t o = off;
        off = (off + bytes + 255) & ~(size_t)255;
        return o;
    }
};

// ---- profiling ------------------------------------------------------------------------------------
struct KScope {
    ngsc_ctx *c;
    hipStream_t st;
    int idx = -1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const char *name_;
    KScope(ngsc_ctx *ctx, hipStream_t s, const char *name) : c(ctx), st(s), name_(name)
    {
        if (!c->profiling) return;
        if (!c->profile_only.empty() && c->profile_only != name) return;
        for (size_t i = 0; i < c->ktimes.size(); ++i)
            if (c->ktimes[i].name == name || !strcmp(c->ktimes[i].name, name)) idx = (int)i;
        if (idx < 0) {
            KTime k;
            k.name = name;
            c->ktimes.push_back(k);
            idx = (int)c->ktimes.size() - 1;
        }
        auto get = [&]() {
            hipEvent_t e;
            if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
            else (void)hipEventCreate(&e);
            return e;
        };
        e0 = get();
        e1 = get();
        (void)hipEventRecord(e0, st);
    }
    ~KScope()
    {
        if (c->debug_sync) {   // NGSC_DEBUG_SYNC=1: name the launch that failed
            hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(st);
            if (e1 != hipSuccess || e2 != hipSuccess)
                fprintf(stderr, "[ngsc] %s: launch %s, run %s\n", name_, hipGetErrorString(e1), hipGetErrorString(e2));
        }
        if (idx < 0) return;
        (void)hipEventRecord(e1, st);
        c->ktimes[idx].pending.emplace_back(e0, e1);
        c->ktimes[idx].launches++;
    }
};

static void collect_times(ngsc_ctx *c)
{
    for (auto &k : c->ktimes) {
        for (auto &pr : k.pending) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) k.ms += ms;
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
        k.pending.clear();
    }
}

// ====================================================================================================
extern "C" {

int ngsc_version(void) { return NGSC_VERSION; }

const char *ngsc_strerror(int code)
{
    switch (code) {
    case NGSC_OK: return "ok";
    case NGSC_ERR_FORMAT: return "input is not 4-line FASTQ with bare '+' lines and equal sequence/quality lengths";
    case NGSC_ERR_FIELDS: return "title separator count differs inside a sub-block";
    case NGSC_ERR_TOOMANY: return "more than 100000 records in one sub-block";
    case NGSC_ERR_COLORSPACE: return "colour-space (SOLiD) sub-block: stored lossily by the format, cannot be decoded";
    case NGSC_ERR_RANGE: return "numeric title field out of int32 range";
    case NGSC_ERR_NOMEM: return "out of memory";
    case NGSC_ERR_ARG: return "invalid argument";
    case NGSC_ERR_UNSUPPORTED: return "engine limit exceeded (title fields / text field length / code length / rank regions of one record's size)";
    case NGSC_ERR_HIP: return "HIP runtime error";
    case NGSC_ERR_CAPACITY: return "output buffer too small";
    case NGSC_ERR_IO: return "I/O error";
    default: return "unknown error";
    }
}

int ngsc_ctx_create(int device_id, ngsc_ctx **out)
{
    if (!out) return NGSC_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n) return NGSC_ERR_HIP;
    ngsc_ctx *c = new ngsc_ctx();
    c->device = device_id;
    // The side stream carries the latency-bound kernels of the staggered schedule (sub-block finalisation, Huffman waves, headers)
    // beside a throughput kernel on the main stream: with a higher priority its few workgroups are dispatched as soon as a
    // slot frees up instead of taking turns with tens of thousands of queued ones
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const int side_prio = prio_hi;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreate(&c->own_stream) != hipSuccess ||
        hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, side_prio) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_zero, hipEventDisableTiming) != hipSuccess) {
        delete c;
        return NGSC_ERR_HIP;
    }
    *out = c;
    return NGSC_OK;
}

void ngsc_ctx_destroy(ngsc_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    for (DevBuf *b : {&c->small, &c->tiles, &c->recs, &c->recs2, &c->persb, &c->zero, &c->slots, &c->misc, &c->cscopy, &c->dense})
        if (b->p) (void)hipFree(b->p);
    for (DevBuf &b : c->dec) if (b.p) (void)hipFree(b.p);
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    collect_times(c);
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_zero) (void)hipEventDestroy(c->ev_zero);
    delete c;
}

const char *ngsc_last_hip_error(ngsc_ctx *c) { return c ? c->hip_err.c_str() : ""; }

// Grow-only device memory of the context for the decoder (decode.cpp), slot 0..3: kept between calls, so that a file decoded
// batch by batch does not pay a multi-gigabyte hipMalloc / hipFree - and the device-wide wait a free implies - per batch.
// nullptr: no memory (the caller falls back or gives up).  Not part of the public ABI.
void *ngsc_ctx_scratch(ngsc_ctx *c, uint32_t slot, size_t bytes)
{
    if (!c || slot >= 4) return nullptr;
    DevBuf &b = c->dec[slot];
    if (bytes <= b.cap) return b.p;
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
    if (hipMalloc(&b.p, bytes) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; return nullptr; }
    b.cap = bytes;
    return b.p;
}

int ngsc_ctx_device(ngsc_ctx *c) { return c ? c->device : -1; }

int ngsc_ctx_trim(ngsc_ctx *c)
{
    if (!c) return NGSC_ERR_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return NGSC_ERR_HIP;
    for (DevBuf &d : c->dec)
        if (d.p) { (void)hipFree(d.p); d.p = nullptr; d.cap = 0; }
    return NGSC_OK;
}

int ngsc_set_profiling(ngsc_ctx *c, int enable)
{
    if (!c) return NGSC_ERR_ARG;
    c->profiling = enable != 0;
    return NGSC_OK;
}

int ngsc_profile_only(ngsc_ctx *c, const char *kernel_name)
{
    if (!c) return NGSC_ERR_ARG;
    c->profile_only = kernel_name ? kernel_name : "";
    return NGSC_OK;
}

int ngsc_reset_kernel_times(ngsc_ctx *c)
{
    if (!c) return NGSC_ERR_ARG;
    collect_times(c);
    for (auto &k : c->ktimes) { k.ms = 0; k.launches = 0; }
    return NGSC_OK;
}

uint32_t ngsc_get_kernel_times(ngsc_ctx *c, ngsc_kernel_time_t *out, uint32_t cap)
{
    if (!c) return 0;
    collect_times(c);
    uint32_t n = 0;
    for (auto &k : c->ktimes) {
        if (out && n < cap) { out[n].name = k.name; out[n].ms = k.ms; out[n].launches = k.launches; }
        ++n;
    }
    return n;
}

uint64_t ngsc_result_bytes(ngsc_ctx *c) { return c ? c->res_bytes : 0; }
uint32_t ngsc_result_num_subblocks(ngsc_ctx *c) { return c ? (uint32_t)c->res_sb.size() : 0; }
uint32_t ngsc_result_num_ranks(ngsc_ctx *c) { return c ? (uint32_t)c->res_rank.size() : 0; }
const ngsc_subblock_t *ngsc_result_subblocks(ngsc_ctx *c) { return c ? c->res_sb.data() : nullptr; }
const ngsc_rank_t *ngsc_result_ranks(ngsc_ctx *c) { return c ? c->res_rank.data() : nullptr; }

}  // extern "C"

// The whole path for ranks [rank_lo, rank_hi).  part != nullptr: ONE rank, walked in several calls (ngsc_compress_rank_part):
// from tells where its loop stands (resume == 0: at its first record), the buffer may end before the rank does, and
// *part receives where the loop stands afterwards.
#ifdef NGSC_HOST_TIMING
// developer build: where a call's host time goes (stderr, every 16 calls): outside = between the previous call's return and this
// call's entry (the caller), first = entry -> first launch enqueued, enqueue = entry -> everything enqueued, wait = the one wait,
// tail = wait over -> return
namespace {
struct HostTiming {
    double t_entry = 0, t_first = 0, t_enq = 0, t_waited = 0, t_prev_ret = 0, acc[5] = {0, 0, 0, 0, 0};
    int n = 0;
    static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    void done()
    {
        const double t_ret = now();
        if (t_prev_ret > 0 && t_first > 0 && t_enq > 0 && t_waited > 0) {
            acc[0] += t_entry - t_prev_ret; acc[1] += t_first - t_entry; acc[2] += t_enq - t_entry; acc[3] += t_waited - t_enq; acc[4] += t_ret - t_waited;
            if (++n == 16) {
                fprintf(stderr, "[host us] outside %.1f first %.1f enqueue %.1f wait %.1f tail %.1f\n", acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n);
                n = 0; for (double &a : acc) a = 0;
            }
        }
        t_prev_ret = t_ret; t_first = t_enq = t_waited = 0;
    }
};
thread_local HostTiming g_ht;
}
#define HT(x) g_ht.x = HostTiming::now()
#else
#define HT(x) (void)0
#endif

static int compress_impl(ngsc_ctx *c, const void *d_buf, uint64_t buf_len, uint64_t buf_off, uint64_t file_size,
                         uint32_t P, uint32_t rank_lo, uint32_t rank_hi, void *d_out, uint64_t out_cap, void *hip_stream,
                         const PlanResume &from, ngsc_rank_part_t *part)
{
    if (!c || !d_buf || !d_out || P < 2 || rank_lo >= rank_hi || rank_hi > P || !file_size) return NGSC_ERR_ARG;
    if (buf_len == 0 || buf_len >= 0xFFFFFF00ull || buf_off + buf_len > file_size) return NGSC_ERR_ARG;
    if (((uintptr_t)d_buf & 15) || ((uintptr_t)d_out & 15)) return NGSC_ERR_ARG;
    if (file_size / P == 0) return NGSC_ERR_ARG;
    const uint32_t n_ranks = rank_hi - rank_lo;
    if (part && n_ranks != 1) return NGSC_ERR_ARG;
    {
        RankGeom g0 = rank_geometry(file_size, P, rank_lo), g1 = rank_geometry(file_size, P, rank_hi - 1);
        if (from.resume) {
            if (buf_off > from.next_pos || from.next_pos >= buf_off + buf_len) return NGSC_ERR_ARG;
        } else if (rank_lo == 0 ? buf_off != 0 : (int64_t)buf_off > g0.wr_start) return NGSC_ERR_ARG;
        // the buffer must reach the end of the last rank's region (+ the reference's overlap) or EOF - unless the rank is
        // walked in parts
        uint64_t need_end = (uint64_t)g1.wr_end + 1;
        if (need_end > file_size) need_end = file_size;
        if (!part && buf_off + buf_len < need_end) return NGSC_ERR_ARG;
    }
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    c->res_sb.clear();
    c->res_rank.clear();
    c->res_bytes = 0;
    const uint8_t *buf = (const uint8_t *)d_buf;
    const uint32_t len = (uint32_t)buf_len;

    // Is there a prediction for this call?  (Not while every kernel is being timed, not for the second run of a job - colour
    // space, text slots, a missed prediction - and not for a rank walked in parts, whose window count follows the buffer.)
    const bool time_all = c->profiling && c->profile_only.empty();
    const JobShape &last = c->shape;
    const bool spec = last.valid && !c->no_spec && !c->cs_retry && !c->slot_retry && !c->wide && !part && !c->debug_sync &&
                      last.file_size == file_size && last.P == P && last.n_ranks == n_ranks && last.part == 0 &&
                      buf_len <= last.buf_len + last.buf_len / 64 && last.buf_len <= buf_len + buf_len / 64;

    // ---- small device state ----
    const uint64_t region = file_size / P;
    const uint32_t max_cuts = (uint32_t)(region >> 22) + 8;
    Carver cs;
    size_t o_err = cs.take(4), o_total = cs.take(4), o_start = cs.take(4), o_maxq = cs.take(4), o_slotc = cs.take(4);
    size_t o_maxrec = cs.take(4), o_maxtitle = cs.take(4), o_done = cs.take(4), o_bigt = cs.take(4), o_wide = cs.take(4), o_numhuff = cs.take(4);
    size_t o_outtot = cs.take(8);
    size_t o_first = cs.take(sizeof(uint64_t) * n_ranks);
    size_t o_plans = cs.take(sizeof(RankPlan) * n_ranks);
    size_t o_cuts = cs.take(sizeof(Cut) * (size_t)n_ranks * max_cuts);
    int rc = ensure(c, c->small, cs.off);
    if (rc) return rc;
    uint8_t *sm = (uint8_t *)c->small.p;
    int32_t *d_err = (int32_t *)(sm + o_err);
    uint32_t *d_total = (uint32_t *)(sm + o_total), *d_start = (uint32_t *)(sm + o_start);
    uint32_t *d_maxq = (uint32_t *)(sm + o_maxq), *d_slotc = (uint32_t *)(sm + o_slotc);
    uint32_t *d_maxrec = (uint32_t *)(sm + o_maxrec), *d_maxtitle = (uint32_t *)(sm + o_maxtitle), *d_done = (uint32_t *)(sm + o_done);
    uint64_t *d_outtot = (uint64_t *)(sm + o_outtot), *d_first = (uint64_t *)(sm + o_first);
    RankPlan *d_plans = (RankPlan *)(sm + o_plans);
    Cut *d_cuts = (Cut *)(sm + o_cuts);
    // (the small state in front of the per-rank words is cleared by the step's first kernel, k_sync_first - by a kernel of ours in
    // front of it when the call has many ranks; the runtime's fill kernel left the GPU idle for ~20 us behind it)

    const uint32_t ntiles = index_num_tiles(len);
    // The record table is sized for the smallest possible record (5 bytes), so the line count never has to come back
    // to the host before the table is written.
    const size_t mask_words = index_mask_words(len);
    rc = ensure(c, c->tiles, 4 * (size_t)ntiles + 64 + 4 * mask_words);
    if (rc) return rc;
    uint32_t *d_tiles = (uint32_t *)c->tiles.p;
    uint32_t *d_nlmask = (uint32_t *)((uint8_t *)c->tiles.p + ((4 * (size_t)ntiles + 63) & ~(size_t)63));
    const size_t rec_bound = (size_t)len / 5 + 2;   // a record is four newlines and a '+' at the very least
    Carver cr;
    size_t o_te = cr.take(4 * rec_bound), o_se = cr.take(4 * rec_bound), o_ls = cr.take(4 * (rec_bound + 1));
    rc = ensure(c, c->recs, cr.off);
    if (rc) return rc;
    uint8_t *rb = (uint8_t *)c->recs.p;
    uint32_t *d_te = (uint32_t *)(rb + o_te), *d_se = (uint32_t *)(rb + o_se), *d_ls = (uint32_t *)(rb + o_ls);

    // The output bytes of the step are cleared EARLY - beside the index kernels' latency-bound tail - as far as the previous
    // call's output reached (jobs of one file are alike; the bench repeats one job); k_zero_output later clears only what
    // lies beyond.
    const char *ov_env = getenv("NGSC_OVERLAP");
    const bool serial = (ov_env && !strcmp(ov_env, "0")) || time_all || c->debug_sync;
    uint64_t out_prefix = 0;
    if (c->out_hint && !c->debug_sync) out_prefix = std::min<uint64_t>(c->out_hint + (c->out_hint >> 6), out_cap) & ~15ull;

    // ---- stage A: line index ----
    {
        KScope k(c, st, "k_sync_first");
        launch_sync_first(st, buf, len, buf_off, file_size, P, rank_lo, n_ranks, d_first, d_start, d_err, from.resume ? from.next_pos : 0, sm, o_first);
    }
    HT(t_first);
    { KScope k(c, st, "k_nl_count"); launch_nl_count(st, buf, len, d_start, d_tiles, d_nlmask, buf_off + buf_len == file_size); }
    { KScope k(c, st, "k_scan_tiles"); launch_scan_tiles(st, d_tiles, ntiles, d_total); }
    { KScope k(c, st, "k_nl_write"); launch_nl_write(st, d_nlmask, len, d_start, d_tiles, d_te, d_se, d_ls, (uint32_t)rec_bound); }
    // beside the cut planner / validation kernel (latency-bound: a chain of searches and a light pass over the record table), on
    // the side stream: the early clearing of the output and - when the step is enqueued in one go - of the call's zero arena
    const bool zero_on_side = !serial;
    void *const prezero_ptr = c->zero.p;
    const size_t prezero = std::min(c->zero_hint, c->zero.cap);
    // (round 4) ... the output's clearing only for longer records.  With short ones (under 128 bytes on average: 36-base reads) the
    // cut planner - 0.05 ms since it looks its windows up with sixteen waves - would wait for the clearing beside it (0.07 ms), while
    // the sequence/quality side chain behind the statistics kernel has 0.1 ms to spare beside the title statistics: the clearing
    // goes there (1.960 -> 1.941 ms per GB of 36-base reads, 1.928 -> 1.901 with the round's later schedule; 150-base and
    // 36..300-base jobs, whose side chain has no slack, lose 1-2 % that way: profiles/r04_ab_zero_late.txt).  The record count is the previous job's (the prediction) or none.
    const bool zero_late = zero_on_side && spec && (uint64_t)last.n_rec_total * 128 > (uint64_t)len;
    if (zero_on_side) {
        HIPCHK(c, hipEventRecord(c->ev_fork, st));
        HIPCHK(c, hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        if (out_prefix && !zero_late) launch_zero_range(c->side_stream, d_out, out_prefix);
        if (spec && prezero) launch_zero_range(c->side_stream, c->zero.p, prezero);
        HIPCHK(c, hipEventRecord(c->ev_join, c->side_stream));
    } else {
        if (out_prefix) { KScope k(c, st, "k_zero_prefix"); launch_zero_range(st, d_out, out_prefix); }
        if (spec && prezero) launch_zero_range(st, c->zero.p, prezero);
    }
    {
        KScope k(c, st, "k_cuts_validate");
        launch_cuts_validate(st, file_size, P, rank_lo, n_ranks, buf_off, d_te, d_se, d_ls, len / 96 + 1, d_total, d_first, d_cuts,
                             max_cuts, d_plans, d_maxq, d_maxrec, d_maxtitle, d_err, buf_off + buf_len == file_size ? buf : nullptr, from,
                             part ? buf_off + buf_len : 0, d_tiles, ntiles);
    }
    if (zero_on_side) HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));

    // ---- the job's shape: read back (one sync) or predicted ----
    rc = ensure_pinned(c, 2 * cs.off + 4096);
    if (rc) return rc;
    uint8_t *hp = (uint8_t *)c->h_pin;
    JobShape g;
    g.file_size = file_size; g.P = P; g.n_ranks = n_ranks; g.part = part ? 1u : 0u; g.buf_len = buf_len;
    if (spec) {
        g = last;
        g.buf_len = buf_len;
    } else {
        HIPCHK(c, hipMemcpyAsync(hp, sm, cs.off, hipMemcpyDeviceToHost, st));
        // while the host waits: the zero-initialised arena of this call, as large as the last call needed it
        if (prezero) launch_zero_range(st, c->zero.p, prezero);
        HIPCHK(c, hipStreamSynchronize(st));
        if (*(int32_t *)(hp + o_err)) return *(int32_t *)(hp + o_err);
        g.n_rec_total = *(uint32_t *)(hp + o_total) / 4;
        if (!g.n_rec_total || (size_t)g.n_rec_total + 2 > rec_bound) return NGSC_ERR_FORMAT;
        const RankPlan *h_plans = (const RankPlan *)(hp + o_plans);
        const Cut *h_cuts = (const Cut *)(hp + o_cuts);
        for (uint32_t i = 0; i < n_ranks; ++i) {
            for (uint32_t k = 0; k < h_plans[i].n_cuts; ++k) {
                const uint32_t n = h_cuts[(size_t)i * max_cuts + k].n_rec;
                ++g.S;
                g.max_rec = std::max(g.max_rec, n);
                g.slots += n;
                g.groups += (n + GROUP - 1) / GROUP;
                g.tiles += (n + 255) / 256;
            }
        }
        if (part) {
            part->next_pos = h_plans[0].next_pos; part->bytes_read = h_plans[0].bytes_read; part->bsize = h_plans[0].bsize;
            part->overlap = h_plans[0].overlap; part->done = h_plans[0].done;
            if (!from.resume) part->overlap_used = h_plans[0].overlap_used;
            part->resume = 1;
        }
        if (!g.S) return part && !part->done ? NGSC_ERR_UNSUPPORTED /* not one window fits the buffer */ : NGSC_ERR_FORMAT;
        g.max_title = *(uint32_t *)(hp + o_maxtitle);
        g.max_q = *(uint32_t *)(hp + o_maxq);
        g.max_rec_bytes = *(uint32_t *)(hp + o_maxrec);
    }
    const uint32_t S = g.S, n_rec_total = g.n_rec_total, max_rec = g.max_rec;
    // per-record arrays
    const size_t rec_cap = (size_t)n_rec_total + 2;
    // Arrays whose per-record values depend on the sub-block (title bits, cached fields, candidate values, payload bit
    // counts) have one slot per (sub-block, record): a sub-block takes its second record whatever its size
    // (phyNGSC.cpp:295-300), so a rank whose last window holds a single record before the overlap also encodes the first
    // record of the next rank, and with rank regions smaller than a record consecutive ranks can even start on the same
    // record.  Slots are dealt out sub-block after sub-block (SbState::slot_bias).
    const size_t slot_cap = (size_t)g.slots + 2;
    const size_t grp_cap = (size_t)g.groups + 2;
    Carver cr2;
    size_t o_slen = cr2.take(4 * rec_cap), o_qoff = cr2.take(4 * slot_cap), o_doff = cr2.take(4 * slot_cap);
    size_t o_goff = cr2.take(4 * grp_cap), o_gflags = cr2.take(sizeof(fmask_t) * grp_cap), o_tbits = cr2.take(4 * slot_cap);
    const uint32_t max_title = g.max_title;
    const uint32_t tsep_words = max_title <= 64 ? 1u : 2u;
    size_t o_tsep = cr2.take(8 * tsep_words * rec_cap), o_apos = cr2.take(8 * rec_cap), o_bcnt = cr2.take(8 * rec_cap);
    rc = ensure(c, c->recs2, cr2.off);
    if (rc) return rc;
    uint8_t *rb2 = (uint8_t *)c->recs2.p;
    const uint32_t max_q = g.max_q;
    const uint32_t max_rec_bytes = g.max_rec_bytes;
    const uint32_t qrows = max_q + 1;

    // ---- per-sub-block arena ----
    // Table slots for the non-constant text fields of all sub-blocks.  FASTQ titles have one or two such fields; a
    // call that needs more finds out from the slot counter and runs again with what it asked for.
    const bool wide = c->wide;
    const uint32_t fcap = wide ? WIDE_FIELDS : MAX_FIELDS;   // fields per sub-block in the per-field arrays
    const uint32_t n_slots = std::min<uint32_t>(S * fcap, std::max<uint32_t>({64u, 2 * S, c->min_text_slots}));
    Carver cz;  // zero-initialised every call
    size_t z_fstat = cz.take(sizeof(FieldStat) * (size_t)S * fcap);
    size_t z_occ = cz.take(4 * (size_t)S * 256);
    size_t z_qhist = cz.take(4 * (size_t)S * qrows * 256);
    size_t z_nhist = cz.take(4 * (size_t)S * fcap * MAX_NUM_HUF);
    size_t z_chist = cz.take(4 * (size_t)n_slots * TEXT_TREES * 256);
    size_t z_sb = cz.take(sizeof(SbState) * (size_t)S);   // (zero but for the four words k_init_sb writes)
    if (spec && (c->zero.p != prezero_ptr || cz.off > c->zero.cap || cz.off > prezero)) {
        // (cannot happen: the arena's size follows from the predicted shape, which is the shape of the call that set zero_hint)
        c->shape.valid = 0;
        HIPCHK(c, hipStreamSynchronize(st));
        HIPCHK(c, hipStreamSynchronize(c->side_stream));
        return compress_impl(c, d_buf, buf_len, buf_off, file_size, P, rank_lo, rank_hi, d_out, out_cap, hip_stream, from, part);
    }
    rc = ensure(c, c->zero, cz.off);
    if (rc) return rc;
    Carver cp;
    size_t p_fdesc = cp.take(sizeof(FieldDesc) * (size_t)S * fcap);
    size_t p_grplo = cp.take(4 * (size_t)S);
    size_t p_qcodes = cp.take(4 * (size_t)S * qrows * 256);
    size_t p_qfmt = cp.take(sq_qfmt_bytes(qrows, S));
    size_t p_qblob = cp.take((size_t)S * qrows * BLOB_SLOT_256);
    size_t p_qblen = cp.take(4 * (size_t)S * qrows);
    size_t p_qrowoff = cp.take(4 * (size_t)S * qrows);
    size_t p_dcodes = cp.take(4 * (size_t)S * 256);
    size_t p_dblob = cp.take((size_t)S * BLOB_SLOT_256);
    size_t p_dblen = cp.take(4 * (size_t)S);
    size_t p_ncodes = cp.take(4 * (size_t)S * fcap * MAX_NUM_HUF);
    size_t p_nblob = cp.take((size_t)S * fcap * BLOB_SLOT_512);
    size_t p_nblen = cp.take(4 * (size_t)S * fcap);
    // records of up to 2 KB: one lane per record straight from global memory, 256 records per "tile" (kernels_sq.hip);
    // longer records, or NGSC_SQ_PATH=wave: the wave-per-record kernels of kernels_stats.hip / kernels_pack.hip (R = 0).
    // (The LDS-tiled and chunk-per-lane families of rounds 1-2 were slower on every configuration and are gone;
    // DESIGN.md section 4 keeps their measurements.)
    const char *sq_env = getenv("NGSC_SQ_PATH");
    const bool sq_reg = max_rec_bytes <= 2048 && !(sq_env && !strcmp(sq_env, "wave"));
    const uint32_t R = sq_reg ? 256u : 0u;
    const size_t n_tiles = R ? (size_t)g.tiles : 0;
    size_t p_tilelo = cp.take(4 * (size_t)S);
    size_t p_wq = cp.take(4 * (n_tiles + 8)), p_wd = cp.take(4 * (n_tiles + 8));
    size_t p_ubits = cp.take(4 * (n_tiles * 256 + 8));
    size_t p_tw = cp.take(4 * (size_t)NCF_CAP * slot_cap);
    size_t p_tv = cp.take(4 * (size_t)NUM_CAND_CAP * slot_cap);
    size_t p_rec0 = cp.take(sizeof(Rec0Tab) * (size_t)S);
    rc = ensure(c, c->persb, cp.off);
    if (rc) return rc;
    Carver ct;
    size_t t_codes = ct.take(4 * (size_t)n_slots * TEXT_TREES * 256);
    size_t t_blob = ct.take((size_t)n_slots * TEXT_TREES * BLOB_SLOT_256);
    size_t t_blen = ct.take(4 * (size_t)n_slots * TEXT_TREES);
    size_t t_owner = ct.take(4 * (size_t)n_slots);
    rc = ensure(c, c->slots, ct.off);
    if (rc) return rc;
    if (!spec && (c->zero.p != prezero_ptr || cz.off > prezero)) launch_zero_range(st, c->zero.p, cz.off);
    c->zero_hint = cz.off;

    uint8_t *zb = (uint8_t *)c->zero.p, *pb = (uint8_t *)c->persb.p, *tb = (uint8_t *)c->slots.p;
    Arena A;
    memset(&A, 0, sizeof A);
    A.buf = buf; A.buf_len = len;
    A.te = d_te; A.se = d_se; A.ls = d_ls;
    A.slen = (uint32_t *)(rb2 + o_slen); A.qoff = (uint32_t *)(rb2 + o_qoff); A.doff = (uint32_t *)(rb2 + o_doff);
    A.goff = (uint32_t *)(rb2 + o_goff);
    A.grp_lo = (uint32_t *)(pb + p_grplo);
    A.tbits = (uint32_t *)(rb2 + o_tbits);
    A.tsep = (uint64_t *)(rb2 + o_tsep);
    A.tsep_words = tsep_words;
    A.apos = (AmbPos *)(rb2 + o_apos);
    A.bcnt = (uint64_t *)(rb2 + o_bcnt);
    A.gflags = (fmask_t *)(rb2 + o_gflags);
    A.wq = (uint32_t *)(pb + p_wq);
    A.wd = (uint32_t *)(pb + p_wd);
    A.ubits = (uint32_t *)(pb + p_ubits);
    A.tw = (uint32_t *)(pb + p_tw);
    A.tv = (uint32_t *)(pb + p_tv);
    A.tw_stride = (uint32_t)slot_cap;
    A.rec0 = (Rec0Tab *)(pb + p_rec0);
    A.tile_lo = (uint32_t *)(pb + p_tilelo);
    A.err = d_err;
    A.sb = (SbState *)(zb + z_sb);
    A.fstat = (FieldStat *)(zb + z_fstat);
    A.fdesc = (FieldDesc *)(pb + p_fdesc);
    A.dna_occ = (uint32_t *)(zb + z_occ);
    A.qhist = (uint32_t *)(zb + z_qhist);
    A.nhist = (uint32_t *)(zb + z_nhist);
    A.qcodes = (uint32_t *)(pb + p_qcodes);
    A.qfmt = (uint32_t *)(pb + p_qfmt);
    A.qfmt_stride = (uint32_t)sq_qfmt_words(qrows);
    A.qblob = pb + p_qblob;
    A.qblob_len = (uint32_t *)(pb + p_qblen);
    A.dcodes = (uint32_t *)(pb + p_dcodes);
    A.dblob = pb + p_dblob;
    A.dblob_len = (uint32_t *)(pb + p_dblen);
    A.ncodes = (uint32_t *)(pb + p_ncodes);
    A.nblob = pb + p_nblob;
    A.nblob_len = (uint32_t *)(pb + p_nblen);
    A.slot_counter = d_slotc;
    A.big_trees = (uint32_t *)(sm + o_bigt);
    A.num_huff = (uint32_t *)(sm + o_numhuff);
    A.wide_needed = (uint32_t *)(sm + o_wide);
    A.fcap = fcap;
    A.slot_owner = (uint32_t *)(tb + t_owner);
    A.n_slots = n_slots;
    A.chist = (uint32_t *)(zb + z_chist);
    A.tcodes = (uint32_t *)(tb + t_codes);
    A.tblob = tb + t_blob;
    A.tblob_len = (uint32_t *)(tb + t_blen);
    A.qrows = qrows;
    A.n_sb = S;
    // Chunk-major copies of every record's bases and qualities, filled by the statistics kernel and read by
    // measure and pack - when they stay within 2.5x the job (address space; only real chunks are touched).  NGSC_SQ_DENSE=0:
    // measure and pack read the file themselves, as in round 1
    A.dense_s = A.dense_q = nullptr;
    A.dstride = 0;
    A.dtail_c = 0xffffffffu;
    A.dtail_b = 16;
    {
        const char *de = getenv("NGSC_SQ_DENSE");
        uint64_t stride = ((uint64_t)max_q + 15) & ~15ull;
        uint32_t tail_c = 0xffffffffu, tail_b = 16;
        if (max_q) {
            const uint32_t lc = (max_q - 1) / 16, rem = max_q - 16 * lc;   // last chunk of the longest read and its bytes
            if (rem <= 12) { tail_c = lc; tail_b = (rem + 3) & ~3u; stride = 16ull * lc + tail_b; }
        }
        const uint64_t per_stream = (stride * (uint64_t)n_rec_total + 15) & ~15ull, need = 2 * per_stream;
        if (!(de && !strcmp(de, "0")) && sq_reg && stride && need <= (uint64_t)len * 5 / 2) {
            rc = ensure(c, c->dense, (size_t)need + 64);
            if (rc) return rc;
            A.dense_s = (uint8_t *)c->dense.p;
            A.dense_q = A.dense_s + per_stream;
            A.dstride = n_rec_total;
            A.dtail_c = tail_c;
            A.dtail_b = tail_b;
        }
    }
    // (round 4, measured and not kept: measure and pack dealing each tile's 256 records to their lanes in the order of their lengths,
    // so that a wave's lanes finish together - no change on 36..300-base reads, 1.849 against 1.854 ms: the pack kernel walks its
    // table windows behind workgroup barriers whatever its waves hold - and the extra registers and LDS cost the 36-base job
    // 4 % of its pack kernel; profiles/r04_ab_planner_sorted_lanes.txt)
    A.out = (uint8_t *)d_out;
    uint32_t *d_qrowoff = (uint32_t *)(pb + p_qrowoff);

    // SbState (rec_lo, n_rec, rank, slot_bias; everything else zero), group and tile offsets per sub-block: written on the
    // device from the cut list that is already there (k_init_sb), which also holds the job against the shape the arrays
    // were sized for - the synchronous path's own numbers, or the prediction
    size_t sb_bytes = sizeof(SbState) * (size_t)S;
    rc = ensure_pinned(c, cs.off + sb_bytes + 8 * (size_t)S + cs.off + 4096);
    if (rc) return rc;
    hp = (uint8_t *)c->h_pin;
    SbState *h_sb = (SbState *)hp;
    {
        KScope k(c, st, "k_init_sb");
        ShapeCaps caps;
        caps.n_rec_total = n_rec_total; caps.S = S; caps.max_rec = max_rec; caps.slots = g.slots; caps.groups = g.groups; caps.tiles = g.tiles;
        caps.max_title = max_title; caps.max_q = max_q; caps.max_rec_bytes = max_rec_bytes; caps.rec_bound = (uint32_t)rec_bound;
        launch_init_sb(st, A, d_cuts, d_plans, n_ranks, max_cuts, R, caps, d_total, d_maxtitle, d_maxq, d_maxrec);
    }

    // ---- stage C: statistics, trees, measure.  The title chain and the sequence/quality chain share nothing
    // until the trees are built and can run on two streams. ----
    // Two streams ("staggered"): the title chain and the sequence/quality chain are forked behind the sequence/quality
    // statistics kernel, so that the latency-bound kernels of one chain (sub-block finalisation, the Huffman waves, later
    // headers and length tables) run beside a throughput kernel of the other instead of alone on the GPU; the payload pack
    // kernel, the longest of all, runs alone.  3 % on the 1 GB bench job, 10-17 % on 400 MB jobs, which is the size the file
    // driver works in.  NGSC_OVERLAP=0: everything on the caller's stream - also taken whenever every kernel is being timed
    // (ngsc_set_profiling without ngsc_profile_only), so that per-kernel durations mean what they say.
    // Colour-space sub-blocks (found by the statistics kernels of a first run, which then ends with ERR_COLORSPACE): the job
    // runs on a copy of its input in which they are decoded to bases as the reference decodes them (kernels_cs.hip)
    if (c->cs_retry) {
        rc = ensure(c, c->cscopy, (size_t)len + 64);
        if (rc) return rc;
        HIPCHK(c, hipMemcpyAsync(c->cscopy.p, buf, len, hipMemcpyDeviceToDevice, st));
        KScope k(c, st, "k_cs_decode");
        launch_cs(st, A, max_rec, buf, (uint8_t *)c->cscopy.p, d_se);
        A.buf = (const uint8_t *)c->cscopy.p;
        A.cs_mode = 1;
    }
    const bool stagger = !serial && sq_reg;
    hipStream_t sts = stagger ? c->side_stream : st;   // the sequence/quality chain's stream behind its statistics kernel
    auto fork_s = [&]() -> int {
        if (!stagger) return NGSC_OK;
        HIPCHK(c, hipEventRecord(c->ev_fork, st));
        HIPCHK(c, hipStreamWaitEvent(sts, c->ev_fork, 0));
        return NGSC_OK;
    };
    auto join_s = [&]() -> int {
        if (!stagger) return NGSC_OK;
        HIPCHK(c, hipEventRecord(c->ev_join, sts));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_join, 0));
        return NGSC_OK;
    };
    // titles of at most 128 bytes take the word-parallel kernels, with one or two 64-bit mask words (NGSC_TITLE_PATH=reg forces the
    // byte-serial ones); a job with more than MAX_FIELDS separators in some first title the wide kernels (found out by a first run)
    const char *tp_env = getenv("NGSC_TITLE_PATH");
    const bool t_swar = !wide && max_title <= 128 && !(tp_env && !strcmp(tp_env, "reg"));
    // record 0's field table - one wave per sub-block, a serial walk over a title: 23 us - is needed by the title kernels
    // only: on the side stream, beside the sequence/quality statistics instead of in front of them
    if ((rc = fork_s())) return rc;
    if (!wide) { KScope k(c, sts, "k_rec0_fields"); launch_rec0_fields(sts, A); }
    if (stagger) HIPCHK(c, hipEventRecord(c->ev_zero, sts));
    if (sq_reg) { KScope k(c, st, "k_sq_stats_reg"); launch_sq_stats_reg(st, A, max_rec, max_rec_bytes, n_rec_total); }
    else { KScope k(c, st, "k_seq_qual_stats"); launch_seq_qual_stats(st, A, max_rec); }
    if ((rc = fork_s())) return rc;
    if (stagger) HIPCHK(c, hipStreamWaitEvent(st, c->ev_zero, 0));   // (the title kernels below read the table)
    if (zero_late && out_prefix && stagger) launch_zero_range(sts, d_out, out_prefix);
    else if (zero_late && out_prefix) launch_zero_range(st, d_out, out_prefix);
    const bool big_lds = !spec || c->big_alphabets, hist_lds = !spec || (c->title_need & 3u) != 0, ttree_lds = hist_lds;
    { KScope k(c, sts, "k_sq_finalize"); launch_sq_finalize(sts, A, !big_lds); }
    // (round 4) A predicted step does NOT launch the kernels that usually have nothing to do - the 512-symbol tree instance, the
    // title's histograms and trees - when the context's last job gave them nothing to do: the finalisation kernels in front of them
    // raise ERR_SHAPE if this job does (the step runs again, the synchronous way, with everything).  Launched, thousands of
    // workgroups queued for 26-40 KB of LDS each on CUs that the title statistics kernel fills, to find out that they could
    // leave: 14-150 us on the side chain, 20-100 + 25 on the title chain of a 36-base step.
    { KScope k(c, sts, "k_huffman_sb"); launch_huffman_sq(sts, A, c->big_alphabets, big_lds); }
    // (round 4, measured and not kept: the payload measure kernel - the side chain's one throughput kernel - on a stream of ordinary
    // priority, so that the title chain's small kernels are not dispatched behind it: no change on 36-base reads, 3 % slower on
    // 150-base reads, profiles/r04_ab_measure_stream_priority.txt; the side stream at low priority altogether: 1-2 % slower)
    if (sq_reg) {
        { KScope k(c, sts, "k_sq_format_qtab"); launch_sq_format_qtab(sts, A); }
        { KScope k(c, sts, "k_sq_measure_reg"); launch_sq_measure_reg(sts, A, max_rec); }
    }
    else { KScope k(c, sts, "k_measure_sq"); launch_measure_sq(sts, A, max_rec); }
    // (round 4, measured and not kept: these small kernels of the title chain on a high-priority stream of their own, so that their
    // workgroups are not dispatched behind the payload measure kernel's - the two extra stream hops cost what the earlier dispatch
    // gains: 1.221 / 1.224 ms on 150-base reads, 1.937 / 1.935 on 36..300-base reads, 0.8 % slower at 36 bases; profiles/r04_ab_*)
    if (wide) { KScope k(c, st, "k_title_wide_stats"); launch_title_wide_stats(st, A, max_rec); }
    else {
        if (t_swar) { KScope k(c, st, "k_title_stats_swar"); launch_title_stats_swar(st, A, max_rec, max_title); }
        else { KScope k(c, st, "k_title_stats_reg"); launch_title_stats_reg(st, A, max_rec); }
        { KScope k(c, st, "k_field_finalize"); launch_field_finalize(st, A, !hist_lds); }
        if (hist_lds) { KScope k(c, st, "k_title_hist"); launch_title_hist(st, A, max_rec, t_swar); }
    }
    if (ttree_lds) { KScope k(c, st, "k_huffman_title"); launch_huffman_title(st, A); }
    if (wide) { KScope k(c, st, "k_title_wide_measure"); launch_title_wide_measure(st, A, max_rec); }
    else { KScope k(c, st, t_swar ? "k_title_measure_swar" : "k_title_measure_reg"); launch_title_measure_reg(st, A, max_rec, t_swar); }
    if ((rc = join_s())) return rc;
    // A predicted step whose output was cleared early as far as the previous job's reached has (almost always) nothing left to clear
    // once its size is known: the layout kernel sees to that itself - up to 1 MiB beyond the early clearing is cleared by its last
    // wave, more raises ERR_SHAPE - and the step saves a launch and its gap on the critical path (12 us)
    const bool zero_by_layout = spec && R && out_prefix;
    if (R) { KScope k(c, st, "k_layout"); launch_layout(st, A, R, d_qrowoff, out_cap, d_outtot, d_done, zero_by_layout ? d_out : nullptr, zero_by_layout ? out_prefix : 0); }
    else {
        { KScope k(c, st, "k_scan_sb"); launch_scan_sb(st, A); }
        { KScope k(c, st, "k_sizes_offsets"); launch_sizes_offsets(st, A, d_qrowoff, out_cap, d_outtot); }
    }
    // ---- sizes and layout are on the device now; their copies for the host travel behind the pack kernels (no sync here) ----
    // (staggered: the copies, the headers and the length tables beside the output clearing and the title pack kernel)
    if ((rc = fork_s())) return rc;
    uint8_t *const h_small = hp + sb_bytes;   // the small state behind the layout kernels (plans and cuts included)
    HIPCHK(c, hipMemcpyAsync(hp, A.sb, sb_bytes, hipMemcpyDeviceToHost, sts));
    HIPCHK(c, hipMemcpyAsync(h_small, sm, cs.off, hipMemcpyDeviceToHost, sts));
    // ---- pack ----
    if (!zero_by_layout) { KScope k(c, st, "k_zero_output"); launch_zero_output(st, d_out, d_outtot, out_cap, out_prefix); }
    if (stagger && !zero_by_layout) {   // the side stream's kernels write into the cleared output (it follows the layout kernel already)
        HIPCHK(c, hipEventRecord(c->ev_fork, st));
        HIPCHK(c, hipStreamWaitEvent(sts, c->ev_fork, 0));
    }
    { KScope k(c, sts, "k_write_headers"); launch_write_headers(sts, A, d_qrowoff); }
    { KScope k(c, sts, "k_pack_lentable"); launch_pack_lentable(sts, A, max_rec); }
    // (measured and not kept: the title pack kernel - latency-bound, 0.11 ms alone - on the side stream BESIDE the payload pack
    // kernel.  The payload kernel then takes 0.55 ms instead of 0.40: it is bound by instruction issue and its LDS queue, and
    // what it shares it loses - 452 against 462 GB/s)
    if (wide) { KScope k(c, st, "k_title_wide_pack"); launch_title_wide_pack(st, A, max_rec); }
    else { KScope k(c, st, "k_title_pack_reg"); launch_title_pack_reg(st, A, max_rec); }
    // (round 4) the side stream's headers, length tables and result copies are joined BEHIND the payload pack kernel, which reads
    // nothing of theirs: with long or unequal reads (few records: a short title pack kernel, but as many header bytes) they
    // used to hold it back - 40 us of a 36..300-base step
    if (sq_reg) {
        // LDS staging of a tile's output: room for 5 quality bits and 2.5 DNA bits per base of R reads of the job's average
        // length, + 10 % (from the record count: no stream size has to come back to the host first; a tile that needs
        // more writes straight to the zeroed output)
        // (average read length from the record count: a record is a title, two lines of the read's length and four more bytes)
        // (+ 10 %, not 25: with long reads the staging is most of the kernel's LDS, and 46 KB against 42 is three workgroups per CU
        // against four - pack 0.335 -> 0.309 ms per 400 MB of 150-base reads; a tile that overflows is still written, directly)
        const uint64_t trec = (uint64_t)R * 11 / 10;
        // ... and, once a call has been through, from the bits per base it saw: jobs of one file look alike)
        const uint64_t avg_l = std::min<uint64_t>(max_q, (uint64_t)len / n_rec_total / 2 + 1);
        const uint64_t qx = c->qbits_x16 ? c->qbits_x16 + 4 : 80, dx = c->dbits_x16 ? c->dbits_x16 + 2 : 40;
        const uint32_t sw_q = (uint32_t)(trec * avg_l * qx / 16 / 32) + 64;
        const uint32_t sw_d = (uint32_t)(trec * avg_l * dx / 16 / 32) + 64;
        { KScope k(c, st, "k_sq_pack_reg"); launch_sq_pack_reg(st, A, max_rec, sw_q, sw_d); }
    }
    else { KScope k(c, st, "k_pack_sq"); launch_pack_sq(st, A, max_rec); }
    if ((rc = join_s())) return rc;
    // (measured and not kept, profiles/r04_ab_fewer_events.txt: (1) no join here - either stream copies the error word behind its own
    // last kernel, the host waits for both and takes the lower: no gain, 1.920 -> 1.920/1.932 ms per GB, 125 MB jobs 0.451 -> 0.455;
    // (2) the zero arena - the context's own - cleared on the side stream from the step's first microsecond, without the event that
    // holds it behind the index kernels: the planner starts 7 us earlier, the clearing slows the newline count beside it down by more)
    // ---- the one sync of a predicted step (the second of a synchronous one): error word, sizes, result tables ----
    uint8_t *hfin = h_small + cs.off + 64;
    HIPCHK(c, hipMemcpyAsync(hfin, d_err, 4, hipMemcpyDeviceToHost, st));
    HT(t_enq);
    {   // the step's one wait: polled for its first millisecond-scale stretch (a blocking wait wakes up 10-20 us after the copy
        // has landed - one percent of a step), then handed to the runtime
        const auto t_poll = std::chrono::steady_clock::now();
        hipError_t q;
        while ((q = hipStreamQuery(st)) == hipErrorNotReady)
            if (std::chrono::steady_clock::now() - t_poll > std::chrono::milliseconds(20)) break;
        if (q != hipSuccess) HIPCHK(c, hipStreamSynchronize(st));
    }
    // (host time of a call, -DNGSC_HOST_TIMING: 4 us to the first launch, 105-145 us to enqueue the step, 1-3 us from here to the
    // return - 6-15 while the cut list below was a vector of the call's own - and 10-20 us in the Python caller: ~25 of the ~45 us
    // the GPU idles between two steps of the bench; the rest is the copy's completion reaching the host and the first launch
    // reaching the GPU.  Reading the result tables once while the pack kernel still runs - they are in host memory by then, not
    // in the host's caches - takes 1 us off the loops below and was not kept; the host thread on the GPU's NUMA node or the
    // other: 105 against 111-118 us to enqueue a step, step times within 0.3 %; profiles/r04_host_time_of_a_call.txt)
    HT(t_waited);
    HIPCHK(c, hipGetLastError());
    const int32_t err3 = *(int32_t *)(h_small + o_err);   // state behind the layout kernels
    const int32_t err4 = *(int32_t *)hfin;
    if (err3 == ERR_SHAPE || err4 == ERR_SHAPE) {
        // the prediction did not hold (nothing was written behind k_init_sb): once more, reading the shape back
        c->shape.valid = 0;
        if (!spec) return NGSC_ERR_HIP;   // (the synchronous path sizes everything from the job's own numbers)
#ifdef NGSC_HOST_TIMING
        fprintf(stderr, "[host] the prediction did not hold: once more\n");
#endif
        HIPCHK(c, hipStreamSynchronize(c->side_stream));
        return compress_impl(c, d_buf, buf_len, buf_off, file_size, P, rank_lo, rank_hi, d_out, out_cap, hip_stream, from, part);
    }
    const uint64_t total = *(uint64_t *)(h_small + o_outtot);
    c->res_bytes = total;
    c->out_hint = total;
    if (err3 == NGSC_ERR_UNSUPPORTED && !c->slot_retry) {
        const uint32_t asked = *(uint32_t *)(h_small + o_slotc);
        if (asked > n_slots) {
            // (kernels that start after the error flag is up return early, so the count can fall short: grow at least 2x)
            c->slot_retry = true;
            int r = NGSC_ERR_UNSUPPORTED;
            for (uint32_t want = std::max(asked, 2 * n_slots);; want *= 2) {
                c->min_text_slots = want;
                r = compress_impl(c, d_buf, buf_len, buf_off, file_size, P, rank_lo, rank_hi, d_out, out_cap, hip_stream, from, part);
                if (r != NGSC_ERR_UNSUPPORTED || want >= S * fcap) break;
            }
            c->slot_retry = false;
            return r;
        }
    }
    if (err3 == NGSC_ERR_UNSUPPORTED && *(uint32_t *)(h_small + o_wide) && !wide) {
        // a first title with more than MAX_FIELDS separators: once more, on the wide title kernels (and so the context's next jobs)
        c->wide = true;
        HIPCHK(c, hipStreamSynchronize(c->side_stream));
        return compress_impl(c, d_buf, buf_len, buf_off, file_size, P, rank_lo, rank_hi, d_out, out_cap, hip_stream, from, part);
    }
    if ((err3 == NGSC_ERR_COLORSPACE || err4 == NGSC_ERR_COLORSPACE) && !c->cs_retry) {
        c->cs_retry = true;
        const int r = compress_impl(c, d_buf, buf_len, buf_off, file_size, P, rank_lo, rank_hi, d_out, out_cap, hip_stream, from, part);
        c->cs_retry = false;
        return r;
    }
    if (err3) return err3;
    if (err4) return err4;
    // ---- results (the planner's tables came back with the small state) ----
    const RankPlan *h_plans = (const RankPlan *)(h_small + o_plans);
    const Cut *h_cuts = (const Cut *)(h_small + o_cuts);
    // (the context's vector, not a local one: the first allocation behind the wait was 6-15 us of host time; -DNGSC_HOST_TIMING found
    // it.  The caller's first allocation pays about as much instead - it is whoever allocates first behind the runtime's clean-up)
    std::vector<Cut> &cuts = c->cuts_tmp;
    cuts.clear();
    cuts.reserve(S);
    c->res_rank.resize(n_ranks);
    JobShape seen = g;   // what this job really looked like: the next call's prediction
    seen.S = 0; seen.max_rec = 0; seen.slots = seen.groups = seen.tiles = 0;
    for (uint32_t i = 0; i < n_ranks; ++i) {
        ngsc_rank_t &rk = c->res_rank[i];
        rk.rank = rank_lo + i;
        rk.n_subblocks = h_plans[i].n_cuts;
        rk.first_subblock = (uint32_t)cuts.size();
        rk.overlap_used = h_plans[i].overlap_used;
        rk.stream_off = rk.stream_bytes = 0;
        for (uint32_t k = 0; k < h_plans[i].n_cuts; ++k) {
            const Cut &ct = h_cuts[(size_t)i * max_cuts + k];
            cuts.push_back(ct);
            ++seen.S;
            seen.max_rec = std::max(seen.max_rec, ct.n_rec);
            seen.slots += ct.n_rec;
            seen.groups += (ct.n_rec + GROUP - 1) / GROUP;
            seen.tiles += (ct.n_rec + 255) / 256;
        }
    }
    if (cuts.size() != S) return NGSC_ERR_HIP;   // (k_init_sb lets no other count through)
    seen.n_rec_total = *(uint32_t *)(h_small + o_total) / 4;
    seen.max_title = *(uint32_t *)(h_small + o_maxtitle);
    seen.max_q = *(uint32_t *)(h_small + o_maxq);
    seen.max_rec_bytes = *(uint32_t *)(h_small + o_maxrec);
    if (!part && !c->cs_retry) {
        // Capacities with a little head room for the counts (the next job of a file has a few records more or less); the
        // maxima as they were: a larger title / read / record in the next job is a miss, and so is another sub-block count
        seen.n_rec_total += seen.n_rec_total / 64 + 256;
        seen.max_rec = std::min<uint32_t>(MAX_RECORDS, seen.max_rec + seen.max_rec / 64 + 64);
        seen.slots += seen.slots / 64 + 256;
        seen.groups += seen.groups / 64 + 64;
        seen.tiles += seen.tiles / 64 + 64;
        seen.valid = 1;
        c->shape = seen;
    }
    c->res_sb.resize(S);
    if (wide) {   // (the context goes back to the ordinary title kernels once a job has no such title any more)
        uint32_t most = 0;
        for (uint32_t s = 0; s < S; ++s) most = std::max(most, h_sb[s].nf);
        if (most <= MAX_FIELDS) c->wide = false;
    }
    c->qbits_x16 = c->dbits_x16 = 0;
    c->big_alphabets = false;
    c->title_need = *(uint32_t *)(h_small + o_numhuff);
    for (uint32_t s = 0; s < S; ++s) {
        const SbState &x = h_sb[s];
        if (x.nqual > 64 || x.nsym > 64) c->big_alphabets = true;
        if (x.n_bases) {
            c->qbits_x16 = std::max<uint32_t>(c->qbits_x16, (uint32_t)(((uint64_t)x.qbits * 16 + x.n_bases - 1) / x.n_bases));
            c->dbits_x16 = std::max<uint32_t>(c->dbits_x16, (uint32_t)(((uint64_t)x.dbits * 16 + x.n_bases - 1) / x.n_bases));
        }
        ngsc_subblock_t &r = c->res_sb[s];
        r.rank = x.rank; r.n_records = x.n_rec; r.win_pos = cuts[s].win_pos; r.wsize = cuts[s].wsize;
        r.out_off = x.out_off;
        r.info_len = x.info_len; r.title_len = x.title_len; r.qual_len = x.qual_len; r.dna_len = x.dna_len;
        r.max_qlen = x.max_q; r.max_slen = x.max_s; r.nsym = x.nsym; r.nqual = x.nqual; r.flags = x.flags;
        r.n_fields = x.nf;
        r.n_bases = x.n_bases;
    }
    for (auto &rk : c->res_rank) {
        if (!rk.n_subblocks) continue;
        rk.stream_off = c->res_sb[rk.first_subblock].out_off;
        uint64_t b = 0;
        for (uint32_t k = 0; k < rk.n_subblocks; ++k) {
            const ngsc_subblock_t &r = c->res_sb[rk.first_subblock + k];
            b += (uint64_t)r.info_len + r.title_len + r.qual_len + r.dna_len;
        }
        rk.stream_bytes = b;
    }
    return NGSC_OK;
}

extern "C" {

int ngsc_compress_device(ngsc_ctx *c, const void *d_buf, uint64_t buf_len, uint64_t buf_off, uint64_t file_size,
                         uint32_t P, uint32_t rank_lo, uint32_t rank_hi, void *d_out, uint64_t out_cap, void *hip_stream)
{
    HT(t_entry);
    PlanResume from;
    memset(&from, 0, sizeof from);
    const int rc = compress_impl(c, d_buf, buf_len, buf_off, file_size, P, rank_lo, rank_hi, d_out, out_cap, hip_stream, from, nullptr);
    if (rc != NGSC_OK && c && c->side_stream) (void)hipStreamSynchronize(c->side_stream);   // (the early clearing of d_out may still run)
#ifdef NGSC_HOST_TIMING
    g_ht.done();
#endif
    return rc;
}

int ngsc_compress_rank_part(ngsc_ctx *c, const void *d_buf, uint64_t buf_len, uint64_t buf_off, uint64_t file_size, uint32_t P,
                            uint32_t rank, ngsc_rank_part_t *state, void *d_out, uint64_t out_cap, void *hip_stream)
{
    if (!state || state->done) return NGSC_ERR_ARG;
    PlanResume from;
    memset(&from, 0, sizeof from);
    if (state->resume) {
        from.resume = 1; from.next_pos = state->next_pos; from.bytes_read = state->bytes_read; from.bsize = state->bsize;
        from.overlap = state->overlap;
    }
    const int rc = compress_impl(c, d_buf, buf_len, buf_off, file_size, P, rank, rank + 1, d_out, out_cap, hip_stream, from, state);
    if (rc != NGSC_OK && c && c->side_stream) (void)hipStreamSynchronize(c->side_stream);
    return rc;
}

}  // extern "C"
