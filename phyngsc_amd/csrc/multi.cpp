// multi.cpp - the single-process host over several GPUs and its one collective.
//
// The reference fans out with `mpiexec -np P`; its ranks meet twice, at the end: MPI_Gather of {n_blocks, n_subblocks,
// last_block_size, wr_overlap} (phyNGSC.cpp:934-964) and MPI_Gatherv of the block timestamps (:994), from which rank 0
// derives the block order and writes the footer.  Here ONE process owns the node's GPUs: a "team" is one host thread
// per context (= per GPU) plus an RCCL communicator created inside this process (ncclCommInitAll over the contexts'
// devices - ranks talk over xGMI, no launcher, no TCP store).  A step (ngsc_team_compress) lets every thread compress
// its shard of consecutive virtual ranks on its GPU - no data-path collective - and then all-gathers the per-rank rows
//   { n_blocks, n_subblocks, overlap_used, block sizes ... }           (a few hundred bytes per GPU)
// so that every GPU / thread holds all P rows and derives the same file offsets by a local prefix sum over the
// deterministic block order (ngsc.h).  RCCL is bound at run time (dlopen of librccl.so.1: the one a host program such
// as torch has already loaded, else ROCm's): the one-GPU paths of the library do not depend on it, and when it is
// missing or refuses the device list (two contexts on one device) the team says so (ngsc_team_rccl_ranks() == 0,
// ngsc_team_describe()) and the rows meet in host memory instead - same rows, same offsets.
#include "../../include/ngsc.h"
#include "ngsc_common.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace ngsc;

namespace {

// ---- RCCL, bound at run time ---------------------------------------------------------------------------
struct RcclApi {
    void *handle = nullptr;
    std::string path, why;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;   // (optional)
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return handle != nullptr; }
};

RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        if (const char *e = getenv("NGSC_RCCL"))
            if (!strcmp(e, "0")) { api.why = "NGSC_RCCL=0"; return; }
        const char *names[] = {getenv("NGSC_RCCL_LIB"), "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            if (void *h = dlopen(n, RTLD_NOW | RTLD_GLOBAL)) { api.handle = h; api.path = n; break; }
            api.why = dlerror();
        }
        if (!api.handle) return;
        bool all = true;
        auto sym = [&](const char *name) { void *p = dlsym(api.handle, name); if (!p) { all = false; api.why = std::string("missing symbol ") + name; } return p; };
        api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
        api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!all) { api.handle = nullptr; }
        else api.CommAbort = (decltype(api.CommAbort))dlsym(api.handle, "ncclCommAbort");
    });
    return api;
}

struct Member {
    ngsc_ctx *ctx = nullptr;
    int device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    uint64_t *d_send = nullptr, *d_recv = nullptr;   // device
    uint64_t *h_send = nullptr, *h_recv = nullptr;   // pinned
    size_t cap_words = 0;                            // capacity of d_send (d_recv: n x)
    std::vector<uint64_t> rows;                      // this member's view of all rows after a step
    int rc = NGSC_OK;
    double ms = 0, xchg_ms = 0;   // host time of the member's last step / of the exchange inside it
};

}  // namespace

struct ngsc_team {
    std::vector<Member> m;
    uint32_t rccl_ranks = 0;       // what the communicator reports (ncclCommCount); 0: rows meet in host memory
    std::string describe;
    // host threads, one per member, parked between steps
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t epoch = 0;
    uint32_t pending = 0;
    bool quit = false;
    std::function<void(uint32_t)> job;
    // host-memory meeting point of the rows (the fallback, and the cross-check of the collective)
    std::vector<uint64_t> host_rows;
    std::mutex hm;
    std::condition_variable hcv;
    uint32_t arrived = 0;
    uint64_t gen = 0;
    int vote = NGSC_OK, agreed = NGSC_OK;   // what the members have reported in the barrier being filled / in the last one
    int aborted = NGSC_OK;                  // a member could not go on (ngsc_team_abort): nobody waits in a barrier any more
    std::atomic<int> coll_failed{0};        // a member's part of the running collective failed: the others stop waiting for it
};

namespace {

void team_thread(ngsc_team *t, uint32_t i)
{
    (void)hipSetDevice(t->m[i].device);
    uint64_t seen = 0;
    for (;;) {
        {
            std::unique_lock<std::mutex> lk(t->mu);
            t->cv_go.wait(lk, [&] { return t->quit || t->epoch != seen; });
            if (t->quit) return;
            seen = t->epoch;
        }
        t->job(i);
        {
            std::lock_guard<std::mutex> lk(t->mu);
            --t->pending;
        }
        t->cv_done.notify_all();
    }
}

void team_run(ngsc_team *t, std::function<void(uint32_t)> fn)
{
    {
        std::lock_guard<std::mutex> lk(t->mu);
        t->job = std::move(fn);
        t->pending = (uint32_t)t->m.size();
        ++t->epoch;
    }
    t->cv_go.notify_all();
    std::unique_lock<std::mutex> lk(t->mu);
    t->cv_done.wait(lk, [&] { return t->pending == 0; });
}

// Host-memory barrier between the team's members that is also a vote: every member hands in its status (0 or an error code),
// all of them leave with the first error handed in - or with the code of ngsc_team_abort(), which also ends the wait of
// whoever is already inside.  A collective is entered only behind such a vote: no member can be left waiting in RCCL for a
// peer that has given up.
int team_vote(ngsc_team *t, int status)
{
    std::unique_lock<std::mutex> lk(t->hm);
    if (t->aborted) return t->aborted;
    const uint64_t g = t->gen;
    if (status && !t->vote) t->vote = status;
    if (++t->arrived == t->m.size()) {
        t->arrived = 0;
        t->agreed = t->vote;
        t->vote = NGSC_OK;
        ++t->gen;
        const int a = t->agreed;
        lk.unlock();
        t->hcv.notify_all();
        return a;
    }
    t->hcv.wait(lk, [&] { return t->gen != g || t->aborted; });
    return t->gen != g ? t->agreed : t->aborted;
}

int member_buffers(Member &mb, size_t words, uint32_t n)
{
    if (words <= mb.cap_words) return NGSC_OK;
    if (mb.d_send) { (void)hipFree(mb.d_send); (void)hipFree(mb.d_recv); (void)hipHostFree(mb.h_send); (void)hipHostFree(mb.h_recv); }
    mb.d_send = mb.d_recv = mb.h_send = mb.h_recv = nullptr;
    mb.cap_words = 0;
    const size_t w = words + 64;
    if (hipMalloc((void **)&mb.d_send, 8 * w) != hipSuccess || hipMalloc((void **)&mb.d_recv, 8 * w * n) != hipSuccess ||
        hipHostMalloc((void **)&mb.h_send, 8 * w, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&mb.h_recv, 8 * w * n, hipHostMallocDefault) != hipSuccess)
        return NGSC_ERR_NOMEM;
    mb.cap_words = w;
    return NGSC_OK;
}

// The collective, from member i's own thread: send[count] -> recv[n * count] (member order).  RCCL when the team has a
// communicator, host memory otherwise.  status: what the member has to report about its own step (0: fine).  Nobody enters
// the collective unless every member is fine and has its buffers and its row on its GPU (a vote in host memory first);
// inside, a member waits for its stream by polling, so that a failure on another member or ngsc_team_abort() ends the
// wait (the communicator is aborted where the library allows it) instead of leaving it in hipStreamSynchronize for ever.
int member_allgather(ngsc_team *t, uint32_t i, int status, const uint64_t *send, uint32_t count, uint64_t *recv)
{
    const uint32_t n = (uint32_t)t->m.size();
    Member &mb = t->m[i];
    const bool use_rccl = t->rccl_ranks == n && n > 0;
    if (const char *e = getenv("NGSC_TEST_FAIL_MEMBER"))   // (test hook: this member fails in front of the collective)
        if (!status && atoi(e) == (int)i) status = NGSC_ERR_NOMEM;
    if (use_rccl && !status) {
        if (hipSetDevice(mb.device) != hipSuccess) status = NGSC_ERR_HIP;
        else if ((status = member_buffers(mb, count, n)) == NGSC_OK) {
            memcpy(mb.h_send, send, 8 * (size_t)count);
            if (hipMemcpyAsync(mb.d_send, mb.h_send, 8 * (size_t)count, hipMemcpyHostToDevice, mb.stream) != hipSuccess ||
                hipStreamSynchronize(mb.stream) != hipSuccess)
                status = NGSC_ERR_HIP;
        }
    }
    if (!use_rccl && !status) {
        std::lock_guard<std::mutex> lk(t->hm);
        if (t->host_rows.size() < (size_t)count * n) t->host_rows.resize((size_t)count * n);
    }
    int rc = team_vote(t, status);   // (also: the host rows are sized before anyone writes)
    if (rc) return rc;
    if (use_rccl) {
        RcclApi &R = rccl();
        t->coll_failed.store(0);
        rc = team_vote(t, NGSC_OK);  // (every member has reset the flag before anyone can raise it)
        if (rc) return rc;
        int mine = NGSC_OK;
        const ncclResult_t nr = R.AllGather(mb.d_send, mb.d_recv, count, ncclUint64, mb.comm, mb.stream);
        if (nr != ncclSuccess) { fprintf(stderr, "[ngsc] ncclAllGather: %s\n", R.GetErrorString(nr)); mine = NGSC_ERR_HIP; }
        else if (hipMemcpyAsync(mb.h_recv, mb.d_recv, 8 * (size_t)count * n, hipMemcpyDeviceToHost, mb.stream) != hipSuccess) mine = NGSC_ERR_HIP;
        if (mine) t->coll_failed.store(1);
        const auto t0 = std::chrono::steady_clock::now();
        while (!mine) {
            const hipError_t q = hipStreamQuery(mb.stream);
            if (q == hipSuccess) break;
            if (q != hipErrorNotReady) { mine = NGSC_ERR_HIP; t->coll_failed.store(1); break; }
            bool gone;
            { std::lock_guard<std::mutex> lk(t->hm); gone = t->aborted != 0; }
            if (gone || t->coll_failed.load() || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                mine = NGSC_ERR_HIP;
                t->coll_failed.store(1);
                if (R.CommAbort && mb.comm) { (void)R.CommAbort(mb.comm); mb.comm = nullptr; t->rccl_ranks = 0; }
                break;
            }
            std::this_thread::yield();
        }
        if (!mine) memcpy(recv, mb.h_recv, 8 * (size_t)count * n);
        return team_vote(t, mine);   // all leave with the same verdict
    }
    memcpy(t->host_rows.data() + (size_t)i * count, send, 8 * (size_t)count);
    if ((rc = team_vote(t, NGSC_OK))) return rc;
    memcpy(recv, t->host_rows.data(), 8 * (size_t)count * n);
    return team_vote(t, NGSC_OK);   // (read before the next step overwrites)
}

// (the calling thread's device is left as it was found)
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
};

}  // namespace

extern "C" {

int ngsc_team_create(ngsc_ctx **ctxs, uint32_t n_ctx, ngsc_team **out)
{
    if (!ctxs || !n_ctx || !out) return NGSC_ERR_ARG;
    for (uint32_t i = 0; i < n_ctx; ++i)
        if (!ctxs[i]) return NGSC_ERR_ARG;
    DeviceGuard keep;
    ngsc_team *t = new ngsc_team();
    t->m.resize(n_ctx);
    std::vector<int> devs(n_ctx);
    bool distinct = true;
    for (uint32_t i = 0; i < n_ctx; ++i) {
        t->m[i].ctx = ctxs[i];
        t->m[i].device = devs[i] = ngsc_ctx_device(ctxs[i]);
        for (uint32_t k = 0; k < i; ++k) distinct = distinct && devs[k] != devs[i];
    }
    RcclApi &R = rccl();
    std::string why;
    if (!R.ok()) why = "librccl not loaded (" + R.why + ")";
    else if (!distinct) why = "two contexts share a device: RCCL wants one rank per GPU";
    else {
        std::vector<ncclComm_t> comms(n_ctx, nullptr);
        const ncclResult_t nr = R.CommInitAll(comms.data(), (int)n_ctx, devs.data());
        if (nr != ncclSuccess) why = std::string("ncclCommInitAll: ") + R.GetErrorString(nr);
        else {
            int cnt = 0, ver = 0;
            (void)R.CommCount(comms[0], &cnt);
            (void)R.GetVersion(&ver);
            bool ok = cnt == (int)n_ctx;
            for (uint32_t i = 0; i < n_ctx && ok; ++i) {
                t->m[i].comm = comms[i];
                ok = hipSetDevice(devs[i]) == hipSuccess && hipStreamCreateWithFlags(&t->m[i].stream, hipStreamNonBlocking) == hipSuccess;
            }
            if (ok) {
                t->rccl_ranks = (uint32_t)cnt;
                char b[256];
                snprintf(b, sizeof b, "rccl %d (%s), ncclCommInitAll in-process, %d ranks, ncclAllGather of u64 rows", ver, R.path.c_str(), cnt);
                t->describe = b;
            } else {
                why = "communicator size / stream creation failed";
                for (auto c : comms) if (c) (void)R.CommDestroy(c);
                for (auto &mb : t->m) { mb.comm = nullptr; if (mb.stream) { (void)hipStreamDestroy(mb.stream); mb.stream = nullptr; } }
            }
        }
    }
    if (!t->rccl_ranks) t->describe = "host memory (no RCCL: " + why + ")";
    for (uint32_t i = 0; i < n_ctx; ++i) t->threads.emplace_back(team_thread, t, i);
    *out = t;
    return NGSC_OK;
}

void ngsc_team_destroy(ngsc_team *t)
{
    if (!t) return;
    {
        std::lock_guard<std::mutex> lk(t->mu);
        t->quit = true;
    }
    t->cv_go.notify_all();
    for (auto &th : t->threads) th.join();
    DeviceGuard keep;
    RcclApi &R = rccl();
    for (auto &mb : t->m) {
        (void)hipSetDevice(mb.device);
        if (mb.comm && R.ok()) (void)R.CommDestroy(mb.comm);
        if (mb.stream) (void)hipStreamDestroy(mb.stream);
        if (mb.d_send) { (void)hipFree(mb.d_send); (void)hipFree(mb.d_recv); (void)hipHostFree(mb.h_send); (void)hipHostFree(mb.h_recv); }
    }
    delete t;
}

uint32_t ngsc_team_size(ngsc_team *t) { return t ? (uint32_t)t->m.size() : 0; }
uint32_t ngsc_team_rccl_ranks(ngsc_team *t) { return t ? t->rccl_ranks : 0; }
const char *ngsc_team_describe(ngsc_team *t) { return t ? t->describe.c_str() : ""; }

int ngsc_team_member_allgather(ngsc_team *t, uint32_t member, int status, const uint64_t *send, uint32_t count, uint64_t *recv)
{
    if (!t || member >= t->m.size() || !send || !recv || !count) return NGSC_ERR_ARG;
    return member_allgather(t, member, status, send, count, recv);
}

void ngsc_team_exchange_ms(ngsc_team *t, double *ms)
{
    if (!t || !ms) return;
    for (size_t i = 0; i < t->m.size(); ++i) ms[i] = t->m[i].xchg_ms;
}

void ngsc_team_abort(ngsc_team *t, int code)
{
    if (!t) return;
    {
        std::lock_guard<std::mutex> lk(t->hm);
        if (!t->aborted) t->aborted = code ? code : NGSC_ERR_HIP;
    }
    t->coll_failed.store(1);
    t->hcv.notify_all();
}

int ngsc_team_allgather_u64(ngsc_team *t, const uint64_t *send, uint32_t count, uint64_t *recv)
{
    if (!t || !send || !recv || !count) return NGSC_ERR_ARG;
    const uint32_t n = (uint32_t)t->m.size();
    std::vector<std::vector<uint64_t>> got(n, std::vector<uint64_t>((size_t)count * n));
    team_run(t, [&](uint32_t i) { t->m[i].rc = member_allgather(t, i, NGSC_OK, send + (size_t)i * count, count, got[i].data()); });
    for (uint32_t i = 0; i < n; ++i) {
        if (t->m[i].rc) return t->m[i].rc;
        memcpy(recv + (size_t)i * count * n, got[i].data(), 8 * (size_t)count * n);
    }
    return NGSC_OK;
}

int ngsc_team_compress(ngsc_team *t, const ngsc_shard_t *shards, uint64_t file_size, uint32_t P, uint64_t *rows, uint32_t row_len,
                       double *shard_ms)
{
    if (!t || !shards || !rows || row_len < 4 || P < 2) return NGSC_ERR_ARG;
    const uint32_t n = (uint32_t)t->m.size();
    if (P % n) return NGSC_ERR_ARG;
    const uint32_t per = P / n;
    for (uint32_t i = 0; i < n; ++i)
        if (shards[i].rank_lo != i * per || shards[i].rank_hi != (i + 1) * per) return NGSC_ERR_ARG;   // member i owns ranks [i P/n, (i+1) P/n)
    const uint32_t count = per * row_len;
    team_run(t, [&](uint32_t i) {
        Member &mb = t->m[i];
        const ngsc_shard_t &sh = shards[i];
        const auto t0 = std::chrono::steady_clock::now();
        mb.rc = ngsc_compress_device(mb.ctx, sh.d_buf, sh.buf_len, sh.buf_off, file_size, P, sh.rank_lo, sh.rank_hi, sh.d_out, sh.out_cap, nullptr);
        // this member's rows: the 8 MiB block plan of each of its ranks (phyNGSC.cpp:842-928)
        std::vector<uint64_t> mine((size_t)count, 0);
        if (!mb.rc) {
            const ngsc_subblock_t *sb = ngsc_result_subblocks(mb.ctx);
            const ngsc_rank_t *rk = ngsc_result_ranks(mb.ctx);
            const uint32_t nr = ngsc_result_num_ranks(mb.ctx);
            std::vector<uint32_t> sizes;
            std::vector<ngsc_block_t> blocks;
            for (uint32_t r = 0; r < nr && r < per && !mb.rc; ++r) {
                sizes.resize(rk[r].n_subblocks);
                uint64_t bytes = 0;
                for (uint32_t k = 0; k < rk[r].n_subblocks; ++k) {
                    const ngsc_subblock_t &x = sb[rk[r].first_subblock + k];
                    sizes[k] = x.info_len + x.title_len + x.qual_len + x.dna_len;
                    bytes += sizes[k];
                }
                blocks.resize((size_t)(bytes / (NGSC_BLOCK_SIZE - 4096)) + 4);
                uint32_t nb = 0;
                mb.rc = ngsc_plan_blocks(P, rk[r].rank, sizes.data(), (uint32_t)sizes.size(), blocks.data(), (uint32_t)blocks.size(), &nb);
                if (!mb.rc && nb + 3 > row_len) mb.rc = NGSC_ERR_CAPACITY;
                if (mb.rc) break;
                uint64_t *row = mine.data() + (size_t)r * row_len;
                row[0] = nb;
                row[1] = rk[r].n_subblocks;
                row[2] = (uint64_t)(int64_t)rk[r].overlap_used;
                for (uint32_t k = 0; k < nb; ++k) row[3 + k] = blocks[k].size;
            }
        }
        mb.rows.resize((size_t)count * n);
        // (a member whose step failed says so in the vote in front of the collective: nobody enters it then)
        const auto tx = std::chrono::steady_clock::now();
        const int xr = member_allgather(t, i, mb.rc, mine.data(), count, mb.rows.data());
        mb.xchg_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tx).count();
        if (!mb.rc) mb.rc = xr;
        mb.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    });
    int rc = NGSC_OK;
    for (uint32_t i = 0; i < n; ++i) {
        if (t->m[i].rc && !rc) rc = t->m[i].rc;
        if (shard_ms) shard_ms[i] = t->m[i].ms;
    }
    if (rc) return rc;
    // every member holds the same rows, or the collective is broken
    for (uint32_t i = 1; i < n; ++i)
        if (memcmp(t->m[i].rows.data(), t->m[0].rows.data(), 8 * (size_t)count * n)) return NGSC_ERR_HIP;
    memcpy(rows, t->m[0].rows.data(), 8 * (size_t)count * n);
    return NGSC_OK;
}

// The local prefix sum behind the exchange: file offset of every block in the deterministic order of ngsc.h (groups of
// consecutive ranks covering at most NGSC_ORDER_GROUP_BYTES of input; inside a group round k = k-th block of each rank).
int ngsc_rows_to_offsets(const uint64_t *rows, uint32_t P, uint32_t row_len, uint64_t file_size, uint64_t *offsets, uint32_t *cbo,
                         uint32_t cbo_cap, uint32_t *n_blocks, uint64_t *end)
{
    if (!rows || !P || row_len < 4 || !offsets) return NGSC_ERR_ARG;
    const uint64_t region = file_size / P;
    uint64_t rpg = region ? NGSC_ORDER_GROUP_BYTES / region : P;
    rpg = std::max<uint64_t>(1, std::min<uint64_t>(P, rpg));
    uint64_t off = 0;
    uint32_t nb = 0;
    for (uint32_t g0 = 0; g0 < P; g0 += (uint32_t)rpg) {
        const uint32_t g1 = (uint32_t)std::min<uint64_t>(P, g0 + rpg);
        uint64_t maxb = 0;
        for (uint32_t r = g0; r < g1; ++r) {
            if (rows[(size_t)r * row_len] > (uint64_t)row_len - 3) return NGSC_ERR_FORMAT;   // (also the all-ones word of a row nobody filled)
            maxb = std::max(maxb, rows[(size_t)r * row_len]);
        }
        for (uint64_t k = 0; k < maxb; ++k)
            for (uint32_t r = g0; r < g1; ++r)
                if (k < rows[(size_t)r * row_len]) {
                    offsets[(size_t)r * (row_len - 3) + k] = off;
                    off += rows[(size_t)r * row_len + 3 + k];
                    if (cbo) { if (nb >= cbo_cap) return NGSC_ERR_CAPACITY; cbo[nb] = r; }
                    ++nb;
                }
    }
    if (n_blocks) *n_blocks = nb;
    if (end) *end = off;
    return NGSC_OK;
}

}  // extern "C"
