// ubench_graph.hip - what does a captured HIP graph save between the dependent kernels of a step on gfx950?
// (measurement helper, not product)
//
// A compress step is ~20 launches, most of them dependent on the one before, with three fork / join pairs to a side stream
// (engine.cpp: compress_impl).  This replays that SHAPE with kernels that do nothing but wait a fixed time on the constant
// clock, so that  step time - sum of the waits  is what launches, dependencies and joins cost:
//   chain   K kernels on one stream, each behind the one before
//   forked  the same K kernels, every fifth one on a second stream between an event fork and an event join
// each as plain launches and as one hipGraphLaunch of the captured sequence, for small (one workgroup) and chip-filling
// (1024 workgroups) grids and for waits of 2 / 20 / 100 us.  Prints the overhead per kernel in microseconds.
//   hipcc --offload-arch=gfx950 -O3 -o ubench_graph ubench_graph.hip && ./ubench_graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_wait(uint64_t ticks, uint32_t *sink)
{
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink += 1;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Shape { int K; bool forked; uint32_t grid; uint64_t ticks; };

static int enqueue(const Shape &s, hipStream_t st, hipStream_t side, hipEvent_t fork, hipEvent_t join, uint32_t *sink)
{
    for (int i = 0; i < s.K; ++i) {
        if (s.forked && i % 5 == 2) {
            CK(hipEventRecord(fork, st));
            CK(hipStreamWaitEvent(side, fork, 0));
            k_wait<<<s.grid, 256, 0, side>>>(s.ticks, nullptr);
            CK(hipEventRecord(join, side));
            k_wait<<<s.grid, 256, 0, st>>>(s.ticks, sink);   // beside the side stream's kernel
            CK(hipStreamWaitEvent(st, join, 0));
            ++i;
        } else {
            k_wait<<<s.grid, 256, 0, st>>>(s.ticks, sink);
        }
    }
    return 0;
}

// the waits on the critical path: a forked pair runs side by side
static int critical_kernels(const Shape &s)
{
    int n = 0;
    for (int i = 0; i < s.K; ++i) { if (s.forked && i % 5 == 2) ++i; ++n; }
    return n;
}

int main()
{
    int dev = 0;
    CK(hipSetDevice(dev));
    int khz = 0;
    CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev));
    if (khz <= 0) { printf("no wall clock rate\n"); return 1; }
    hipStream_t st, side;
    CK(hipStreamCreate(&st));
    CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, -1));
    hipEvent_t fork, join, e0, e1;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    uint32_t *sink;
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(sink, 0, 4));
    printf("wall clock %d kHz; overhead = (time of a step - waits on its critical path) / kernels on the critical path\n", khz);
    printf("%-8s %6s %8s %10s | %12s %12s | %12s %12s\n", "shape", "grid", "wait us", "kernels", "plain us/k", "host us/k", "graph us/k", "host us/k");
    const int REPS = 200, K = 20;
    for (int forked = 0; forked < 2; ++forked)
        for (uint32_t grid : {1u, 1024u})
            for (double wait_us : {2.0, 20.0, 100.0}) {
                Shape s{K, forked != 0, grid, (uint64_t)(wait_us * khz / 1000.0)};
                const int nc = critical_kernels(s);
                const double waits = nc * wait_us;
                // plain launches
                for (int r = 0; r < 5; ++r) if (enqueue(s, st, side, fork, join, sink)) return 1;
                CK(hipStreamSynchronize(st));
                double host_plain = 0;
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < REPS; ++r) {
                    const double h0 = now_us();
                    if (enqueue(s, st, side, fork, join, sink)) return 1;
                    host_plain += now_us() - h0;
                }
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms_plain = 0;
                CK(hipEventElapsedTime(&ms_plain, e0, e1));
                // one graph per step
                hipGraph_t g;
                hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                if (enqueue(s, st, side, fork, join, sink)) return 1;
                CK(hipStreamEndCapture(st, &g));
                CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, st));
                CK(hipStreamSynchronize(st));
                double host_graph = 0;
                CK(hipEventRecord(e0, st));
                for (int r = 0; r < REPS; ++r) {
                    const double h0 = now_us();
                    CK(hipGraphLaunch(ge, st));
                    host_graph += now_us() - h0;
                }
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                float ms_graph = 0;
                CK(hipEventElapsedTime(&ms_graph, e0, e1));
                CK(hipGraphExecDestroy(ge));
                CK(hipGraphDestroy(g));
                printf("%-8s %6u %8.0f %10d | %12.2f %12.2f | %12.2f %12.2f\n", forked ? "forked" : "chain", grid, wait_us, nc,
                       (ms_plain * 1000.0 / REPS - waits) / nc, host_plain / REPS / s.K, (ms_graph * 1000.0 / REPS - waits) / nc,
                       host_graph / REPS / s.K);
                fflush(stdout);
            }
    // every step waited for by the host (as a call of the engine is): wall clock of launch + sync, the 20 us chain
    for (int graph = 0; graph < 2; ++graph) {
        Shape s{K, true, 1024u, (uint64_t)(20.0 * khz / 1000.0)};
        hipGraph_t g = nullptr;
        hipGraphExec_t ge = nullptr;
        if (graph) {
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            if (enqueue(s, st, side, fork, join, sink)) return 1;
            CK(hipStreamEndCapture(st, &g));
            CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        }
        double tot = 0;
        for (int r = 0; r < REPS + 5; ++r) {
            const double h0 = now_us();
            if (graph) CK(hipGraphLaunch(ge, st));
            else if (enqueue(s, st, side, fork, join, sink)) return 1;
            CK(hipStreamSynchronize(st));
            if (r >= 5) tot += now_us() - h0;
        }
        printf("synchronous steps (forked, 1024 workgroups, 20 us waits, %d on the critical path): %s %.1f us per step (waits %.0f)\n",
               critical_kernels(s), graph ? "graph" : "plain", tot / REPS, critical_kernels(s) * 20.0);
        if (graph) { CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
    }
    return 0;
}
