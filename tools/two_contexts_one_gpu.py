#!/usr/bin/env python3
"""The bench workload (1 GB of 36-base reads, 8 virtual ranks, one GPU) compressed by ONE context that takes all eight ranks per
call - what bench.py times - against N contexts on the same GPU, a host thread each, that take 8 / N consecutive ranks each:
the ranks are independent, so the output is the same bytes, and the latency-bound stretches of one context's step (index tail,
small kernels, the host's wait) could be filled by the throughput kernels of another.  (measurement helper, not product)

    python tools/two_contexts_one_gpu.py [--size-mb 1000] [--steps 20] [--contexts 2 4]

lockstep: every step starts all contexts together and ends when the last is done (a step = the whole file, as in bench.py);
free:     every context runs its K calls without waiting for the others (their phases drift apart).
"""
from __future__ import annotations

import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from phyngsc_amd import api, fastq_gen  # noqa: E402

P = 8


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--size-mb", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--contexts", type=int, nargs="*", default=[2, 4])
    ap.add_argument("--cfg", default="CFG2")
    args = ap.parse_args()
    host = fastq_gen.generate(getattr(fastq_gen, args.cfg), target_bytes=args.size_mb * 1_000_000)
    fs = int(host.size)
    d_in = torch.from_numpy(host).cuda()
    print(f"{args.cfg}: {fs} B, {P} virtual ranks, {args.steps} steps after {args.warmup}")

    # ---- one context, all ranks (bench.py's step) ----
    eng = api.Engine(0)
    d_out = torch.empty(int(fs * 0.45) + (32 << 20), dtype=torch.uint8, device="cuda")
    for _ in range(args.warmup):
        res = eng.compress_device(d_in, fs, P, out=d_out, stream=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = eng.compress_device(d_in, fs, P, out=d_out, stream=0)
    torch.cuda.synchronize()
    ms1 = (time.perf_counter() - t0) / args.steps * 1e3
    print(f"1 context  x {P} ranks: {ms1:.3f} ms per file = {fs / ms1 / 1e6:.1f} GB/s")
    want = {}
    for rk in res.ranks:
        want[rk.rank] = bytes(res.out[rk.stream_off:rk.stream_off + rk.stream_bytes].cpu().numpy())

    for n in args.contexts:
        per = P // n
        engs, ins, outs, offs, ranks = [], [], [], [], []
        for i in range(n):
            lo, hi = api.job_byte_range(fs, P, i * per, (i + 1) * per)
            off = lo & ~15
            engs.append(api.Engine(0))
            ins.append(d_in[off:hi])
            offs.append(off)
            outs.append(torch.empty(int((hi - off) * 0.45) + (32 << 20), dtype=torch.uint8, device="cuda"))
            ranks.append((i * per, (i + 1) * per))
        results = [None] * n

        def call(i):
            results[i] = engs[i].compress_device(ins[i], fs, P, rank_lo=ranks[i][0], rank_hi=ranks[i][1], buf_off=offs[i], out=outs[i], stream=0)

        for mode in ("lockstep", "free"):
            bar = threading.Barrier(n)

            def work(i, steps, sync):
                for _ in range(steps):
                    if sync:
                        bar.wait()
                    call(i)

            def run(steps):
                th = [threading.Thread(target=work, args=(i, steps, mode == "lockstep")) for i in range(n)]
                for t in th:
                    t.start()
                for t in th:
                    t.join()

            run(args.warmup)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(args.steps)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            same = all(bytes(r.out[rk.stream_off:rk.stream_off + rk.stream_bytes].cpu().numpy()) == want[rk.rank] for r in results for rk in r.ranks)
            print(f"{n} contexts x {per} ranks, {mode:8s}: {ms:.3f} ms per file = {fs / ms / 1e6:.1f} GB/s ({ms1 / ms:.3f}x); "
                  f"rank streams identical to the one-context call: {same}", flush=True)
            if not same:
                return 1
        for e in engs:
            e.close() if hasattr(e, "close") else None
        del engs, outs
        torch.cuda.empty_cache()
    return 0


if __name__ == "__main__":
    sys.exit(main())
