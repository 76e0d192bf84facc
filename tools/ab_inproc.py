#!/usr/bin/env python3
"""Developer tool (GPU box): A/B of environment settings inside ONE process, the variants taking turns (A B A B ...), so that
box-to-box and minute-to-minute drift (clocks, neighbours) hits them alike.

    python tools/ab_inproc.py [--configs cfg2,cfg3,cfg5] [--mb 1000,400,400] [--rounds 4] [--steps 8] "-" "NGSC_NO_SPEC=1" ...

Per configuration and variant: the median and the minimum of the per-round step times (a round = `steps` back-to-back calls,
timed as a whole), and the per-kernel table of one profiled call.  "-" = no setting.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from phyngsc_amd import api, fastq_gen  # noqa: E402

CFGS = {"cfg2": fastq_gen.CFG2, "cfg3": fastq_gen.CFG3, "cfg5": fastq_gen.CFG5_300, "cfg1": fastq_gen.CFG1}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg2")
    ap.add_argument("--mb", default="1000")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--kernels", type=int, default=9)
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    names = a.configs.split(",")
    mbs = [int(x) for x in a.mb.split(",")]
    if len(mbs) == 1:
        mbs = mbs * len(names)
    for name, mb in zip(names, mbs):
        data = fastq_gen.generate(CFGS[name], target_bytes=mb * 1_000_000)
        d_in = torch.from_numpy(data).cuda()
        out = torch.empty(int(data.size * 0.6) + (32 << 20), dtype=torch.uint8, device="cuda")
        engines, times, kts = [], [[] for _ in a.variants], []
        base_env = dict(os.environ)

        def apply(v):
            os.environ.clear()
            os.environ.update(base_env)
            if v != "-":
                for kv in v.split(","):
                    k, _, val = kv.partition("=")
                    os.environ[k] = val
        ref_bytes = None
        for v in a.variants:
            apply(v)
            eng = api.Engine(0)
            for _ in range(3):
                res = eng.compress_device(d_in, data.size, 8, out=out)
            got = out[:res.total_bytes].cpu().numpy().tobytes()
            if ref_bytes is None:
                ref_bytes = got
            elif got != ref_bytes:
                print(f"!! {name} {v}: output differs from the first variant's", flush=True)
            eng.set_profiling(True)
            eng.reset_kernel_times()
            eng.compress_device(d_in, data.size, 8, out=out)
            torch.cuda.synchronize()
            kt = eng.kernel_times()
            eng.set_profiling(False)
            kts.append(sorted(((ms / max(n, 1), k) for k, (ms, n) in kt.items()), reverse=True))
            engines.append(eng)
        for _ in range(a.rounds):
            for i, v in enumerate(a.variants):
                apply(v)
                eng = engines[i]
                eng.compress_device(d_in, data.size, 8, out=out)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    eng.compress_device(d_in, data.size, 8, out=out)
                torch.cuda.synchronize()
                times[i].append((time.perf_counter() - t0) / a.steps)
        for i, v in enumerate(a.variants):
            t = sorted(times[i])
            med, mn = t[len(t) // 2], t[0]
            print(f"{name} {data.size / 1e6:7.1f} MB  {v:>28s}  median {med * 1e3:6.3f} ms = {data.size / med / 1e9:6.1f} GB/s   min {mn * 1e3:6.3f} ms = {data.size / mn / 1e9:6.1f} GB/s   "
                  + ", ".join(f"{k}={ms:.3f}" for ms, k in kts[i][:a.kernels]) + f"  sum={sum(ms for ms, _ in kts[i]):.3f}", flush=True)
        for e in engines:
            e.close()
        os.environ.clear()
        os.environ.update(base_env)
        del d_in, out


if __name__ == "__main__":
    main()
