#!/usr/bin/env python3
"""Developer tool (GPU box): A/B of environment settings and / or builds, the variants taking turns (A B A B ...) on the same
GPU within seconds of each other, so that box-to-box and minute-to-minute drift (clocks, neighbours) hits them alike.

    python tools/ab_inproc.py [--configs cfg2,cfg3,cfg5] [--mb 1000,400,400] [--rounds 4] [--steps 8] "-" "NGSC_NO_SPEC=1" "LIB=/path/libB.so" ...

A variant is a comma-separated list of NAME=value environment settings ("-" = none); LIB=path loads another build of
libngsc_hip.so (NGSC_LIB_PATH, read by phyngsc_amd/_lib.py).  Every variant runs in a worker process of its own (its
environment and library are fixed at start); the parent tells the workers in turn to time `steps` back-to-back calls.
Per configuration and variant: the median and the minimum of the per-round step times, the per-kernel table of one
profiled call, and whether the output equals the first variant's.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(name, mb, steps):
    import torch
    from phyngsc_amd import api, fastq_gen
    cfgs = {"cfg2": fastq_gen.CFG2, "cfg3": fastq_gen.CFG3, "cfg5": fastq_gen.CFG5_300, "cfg1": fastq_gen.CFG1}
    data = fastq_gen.generate(cfgs[name], target_bytes=mb * 1_000_000)
    d_in = torch.from_numpy(data).cuda()
    out = torch.empty(int(data.size * 0.6) + (32 << 20), dtype=torch.uint8, device="cuda")
    eng = api.Engine(0)
    for _ in range(3):
        res = eng.compress_device(d_in, data.size, 8, out=out)
    digest = hashlib.sha256(out[:res.total_bytes].cpu().numpy().tobytes()).hexdigest()
    eng.set_profiling(True)
    eng.reset_kernel_times()
    eng.compress_device(d_in, data.size, 8, out=out)
    torch.cuda.synchronize()
    kt = eng.kernel_times()
    eng.set_profiling(False)
    kts = sorted(((ms / max(n, 1), k) for k, (ms, n) in kt.items()), reverse=True)
    print(json.dumps({"ready": True, "bytes": int(data.size), "sha": digest, "kernels": kts}), flush=True)
    for line in sys.stdin:
        if line.strip() != "go":
            break
        eng.compress_device(d_in, data.size, 8, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            eng.compress_device(d_in, data.size, 8, out=out)
        torch.cuda.synchronize()
        print(json.dumps({"t": (time.perf_counter() - t0) / steps}), flush=True)
    eng.close()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        return
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
        procs, infos = [], []
        for v in a.variants:
            env = dict(os.environ)
            if v != "-":
                for kv in v.split(","):
                    k, _, val = kv.partition("=")
                    env["NGSC_LIB_PATH" if k == "LIB" else k] = val
            p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", name, str(mb), str(a.steps)], env=env,
                                 stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
            procs.append(p)
        for p in procs:
            line = p.stdout.readline()
            while line and not line.startswith("{"):
                line = p.stdout.readline()
            infos.append(json.loads(line) if line else {"ready": False})
        times = [[] for _ in a.variants]
        for _ in range(a.rounds):
            for i, p in enumerate(procs):
                if not infos[i].get("ready"):
                    continue
                p.stdin.write("go\n")
                p.stdin.flush()
                times[i].append(json.loads(p.stdout.readline())["t"])
        for i, v in enumerate(a.variants):
            if not infos[i].get("ready"):
                print(f"{name} {v}: worker failed", flush=True)
                continue
            t = sorted(times[i])
            med, mn, nbytes = t[len(t) // 2], t[0], infos[i]["bytes"]
            first = next(x for x in infos if x.get("ready"))
            same = "" if infos[i]["sha"] == first["sha"] else "  !! OUTPUT DIFFERS from the first variant's"
            kts = infos[i]["kernels"]
            print(f"{name} {nbytes / 1e6:7.1f} MB  {v:>34s}  median {med * 1e3:6.3f} ms = {nbytes / med / 1e9:6.1f} GB/s   min {mn * 1e3:6.3f} ms = {nbytes / mn / 1e9:6.1f} GB/s   "
                  + ", ".join(f"{k}={ms:.3f}" for ms, k in kts[:a.kernels]) + f"  sum={sum(ms for ms, _ in kts):.3f}{same}", flush=True)
        for p in procs:
            try:
                p.stdin.write("quit\n")
                p.stdin.flush()
                p.wait(timeout=60)
            except Exception:
                p.kill()


if __name__ == "__main__":
    main()
