#!/usr/bin/env python3
"""Developer tool (GPU box): throughput on 150-base reads with dual-index Illumina headers (65..70 byte title lines),
word-parallel title kernels (two mask words) against the byte-serial fallback (NGSC_TITLE_PATH=reg)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phyngsc_amd import api

rnd = random.Random(1)
nrng = np.random.default_rng(1)
recs = []
for i in range(60000):
    t = b"@A00738:123:HLK2TDSXY:%d:%d:%d:%d 1:N:0:ACGTACGT+TGCATGCA" % (1 + i % 4, 1101 + i // 3000, rnd.randint(1000, 32000), rnd.randint(1000, 40000))
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[nrng.integers(0, 4, 150)].tobytes()
    q = np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8)[nrng.integers(0, 11, 150)].tobytes()
    recs.append(t + b"\n" + s + b"\n+\n" + q + b"\n")
block = b"".join(recs)
data = np.frombuffer(block * (400_000_000 // len(block)), dtype=np.uint8)
tl = [r.index(b"\n") + 1 for r in recs]
print(f"{data.size/1e6:.0f} MB, title lines of {min(tl)}..{max(tl)} bytes", flush=True)
eng = api.Engine(0)
d_in = torch.from_numpy(data.copy()).cuda()
out = torch.empty(int(data.size * 0.6) + (32 << 20), dtype=torch.uint8, device="cuda")
for path in ("swar", "reg"):
    os.environ["NGSC_TITLE_PATH"] = path
    for _ in range(2):
        res = eng.compress_device(d_in, data.size, 8, out=out)
    eng.set_profiling(True); eng.reset_kernel_times()
    res = eng.compress_device(d_in, data.size, 8, out=out)
    torch.cuda.synchronize()
    kt = eng.kernel_times(); eng.set_profiling(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        res = eng.compress_device(d_in, data.size, 8, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    top = sorted(((ms / max(n, 1), k) for k, (ms, n) in kt.items()), reverse=True)[:6]
    print(f"NGSC_TITLE_PATH={path:4s} {data.size/dt/1e9:6.1f} GB/s  {dt*1e3:6.2f} ms  " + ", ".join(f"{k}={ms:.3f}" for ms, k in top), flush=True)
