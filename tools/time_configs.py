#!/usr/bin/env python3
"""Developer tool (GPU box): throughput and per-kernel times of the engine on the other generator configurations
(variable read lengths, ambiguity codes, long titles) - the bench workload is cfg2 only."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phyngsc_amd import api, fastq_gen

size = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
eng = api.Engine(0)
only = os.environ.get("NGSC_TC_ONLY")
for name, cfg in (("cfg1", fastq_gen.CFG1), ("cfg2", fastq_gen.CFG2), ("cfg3_150bp", fastq_gen.CFG3), ("cfg5_mixed", fastq_gen.CFG5), ("cfg5_36_300", fastq_gen.CFG5_300)):
    if only and only != name:
        continue
    data = fastq_gen.generate(cfg, target_bytes=size)
    d_in = torch.from_numpy(data).cuda()
    out = torch.empty(int(data.size * 0.6) + (32 << 20), dtype=torch.uint8, device="cuda")
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
    top = sorted(((ms / max(n, 1), k) for k, (ms, n) in kt.items()), reverse=True)[:7]
    print(f"{name:11s} {data.size/1e6:7.1f} MB  {data.size/dt/1e9:6.1f} GB/s  {dt*1e3:6.2f} ms  ratio {res.total_bytes/data.size:.3f}  " +
          ", ".join(f"{k}={ms:.3f}" for ms, k in top))
    del d_in, out
