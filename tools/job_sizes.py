#!/usr/bin/env python3
"""Developer tool (GPU box): step time of the 36-base workload at several job sizes (what the file driver's jobs and small files see).
    python tools/job_sizes.py [MB ...]        default 125 250 500 1000"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from phyngsc_amd import api, fastq_gen

sizes = [int(x) for x in sys.argv[1:]] or [125, 250, 500, 1000]
eng = api.Engine(0)
for mb in sizes:
    data = fastq_gen.generate(fastq_gen.CFG2, target_bytes=mb * 1_000_000)
    d_in = torch.from_numpy(data).cuda()
    out = torch.empty(int(data.size * 0.6) + (32 << 20), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        eng.compress_device(d_in, data.size, 8, out=out)
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8):
            eng.compress_device(d_in, data.size, 8, out=out)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 8)
    ts.sort()
    print(f"size {mb:5d} MB  {data.size / ts[2] / 1e9:6.1f} GB/s  median {ts[2] * 1e3:.3f} ms  min {ts[0] * 1e3:.3f} ms", flush=True)
