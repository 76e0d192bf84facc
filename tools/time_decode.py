#!/usr/bin/env python3
"""Developer tool (GPU box): compress a generated FASTQ on the GPU, decode it on the GPU, compare, print the decode rate."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from phyngsc_amd import api, fastq_gen

size = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
cfg = {"cfg2": fastq_gen.CFG2, "cfg3": fastq_gen.CFG3, "cfg5": fastq_gen.CFG5_300}[sys.argv[2] if len(sys.argv) > 2 else "cfg2"]
data = fastq_gen.generate(cfg, target_bytes=size)
eng = api.Engine(0)
d_in = torch.from_numpy(data).cuda()
res = eng.compress_device(d_in, data.size, 8)
offs = [sb.out_off for sb in res.subblocks]
lens = [sb.size for sb in res.subblocks]
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    text, toff, nrec = eng.decompress_device(res.out, offs, lens, text_cap=data.size + (1 << 20))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"decode {len(offs)} sub-blocks, {toff[-1]/1e9:.3f} GB text in {dt*1e3:.1f} ms = {toff[-1]/dt/1e9:.2f} GB/s", flush=True)
same = bool(torch.equal(text[:data.size], d_in))
print("IDENTICAL" if same and toff[-1] == data.size else f"DIFFERENT ({toff[-1]} vs {data.size})")
