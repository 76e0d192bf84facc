#!/usr/bin/env python3
"""Developer tool (GPU box): a FASTQ file larger than one engine job (4 GiB) through the command line tool, checked
byte for byte against the oracle.  usage: big_file_check.py [GB=5] [ranks=16[,ranks..]] [cfg2|cfg3|cfg5] [io=8,16,..] [decode]
io=..: the last rank count again under these NGSC_IO_THREADS values (output compared with the verified one); decode: `ngsc -d`
of the last output, compared with the input file."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from phyngsc_amd import fastq_gen
from oracle import binding as orc

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
PS = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16").split(",")]
cfg = {"cfg2": fastq_gen.CFG2, "cfg3": fastq_gen.CFG3, "cfg5": fastq_gen.CFG5}[sys.argv[3] if len(sys.argv) > 3 else "cfg2"]
IO = [int(x) for a in sys.argv[4:] if a.startswith("io=") for x in a[3:].split(",")]
DECODE = "decode" in sys.argv[4:]
exe = os.path.join(ROOT, "phyngsc_amd", "bin", "ngsc")
t0 = time.time()
data = fastq_gen.generate(cfg, target_bytes=int(gb * 1e9))
inp, outp = "/tmp/big_in.fastq", "/tmp/big_out.ngsc"
data.tofile(inp)
print(f"generated {data.size/1e9:.2f} GB in {time.time()-t0:.1f} s", flush=True)
ok_all = True
for P in PS:
    t0 = time.time()
    r = subprocess.run([os.path.join(ROOT, "phyngsc_amd", "bin", "ngsc"), inp, outp, "--ranks", str(P)], capture_output=True, text=True, env=dict(os.environ, NGSC_VERBOSE="1"))
    print(r.stdout.strip(), r.stderr.strip(), f"[ngsc wall {time.time()-t0:.1f} s, rc {r.returncode}]", flush=True)
    t0 = time.time()
    want, st = orc.compress_file(data, P, threads=min(P, os.cpu_count() or 1))
    print(f"oracle {time.time()-t0:.1f} s, {len(want)} B", flush=True)
    got = np.fromfile(outp, dtype=np.uint8)
    want = np.frombuffer(want, dtype=np.uint8) if isinstance(want, (bytes, bytearray)) else want
    same = got.size == want.size and bool(np.array_equal(got, want))
    ok_all = ok_all and same and r.returncode == 0
    print(f"ranks {P}: " + ("IDENTICAL" if same else f"DIFFERENT ({got.size} vs {want.size} bytes)"), flush=True)
    if P == PS[-1]:
        for io in IO:
            t0 = time.time()
            r = subprocess.run([exe, inp, outp + ".io", "--ranks", str(P)], capture_output=True, text=True,
                               env=dict(os.environ, NGSC_VERBOSE="1", NGSC_IO_THREADS=str(io)))
            wall = time.time() - t0
            same_io = r.returncode == 0 and bool(np.array_equal(np.fromfile(outp + ".io", dtype=np.uint8), got))
            print(f"NGSC_IO_THREADS={io}: " + r.stderr.strip() + f" [ngsc wall {wall:.1f} s, rc {r.returncode}, {'same file' if same_io else 'DIFFERENT'}]", flush=True)
            ok_all = ok_all and same_io
            os.remove(outp + ".io")
    del want, got
BACK = os.environ.get("BIG_BACK", "/tmp/big_back.fastq")   # (the GPU box's /tmp holds 79 GB: a 50 GB file is decoded to /dev/shm)
if DECODE:
    del data
    t0 = time.time()
    r = subprocess.run([exe, "-d", outp, BACK], capture_output=True, text=True, env=dict(os.environ, NGSC_VERBOSE="1"))
    print(r.stdout.strip(), r.stderr.strip()[-600:], f"[ngsc -d wall {time.time()-t0:.1f} s, rc {r.returncode}]", flush=True)
    same = r.returncode == 0 and subprocess.run(["cmp", "-s", inp, BACK]).returncode == 0
    print("decoded file: " + ("IDENTICAL to the input" if same else "DIFFERENT"), flush=True)
    ok_all = ok_all and same
    if os.path.exists(BACK):
        os.remove(BACK)
os.remove(inp); os.remove(outp)
sys.exit(0 if ok_all else 1)
