#!/bin/bash
# Developer tool (GPU box): bench.py under several values of one environment variable.
#   tools/sweep_env.sh NAME v1 v2 ...   -> gpurun_out/sweep_NAME.txt (value, GB/s, ms/step, per-kernel ms)
name=$1; shift
out=gpurun_out/sweep_$name.txt
: > $out
for v in "$@"; do
    env $name=$v python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify > gpurun_out/sweep_tmp.json 2> gpurun_out/sweep_tmp.err || { echo "$v FAILED" >> $out; continue; }
    python3 - "$v" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/sweep_tmp.json"))
k = d["kernels_ms"]
top = ", ".join(f"{n}={ms:.3f}" for n, ms in list(k.items())[:8])
print(f"{sys.argv[1]:>8}  {d['value']:.1f} GB/s  {d['ms_per_step']:.3f} ms  sum={d['pipeline']['kernel_ms_sum']:.3f}  {top}")
PY
done
cat $out
