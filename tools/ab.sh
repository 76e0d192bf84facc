#!/bin/bash
# Developer tool (GPU box): bench.py under several environment settings, one line each.
#   tools/ab.sh OUT "NAME=v NAME2=v2" "NAME=v" ...      ("-" = no setting)
out=$1; shift
: > $out
for setting in "$@"; do
    envs=""; [ "$setting" != "-" ] && envs="$setting"
    env $envs python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$setting FAILED" >> $out; tail -3 gpurun_out/ab_tmp.err >> $out; continue; }
    python3 - "$setting" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
k = d["kernels_ms"]
top = ", ".join(f"{n}={ms:.3f}" for n, ms in list(k.items())[:12])
print(f"{sys.argv[1]:>24}  {d['value']:.1f} GB/s  {d['ms_per_step']:.3f} ms  sum={d['pipeline']['kernel_ms_sum']:.3f}  {top}")
PY
done
cat $out
