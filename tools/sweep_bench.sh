#!/bin/bash
# Developer tool (GPU box): bench.py under several settings of one environment variable; prints the kernel table of each.
#   tools/sweep_bench.sh VAR v1 v2 ...      (extra bench arguments in BENCH_ARGS)
var=$1; shift
for v in "$@"; do
    echo "== $var=$v"
    env $var=$v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-verify $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  %.1f GB/s  %.3f ms  sum %.3f' % (d['value'], d['ms_per_step'], d['pipeline']['kernel_ms_sum'])); print('  ', {k:v for k,v in list(d['kernels_ms'].items())[:8]})
"
done
