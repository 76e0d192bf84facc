#!/bin/bash
# Developer tool (GPU box): the measurements a round's DESIGN.md and profiles/ quote, from one build, in one call.
#   tools/round_evidence.sh r04            (everything)        tools/round_evidence.sh r04 cfg2|cfg3|cfg5|bench|trace|decode   (one part)
# Writes under gpurun_out/evidence_<tag>/ (and gpurun_out/{pmc,traffic}_<tag>_*); copy what is to be judged into profiles/.
tag=${1:-rXX}
part=${2:-all}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/evidence_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
chk() { rc=$?; if [ $rc -ge 124 ]; then echo "step killed (rc $rc): stopping"; exit $rc; fi; }
stats() {   # $1 = name, rest = program and arguments: rocprofv3 --kernel-trace --stats summary of it
    name=$1; shift
    rm -rf $out/stats_$name
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$name -- "$@" > $out/stats_$name.log 2>&1; chk
    f=$(find $out/stats_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $out/${tag}_kernel_stats_$name.csv
    find $out/stats_$name -name "*.csv" -size +8M -delete
}
if [ $part = all ] || [ $part = bench ]; then
    timeout -k 10 600 python3 $repo/bench.py --steps 20 --warmup 5 > $out/${tag}_bench_final.json 2> $out/bench.err; chk
    tail -c 700 $out/${tag}_bench_final.json; echo
fi
if [ $part = all ] || [ $part = trace ]; then
    rm -rf $out/trace
    timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace -- python3 $repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify > $out/trace.log 2>&1; chk
    python3 $repo/tools/trace_gaps.py $out/trace 5 > $out/${tag}_trace_cfg2_step.txt 2>&1
    python3 $repo/tools/trace_gaps.py $out/trace 12 > $out/${tag}_trace_cfg3_step.txt 2>&1
    python3 $repo/tools/trace_gaps.py $out/trace 20 > $out/${tag}_trace_cfg5_step.txt 2>&1
    find $out/trace -name "*.csv" -size +8M -delete
    echo "traces done"
fi
if [ $part = all ] || [ $part = cfg2 ]; then
    stats cfg2 python3 $repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-other-configs
    (cd $repo && timeout -k 10 900 tools/pmc_pass.sh ${tag}_cfg2 && timeout -k 10 900 tools/pmc_mem.sh ${tag}_cfg2 && timeout -k 10 600 tools/pmc_traffic.sh ${tag}_cfg2) > $out/pmc_cfg2.log 2>&1; chk
    echo "cfg2 counters done"
fi
for c in cfg3_150bp cfg5_36_300; do
    short=${c%%_*}
    if [ $part = all ] || [ $part = $short ]; then
        export NGSC_TC_ONLY=$c PMC_SCRIPT=tools/time_configs.py PMC_ARGS=400000000
        stats $short python3 $repo/tools/time_configs.py 400000000
        (cd $repo && timeout -k 10 900 tools/pmc_pass.sh ${tag}_$short && timeout -k 10 600 tools/pmc_traffic.sh ${tag}_$short) > $out/pmc_$short.log 2>&1; chk
        unset NGSC_TC_ONLY PMC_SCRIPT PMC_ARGS
        echo "$short counters done"
    fi
done
if [ $part = all ] || [ $part = decode ]; then
    for c in cfg2 cfg3 cfg5; do
        NGSC_VERBOSE=1 timeout -k 10 300 python3 $repo/tools/time_decode.py 1000000000 $c > $out/${tag}_decode_1GB_$c.txt 2>&1; chk
        tail -2 $out/${tag}_decode_1GB_$c.txt
    done
fi
ls $out | head -40
