#!/bin/bash
# Developer tool (GPU box): the measurements a round's DESIGN.md and profiles/ quote, from one build, in one call.
#   tools/round_evidence.sh r04            (everything)        tools/round_evidence.sh r04 cfg2|cfg3|cfg5|bench   (one part)
# Writes under gpurun_out/evidence_<tag>/ ; copy what is to be judged into profiles/.
tag=${1:-rXX}
part=${2:-all}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out/evidence_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
stats() {   # $1 = name, rest = program and arguments: rocprofv3 --kernel-trace --stats summary of it
    name=$1; shift
    rm -rf $out/stats_$name
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$name -- "$@" > $out/stats_$name.log 2>&1
    f=$(find $out/stats_$name -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && cp $f $out/${tag}_kernel_stats_$name.csv
    find $out/stats_$name -name "*.csv" -size +8M -delete
}
if [ $part = all ] || [ $part = bench ]; then
    python3 $repo/bench.py --steps 20 --warmup 5 > $out/${tag}_bench_final.json 2> $out/bench.err
    tail -c 600 $out/${tag}_bench_final.json; echo
fi
if [ $part = all ] || [ $part = cfg2 ]; then
    stats cfg2 python3 $repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-verify --no-other-configs
    (cd $repo && tools/pmc_pass.sh ${tag}_cfg2 && tools/pmc_mem.sh ${tag}_cfg2 && tools/pmc_traffic.sh ${tag}_cfg2) > $out/pmc_cfg2.log 2>&1
    echo "cfg2 counters done"
fi
for c in cfg3_150bp cfg5_36_300; do
    short=${c%%_*}
    if [ $part = all ] || [ $part = $short ]; then
        export NGSC_TC_ONLY=$c PMC_SCRIPT=tools/time_configs.py PMC_ARGS=400000000
        stats $short python3 $repo/tools/time_configs.py 400000000
        (cd $repo && tools/pmc_pass.sh ${tag}_$short && tools/pmc_traffic.sh ${tag}_$short) > $out/pmc_$short.log 2>&1
        unset NGSC_TC_ONLY PMC_SCRIPT PMC_ARGS
        echo "$short counters done"
    fi
done
ls $out $repo/gpurun_out | grep -c "$tag" || true
