#!/bin/bash
# Developer tool (GPU box): PMC passes over bench.py, one rocprofv3 run per counter group.
#   tools/pmc_pass.sh <tag>      -> gpurun_out/pmc_<tag>_{a,b}/ + gpurun_out/pmc_<tag>_{a,b}.json (per-kernel averages)
#   PMC_SCRIPT=tools/time_configs.py PMC_ARGS=400000000 NGSC_TC_ONLY=cfg3_150bp tools/pmc_pass.sh cfg3   (other workloads)
set -e
tag=${1:-x}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # $1 = suffix, rest = counters
    sfx=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_${tag}_$sfx -- \
        python3 $repo/${PMC_SCRIPT:-bench.py} ${PMC_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-other-configs} > $out/pmc_${tag}_$sfx.log 2>&1
    python3 $repo/tools/pmc_summary.py $out/pmc_${tag}_$sfx > $out/pmc_${tag}_$sfx.json
    find $out/pmc_${tag}_$sfx -name "*.csv" -size +8M -delete
}
run a SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY GRBM_GUI_ACTIVE
echo done
