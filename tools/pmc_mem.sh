#!/bin/bash
# Developer tool (GPU box): memory-path counters per kernel (texture addresser / L1 busy and stalls, LDS), ONE hardware
# block per rocprofv3 pass.  Observed in round 1 (no log of it was kept, so the cause is not proven): a pass that asked for
# several TA_/TCP_ counters at once did not finish - presumably more than the block's counter registers - whereas one
# block per pass always has.  Every pass runs under its own timeout.  PMC_SCRIPT / PMC_ARGS as in pmc_pass.sh.
#   tools/pmc_mem.sh <tag>   -> gpurun_out/pmc_<tag>_m*.json
tag=${1:-x}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {
    sfx=$1; shift
    echo "pass $sfx: $*" >> $out/pmc_${tag}_progress.txt
    timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_${tag}_$sfx -- \
        python3 $repo/${PMC_SCRIPT:-bench.py} ${PMC_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-other-configs} > $out/pmc_${tag}_$sfx.log 2>&1 || { echo "pass $sfx failed" >> $out/pmc_${tag}_progress.txt; return 1; }
    python3 $repo/tools/pmc_summary.py $out/pmc_${tag}_$sfx > $out/pmc_${tag}_$sfx.json
    find $out/pmc_${tag}_$sfx -name "*.csv" -size +8M -delete
}
run m1 TA_BUSY_avr GRBM_GUI_ACTIVE || exit 1
run m2 TA_ADDR_STALLED_BY_TC_CYCLES_sum || exit 1
run m3 TCP_PENDING_STALL_CYCLES_sum || exit 1
run m4 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE || exit 1
echo done
