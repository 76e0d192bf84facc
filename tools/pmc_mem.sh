#!/bin/bash
# Developer tool (GPU box): memory-path counters per kernel (texture addresser / L1 busy and stalls, LDS), ONE hardware
# block per rocprofv3 pass: several TA_/TCP_ counters in one pass exceed what the hardware can collect, and rocprofv3
# then aborts and hangs in its finaliser.  Every pass runs under its own timeout.
#   tools/pmc_mem.sh <tag>   -> gpurun_out/pmc_<tag>_m*.json
tag=${1:-x}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {
    sfx=$1; shift
    echo "pass $sfx: $*" >> $out/pmc_${tag}_progress.txt
    timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/pmc_${tag}_$sfx -- \
        python3 $repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-verify > $out/pmc_${tag}_$sfx.log 2>&1 || { echo "pass $sfx failed" >> $out/pmc_${tag}_progress.txt; return 1; }
    python3 $repo/tools/pmc_summary.py $out/pmc_${tag}_$sfx > $out/pmc_${tag}_$sfx.json
    find $out/pmc_${tag}_$sfx -name "*.csv" -size +8M -delete
}
run m1 TA_BUSY_avr GRBM_GUI_ACTIVE || exit 1
run m2 TA_ADDR_STALLED_BY_TC_CYCLES_sum || exit 1
run m3 TCP_PENDING_STALL_CYCLES_sum || exit 1
run m4 SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE || exit 1
echo done
