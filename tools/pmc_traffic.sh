#!/bin/bash
# Developer tool (GPU box): HBM traffic per kernel launch, collected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (kilobytes; gfx950 reports half of wide coalesced
# reads, so fetch is doubled = an upper estimate for byte-granular access).  Writes gpurun_out/traffic_<tag>.json.
#   PMC_SCRIPT=tools/time_configs.py PMC_ARGS=400000000 NGSC_TC_ONLY=cfg3_150bp tools/pmc_traffic.sh cfg3   (other workloads)
set -e
tag=${1:-x}
repo=${GRAFT_REPO_ROOT:-/root/repo}
out=$repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${tag}_$c -- \
        python3 $repo/${PMC_SCRIPT:-bench.py} ${PMC_ARGS:---steps 2 --warmup 1 --no-cpu-baseline --no-verify --no-other-configs} > $out/pmc_${tag}_$c.log 2>&1
    python3 $repo/tools/pmc_summary.py $out/pmc_${tag}_$c > $out/pmc_${tag}_$c.json
    find $out/pmc_${tag}_$c -name "*.csv" -size +8M -delete
done
khash=$(cd $repo && python3 -c "import bench; print(bench.kernel_source_hash())" 2>/dev/null | tail -1)
python3 - "$out" "$tag" "$khash" <<'PY'
import json, re, sys
def norm(k):   # template arguments off; the word-parallel instance of the title measure kernel has its own name
    k = re.sub(r'k_title_measure_reg<true, \d+u?>', 'k_title_measure_swar', k)
    return re.sub(r'<[^>]*>', '', k)
out, tag = sys.argv[1], sys.argv[2]
f = json.load(open(f"{out}/pmc_{tag}_FETCH_SIZE.json")); w = json.load(open(f"{out}/pmc_{tag}_WRITE_SIZE.json"))
import os
res = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), per launch, " + (os.environ.get("PMC_SCRIPT", "bench.py 1 GB cfg2 workload") + " " + os.environ.get("NGSC_TC_ONLY", "")).strip() + "; FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads) - an upper estimate for byte-granular access",
       "build": tag, "kernel_source_hash": sys.argv[3]}
for k in f:
    if not isinstance(f[k], dict): continue
    fk = f[k].get("FETCH_SIZE", 0.0); wk = w.get(k, {}).get("WRITE_SIZE", 0.0)
    res[norm(k)] = {"fetch_size_kb_raw": fk, "write_size_kb": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
              "launches_sampled": f[k].get("launches")}
json.dump(res, open(f"{out}/traffic_{tag}.json", "w"), indent=1)
for k, v in sorted(res.items(), key=lambda kv: -(kv[1]["hbm_bytes_per_launch"] if isinstance(kv[1], dict) else 0)):
    if isinstance(v, dict): print(f"{k:28s} fetch_raw {v['fetch_size_kb_raw']/1e6:7.3f} GB  write {v['write_size_kb']/1e6:7.3f} GB")
PY
