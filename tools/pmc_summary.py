#!/usr/bin/env python3
"""Developer tool: averages rocprofv3 --pmc counters per kernel.

  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... --kernel-trace --output-format csv -d gpurun_out/pmcX -- python3 bench.py ...
  python3 tools/pmc_summary.py gpurun_out/pmcX [kernel-name-prefix]

Prints, per kernel (launch average): every counter collected, plus VGPR/LDS/scratch of the code object.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else "k_"
    rows = defaultdict(lambda: defaultdict(list))
    meta = {}
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0].replace("ngsc::", "").replace("void ", "")
                if not name.startswith(prefix):
                    continue
                rows[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                meta[name] = dict(vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]),
                                  lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]), wg=int(r["Workgroup_Size"]),
                                  grid=int(r["Grid_Size"]))
    out = {}
    for name, ctr in rows.items():
        d = dict(meta[name])
        for c, vals in ctr.items():
            per_dispatch = defaultdict(float)
            for disp, v in vals:
                per_dispatch[disp] += v      # one row per dispatch (and per dimension, if any): sum those
            d[c] = sum(per_dispatch.values()) / len(per_dispatch)
            d["launches"] = len(per_dispatch)
        out[name] = d
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
