#!/usr/bin/env python3
"""Developer tool: timeline of one bench step from a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV dump.
Prints every kernel / copy of one step (the LAST one, or the step with the index given as second argument; the steps are listed
first) with its start offset, duration and the idle gap before it."""
import csv, glob, os, sys

root = sys.argv[1]
ev = []
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("ngsc::", "").replace("void ", "")))
for path in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path, newline="")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", "?") ))
ev.sort()
# last occurrence of k_sync_first starts the last step
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_sync_first")]
for j, i in enumerate(starts):
    i1 = starts[j + 1] if j + 1 < len(starts) else len(ev)
    seg = [x for x in ev[i:i1] if not x[2].startswith("copy:MEMORY_COPY_HOST_TO_DEVICE")]
    print(f"step {j}: {len(seg)} events, {(max(x[1] for x in seg) - seg[0][0]) / 1e3:.1f} us, statistics kernel {[x[2] for x in seg if 'sq_stats' in x[2] or 'seq_qual' in x[2]][:1]}")
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 1
i0 = starts[which]
i_end = starts[which + 1] if which + 1 < len(starts) else len(ev)
t0 = ev[i0][0]
prev_end = t0
busy = 0
print(f"---- step {which} ----")
for s, e, n in [x for x in ev[i0:i_end] if not x[2].startswith("copy:MEMORY_COPY_HOST_TO_DEVICE")]:
    gap = s - prev_end
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {n}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
