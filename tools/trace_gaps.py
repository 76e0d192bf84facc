#!/usr/bin/env python3
"""Developer tool: timeline of one bench step from a rocprofv3 --kernel-trace (+ --memory-copy-trace) CSV dump.
Prints every kernel / copy of the LAST step with its start offset, duration and the idle gap before it."""
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
i0 = starts[-1]
t0 = ev[i0][0]
prev_end = t0
busy = 0
for s, e, n in ev[i0:]:
    gap = s - prev_end
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {n}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us, busy {busy / 1e3:.1f} us")
