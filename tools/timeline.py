#!/usr/bin/env python
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV: start offset, duration and the gap to the
previous kernel's end, per launch.    python tools/timeline.py <kernel_trace.csv> [step-index-from-the-end]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("g4s::", "").replace("void ", "")[:44]
# a step starts at preprocess_fwd
starts = [i for i, r in enumerate(rows) if "preprocess_fwd" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, b = starts[-k - 1], starts[-k]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name(r)}")
    busy += e - s
    prev_end = max(prev_end, e)
print(f"step {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, sum of kernels {busy / 1e3:.1f} us, launches {b - a}")
