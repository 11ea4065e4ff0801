#!/usr/bin/env python3
"""Per-step kernel timeline from a rocprofv3 --kernel-trace CSV: tools/timeline.py <kernel_trace.csv> [min_us] [until_us]
(one middle step, delimited by k_ortho9d; start/end/duration in us, queue id, kernel name)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 40.0
until = float(sys.argv[3]) if len(sys.argv) > 3 else 1e12
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_ortho9d" in r["Kernel_Name"]] or [i for i, r in enumerate(rows) if "k_heads_l23" in r["Kernel_Name"]]
k = len(ends) // 2
t0 = int(rows[ends[k]]["End_Timestamp"])
step = rows[ends[k] + 1:ends[k + 1] + 1]
print("kernels %d, span %.3f ms" % (len(step), (int(step[-1]["End_Timestamp"]) - t0) / 1e6))
for r in step:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    if e - s >= min_us and s < until:
        print("%9.1f %9.1f %8.1f q%s %s" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"][:64]))
