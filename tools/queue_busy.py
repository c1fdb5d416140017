#!/usr/bin/env python3
"""Per hardware queue: how busy it was over the last frames of a tools/pipe_trace.sh trace, and the kernels that kept it busy.
usage: queue_busy.py <run_kernel_trace.csv> [frames]"""
import csv
import sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_points_" in r["Kernel_Name"]]
start, end = idx[-2 * nf], idx[-nf]
t0 = int(rows[start]["Start_Timestamp"])
T = (int(rows[end]["Start_Timestamp"]) - t0) / 1e3
print(f"# {nf} frames in {T:.0f} us = {T / nf:.0f} us/frame (under the tracer)")
busy, byk, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for r in rows[start:end]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy[r["Queue_Id"]] += d
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[-44:]
    byk[(r["Queue_Id"], n)] += d
    cnt[(r["Queue_Id"], n)] += 1
for q, b in sorted(busy.items()):
    print(f"queue {q}: kernels executing {b / nf:.0f} us/frame = {100 * b / T:.0f} % of the time")
for k, v in sorted(byk.items(), key=lambda kv: -kv[1])[:24]:
    print(f"  q{k[0]}  {v / nf:8.1f} us/frame  {cnt[k] / nf:5.1f} launches/frame  {k[1]}")
