#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of tools/exact_tune.py: do the stage-B chains of consecutive frames overlap?
usage: exact_overlap.py <dir with *kernel_trace.csv>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# frames: k_eo2_begin .. k_publish on the same queue
begins = [r for r in rows if "k_eo2_begin" in r["Kernel_Name"]]
pubs = [r for r in rows if "k_publish" in r["Kernel_Name"]]
tests = [r for r in rows if "k_test" in r["Kernel_Name"]]
print("frames", len(begins), "queues used by k_eo2_begin:", sorted({r.get("Queue_Id", "?") for r in begins}))
iv = []
for b in begins[-24:]:
    q = b.get("Queue_Id")
    s = int(b["Start_Timestamp"])
    e = min((int(p["End_Timestamp"]) for p in pubs if int(p["Start_Timestamp"]) > s and p.get("Queue_Id") == q), default=None)
    if e:
        iv.append((s, e, q))
for s, e, q in iv:
    over = sum(1 for s2, e2, _ in iv if s2 < e and e2 > s) - 1
    print(f"  fix point + emission of a frame: start {(s - t0) / 1e3:10.1f} us  length {(e - s) / 1e3:8.1f} us  queue {q}  overlapping frames {over}")
# concurrency histogram over the last third of the trace
ev = []
cut = int(rows[len(rows) * 2 // 3]["Start_Timestamp"])
for r in rows:
    if int(r["Start_Timestamp"]) >= cut:
        ev.append((int(r["Start_Timestamp"]), 1))
        ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
cur, last, hist = 0, ev[0][0], {}
for t, d in ev:
    hist[cur] = hist.get(cur, 0) + (t - last)
    cur += d
    last = t
tot = sum(hist.values())
print("kernels executing concurrently (share of time):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
