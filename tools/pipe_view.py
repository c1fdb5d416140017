#!/usr/bin/env python3
"""Per-queue picture of a pipelined run from a rocprofv3 --kernel-trace CSV: busy share of every hardware queue over the
last 100 frames, then the kernels of a window of `span` us in the middle of them, one line per dispatch with its queue.
usage: pipe_view.py <run_kernel_trace.csv> [span_us=3000] [first-kernel=k_points_]"""
import collections
import csv
import sys


def short(n):
    return n.split("(")[0].replace("void ", "").replace("ksk::", "").replace("ksrs::", "")[:34]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    span = float(sys.argv[2]) if len(sys.argv) > 2 else 3000.0
    first = sys.argv[3] if len(sys.argv) > 3 else "k_points_"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    sp = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    seg = rows[sp[-101]:sp[-1]]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    q = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    for r in seg:
        k = r["Queue_Id"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        q[k][0] += 1
        q[k][1] += d
        q[k][2][short(r["Kernel_Name"])] += d
    print(f"# frame period over the last 100 frames: {(t1 - t0) / 1e5:.1f} us")
    names = {}
    for i, (k, v) in enumerate(sorted(q.items(), key=lambda kv: -kv[1][1])):
        names[k] = chr(ord('A') + i)
        top = ", ".join(f"{n} {d / 100:.0f}" for n, d in v[2].most_common(6))
        print(f"# queue {names[k]}: {v[0] / 100:.1f} dispatches/frame, busy {100 * v[1] * 1e3 / (t1 - t0):.0f} %, {v[1] / 100:.0f} us/frame: {top}")
    mid = (t0 + t1) // 2
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s < mid or s > mid + span * 1e3:
            continue
        col = ord(names[r["Queue_Id"]]) - ord('A')
        print(f"{(s - mid) / 1e3:9.1f} +{(e - s) / 1e3:7.1f}  " + "  " * col * 6 + f"{names[r['Queue_Id']]}:{short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
