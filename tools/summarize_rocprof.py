#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats` output directory into the text summary kept under
profiles/: per-kernel calls / total / average, plus (from the trace) how much of the wall-clock
span had 1, 2, 3... kernels executing at once (the frame pipeline overlaps three streams).
usage: summarize_rocprof.py <dir with run_kernel_stats.csv, run_kernel_trace.csv> "<command line>" > profiles/xxx.txt
"""
import csv
import sys


def main():
    d, cmd = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(f"{d}/run_kernel_stats.csv")))
    print(f"# rocprofv3 --kernel-trace --stats -- {cmd}  (MI355X)")
    print(f"{'kernel':100s} {'calls':>8s} {'total_us':>14s} {'avg_us':>12s} {'pct':>8s}")
    for r in rows:
        print(f"{r['Name'][:100]:100s} {int(r['Calls']):8d} {int(r['TotalDurationNs']) / 1e3:14.1f} "
              f"{float(r['AverageNs']) / 1e3:12.2f} {float(r['Percentage']):8.2f}")
    tr = list(csv.DictReader(open(f"{d}/run_kernel_trace.csv")))
    ev = []
    for r in tr:
        ev.append((int(r["Start_Timestamp"]), 1))
        ev.append((int(r["End_Timestamp"]), -1))
    ev.sort()
    depth, last, hist = 0, ev[0][0], {}
    for t, dlt in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        last = t
        depth += dlt
    # frame markers: the first kernel of a frame (k_points_fast / k_points_merged).  bench.py --steps 20 --warmup 2:
    # 60 untimed frames (two turns of the ring of 20: >= PRIME + warm-up, and one untimed region), then the timed regions: frames [60, 160) = five of them
    first = sorted(int(r["Start_Timestamp"]) for r in tr if "k_points_" in r["Kernel_Name"])
    hist = {}
    B0, B1 = 60, 160
    if len(first) >= B1:
        per = (first[B1 - 1] - first[B0]) / (B1 - 1 - B0) / 1e3
        print(f"\n# frame period inside the timed regions (k_points start to start, frames {B0}..{B1 - 1}), with tracing on: {per:.1f} us")
        lo, hi = first[B0], first[B1 - 1]
        depth, last = 0, lo
        for t, dlt in ev:
            if t > hi:
                break
            if t >= lo:
                hist[depth] = hist.get(depth, 0) + (t - max(last, lo))
            last = t
            depth += dlt
    ap = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                for r in tr if "k_apply<" in r["Kernel_Name"] and "k_apply_long" not in r["Kernel_Name"])
    if len(ap) >= B1:
        timed = [d for _, d in ap[B0:B1]]
        print(f"# k_apply average duration: {sum(timed) / len(timed) / 1e3:.2f} us over the 100 timed (pipelined, overlapped) launches "
              f"(bench.py's roofline.k_apply.avg_launch_ms of the same run is in the log line below)")
    tot = sum(hist.values()) or 1
    print("# kernels executing concurrently (share of the steady-state span): " +
          ", ".join(f"{k}: {100.0 * v / tot:.1f}%" for k, v in sorted(hist.items())))
    if len(sys.argv) > 3:
        import json
        for line in open(sys.argv[3]):
            if line.startswith("{"):
                j = json.loads(line)
                rf = j["roofline"]
                print("# bench.py line of this traced run: value", j["value"], j["unit"], "ms_per_step", j["ms_per_step"],
                      "whole-frame frac", rf["frac"], "k_apply", json.dumps(rf["k_apply"]))


if __name__ == "__main__":
    main()
