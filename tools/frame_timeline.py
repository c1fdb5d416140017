#!/usr/bin/env python3
"""Per-dispatch timeline of the last complete frame in a rocprofv3 --kernel-trace CSV
(usage: frame_timeline.py <run_kernel_trace.csv> [first-kernel-substring])."""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    first = sys.argv[2] if len(sys.argv) > 2 else "k_points_"
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    a, b = idx[-2], idx[-1]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows[a:b]:
        name = r["Kernel_Name"].split("(")[0][-44:]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{name:46s} start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size_X', '?'):>8s} x {r.get('Workgroup_Size_X', '?')}")


if __name__ == "__main__":
    main()
