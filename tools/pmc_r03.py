#!/usr/bin/env python3
"""Summarises tools/profile_r03.sh's PMC passes.  FETCH_SIZE / WRITE_SIZE are collected in SEPARATE rocprofv3 runs
(they do not fit one TCC pass: /opt/skills/guides/MI355X_MICROARCH.md, HBM/rocprofv3 section); both report KB per
dispatch; on gfx950 FETCH_SIZE reads high by a constant factor which is calibrated here on k_export_tiles (it reads and
writes exactly 65536 B per tile).  Per workload and kernel: calls, average duration (the timing pass / the PMC
pass's own trace), corrected HBM bytes per launch, real GB/s.
For the bench commands (fast = C2, merged = C3) only the launches of the timed regions are averaged and the result for
k_apply goes to profiles-ready JSON (r03_pmc_c2.json / r03_pmc_c3.json: bench.py's roofline.traffic reads them).
usage: pmc_r03.py <gpurun_out/prof_r03>"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("ksk::", "").replace("ksrs::", "").replace("void ", "")
    return n.split("(")[0]


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return []
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def trace(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not f:
        return []
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return rows


def per_kernel(rows, key="Counter_Value", lo=None, hi=None):
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(short(r["Kernel_Name"]), []).append(float(r[key]))
    if lo is not None:   # keep launches [lo, hi) of kernels launched once per frame; scale others proportionally
        out = collections.OrderedDict()
        n_frames = len(per.get("k_publish", per.get("k_scan_local", [])))
        for k, v in per.items():
            per_frame = len(v) / max(1, n_frames)
            a, b = int(lo * per_frame), int(hi * per_frame)
            out[k] = v[a:b] if b > a else v
        return out
    return per


TAG = os.environ.get("PMC_TAG", "r03")
SCRIPT = os.environ.get("PMC_SCRIPT", "profile_" + TAG + ".sh")


def main():
    root = sys.argv[1]
    out = {}
    log = open(os.path.join(root, "pmc_calib_FETCH_SIZE.log")).read()
    m = re.search(r"calib tiles (\d+) bytes_each_way (\d+)", log)
    known = float(m.group(2)) if m else None
    cf = per_kernel(load(os.path.join(root, "pmc_calib_FETCH_SIZE"), "FETCH_SIZE")).get("k_export_tiles", [])
    cw = per_kernel(load(os.path.join(root, "pmc_calib_WRITE_SIZE"), "WRITE_SIZE")).get("k_export_tiles", [])
    corr_f = known / (sum(cf) / len(cf) * 1024) if known and cf else 0.5
    corr_w = known / (sum(cw) / len(cw) * 1024) if known and cw else 1.0
    out["calibration"] = {"kernel": "k_export_tiles (65536 B read + 65536 B written per tile, 16 B per lane, coalesced)",
                          "known_bytes_each_way": known, "fetch_correction": round(corr_f, 4), "write_correction": round(corr_w, 4)}
    print("# calibration:", json.dumps(out["calibration"]))
    for wl, tag, timed in (("fast", "c2", True), ("merged", "c3", True), ("C4-fast", "c4_fast", False), ("C4-merged", "c4_merged", False)):
        fe_rows = load(os.path.join(root, f"pmc_{wl}_FETCH_SIZE"), "FETCH_SIZE")
        wr_rows = load(os.path.join(root, f"pmc_{wl}_WRITE_SIZE"), "WRITE_SIZE")
        if not fe_rows or not wr_rows:
            print(f"# {wl}: no counters")
            continue
        # bench commands (--steps 20 --warmup 2): 60 untimed frames, then the timed regions: frames [60, 160) = five of them
        lo, hi = (60, 160) if timed else (None, None)
        fe, wr = per_kernel(fe_rows, lo=lo, hi=hi), per_kernel(wr_rows, lo=lo, hi=hi)
        tdir = os.path.join(root, wl if timed else f"time_{wl}")
        du = collections.OrderedDict()
        for r in trace(tdir):
            du.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        rows = {}
        print(f"# {wl}")
        print(f"{'kernel':44s} {'calls':>6s} {'avg_us':>9s} {'FETCH_MB':>9s} {'WRITE_MB':>9s} {'real_GB/s':>10s} {'frac_8TB/s':>10s}")
        tot_t = sum(sum(v) for v in du.values()) or 1.0
        for k in sorted(fe, key=lambda k: -sum(du.get(k, [0]))):
            f_b = sum(fe[k]) / len(fe[k]) * 1024 * corr_f
            w_b = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0]))) * 1024 * corr_w
            d = du.get(k, [])
            avg_us = sum(d) / len(d) if d else 0.0
            gbs = (f_b + w_b) / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
            rows[k] = {"calls_averaged": len(fe[k]), "avg_us": round(avg_us, 2), "fetch_bytes_per_launch": int(f_b), "write_bytes_per_launch": int(w_b),
                       "real_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / 8000.0, 4)}
            if sum(d) > 0.01 * tot_t:
                print(f"{k[:44]:44s} {len(fe[k]):6d} {avg_us:9.1f} {f_b / 1e6:9.2f} {w_b / 1e6:9.2f} {gbs:10.1f} {gbs / 8000.0:10.4f}")
        out[wl] = rows
        # whole frame (round 6): every kernel's bytes per launch x its launches per frame (a kernel launched once per frame
        # defines the frame count of the averaged window)
        kp = [k for k in fe if k.startswith("k_points_")]   # one launch per frame (k_publish is one per BATCH of frames in pipelined contexts)
        n_frames_avg = max(1, len(fe[kp[0]]) if kp else len(fe.get("k_publish", [0])))
        whole = sum((sum(fe[k]) * corr_f + sum(wr.get(k, [0])) * corr_w) * 1024 for k in fe) / n_frames_avg
        print(f"# {wl}: whole frame, all kernels: {whole / 1e6:.1f} MB of HBM traffic per frame (FETCH x {corr_f:.3f} + WRITE, averaged over {n_frames_avg} frames)")
        ka = [k for k in rows if k.startswith("k_apply<")] or [k for k in rows if k.startswith("k_apply_runs<")]
        if ka:
            r = rows[ka[0]]
            json.dump({"whole_frame_hbm_bytes_per_frame": int(whole), "frames_averaged": n_frames_avg, "update_kernel": ka[0],
                       "k_apply_hbm_bytes_per_launch": r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"],
                       "fetch_bytes_per_launch": r["fetch_bytes_per_launch"], "write_bytes_per_launch": r["write_bytes_per_launch"],
                       "launches_averaged": r["calls_averaged"], "calibration": out["calibration"],
                       "source": f"profiles/{TAG}_pmc_{tag}.json <- tools/{SCRIPT}: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --kernel-trace -- "
                                 + ("python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count" + (" --method merged" if wl == "merged" else "")
                                    if timed else f"python tools/probe.py {wl} 3")},
                      open(os.path.join(root, f"{TAG}_pmc_{tag}.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
