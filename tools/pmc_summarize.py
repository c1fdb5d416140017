#!/usr/bin/env python3
"""Summarises tools/pmc_c4.sh: per kernel, average duration (timing pass), FETCH_SIZE / WRITE_SIZE per launch (two
PMC passes), the gfx950 corrections calibrated on k_export_tiles (known 65536 B read + written per tile), real
HBM-side GB/s, and for the voxel-update kernels the algorithmic 208 B / update rate next to it.
usage: pmc_summarize.py <gpurun_out/pmc_r02> -> text on stdout, JSON in <dir>/summary.json"""
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


def counters(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    per = collections.OrderedDict()
    if not f:
        return per
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == counter:
            per.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return per


def durations(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    per = collections.OrderedDict()
    if not f:
        return per
    for r in csv.DictReader(open(f[0])):
        per.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return per


def main():
    root = sys.argv[1]
    out = {}
    # calibration: bytes per KB reported
    log = open(os.path.join(root, "calib_FETCH_SIZE.log")).read()
    m = re.search(r"calib tiles (\d+) bytes_each_way (\d+)", log)
    known = float(m.group(2)) if m else None
    cf = counters(os.path.join(root, "calib_FETCH_SIZE"), "FETCH_SIZE").get("k_export_tiles", [])
    cw = counters(os.path.join(root, "calib_WRITE_SIZE"), "WRITE_SIZE").get("k_export_tiles", [])
    corr_f = known / (sum(cf) / len(cf) * 1024) if known and cf else 2.0
    corr_w = known / (sum(cw) / len(cw) * 1024) if known and cw else 1.0
    out["calibration"] = {"kernel": "k_export_tiles (65536 B read + 65536 B written per tile, 16 B per lane, coalesced)",
                          "known_bytes_each_way": known, "FETCH_SIZE_KB_reported": sum(cf) / len(cf) if cf else None,
                          "WRITE_SIZE_KB_reported": sum(cw) / len(cw) if cw else None,
                          "fetch_correction": round(corr_f, 4), "write_correction": round(corr_w, 4)}
    print("# calibration:", json.dumps(out["calibration"]))
    for wl in ("C4-fast", "C4-merged"):
        fe = counters(os.path.join(root, wl + "_FETCH_SIZE"), "FETCH_SIZE")
        wr = counters(os.path.join(root, wl + "_WRITE_SIZE"), "WRITE_SIZE")
        du = durations(os.path.join(root, wl + "_time"))
        log = open(os.path.join(root, wl + "_time.log")).read()
        upd = [int(x) for x in re.findall(r"updates (\d+)", log)]
        rows = {}
        print(f"# {wl}: frames {len(upd)}, updates per frame {upd}")
        print(f"{'kernel':44s} {'calls':>6s} {'avg_us':>10s} {'FETCH_MB':>10s} {'WRITE_MB':>10s} {'real_GB/s':>10s} {'frac_8TB/s':>10s}")
        for k in sorted(du, key=lambda k: -sum(du[k])):
            d = du[k]
            f_kb = sum(fe.get(k, [0])) / max(1, len(fe.get(k, [0])))
            w_kb = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
            avg_us = sum(d) / len(d)
            real = (f_kb * corr_f + w_kb * corr_w) * 1024
            gbs = real / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
            rows[k] = {"calls": len(d), "avg_us": round(avg_us, 2), "fetch_bytes": int(f_kb * corr_f * 1024), "write_bytes": int(w_kb * corr_w * 1024),
                       "real_GBs": round(gbs, 1), "frac_of_8TBs": round(gbs / 8000.0, 4)}
            if sum(d) > 0.01 * sum(sum(v) for v in du.values()):
                print(f"{k[:44]:44s} {len(d):6d} {avg_us:10.1f} {f_kb * corr_f / 1024:10.1f} {w_kb * corr_w / 1024:10.1f} {gbs:10.1f} {gbs / 8000.0:10.4f}")
        # algorithmic rate of the voxel update (208 B per update) over k_apply + k_apply_long of the same frames
        ka = [k for k in du if k.startswith("k_apply<")]
        kl = [k for k in du if k.startswith("k_apply_long")]
        if ka and upd:
            t_apply = sum(du[ka[0]]) / len(du[ka[0]])
            t_long = sum(du[kl[0]]) / len(du[kl[0]]) if kl else 0.0
            mean_upd = sum(upd) / len(upd)
            alg = 208.0 * mean_upd
            rows["voxel_update_algorithmic"] = {
                "updates_per_frame": mean_upd, "algorithmic_bytes": alg,
                "k_apply_only_GBs": round(alg / (t_apply * 1e-6) / 1e9, 1), "k_apply_only_frac": round(alg / (t_apply * 1e-6) / 1e9 / 8000.0, 4),
                "apply_and_long_serial_GBs": round(alg / ((t_apply + t_long) * 1e-6) / 1e9, 1),
                "note": "k_apply and k_apply_long run side by side on two streams; k_apply_only credits k_apply with every update"}
            print("# voxel update, algorithmic:", json.dumps(rows["voxel_update_algorithmic"]))
        out[wl] = rows
    json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
