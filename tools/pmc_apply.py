#!/usr/bin/env python3
"""HBM traffic of k_apply from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE must be
collected in separate runs; /opt/skills/guides/MI355X_MICROARCH.md, HBM/rocprofv3 section).
usage: pmc_apply.py <dir of the FETCH_SIZE run> <dir of the WRITE_SIZE run> "<bench command>"
writes profiles/pmc_apply.json + per-kernel text summaries profiles/r01_pmc_{FETCH,WRITE}_SIZE.txt.
Only launches of the timed, pipelined context (the first steps+warmup k_apply launches of the
bench command) are averaged, the same launches bench.py's roofline.achieved is computed on."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("ksk::", "").replace("void ", "")
    return n.split("(")[0]


def summary(rows, counter, cmd, path):
    per = collections.OrderedDict()
    for r in rows:
        per.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    with open(path, "w") as fh:
        fh.write(f"# rocprofv3 --pmc {counter} --kernel-trace -- {cmd} (MI355X; counter unit: KB per dispatch)\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            fh.write(f"{k[:70]:70s} calls {len(v):5d} avg_KB {sum(v) / len(v):12.1f}\n")


def main():
    dfetch, dwrite, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    n_timed = int(sys.argv[4]) if len(sys.argv) > 4 else None
    out = {}
    for counter, d in (("FETCH_SIZE", dfetch), ("WRITE_SIZE", dwrite)):
        rows = load(d, counter)
        summary(rows, counter, cmd, os.path.join(ROOT, "profiles", f"r01_pmc_{counter}.txt"))
        ap = [float(r["Counter_Value"]) for r in rows if "k_apply<" in r["Kernel_Name"] and "k_apply_long" not in r["Kernel_Name"]]
        if n_timed:
            ap = ap[:n_timed]
        out[counter] = (sum(ap) / len(ap), len(ap))
    fetch_kb, n = out["FETCH_SIZE"]
    write_kb, _ = out["WRITE_SIZE"]
    res = {
        "kernel": "k_apply<KS_COLOR_MODE_SEMANTIC>",
        "workload": f"{cmd}: the {n} k_apply launches of the pipelined context (warm-up + timed frames)",
        "FETCH_SIZE_KB_per_launch": round(fetch_kb, 1),
        "WRITE_SIZE_KB_per_launch": round(write_kb, 1),
        "correction": "gfx950: FETCH_SIZE under-reports 16-B/lane coalesced reads by 2x (MI355X_MICROARCH.md HBM "
                      "section) -> doubled; WRITE_SIZE taken as is; separate --pmc passes",
        "hbm_bytes_per_launch": int(round((2.0 * fetch_kb + write_kb) * 1024)),
    }
    json.dump(res, open(os.path.join(ROOT, "profiles", "pmc_apply.json"), "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
