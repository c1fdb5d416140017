#!/usr/bin/env python3
"""Summarises tools/sq_pass.sh: per kernel (last frame's launches averaged), the SQ counters of the three passes and what they
say — waves, wave-cycles split into parked (SQ_WAIT_ANY: s_waitcnt / barrier), issue-stalled (SQ_WAIT_INST_ANY) and issuing
(SQ_ACTIVE_INST_ANY), instruction mix per wave, VALU lane utilisation (SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU),
mean occupancy (SQ_LEVEL_WAVES / SQ_BUSY_CYCLES... reported raw).  SQ cycle counters count quad-cycles
(/opt/skills/guides/MI355X_MICROARCH.md).  usage: sq_summarize.py <dir> <workload>"""
import collections
import csv
import glob
import os
import sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("ksk::", "").replace("ksrs::", "").replace("void ", "")
    return n.split("(")[0]


def main():
    root = sys.argv[1]
    per = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values per dispatch
    for p in sorted(glob.glob(os.path.join(root, "p*"))):
        if not os.path.isdir(p):
            continue
        f = glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
        if not f:
            print("# no counters in", p)
            continue
        for r in csv.DictReader(open(f[0])):
            per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    f = glob.glob(os.path.join(root, "time", "**", "*kernel_trace.csv"), recursive=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"# workload {sys.argv[2] if len(sys.argv) > 2 else ''}: per kernel, mean over its launches (us from the un-instrumented timing pass)")
    want = ["k_xl", "k_apply", "k_apply_xlong", "k_apply_long", "k_emit_lane", "k_rs_pass", "k_rs_hist", "k_find_long", "k_apply_runs", "k_list_runs", "k_eo2_sweep", "k_test"]
    for k in sorted(per, key=lambda k: -sum(dur.get(k, [0]))):
        if not any(k.startswith(w) for w in want):
            continue
        c = {n: sum(v) / len(v) for n, v in per[k].items()}
        us = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
        g = lambda n: c.get(n, float("nan"))
        waves = g("SQ_WAVES")
        wc = g("SQ_WAVE_CYCLES")
        print(f"\n{k}: launches {len(next(iter(per[k].values())))}, {us:.1f} us, waves {waves:.0f}")
        print(f"  wave-cycles (quad): {wc:.3g}  = parked {g('SQ_WAIT_ANY') / wc:.2f} + issue-stalled {g('SQ_WAIT_INST_ANY') / wc:.2f} + issuing {g('SQ_ACTIVE_INST_ANY') / wc:.2f}"
              f"   (VALU {g('SQ_ACTIVE_INST_VALU') / wc:.2f}, LDS {g('SQ_ACTIVE_INST_LDS') / wc:.2f}, VMEM {g('SQ_ACTIVE_INST_VMEM') / wc:.2f}, scalar {g('SQ_ACTIVE_INST_SCA') / wc:.2f}, misc {g('SQ_ACTIVE_INST_MISC') / wc:.2f}; LDS-issue-stall {g('SQ_WAIT_INST_LDS') / wc:.2f})")
        print(f"  per wave: VALU {g('SQ_INSTS_VALU') / waves:.0f} (trans {g('SQ_INSTS_VALU_TRANS_F32') / waves:.1f}), SALU {g('SQ_INSTS_SALU') / waves:.0f}, LDS {g('SQ_INSTS_LDS') / waves:.0f}, "
              f"VMEM rd {g('SQ_INSTS_VMEM_RD') / waves:.1f} wr {g('SQ_INSTS_VMEM_WR') / waves:.1f}, SMEM {g('SQ_INSTS_SMEM') / waves:.1f}; wave lifetime {4 * wc / waves:.0f} cycles")
        busy = g("SQ_BUSY_CYCLES")
        print(f"  SQ_BUSY_CYCLES {busy:.3g}, GRBM_GUI_ACTIVE {g('GRBM_GUI_ACTIVE'):.3g}; mean waves in flight (SQ_LEVEL_WAVES / SQ_BUSY_CYCLES) {g('SQ_LEVEL_WAVES') / busy:.2f}; "
              f"VALU lane utilisation {g('SQ_THREAD_CYCLES_VALU') / 64.0 / max(1e-9, g('SQ_ACTIVE_INST_VALU')):.2f}; LDS bank-conflict cycles / LDS active {g('SQ_LDS_BANK_CONFLICT') / max(1e-9, g('SQ_LDS_IDX_ACTIVE')):.2f}; "
              f"VMEM in flight (SQ_INST_LEVEL_VMEM / SQ_BUSY_CYCLES) {g('SQ_INST_LEVEL_VMEM') / busy:.1f}")
        print("  raw:", {n: float(f"{v:.4g}") for n, v in sorted(c.items())})


if __name__ == "__main__":
    main()
