#!/usr/bin/env python3
"""Reads a bench.py JSON line (file or stdin) and prints its `*-switches` sub-records as a table: every opt-in switch of the
library against the defaults of the same context, measured in that run (bench.py: switch_records), and what the numbers
say about each default:   python tools/switch_report.py BENCH_r03.json
A switch "wins" when its median region is faster than BOTH default measurements (first and last of the A/B sequence) by
more than the spread of the default's own regions; a switch that must not change the result and did is flagged."""
import json
import sys


def load(path):
    txt = sys.stdin.read() if path == "-" else open(path).read()
    d = json.loads(txt)
    return d.get("parsed", d)     # (the driver wraps the line: {"parsed": {...}, ...})


def main():
    d = load(sys.argv[1] if len(sys.argv) > 1 else "-")
    recs = [s for s in d.get("secondary", []) if str(s.get("config", "")).endswith("-switches") or s.get("config") == "switches"]
    if not recs:
        print("no *-switches sub-records in this line")
        return
    for r in recs:
        if "error" in r and "default" not in r:
            print(f"{r['config']}: {r['error']}")
            continue
        base = r["default"]
        regs = base["ms_per_frame_all_regions"]
        spread = max(regs) - min(regs)
        again = r.get("default_again_ms_per_frame", base["ms_per_frame"])
        slow_default = max(base["ms_per_frame"], again)
        fast_default = min(base["ms_per_frame"], again)
        print(f"\n{r['config']}  (pipeline_frames {r['pipeline_frames']}, {r['regions']} regions of {r['steps_per_region']} frames)")
        print(f"  default                       {base['ms_per_frame']:.4f} ms/frame  (again at the end: {again:.4f}; spread of its regions {spread:.4f})")
        for v in r["variants"]:
            if "error" in v and "ms_per_frame" not in v:
                print(f"  {v['switch']:<29} ERROR {v['error']}")
                continue
            ms = v["ms_per_frame"]
            verdict = "faster than the default" if ms < fast_default - spread else "slower than the default" if ms > slow_default + spread else "within the noise"
            eq = v.get("map_equals_default")
            note = ""
            if v.get("result_must_equal_default") and eq is False:
                note = "   *** RESULT DIFFERS FROM THE DEFAULT'S ***"
            elif eq is not None:
                note = f"   map {'==' if eq else '!='} default"
            stages = " ".join(f"{k}={x:.3f}" for k, x in v.get("stage_ms", {}).items())
            print(f"  {v['switch']:<29} {ms:.4f} ms/frame  x{ms / base['ms_per_frame']:.3f}  {verdict}{note}")
            if "--stages" in sys.argv:
                print(f"      stages[{stages}]")


if __name__ == "__main__":
    main()
