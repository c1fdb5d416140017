"""A/B of KS_XLONG (1, the default: the runs of more than 1024 updates on a list and a stream of their own, four waves per run —
k_apply_xlong; 0: one list, k_apply_long) on the `merged` workloads: C3 (640x480) and, with --c4, C4-merged (1280x720).
Same map in every variant (tests/test_parity_gpu.py::test_runs_next_to_the_sensor_on_their_own_list_exact); this tool only
reads the clock.  One JSON line per variant; run it under `rocprofv3 --kernel-trace --stats` for the kernel durations.

    python tools/xlong_ab.py [--c4] [--steps K] [--repeats R]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c4", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--variants", default="1,0")
    args = ap.parse_args()
    import torch
    import bench
    from kimera_semantics_amd import binding as B
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    plans = [("C3", bench.WORKLOADS["C3"], 24, args.steps, 1 << 13)]
    if args.c4:
        plans.append(("C4-merged", bench.WORKLOADS["C4-merged"], 12, max(6, args.steps // 2), 1 << 16))
    for name, wl, n_frames, K, tiles in plans:
        ring = bench.FrameRing(bench.make_frames(wl, range(n_frames)), torch, dev)
        for pf in args.variants.split(","):
            os.environ["KS_DEBUG"] = "1"
            os.environ["KS_XLONG"] = pf
            m = bench.measure(B, torch, None, dev, wl, ring, 2, K, args.repeats, 8, tiles, 1, prime=8 if name != "C3" else None)
            ms = sorted(r["dt"] / K * 1e3 for r in m["regions"])
            sp = m["stage_prof"]
            st = {k: round(v / max(1, sp["frames"]), 4) for k, v in sp["ms"].items()}
            print(json.dumps({"config": name, "KS_XLONG": int(pf), "ms_per_frame_median": round(ms[len(ms) // 2], 4),
                              "ms_per_frame_all": [round(x, 4) for x in ms], "updates_per_frame": m["regions"][0]["updates"] // K,
                              "stage_ms": st}), flush=True)
        del ring
        torch.cuda.empty_cache()
    os.environ.pop("KS_XLONG", None)


if __name__ == "__main__":
    main()
