"""A/B of an environment switch of libks_hip.so on one of bench.py's workloads, in ONE process (the frames are made
once): python tools/ab_env.py [ENV_NAME] [values...]   default: KS_TEST_PRE 0 1
    AB_WORKLOAD = C2 (default) | C3 | C4-fast | C4-merged    AB_FRAMES distinct frames (48)    AB_REPS repetitions (2)
Prints ms/frame (median of the regions) for pipeline_frames = 4 and 0, and the per-stage HIP-event times.
Switches worth a run (opt-in variants checked on the functional model): KS_TEST_PRE 0 1 7,
KS_EMIT_STAGE 0 1 (AB_WORKLOAD=C4-fast)."""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "KS_TEST_PRE"
    values = sys.argv[2:] or ["0", "1"]
    import torch
    from kimera_semantics_amd import binding as B
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = dict(bench.WORKLOADS[os.environ.get("AB_WORKLOAD", "C2")])
    n_frames = int(os.environ.get("AB_FRAMES", "48"))
    frames = bench.make_frames(wl, list(range(n_frames)))
    ring = bench.FrameRing(frames, torch, dev)
    big = wl["w"] * wl["h"] > 640 * 480
    K, W, R = (10, 2, 3) if big else (40, 5, 3)
    for rep in range(int(os.environ.get("AB_REPS", "2"))):
        for v in values:
            os.environ[name] = v
            for pipe in (4, 0):
                m = bench.measure(B, torch, None, dev, wl, ring, W, K, R, pipe, (1 << 16) if big else (1 << 13), 1)
                ms = statistics.median(r["dt"] for r in m["regions"]) / K * 1e3
                upd = m["regions"][0]["updates"]
                sp = m["stage_prof"]
                stages = " ".join(f"{k}={sp['ms'][k] / max(1, sp['launches'][k]):.3f}" for k in sp["ms"] if sp["launches"][k])
                print(f"{name}={v} pipeline={pipe}: {ms:.4f} ms/frame  updates/region {upd}  stages[{stages}]", flush=True)


if __name__ == "__main__":
    main()
