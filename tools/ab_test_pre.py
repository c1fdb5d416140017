"""A/B of an environment switch of libks_hip.so on bench.py's headline workload, in ONE process (the frames are made
once): python tools/ab_test_pre.py [ENV_NAME] [values...]   default: KS_TEST_PRE 0 1
Prints ms/frame (median of the regions) for pipeline_frames = 4 and 0, and the per-stage HIP-event times."""
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
    wl = dict(bench.WORKLOADS["C2"], w=640, h=480, method="fast")
    n_frames = int(os.environ.get("AB_FRAMES", "48"))
    frames = bench.make_frames(wl, list(range(n_frames)))
    ring = bench.FrameRing(frames, torch, dev)
    K, W, R = 40, 5, 3
    for rep in range(int(os.environ.get("AB_REPS", "2"))):
        for v in values:
            os.environ[name] = v
            for pipe in (4, 0):
                m = bench.measure(B, torch, None, dev, wl, ring, W, K, R, pipe, 1 << 13, 1)
                ms = statistics.median(r["dt"] for r in m["regions"]) / K * 1e3
                upd = m["regions"][0]["updates"]
                sp = m["stage_prof"]
                stages = " ".join(f"{k}={sp['ms'][k] / max(1, sp['launches'][k]):.3f}" for k in sp["ms"] if sp["launches"][k])
                print(f"{name}={v} pipeline={pipe}: {ms:.4f} ms/frame  updates/region {upd}  stages[{stages}]", flush=True)


if __name__ == "__main__":
    main()
