#!/usr/bin/env python3
"""Times the exact early-out mode (event-driven fix point) at a bench workload for a list of settings, on the GPU:
   python tools/exact_tune.py [C2|C4-fast] "pipe=8" "pipe=8,KS_EXACT_BULK_ROUNDS=10" "pipe=4,KS_EXACT_SEED_GROWTH=64" "pipe=0,growth=32" ...
growth=<n>: the ordered-phase schedule alone (early_out_phase_growth = n), for comparison."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from kimera_semantics_amd import binding as B  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    wl = bench.WORKLOADS[name]
    big = wl["w"] > 640
    n_frames = 10 if big else 40
    frames = bench.make_frames(wl, range(n_frames))
    dev = torch.device("cuda", 0)
    ring = bench.FrameRing(frames, torch, dev)
    for spec in sys.argv[2:] or ["pipe=8"]:
        kv = dict(x.split("=") for x in spec.split(","))
        pipe = int(kv.pop("pipe", "8"))
        growth = int(kv.pop("growth", "0"))
        K = int(kv.pop("frames", "30" if big else "120"))
        saved = {k: os.environ.get(k) for k in kv}
        os.environ.update(kv)
        try:
            cfg = B.default_config(device_id=0, max_tiles=(1 << 16) if big else (1 << 13), max_points=wl["w"] * wl["h"], pipeline_frames=pipe,
                                   **bench.integ_cfg(wl, early_out_phase_growth=growth))
            h = B.HipIntegrator(cfg)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v

        def step(i):
            x, c, l = ring.dev(i)
            return h.integrate_device(ring.host(i).T_G_C, x.data_ptr(), c.data_ptr(), l.data_ptr(), x.shape[0])
        for i in range(n_frames):
            step(i)
        h.flush()
        h.synchronize()
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            upd = 0
            for i in range(K):
                upd += step(n_frames + rep * K + i).n_voxel_updates
            upd += h.flush().n_voxel_updates
            h.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        st = h.early_out_stats()
        print(f"{name} {spec:50s} {best / K * 1e3:8.4f} ms/frame  {upd / K / 1e3:8.1f} k updates/frame  rounds/frame {st['rounds'] / max(1, st['frames']):5.1f} "
              f"fallbacks {st['fallbacks']} event_driven {st['event_driven']} pipelined {st['pipelined']}", flush=True)
        h.close()


if __name__ == "__main__":
    main()
