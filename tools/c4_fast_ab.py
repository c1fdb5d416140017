"""C4-`fast` in the default mode (the reference's serial result) on the device: ms per frame, fix-point rounds and fallbacks, for
the launch-shape knobs that do not change the map (KS_EXACT_SWEEPS / KS_EXACT_SWEEP_ORDER from the environment; KS_EXACT_TRACE=1
prints every frame's sweeps).
usage: python tools/c4_fast_ab.py [frames=8] [pipeline=0]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def main():
    import torch
    import bench
    from kimera_semantics_amd import binding as B
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS["C4-fast"]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    pipe = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ring = bench.FrameRing(bench.make_frames(wl, range(n)), torch, dev)
    integ = B.HipIntegrator(B.default_config(device_id=0, max_tiles=1 << 16, max_points=wl["w"] * wl["h"], pipeline_frames=pipe, **bench.integ_cfg(wl)))

    def turn():
        upd = 0
        for i in range(n):
            x, c, l = ring.dev(i)
            upd += integ.integrate_device(ring.host(i).T_G_C, x.data_ptr(), c.data_ptr(), l.data_ptr(), x.shape[0]).n_voxel_updates
        upd += integ.flush().n_voxel_updates
        integ.synchronize()
        return upd
    turn()   # buffers grow here
    s0 = integ.early_out_stats()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        upd = turn()
        ts.append((time.perf_counter() - t0) * 1e3 / n)
    s1 = integ.early_out_stats()
    print({k: os.environ.get(k) for k in ("KS_EXACT_SWEEPS", "KS_EXACT_SWEEP_ORDER", "KS_EXACT_SEED_GROWTH") if os.environ.get(k)},
          "pipeline", pipe, "ms/frame", [round(t, 2) for t in ts], "updates/frame", upd // n,
          "rounds/frame", (s1["rounds"] - s0["rounds"]) / (3 * n), "fallbacks", s1["fallbacks"] - s0["fallbacks"], "first turn fallbacks", s0["fallbacks"],
          "event_driven", s1["event_driven"], flush=True)
    integ.close()


if __name__ == "__main__":
    main()
