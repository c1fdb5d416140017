"""Wall time of every integrate call of a pipelined stream (C2, default mode): which calls wait, and for how long.
KS_HOST_PROF=1 adds the library's own account of the host time spent enqueueing stage A / B / T."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("KS_DEBUG", "1")
os.environ.setdefault("KS_HOST_PROF", "1")


def main():
    import torch
    import bench
    from kimera_semantics_amd import binding as B
    dev = torch.device("cuda:0")
    wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "C2"]
    pipe = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    ring = bench.FrameRing(bench.make_frames(wl, range(40)), torch, dev)
    cfg = B.default_config(device_id=0, max_tiles=1 << 13, max_points=max(f.xyz.shape[0] for f in ring.frames),
                           pipeline_frames=pipe, **bench.integ_cfg(wl))
    integ = B.HipIntegrator(cfg)
    for i in range(40):
        x, c, l = ring.dev(i)
        integ.integrate_device(ring.host(i).T_G_C, x.data_ptr(), c.data_ptr(), l.data_ptr(), x.shape[0])
    integ.flush()
    integ.synchronize()
    t_all = time.perf_counter()
    ts = []
    for i in range(40, 40 + 80):
        x, c, l = ring.dev(i)
        t0 = time.perf_counter()
        integ.integrate_device(ring.host(i).T_G_C, x.data_ptr(), c.data_ptr(), l.data_ptr(), x.shape[0])
        ts.append((time.perf_counter() - t0) * 1e3)
    integ.flush()
    integ.synchronize()
    print("ms per frame %.3f" % ((time.perf_counter() - t_all) * 1e3 / 80))
    print("call ms:", " ".join("%.2f" % t for t in ts))
    print(integ.early_out_stats())
    integ.close()


if __name__ == "__main__":
    main()
