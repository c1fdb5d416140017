#!/usr/bin/env python3
"""Integrates a few frames of one bench workload (C2 | C3 | C4-fast | C4-merged), unpipelined: meant to run under
rocprofv3 --kernel-trace (then tools/frame_timeline.py) or with KS_HIP_LIB pointing at a diagnostics build."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kimera_semantics_amd import binding as B


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    if name == "calib":
        # PMC calibration: k_export_tiles reads and writes exactly 65536 B per tile, 16 B per lane, coalesced
        import torch
        import numpy as np
        wl = bench.WORKLOADS["C3"]
        f = bench.make_frames(wl, range(1))[0]
        h = B.HipIntegrator(B.default_config(max_tiles=1 << 13, max_points=wl["w"] * wl["h"], **bench.integ_cfg(wl)))
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        keys = h.tile_keys()
        buf = torch.empty((len(keys), 16384), dtype=torch.int32, device="cuda")
        for _ in range(n):
            h.export_tiles(np.arange(len(keys), dtype=np.uint32), buf.data_ptr())
        print("calib tiles", len(keys), "bytes_each_way", len(keys) * 65536, flush=True)
        h.close()
        return
    wl = bench.WORKLOADS[name]
    frames = bench.make_frames(wl, range(n))
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 16 if name.startswith("C4") else 1 << 13, max_points=wl["w"] * wl["h"],
                                         **bench.integ_cfg(wl)))
    for f in frames:
        st = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        print(name, "rays", st.n_rays_cast, "updates", st.n_voxel_updates, flush=True)
    h.close()


if __name__ == "__main__":
    main()
