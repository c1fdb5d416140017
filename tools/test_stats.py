#!/usr/bin/env python3
"""Diagnostics: per-wave counters of k_test (build libks_hip_stats.so with -DKS_STATS, run with
KS_HIP_LIB=.../libks_hip_stats.so)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kimera_semantics_amd import binding as B, synth
import bench
import torch


def main():
    sc = synth.make_scene("room")
    cfg = B.default_config(max_tiles=1 << 13, max_points=640 * 480, **bench.common_cfg("fast"))
    h = B.HipIntegrator(cfg)
    L = B.lib()
    out = (C.c_ulonglong * 16)()
    for k in range(6):
        f = synth.render_frame(sc, synth.trajectory_pose(k), 640, 480, seed=k)
        L.ks_debug_test_stats(out)
        st = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        L.ks_debug_test_stats(out)
        v = list(out)
        waves = 1024 * 10
        print(f"frame {k}: rays {st.n_rays_cast} updates {st.n_voxel_updates} | all phases: cycles/wave avg {v[0]/waves:.0f} max {v[1]} "
              f"batches avg {v[2]/waves:.1f} max {v[3]} rays {v[4]} long {v[5]} rounds {v[6]} max/wave {v[7]} | phase7: cycles avg {v[8]/1024:.0f} max {v[9]} "
              f"batches avg {v[10]/1024:.1f} max {v[11]} long {v[12]} rounds {v[13]} maxlong/wave {v[14]} maxrounds/wave {v[15]}")


if __name__ == "__main__":
    main()
