#!/usr/bin/env python3
"""Diagnostics: where a k_test_pre wavefront spends its time (sections A walk / B shared-set look-ups / C rays in
order), from in-kernel counters.  Needs the -DKS_STATS build:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DKS_STATS -shared -o kimera_semantics_amd/libks_hip_stats.so kimera_semantics_amd/csrc/ks_hip.hip
  KS_HIP_LIB=$PWD/kimera_semantics_amd/libks_hip_stats.so KS_TEST_PRE=1 python tools/test_pre_stats.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from kimera_semantics_amd import binding as B  # noqa: E402


def main():
    wl = dict(bench.WORKLOADS["C2"], w=640, h=480, method="fast")
    frames = bench.make_frames(wl, list(range(4)))
    cfg = B.default_config(max_tiles=1 << 13, max_points=640 * 480, pipeline_frames=0, **bench.integ_cfg(wl))
    h = B.HipIntegrator(cfg)
    L = B.lib()
    out = (C.c_ulonglong * 16)()
    for k, f in enumerate(frames):
        L.ks_debug_test_stats(out)
        st = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        h.synchronize()
        L.ks_debug_test_stats(out)
        v = list(out)
        n = max(1, v[11])
        print(f"frame {k}: updates {st.n_voxel_updates} | k_test: waves {v[2]} cycles/wave {v[0] / max(1, v[2]):.0f} | k_test_pre: waves {v[11]} "
              f"cycles/wave A {v[8] / n:.0f} B {v[9] / n:.0f} C {v[10] / n:.0f}; wall/wave {v[12] / n / 100.0:.2f} us (max {v[15] / 100.0:.2f} us) "
              f"steps walked/wave {v[13] / n:.1f} chunks/wave {v[14] / n:.1f}", flush=True)
    h.close()


if __name__ == "__main__":
    main()
