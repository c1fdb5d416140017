#!/usr/bin/env python3
"""Integrates TURNS turns of a ring of 6 frames of one bench workload, unpipelined (under rocprofv3 --kernel-trace: the last
frame is a steady-state frame — every voxel it touches has been touched before).  usage: probe_ring.py <workload> <turns>"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kimera_semantics_amd import binding as B

name = sys.argv[1]
turns = int(sys.argv[2])
wl = bench.WORKLOADS[name]
frames = bench.make_frames(wl, range(6))
h = B.HipIntegrator(B.default_config(max_tiles=1 << 16 if name.startswith("C4") else 1 << 13, max_points=wl["w"] * wl["h"], **bench.integ_cfg(wl)))
for t in range(turns):
    for f in frames:
        st = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
print(name, "updates of the last frame", st.n_voxel_updates, h.update_stats())
h.close()
