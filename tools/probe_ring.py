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
pipe = int(os.environ.get("KS_PROBE_PIPE", "0"))   # > 0: pipelined, that many frames in flight (tools/pipe_trace.sh)
h = B.HipIntegrator(B.default_config(pipeline_frames=pipe, max_tiles=1 << 16 if name.startswith("C4") else 1 << 13, max_points=wl["w"] * wl["h"], **bench.integ_cfg(wl)))
if pipe:
    import torch   # device-resident inputs, as the bench's timed region has them
    ring = bench.FrameRing(frames, torch, torch.device("cuda:0"))
    for i in range(turns * len(frames)):
        x, c, l = ring.dev(i)
        st = h.integrate_device(ring.host(i).T_G_C, x.data_ptr(), c.data_ptr(), l.data_ptr(), x.shape[0])
    st = h.flush()
else:
    for t in range(turns):
        for f in frames:
            st = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
print(name, "updates of the last frame", st.n_voxel_updates, h.update_stats())
h.close()
