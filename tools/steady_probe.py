#!/usr/bin/env python3
"""Steady-state rate of one bench workload, pipelined, device-resident inputs: N frames in ONE region (the bench's regions of K frames
pay the pipeline's fill and drain once per K).   usage: steady_probe.py <workload> <frames> [pipeline_frames]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kimera_semantics_amd import binding as B

name, N = sys.argv[1], int(sys.argv[2])
pipe = int(sys.argv[3]) if len(sys.argv) > 3 else 8
wl = bench.WORKLOADS[name]
n_ring = 12 if name.startswith("C4") else 40
frames = bench.make_frames(wl, range(n_ring))
ring = bench.FrameRing(frames, torch, torch.device("cuda:0"))
h = B.HipIntegrator(B.default_config(pipeline_frames=pipe, max_tiles=1 << 16 if name.startswith("C4") else 1 << 13, max_points=wl["w"] * wl["h"],
                                     **bench.integ_cfg(wl)))


def run(first, count):
    upd = 0
    for i in range(first, first + count):
        x, c, l = ring.dev(i)
        upd += h.integrate_device(ring.host(i).T_G_C, x.data_ptr(), c.data_ptr(), l.data_ptr(), x.shape[0]).n_voxel_updates
    upd += h.flush().n_voxel_updates
    h.synchronize()
    return upd


run(0, 2 * n_ring)
for count in (n_ring, N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    upd = run(2 * n_ring, count)
    dt = time.perf_counter() - t0
    print(f"{name} pipeline {pipe}: {count} frames in one region: {1e3 * dt / count:.4f} ms/frame, {upd / dt / 1e6:.1f} M updates/s")
h.close()
