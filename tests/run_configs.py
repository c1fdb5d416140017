#!/usr/bin/env python3
"""Runs the BASELINE.json configurations C1..C4 on one MI355X next to the CPU oracle and
prints a markdown table (results are pasted into BASELINE.md §3).  Needs a GPU."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kimera_semantics_amd import binding as B  # noqa: E402
from kimera_semantics_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

CFG = {
    "C1": dict(scene="room", w=640, h=480, hfov=90.0, voxel=0.05, max_ray=5.0, frames=1, traj="single"),
    "C2": dict(scene="room", w=640, h=480, hfov=90.0, voxel=0.05, max_ray=5.0, frames=200, traj="circle", method="fast"),
    "C3": dict(scene="room", w=640, h=480, hfov=90.0, voxel=0.05, max_ray=5.0, frames=200, traj="circle", method="merged"),
    "C4": dict(scene="hall", w=1280, h=720, hfov=75.0, voxel=0.02, max_ray=10.0, frames=90, traj="circle"),
}


def run(name, method, n_frames, cpu_frames, max_tiles):
    import torch
    c = CFG[name]
    sc = synth.make_scene(c["scene"])
    kw = dict(method=0 if method == "fast" else 1, voxel_size=c["voxel"], voxels_per_side=16,
              truncation_distance=4 * c["voxel"], max_ray_length_m=c["max_ray"], semantic_measurement_probability=0.8,
              dynamic_labels=[20], label_rgba=synth.default_label_colors())
    n_frames = min(n_frames, c["frames"])
    radius = 1.5 if c["scene"] == "room" else 3.0
    frames = []
    for k in range(n_frames):
        T = synth.single_pose() if c["traj"] == "single" else synth.trajectory_pose(k, radius=radius)
        frames.append(synth.render_frame(sc, T, c["w"], c["h"], hfov_deg=c["hfov"], seed=k))
    dev = [(torch.from_numpy(f.xyz).cuda(), torch.from_numpy(f.rgba).cuda(), torch.from_numpy(f.labels).cuda()) for f in frames]
    # timed pass: pipelined frames (a stream of frames), no per-stage events
    h = B.HipIntegrator(B.default_config(max_tiles=max_tiles, max_points=c["w"] * c["h"], pipeline_frames=2, **kw))
    torch.cuda.synchronize()
    upd = 0
    t0 = time.perf_counter()
    for f, (x, col, lab) in zip(frames, dev):
        st = h.integrate_device(f.T_G_C, x.data_ptr(), col.data_ptr(), lab.data_ptr(), x.shape[0])
        upd += st.n_voxel_updates
    upd += h.flush().n_voxel_updates
    h.synchronize()
    dt = time.perf_counter() - t0
    tiles = len(h.tile_keys())
    h.close()
    # per-stage pass: unpipelined context, events around every stage (apply figures = kernel alone on the GPU)
    h = B.HipIntegrator(B.default_config(max_tiles=max_tiles, max_points=c["w"] * c["h"], pipeline_frames=0, **kw))
    h.profile_enable(1)
    t1 = time.perf_counter()
    upd1 = 0
    for f, (x, col, lab) in zip(frames, dev):
        upd1 += h.integrate_device(f.T_G_C, x.data_ptr(), col.data_ptr(), lab.data_ptr(), x.shape[0]).n_voxel_updates
    h.synchronize()
    dt1 = time.perf_counter() - t1
    prof = h.profile()
    apply_ms = (prof["ms"]["apply"] + prof["ms"]["apply_long"]) / max(1, prof["frames"])
    res = dict(config=name, method=method, frames=n_frames, points=int(np.mean([len(f.xyz) for f in frames])),
               updates_per_frame=int(upd / n_frames), gpu_ms_per_frame=round(dt / n_frames * 1e3, 3),
               gpu_Mupd_s=round(upd / dt / 1e6, 1), gpu_fps=round(n_frames / dt, 1),
               unpipelined_ms_per_frame_with_stage_events=round(dt1 / n_frames * 1e3, 3),
               apply_ms=round(apply_ms, 4),
               apply_alg_GBs=round(208 * upd1 / n_frames / (apply_ms * 1e-3) / 1e9, 1) if apply_ms else 0,
               tiles=tiles,
               stage_ms={k: round(v / n_frames, 3) for k, v in prof["ms"].items()})
    # CPU oracle on a bounded sample
    cores = os.cpu_count() or 1
    nc = min(cpu_frames, n_frames)
    for tag, threads in (("mt", cores), ("st", 1)):
        o = O.Oracle(O.default_config(integrator_threads=threads, **kw))
        u = 0
        t0 = time.perf_counter()
        for f in frames[:nc]:
            u += o.integrate(f.T_G_C, f.xyz, f.rgba if method == "fast" else None, f.labels).n_voxel_updates
        d = time.perf_counter() - t0
        res[f"cpu_{tag}_Mupd_s"] = round(u / d / 1e6, 3)
        res[f"cpu_{tag}_fps"] = round(nc / d, 2)
        res[f"cpu_{tag}_updates_per_frame"] = int(u / nc)
        o.close()
    res["cpu_cores"] = cores
    res["cpu_sample_frames"] = nc
    h.close()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C1,C2,C3,C4")
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--cpu-frames", type=int, default=3)
    a = ap.parse_args()
    out = []
    for name in a.configs.split(","):
        methods = [CFG[name].get("method")] if CFG[name].get("method") else ["fast", "merged"]
        for m in methods:
            big = name == "C4"
            r = run(name, m, 12 if big else a.frames, 1 if big else a.cpu_frames, (1 << 18) if big else (1 << 13))
            print(json.dumps(r), flush=True)
            out.append(r)
