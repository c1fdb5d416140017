#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry (ks_integrate_points: pageable host buffers ->
H2D copy -> integrate), reported in DESIGN.md next to the HBM-resident `value` of bench.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kimera_semantics_amd import binding as B  # noqa: E402
from kimera_semantics_amd import synth  # noqa: E402

sc = synth.make_scene("room")
frames = [synth.render_frame(sc, synth.trajectory_pose(k), 640, 480, seed=k) for k in range(45)]
for method, pipe in ((0, 0), (0, 1), (1, 0), (1, 1)):
    h = B.HipIntegrator(B.default_config(method=method, max_tiles=1 << 13, max_points=640 * 480,
                                         semantic_measurement_probability=0.8, dynamic_labels=[20],
                                         pipeline_frames=pipe, label_rgba=synth.default_label_colors()))
    for f in frames[:5]:
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    h.synchronize()
    upd = 0
    t0 = time.perf_counter()
    for f in frames[5:]:
        upd += h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates
    upd += h.flush().n_voxel_updates
    h.synchronize()
    dt = time.perf_counter() - t0
    print(f"method={'merged' if method else 'fast'} pipeline_frames={pipe} host-pointer entry: {dt / 40 * 1e3:.3f} ms/frame, "
          f"{upd / dt / 1e6:.1f} Mvoxel-updates/s, {40 / dt:.1f} frames/s (5.2 MB H2D per frame)")
    t0 = time.perf_counter()
    for f in frames[5:25]:
        h.integrate_depth(f.T_G_C, f.depth, f.K, label_img=f.label_img)
    h.synchronize()
    dt = time.perf_counter() - t0
    print(f"   depth+label image entry (1.5 MB H2D per frame): {dt / 20 * 1e3:.3f} ms/frame, {20 / dt:.1f} frames/s")
    h.close()
