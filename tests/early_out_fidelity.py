#!/usr/bin/env python3
"""Distance of the ORDERED-PHASE early-out schedule (what the GPU runs; restated in oracle/ks_oracle.cpp,
integrate_fast_phased) from the reference's SERIAL order (semantic_tsdf_integrator_fast.cpp:110-122), on
the CPU oracle alone: touched-voxel Jaccard, update-count ratio, label agreement on common voxels, per
phase growth factor (ks_config.early_out_phase_growth / 16).  CPU only; the numbers in DESIGN.md §3.2
come from here:   python tests/early_out_fidelity.py [640x480 | c4geom | c4]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kimera_semantics_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.util import COMMON, compare_maps  # noqa: E402

CASES = {
    "320x240": dict(scene="room", w=320, h=240, hfov=90.0, pose=7, geom={}),
    "640x480": dict(scene="room", w=640, h=480, hfov=90.0, pose=7, geom={}),
    "c4geom": dict(scene="hall", w=320, h=180, hfov=75.0, pose=3,
                   geom=dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=10.0)),
    "c4geom_small": dict(scene="hall", w=144, h=81, hfov=75.0, pose=3,
                         geom=dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=10.0)),
    "c4": dict(scene="hall", w=1280, h=720, hfov=75.0, pose=3,
               geom=dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=10.0)),
}


def fidelity(case, growths, with_maps=True):
    c = CASES[case]
    sc = synth.make_scene(c["scene"])
    f = synth.render_frame(sc, synth.trajectory_pose(c["pose"]), c["w"], c["h"], hfov_deg=c["hfov"], seed=c["pose"])
    kw = dict(COMMON, method=0, **c["geom"])
    serial = O.Oracle(O.default_config(**kw))
    ss = serial.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    rows = []
    for g in growths:
        o = O.Oracle(O.default_config(early_out_phase_growth=g, **kw))
        t0 = time.time()
        s = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        rep = compare_maps(serial, o, exact=False) if with_maps else {}   # (the map comparison dominates at 2 cm voxels)
        rows.append(dict(growth=g, phases_factor=g / 16.0, touched_jaccard=rep.get("touched_jaccard"),
                         block_jaccard=rep.get("block_jaccard"), updates_ratio=s.n_voxel_updates / ss.n_voxel_updates,
                         label_agreement=rep.get("label_agreement"), seconds=time.time() - t0))
        o.close()
    serial.close()
    return ss, rows


if __name__ == "__main__":
    case = sys.argv[1] if len(sys.argv) > 1 else "640x480"
    ss, rows = fidelity(case, [16, 20, 24, 32, 48, 64, 128])
    print(f"{case}: serial order {ss.n_rays_cast} rays, {ss.n_voxel_updates} updates")
    print("| growth/16 | touched-set Jaccard | block Jaccard | updates / serial | label agreement (common voxels) |")
    print("|---|---|---|---|---|")
    for r in rows:
        la = "-" if r["label_agreement"] is None else f"{r['label_agreement']:.4f}"
        print(f"| {r['phases_factor']:.2f} | {r['touched_jaccard']:.4f} | {r['block_jaccard']:.4f} | {r['updates_ratio']:.3f} | {la} |")
