#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (the reference itself has no tests or
golden vectors to import — SURVEY.md §4/§8c — and cannot be built without Voxblox).
Run from the repo root:  python tests/golden/make_golden.py
Each fixture holds the INPUT cloud and the oracle's OUTPUT map (touched voxels only)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from kimera_semantics_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
PRIOR_INIT = np.float32(-0.60205999132)

CASES = {
    # name: (method, extra config)
    "fast_noearlyout": (0, dict(max_consecutive_ray_collisions=1 << 30)),
    "fast_default": (0, dict()),
    "merged": (1, dict(bundle_order=1)),                    # first-insertion bundle order (the container-independent one)
    "merged_color_mode": (1, dict(color_mode=0, bundle_order=1)),
}
GEOM = dict(voxel_size=0.2, voxels_per_side=16, truncation_distance=0.8, max_ray_length_m=5.0,
            semantic_measurement_probability=0.8, dynamic_labels=[20])


def flatten(indices, t, s):
    """Touched voxels only, in (block, linear index) order."""
    touched = (t["weight"] > 0) | (s["label"] != 0) | (np.abs(s["priors"] - PRIOR_INIT).max(axis=-1) > 0)
    b, v = np.nonzero(touched)
    return dict(block_indices=indices.astype(np.int32), vox_block=b.astype(np.int32), vox_linear=v.astype(np.int32),
                distance=t["distance"][b, v], weight=t["weight"][b, v], tsdf_color=t["color"][b, v],
                label=s["label"][b, v], sem_color=s["color"][b, v], priors=s["priors"][b, v])


def main():
    sc = synth.make_scene("room")
    frames = [synth.render_frame(sc, synth.trajectory_pose(3 * k), 64, 48, seed=100 + k) for k in range(2)]
    for name, (method, extra) in CASES.items():
        cfg = O.default_config(method=method, label_rgba=synth.default_label_colors(), **GEOM, **extra)
        o = O.Oracle(cfg)
        stats = []
        for f in frames:
            st = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
            stats.append([st.n_valid_points, st.n_rays_cast, st.n_voxel_updates])
        out = flatten(*o.download())
        priors = out.pop("priors")
        out["priors_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(priors).tobytes()).digest(), dtype=np.uint8)
        out["priors_label_value"] = priors[np.arange(len(priors)), out["label"]]  # the winning class prior
        for k, f in enumerate(frames):
            out[f"T{k}"], out[f"xyz{k}"], out[f"rgba{k}"], out[f"labels{k}"] = f.T_G_C, f.xyz, f.rgba, f.labels
        out["stats"] = np.array(stats, dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, "touched voxels", len(out["label"]), "stats", stats)


if __name__ == "__main__":
    main()
