"""Shared helpers for the parity tests: one config dict feeds both the oracle and the HIP
integrator; compare_maps() checks the parity contract of SURVEY.md §8c."""
import numpy as np

from kimera_semantics_amd import synth

COMMON = dict(semantic_measurement_probability=0.8, dynamic_labels=[20],
              label_rgba=synth.default_label_colors())
NO_EARLY_OUT = 1 << 30  # max_consecutive_ray_collisions value that disables fast's early termination

# Stated float tolerances (BASELINE.json north_star: labels/indices bit-exact, TSDF within tolerance)
TOL_DISTANCE_ABS = 1e-5
TOL_WEIGHT_REL = 1e-5
TOL_PRIORS_ABS = 1e-4


def small_frame(seed=0, w=160, h=120, pose=None, scene="room", hfov=90.0):
    sc = synth.make_scene(scene)
    T = synth.single_pose() if pose is None else pose
    return synth.render_frame(sc, T, w, h, hfov_deg=hfov, seed=seed)


def compare_maps(oracle, hip, exact=True):
    """Returns a report dict; raises AssertionError on contract violations.
    exact=True: labels/priors/distance/weight/colours must all be bit-identical (the HIP
    path replays the oracle's per-voxel order).  exact=False: statistical comparison."""
    oi = oracle.block_indices()
    hi = hip.block_indices()
    rep = {"oracle_blocks": len(oi), "hip_blocks": len(hi)}
    so = {tuple(x) for x in oi.tolist()}
    sh = {tuple(x) for x in hi.tolist()}
    rep["block_jaccard"] = len(so & sh) / max(1, len(so | sh))
    if exact:
        assert so == sh, f"allocated block sets differ: only oracle {sorted(so - sh)[:5]}, only hip {sorted(sh - so)[:5]}"
        if hasattr(oracle, "semantic_block_indices"):
            assert {tuple(x) for x in oracle.semantic_block_indices().tolist()} == so
    common = np.array(sorted(so & sh), dtype=np.int32).reshape(-1, 3)
    _, ot, osem = oracle.download(common)
    _, ht, hsem = hip.download(common)
    o_touched = (ot["weight"] > 0) | (osem["label"] != 0) | (np.abs(osem["priors"] - np.float32(-0.60205999132)).max(axis=-1) > 0)
    h_touched = (ht["weight"] > 0) | (hsem["label"] != 0) | (np.abs(hsem["priors"] - np.float32(-0.60205999132)).max(axis=-1) > 0)
    rep["voxels_compared"] = int(ot.size)
    rep["oracle_touched"] = int(o_touched.sum())
    rep["hip_touched"] = int(h_touched.sum())
    rep["touched_jaccard"] = float((o_touched & h_touched).sum() / max(1, (o_touched | h_touched).sum()))
    rep["label_mismatches"] = int((osem["label"] != hsem["label"]).sum())
    rep["max_abs_distance_err"] = float(np.abs(ot["distance"] - ht["distance"]).max()) if ot.size else 0.0
    w_o, w_h = ot["weight"], ht["weight"]
    rep["max_rel_weight_err"] = float((np.abs(w_o - w_h) / np.maximum(np.abs(w_o), 1e-12)).max()) if ot.size else 0.0
    rep["max_abs_priors_err"] = float(np.abs(osem["priors"] - hsem["priors"]).max()) if ot.size else 0.0
    rep["tsdf_color_mismatches"] = int((ot["color"] != ht["color"]).any(axis=-1).sum())
    rep["sem_color_mismatches"] = int((osem["color"] != hsem["color"]).any(axis=-1).sum())
    if exact:
        assert rep["label_mismatches"] == 0, rep
        assert np.array_equal(osem["priors"].view(np.uint32), hsem["priors"].view(np.uint32)), rep
        assert np.array_equal(ot["distance"].view(np.uint32), ht["distance"].view(np.uint32)), rep
        assert np.array_equal(ot["weight"].view(np.uint32), ht["weight"].view(np.uint32)), rep
        assert rep["tsdf_color_mismatches"] == 0 and rep["sem_color_mismatches"] == 0, rep
        # and therefore within the stated tolerances
        assert rep["max_abs_distance_err"] <= TOL_DISTANCE_ABS
        assert rep["max_rel_weight_err"] <= TOL_WEIGHT_REL
        assert rep["max_abs_priors_err"] <= TOL_PRIORS_ABS
    return rep
