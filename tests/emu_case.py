"""One parity case through the HOST FUNCTIONAL MODEL of the library (tools/emu: ks_hip.hip and every kernel header
compiled for the CPU, work-items as fibers), run as a subprocess by tests/test_emu_parity.py:
    KS_HIP_LIB=tools/emu/_build/libks_hip_emu.so python -m tests.emu_case '<json spec>'
The checker is the oracle, exactly as in the GPU tier (tests/util.compare_maps, bit-exact)."""
import json
import os
import sys


def main():
    spec = json.loads(sys.argv[1])
    assert os.environ.get("KS_HIP_LIB", "").endswith("libks_hip_emu.so"), "this script drives the functional model only"
    from kimera_semantics_amd import binding as B
    from kimera_semantics_amd import synth
    from oracle import oracle_py as O
    from tests.util import COMMON, NO_EARLY_OUT, compare_maps
    from tests.variants import random_combo
    cfg = dict(COMMON, method=spec["method"])
    if spec.get("random_combo") is not None:
        cfg.update(random_combo(spec["random_combo"]))
    cfg.update(spec.get("cfg", {}))
    if spec.get("no_early_out"):
        cfg["max_consecutive_ray_collisions"] = NO_EARLY_OUT
    ocfg, hcfg = dict(cfg), dict(cfg)
    if spec.get("exact"):
        hcfg["early_out_phase_growth"] = B.KS_EARLY_OUT_EXACT   # oracle: growth 0 = the reference's serial loop
    w, h = spec["size"]
    o = O.Oracle(O.default_config(integrator_threads=1, **ocfg))
    g = B.HipIntegrator(B.default_config(max_tiles=spec.get("max_tiles", 2048), max_points=w * h, pipeline_frames=spec.get("pipeline", 0), **hcfg))
    sc = synth.make_scene("room")
    tot_o = tot_g = 0
    for k in range(spec.get("frames", 1)):
        T = synth.pose_to_T((3.5, 0.3, 1.2), 0.1) if (spec.get("close_up_first") and k == 0) else synth.trajectory_pose(0 if spec.get("fixed_pose") else 5 * k)
        f = synth.render_frame(sc, T, w, h, seed=40 + k)   # (close_up_first: 0.45 m from a wall — bundles of thousands of points)
        if spec.get("cloud") == "axis":
            # rays with one or two zero components seen from an unrotated sensor (crossing times inf / NaN: the serial
            # caster of the owner lane instead of the parallel one), among ordinary ones
            import numpy as np
            rng = np.random.default_rng(3 + k)
            n = w * h
            pts = rng.uniform(-2.5, 2.5, size=(n, 3)).astype(np.float32)
            kind = rng.integers(0, 4, size=n)
            pts[kind == 0, 1:] = 0
            pts[kind == 1, 0] = 0
            pts[kind == 2, :2] = 0
            pts[np.linalg.norm(pts, axis=1) < 0.3] += 1.0
            f.xyz = pts
            f.labels = rng.integers(0, 21, size=n, dtype=np.uint8)
            f.rgba = synth.default_label_colors()[f.labels]
            f.T_G_C = np.array([1, 0, 0, 0, 0.3 * k, -0.2 * k, 0.025 * k], np.float32)
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sg = g.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        if spec.get("rays_at_least") is not None:   # (merged: rays = bundles — the case is meant to reach the bundle order's large epochs)
            assert so.n_rays_cast >= spec["rays_at_least"], so.n_rays_cast
        tot_o += so.n_voxel_updates
        tot_g += sg.n_voxel_updates
    tot_g += g.flush().n_voxel_updates
    assert tot_o == tot_g, (tot_o, tot_g)
    if spec.get("fallbacks_exactly") is not None:
        st = g.early_out_stats()
        assert st["fallbacks"] == spec["fallbacks_exactly"] and st["event_driven"], st
    if spec.get("fallbacks_below") is not None:
        st = g.early_out_stats()
        assert 0 < st["fallbacks"] < spec["fallbacks_below"], st   # (some frames repeat on the host, then the device loop takes over again)
    if spec.get("xl_walked_at_least") is not None:
        st = g.update_stats()
        assert st["walked"] >= spec["xl_walked_at_least"] and st["chunks"] > 0, st
        print("update_stats", st)
    rep = compare_maps(o, g, exact=True)
    assert rep["oracle_touched"] > 500, rep
    print("EMU_CASE_OK", json.dumps({"updates": tot_g, "voxels": rep["oracle_touched"]}))


if __name__ == "__main__":
    main()
