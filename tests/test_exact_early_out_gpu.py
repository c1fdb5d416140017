"""-m gpu: `fast` with the early-out enabled in the reference's SERIAL order (KS_EARLY_OUT_EXACT,
csrc/ks_k_exact.h).  The reference's loop (semantic_tsdf_integrator_fast.cpp:110-122) is serial by construction;
at integrator_threads = 1 it is deterministic, and that result is what this mode must reproduce bit for bit:
against the oracle's serial restatement (several frames, configuration variants, the zero-hash slot artefact of
ApproxHashSet), against the REAL reference sources (oracle/_ref), and against the golden digest generated
from them (tests/golden/ref_fast_default_640x480.npz)."""
import math
import os

import numpy as np
import pytest

from kimera_semantics_amd import binding as B
from kimera_semantics_amd import synth
from oracle import oracle_py as O
from oracle import ref_py as R
from tests.util import COMMON, compare_maps

pytestmark = pytest.mark.gpu


def _pair(max_points=1 << 19, **kw):
    okw = dict(COMMON, method=0, **kw)
    o = O.Oracle(O.default_config(**okw))     # early_out_phase_growth = 0: the serial loop, one thread
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=max_points, early_out_phase_growth=B.KS_EARLY_OUT_EXACT, **okw))
    return o, h


def _run(o, h, frames):
    for k, f in enumerate(frames):
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        assert (so.n_valid_points, so.n_rays_cast, so.n_voxel_updates) == (sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates), k
    return compare_maps(o, h, exact=True)


@pytest.mark.parametrize("size,frames", [((160, 120), 4), ((320, 240), 3), ((640, 480), 3)])
def test_exact_early_out_equals_serial_oracle(size, frames):
    o, h = _pair()
    sc = synth.make_scene("room")
    rep = _run(o, h, [synth.render_frame(sc, synth.trajectory_pose(5 * k), size[0], size[1], seed=30 + k) for k in range(frames)])
    assert rep["oracle_touched"] > 1000
    nf, it = h.early_out_iterations()
    assert nf == frames and it >= frames
    print(f"exact early-out {size}: {it / nf:.1f} fix-point iterations per frame")


@pytest.mark.parametrize("variant", ["clear_every_3", "sorted_order", "subsample_1", "limit_0", "limit_5", "no_carving", "pipeline_ignored"])
def test_exact_early_out_variants(variant):
    kw = {}
    pipe = 0
    if variant == "clear_every_3":
        kw["clear_checks_every_n_frames"] = 3     # marks of earlier frames stay valid (same offset)
    elif variant == "sorted_order":
        kw["integration_order_mode"] = 1
    elif variant == "subsample_1":
        kw["start_voxel_subsampling_factor"] = 1.0
    elif variant == "limit_0":
        kw["max_consecutive_ray_collisions"] = 0
    elif variant == "limit_5":
        kw["max_consecutive_ray_collisions"] = 5
    elif variant == "no_carving":
        kw["voxel_carving_enabled"] = 0
    elif variant == "pipeline_ignored":
        pipe = 3
    okw = dict(COMMON, method=0, **kw)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, pipeline_frames=pipe,
                                         early_out_phase_growth=B.KS_EARLY_OUT_EXACT, **okw))
    sc = synth.make_scene("room")
    for k in range(5):
        f = synth.render_frame(sc, synth.trajectory_pose(2 * k), 160, 120, seed=60 + k)
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=(k == 3))
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=(k == 3))
    h.flush()
    compare_maps(o, h, exact=True)


def test_exact_early_out_zero_hash_slot_artefact():
    """ApproxHashSet's zero-initialised slots "contain" hash 0 once the offset is > 0: the voxel whose hash is 0
    — voxel (0, 0, 0) — looks already observed the first time a frame reaches it
    (semantic_tsdf_integrator_fast.h:102-130).  Rays looking down at the floor around the world origin cross it."""
    o, h = _pair()
    sc = synth.make_scene("room")
    frames = [synth.render_frame(sc, synth.pose_to_T((-1.5 + 0.1 * k, 0.02, 1.5), 0.0, math.radians(45.0)), 200, 150, seed=k)
              for k in range(3)]
    _run(o, h, frames)
    # the origin voxel is inside the map (so the artefact was exercised, not dodged)
    idx, t, _ = o.download(np.array([[0, 0, 0]], dtype=np.int32))
    assert t["weight"][0, 0] > 0 or t["weight"][0].max() > 0


def test_exact_early_out_c4_geometry():
    geom = dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=10.0)
    okw = dict(COMMON, method=0, **geom)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 15, max_points=1 << 17, early_out_phase_growth=B.KS_EARLY_OUT_EXACT, **okw))
    sc = synth.make_scene("hall")
    frames = [synth.render_frame(sc, synth.trajectory_pose(3 + k, radius=3.0), 160, 90, hfov_deg=75.0, seed=3 + k) for k in range(2)]
    _run(o, h, frames)


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
def test_exact_early_out_equals_real_reference_sources(tmp_path):
    """default fast (max_consecutive_ray_collisions = 2) over 3 frames at 640x480: HIP == the real sources."""
    csv = str(tmp_path / "labels.csv")
    R.write_label_csv(csv, synth.default_label_colors())
    r = R.Reference("fast", csv)
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 14, max_points=1 << 19, early_out_phase_growth=B.KS_EARLY_OUT_EXACT,
                                         **dict(COMMON, method=0)))
    sc = synth.make_scene("room")
    for k in range(3):
        f = synth.render_frame(sc, synth.trajectory_pose(5 + 2 * k), 640, 480, seed=5 + k)
        r.integrate(f.T_G_C, f.xyz, f.rgba)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    ri, hi = r.block_indices(), h.block_indices()
    assert np.array_equal(ri, hi)
    _, rt, rs = r.download(ri)
    _, ht, hs = h.download(ri)
    assert int((rt["weight"] > 0).sum()) > 100000
    assert np.array_equal(rs["label"], hs["label"])
    assert np.array_equal(rs["priors"].view(np.uint32), hs["priors"].view(np.uint32))
    assert np.array_equal(rt["distance"].view(np.uint32), ht["distance"].view(np.uint32))
    assert np.array_equal(rt["weight"].view(np.uint32), ht["weight"].view(np.uint32))
    assert np.array_equal(rt["color"], ht["color"]) and np.array_equal(rs["color"], hs["color"])


def test_exact_early_out_reproduces_reference_golden():
    from tests.test_golden_ref import _cfg, _check
    name = "ref_fast_default_640x480"
    _check(B.HipIntegrator(B.default_config(max_tiles=1 << 15, max_points=1 << 19, early_out_phase_growth=B.KS_EARLY_OUT_EXACT,
                                            **_cfg(name))), name)
