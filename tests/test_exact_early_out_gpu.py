"""-m gpu: `fast` with the early-out enabled in the reference's SERIAL order (the library's default since round 4;
KS_EARLY_OUT_EXACT / early_out_phase_growth = 0, csrc/ks_k_exact.h).  The reference's loop (semantic_tsdf_integrator_fast.cpp:110-122) is serial by construction;
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


@pytest.mark.parametrize("variant", ["clear_every_3", "sorted_order", "subsample_1", "limit_0", "limit_5", "no_carving", "pipelined_3", "clear_every_3_pipelined"])
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
    elif variant == "pipelined_3":
        pipe = 3
    elif variant == "clear_every_3_pipelined":   # (a frame's marks are inputs of the next frame: the library runs one frame at a time)
        kw["clear_checks_every_n_frames"] = 3
        pipe = 4
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
@pytest.mark.parametrize("order", [pytest.param(0, id="mixed_upstream"), pytest.param(2, id="mixed_1024_groups")])
def test_exact_early_out_equals_real_reference_sources(tmp_path, order):
    """default fast (max_consecutive_ray_collisions = 2) over 3 frames at 640x480: HIP == the real sources, in either
    reading of Voxblox's "mixed" order (the early-out is where the order matters most: which ray meets which ray's marks)."""
    csv = str(tmp_path / "labels.csv")
    R.write_label_csv(csv, synth.default_label_colors())
    r = R.Reference("fast", csv, order_mode="mixed" if order == 0 else "mixed_1024_groups")
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 14, max_points=1 << 19, early_out_phase_growth=B.KS_EARLY_OUT_EXACT,
                                         integration_order_mode=order, **dict(COMMON, method=0)))
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


def _frames(n, size=(320, 240), first=0):
    sc = synth.make_scene("room")
    return [synth.render_frame(sc, synth.trajectory_pose(first + 3 * k), size[0], size[1], seed=70 + k) for k in range(n)]


def _totals(o, h, frames):
    to = tg = 0
    for f in frames:
        to += o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates
        tg += h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates
    tg += h.flush().n_voxel_updates
    assert to == tg
    return compare_maps(o, h, exact=True)


def test_default_configuration_is_the_serial_result():
    """early_out_phase_growth = 0 (ks_default_config): the reference's serial result, by the event-driven loop."""
    okw = dict(COMMON, method=0)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 17, **okw))
    _totals(o, h, _frames(3))
    st = h.early_out_stats()
    assert st["event_driven"] and st["frames"] == 3 and st["fallbacks"] == 0 and st["rounds"] >= 3, st


@pytest.mark.parametrize("pipe", [2, 4, 8, 16])
def test_exact_early_out_pipelined(pipe):
    """frames in flight: the fix point of frame i + 1 runs beside that of frame i; what is left of the dependence
    between frames (the table earlier frames leave behind) is carried by the commit events.  (8: batches of four frames
    per launch sequence; 16: batches of eight, 24 frame slots — two full batches and a partial one.)"""
    okw = dict(COMMON, method=0)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 17, pipeline_frames=pipe, **okw))
    n = 14 if pipe < 16 else 21
    _totals(o, h, _frames(n))
    st = h.early_out_stats()
    assert st["event_driven"] and st["pipelined"] and st["frames"] == n and st["fallbacks"] == 0, st
    print(f"pipeline {pipe}: {st['rounds'] / st['frames']:.1f} rounds per frame")


def test_exact_early_out_zero_hash_slot_artefact_pipelined():
    okw = dict(COMMON, method=0)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 17, pipeline_frames=4, **okw))
    sc = synth.make_scene("room")
    frames = [synth.render_frame(sc, synth.pose_to_T((-1.5 + 0.1 * k, 0.02, 1.5), 0.0, math.radians(45.0)), 200, 150, seed=k) for k in range(9)]
    _totals(o, h, frames)
    assert h.early_out_stats()["fallbacks"] == 0


@pytest.mark.parametrize("pipe", [0, 4])
def test_exact_early_out_falls_back_to_the_host_loop_and_grows(monkeypatch, pipe):
    """marks / X marks that do not fit: the frame (and the frames in flight behind it) repeat their fix point through
    the host-driven loop, the buffers grow, later frames run on the device again — same map."""
    monkeypatch.setenv("KS_DEBUG", "1")   # the one gate in front of the library's diagnostic switches
    monkeypatch.setenv("KS_EXACT_CAP_MARKS", "30000")
    monkeypatch.setenv("KS_EXACT_CAP_X", "16")
    okw = dict(COMMON, method=0)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 17, pipeline_frames=pipe, **okw))
    monkeypatch.delenv("KS_EXACT_CAP_MARKS")
    monkeypatch.delenv("KS_EXACT_CAP_X")
    _totals(o, h, _frames(24))
    st = h.early_out_stats()
    assert 0 < st["fallbacks"] < 24, st


def test_exact_early_out_host_loop_switch(monkeypatch):
    monkeypatch.setenv("KS_DEBUG", "1")
    monkeypatch.setenv("KS_EXACT_HOST_LOOP", "1")
    okw = dict(COMMON, method=0)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 17, pipeline_frames=4, **okw))
    monkeypatch.delenv("KS_EXACT_HOST_LOOP")
    _totals(o, h, _frames(3))
    st = h.early_out_stats()
    assert not st["event_driven"] and not st["pipelined"], st


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
def test_full_size_c4_frame_exact_early_out_vs_real_reference(tmp_path):
    """One FULL-SIZE C4 frame (1280x720, 2 cm voxels, 10 m rays, 75 deg), default `fast` (early-out after 2 consecutive
    observed voxels): HIP == the real reference sources at integrator_threads = 1, every voxel."""
    geom = dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=10.0)
    csv = str(tmp_path / "labels.csv")
    R.write_label_csv(csv, synth.default_label_colors())
    r = R.Reference("fast", csv, voxel_size=geom["voxel_size"], truncation=geom["truncation_distance"], max_ray=geom["max_ray_length_m"])
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 16, max_points=1280 * 720, **dict(COMMON, method=0, **geom)))
    f = synth.render_frame(synth.make_scene("hall"), synth.trajectory_pose(3, radius=3.0), 1280, 720, hfov_deg=75.0, seed=3)
    r.integrate(f.T_G_C, f.xyz, f.rgba)
    st = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert st.n_voxel_updates > 1e7
    print("C4 frame, exact early-out:", st.n_voxel_updates, "updates;", h.early_out_stats())
    ri, hi = r.block_indices(), h.block_indices()
    assert np.array_equal(ri, hi)
    _, rt, rs = r.download(ri)
    _, ht, hs = h.download(ri)
    assert np.array_equal(rs["label"], hs["label"])
    assert np.array_equal(rs["priors"].view(np.uint32), hs["priors"].view(np.uint32))
    assert np.array_equal(rt["distance"].view(np.uint32), ht["distance"].view(np.uint32))
    assert np.array_equal(rt["weight"].view(np.uint32), ht["weight"].view(np.uint32))
    assert np.array_equal(rt["color"], ht["color"]) and np.array_equal(rs["color"], hs["color"])


@pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")
def test_three_full_size_c4_frames_on_the_device_vs_real_reference(tmp_path):
    """Three consecutive FULL-SIZE C4 frames (1280x720, 2 cm voxels, 10 m rays), default `fast`: the reference's serial result
    from the DEVICE loop — marks over the rays' views, dense iterations, then the event-driven rounds (csrc/ks_k_exact.h) —
    not from the host-driven one: at most the first frame may repeat on the host (its mark buffers grow from their initial size),
    and the context must still be event-driven afterwards.  HIP == the real reference sources, every voxel."""
    geom = dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=10.0)
    csv = str(tmp_path / "labels.csv")
    R.write_label_csv(csv, synth.default_label_colors())
    r = R.Reference("fast", csv, voxel_size=geom["voxel_size"], truncation=geom["truncation_distance"], max_ray=geom["max_ray_length_m"])
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 17, max_points=1280 * 720, **dict(COMMON, method=0, **geom)))
    sc = synth.make_scene("hall")
    for k in range(3):
        f = synth.render_frame(sc, synth.trajectory_pose(3 + k, radius=3.0), 1280, 720, hfov_deg=75.0, seed=3 + k)
        r.integrate(f.T_G_C, f.xyz, f.rgba)
        st = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        assert st.n_voxel_updates > 1e7
    eo = h.early_out_stats()
    print("three C4 frames, default mode:", eo)
    assert eo["event_driven"] and eo["frames"] == 3 and eo["fallbacks"] <= 1, eo
    ri, hi = r.block_indices(), h.block_indices()
    assert np.array_equal(ri, hi)
    _, rt, rs = r.download(ri)
    _, ht, hs = h.download(ri)
    assert np.array_equal(rs["label"], hs["label"])
    assert np.array_equal(rs["priors"].view(np.uint32), hs["priors"].view(np.uint32))
    assert np.array_equal(rt["distance"].view(np.uint32), ht["distance"].view(np.uint32))
    assert np.array_equal(rt["weight"].view(np.uint32), ht["weight"].view(np.uint32))
    assert np.array_equal(rt["color"], ht["color"]) and np.array_equal(rs["color"], hs["color"])


@pytest.mark.parametrize("pipe", [8, 16])
def test_full_reset_of_the_sets_after_10000_frames_pipelined_with_a_fallback_beside_it(monkeypatch, pipe):
    """ApproxHashSet::resetApproxSet clears the whole table every 10 000 offsets ([K:include/kimera_semantics/
    semantic_tsdf_integrator_fast.h:107]; kFullResetThreshold): with clear_checks_every_n_frames = 1 that is frame 10 000.  A
    pipelined default-mode stream of 10 050 small frames crosses it with frames in flight — reset_set completes the pending tails
    first — and FOUR much larger frames sit right at the crossing: their marks do not fit buffers sized (KS_EXACT_CAP_*) for the
    small ones, so a fallback to the host-driven loop, the growth of the buffers and the drain of the pipeline coincide with the
    full reset.  Every frame's counts and the final map equal the serial oracle's, bit for bit."""
    monkeypatch.setenv("KS_DEBUG", "1")
    monkeypatch.setenv("KS_EXACT_CAP_MARKS", "60000")
    monkeypatch.setenv("KS_EXACT_CAP_X", "2048")
    okw = dict(COMMON, method=0)
    o = O.Oracle(O.default_config(integrator_threads=1, **okw))
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 14, pipeline_frames=pipe, **okw))
    monkeypatch.delenv("KS_EXACT_CAP_MARKS")
    monkeypatch.delenv("KS_EXACT_CAP_X")
    assert h.pipeline_shape()["lag"] == pipe and h.pipeline_shape()["batch"] == (8 if pipe == 16 else 4)
    sc = synth.make_scene("room")
    small = [synth.render_frame(sc, synth.trajectory_pose(k), 24, 18, seed=k) for k in range(8)]   # (the serial oracle is the cost: ~5 ms per such frame)
    big = [synth.render_frame(sc, synth.trajectory_pose(2 * k), 128, 96, seed=500 + k) for k in range(4)]
    n_frames, first_big = 10050, 9998   # the table is cleared when the 10 000th offset is reached
    fallbacks_before = None
    for k in range(n_frames):
        f = big[k - first_big] if first_big <= k < first_big + 4 else small[k % 8]
        if k == first_big:
            fallbacks_before = h.early_out_stats()["fallbacks"]
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    st_h = h.flush()
    st = h.early_out_stats()
    assert st["frames"] == n_frames and st["pipelined"], st
    assert st["fallbacks"] > fallbacks_before, (st, fallbacks_before)   # the large frames did overflow, at the crossing
    compare_maps(o, h, exact=True)
