"""-m gpu: k_apply_runs — the voxel update with ONE LANE PER RUN, a tile's runs bucketed by length (csrc/ks_k_apply.h) — is
what the library uses from 2^20 pairs per frame on, and k_apply_long_lanes (the runs of 33 .. 256 updates, a lane per run, bucketed
by length over the frame) from 2^24 on (the full-size C4 tests reach both by themselves); here they are forced for
frames of every size (KS_DEBUG=1 KS_APPLY_RUNS=1 KS_LONG_LANES=2) and must leave the oracle's records, bit for bit: both integrators, the
three colour modes, mixed-label bundles, runs beside the long-run kernels, 2 cm geometry, frames in flight, and A/B against
k_apply (KS_APPLY_RUNS=0) on the same frames."""
import pytest

from kimera_semantics_amd import binding as B
from kimera_semantics_amd import synth
from oracle import oracle_py as O
from tests.util import COMMON, NO_EARLY_OUT, compare_maps

pytestmark = pytest.mark.gpu


def _pair(monkeypatch, method, force="1", pipe=0, max_tiles=8192, **kw):
    okw = dict(COMMON, method=method, **kw)
    o = O.Oracle(O.default_config(integrator_threads=1, **okw))
    monkeypatch.setenv("KS_DEBUG", "1")
    monkeypatch.setenv("KS_APPLY_RUNS", force)
    monkeypatch.setenv("KS_LONG_LANES", "2" if force == "1" else "0")   # the runs of 33 .. 256 updates a lane per run, whatever the frame's size
    h = B.HipIntegrator(B.default_config(max_tiles=max_tiles, max_points=1 << 18, pipeline_frames=pipe, **okw))
    monkeypatch.delenv("KS_APPLY_RUNS")
    monkeypatch.delenv("KS_LONG_LANES")
    monkeypatch.delenv("KS_DEBUG")
    return o, h


def _run(o, h, frames, exact=True):
    for f in frames:
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    h.flush()
    return compare_maps(o, h, exact=exact)


def _frames(n, w=320, h=240, seed=600, step=4):
    sc = synth.make_scene("room")
    return [synth.render_frame(sc, synth.trajectory_pose(step * k), w, h, seed=seed + k) for k in range(n)]


@pytest.mark.parametrize("method,color_mode,early_out", [(1, 1, False), (1, 0, False), (0, 1, False), (0, 0, False), (0, 1, True)])
def test_lane_per_run_update_is_exact(monkeypatch, method, color_mode, early_out):
    kw = {} if early_out else dict(max_consecutive_ray_collisions=NO_EARLY_OUT)
    o, h = _pair(monkeypatch, method, color_mode=color_mode, **kw)
    rep = _run(o, h, _frames(3))
    assert rep["oracle_touched"] > 10000


def test_lane_per_run_update_probability_colours(monkeypatch):
    o, h = _pair(monkeypatch, 1, color_mode=2)
    rep = _run(o, h, _frames(2), exact=False)   # (colours go through exp(): one LSB allowed on the TSDF colour only)
    assert rep["label_mismatches"] == 0 and rep["max_abs_priors_err"] == 0.0 and rep["max_abs_distance_err"] == 0.0
    assert rep["tsdf_color_mismatches"] <= rep["voxels_compared"] * 1e-3


@pytest.mark.parametrize("method", [0, 1])
def test_lane_per_run_update_beside_the_long_run_kernels(monkeypatch, method):
    """0.45 m from a wall: the voxels next to the sensor are k_apply_long's / k_apply_xlong's (runs of more than 32 updates),
    everything else this kernel's — the split must be the one k_find_long makes."""
    sc = synth.make_scene("room")
    frames = [synth.render_frame(sc, synth.pose_to_T((3.5, 0.3, 1.2), 0.1), 320, 240, seed=7)] + _frames(2)
    o, h = _pair(monkeypatch, method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    _run(o, h, frames)


@pytest.mark.parametrize("method", [0, 1])
def test_lane_per_run_update_c4_geometry_pipelined(monkeypatch, method):
    geom = dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=10.0)
    o, h = _pair(monkeypatch, method, pipe=2, max_tiles=1 << 16, max_consecutive_ray_collisions=NO_EARLY_OUT, **geom)
    rep = _run(o, h, _frames(3, 240, 135))
    assert rep["oracle_touched"] > 10000


def test_both_update_kernels_leave_the_same_map(monkeypatch):
    """k_apply (KS_APPLY_RUNS=0) and k_apply_runs (=1) on the same 640x480 `merged` frames: identical records."""
    frames = _frames(2, 640, 480)
    o, h1 = _pair(monkeypatch, 1, force="1")
    _run(o, h1, frames)
    o2, h0 = _pair(monkeypatch, 1, force="0")
    _run(o2, h0, frames)
    compare_maps(h0, h1, exact=True)


# ---- the runs of more than 1024 updates through integer sums per chunk (csrc/ks_k_apply_xl.h) ----

def _fixed_pose_frames(n, w=320, h=240, seed=700):
    sc = synth.make_scene("room")
    return [synth.render_frame(sc, synth.trajectory_pose(0), w, h, seed=seed + k) for k in range(n)]


@pytest.mark.parametrize("method,kw", [(1, dict(max_weight=2.0)), (1, dict()), (0, dict(max_weight=2.0, max_consecutive_ray_collisions=NO_EARLY_OUT)),
                                       (1, dict(max_weight=2.0, color_mode=2)), (1, dict(max_weight=2.0, semantic_measurement_probability=0.9, dynamic_labels=[]))])
def test_sensor_voxel_runs_as_integer_sums_are_exact(monkeypatch, method, kw):
    """Five frames from ONE pose: the voxel that holds the sensor (and its neighbours) collects a run of thousands of updates per
    frame; once its weight sits at max_weight and its distance at the truncation, its 21 class sums go through the chunked
    integer sums (fl(p + x) = -(M + R) u inside a binade) instead of 21 serial chains.  Bit for bit the oracle's records — also where the
    sums cross binades within a run, with mixed-label bundles, and through the exp() colours (one LSB allowed there)."""
    okw = dict(COMMON, method=method)
    okw.update(kw)
    o = O.Oracle(O.default_config(integrator_threads=1, **okw))
    monkeypatch.setenv("KS_DEBUG", "1")
    monkeypatch.setenv("KS_XL_PARALLEL", "2")   # (the library takes this path from 2^23 pairs per frame on)
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 18, **okw))
    monkeypatch.delenv("KS_XL_PARALLEL")
    monkeypatch.delenv("KS_DEBUG")
    prob = okw.get("color_mode") == 2
    rep = _run(o, h, _fixed_pose_frames(5), exact=not prob)
    if prob:
        assert rep["label_mismatches"] == 0 and rep["max_abs_priors_err"] == 0.0 and rep["max_abs_distance_err"] == 0.0
    st = h.update_stats()
    print("update_stats", st)
    if "max_weight" in kw:
        assert st["walked"] > 0 and st["chunks"] > st["replayed"], st


def test_sensor_voxel_runs_both_ways_leave_the_same_map(monkeypatch):
    frames = _fixed_pose_frames(4, 640, 480)
    okw = dict(COMMON, method=1, max_weight=50.0)
    maps = []
    for par in ("2", "0"):
        monkeypatch.setenv("KS_DEBUG", "1")
        monkeypatch.setenv("KS_XL_PARALLEL", par)
        h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 19, pipeline_frames=2, **okw))
        monkeypatch.delenv("KS_XL_PARALLEL")
        monkeypatch.delenv("KS_DEBUG")
        for f in frames:
            h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        h.flush()
        st = h.update_stats()
        assert (st["walked"] > 0) == (par == "2"), st
        maps.append(h)
    compare_maps(maps[0], maps[1], exact=True)
