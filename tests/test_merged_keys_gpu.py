"""-m gpu: `merged` stage A groups the points of a frame by end voxel with a 32-bit key — the voxel relative to a window around
the sensor that holds every point within max_ray, anything else through a hash table (FrameParams::key_bits, csrc/ks_k_rays.h) —
sorted in four passes instead of the eight of the 64-bit end-voxel keys.  What comes out must be the oracle's map bit for bit:
with the default window, with a window of 3 bits per axis (nearly every voxel through the table, frames in flight), with short
rays (clearing points far outside the window), and A/B against the 64-bit keys (KS_KEY_WINDOW_BITS=0)."""
import numpy as np
import pytest

from kimera_semantics_amd import binding as B
from kimera_semantics_amd import synth
from oracle import oracle_py as O
from tests.util import COMMON, compare_maps

pytestmark = pytest.mark.gpu


def _pair(monkeypatch, bits=None, pipe=0, **kw):
    okw = dict(COMMON, method=1, **kw)
    o = O.Oracle(O.default_config(integrator_threads=1, **okw))
    if bits is not None:
        monkeypatch.setenv("KS_DEBUG", "1")
        monkeypatch.setenv("KS_KEY_WINDOW_BITS", str(bits))
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 19, pipeline_frames=pipe, **okw))
    if bits is not None:
        monkeypatch.delenv("KS_KEY_WINDOW_BITS")
        monkeypatch.delenv("KS_DEBUG")
    return o, h


def _frames(n, w=320, h=240, seed=900, step=4):
    sc = synth.make_scene("room")
    return [synth.render_frame(sc, synth.trajectory_pose(step * k), w, h, seed=seed + k) for k in range(n)]


def _run(o, h, frames):
    for f in frames:
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    h.flush()
    return compare_maps(o, h, exact=True)


@pytest.mark.parametrize("bits,pipe", [(None, 0), (None, 4), (3, 0), (3, 4), (1, 0), (0, 0)])
def test_grouping_keys_leave_the_oracles_map(monkeypatch, bits, pipe):
    o, h = _pair(monkeypatch, bits, pipe)
    rep = _run(o, h, _frames(3))
    assert rep["oracle_touched"] > 10000
    h.close()


def test_clearing_points_far_outside_the_window(monkeypatch):
    """max_ray 1.5 m in a 6 m room: most points are clearing points whose end voxels lie outside the window of +-1.5 m; a few
    points at 500 m on top (one clearing bundle each)."""
    frames = _frames(3)
    for f in frames:
        f.xyz[::997] *= np.float32(100.0)
    o, h = _pair(monkeypatch, None, 0, max_ray_length_m=1.5)
    rep = _run(o, h, frames)
    assert rep["oracle_touched"] > 1000
    h.close()


def test_full_size_frames_both_key_forms_same_map(monkeypatch):
    frames = _frames(2, 640, 480)
    o, h32 = _pair(monkeypatch, None, 4)
    _run(o, h32, frames)
    o2, h64 = _pair(monkeypatch, 0, 4)
    _run(o2, h64, frames)
    assert compare_maps(o, h64, exact=True)["voxels_compared"] > 10000
    h32.close()
    h64.close()


def _pair_env(monkeypatch, env, pipe=0, **kw):
    okw = dict(COMMON, method=1, **kw)
    o = O.Oracle(O.default_config(integrator_threads=1, **okw))
    monkeypatch.setenv("KS_DEBUG", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 19, pipeline_frames=pipe, **okw))
    for k in env:
        monkeypatch.delenv(k)
    monkeypatch.delenv("KS_DEBUG")
    return o, h


@pytest.mark.parametrize("hint,pipe", [("1", 0), ("1", 4), ("12000", 4), ("0", 0)])
def test_bundle_order_epochs_beyond_the_hint(monkeypatch, hint, pipe):
    """The rehash recurrence of the bundle order is launched for the bundle counts of the frames before (a hint); a frame with more
    bundles finishes in k_bo_rest.  640x480 frames have 1-1.5e4 bundles: hint 1 = every epoch past k_bo_small through k_bo_rest,
    12000 = the last one or two, 0 = the launches a frame of n points could need (rounds 3-5)."""
    o, h = _pair_env(monkeypatch, {"KS_BO_HINT": hint}, pipe)
    rep = _run(o, h, _frames(3, 640, 480))
    assert rep["oracle_touched"] > 10000
    h.close()


def test_bundle_order_hint_follows_the_stream(monkeypatch):
    """No fixed hint: the first frames are launched for n points, later ones for the counts seen; a jump from 160x120 frames to
    640x480 ones (10x the bundles) goes through k_bo_rest once and through the launches afterwards."""
    okw = dict(COMMON, method=1)
    o = O.Oracle(O.default_config(integrator_threads=1, **okw))
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 19, pipeline_frames=0, **okw))
    frames = _frames(3, 160, 120, seed=40) + _frames(3, 640, 480, seed=50) + _frames(2, 160, 120, seed=60)
    rep = _run(o, h, frames)
    assert rep["oracle_touched"] > 10000
    h.close()


def test_point_buffers_grow_with_the_key_table(monkeypatch):
    """A context created for 1000 points takes 160x120, then 320x240 frames: stage A's buffers — the key table and the bundle
    order's slab among them — are re-allocated between frames; window of 3 bits so that the table is in use when it happens."""
    okw = dict(COMMON, method=1)
    o = O.Oracle(O.default_config(integrator_threads=1, **okw))
    monkeypatch.setenv("KS_DEBUG", "1")
    monkeypatch.setenv("KS_KEY_WINDOW_BITS", "3")
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1000, pipeline_frames=4, **okw))
    monkeypatch.delenv("KS_KEY_WINDOW_BITS")
    monkeypatch.delenv("KS_DEBUG")
    rep = _run(o, h, _frames(2, 160, 120, seed=70) + _frames(2, 320, 240, seed=80) + _frames(1, 160, 120, seed=90))
    assert rep["oracle_touched"] > 10000
    h.close()
