"""CPU tier: the frame source of the bag-replay path (kimera_semantics_amd/frame_source.py):
tf composition T_G_C = T_G_B * T_B_C (kimera_semantics_rosbag.cpp:124-134), tf lookup with interpolation and no
extrapolation, the ROS1 bag reader against bags written on the fly (uncompressed and bz2 chunks), the stamp CHECK."""
import numpy as np
import pytest

from kimera_semantics_amd import frame_source as FS
from kimera_semantics_amd import synth
from oracle import oracle_py as O


def test_compose_equals_applying_the_transforms_one_after_the_other():
    rng = np.random.default_rng(0)
    for _ in range(50):
        def rand_T():
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            return np.concatenate([q, rng.uniform(-3, 3, size=3)]).astype(np.float32)
        A, Bc = rand_T(), rand_T()
        C = FS.compose(A, Bc)
        assert abs(np.linalg.norm(C[:4]) - 1.0) < 1e-5
        p = rng.uniform(-2, 2, size=3).astype(np.float32)
        direct = O.transform_point(C, p)                       # the oracle's minkindr arithmetic
        chained = O.transform_point(A, O.transform_point(Bc, p))
        assert np.allclose(direct, chained, atol=2e-5)
        I = FS.compose(A, FS.inverse(A))
        assert np.allclose(I, [1, 0, 0, 0, 0, 0, 0], atol=1e-5) or np.allclose(I, [-1, 0, 0, 0, 0, 0, 0], atol=1e-5)


def test_tf_buffer_interpolates_and_never_extrapolates():
    tf = FS.TfBuffer()
    T0 = synth.pose_to_T((0.0, 0.0, 1.0), 0.0)
    T1 = synth.pose_to_T((1.0, 2.0, 1.0), np.pi / 2)
    tf.set_transform(1000, "world", "base", T0)
    tf.set_transform(2000, "world", "base", T1)
    tf.set_transform(0, "base", "cam", np.array([1, 0, 0, 0, 0.1, 0, 0], np.float32), static=True)
    assert np.array_equal(tf.lookup("world", "base", 1000), T0)
    mid = tf.lookup("world", "base", 1500)
    assert np.allclose(mid[4:7], [0.5, 1.0, 1.0], atol=1e-6)
    # rotation half-way between the two: slerp keeps it on the geodesic
    assert abs(abs(np.dot(mid[:4], T0[:4])) - abs(np.dot(mid[:4], T1[:4]))) < 1e-5
    assert tf.lookup("world", "base", 999) is None and tf.lookup("world", "base", 2001) is None
    chain = tf.lookup("world", "cam", 1000)
    assert np.allclose(chain, FS.compose(T0, np.array([1, 0, 0, 0, 0.1, 0, 0], np.float32)), atol=1e-6)
    assert tf.lookup("world", "nowhere", 1000) is None


@pytest.mark.parametrize("compression", ["none", "bz2"])
def test_rosbag_round_trip(tmp_path, compression):
    seq = FS.synthetic_sequence(5, width=64, height=48, tf_rate_divisor=2)
    path = str(tmp_path / "demo.bag")
    FS.write_bag(path, FS.sequence_to_messages(seq), compression=compression, chunk_messages=3)
    got = FS.read_rosbag(path, "/depth", "/semantic", "/camera_info", "left_cam")
    assert len(got.frames) == 5
    assert np.allclose(got.T_B_C, seq.T_B_C)
    for a, b in zip(seq.frames, got.frames):
        assert a.stamp_ns == b.stamp_ns == b.semantic_stamp_ns
        assert np.array_equal(np.nan_to_num(a.depth, nan=-1), np.nan_to_num(b.depth, nan=-1))
        assert np.array_equal(a.semantic_rgba, b.semantic_rgba)
        assert np.allclose(a.K, b.K)
        Ta, Tb = seq.tf.lookup("world", "base_link_gt", a.stamp_ns), got.tf.lookup("world", "base_link_gt", a.stamp_ns)
        assert (Ta is None) == (Tb is None)
        if Ta is not None:
            assert np.allclose(Ta, Tb, atol=1e-6)


class _Recorder:
    def __init__(self):
        self.calls = []

    def integrate_depth(self, T, depth, K, label_img=None, rgba_img=None, freespace=False):
        self.calls.append((np.array(T), depth.shape, label_img is not None, rgba_img is not None))
        return None


def test_replay_composes_poses_skips_frames_without_tf_and_checks_stamps():
    seq = FS.synthetic_sequence(6, width=32, height=24, tf_rate_divisor=4)   # poses at frames 0, 4, 5: 1..3 interpolated
    rec = _Recorder()
    out = FS.replay(seq, rec)
    assert out == {"integrated": 6, "skipped_no_tf": 0}
    for k, (T, shape, has_lab, _) in enumerate(rec.calls):
        want = synth.trajectory_pose(k)
        assert shape == (24, 32) and has_lab
        if k in (0, 4, 5):     # published poses: T_G_B * T_B_C gives the camera pose back
            assert np.allclose(T[4:7], want[4:7], atol=1e-5) and abs(abs(np.dot(T[:4], want[:4])) - 1) < 1e-5
        else:                  # interpolated between published poses: close to the true one on a smooth trajectory
            assert np.linalg.norm(T[4:7] - want[4:7]) < 0.02
    # a frame outside the tf interval is skipped, as the reference does ("Couldn't find tf ...")
    seq.frames[0].stamp_ns -= 10_000_000
    rec2 = _Recorder()
    assert FS.replay(seq, rec2) == {"integrated": 5, "skipped_no_tf": 1}
    seq.frames[2].semantic_stamp_ns = seq.frames[2].stamp_ns + 1
    with pytest.raises(ValueError, match="timestamps do not match"):
        FS.replay(seq, _Recorder())


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1])
def test_bag_replay_through_the_gpu_integrator_equals_the_oracle(tmp_path, method):
    """A bag written with the demo layout (depth, colour-coded semantic image, CameraInfo, /tf, /tf_static), read back
    by the pure-Python reader, replayed through compose() + ks_integrate_depth (colour -> label on the GPU): the map
    equals the oracle's, fed the reference way (back-projected finite cloud, labels from the colours, T_G_B * T_B_C)."""
    from kimera_semantics_amd import binding as B
    from tests.util import COMMON, NO_EARLY_OUT, compare_maps
    seq0 = FS.synthetic_sequence(4, width=128, height=96)
    path = str(tmp_path / "demo.bag")
    FS.write_bag(path, FS.sequence_to_messages(seq0), compression="bz2")
    seq = FS.read_rosbag(path, "/depth", "/semantic", "/camera_info", "left_cam")
    kw = dict(COMMON, method=method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    lut = synth.default_label_colors()
    h.set_color_to_label(lut[:21], np.arange(21, dtype=np.uint8))
    o = O.Oracle(O.default_config(**kw))
    poses = []
    out = FS.replay(seq, h, on_frame=lambda fr, T, st: poses.append(T))
    assert out == {"integrated": 4, "skipped_no_tf": 0}
    color_to_label = {tuple(int(x) for x in lut[i]): i for i in range(21)}
    for fr, T in zip(seq.frames, poses):
        pts = synth.backproject(fr.depth, fr.K).reshape(-1, 3)
        ok = np.isfinite(pts).all(axis=1)
        rgba = fr.semantic_rgba.reshape(-1, 4)[ok]
        labels = np.array([color_to_label.get(tuple(int(x) for x in c), 0) for c in rgba], dtype=np.uint8)
        o.integrate(T, np.ascontiguousarray(pts[ok]), None if method == 1 else np.ascontiguousarray(rgba), labels)
    compare_maps(o, h, exact=True)
