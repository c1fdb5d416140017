"""Hand-derived known-answer tests that pin the CPU oracle (the reference has no tests of its
own — SURVEY.md §4 — so these are the repo's golden vectors for the pure functions)."""
import math

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from kimera_semantics_amd import synth
from oracle import oracle_py as O
from tests.util import COMMON

f32 = np.float32


def test_log_likelihood_matrix():
    # semantic_integrator_base.cpp:93-128: off-diagonal log(1-p), diagonal log(p), column 0 zero
    L = O.log_likelihood(0.8)
    lp = f32(math.log(float(f32(0.8))))
    lnp = f32(math.log(float(f32(1.0) - f32(0.8))))
    assert L.shape == (21, 21)
    assert np.all(L[:, 0] == 0.0)
    for i in range(21):
        for j in range(1, 21):
            assert L[i, j] == (lp if i == j else lnp)
    assert lp > lnp
    # invalid probabilities are rejected like the CHECKs
    assert np.isnan(O.log_likelihood(1.0)).all()
    assert np.isnan(O.log_likelihood(0.4)).all()  # log p must exceed log(1-p)


def test_index_hash_known_values():
    # AnyIndexHash/LongIndexHash: x + y*17191 + z*17191^2 truncated to 32 bits
    assert O.long_index_hash([1, 0, 0]) == 1
    assert O.long_index_hash([0, 1, 0]) == 17191
    assert O.long_index_hash([0, 0, 1]) == 17191 * 17191
    assert O.long_index_hash([-1, 0, 0]) == 2 ** 32 - 1
    assert O.long_index_hash([3, -2, 5]) == (3 - 2 * 17191 + 5 * 17191 ** 2) % 2 ** 32


def test_grid_index_negative_and_boundary():
    # floor(p * inv + 1e-6) in f32
    assert O.grid_index_from_point([-0.01, 0.0, 0.0499], 20.0).tolist() == [-1, 0, 0]
    assert O.grid_index_from_point([0.05, -0.05, 0.1], 20.0).tolist() == [1, -1, 2]
    assert O.grid_index_from_point([-1e-8, 1e-8, -0.050001], 20.0).tolist() == [0, 0, -2]


def test_mixed_index_permutation():
    # upstream form: number_of_groups_ = n / 1024 = 2 groups of step_size_ = 1024 points, 5 stragglers
    n = 2048 + 5
    seq = [O.mixed_index(s, n) for s in range(n)]
    assert sorted(seq) == list(range(n))
    assert seq[:4] == [0, 1024, 1, 1025] and seq[2046:2048] == [1023, 2047]
    assert seq[2048:] == [2048, 2049, 2050, 2051, 2052]
    assert [O.mixed_index(s, 10) for s in range(10)] == list(range(10))  # fewer points than one group
    assert O.mixed_chains(n) == 2 and O.mixed_chains(10) == 1024 and O.mixed_chains(640 * 480) == 300
    # the other reading (rounds 1-4): 1024 groups of n / 1024 = 2 points
    seq = [O.mixed_index(s, n, O.ORDER_MIXED_1024_GROUPS) for s in range(n)]
    assert sorted(seq) == list(range(n))
    assert seq[0] == 0 and seq[1] == 2 and seq[1024] == 1 and seq[1025] == 3
    assert seq[2048:] == [2048, 2049, 2050, 2051, 2052]
    assert O.mixed_chains(n, O.ORDER_MIXED_1024_GROUPS) == 1024


def test_transform_point_quaternion():
    # 90 deg about z: (1,0,0) -> (0,1,0), then translate
    s = math.sqrt(0.5)
    T = [s, 0, 0, s, 1.0, 2.0, 3.0]
    out = O.transform_point(T, [1.0, 0.0, 0.0])
    np.testing.assert_allclose(out, [1.0, 3.0, 3.0], atol=1e-6)
    # identity is exact
    assert O.transform_point([1, 0, 0, 0, 0, 0, 0], [0.3, -0.2, 5.0]).tolist() == [f32(0.3), f32(-0.2), f32(5.0)]


def test_dda_hand_derived():
    # start (0.5,0.5,0.5) -> end (2.5,1.5,0.6) in voxel units: t_to_next = (.25,.5,5), t_step = (.5,1,10)
    vox = O.cast_ray([0.5, 0.5, 0.5], [2.5, 1.5, 0.6], carving=True, voxel_size_inv=1.0, truncation=0.0)
    assert vox.tolist() == [[0, 0, 0], [1, 0, 0], [1, 1, 0], [2, 1, 0]]
    # reversed direction (fast casts surface -> origin) visits a (generally different) 4-voxel chain
    rev = O.cast_ray([0.5, 0.5, 0.5], [2.5, 1.5, 0.6], carving=True, voxel_size_inv=1.0, truncation=0.0,
                     cast_from_origin=False)
    assert rev[0].tolist() == [2, 1, 0] and rev[-1].tolist() == [0, 0, 0] and len(rev) == 4


def test_dda_axis_aligned_quirk_is_preserved():
    # A zero ray component divides by zero upstream (the "|r| < 0" guard is dead code): the
    # -inf / NaN t-values make the caster emit the start voxel three times and stop two voxels
    # short.  The oracle (and the GPU) reproduce this literally.
    vox = O.cast_ray([0.025, 0.025, 0.025], [0.525, 0.025, 0.025], carving=True, voxel_size_inv=20.0, truncation=0.2)
    assert len(vox) == 15
    assert vox[:3].tolist() == [[0, 0, 0]] * 3
    assert vox[-1].tolist() == [12, 0, 0]


def test_clearing_ray_and_no_carving():
    # clearing: ends trunc before the point, capped at max_ray_length
    v = O.cast_ray([0.5, 0.5, 0.5], [10.5, 0.6, 0.7], is_clearing=True, max_ray_length_m=5.0, voxel_size_inv=1.0,
                   truncation=1.0)
    assert v[0].tolist() == [0, 0, 0] and v[-1][0] == 5
    # no carving: only the truncation band around the surface
    v = O.cast_ray([0.5, 0.5, 0.5], [10.5, 0.6, 0.7], carving=False, voxel_size_inv=1.0, truncation=1.0)
    assert v[0][0] == 9 and v[-1][0] == 11


@settings(max_examples=200, deadline=None)
@given(st.lists(st.floats(-20, 20, allow_nan=False, width=32), min_size=6, max_size=6))
def test_dda_properties(c):
    o, p = c[:3], c[3:]
    d = np.array(p, dtype=f32) - np.array(o, dtype=f32)
    if np.any(np.abs(d) < 1e-3):
        return  # axis-aligned degenerate case is pinned separately
    vox = O.cast_ray(o, p, carving=True, voxel_size_inv=20.0, truncation=0.2)
    diff = np.abs(np.diff(vox, axis=0))
    assert np.all(diff.sum(axis=1) == 1)  # 6-connected chain
    assert len(vox) == np.abs(vox[-1] - vox[0]).sum() + 1  # L1 distance + 1
    start = O.grid_index_from_point(o, 20.0)
    assert vox[0].tolist() == start.tolist()


def test_update_tsdf_voxel_known_answers():
    cfg = O.default_config()
    # in front of the surface: sdf = 1 - 0.525 = 0.475 -> clamped to +trunc, no colour blend
    d, w, c = O.update_tsdf_voxel(cfg, [0, 0, 0], [1, 0, 0], [10, 0, 0], [200, 100, 50, 255], 1.0, 0.0, 0.0, [0, 0, 0, 0])
    assert d == pytest.approx(0.2, abs=1e-7) and w == 1.0 and c.tolist() == [0, 0, 0, 0]
    # behind the surface with drop-off: centre 1.125 -> sdf -0.125 < -0.05 -> w = (0.2-0.125)/(0.2-0.05) = 0.5
    d, w, c = O.update_tsdf_voxel(cfg, [0, 0, 0], [1, 0, 0], [22, 0, 0], [200, 100, 50, 255], 1.0, 0.0, 0.0, [0, 0, 0, 0])
    assert d == pytest.approx(-0.125, abs=1e-6) and w == pytest.approx(0.5, rel=1e-5)
    assert c.tolist() == [200, 100, 50, 255]  # blended with zero prior weight
    # weighted running mean + max_weight clamp
    d, w, _ = O.update_tsdf_voxel(cfg, [0, 0, 0], [1, 0, 0], [19, 0, 0], [0, 0, 0, 0], 1.0, 0.1, 9999.5, [0, 0, 0, 0])
    assert w == 10000.0 and d == pytest.approx((0.025 * 1 + 0.1 * 9999.5) / 10000.5, rel=1e-5)
    # vanishing weight: voxel untouched
    d, w, _ = O.update_tsdf_voxel(cfg, [0, 0, 0], [1, 0, 0], [19, 0, 0], [0, 0, 0, 0], 1e-9, 0.1, 0.0, [0, 0, 0, 0])
    assert (d, w) == (pytest.approx(0.1), 0.0)


def test_blend_two_colors_rounding():
    assert O.blend_two_colors([10, 0, 0, 0], 1.0, [11, 0, 0, 0], 1.0).tolist() == [11, 0, 0, 0]  # 10.5 rounds away from zero
    assert O.blend_two_colors([0, 0, 0, 0], 0.0, [9, 8, 7, 255], 0.25).tolist() == [9, 8, 7, 255]
    assert O.blend_two_colors([100, 100, 100, 100], 3.0, [200, 0, 50, 255], 1.0).tolist() == [125, 75, 88, 139]


def test_rainbow_color_map():
    assert O.rainbow_color_map(0.0).tolist() == [255, 0, 0, 255]
    assert O.rainbow_color_map(1.0 / 3.0).tolist()[1] == 255
    assert O.rainbow_color_map(0.5).tolist() == [0, 255, 255, 255]


def _one_point(label, xyz=(0.013, 0.021, 1.0)):  # off-axis: exact axis-aligned rays hit the upstream 0/0 quirk
    return np.array([xyz], dtype=f32), np.array([synth.default_label_colors()[label]], dtype=np.uint8), np.array([label], dtype=np.uint8)


def test_semantic_update_order_and_argmax_tie():
    """One observation of label 5 then one of label 9 on the same ray: both priors hold the
    same multiset of addends in a different order; the label is whatever f32 rounding makes it,
    with ties going to the LOWEST index (Eigen maxCoeff)."""
    cfg = O.default_config(**dict(COMMON, method=0))
    o = O.Oracle(cfg)
    T = np.array([1, 0, 0, 0, 0, 0, 0], dtype=f32)
    for lab in (5, 9):
        x, c, l = _one_point(lab)
        st_ = o.integrate(T, x, c, l)
        assert st_.n_rays_cast == 1
    _, t, s = o.download()
    touched = s["label"] != 0
    init = f32(-0.60205999132)
    lp, lnp = f32(math.log(float(f32(0.8)))), f32(math.log(float(f32(1) - f32(0.8))))
    p5 = f32(f32(init + lp) + lnp)
    p9 = f32(f32(init + lnp) + lp)
    other = f32(f32(init + lnp) + lnp)
    pri = s["priors"][touched]
    assert pri.shape[0] > 10
    assert np.all(pri[:, 5] == p5) and np.all(pri[:, 9] == p9) and np.all(pri[:, 1] == other)
    expect = 5 if p5 >= p9 else 9
    assert np.all(s["label"][touched] == expect)
    assert np.all(t["color"][touched] == synth.default_label_colors()[expect])  # ColorMode::kSemantic


def test_unknown_label_changes_nothing_but_marks_voxel():
    cfg = O.default_config(**dict(COMMON, method=0))
    o = O.Oracle(cfg)
    T = np.array([1, 0, 0, 0, 0, 0, 0], dtype=f32)
    x, c, l = _one_point(0)
    o.integrate(T, x, c, l)
    _, t, s = o.download()
    upd = t["weight"] > 0
    assert upd.sum() > 10
    assert np.all(s["priors"][upd] == f32(-0.60205999132))  # column 0 of L is zero
    assert np.all(s["label"][upd] == 0)
    assert np.all(s["color"][upd] == [255, 255, 255, 255])  # id 0 -> White (color.cpp:64-66)
    assert np.all(s["color"][~upd] == [127, 127, 127, 255])  # untouched voxels stay Gray


def test_fast_start_voxel_dedup_and_dynamic_labels():
    cfg = O.default_config(**dict(COMMON, method=0))
    o = O.Oracle(cfg)
    T = np.array([1, 0, 0, 0, 0, 0, 0], dtype=f32)
    # two points in the same 2.5 cm start cell, one in another cell, one dynamic (label 20), one too close
    xyz = np.array([[0.013, 0.021, 1.0], [0.014, 0.022, 1.001], [0.2, 0.01, 1.0], [0.4, 0.01, 1.0], [0.001, 0.002, 0.05]], dtype=f32)
    lab = np.array([3, 3, 3, 20, 3], dtype=np.uint8)
    st_ = o.integrate(T, xyz, synth.default_label_colors()[lab], lab)
    assert st_.n_valid_points == 3  # dynamic + too-close dropped
    assert st_.n_rays_cast == 2      # second point deduplicated


def test_merged_bundle_weighted_mean_and_histogram():
    cfg = O.default_config(**dict(COMMON, method=1))
    o = O.Oracle(cfg)
    T = np.array([1, 0, 0, 0, 0, 0, 0], dtype=f32)
    xyz = np.array([[0.01, 0.01, 1.01], [0.02, 0.02, 1.02], [0.03, 0.01, 1.03]], dtype=f32)  # same 5 cm voxel
    lab = np.array([4, 4, 7], dtype=np.uint8)
    st_ = o.integrate(T, xyz, None, lab)
    assert st_.n_rays_cast == 1
    _, t, s = o.download()
    upd = t["weight"] > 0
    lp, lnp = f32(math.log(float(f32(0.8)))), f32(math.log(float(f32(1) - f32(0.8))))
    init = f32(-0.60205999132)
    # priors += L*freq, j ascending: label 4 seen twice, label 7 once
    p4 = f32(init + f32(f32(lp * f32(2)) + f32(lnp * f32(1))))
    p7 = f32(init + f32(f32(lnp * f32(2)) + f32(lp * f32(1))))
    p1 = f32(init + f32(f32(lnp * f32(2)) + f32(lnp * f32(1))))
    pri = s["priors"][upd]
    assert np.all(pri[:, 4] == p4) and np.all(pri[:, 7] == p7) and np.all(pri[:, 1] == p1)
    assert np.all(s["label"][upd] == 4)


def test_label_out_of_range_is_an_error():
    o = O.Oracle(O.default_config(**dict(COMMON, method=0)))
    x, c, _ = _one_point(3)
    with pytest.raises(RuntimeError):
        o.integrate(np.array([1, 0, 0, 0, 0, 0, 0], dtype=f32), x, c, np.array([21], dtype=np.uint8))


def test_empty_cloud():
    for m in (0, 1):
        o = O.Oracle(O.default_config(**dict(COMMON, method=m)))
        st_ = o.integrate(np.array([1, 0, 0, 0, 0, 0, 0], dtype=f32), np.zeros((0, 3), f32), np.zeros((0, 4), np.uint8),
                          np.zeros((0,), np.uint8))
        assert st_.n_voxel_updates == 0 and len(o.block_indices()) == 0


def test_merged_reference_order_vs_canonical_order_same_sets():
    """std::unordered_map iteration order (what the reference does) vs first-insertion order
    (what the GPU reproduces): identical voxel sets / counts, float state differs only by
    summation order."""
    from tests.util import small_frame
    f = small_frame(seed=4, w=96, h=72)
    a = O.Oracle(O.default_config(**dict(COMMON, method=1, bundle_order=0)))
    b = O.Oracle(O.default_config(**dict(COMMON, method=1, bundle_order=1)))
    sa = a.integrate(f.T_G_C, f.xyz, None, f.labels)
    sb = b.integrate(f.T_G_C, f.xyz, None, f.labels)
    assert (sa.n_rays_cast, sa.n_voxel_updates) == (sb.n_rays_cast, sb.n_voxel_updates)
    ia, ta, sa_ = a.download()
    ib, tb, sb_ = b.download()
    assert np.array_equal(ia, ib)
    assert np.array_equal(ta["weight"] > 0, tb["weight"] > 0)
    # The clamp to +-truncation is applied after every update, so the running mean is order
    # dependent for the few voxels that see both free-space and surface observations (the
    # multi-threaded reference has the same spread); everywhere else only rounding differs.
    dd = np.abs(ta["distance"] - tb["distance"])
    assert (dd > 1e-4).mean() < 0.02 and np.median(dd) < 1e-6
    dp = np.abs(sa_["priors"] - sb_["priors"])
    assert np.all(dp <= 1e-4 * np.abs(sa_["priors"]) + 1e-3)  # f32 summation order only
    mism = (sa_["label"] != sb_["label"]).sum()
    assert mism <= 0.005 * sa_["label"].size  # only exact count-ties may flip
