"""Pins the Kimera half of the oracle: the REAL reference sources
(/root/reference/kimera_semantics/src/{semantic_integrator_base,semantic_tsdf_integrator_fast,
semantic_tsdf_integrator_merged,semantic_tsdf_integrator_factory,color,csv_iterator}.cpp),
compiled into oracle/_ref/libks_ref.so, must produce bit-identical maps to the oracle's
restatement on the same inputs (single-threaded, where the reference is deterministic).
Both sit on the same restated Voxblox primitives, so this checks everything that lives
under /root/reference; the Voxblox half stays 'parity unpinned' (no upstream sources here)."""
import numpy as np
import pytest

from kimera_semantics_amd import synth
from oracle import oracle_py as O
from oracle import ref_py as R
from tests.util import COMMON, small_frame

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference at build time)")


def _csv(tmp_path):
    p = str(tmp_path / "labels.csv")
    R.write_label_csv(p, synth.default_label_colors())
    return p


def _same(o, r):
    oi, ri = o.block_indices(), r.block_indices()
    assert np.array_equal(oi, ri)
    assert r.n_semantic_blocks() == len(ri)
    _, ot, os_ = o.download(oi)
    _, rt, rs = r.download(oi)
    assert np.array_equal(os_["label"], rs["label"])
    assert np.array_equal(os_["priors"].view(np.uint32), rs["priors"].view(np.uint32))
    assert np.array_equal(os_["color"], rs["color"])
    assert np.array_equal(ot["distance"].view(np.uint32), rt["distance"].view(np.uint32))
    assert np.array_equal(ot["weight"].view(np.uint32), rt["weight"].view(np.uint32))
    assert np.array_equal(ot["color"], rt["color"])
    return int((ot["weight"] > 0).sum())


@pytest.mark.parametrize("mc", [2, 1 << 30])
@pytest.mark.parametrize("color_mode", [1, 0, 2])
def test_fast_matches_reference(tmp_path, mc, color_mode):
    csv = _csv(tmp_path)
    o = O.Oracle(O.default_config(**dict(COMMON, method=0, max_consecutive_ray_collisions=mc, color_mode=color_mode)))
    r = R.Reference("fast", csv, max_consecutive_ray_collisions=mc, color_mode=color_mode)
    sc = synth.make_scene("room")
    for k in range(3):  # several frames: exercises the approximate-set reset across frames
        f = synth.render_frame(sc, synth.trajectory_pose(5 * k), 96, 72, seed=40 + k)
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        r.integrate(f.T_G_C, f.xyz, f.rgba)
    assert _same(o, r) > 1000


@pytest.mark.parametrize("color_mode", [1, 0])
def test_merged_matches_reference(tmp_path, color_mode):
    csv = _csv(tmp_path)
    # reference bundle order = std::unordered_map iteration order; colours are all-zero in the
    # reference's colour overload (hash_colors is never filled, merged.cpp:70,92-93)
    o = O.Oracle(O.default_config(**dict(COMMON, method=1, bundle_order=0, color_mode=color_mode)))
    r = R.Reference("merged", csv, color_mode=color_mode)
    sc = synth.make_scene("room")
    for k in range(2):
        f = synth.render_frame(sc, synth.trajectory_pose(5 * k), 96, 72, seed=50 + k)
        o.integrate(f.T_G_C, f.xyz, None, f.labels)
        r.integrate(f.T_G_C, f.xyz, f.rgba)
    assert _same(o, r) > 1000


def test_sorted_order_and_close_up_match_reference(tmp_path):
    csv = _csv(tmp_path)
    f = synth.render_frame(synth.make_scene("room"), synth.pose_to_T((3.5, 0.3, 1.2), 0.1), 96, 72, seed=7)
    for method, name in ((0, "fast"), (1, "merged")):
        o = O.Oracle(O.default_config(**dict(COMMON, method=method, bundle_order=0, integration_order_mode=1)))
        r = R.Reference(name, csv, order_mode="sorted")
        o.integrate(f.T_G_C, f.xyz, f.rgba if method == 0 else None, f.labels)
        r.integrate(f.T_G_C, f.xyz, f.rgba)
        _same(o, r)


def test_mixed_order_both_forms_match_the_index_behind_the_reference(tmp_path):
    """Voxblox's MixedThreadSafeIndex is not in the reference tree: both readings of it exist on every side.  The sequence
    the shim hands the REAL Kimera sources (ThreadSafeIndexFactory::get("mixed", ...)) equals the oracle's closed form in
    either setting, and whole maps (fast with the early-out = the order-sensitive case, and merged) agree bit for bit."""
    import numpy as np
    for form, mode in ((0, O.ORDER_MIXED), (1, O.ORDER_MIXED_1024_GROUPS)):
        R.set_mixed_order_form(form)
        try:
            for n in (5 * 1024 + 7, 1023, 1024, 2048, 640 * 480):
                seq = R.mixed_sequence(n)
                assert sorted(seq.tolist()) == list(range(n))
                step = max(1, n // 5000)
                assert all(int(seq[s]) == O.mixed_index(s, n, mode) for s in list(range(0, n, step)) + [n - 1]), (form, n)
        finally:
            R.set_mixed_order_form(0)
    assert O.mixed_index(1, 5 * 1024, O.ORDER_MIXED) == 1024 and O.mixed_index(1, 5 * 1024, O.ORDER_MIXED_1024_GROUPS) == 5
    csv = _csv(tmp_path)
    sc = synth.make_scene("room")
    for method, name in ((0, "fast"), (1, "merged")):
        maps = []
        for mode, rname in ((0, "mixed"), (2, "mixed_1024_groups")):
            o = O.Oracle(O.default_config(**dict(COMMON, method=method, bundle_order=0, integration_order_mode=mode)))
            r = R.Reference(name, csv, order_mode=rname)
            for k in range(2):
                f = synth.render_frame(sc, synth.trajectory_pose(4 * k), 128, 96, seed=70 + k)
                o.integrate(f.T_G_C, f.xyz, f.rgba if method == 0 else None, f.labels)
                r.integrate(f.T_G_C, f.xyz, f.rgba)
            assert _same(o, r) > 1000
            maps.append(o.download()[2]["priors"].copy())
        assert maps[0].shape != maps[1].shape or not np.array_equal(maps[0], maps[1]), "the two forms are different orders: the maps must differ somewhere"


from tests.variants import VARIANTS  # noqa: E402


@pytest.mark.parametrize("name", sorted(VARIANTS))
@pytest.mark.parametrize("method", ["fast", "merged"])
def test_config_variants_match_reference(tmp_path, name, method):
    """Every configuration knob the hot path reads, oracle restatement vs the real reference."""
    csv = _csv(tmp_path)
    v = dict(VARIANTS[name])
    okw = dict(COMMON, method=0 if method == "fast" else 1, bundle_order=0)
    okw.update(v)
    rkw = dict(v)
    ref_args = {}
    if "voxel_size" in rkw:
        ref_args["voxel_size"] = rkw.pop("voxel_size")
    if "truncation_distance" in rkw:
        ref_args["truncation"] = rkw.pop("truncation_distance")
    if "max_ray_length_m" in rkw:
        ref_args["max_ray"] = rkw.pop("max_ray_length_m")
    if "semantic_measurement_probability" in rkw:
        ref_args["p_match"] = rkw.pop("semantic_measurement_probability")
    if "dynamic_labels" in rkw:
        ref_args["dynamic_labels"] = tuple(rkw.pop("dynamic_labels"))
    o = O.Oracle(O.default_config(**okw))
    r = R.Reference(method, csv, **ref_args, **rkw)
    sc = synth.make_scene("room")
    for k in range(4):
        f = synth.render_frame(sc, synth.trajectory_pose(7 * k), 80, 60, seed=60 + k)
        fs = (k == 2)  # one frame flagged as freespace points
        o.integrate(f.T_G_C, f.xyz, f.rgba if method == "fast" else None, f.labels, freespace=fs)
        r.integrate(f.T_G_C, f.xyz, f.rgba, freespace=fs)
    assert _same(o, r) > 300


from tests.variants import ORDER_MODE_NAMES, random_combo  # noqa: E402


@pytest.mark.parametrize("seed", range(10))
@pytest.mark.parametrize("method", ["fast", "merged"])
def test_random_knob_combinations_match_reference(tmp_path, seed, method):
    """Seeded random combinations of the configuration knobs (their interactions), several frames
    incl. a free-space cloud: oracle restatement == real reference sources, bit for bit."""
    csv = _csv(tmp_path)
    v = random_combo(seed)
    okw = dict(COMMON, method=0 if method == "fast" else 1, bundle_order=0)
    okw.update(v)
    rkw = dict(v)
    ref_args = {"color_mode": rkw.pop("color_mode"),
                "order_mode": ORDER_MODE_NAMES[rkw.pop("integration_order_mode")]}
    for src, dst in (("voxel_size", "voxel_size"), ("truncation_distance", "truncation"), ("max_ray_length_m", "max_ray"),
                     ("semantic_measurement_probability", "p_match")):
        if src in rkw:
            ref_args[dst] = rkw.pop(src)
    if "dynamic_labels" in rkw:
        ref_args["dynamic_labels"] = tuple(rkw.pop("dynamic_labels"))
    o = O.Oracle(O.default_config(**okw))
    r = R.Reference(method, csv, **ref_args, **rkw)
    sc = synth.make_scene("room")
    for k in range(3):
        f = synth.render_frame(sc, synth.trajectory_pose(9 * k + seed), 80, 60, seed=900 + 10 * seed + k)
        fs = (k == 1)
        o.integrate(f.T_G_C, f.xyz, f.rgba if method == "fast" else None, f.labels, freespace=fs)
        r.integrate(f.T_G_C, f.xyz, f.rgba, freespace=fs)
    assert _same(o, r) > 100, v


@pytest.mark.parametrize("left_over", [1, 2])
def test_reference_contexts_do_not_inherit_the_static_reset_counter(tmp_path, left_over):
    """[K:src/semantic_tsdf_integrator_fast.cpp:165] counts frames in a function-static: a context with
    clear_checks_every_n_frames = 3 that integrated 1 or 2 frames leaves them on the counter for every later integrator of the
    process.  The checker (oracle/ref_shim/ref_driver.cpp: kr_create) realigns the counter, so what a Reference computes
    does not depend on which tests ran before it in the same (xdist worker) process."""
    csv = _csv(tmp_path)
    sc = synth.make_scene("room")
    frames = [synth.render_frame(sc, synth.trajectory_pose(7 * k), 80, 60, seed=60 + k) for k in range(4)]
    polluter = R.Reference("fast", csv, clear_checks_every_n_frames=3)
    for f in frames[:left_over]:
        polluter.integrate(f.T_G_C, f.xyz, f.rgba)
    polluter.close()
    o = O.Oracle(O.default_config(**dict(COMMON, method=0, clear_checks_every_n_frames=3)))
    r = R.Reference("fast", csv, clear_checks_every_n_frames=3)
    for f in frames:
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        r.integrate(f.T_G_C, f.xyz, f.rgba)
    assert _same(o, r) > 300
