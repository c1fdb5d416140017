"""-m gpu: HIP path vs CPU oracle through the C ABI (bit-exact labels / indices; TSDF
distance & weight bit-exact too because the HIP path replays the oracle's per-voxel order)."""
import numpy as np
import pytest

from kimera_semantics_amd import binding as B
from kimera_semantics_amd import synth
from oracle import oracle_py as O
from tests.util import COMMON, NO_EARLY_OUT, compare_maps, small_frame

pytestmark = pytest.mark.gpu


def _pair(method, **kw):
    okw = dict(COMMON, method=method, **kw)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 18, **okw))
    return o, h


@pytest.mark.parametrize("vps", [8, 16, 32])
def test_merged_single_frame_exact(vps):
    f = small_frame(seed=1)
    o, h = _pair(1, voxels_per_side=vps)
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert (so.n_valid_points, so.n_rays_cast, so.n_voxel_updates) == (sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates)
    rep = compare_maps(o, h, exact=True)
    assert rep["oracle_touched"] > 1000


def test_fast_no_early_out_exact():
    f = small_frame(seed=2)
    o, h = _pair(0, max_consecutive_ray_collisions=NO_EARLY_OUT)
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert (so.n_valid_points, so.n_rays_cast, so.n_voxel_updates) == (sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates)
    compare_maps(o, h, exact=True)


@pytest.mark.parametrize("method", [0, 1])
def test_multi_frame_exact(method):
    o, h = _pair(method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    sc = synth.make_scene("room")
    for k in range(4):
        f = synth.render_frame(sc, synth.trajectory_pose(4 * k), 128, 96, seed=10 + k)
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        assert so.n_voxel_updates == sh.n_voxel_updates and so.n_rays_cast == sh.n_rays_cast
    compare_maps(o, h, exact=True)


@pytest.mark.parametrize("growth,size", [(32, (320, 240)), (16, (160, 120)), (24, (160, 120)), (32, (640, 480))])
def test_fast_early_out_ordered_phases_exact(growth, size):
    """The DEFAULT fast configuration (max_consecutive_ray_collisions = 2).  The reference's loop is serial
    (ray k stops on marks of rays 1..k-1); the GPU runs the ordered-phase schedule, which the oracle
    restates (early_out_phase_growth): bit-exact against that restatement, deterministic, over several
    frames (the approximate sets carry over)."""
    o, h = _pair(0, early_out_phase_growth=growth)
    sc = synth.make_scene("room")
    for k in range(3):
        f = synth.render_frame(sc, synth.trajectory_pose(5 * k), size[0], size[1], seed=30 + k)
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        assert (so.n_valid_points, so.n_rays_cast, so.n_voxel_updates) == (sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates), k
    compare_maps(o, h, exact=True)


def test_fast_early_out_rounds_one_after_the_other_exact(monkeypatch):
    """KS_TEST_OVERLAP=0: k_test casts a long ray's next 64 voxels only after the current 64 have been decided (the code
    measured until round 3; the default overlaps the two).  Same schedule, same result: both against the oracle, 2 cm voxels
    and 9 m rays so that rays take many rounds."""
    geom = dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=9.0)
    sc = synth.make_scene("hall")
    frames = [synth.render_frame(sc, synth.trajectory_pose(3 * k), 320, 180, hfov_deg=75.0, seed=70 + k) for k in range(2)]
    for overlap in ("0", "1"):
        monkeypatch.setenv("KS_DEBUG", "1")
        monkeypatch.setenv("KS_TEST_OVERLAP", overlap)
        okw = dict(COMMON, method=0, early_out_phase_growth=32, **geom)
        o = O.Oracle(O.default_config(**okw))
        h = B.HipIntegrator(B.default_config(max_tiles=1 << 16, max_points=1 << 16, **okw))
        for f in frames:
            so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
            sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
            assert (so.n_rays_cast, so.n_voxel_updates) == (sh.n_rays_cast, sh.n_voxel_updates), overlap
        compare_maps(o, h, exact=True)
        h.close()
        o.close()


@pytest.mark.parametrize("variant", ["clear_every_3", "sorted_order", "pipelined", "subsample_1", "limit_0", "limit_5"])
def test_fast_early_out_ordered_phases_variants_exact(variant):
    kw = dict(early_out_phase_growth=32)
    pipe = 0
    if variant == "clear_every_3":
        kw["clear_checks_every_n_frames"] = 3     # marks of earlier frames stay valid (same offset)
    elif variant == "sorted_order":
        kw["integration_order_mode"] = 1
    elif variant == "pipelined":
        pipe = 2
    elif variant == "subsample_1":
        kw["start_voxel_subsampling_factor"] = 1.0
    elif variant == "limit_0":
        kw["max_consecutive_ray_collisions"] = 0
    elif variant == "limit_5":
        kw["max_consecutive_ray_collisions"] = 5
    okw = dict(COMMON, method=0, **kw)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, pipeline_frames=pipe, **okw))
    sc = synth.make_scene("room")
    for k in range(5):
        f = synth.render_frame(sc, synth.trajectory_pose(2 * k), 160, 120, seed=60 + k)
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=(k == 3))
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=(k == 3))
    h.flush()
    compare_maps(o, h, exact=True)


def test_fast_early_out_fidelity_vs_serial_reference():
    """Distance of the ordered-phase schedule from the SERIAL reference order (oracle, one thread).
    Measured on the CPU restatement (tests/early_out_fidelity.py): touched-set Jaccard 0.976 with doubling
    phases (default), 0.987 with growth 1.5, 0.995 with one generation per phase; the reference's own
    1-thread vs 8-thread spread is 0.998."""
    sc = synth.make_scene("room")
    f = synth.render_frame(sc, synth.trajectory_pose(7), 640, 480, seed=7)
    o, h = _pair(0)                      # oracle: serial reference order; GPU: default schedule
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert so.n_rays_cast == sh.n_rays_cast  # start-voxel dedup is exact
    rep = compare_maps(o, h, exact=False)
    assert rep["block_jaccard"] > 0.99
    assert rep["touched_jaccard"] >= 0.95, rep
    assert 1.0 <= sh.n_voxel_updates / so.n_voxel_updates < 1.2
    o1, h1 = _pair(0, early_out_phase_growth=16)
    o1.close()
    o1 = O.Oracle(O.default_config(**dict(COMMON, method=0)))
    o1.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh1 = h1.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    rep1 = compare_maps(o1, h1, exact=False)
    assert rep1["touched_jaccard"] >= 0.99, rep1
    assert 1.0 <= sh1.n_voxel_updates / so.n_voxel_updates < 1.1


def test_fast_early_out_c4_geometry():
    """C4 geometry (2 cm voxels, 10 m rays, trunc 8 cm; 320x180 here): a whole wavefront per ray (rays of
    ~600 steps).  One frame makes far more voxel visits than the approximate set has slots, so the serial
    reference's set thrashes; with one generation per phase the update count stays within 10 % of the
    serial oracle (measured 0.98), and the GPU is bit-exact against the restated schedule."""
    geom = dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=10.0)
    sc = synth.make_scene("hall")
    f = synth.render_frame(sc, synth.trajectory_pose(3), 320, 180, hfov_deg=75.0, seed=3)
    okw = dict(COMMON, method=0, early_out_phase_growth=16, **geom)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 15, max_points=1 << 16, **okw))
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert (so.n_rays_cast, so.n_voxel_updates) == (sh.n_rays_cast, sh.n_voxel_updates)
    compare_maps(o, h, exact=True)
    serial = O.Oracle(O.default_config(**dict(COMMON, method=0, **geom)))
    ss = serial.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert abs(sh.n_voxel_updates / ss.n_voxel_updates - 1.0) < 0.10, (sh.n_voxel_updates, ss.n_voxel_updates)
    # default (doubling) schedule at this geometry: exact against its restatement as well
    okw["early_out_phase_growth"] = 32
    o2 = O.Oracle(O.default_config(**okw))
    h2 = B.HipIntegrator(B.default_config(max_tiles=1 << 15, max_points=1 << 16, **okw))
    s2o = o2.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    s2h = h2.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert s2o.n_voxel_updates == s2h.n_voxel_updates
    compare_maps(o2, h2, exact=True)


def test_full_size_c4_default_early_out_exact():
    """One FULL-SIZE C4 frame (1280x720, 2 cm voxels, 10 m rays) through the DEFAULT fast configuration: 900 generations, the last
    phase 388 generations long — k_test ranks a chain's live rays over more than one window of 256 generations, up to 25 sub-runs
    per chain are launched.  Update and ray counts, the allocated block set, and every voxel of a sample of 600 blocks against the
    restated schedule (the whole map is 3 GB of host layout)."""
    geom = dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=10.0)
    f = synth.render_frame(synth.make_scene("hall"), synth.trajectory_pose(3, radius=3.0), 1280, 720, hfov_deg=75.0, seed=3)
    okw = dict(COMMON, method=0, early_out_phase_growth=32, **geom)
    o = O.Oracle(O.default_config(**okw))
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 16, max_points=1280 * 720, **okw))
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert (so.n_valid_points, so.n_rays_cast, so.n_voxel_updates) == (sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates)
    oi, hi = o.block_indices(), h.block_indices()
    assert np.array_equal(oi, hi)
    sample = np.ascontiguousarray(oi[:: max(1, len(oi) // 600)])
    _, ot, osem = o.download(sample)
    _, ht, hsem = h.download(sample)
    assert (ot["weight"] > 0).sum() > 1e5
    for a, b in ((ot, ht), (osem, hsem)):
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("pipe,growth", [(1, 32), (8, 32), (8, 0), (16, 0)])
def test_observed_set_tag_wrap(pipe, growth):
    """The early-out set's entries carry a 10-bit frame tag; every ~1000 frames the stale entries are retired
    and the tags restart.  1100 small frames stay bit-exact against the oracle — also with batches of four frames per
    launch (pipeline_frames = 8) whose alignment a query in mid-stream has shifted, so that frames are waiting for their
    batch to fill when the retag comes: they go out with their OLD tag before the tables are rewritten (reset_set);
    and in the default mode (growth 0), whose seed runs on the same tables."""
    okw = dict(COMMON, method=0, early_out_phase_growth=growth)
    o = O.Oracle(O.default_config(integrator_threads=1, **okw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 12, pipeline_frames=pipe, **okw))
    sc = synth.make_scene("room")
    frames = [synth.render_frame(sc, synth.trajectory_pose(k), 48, 36, seed=k) for k in range(8)]
    for k in range(1100):
        f = frames[k % 8]
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        if pipe > 1 and k in (5, 1001, 1015):
            h.block_indices()   # (completes the frames in flight: the next batch starts off the multiple of four / eight)
    h.flush()
    compare_maps(o, h, exact=True)


@pytest.mark.parametrize("method,color_mode", [(1, 1), (1, 0), (0, 0), (1, 2)])
def test_close_up_long_runs_exact(method, color_mode):
    """Camera 0.45 m from a wall: thousands of pixels fall into one voxel (wave-per-bundle
    path) and voxels near the sensor collect thousands of updates (wave-per-voxel path)."""
    sc = synth.make_scene("room")
    T = synth.pose_to_T((3.5, 0.3, 1.2), 0.1)
    f = synth.render_frame(sc, T, 320, 240, seed=7)
    o, h = _pair(method, max_consecutive_ray_collisions=NO_EARLY_OUT, color_mode=color_mode)
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert (so.n_rays_cast, so.n_voxel_updates) == (sh.n_rays_cast, sh.n_voxel_updates)
    if color_mode == 2:
        # kSemanticProbability colours go through exp(): allow 1 LSB on the TSDF colour only
        rep = compare_maps(o, h, exact=False)
        assert rep["label_mismatches"] == 0 and rep["max_abs_priors_err"] == 0.0 and rep["max_abs_distance_err"] == 0.0
        assert rep["tsdf_color_mismatches"] <= rep["voxels_compared"] * 1e-3
    else:
        compare_maps(o, h, exact=True)


@pytest.mark.parametrize("xlong", ["1", "0"])
@pytest.mark.parametrize("method,color_mode,pipe", [(1, 1, 0), (1, 0, 4), (1, 2, 0), (0, 1, 8)])
def test_runs_next_to_the_sensor_on_their_own_list_exact(monkeypatch, method, color_mode, pipe, xlong):
    """The runs of more than 1024 updates (the voxels next to the sensor: every ray of `merged` starts there) are listed apart
    and walked by k_apply_xlong — four waves per run: two prepare the batches alternately, one walks the TSDF recurrences, one
    the class sums — on a stream of its own (KS_XLONG=0: one list, k_apply_long).  Same arithmetic in the same order: the map
    is the oracle's bit for bit, over frames that share voxels."""
    monkeypatch.setenv("KS_DEBUG", "1")
    monkeypatch.setenv("KS_XLONG", xlong)
    sc = synth.make_scene("room")
    okw = dict(COMMON, method=method, max_consecutive_ray_collisions=NO_EARLY_OUT, color_mode=color_mode)
    o = O.Oracle(O.default_config(integrator_threads=1, **okw))
    h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=320 * 240, pipeline_frames=pipe, **okw))
    for k, T in enumerate((synth.pose_to_T((3.5, 0.3, 1.2), 0.1), synth.trajectory_pose(1), synth.trajectory_pose(1), synth.trajectory_pose(2))):
        f = synth.render_frame(sc, T, 320, 240, seed=k)
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    h.flush()
    if color_mode == 2:
        rep = compare_maps(o, h, exact=False)    # (kSemanticProbability colours go through exp(): 1 LSB on the TSDF colour only)
        assert rep["label_mismatches"] == 0 and rep["max_abs_priors_err"] == 0.0 and rep["max_abs_distance_err"] == 0.0
    else:
        compare_maps(o, h, exact=True)


@pytest.mark.parametrize("method", [0, 1])
def test_sorted_integration_order_exact(method):
    f = small_frame(seed=5, w=96, h=72)
    o, h = _pair(method, max_consecutive_ray_collisions=NO_EARLY_OUT, integration_order_mode=1)
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert (so.n_rays_cast, so.n_voxel_updates) == (sh.n_rays_cast, sh.n_voxel_updates)
    compare_maps(o, h, exact=True)


@pytest.mark.parametrize("dtype,n,end_bit", [(np.uint32, 1, 21), (np.uint32, 2047, 21), (np.uint32, 305619, 21),
                                             (np.uint64, 2049, 64), (np.uint64, 700001, 41), (np.uint64, 3000000, 64),
                                             (np.uint32, 100000, 32), (np.uint64, 1 << 20, 40), (np.uint64, (1 << 20) + 1, 40),
                                             (np.uint64, 17000001, 33)])
def test_device_radix_sort_is_correct_and_stable(dtype, n, end_bit):
    rng = np.random.default_rng(n)
    h = B.HipIntegrator(B.default_config(max_tiles=64, max_points=1024, **dict(COMMON, method=0)))
    hi = (1 << end_bit) - 1
    if n > 4:
        keys = rng.integers(0, min(hi, 5000), size=n, dtype=np.uint64)      # many duplicates: stability matters
        keys[: n // 2] = rng.integers(0, hi, size=n // 2, dtype=np.uint64, endpoint=True)
    else:
        keys = rng.integers(0, hi, size=n, dtype=np.uint64, endpoint=True)
    keys = keys.astype(dtype)
    vals = np.arange(n, dtype=np.uint32)
    k2, v2 = h.debug_radix_sort(keys, vals, end_bit)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(k2, keys[order])
    assert np.array_equal(v2, vals[order])
    k3, _ = h.debug_radix_sort(keys, None, end_bit)
    assert np.array_equal(k3, keys[order])


@pytest.mark.parametrize("method", [0, 1])
def test_depth_image_entry_equals_cloud_entry(method):
    """f-1: ks_integrate_depth(depth, labels) == ks_integrate_points(back-projected finite cloud)."""
    f = small_frame(seed=11, w=160, h=120)
    kw = dict(COMMON, method=method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    a = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    b = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    sa = a.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sb = b.integrate_depth(f.T_G_C, f.depth, f.K, label_img=f.label_img)
    assert sb.n_points == len(f.xyz)  # NaN pixels dropped
    assert (sa.n_valid_points, sa.n_rays_cast, sa.n_voxel_updates) == (sb.n_valid_points, sb.n_rays_cast, sb.n_voxel_updates)
    ia, ta, sa_ = a.download()
    ib, tb, sb_ = b.download()
    assert np.array_equal(ia, ib) and ta.tobytes() == tb.tobytes() and sa_.tobytes() == sb_.tobytes()
    # and against the oracle fed with the reference-style cloud
    o = O.Oracle(O.default_config(**kw))
    o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    compare_maps(o, b, exact=True)


def test_depth_image_u16_and_colour_decoding():
    f = small_frame(seed=12, w=96, h=72)
    mm = np.where(np.isfinite(f.depth), np.round(f.depth * 1000.0), 0).astype(np.uint16)
    kw = dict(COMMON, method=1)
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    lut = synth.default_label_colors()
    h.set_color_to_label(lut[:21], np.arange(21, dtype=np.uint8))
    rgba_img = lut[f.label_img]
    st = h.integrate_depth(f.T_G_C, mm, f.K, rgba_img=rgba_img)
    # reference-style cloud from the same u16 image (DepthTraits<uint16_t>)
    fx, fy, cx, cy = [np.float32(v) for v in f.K]
    u = np.arange(mm.shape[1], dtype=np.float32)[None, :]
    v = np.arange(mm.shape[0], dtype=np.float32)[:, None]
    d = mm.astype(np.float32)
    cxs, cys = np.float32(np.float64(0.001) / np.float64(fx)), np.float32(np.float64(0.001) / np.float64(fy))
    pts = np.stack([((u - cx) * d) * cxs, ((v - cy) * d) * cys, d * np.float32(0.001)], axis=-1).reshape(-1, 3).astype(np.float32)
    ok = (mm.reshape(-1) != 0)
    o = O.Oracle(O.default_config(**kw))
    so = o.integrate(f.T_G_C, pts[ok], None, f.label_img.reshape(-1)[ok])
    assert (so.n_rays_cast, so.n_voxel_updates) == (st.n_rays_cast, st.n_voxel_updates)
    compare_maps(o, h, exact=True)


def test_merged_anti_grazing_exact():
    """vxb Config::enable_anti_grazing: rays skip voxels that are another bundle's end voxel."""
    f = small_frame(seed=13, w=128, h=96)
    o, h = _pair(1, enable_anti_grazing=1)
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    o2 = O.Oracle(O.default_config(**dict(COMMON, method=1)))
    s2 = o2.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert so.n_voxel_updates < s2.n_voxel_updates  # the flag does something
    assert (so.n_rays_cast, so.n_voxel_updates) == (sh.n_rays_cast, sh.n_voxel_updates)
    compare_maps(o, h, exact=True)


@pytest.mark.parametrize("method", [0, 1])
def test_saturated_weights_exact(method):
    """Small max_weight: voxel weights clamp within a frame or two, which exercises the
    saturated-weight shortcut of the wave-per-voxel kernel (and the clamp itself)."""
    o, h = _pair(method, max_consecutive_ray_collisions=NO_EARLY_OUT, max_weight=3.0)
    sc = synth.make_scene("room")
    T = synth.pose_to_T((3.5, 0.3, 1.2), 0.1)
    for k in range(3):
        f = synth.render_frame(sc, T, 192, 144, seed=90 + k)
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        assert so.n_voxel_updates == sh.n_voxel_updates
    rep = compare_maps(o, h, exact=True)
    _, t, _ = h.download()
    assert (t["weight"] == np.float32(3.0)).sum() > 1000  # the clamp was hit


@pytest.mark.parametrize("method", [0, 1])
def test_full_size_640x480_frame_exact(method):
    """BASELINE.json's frame size (640x480 @ 5 cm, 5 m rays), one frame, bit-exact vs the oracle."""
    sc = synth.make_scene("room")
    f = synth.render_frame(sc, synth.trajectory_pose(5), 640, 480, seed=5)  # includes a 4.6k-point bundle
    o, h = _pair(method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert (so.n_valid_points, so.n_rays_cast, so.n_voxel_updates) == (sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates)
    rep = compare_maps(o, h, exact=True)
    assert rep["oracle_touched"] > 150000


def test_full_size_properties():
    """Size-independent properties at full frame size: determinism, unknown-label no-op,
    per-frame update counts add up, untouched voxels keep the initial state."""
    sc = synth.make_scene("room")
    f = synth.render_frame(sc, synth.trajectory_pose(9), 640, 480, seed=9)
    kw = dict(COMMON, method=1)
    a = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 19, **kw))
    b = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 19, **kw))
    sa = a.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    sb = b.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    ia, ta, sea = a.download()
    ib, tb, seb = b.download()
    assert sa.n_voxel_updates == sb.n_voxel_updates
    assert np.array_equal(ia, ib) and ta.tobytes() == tb.tobytes() and sea.tobytes() == seb.tobytes()  # deterministic
    # a second pass with every label forced to 0 (unknown) must not move any prior or label
    s2 = a.integrate(f.T_G_C, f.xyz, f.rgba, np.zeros_like(f.labels))
    assert s2.n_voxel_updates == sa.n_voxel_updates  # same rays, same voxels
    _, t2, se2 = a.download(ia)
    assert np.array_equal(se2["priors"].view(np.uint32), sea["priors"].view(np.uint32))
    assert np.array_equal(se2["label"], sea["label"])
    touched = ta["weight"] > 0
    assert np.all(t2["weight"][touched] >= ta["weight"][touched])  # weights only grow
    untouched = (t2["weight"] == 0) & (se2["label"] == 0)
    assert np.all(se2["priors"][untouched] == np.float32(-0.60205999132))


def test_error_codes_mirror_reference_checks():
    f = small_frame(seed=1, w=64, h=48)
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 14, **dict(COMMON, method=0)))
    bad = f.labels.copy()
    bad[7] = 21  # CHECK_LT(label, 21) in the reference
    with pytest.raises(B.KsError) as e:
        h.integrate(f.T_G_C, f.xyz, f.rgba, bad)
    assert e.value.code == B.KS_ERR_LABEL_RANGE
    assert len(h.block_indices()) == 0  # nothing was integrated
    st = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)  # context stays usable
    assert st.n_voxel_updates > 0
    # empty cloud is a no-op
    st = h.integrate(f.T_G_C, np.zeros((0, 3), np.float32), np.zeros((0, 4), np.uint8), np.zeros(0, np.uint8))
    assert st.n_voxel_updates == 0
    # invalid probabilities are rejected at construction (CHECKs of setSemanticProbabilities)
    for p in (0.0, 1.0, 0.4):
        with pytest.raises(B.KsError) as e:
            B.HipIntegrator(B.default_config(**dict(COMMON, method=0, semantic_measurement_probability=p)))
        assert e.value.code == B.KS_ERR_PROBABILITY
    # tile pool exhaustion is reported, not silently dropped
    small = B.HipIntegrator(B.default_config(max_tiles=16, max_points=1 << 14, **dict(COMMON, method=1)))
    with pytest.raises(B.KsError) as e:
        small.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert e.value.code == B.KS_ERR_POOL_FULL


from tests.variants import VARIANTS  # noqa: E402


@pytest.mark.parametrize("name", sorted(VARIANTS))
@pytest.mark.parametrize("method", [0, 1])
def test_config_variants_exact(name, method):
    """Every configuration knob the hot path reads (incl. a freespace frame), HIP vs oracle, bit-exact."""
    o, h = _pair(method, **dict(dict(max_consecutive_ray_collisions=NO_EARLY_OUT), **VARIANTS[name]))
    sc = synth.make_scene("room")
    for k in range(4):
        f = synth.render_frame(sc, synth.trajectory_pose(7 * k), 80, 60, seed=60 + k)
        fs = (k == 2)
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=fs)
        sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=fs)
        assert (so.n_valid_points, so.n_rays_cast, so.n_voxel_updates) == (sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates), (name, k)
    compare_maps(o, h, exact=True)


def test_device_pointer_entries_equal_host_entries():
    """ks_integrate_points_device / ks_integrate_depth_device (inputs already in HBM, as bench.py
    uses them) give the same map as the host-pointer entries."""
    import torch
    f = small_frame(seed=21, w=128, h=96)
    kw = dict(COMMON, method=0, max_consecutive_ray_collisions=NO_EARLY_OUT)
    a = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    b = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    c = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    a.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    x, col, lab = torch.from_numpy(f.xyz).cuda(), torch.from_numpy(f.rgba).cuda(), torch.from_numpy(f.labels).cuda()
    torch.cuda.synchronize()
    b.integrate_device(f.T_G_C, x.data_ptr(), col.data_ptr(), lab.data_ptr(), x.shape[0])
    d, li = torch.from_numpy(f.depth).cuda(), torch.from_numpy(f.label_img).cuda()
    torch.cuda.synchronize()
    c.integrate_depth_device(f.T_G_C, d.data_ptr(), 0, li.data_ptr(), 0, f.depth.shape[1], f.depth.shape[0], f.K)
    ia, ta, sa = a.download()
    for other in (b, c):
        io, to, so = other.download()
        assert np.array_equal(ia, io) and ta.tobytes() == to.tobytes() and sa.tobytes() == so.tobytes()


@pytest.mark.parametrize("method", [0, 1])
def test_long_sequence_exact(method):
    """12 consecutive trajectory frames at 320x240: tile growth, approximate-set resets across
    frames, weights accumulating over many frames — still bit-exact at the end."""
    o, h = _pair(method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    sc = synth.make_scene("room")
    for k in range(12):
        f = synth.render_frame(sc, synth.trajectory_pose(3 * k), 320, 240, seed=500 + k)
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        assert (so.n_rays_cast, so.n_voxel_updates) == (sh.n_rays_cast, sh.n_voxel_updates), k
    compare_maps(o, h, exact=True)


@pytest.mark.parametrize("method", [0, 1])
def test_degenerate_inputs_exact(method):
    """Edge cases the reference code paths contain: zero-weight points (|z| <= 1e-6), duplicate
    points, points at the min/max ray-length limits, an all-invalid cloud, a one-point cloud, an
    all-dynamic cloud."""
    o, h = _pair(method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    T = synth.pose_to_T((0.3, -0.2, 1.1), 0.4)
    rng = np.random.default_rng(5)
    base = rng.uniform(-1.5, 1.5, size=(400, 3)).astype(np.float32)
    base[:, 2] = np.abs(base[:, 2]) + 0.3
    pts = np.concatenate([
        base,
        base[:50],                                                   # exact duplicates
        np.array([[0.7, 0.2, 0.0], [0.7, 0.2, 1e-7], [1.2, -0.4, -1e-7]], np.float32),  # zero weight
        np.array([[0.0, 0.0, 0.1], [0.0, 0.06, 0.08], [0.0, 0.0, 0.0999]], np.float32),  # around min_ray_length
        np.array([[3.0, 4.0, 0.0], [0.0, 3.0, 4.0], [0.0, 3.0, 4.0001], [30.0, 40.0, 5.0]], np.float32),  # around/beyond max
    ]).astype(np.float32)
    labels = rng.integers(0, 21, size=len(pts), dtype=np.uint8)
    rgba = synth.default_label_colors()[labels]
    for cloud, lab, col in ((pts, labels, rgba),
                            (np.full((5, 3), 0.01, np.float32), labels[:5], rgba[:5]),   # all too close: invalid
                            (pts[:1], labels[:1], rgba[:1]),                               # single point
                            (pts[:64], np.full(64, 20, np.uint8), synth.default_label_colors()[np.full(64, 20)])):
        so = o.integrate(T, cloud, col, lab)
        sh = h.integrate(T, cloud, col, lab)
        assert (so.n_valid_points, so.n_rays_cast, so.n_voxel_updates) == (sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates)
    rep = compare_maps(o, h, exact=False)
    # NaN distances (0/0 in computeDistance for a zero-length ray) must be NaN on both sides
    assert rep["label_mismatches"] == 0 and rep["max_abs_priors_err"] == 0.0
    idx = o.block_indices()
    _, ot, _ = o.download(idx)
    _, ht, _ = h.download(idx)
    assert np.array_equal(np.isnan(ot["distance"]), np.isnan(ht["distance"]))
    ok = ~np.isnan(ot["distance"])
    assert np.array_equal(ot["distance"][ok].view(np.uint32), ht["distance"][ok].view(np.uint32))
    assert np.array_equal(ot["weight"].view(np.uint32), ht["weight"].view(np.uint32))


@pytest.mark.parametrize("color_mode", [1, 0])
def test_long_bundles_edge_cases_exact(color_mode):
    """`merged`, bundles of >= 32 points (one workgroup each, 64 points per step): bundles of exactly 64 / 128 / 192 points (the
    bundle ends on a step boundary), of 33 and 65, zero-weight points (|z| <= 1e-6 after the pose) inside a long bundle, points
    whose coordinate is exactly 0 (the exponent window of the division), and long bundles of points beyond the ray-length limit
    (clearing rays: the first usable point only)."""
    o, h = _pair(1, max_consecutive_ray_collisions=NO_EARLY_OUT, color_mode=color_mode)
    T = np.array([1, 0, 0, 0, 0, 0, 0], np.float32)    # identity: the cloud's z is the weight's z
    rng = np.random.default_rng(11)
    vs = 0.05   # (the default voxel size; max_ray_length_m = 5)

    def cluster(center, n, spread=0.4):
        c = (np.floor(np.asarray(center) / vs) + 0.5) * vs
        return (c + rng.uniform(-spread, spread, size=(n, 3)) * vs).astype(np.float32)

    parts = [cluster((0.8, 0.3, 1.5), 64), cluster((-0.6, 0.2, 1.2), 128), cluster((0.1, -0.7, 2.0), 192),
             cluster((1.1, 1.0, 0.9), 33), cluster((-1.0, -0.5, 1.7), 65), cluster((0.4, 0.9, 2.6), 1000)]
    zero_w = cluster((0.52, 0.52, 0.0), 80, spread=0.45)          # voxel around z = 0: some weights 1/z^2 are huge, some points
    zero_w[::3, 2] = 0.0                                          # have z = 0 exactly: zero weight, skipped inside the bundle
    on_axis = cluster((0.0, 0.0, 1.0), 100)
    on_axis[::2, 0] = 0.0                                         # x exactly 0: the mean's numerator is 0 in the first steps
    on_axis[:10, 1] = 0.0
    far = cluster((3.0, 4.0, 12.0), 150)                          # beyond max_ray_length: clearing bundle, long
    pts = np.concatenate(parts + [zero_w, on_axis, far]).astype(np.float32)
    pts = pts[rng.permutation(len(pts))]
    labels = rng.integers(1, 21, size=len(pts), dtype=np.uint8)
    rgba = synth.default_label_colors()[labels].copy()
    rgba[:, :3] = rng.integers(0, 255, size=(len(pts), 3), dtype=np.uint8)
    for rep in range(2):
        so = o.integrate(T, pts, rgba, labels)
        sh = h.integrate(T, pts, rgba, labels)
        assert (so.n_valid_points, so.n_rays_cast, so.n_voxel_updates) == (sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates)
    compare_maps(o, h, exact=True)


def test_out_of_range_coordinates_are_reported():
    h = B.HipIntegrator(B.default_config(max_tiles=1024, max_points=1 << 12, **dict(COMMON, method=1, max_ray_length_m=1e9)))
    T = np.array([1, 0, 0, 0, 0, 0, 0], np.float32)
    far = np.array([[1e6, 2e6, 3e6]], np.float32)   # voxel index beyond the packed +-2^20 range
    with pytest.raises(B.KsError) as e:
        h.integrate(T, far, None, np.array([1], np.uint8))
    assert e.value.code == -6  # KS_ERR_INDEX_RANGE


@pytest.mark.parametrize("method", [0, 1])
def test_upload_roundtrip_and_resume(method):
    """ks_upload_blocks is the inverse of ks_download_blocks: a map moved host-side into a fresh
    context downloads identically, and both contexts then integrate further frames identically."""
    kw = dict(COMMON, method=method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    a = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    b = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    sc = synth.make_scene("room")
    fr = [synth.render_frame(sc, synth.trajectory_pose(4 * k), 160, 120, seed=40 + k) for k in range(4)]
    for f in fr[:2]:
        a.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    idx, t, s = a.download()
    b.upload(idx, t, s)
    i2, t2, s2 = b.download()
    assert np.array_equal(idx, i2) and t.tobytes() == t2.tobytes() and s.tobytes() == s2.tobytes()
    for f in fr[2:]:
        a.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        b.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    compare_maps(a, b, exact=True)
    # half uploads leave the other half alone
    c = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
    c.upload(idx, tsdf=t)
    _, tc, sc_ = c.download(idx)
    assert tc.tobytes() == t.tobytes()
    assert np.all(sc_["label"] == 0) and np.all(sc_["color"] == np.array([127, 127, 127, 255], np.uint8))
    c.upload(idx, sem=s)
    _, tc, sc_ = c.download(idx)
    assert tc.tobytes() == t.tobytes() and sc_.tobytes() == s.tobytes()
    with pytest.raises(B.KsError):
        c.upload(np.array([[1 << 20, 0, 0]], np.int32), tsdf=t[:1])


@pytest.mark.parametrize("method", [0, 1])
def test_pipelined_frames_equal_unpipelined(method):
    """ks_config.pipeline_frames only moves WHEN a frame's second half is enqueued: the map is
    bit-identical, the statistics arrive one call later, queries complete the outstanding frame."""
    kw = dict(COMMON, method=method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    o = O.Oracle(O.default_config(**kw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, pipeline_frames=1, **kw))
    sc = synth.make_scene("room")
    # growing clouds force a buffer re-allocation while a frame is outstanding
    sizes = [(96, 72), (160, 120), (160, 120), (240, 180), (160, 120), (160, 120)]
    ref_stats, got = [], []
    for k, (w, hh) in enumerate(sizes):
        f = synth.render_frame(sc, synth.trajectory_pose(4 * k), w, hh, seed=700 + k)
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        ref_stats.append((so.n_points, so.n_valid_points, so.n_rays_cast, so.n_voxel_updates))
        sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        got.append((sh.n_points, sh.n_valid_points, sh.n_rays_cast, sh.n_voxel_updates))
        if k == 2:  # a query in the middle of the stream sees frames 0..2 complete
            assert len(h.block_indices()) == len(o.block_indices())
    last = h.flush()
    got.append((last.n_points, last.n_valid_points, last.n_rays_cast, last.n_voxel_updates))
    assert h.flush().n_points == 0  # nothing left
    # call k returns frame k-1 (also when a query or a re-allocation completed that frame early)
    assert got[0] == (0, 0, 0, 0) and got[1:] == ref_stats, (ref_stats, got)
    compare_maps(o, h, exact=True)


def test_pipelined_label_error_is_reported_by_the_next_call():
    h = B.HipIntegrator(B.default_config(max_tiles=1024, max_points=1 << 12, pipeline_frames=1, **dict(COMMON, method=0)))
    T = np.array([1, 0, 0, 0, 0, 0, 0], np.float32)
    pts = np.array([[0.5, 0.2, 1.0], [0.4, 0.1, 1.2]], np.float32)
    h.integrate(T, pts, None, np.array([3, 21], np.uint8))      # bad label: found when the frame completes
    with pytest.raises(B.KsError) as e:
        h.flush()
    assert e.value.code == -2  # KS_ERR_LABEL_RANGE
    h.integrate(T, pts, None, np.array([3, 4], np.uint8))       # context still usable
    assert h.flush().n_voxel_updates > 0


def test_profile_levels():
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, pipeline_frames=1, **dict(COMMON, method=0)))
    sc = synth.make_scene("room")
    fr = [synth.render_frame(sc, synth.trajectory_pose(k), 160, 120, seed=k) for k in range(8)]
    h.profile_enable(1)
    tot = 0
    for f in fr:
        tot += h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates
    tot += h.flush().n_voxel_updates
    p = h.profile(reset=True)
    assert p["frames"] == 8 and p["updates"] == tot and p["apply_kernel_launches"] == 8
    assert all(v >= 0 for v in p["ms"].values()) and p["ms"]["march"] > 0 and p["apply_kernel_ms"] > 0
    h.profile_enable(2)
    for f in fr:
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    p = h.profile()
    assert p["frames"] == 8 and p["apply_kernel_launches"] == 2 and sum(p["ms"].values()) == 0
    assert 0 < p["apply_kernel_updates"] < p["updates"]


@pytest.mark.parametrize("variant", ["sorted_order", "anti_grazing", "depth_entry", "clearing"])
def test_pipelined_variants_exact(variant):
    """Configurations that take special paths under ks_config.pipeline_frames: sorted integration
    order (falls back to one frame at a time), anti-grazing (the march stays on stage A's stream),
    the depth-image entry, free-space clouds."""
    method = 0 if variant in ("sorted_order", "depth_entry") else 1
    kw = dict(COMMON, method=method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    if variant == "sorted_order":
        kw["integration_order_mode"] = 1
    if variant == "anti_grazing":
        kw["enable_anti_grazing"] = 1
    o = O.Oracle(O.default_config(**kw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, pipeline_frames=1, **kw))
    sc = synth.make_scene("room")
    for k in range(5):
        f = synth.render_frame(sc, synth.trajectory_pose(3 * k), 160, 120, seed=900 + k)
        free = 1 if (variant == "clearing" and k % 2 == 1) else 0
        if variant == "depth_entry":
            o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)  # f.xyz is the reference-style cloud of f.depth
            h.integrate_depth(f.T_G_C, f.depth, f.K, label_img=f.label_img)
        else:
            o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=free)
            h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=free)
    compare_maps(o, h, exact=True)


def test_pipelined_soak_equals_unpipelined():
    """150 frames through the three-stream pipeline with queries sprinkled in: same map as the
    unpipelined context, bit for bit (deterministic mode), and identical statistics."""
    kw = dict(COMMON, method=0, max_consecutive_ray_collisions=NO_EARLY_OUT)
    a = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 15, pipeline_frames=0, **kw))
    b = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 15, pipeline_frames=1, **kw))
    sc = synth.make_scene("room")
    frames = [synth.render_frame(sc, synth.trajectory_pose(k), 128, 96, seed=k) for k in range(30)]
    sa, sb = [], []
    for k in range(150):
        f = frames[k % 30]
        s = a.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sa.append((s.n_rays_cast, s.n_voxel_updates, s.n_blocks_allocated))
        s = b.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sb.append((s.n_rays_cast, s.n_voxel_updates, s.n_blocks_allocated))
        if k % 37 == 36:
            assert len(b.tile_keys()) == len(a.tile_keys())
    s = b.flush()
    sb.append((s.n_rays_cast, s.n_voxel_updates, s.n_blocks_allocated))
    assert sb[1:] == sa
    compare_maps(a, b, exact=True)


def test_pipelined_pool_exhaustion_is_reported():
    h = B.HipIntegrator(B.default_config(max_tiles=8, max_points=1 << 15, pipeline_frames=1, **dict(COMMON, method=0)))
    f = small_frame(seed=3, w=128, h=96)
    h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    with pytest.raises(B.KsError) as e:
        h.flush()
    assert e.value.code == B.KS_ERR_POOL_FULL if hasattr(B, "KS_ERR_POOL_FULL") else e.value.code < 0
    with pytest.raises(B.KsError):          # the context stays in its failed state
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert len(h.tile_keys()) <= 8          # queries still work; frames in flight were dropped, not applied
    h.clear()                               # and the context is usable again after a clear
    small = np.array([[0.3, 0.1, 1.0]], np.float32)
    h.integrate(np.array([1, 0, 0, 0, 0, 0, 0], np.float32), small, None, np.array([2], np.uint8))
    assert h.flush().n_voxel_updates > 0


def test_pool_exhaustion_with_frames_in_flight():
    """pipeline depth 2: the frame after the one that exhausts the pool is already marching when the
    failure is noticed; it must be dropped, not applied on tiles that do not exist."""
    h = B.HipIntegrator(B.default_config(max_tiles=8, max_points=1 << 15, pipeline_frames=2, **dict(COMMON, method=1)))
    sc = synth.make_scene("room")
    with pytest.raises(B.KsError):
        for k in range(4):
            f = synth.render_frame(sc, synth.trajectory_pose(k), 128, 96, seed=k)
            h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        h.flush()
    assert len(h.tile_keys()) <= 8
    h.synchronize()


from tests.variants import random_combo  # noqa: E402


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("method", [0, 1])
def test_random_knob_combinations_exact(seed, method):
    """Seeded random combinations of the configuration knobs (the same ones the CPU tier checks
    oracle-vs-reference), alternately unpipelined / pipelined: HIP vs oracle, bit-exact."""
    v = random_combo(seed)
    kw = dict(COMMON, method=method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    kw.update(v)
    o = O.Oracle(O.default_config(**kw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 14, pipeline_frames=seed % 2, **kw))
    sc = synth.make_scene("room")
    for k in range(3):
        f = synth.render_frame(sc, synth.trajectory_pose(9 * k + seed), 80, 60, seed=900 + 10 * seed + k)
        fs = (k == 1)
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=fs)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels, freespace=fs)
    compare_maps(o, h, exact=True)


@pytest.mark.parametrize("method", [0, 1])
def test_pipeline_depth_two(method):
    """pipeline_frames = 2: the tail of frame i is enqueued by the call for frame i+2.  Same map,
    statistics lag by two calls, every frame's statistics are handed over exactly once."""
    kw = dict(COMMON, method=method, max_consecutive_ray_collisions=NO_EARLY_OUT)
    o = O.Oracle(O.default_config(**kw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, pipeline_frames=2, **kw))
    sc = synth.make_scene("room")
    ref, got = [], []
    for k in range(9):
        f = synth.render_frame(sc, synth.trajectory_pose(3 * k), 128, 96, seed=1300 + k)
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        ref.append((so.n_points, so.n_rays_cast, so.n_voxel_updates))
        sh = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        got.append((sh.n_points, sh.n_rays_cast, sh.n_voxel_updates))
    assert got[0] == (0, 0, 0) and got[1] == (0, 0, 0) and got[2:] == ref[:7], (ref, got)
    last = h.flush()                       # frames 7 and 8, summed
    assert (last.n_points, last.n_rays_cast, last.n_voxel_updates) == tuple(a + b for a, b in zip(ref[7], ref[8]))
    assert h.flush().n_points == 0
    compare_maps(o, h, exact=True)


def test_clear_then_integrate_equals_fresh_context():
    """ks_clear drops the map AND the integrator's approximate sets: with
    clear_checks_every_n_frames > 1 stale set entries would otherwise suppress rays of the first
    frames after the clear."""
    kw = dict(COMMON, method=0, clear_checks_every_n_frames=3)
    sc = synth.make_scene("room")
    frames = [synth.render_frame(sc, synth.trajectory_pose(2 * k), 128, 96, seed=40 + k) for k in range(4)]
    a = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **kw))
    for f in frames[:2]:
        a.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    a.clear()
    b = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **kw))
    for f in frames[2:]:
        sa = a.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        sb = b.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        assert (sa.n_valid_points, sa.n_rays_cast, sa.n_voxel_updates) == (sb.n_valid_points, sb.n_rays_cast, sb.n_voxel_updates)
    compare_maps(b, a, exact=True)


def test_tile_pool_grows_between_frames():
    """ks_config.max_tiles is only the initial capacity: the pool doubles between frames when more than
    half of it is in use (the reference allocates blocks on demand without a cap).  The map stays bit-exact
    across the re-allocation, in the pipelined mode too."""
    kw = dict(COMMON, method=0, max_consecutive_ray_collisions=NO_EARLY_OUT)
    o = O.Oracle(O.default_config(**kw))
    h = B.HipIntegrator(B.default_config(max_tiles=2048, max_points=1 << 14, pipeline_frames=2, **kw))
    sc = synth.make_scene("room")
    for k in range(40):
        f = synth.render_frame(sc, synth.trajectory_pose(8 * k), 128, 96, seed=500 + k)
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    h.flush()
    assert len(h.tile_keys()) > 2048, len(h.tile_keys())
    compare_maps(o, h, exact=True)


@pytest.mark.parametrize("method,vps", [(0, 16), (1, 32), (0, 8)])
def test_voxel_level_sync_rebuilds_the_map(method, vps):
    """ks_download_updated_voxels (the strict drop-in's per-frame sync): replaying its records frame after frame
    into an empty host map gives exactly the map ks_download_blocks reports, and a second call returns nothing."""
    kw = dict(COMMON, method=method, voxels_per_side=vps, max_consecutive_ray_collisions=NO_EARLY_OUT)
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **kw))
    sc = synth.make_scene("room")
    nv = vps ** 3
    host = {}
    total = 0
    for k in range(3):
        f = synth.render_frame(sc, synth.trajectory_pose(4 * k), 128, 96, seed=70 + k)
        st = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        rec = h.download_updated_voxels()
        assert 0 < len(rec) <= st.n_voxel_updates
        total += len(rec)
        for r in rec:
            key = tuple(int(x) for x in r["block"])
            if key not in host:
                t = np.zeros(nv, dtype=B.TSDF_DTYPE)
                s = np.zeros(nv, dtype=B.SEM_DTYPE)
                s["priors"] = np.float32(-0.60205999132)
                s["color"] = (127, 127, 127, 255)
                host[key] = (t, s)
            host[key][0][r["linear"]] = r["tsdf"]
            host[key][1][r["linear"]] = r["sem"]
        assert len(h.download_updated_voxels()) == 0
    idx, t, s = h.download()
    assert {tuple(x) for x in idx.tolist()} >= set(host)
    for b, key in enumerate(tuple(x) for x in idx.tolist()):
        if key in host:
            assert host[key][0].tobytes() == t[b].tobytes() and host[key][1].tobytes() == s[b].tobytes(), key
        else:  # a block whose tiles were allocated but never updated
            assert not (t[b]["weight"] > 0).any()


@pytest.mark.parametrize("method,early_out", [(0, False), (0, True), (1, False)])
def test_cloud_outgrows_max_points_with_frames_in_flight(method, early_out):
    """A later frame is larger than ks_config.max_points while earlier frames still wait for their tails
    (pipelined): the point buffers are re-allocated only after those frames have completed, and
    stage B's replayed launch sequence is re-captured for the new capacity."""
    kw = dict(COMMON, method=method)
    if early_out:
        kw["early_out_phase_growth"] = 32
    else:
        kw["max_consecutive_ray_collisions"] = NO_EARLY_OUT
    o = O.Oracle(O.default_config(**kw))
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=80 * 60, pipeline_frames=3, **kw))
    sc = synth.make_scene("room")
    sizes = [(80, 60)] * 4 + [(160, 120)] * 3 + [(96, 72)] * 2 + [(200, 150)] * 2
    for k, (w, hh) in enumerate(sizes):
        f = synth.render_frame(sc, synth.trajectory_pose(4 * k), w, hh, seed=2100 + k)
        o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    h.flush()
    compare_maps(o, h, exact=True)


@pytest.mark.parametrize("method,pipe", [(0, 12), (0, 8), (1, 8)])
def test_benched_configuration_map_is_exact(method, pipe):
    """EXACTLY what bench.py times: pipeline_frames = 12 (`fast`, the headline; 8 = rounds 4 / 5 and the sub-record) / 8 (`merged`), 640x480, reference defaults (fast: early-out on, the library's
    default mode = the reference's serial result; merged: reference bundle order), device-pointer entry, 18 trajectory
    frames so that every frame slot, march stream and captured stage-B graph is reused — final map bit-for-bit against
    the oracle (fast: the serial loop, one thread; merged: unordered_map order)."""
    import torch
    kw = dict(COMMON, method=method, voxel_size=0.05, voxels_per_side=16, truncation_distance=0.2, max_ray_length_m=5.0)
    o = O.Oracle(O.default_config(integrator_threads=1, **kw))
    h = B.HipIntegrator(B.default_config(max_tiles=1 << 13, max_points=640 * 480, pipeline_frames=pipe, **kw))
    sc = synth.make_scene("room")
    n_frames = (18 if pipe == 8 else 30) if method == 0 else 8
    upd_o = upd_h = 0
    keep = []
    for k in range(n_frames):
        f = synth.render_frame(sc, synth.trajectory_pose(k), 640, 480, seed=k)
        upd_o += o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates
        d = (torch.from_numpy(f.xyz).cuda(), torch.from_numpy(f.rgba).cuda(), torch.from_numpy(f.labels).cuda())
        keep.append(d)   # inputs stay resident until their (pipelined) frame has been integrated
        torch.cuda.synchronize()
        upd_h += h.integrate_device(f.T_G_C, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[0].shape[0]).n_voxel_updates
    upd_h += h.flush().n_voxel_updates
    assert upd_o == upd_h
    rep = compare_maps(o, h, exact=True)
    assert rep["oracle_touched"] > 200000
    if method == 0:
        st = h.early_out_stats()
        assert st["event_driven"] and st["pipelined"] and st["fallbacks"] == 0, st


@pytest.mark.parametrize("pipe", [8, 12, 16])
@pytest.mark.parametrize("method", [0, 1])
def test_batched_stage_b_equals_unpipelined(method, pipe):
    """pipeline_frames = 8 / 12 / 16: stage B of four / eight consecutive frames is ONE batched launch sequence (blockIdx.y = frame).
    22 / 38 frames (full batches, a partial one at the flush, a query in between that forces a partial batch): same map
    and statistics as the unpipelined context; fast runs the default early-out schedule."""
    kw = dict(COMMON, method=method)
    a = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 15, pipeline_frames=0, **kw))
    b = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 15, pipeline_frames=pipe, **kw))
    sc = synth.make_scene("room")
    ua = ub = 0
    for k in range(22 if pipe == 8 else 30 if pipe == 12 else 38):
        f = synth.render_frame(sc, synth.trajectory_pose(k), 128, 96, seed=k)
        ua += a.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates
        ub += b.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates
        if k == 13:
            assert len(b.tile_keys()) == len(a.tile_keys())   # completes the frames in flight (a partial batch)
    ub += b.flush().n_voxel_updates
    assert ua == ub
    compare_maps(a, b, exact=True)
