"""CPU tier: the DEVICE CODE ITSELF against the oracle, without a GPU.  tools/emu/build_emu.sh compiles the whole
library (ks_hip.hip + every kernel header) for the host against a stand-in <hip/hip_runtime.h> (work-items are fibers,
wave collectives rendezvous points, device memory host memory); the same C ABI, the same Python binding, the same
bit-exact comparison as the GPU tier — on frames small enough for a functional model (seconds per frame).  What this
pins without hardware: indexing, control flow, LDS layouts, the f32 arithmetic order, the host orchestration.  What
only the GPU tier can: the memory model between workgroups, timing, the real ISA."""
import json
import os
import subprocess
import sys
import threading
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "emu", "_build", "libks_hip_emu.so")

# Every test of this module is ONE child process (seconds to a minute of single-threaded emulation): JOBS maps a test's
# node name to its command line, the `emu_jobs` fixture starts the selected ones side by side (one per core) as soon as the
# library is built, and a test only collects its own child's verdict — the same cases, the same assertions, a fifth of the
# wall-clock time on 8 cores.
JOBS = {}   # pytest node name -> (argv, extra environment, timeout in seconds, expected seconds: the long ones start first)


def case_job(node, spec, env_extra=None, timeout=900, weight=10):
    JOBS[node] = ([sys.executable, "-m", "tests.emu_case", json.dumps(spec)], dict(env_extra or {}), timeout, weight)


class _Jobs:
    def __init__(self, lib, nodes):
        self.lib, self.lock, self.children = lib, threading.Lock(), set()
        # one child per core — of THIS worker's share of them when pytest-xdist runs several workers side by side
        share = max(1, int(os.environ.get("PYTEST_XDIST_WORKER_COUNT", "1")))
        self.pool = ThreadPoolExecutor(max_workers=max(1, min((os.cpu_count() or 1) // share, 16)))
        self.futures = {n: self.pool.submit(self._run, *JOBS[n][:3]) for n in nodes}

    def _run(self, argv, env_extra, timeout):
        child = subprocess.Popen(argv, cwd=ROOT, env=dict(os.environ, KS_HIP_LIB=self.lib, **env_extra), stdout=subprocess.PIPE,
                                 stderr=subprocess.PIPE, text=True)
        with self.lock:
            self.children.add(child)
        try:
            out, err = child.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            child.kill()
            out, err = child.communicate()
            err += "\n[timed out after %d s]" % timeout
        finally:
            with self.lock:
                self.children.discard(child)
        return child.returncode, out, err

    def result(self, node):
        if node not in self.futures:   # (a test selected in a way the fixture did not foresee: run it now)
            self.futures[node] = self.pool.submit(self._run, *JOBS[node][:3])
        return self.futures[node].result()

    def close(self):
        for f in self.futures.values():
            f.cancel()
        with self.lock:
            for child in list(self.children):   # (exactly the children started here)
                child.kill()
        self.pool.shutdown(wait=True)


@pytest.fixture(scope="module")
def emu_jobs(request):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("host clang++ of the ROCm toolchain not found")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "emu", "build_emu.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    mine = [it.name for it in request.session.items if str(it.fspath) == str(request.fspath) and it.name in JOBS]
    jobs = _Jobs(LIB, sorted(mine, key=lambda n: -JOBS[n][3]))   # (the long ones first)
    yield jobs
    jobs.close()


def check_case(emu_jobs, request):
    rc, out, err = emu_jobs.result(request.node.name)
    assert rc == 0 and "EMU_CASE_OK" in out, out[-3000:] + err[-3000:]


CASES = {
    "fast_no_early_out": dict(method=0, size=[64, 48], frames=2, no_early_out=True),
    "merged_reference_bundle_order": dict(method=1, size=[64, 48], frames=2),
    "merged_colour_mode_anti_grazing": dict(method=1, size=[64, 48], frames=1, cfg=dict(color_mode=0, enable_anti_grazing=1)),
    # (the oracle restates the ordered-phase schedule when early_out_phase_growth >= 16; 0 = the reference's serial loop)
    "fast_ordered_phases_pipelined": dict(method=0, size=[96, 72], frames=3, pipeline=2, cfg=dict(early_out_phase_growth=32)),
    "fast_exact_serial_early_out": dict(method=0, size=[64, 48], frames=1, exact=True),
    "fast_sorted_order_limit_0": dict(method=0, size=[64, 48], frames=1,
                                      cfg=dict(integration_order_mode=1, max_consecutive_ray_collisions=0, early_out_phase_growth=32)),
    # the other reading of Voxblox's "mixed" order (1024 groups of N / 1024 points): schedule alone, and the serial result
    "fast_ordered_phases_mixed_1024_groups": dict(method=0, size=[96, 72], frames=2, cfg=dict(early_out_phase_growth=32, integration_order_mode=2)),
    "fast_exact_mixed_1024_groups": dict(method=0, size=[96, 72], frames=2, exact=True, cfg=dict(integration_order_mode=2)),
    "merged_mixed_1024_groups": dict(method=1, size=[64, 48], frames=2, cfg=dict(integration_order_mode=2)),
    "random_knobs_fast": dict(method=0, size=[64, 48], frames=2, random_combo=3, cfg=dict(early_out_phase_growth=32)),
    "random_knobs_merged": dict(method=1, size=[64, 48], frames=2, random_combo=5),
    # k_bundles_long (two waves per >= 32-point bundle: the first frame looks at a wall from 0.45 m) and k_apply_xlong (four waves
    # per run of more than 1024 updates: the voxels next to the sensor once a frame has more than 1024 bundles), frames in flight
    "merged_long_bundles_and_long_runs": dict(method=1, size=[192, 144], frames=3, pipeline=2, max_tiles=8192, close_up_first=True),
}


for _name, _spec in CASES.items():
    case_job("test_device_code_on_the_host_equals_oracle[%s]" % _name, _spec, weight=50 if "long_bundles" in _name else 10)


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_code_on_the_host_equals_oracle(emu_jobs, request, name):
    check_case(emu_jobs, request)


case_job("test_late_phases_with_several_sub_runs_per_chain_equal_oracle",
         dict(method=0, size=[320, 240], frames=2, max_tiles=8192, cfg=dict(early_out_phase_growth=32)), weight=30)


def test_late_phases_with_several_sub_runs_per_chain_equal_oracle(emu_jobs, request):
    """320x240 = 75 generations: the phases [32, 64) and [64, 75) have up to two sub-runs per chain, cut over the chain's LIVE rays."""
    check_case(emu_jobs, request)


EXACT_CASES = {
    "default": dict(method=0, size=[128, 96], frames=2),                                    # the default: the reference's serial result, event-driven
    "pipelined": dict(method=0, size=[64, 48], frames=4, pipeline=3),                       # frames in flight, commit chain
    "batched": dict(method=0, size=[64, 48], frames=5, pipeline=8),                         # batches of four frames per launch (and one left over)
    "batched_8": dict(method=0, size=[48, 36], frames=10, pipeline=16),                     # batches of eight (24 slots, 16 tables) and a partial one at the flush
    "clear_every_3": dict(method=0, size=[64, 48], frames=4, cfg=dict(clear_checks_every_n_frames=3)),   # a frame's marks are inputs of the next frame
    "axis_parallel": dict(method=0, size=[96, 72], frames=2, cloud="axis", max_tiles=8192),   # axis-parallel rays: the serial caster inside the rounds
    # long rays, pipelining asked for (such contexts run one frame at a time): whole-ray marks, sweeps along the chains (ks_k_exact.h);
    # BOTH frames on the device, no fallback (the mark buffers of such a context start at a third of the longest ray per point)
    "long_rays": dict(method=0, size=[48, 27], frames=2, pipeline=8, max_tiles=32768,
                      cfg=dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=9.0), fallbacks_exactly=0),
}
for _name, _spec in EXACT_CASES.items():
    case_job("test_event_driven_exact_early_out_equals_serial_oracle[%s]" % _name, _spec, weight=110 if _name == "batched_8" else 60)


@pytest.mark.parametrize("name", list(EXACT_CASES))
def test_event_driven_exact_early_out_equals_serial_oracle(emu_jobs, request, name):
    check_case(emu_jobs, request)


case_job("test_overflow_falls_back_to_the_host_loop_and_the_device_loop_takes_over_again",
         dict(method=0, size=[40, 30], frames=6, pipeline=2, fallbacks_below=5),
         env_extra={"KS_DEBUG": "1", "KS_EXACT_CAP_MARKS": "8000", "KS_EXACT_CAP_X": "16384"}, weight=65)


def test_overflow_falls_back_to_the_host_loop_and_the_device_loop_takes_over_again(emu_jobs, request):
    """Marks that do not fit their buffer: host-driven loop for the frame AND for the frames in flight behind it (their
    predecessor's marks are not in the table when their finisher runs); the next call completes them all once, the buffers
    grow, and the frames after that run on the device again: fewer fallbacks than frames, same map."""
    check_case(emu_jobs, request)


for _overlap in ("1", "0"):
    case_job("test_axis_parallel_rays_under_the_early_out_equal_oracle[%s]" % _overlap,
             dict(method=0, size=[96, 72] if _overlap == "1" else [64, 48], frames=1, max_tiles=8192, cloud="axis",
                  cfg=dict(early_out_phase_growth=32)), env_extra={"KS_DEBUG": "1", "KS_TEST_OVERLAP": _overlap}, weight=15)


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_axis_parallel_rays_under_the_early_out_equal_oracle(emu_jobs, request, overlap):
    """Long rays with zero components (the owner lane's serial caster inside k_test's 64-voxel rounds), with the next round
    cast while the current round's shared-set entries are in flight (default) and one after the other (KS_TEST_OVERLAP=0)."""
    check_case(emu_jobs, request)


# k_apply_runs (a lane per voxel run, runs bucketed by length in a tile of 2048 pairs) instead of k_apply, and k_apply_long_lanes
# (the runs of 33 .. 256 updates a lane per run, bucketed by length over the frame) beside k_apply_long: forced for frames of
# any size (the library takes them from 2^20 / 2^24 pairs per frame on); the records must be the same bit for bit
RUNS_CASES = {
    "fast_no_early_out": dict(method=0, size=[64, 48], frames=2, no_early_out=True),
    "fast_colour_blend": dict(method=0, size=[64, 48], frames=2, no_early_out=True, cfg=dict(color_mode=0)),
    "fast_default_pipelined": dict(method=0, size=[64, 48], frames=3, pipeline=2),
    "merged": dict(method=1, size=[64, 48], frames=2),
    "merged_colour_blend": dict(method=1, size=[64, 48], frames=2, cfg=dict(color_mode=0)),
    "merged_probability_colours_random_knobs": dict(method=1, size=[64, 48], frames=2, random_combo=5),
    "merged_close_up_long_runs_beside": dict(method=1, size=[160, 120], frames=2, pipeline=2, max_tiles=8192, close_up_first=True),
    # four frames from one pose, weights saturating at once: the sensor voxel's runs of more than 1024 updates through the chunked
    # integer sums of ks_k_apply_xl.h (binade crossings and replays included), the first frame's through k_apply_xlong
    "merged_sensor_voxel_integer_sums": dict(method=1, size=[96, 72], frames=4, fixed_pose=True, cfg=dict(max_weight=2.0), xl_walked_at_least=1),
}
for _name, _spec in RUNS_CASES.items():
    case_job("test_lane_per_run_update_kernel_equals_oracle[%s]" % _name, _spec, env_extra={"KS_DEBUG": "1", "KS_APPLY_RUNS": "1", "KS_LONG_LANES": "2", "KS_XL_PARALLEL": "2"},
             weight=40 if "close_up" in _name else 10)


@pytest.mark.parametrize("name", sorted(RUNS_CASES))
def test_lane_per_run_update_kernel_equals_oracle(emu_jobs, request, name):
    check_case(emu_jobs, request)


# merged stage A groups the points by a 32-bit key (the end voxel inside a window around the sensor, anything else through a hash
# table: FrameParams::key_bits): a window of 3 bits per axis (nearly every voxel through the table), short rays (clearing points far
# outside the window), and the 64-bit keys of rounds 1-5 — all the same map
KEY_CASES = {
    "window_of_3_bits": (dict(method=1, size=[64, 48], frames=2), {"KS_DEBUG": "1", "KS_KEY_WINDOW_BITS": "3"}),
    "window_of_3_bits_pipelined_colour_blend": (dict(method=1, size=[64, 48], frames=3, pipeline=2, cfg=dict(color_mode=0)),
                                                {"KS_DEBUG": "1", "KS_KEY_WINDOW_BITS": "3"}),
    "clearing_points_outside_the_window": (dict(method=1, size=[64, 48], frames=2, cfg=dict(max_ray_length_m=1.5)), {}),
    "sorted_64_bit_keys": (dict(method=1, size=[64, 48], frames=2), {"KS_DEBUG": "1", "KS_KEY_WINDOW_BITS": "0"}),
    # more than kBoSmallBuckets bundles: the large epochs of the bundle order, every one of them through k_bo_rest (hint 1) / through
    # the launches a frame of n points could need (hint 0)
    "bundle_order_epochs_beyond_the_hint": (dict(method=1, size=[224, 168], frames=2, max_tiles=4096, rays_at_least=8000), {"KS_DEBUG": "1", "KS_BO_HINT": "1"}),
    "bundle_order_epochs_for_n_points": (dict(method=1, size=[224, 168], frames=2, max_tiles=4096, rays_at_least=8000), {"KS_DEBUG": "1", "KS_BO_HINT": "0"}),
}
for _name, (_spec, _env) in KEY_CASES.items():
    case_job("test_merged_grouping_keys_equal_oracle[%s]" % _name, _spec, env_extra=_env, weight=30 if "bundle_order" in _name else 10)


@pytest.mark.parametrize("name", sorted(KEY_CASES))
def test_merged_grouping_keys_equal_oracle(emu_jobs, request, name):
    check_case(emu_jobs, request)


JOBS["test_gpu_tier_cases_unchanged_on_the_functional_model"] = (
    [sys.executable, "-m", "pytest", "tests/test_parity_gpu.py", "-m", "gpu", "-q", "-x", "-k",
     "error_codes or saturated or degenerate or depth_image_u16"], {"KS_TESTS_ON_FUNCTIONAL_MODEL": "1"}, 1200, 55)


def test_gpu_tier_cases_unchanged_on_the_functional_model(emu_jobs, request):
    """A selection of the GPU tier's own tests (tests/test_parity_gpu.py), UNCHANGED, against the functional model:
    the reference's CHECKs as error codes, saturated weights, degenerate inputs (NaN / zero-length / out-of-range rays),
    the depth-image entry with u16 depth and colour-coded labels.  (Any other `-m gpu` test runs the same way —
    minutes instead of seconds: KS_TESTS_ON_FUNCTIONAL_MODEL=1 KS_HIP_LIB=tools/emu/_build/libks_hip_emu.so pytest -m gpu -k ...)"""
    rc, out, err = emu_jobs.result(request.node.name)
    assert rc == 0 and " passed" in out and "skipped" not in out.splitlines()[-1], out[-3000:] + err[-2000:]
