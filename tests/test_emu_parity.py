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

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "emu", "_build", "libks_hip_emu.so")


@pytest.fixture(scope="module")
def emu_lib():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("host clang++ of the ROCm toolchain not found")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "emu", "build_emu.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return LIB


def run_case(lib, spec, env_extra=None, timeout=900):
    env = dict(os.environ, KS_HIP_LIB=lib)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, "-m", "tests.emu_case", json.dumps(spec)], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0 and "EMU_CASE_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


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


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_code_on_the_host_equals_oracle(emu_lib, name):
    run_case(emu_lib, CASES[name])


def test_late_phases_with_several_sub_runs_per_chain_equal_oracle(emu_lib):
    """320x240 = 75 generations: the phases [32, 64) and [64, 75) have up to two sub-runs per chain, cut over the chain's LIVE rays."""
    run_case(emu_lib, dict(method=0, size=[320, 240], frames=2, max_tiles=8192, cfg=dict(early_out_phase_growth=32)))


@pytest.mark.parametrize("spec", [
    dict(method=0, size=[128, 96], frames=2),                                    # the default: the reference's serial result, event-driven
    dict(method=0, size=[64, 48], frames=4, pipeline=3),                         # frames in flight, commit chain
    dict(method=0, size=[64, 48], frames=5, pipeline=8),                         # batches of four frames per launch (and one left over)
    dict(method=0, size=[64, 48], frames=4, cfg=dict(clear_checks_every_n_frames=3)),   # a frame's marks are inputs of the next frame
    dict(method=0, size=[96, 72], frames=2, cloud="axis", max_tiles=8192),       # axis-parallel rays: the serial caster inside the rounds
    # long rays, pipelining asked for (such contexts run one frame at a time): whole-ray marks, sweeps along the chains (ks_k_exact.h);
    # BOTH frames on the device, no fallback (the mark buffers of such a context start at a third of the longest ray per point)
    dict(method=0, size=[48, 27], frames=2, pipeline=8, max_tiles=32768, cfg=dict(voxel_size=0.02, truncation_distance=0.08, max_ray_length_m=9.0),
         fallbacks_exactly=0),
], ids=["default", "pipelined", "batched", "clear_every_3", "axis_parallel", "long_rays"])
def test_event_driven_exact_early_out_equals_serial_oracle(emu_lib, spec):
    spec = dict(spec)
    run_case(emu_lib, spec, env_extra=spec.pop("env", None))


def test_overflow_falls_back_to_the_host_loop_and_the_device_loop_takes_over_again(emu_lib):
    """Marks that do not fit their buffer: host-driven loop for the frame AND for the frames in flight behind it (their
    predecessor's marks are not in the table when their finisher runs); the next call completes them all once, the buffers
    grow, and the frames after that run on the device again: fewer fallbacks than frames, same map."""
    run_case(emu_lib, dict(method=0, size=[40, 30], frames=6, pipeline=2, fallbacks_below=5),
             env_extra={"KS_EXACT_CAP_MARKS": "8000", "KS_EXACT_CAP_X": "16384"})


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_axis_parallel_rays_under_the_early_out_equal_oracle(emu_lib, overlap):
    """Long rays with zero components (the owner lane's serial caster inside k_test's 64-voxel rounds), with the next round
    cast while the current round's shared-set entries are in flight (default) and one after the other (KS_TEST_OVERLAP=0)."""
    run_case(emu_lib, dict(method=0, size=[96, 72] if overlap == "1" else [64, 48], frames=1, max_tiles=8192, cloud="axis",
                           cfg=dict(early_out_phase_growth=32)), env_extra={"KS_TEST_OVERLAP": overlap})


def test_gpu_tier_cases_unchanged_on_the_functional_model(emu_lib):
    """A selection of the GPU tier's own tests (tests/test_parity_gpu.py), UNCHANGED, against the functional model:
    the reference's CHECKs as error codes, saturated weights, degenerate inputs (NaN / zero-length / out-of-range rays),
    the depth-image entry with u16 depth and colour-coded labels.  (Any other `-m gpu` test runs the same way —
    minutes instead of seconds: KS_TESTS_ON_FUNCTIONAL_MODEL=1 KS_HIP_LIB=tools/emu/_build/libks_hip_emu.so pytest -m gpu -k ...)"""
    env = dict(os.environ, KS_HIP_LIB=emu_lib, KS_TESTS_ON_FUNCTIONAL_MODEL="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_parity_gpu.py", "-m", "gpu", "-q", "-x", "-k",
                        "error_codes or saturated or degenerate or depth_image_u16"], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=1200)
    assert r.returncode == 0 and " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-3000:] + r.stderr[-2000:]
