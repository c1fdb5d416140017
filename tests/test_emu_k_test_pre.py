"""CPU tier: the early-out kernels of kimera_semantics_amd/csrc/ks_k_march.h run on the host through the functional
model under tools/emu (no GPU): k_test == k_test_pre (on the phases it takes over) == a serial restatement of the
ordered-phase schedule, compared after every phase (per-ray update counts and the shared set's newest entries).
The GPU tier pins k_test against the oracle; this pins k_test_pre against k_test."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("host clang++ of the ROCm toolchain not found")
    out = tmp_path_factory.mktemp("emu") / "test_k_test_pre"
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wno-unknown-attributes", "-Wno-unused-value",
           "-I", os.path.join(ROOT, "tools", "emu"), "-I", os.path.join(ROOT, "kimera_semantics_amd", "csrc"),
           "-o", str(out), os.path.join(ROOT, "tools", "emu", "test_k_test_pre.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    return str(out)


@pytest.mark.parametrize("n,limit,seed,voxel", [(9100, 2, 1, 0.05), (5000, 5, 3, 0.03)])
def test_k_test_pre_equals_k_test_and_serial_schedule(harness, n, limit, seed, voxel):
    r = subprocess.run([harness, str(n), str(limit), str(seed), str(voxel)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert "k_test_pre: identical" in r.stdout and "DIFFERENT" not in r.stdout and r.stdout.strip().endswith("OK")


@pytest.mark.parametrize("legacy", [False, True])
def test_k_test_late_phases_equal_serial_schedule(harness, legacy):
    """117 generations: phases of 32 and 53 generations, i.e. several (chain, sub-run) wavefronts per chain in k_test —
    sub-runs over the chain's live rays (default) and over generations (EMU_SUB_RUN_GENERATIONS: the A/B switch)."""
    env = dict(os.environ)
    if legacy:
        env["EMU_SUB_RUN_GENERATIONS"] = "1"
    n, last = ("70000", "[64,69)") if legacy else ("120000", "[64,118)")
    r = subprocess.run([harness, n, "2", "6", "0.05"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert f"phase {last} k_test    : identical" in r.stdout and "DIFFERENT" not in r.stdout and r.stdout.strip().endswith("OK")
