"""CPU tier: properties of the gfx950 code hipcc generates for the kernels (no GPU: `--cuda-device-only -S`).  What
is pinned here was found by reading that code, costs time on the GPU when it comes back, and is invisible to every parity test:
  * no kernel keeps anything in scratch memory (selecting a struct field by lane number once pinned the ray caster's
    state there: a memory round trip per 64-voxel round in seven kernels, ks_types.h KS_VALUE_BARRIER);
  * k_test reaches the shared set of the early-out with GLOBAL instructions only (FLAT loads count as LDS operations too:
    every LDS wait of the caster then waits for the set's entries in flight, ks_k_march.h obs_global_u64)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kimera_semantics_amd", "csrc")


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not found")
    out = str(tmp_path_factory.mktemp("isa") / "ks_hip.s")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]   # = csrc/Makefile
    r = subprocess.run(["hipcc", *flags, "--cuda-device-only", "-S", "-o", out, "ks_hip.hip"], cwd=CSRC, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return open(out).read()


def test_no_kernel_uses_scratch_memory(asm):
    per_kernel = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", asm)
    assert len(per_kernel) > 50
    bad = {k: int(v) for k, v in per_kernel if int(v) != 0}
    assert not bad, bad
    assert "scratch_load" not in asm and "scratch_store" not in asm


def test_k_test_reaches_the_shared_set_with_global_instructions(asm):
    bodies = re.findall(r"^(_ZN3ksk6k_testILb[01]EE\w+):[^\n]*\n(.*?)^\.Lfunc_end", asm, re.S | re.M)
    assert len(bodies) == 2
    for name, body in bodies:
        assert "flat_atomic" not in body and "flat_load_dwordx4" not in body and "flat_store" not in body, name
        assert body.count("global_atomic_umax_x2") >= 4, name     # the saves of phase B + the marks
        # (the one FLAT load left reads the integration order of the sorted mode through a pointer kept in FrameParams)
        assert len(re.findall(r"\bflat_load", body)) <= 1, name


def test_k_test_casts_the_next_round_while_the_look_ups_are_in_flight(asm):
    """k_test<OVERLAP = true>: between the request for a round's shared-set entries (the last 16-byte load of the kernel) and the
    wait for them lies the ray caster's accumulation of the NEXT round (its 65 crossing times go to LDS two at a time) — the
    compiler has neither sunk the load nor hoisted the wait.  k_test<false> waits right after the request."""
    for overlap, want in (("1", True), ("0", False)):
        m = re.search(r"^_ZN3ksk6k_testILb" + overlap + r"EE\w+:[^\n]*\n(.*?)^\.Lfunc_end", asm, re.S | re.M)
        lines = m.group(1).split("\n")
        loads = [i for i, l in enumerate(lines) if "global_load_dwordx4" in l]
        assert len(loads) >= 5        # four of phase B, one per round of a long ray
        after = lines[loads[-1] + 1:]
        wait = next(i for i, l in enumerate(after) if "s_waitcnt" in l and "vmcnt(0)" in l)
        caster = sum("ds_write2_b32" in l for l in after[:wait])
        assert (caster >= 5) == want, (overlap, caster, wait)


def test_k_bundles_long_runs_the_weight_recurrence_through_the_lanes(asm):
    """k_bundles_long: the weights of a 64-point batch are summed in order by 63 DPP steps (wave_shr:1: lane k takes lane k-1's
    sum) plus the one that hands every lane the weight before its point — no LDS round trip, no scalar broadcast per point."""
    m = re.search(r"^_ZN3ksk14k_bundles_long\w+:[^\n]*\n(.*?)^\.Lfunc_end", asm, re.S | re.M)
    body = m.group(1)
    assert body.count("wave_shr:1") >= 64
    assert "v_readlane_b32" not in body.split("wave_shr:1")[1]     # nothing scalar between two steps of the chain
