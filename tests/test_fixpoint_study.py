"""The CPU study of the event-driven fix point (oracle/ks_oracle.cpp: ko_sim_fixpoint; tools/fixpoint_study.py prints its
per-round statistics): from the ordered-phase seed, re-evaluating only the rays whose input changed and dirtying only the
reader behind every mark that flipped reaches EXACTLY the lengths of the reference's serial loop
(semantic_tsdf_integrator_fast.cpp:110-122) — the argument the GPU kernels of csrc/ks_k_exact.h rest on, checked here without
a GPU, in place and as Jacobi rounds (what the kernels do), with and without removing the dead marks after round 0."""
import ctypes as C

import numpy as np
import pytest

from kimera_semantics_amd import synth
from oracle import oracle_py as O
from tests.util import COMMON


@pytest.mark.parametrize("mode", ["in_place", "jacobi", "jacobi_compact"])
@pytest.mark.parametrize("seed_growth", [32, 256])
def test_event_driven_fix_point_reaches_the_serial_lengths(monkeypatch, mode, seed_growth):
    monkeypatch.delenv("KO_STUDY_JACOBI", raising=False)
    monkeypatch.delenv("KO_STUDY_COMPACT", raising=False)
    if mode.startswith("jacobi"):
        monkeypatch.setenv("KO_STUDY_JACOBI", "1")
    if mode == "jacobi_compact":
        monkeypatch.setenv("KO_STUDY_COMPACT", "1")
    sc = synth.make_scene("room")
    f = synth.render_frame(sc, synth.trajectory_pose(7), 200, 150, seed=7)
    cfg = O.default_config(**dict(COMMON, method=0))
    L = O.lib()
    L.ko_sim_fixpoint.restype = C.c_size_t
    L.ko_sim_fixpoint.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
    T = np.ascontiguousarray(f.T_G_C, dtype=np.float32)
    stats = np.zeros(300, dtype=np.uint64)
    wrong = L.ko_sim_fixpoint(C.byref(cfg), T.ctypes.data, f.xyz.ctypes.data, f.labels.ctypes.data, len(f.xyz), 0, seed_growth, stats.ctypes.data, len(stats))
    assert wrong == 0
    rounds = int((stats[0::3] > 0).sum())
    assert 2 <= rounds <= 40 and int(stats[0]) > 5000          # every ray in round 0, then a shrinking tail
    assert int(stats[3]) < int(stats[0]) * (1 if seed_growth > 32 else 0.5)   # round 1 looks at a fraction of the rays (a coarse seed: most of them)
