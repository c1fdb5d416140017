#!/usr/bin/env python3
"""Design study of the event-driven fix point for the serial early-out (oracle/ks_oracle.cpp: ko_sim_fixpoint), CPU only:
   python tools/fixpoint_study.py [640x480|c4geom|c4] [pad] [seed_mode: 0 = full rays, 32 = doubling chain phases, 64, ...] [pose]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kimera_semantics_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.early_out_fidelity import CASES  # noqa: E402
from tests.util import COMMON  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "640x480"
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 32
c = CASES[case]
pose = int(sys.argv[4]) if len(sys.argv) > 4 else c["pose"]
sc = synth.make_scene(c["scene"])
f = synth.render_frame(sc, synth.trajectory_pose(pose), c["w"], c["h"], hfov_deg=c["hfov"], seed=pose)
cfg = O.default_config(**dict(COMMON, method=0, **c["geom"]))
L = O.lib()
L.ko_sim_fixpoint.restype = C.c_size_t
L.ko_sim_fixpoint.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t]
T = np.ascontiguousarray(f.T_G_C, dtype=np.float32)
stats = np.zeros(300, dtype=np.uint64)
r = L.ko_sim_fixpoint(C.byref(cfg), T.ctypes.data, f.xyz.ctypes.data, f.labels.ctypes.data, len(f.xyz), pad, seed, stats.ctypes.data, len(stats))
print("result:", "OVERFLOW" if r == 2**64 - 1 else f"{r} rays wrong")
