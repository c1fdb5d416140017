"""Work per (chain, sub-run) wavefront of k_test and phase, from the CPU checker (oracle: KO_WAVE_STATS=1 -> stderr):
    KO_WAVE_STATS=1 python tools/wave_stats.py
profiles/r03_k_test_wavefront_work_cpu_model.txt holds the three schedules DESIGN.md 3.2 compares."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kimera_semantics_amd import synth
from oracle import oracle_py as O
from tests.util import COMMON
sc = synth.make_scene("room")
o = O.Oracle(O.default_config(early_out_phase_growth=32, **dict(COMMON, method=0)))
for k in (6,7):
    f = synth.render_frame(sc, synth.trajectory_pose(k), 640, 480, seed=k)
    if k==7: sys.stderr.write("==== second frame\n")
    o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
