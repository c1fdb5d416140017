cd $GRAFT_REPO_ROOT
O=gpurun_out/call8; rm -rf $O; mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -q -x -n 4 --durations=8 2>&1 | tail -16 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
python - <<'PY' > $O/small_frame_cost.txt 2>&1
import time, sys
sys.path.insert(0, '.')
from kimera_semantics_amd import binding as B, synth
from oracle import oracle_py as O
from tests.util import COMMON
okw = dict(COMMON, method=0)
o = O.Oracle(O.default_config(integrator_threads=1, **okw))
h = B.HipIntegrator(B.default_config(max_tiles=8192, max_points=1 << 14, pipeline_frames=8, **okw))
sc = synth.make_scene("room")
small = [synth.render_frame(sc, synth.trajectory_pose(k), 48, 36, seed=k) for k in range(8)]
t0 = time.time()
for k in range(500): o.integrate(*[getattr(small[k % 8], a) for a in ("T_G_C", "xyz", "rgba", "labels")])
t1 = time.time()
for k in range(500): h.integrate(*[getattr(small[k % 8], a) for a in ("T_G_C", "xyz", "rgba", "labels")])
h.flush(); t2 = time.time()
print(f"oracle {1e3*(t1-t0)/500:.2f} ms/frame, hip pipelined {1e3*(t2-t1)/500:.2f} ms/frame")
PY
grep -v amdgpu $O/small_frame_cost.txt
sh tools/ring_trace.sh C4-merged $O
sh tools/ring_trace.sh C3 $O
echo == C4-merged; grep "k_apply\|k_find_long\|k_xl\|k_long\|# frame\|k_emit\|rs_hist" $O/last_frame_C4-merged.txt
echo == C3; grep "k_apply\|k_find_long\|k_xl\|k_long\|# frame" $O/last_frame_C3.txt
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 400 $O/bench.err | grep -v amdgpu; cat $O/bench_line.json | cut -c1-3000
cp profiles/bench_full_r05.json $O/bench_full.json 2>/dev/null
