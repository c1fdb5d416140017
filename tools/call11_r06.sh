cd $GRAFT_REPO_ROOT
O=gpurun_out/call11; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_apply_runs_gpu.py tests/test_reduce_multiprocess_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -6 > $O/pytest.txt; tail -3 $O/pytest.txt
sh tools/ring_trace.sh C4-merged $O
echo == C4-merged; grep "k_apply\|k_find_long\|k_xl\|k_long\|# frame" $O/last_frame_C4-merged.txt
timeout 900 python bench.py --only-secondary C4-merged,C3 --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
python - <<'PY'
import json
d = json.loads(open("gpurun_out/call11/bench_line.json").read())
print(d["value"], d["ms_per_step"], d.get("frames_per_s"))
for r in d.get("secondary", []): print(r)
PY
KS_BENCH_C5=1 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 600 $O/bench_c5.err | grep -v amdgpu
python - <<'PY'
import json
d = json.loads(open("gpurun_out/call11/bench_c5.json").read())
for r in d.get("secondary", []): print(r)
PY
