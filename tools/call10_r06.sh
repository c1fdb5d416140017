cd $GRAFT_REPO_ROOT
O=gpurun_out/call10; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_apply_runs_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4 > $O/pytest_runs.txt; tail -2 $O/pytest_runs.txt
timeout 600 python -m pytest tests/test_hip_vs_ref_gpu.py tests/test_parity_gpu.py -m gpu -q -x -n 4 -k "full_size_c4 or merged_bit_exact or close_up or long or xlong or sensor" 2>&1 | tail -5 > $O/pytest_c4.txt; tail -2 $O/pytest_c4.txt
sh tools/ring_trace.sh C4-merged $O
KS_BENCH_GROWTH=32 sh tools/ring_trace.sh C4-fast $O
echo == C4-merged; grep "k_apply\|k_find_long\|k_xl\|k_long\|# frame" $O/last_frame_C4-merged.txt
echo == C4-fast-ordered; grep "k_apply\|k_find_long\|k_xl\|k_long\|# frame" $O/last_frame_C4-fast.txt
timeout 900 python bench.py --only-secondary C4-merged,C4-fast-ordered-phases,C3 --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
python - <<'PY'
import json
d = json.loads(open("gpurun_out/call10/bench_line.json").read())
print(d["value"], d["ms_per_step"])
for r in d.get("secondary", []): print(r)
PY
