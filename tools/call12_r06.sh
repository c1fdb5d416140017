cd $GRAFT_REPO_ROOT
O=gpurun_out/call12; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_reduce_multiprocess_gpu.py -m gpu -q -x -k "exact_round" 2>&1 | grep -v amdgpu.ids | tail -6 > $O/pytest.txt; tail -3 $O/pytest.txt
KS_DEBUG=1 KS_RUN_THREADS=256 timeout 600 python -m pytest tests/test_apply_runs_gpu.py tests/test_hip_vs_ref_gpu.py -m gpu -q -x -k "lane_per_run or full_size_c4" 2>&1 | tail -3
KS_DEBUG=1 KS_RUN_THREADS=256 sh tools/ring_trace.sh C4-merged $O; cp $O/last_frame_C4-merged.txt $O/last_frame_C4-merged_t256.txt
sh tools/ring_trace.sh C4-merged $O; cp $O/last_frame_C4-merged.txt $O/last_frame_C4-merged_t512.txt
for f in t256 t512; do echo == $f; grep "k_apply\|# frame" $O/last_frame_C4-merged_$f.txt; done
KS_DEBUG=1 KS_RUN_THREADS=256 timeout 900 python bench.py --only-secondary C4-merged,C3,C4-fast-ordered-phases --no-cpu-baseline > $O/bench_t256.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
python - <<'PY'
import json
d = json.loads(open("gpurun_out/call12/bench_t256.json").read())
print("t256", d["value"], d["ms_per_step"])
for r in d.get("secondary", []): print(r)
PY
KS_BENCH_C5=1 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count > $O/bench_c5.json 2> $O/bench_c5.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/call12/bench_c5.json").read())
for r in d.get("secondary", []): print(r)
PY
