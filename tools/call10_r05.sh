# Round 5, tenth GPU call: C4-fast after the chain's next ray is looked at whatever its flag says; the C4 tests; the C4-fast bench record
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call10_r05
rm -rf $O; mkdir -p $O
cd $R
env KS_EXACT_TRACE=1 timeout 300 python tools/c4_fast_ab.py 4 0 2>&1 | grep -v amdgpu.ids | grep "ks exact\|ms/frame" | tail -3 | cut -c1-700 | tee $O/c4_trace.txt
timeout 400 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x -k "c4 or full_size" 2>&1 | tail -3 | tee $O/pytest_c4.txt
sh tools/frame_trace.sh C4-fast > $O/c4_frame.log 2>&1; cp gpurun_out/frame_trace_C4-fast/one_frame.txt $O/one_frame_C4-fast.txt
timeout 600 python bench.py --no-cpu-baseline --only-secondary C4-fast > $O/bench_c4fast.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_c4fast.json')); print(d['ms_per_step'], d['secondary'])"
