cd $GRAFT_REPO_ROOT
O=gpurun_out/call2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_apply_runs_gpu.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest_runs.txt; tail -3 $O/pytest_runs.txt
timeout 600 python -m pytest tests/test_hip_vs_ref_gpu.py -m gpu -q -x -k "full_size_c4 or merged_bit_exact" 2>&1 | tail -5 > $O/pytest_c4.txt; tail -2 $O/pytest_c4.txt
for W in C4-merged C3; do
  sh tools/frame_trace.sh $W > $O/frame_$W.log 2>&1; cp gpurun_out/frame_trace_$W/one_frame.txt $O/one_frame_$W.txt
done
KS_BENCH_GROWTH=32 sh tools/frame_trace.sh C4-fast > $O/frame_C4-fast-ordered.log 2>&1; cp gpurun_out/frame_trace_C4-fast/one_frame.txt $O/one_frame_C4-fast-ordered.txt
# A/B old kernel on the same workloads
for W in C4-merged C3; do
  KS_DEBUG=1 KS_APPLY_RUNS=0 sh tools/frame_trace.sh $W > $O/frame_old_$W.log 2>&1; cp gpurun_out/frame_trace_$W/one_frame.txt $O/one_frame_old_$W.txt
done
KS_DEBUG=1 KS_APPLY_RUNS=1 sh tools/frame_trace.sh C2 > $O/frame_runs_C2.log 2>&1; cp gpurun_out/frame_trace_C2/one_frame.txt $O/one_frame_runs_C2.txt
sh tools/frame_trace.sh C2 > $O/frame_C2.log 2>&1; cp gpurun_out/frame_trace_C2/one_frame.txt $O/one_frame_C2.txt
bash tools/sq_pass.sh C4-merged 3 sq_c4_merged_runs > $O/sq.log 2>&1
grep -h "k_apply\|k_find_long" $O/one_frame_*.txt | head -40
