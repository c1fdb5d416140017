cd $GRAFT_REPO_ROOT
O=gpurun_out/call6; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_apply_runs_gpu.py -m gpu -q -x 2>&1 | grep -v amdgpu.ids | tail -4 > $O/pytest_runs.txt; tail -2 $O/pytest_runs.txt
timeout 600 python -m pytest tests/test_hip_vs_ref_gpu.py tests/test_parity_gpu.py -m gpu -q -x -n 4 -k "full_size_c4 or merged_bit_exact or close_up or long or xlong or sensor" 2>&1 | tail -5 > $O/pytest_c4.txt; tail -2 $O/pytest_c4.txt
sh tools/ring_trace.sh C4-merged $O
sh tools/ring_trace.sh C3 $O
cp $O/last_frame_C4-merged.txt $O/last_frame_C4-merged_lanes.txt
KS_DEBUG=1 KS_LONG_LANES=0 sh tools/ring_trace.sh C4-merged $O; cp $O/last_frame_C4-merged.txt $O/last_frame_C4-merged_nolanes.txt
KS_DEBUG=1 KS_LONG_LANES=0 KS_XL_PARALLEL=0 KS_APPLY_RUNS=0 sh tools/ring_trace.sh C4-merged $O; cp $O/last_frame_C4-merged.txt $O/last_frame_C4-merged_r05kernels.txt
for f in lanes nolanes r05kernels; do echo == $f; grep "k_apply\|k_find_long\|k_xl\|k_long\|# frame" $O/last_frame_C4-merged_$f.txt; done
echo == C3; grep "k_apply\|k_find_long\|k_xl\|k_long\|# frame" $O/last_frame_C3.txt
