cd $GRAFT_REPO_ROOT
O=gpurun_out/call9; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_reduce_multiprocess_gpu.py -m gpu -q -x -k "exact_round" 2>&1 | grep -v amdgpu.ids | tail -30 > $O/pytest_round.txt; tail -8 $O/pytest_round.txt
timeout 900 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x -k "full_reset" --durations=3 2>&1 | grep -v amdgpu.ids | tail -8 > $O/pytest_reset.txt; tail -5 $O/pytest_reset.txt
sh tools/ring_trace.sh C4-merged $O
echo == C4-merged; grep "k_apply\|k_find_long\|k_xl\|k_long\|# frame" $O/last_frame_C4-merged.txt
