# Round 4: k_apply_xlong (four waves per run of more than 1024 updates): parity, then one C3 and one C4-merged frame kernel by kernel
set -x
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_parity_gpu.py tests/test_hip_vs_ref_gpu.py -m gpu -q -x -k "runs_next_to_the_sensor or close_up_long_runs or merged_bit_exact or merged_colour" 2>&1 | tail -3
sh tools/frame_trace.sh C3 2>&1 | grep -v "^+" | grep "apply\|find_long\|bundles" | head
sh tools/frame_trace.sh C4-merged 2>&1 | grep -v "^+" | grep "apply\|find_long\|bundles\|emit" | head
