# Round 4: k_bundles_long (weight recurrence through the lanes, operands of the mean recurrence from an LDS table):
# parity on the GPU, then one C3 frame kernel by kernel.
set -x
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_parity_gpu.py tests/test_hip_vs_ref_gpu.py -m gpu -q -x -k "close_up_long_runs or merged_single_frame or merged_bit_exact or merged_colour or long_bundles_edge or degenerate" 2>&1 | tail -4
sh tools/frame_trace.sh C3 2>&1 | grep -v "^+" | tail -48
