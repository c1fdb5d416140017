# Round 4: radix sort look-back window 32 for sorts of <= 1024 tiles: sort tests, then one C3 and one C2 frame kernel by kernel.
set -x
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "radix or sort or merged_single_frame or benched" 2>&1 | tail -3
sh tools/frame_trace.sh C3 2>&1 | grep -v "^+" | grep "rs_\|bundles\|apply" | head -30
sh tools/frame_trace.sh C2 2>&1 | grep -v "^+" | tail -60
