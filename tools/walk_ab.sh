cd $GRAFT_REPO_ROOT
env KS_DEBUG=1 timeout 300 python tools/steady_probe.py C3 400 2>&1 | grep -v amdgpu | tail -2
sh tools/ring_trace.sh C3 gpurun_out/walk 3 >/dev/null 2>&1; grep "k_bundles\|# frame" gpurun_out/walk/last_frame_C3.txt
timeout 900 python -m pytest tests/test_merged_keys_gpu.py tests/test_apply_runs_gpu.py -m gpu -q -x -n 4 2>&1 | tail -2
