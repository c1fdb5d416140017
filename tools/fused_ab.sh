cd $GRAFT_REPO_ROOT
env KS_DEBUG=1 timeout 300 python tools/steady_probe.py C3 400 2>&1 | grep -v amdgpu | tail -2
env KS_DEBUG=1 timeout 300 python tools/steady_probe.py C4-merged 72 2>&1 | grep -v amdgpu | tail -1
timeout 1500 python -m pytest tests -m gpu -q -x -n 4 -k "merged or bundle or pipelin" 2>&1 | tail -3
sh tools/ring_trace.sh C3 gpurun_out/fused 3 >/dev/null 2>&1; grep -v "k_bo_\|_pass" gpurun_out/fused/last_frame_C3.txt
