cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_merged_keys_gpu.py -m gpu -q -x 2>&1 | tail -3
for E in "KS_KEY_WINDOW_BITS=0" "KS_NONE=1"; do
  echo "== $E"
  env KS_DEBUG=1 $E timeout 300 python tools/steady_probe.py C3 400 2>&1 | grep -v amdgpu | tail -2
  env KS_DEBUG=1 $E timeout 300 python tools/steady_probe.py C4-merged 72 2>&1 | grep -v amdgpu | tail -1
done
timeout 1500 python -m pytest tests -m gpu -q -x -n 4 -k "merged or bundle or pipelin" 2>&1 | tail -3
sh tools/ring_trace.sh C3 gpurun_out/keys 3 >/dev/null 2>&1; grep -v "k_bo_" gpurun_out/keys/last_frame_C3.txt | head -20
