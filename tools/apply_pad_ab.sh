cd $GRAFT_REPO_ROOT
for E in "KS_APPLY_STREAM=0" "KS_APPLY_STREAM=1 KS_STREAM_PAD=0" "KS_APPLY_STREAM=1 KS_STREAM_PAD=1" "KS_APPLY_STREAM=1 KS_STREAM_PAD=2" "KS_APPLY_STREAM=1 KS_STREAM_PAD=3"; do
  echo "== $E"
  env KS_DEBUG=1 $E timeout 300 python tools/steady_probe.py C4-merged 72 2>&1 | grep -v amdgpu | tail -1
  env KS_DEBUG=1 $E timeout 300 python tools/steady_probe.py C3 400 2>&1 | grep -v amdgpu | tail -1
done
