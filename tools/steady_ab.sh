cd $GRAFT_REPO_ROOT
for E in "KS_BUNDLE_STREAM=0 KS_EMIT_ON_TAIL=0" "KS_BUNDLE_STREAM=1 KS_EMIT_ON_TAIL=0 KS_STREAM_PAD=2" "KS_BUNDLE_STREAM=1 KS_STREAM_PAD=2" "KS_BUNDLE_STREAM=2"; do
  echo "== $E"
  env KS_DEBUG=1 KS_HOST_PROF=1 $E timeout 300 python tools/steady_probe.py C3 400 2>&1 | grep -v amdgpu
done
echo "== C4-merged"
env KS_DEBUG=1 KS_HOST_PROF=1 KS_BUNDLE_STREAM=0 KS_EMIT_ON_TAIL=0 timeout 600 python tools/steady_probe.py C4-merged 72 2>&1 | grep -v amdgpu
