cd $GRAFT_REPO_ROOT
for E in "KS_TAIL_PRIO=0" "KS_TAIL_PRIO=1" "KS_TAIL_PRIO=2" "KS_FRONT_PRIO=1" "KS_FRONT_PRIO=1 KS_TAIL_PRIO=1" "KS_FRONT_PRIO=2"; do
  echo "== $E"
  env KS_DEBUG=1 $E timeout 300 python tools/steady_probe.py C4-merged 72 2>&1 | grep -v amdgpu | tail -1
  env KS_DEBUG=1 $E timeout 300 python tools/steady_probe.py C3 400 2>&1 | grep -v amdgpu | tail -1
  env KS_DEBUG=1 $E timeout 300 python tools/steady_probe.py C2 400 2>&1 | grep -v amdgpu | tail -1
done
