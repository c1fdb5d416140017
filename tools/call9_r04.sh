set -x
R=$GRAFT_REPO_ROOT
cd $R
KS_EXACT_DEBUG=1 timeout 400 python tools/exact_tune.py C4-fast "pipe=0,frames=6" 2>&1 | grep -v amdgpu.ids | grep "ks exact\|^C4" | tail -30
timeout 400 python tools/exact_tune.py C4-fast "pipe=8,frames=12" 2>&1 | grep "^C4"
timeout 300 python tools/exact_tune.py C2 "pipe=8,KS_EXACT_SEED_FULL=1" "pipe=0,KS_EXACT_SEED_FULL=1" 2>&1 | grep "^C2"
