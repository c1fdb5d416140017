# Round 4: tuning sweep of the exact mode's pipelining (march streams, hardware queues, bulk rounds)
set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x 2>&1 | tail -3
timeout 300 python tools/exact_tune.py C2 "pipe=0" "pipe=8" "pipe=8,KS_MARCH_STREAMS=2" "pipe=8,KS_MARCH_STREAMS=3" "pipe=8,KS_MARCH_STREAMS=4" "pipe=8,KS_EXACT_BULK_ROUNDS=14" "pipe=8,KS_EXACT_BULK_ROUNDS=14,KS_MARCH_STREAMS=3" "pipe=4" "pipe=8,growth=32" 2>&1 | grep "^C2"
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/exact_tune.py C2 "pipe=8" "pipe=8,KS_MARCH_STREAMS=4" "pipe=8,KS_MARCH_STREAMS=6" "pipe=8,growth=32" 2>&1 | grep "^C2" | sed 's/^/HWQ8 /'
GPU_MAX_HW_QUEUES=12 timeout 300 python tools/exact_tune.py C2 "pipe=8" "pipe=8,KS_MARCH_STREAMS=6" 2>&1 | grep "^C2" | sed 's/^/HWQ12 /'
KS_EXACT_DEBUG=1 timeout 300 python tools/exact_tune.py C4-fast "pipe=0,frames=8" 2>&1 | grep -v amdgpu.ids | tail -12
