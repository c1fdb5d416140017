set -x
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 python -m pytest tests/test_exact_early_out_gpu.py tests/test_parity_gpu.py -m gpu -q -x -k "exact or benched or pipelin or tag_wrap" 2>&1 | tail -4
timeout 300 python tools/exact_tune.py C2 "pipe=8" "pipe=4" "pipe=8,KS_MARCH_STREAMS=1" "pipe=8,KS_BATCH=2" "pipe=8,growth=32" 2>&1 | grep "^C2"
