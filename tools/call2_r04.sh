# Round 4, second GPU call: the event-driven exact early-out on hardware (parity tests, then timings over its settings)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04b
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x --durations=8 2>&1 | tail -25
timeout 600 python tools/exact_tune.py C2 "pipe=0,growth=32" "pipe=8,growth=32" "pipe=0" "pipe=4" "pipe=8" "pipe=8,KS_EXACT_BULK_ROUNDS=4" "pipe=8,KS_EXACT_BULK_ROUNDS=10" "pipe=8,KS_EXACT_BULK_ROUNDS=14" "pipe=8,KS_EXACT_SEED_GROWTH=64" "pipe=8,KS_EXACT_SEED_GROWTH=256" "pipe=8,KS_EXACT_SEED_GROWTH=4096" "pipe=8,KS_MARCH_STREAMS=4" "pipe=0,KS_EXACT_HOST_LOOP=1" 2>&1 | tee $O/tune_c2.txt | grep -v amdgpu.ids
timeout 600 python tools/exact_tune.py C4-fast "pipe=8,growth=32" "pipe=8" "pipe=0" 2>&1 | tee $O/tune_c4.txt | grep -v amdgpu.ids
