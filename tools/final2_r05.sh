# Round 5, the last GPU call (gpurun, repo root:  bash tools/final2_r05.sh), after the seed's phase growth went from 32 to 22 and
# pipeline_frames = 16 (batches of eight) was added: smoke, the golden / default-mode tests, the headline record (ring of 40, with the
# CPU baseline, without the secondary records: those of profiles/bench_full_r05.json stand — C3 / C4-merged do not use the early-out,
# C4-fast is indifferent to the seed: tools/call13_r05.sh) and the driver's shape.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final2_r05
rm -rf $O; mkdir -p $O
cd $R
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/smoke.txt
timeout 200 python -m pytest tests/test_golden_ref.py tests/test_exact_early_out_gpu.py -m gpu -q -x -k "golden or default_configuration or zero_hash" 2>&1 | tail -3 | tee $O/pytest_subset.txt
KS_BENCH_FULL=$O/bench_full_seed22.json timeout 300 python bench.py --no-secondary > $O/bench_line_seed22.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; cut -c1-900 $O/bench_line_seed22.json
KS_BENCH_FULL=$O/bench_full_driver_shape_seed22.json timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_line_driver_shape_seed22.json 2>/dev/null; cut -c1-600 $O/bench_line_driver_shape_seed22.json
