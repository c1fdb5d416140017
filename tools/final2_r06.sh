# Round 6: the two bench lines again after the whole-frame traffic of profiles/r06_pmc_*.json was recomputed per FRAME (k_points_* launches,
# not k_publish: one per batch) and the C2-depth-host-inputs record was added.   bash tools/final2_r06.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/final2_r06; rm -rf $O; mkdir -p $O
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_shape.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line_driver_shape.json
cp profiles/bench_full_r06.json $O/bench_full_driver_shape.json 2>/dev/null
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line.json; cat $O/bench_line.json
cp profiles/bench_full_r06.json $O/bench_full.json 2>/dev/null
KS_BENCH_C5=1 timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count > $O/bench_c5_world1.json 2> $O/bench_c5.err
timeout 600 python -m pytest tests/test_reduce_multiprocess_gpu.py tests/test_host_logic.py -q -x -m "gpu or not gpu" 2>&1 | tail -3
