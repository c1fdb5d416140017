# Round 5: the bench line in the shape the driver runs it (python3 bench.py --gpus 1 --steps 20 --warmup 5) and the default one, after
# bench.py got its untimed first region; plus the exact-mode tests at the last code (k_eo2_full lost a parameter).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/drivershape_r05
rm -rf $O; mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest_exact.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_shape.json 2> $O/bench.err; tail -c 200 $O/bench.err | grep -v amdgpu.ids; cut -c1-700 $O/bench_line_driver_shape.json
cp profiles/bench_full_r05.json $O/bench_full_driver_shape.json
timeout 300 python bench.py --no-secondary --no-cpu-baseline > $O/bench_line_default_nosec.json 2>/dev/null; cut -c1-400 $O/bench_line_default_nosec.json
cp profiles/bench_full_r05.json $O/bench_full_default_nosec.json
