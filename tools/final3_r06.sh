# Round 6, after the merged stage-A work (32-bit grouping keys, k_bundles_all): whole GPU tier, the two bench lines, the full records.   bash tools/final3_r06.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/final3_r06; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest_gpu.txt
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_shape.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line_driver_shape.json
cp profiles/bench_full_r06.json $O/bench_full_driver_shape.json 2>/dev/null
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line.json; cat $O/bench_line.json
cp profiles/bench_full_r06.json $O/bench_full.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
sh tools/ring_trace.sh C3 $O 3 >/dev/null 2>&1
