# Round 5, first GPU call (gpurun, repo root:  bash tools/call1_r05.sh): everything under the re-pinned "mixed" order.
#   1. the whole GPU tier + smoke at HEAD
#   2. the default bench line (every region = the same K frames now)
#   3. rocprofv3 --kernel-trace --stats of the headline command; one unpipelined C2 frame kernel by kernel
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call1_r05
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/smoke.txt
timeout 500 python bench.py --steps 20 --warmup 2 > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line.json; head -c 3000 $O/bench_line.json; echo
cp profiles/bench_full_r05.json $O/bench_full.json 2>/dev/null
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
cd /tmp && export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast -o run -- python $R/bench.py $BENCH > $O/fast.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/fast "python bench.py $BENCH" $O/fast.log > $O/fast_kernel_stats.txt 2>&1)
cd $R
sh tools/frame_trace.sh C2 > $O/c2_frame.log 2>&1; cp gpurun_out/frame_trace_C2/one_frame.txt $O/c2_one_frame.txt
find $O -name "*.csv" -size +2M -delete
cut -c1-150 $O/fast_kernel_stats.txt | head -50
