# Round 5: refresh of what the last code changes touch (20 fix-point rounds as launches; the sweeps read their flags in parallel):
# targeted tests, C4-fast, the bench line, kernel stats + PMC passes of the headline command and of C4-fast.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh_r05
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_exact_early_out_gpu.py tests/test_golden_ref.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest_exact.txt
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_host_adapter_gpu.py -m gpu -q -x -k "benched or early_out or patched_server or real_factory" 2>&1 | tail -4 | tee $O/pytest_subset.txt
env KS_EXACT_TRACE=1 timeout 300 python tools/c4_fast_ab.py 4 0 2>&1 | grep -v amdgpu.ids | grep "ks exact\|ms/frame" | tail -3 | cut -c1-700 | tee $O/c4_trace.txt
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast -o run -- python $R/bench.py $BENCH > $O/fast.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/fast "python bench.py $BENCH" $O/fast.log > $O/fast_kernel_stats.txt 2>&1)
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/time_C4-fast -o run -- python $R/tools/probe.py C4-fast 4 > $O/time_C4-fast.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_fast_$C -o run -- python $R/bench.py $BENCH > $O/pmc_fast_$C.log 2>&1
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_C4-fast_$C -o run -- python $R/tools/probe.py C4-fast 4 > $O/pmc_C4-fast_$C.log 2>&1
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_calib_$C -o run -- python $R/tools/probe.py calib 3 > $O/pmc_calib_$C.log 2>&1
done
cd $R
PMC_TAG=r05 PMC_SCRIPT=refresh_r05.sh python tools/pmc_r03.py $O | tee $O/pmc_summary.txt | cut -c1-140 | head -50
cp $O/r05_pmc_c2.json $O/r05_pmc_c4_fast.json profiles/ 2>/dev/null
for W in C2 C4-fast; do
  sh tools/frame_trace.sh $W > $O/frame_$W.log 2>&1; cp gpurun_out/frame_trace_$W/one_frame.txt $O/one_frame_$W.txt
done
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line.json; cat $O/bench_line.json
cp profiles/bench_full_r05.json $O/bench_full.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 2 --no-secondary > $O/bench_line_steps20.json 2>/dev/null; cat $O/bench_line_steps20.json | cut -c1-600
find $O -name "*.csv" -size +1M -delete
