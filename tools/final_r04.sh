# Round-4 profiles at HEAD, most important first (gpurun, repo root:  bash tools/final_r04.sh).  Results under
# gpurun_out/prof_r04; the summaries that are kept go to profiles/r04_* (copied by hand after looking at them).
#   1. the default bench line (short stdout line + profiles/bench_full_r04.json)
#   2. rocprofv3 --kernel-trace --stats of the headline command (fast, default mode = exact serial early-out)
#   3. one unpipelined C3 and C4-merged frame, kernel by kernel
#   4. PMC: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (with --kernel-trace only) of the same command and of the
#      calibration kernel (k_export_tiles: known bytes in / out)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r04
rm -rf $O; mkdir -p $O
cd $R
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 400 $O/bench.err | grep -v amdgpu.ids
wc -c $O/bench_line.json; cat $O/bench_line.json
cp profiles/bench_full_r04.json $O/ 2>/dev/null
timeout 120 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "runs_next_to_the_sensor or long_bundles_edge or close_up_long_runs or merged_single_frame" 2>&1 | tail -2 | tee $O/pytest_merged_chains.txt
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast -o run -- python $R/bench.py $BENCH > $O/fast.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/fast "python bench.py $BENCH" $O/fast.log > $O/fast_kernel_stats.txt 2>&1)
cd $R
sh tools/frame_trace.sh C3 > $O/c3_frame.log 2>&1; cp gpurun_out/frame_trace_C3/one_frame.txt $O/c3_one_frame.txt
sh tools/frame_trace.sh C4-merged > $O/c4m_frame.log 2>&1; cp gpurun_out/frame_trace_C4-merged/one_frame.txt $O/c4_merged_one_frame.txt
grep "apply\|bundles" $O/c3_one_frame.txt $O/c4_merged_one_frame.txt
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_fast_$C -o run -- python $R/bench.py $BENCH > $O/pmc_fast_$C.log 2>&1
  timeout 90 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_calib_$C -o run -- python $R/tools/probe.py calib 3 > $O/pmc_calib_$C.log 2>&1
done
cd $R
PMC_TAG=r04 python tools/pmc_r03.py $O | tee $O/pmc_summary.txt | cut -c1-200
find $O -name "*.csv" -size +2M -delete
cut -c1-150 $O/fast_kernel_stats.txt | head -45
