# Round 6, evidence refresh at HEAD after the merged stage-A work (the tiers and the bench lines: tools/final3_r06.sh):
# rocprofv3 --kernel-trace --stats of both integrators, FETCH_SIZE / WRITE_SIZE in separate passes, the steady-state frames.   bash tools/final4_r06.sh
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final4_r06
rm -rf $O; mkdir -p $O
cd $R
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast -o run -- python $R/bench.py $BENCH > $O/fast.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/fast "python bench.py $BENCH" $O/fast.log > $O/fast_kernel_stats.txt 2>&1)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/merged -o run -- python $R/bench.py $BENCH --method merged > $O/merged.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/merged "python bench.py $BENCH --method merged" $O/merged.log > $O/merged_kernel_stats.txt 2>&1)
for W in C4-fast C4-merged; do
  KS_BENCH_GROWTH=32 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/time_$W -o run -- python $R/tools/probe.py $W 4 > $O/time_$W.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_fast_$C -o run -- python $R/bench.py $BENCH > $O/pmc_fast_$C.log 2>&1
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_merged_$C -o run -- python $R/bench.py $BENCH --method merged > $O/pmc_merged_$C.log 2>&1
  for W in C4-fast C4-merged; do
    KS_BENCH_GROWTH=32 timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${W}_$C -o run -- python $R/tools/probe.py $W 4 > $O/pmc_${W}_$C.log 2>&1
  done
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_calib_$C -o run -- python $R/tools/probe.py calib 3 > $O/pmc_calib_$C.log 2>&1
done
cd $R
PMC_TAG=r06 PMC_SCRIPT=final4_r06.sh python tools/pmc_r03.py $O | tee $O/pmc_summary.txt | cut -c1-160 | head -60
sh tools/ring_trace.sh C4-merged gpurun_out/final4_r06; sh tools/ring_trace.sh C3 gpurun_out/final4_r06
sh tools/pipe_trace.sh C3 gpurun_out/final4_r06 c3 KS_DEBUG=1; python tools/queue_busy.py $O/pt_c3/run_kernel_trace.csv 8 > $O/queue_busy_c3.txt
sh tools/pipe_trace.sh C4-merged gpurun_out/final4_r06 c4m KS_DEBUG=1; python tools/queue_busy.py $O/pt_c4m/run_kernel_trace.csv 8 > $O/queue_busy_c4_merged.txt
find $O -name "*.csv" -size +1M -delete
