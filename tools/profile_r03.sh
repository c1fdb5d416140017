# Round-3 profiles at HEAD (gpurun, repo root:  sh tools/profile_r03.sh).  Results under gpurun_out/prof_r03; the
# summaries that are kept go to profiles/r03_* (copied by hand after looking at them).
#   1. the default bench line
#   2. rocprofv3 --kernel-trace --stats of the bench command (fast = the headline, merged)
#   3. PMC: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (with --kernel-trace only) of the same bench commands, of the C4
#      workloads (tools/probe.py) and of the calibration kernel (k_export_tiles: known bytes in / out)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
rm -rf $O; mkdir -p $O
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
timeout 900 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
for M in fast merged; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$M -o run -- python $R/bench.py $BENCH --method $M > $O/$M.log 2>&1
  (cd $R; python tools/summarize_rocprof.py $O/$M "python bench.py $BENCH --method $M" $O/$M.log > $O/${M}_kernel_stats.txt 2>&1)
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${M}_$C -o run -- python $R/bench.py $BENCH --method $M > $O/pmc_${M}_$C.log 2>&1
    tail -c 200 $O/pmc_${M}_$C.log
  done
done
for W in calib C4-fast C4-merged; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${W}_$C -o run -- python $R/tools/probe.py $W 3 > $O/pmc_${W}_$C.log 2>&1
    tail -2 $O/pmc_${W}_$C.log | cut -c1-200
  done
done
for W in C4-fast C4-merged; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/time_$W -o run -- python $R/tools/probe.py $W 3 > $O/time_$W.log 2>&1
done
cd $R
python tools/pmc_r03.py $O | tee $O/pmc_summary.txt | cut -c1-220
# keep the raw per-dispatch csv files out of the merge-back (64 MiB limit): only summaries travel
find $O -name "*.csv" -size +2M -delete
ls -la $O | head -40
