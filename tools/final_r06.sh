# Round 6, final GPU call (gpurun, repo root:  bash tools/final_r06.sh).  Results under gpurun_out/final_r06; what is kept goes to profiles/r06_*.
#   1. the whole GPU tier (serially, as the driver runs it) + smoke at HEAD
#   2. PMC: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (with --kernel-trace only) of C2 / C3 (the bench command), C4-merged / C4-fast ordered
#      (tools/probe.py, unpipelined frames) and of the calibration kernel; SQ passes of C4-merged (tools/sq_pass.sh)
#   3. rocprofv3 --kernel-trace --stats of the headline command; the steady-state frame of C4-merged and C3 kernel by kernel
#   4. the bench line in the driver's shape and the default one (their roofline.traffic comes from step 2's passes)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r06
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/smoke.txt
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
PMC_TAG=r06 PMC_SCRIPT=final_r06.sh python tools/pmc_r03.py $O | tee $O/pmc_summary.txt | cut -c1-160 | head -100
cp $O/r06_pmc_*.json profiles/ 2>/dev/null
bash tools/sq_pass.sh C4-merged 3 sq_c4_merged_final > $O/sq.log 2>&1; cp gpurun_out/sq_c4_merged_final/summary.txt $O/sq_c4_merged.txt
sh tools/ring_trace.sh C4-merged gpurun_out/final_r06; sh tools/ring_trace.sh C3 gpurun_out/final_r06
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line_driver_shape.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line_driver_shape.json
cp profiles/bench_full_r06.json $O/bench_full_driver_shape.json 2>/dev/null
timeout 1200 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line.json; cat $O/bench_line.json
cp profiles/bench_full_r06.json $O/bench_full.json 2>/dev/null
find $O -name "*.csv" -size +1M -delete
cut -c1-150 $O/fast_kernel_stats.txt | head -40
