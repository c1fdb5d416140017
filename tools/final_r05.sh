# Round 5, final GPU call (gpurun, repo root:  bash tools/final_r05.sh).  Results under gpurun_out/final_r05; what is kept goes to
# profiles/r05_* (copied after looking at them).
#   1. the whole GPU tier + smoke at HEAD
#   2. PMC: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (with --kernel-trace only) of C2 / C3 (the bench command), C4-fast /
#      C4-merged (tools/probe.py, unpipelined frames) and of the calibration kernel (k_export_tiles: known bytes in / out)
#   3. rocprofv3 --kernel-trace --stats of the headline command; one unpipelined frame of every workload, kernel by kernel
#   4. the default bench line (its roofline.traffic comes from step 2's passes)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r05
rm -rf $O; mkdir -p $O
cd $R
timeout 1300 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/smoke.txt
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast -o run -- python $R/bench.py $BENCH > $O/fast.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/fast "python bench.py $BENCH" $O/fast.log > $O/fast_kernel_stats.txt 2>&1; python tools/pipe_view.py $(ls $O/fast/*kernel_trace.csv $O/fast/*/*kernel_trace.csv 2>/dev/null | head -1) 3000 > $O/c2_pipelined_queue_view.txt 2>&1)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/merged -o run -- python $R/bench.py $BENCH --method merged > $O/merged.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/merged "python bench.py $BENCH --method merged" $O/merged.log > $O/merged_kernel_stats.txt 2>&1)
for W in C4-fast C4-merged; do
  timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/time_$W -o run -- python $R/tools/probe.py $W 4 > $O/time_$W.log 2>&1
done
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_fast_$C -o run -- python $R/bench.py $BENCH > $O/pmc_fast_$C.log 2>&1
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_merged_$C -o run -- python $R/bench.py $BENCH --method merged > $O/pmc_merged_$C.log 2>&1
  for W in C4-fast C4-merged; do
    timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${W}_$C -o run -- python $R/tools/probe.py $W 4 > $O/pmc_${W}_$C.log 2>&1
  done
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_calib_$C -o run -- python $R/tools/probe.py calib 3 > $O/pmc_calib_$C.log 2>&1
done
cd $R
PMC_TAG=r05 PMC_SCRIPT=final_r05.sh python tools/pmc_r03.py $O | tee $O/pmc_summary.txt | cut -c1-160 | head -90
cp $O/r05_pmc_*.json profiles/ 2>/dev/null
for W in C2 C3 C4-fast C4-merged; do
  sh tools/frame_trace.sh $W > $O/frame_$W.log 2>&1; cp gpurun_out/frame_trace_$W/one_frame.txt $O/one_frame_$W.txt
done
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line.json; cat $O/bench_line.json
cp profiles/bench_full_r05.json $O/bench_full.json 2>/dev/null
find $O -name "*.csv" -size +1M -delete
cut -c1-150 $O/fast_kernel_stats.txt | head -45
