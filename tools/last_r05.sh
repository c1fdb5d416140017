# Round 5, last GPU call at HEAD: the whole GPU tier + smoke, the C4-fast timing + PMC passes (the sweeps changed after tools/final_r05.sh),
# the default bench line (and the driver's shape: --steps 20).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/last_r05
rm -rf $O; mkdir -p $O
cd $R
timeout 1300 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -12 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/smoke.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/time_C4-fast -o run -- python $R/tools/probe.py C4-fast 4 > $O/time_C4-fast.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_C4-fast_$C -o run -- python $R/tools/probe.py C4-fast 4 > $O/pmc_C4-fast_$C.log 2>&1
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_calib_$C -o run -- python $R/tools/probe.py calib 3 > $O/pmc_calib_$C.log 2>&1
done
cd $R
PMC_TAG=r05 PMC_SCRIPT=last_r05.sh python tools/pmc_r03.py $O | tee $O/pmc_summary.txt | cut -c1-140 | head -30
cp $O/r05_pmc_c4_fast.json profiles/ 2>/dev/null
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu.ids; wc -c $O/bench_line.json; cat $O/bench_line.json
cp profiles/bench_full_r05.json $O/bench_full.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 2 --no-secondary > $O/bench_line_steps20.json 2>/dev/null; cut -c1-500 $O/bench_line_steps20.json
cp profiles/bench_full_r05.json $O/bench_full_steps20.json 2>/dev/null
find $O -name "*.csv" -size +1M -delete
