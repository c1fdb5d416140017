# PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, with --kernel-trace only) + a timing pass, on the
# C4 workloads and on the calibration kernel.  Run through gpurun from the repo root; results under gpurun_out/pmc_r02.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r02
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# PMC_SKIP_CALIB=1: keep the calibration passes of an earlier run (they import torch: slow)
for W in $([ -n "$PMC_SKIP_CALIB" ] || echo calib) C4-fast C4-merged; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${W}_$C -o run -- python $R/tools/probe.py $W 3 > $O/${W}_$C.log 2>&1
    tail -2 $O/${W}_$C.log | cut -c1-200
  done
done
for W in C4-fast C4-merged; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${W}_time -o run -- python $R/tools/probe.py $W 3 > $O/${W}_time.log 2>&1
done
cd $R
[ -n "$PMC_SKIP_CALIB" ] || python tools/pmc_summarize.py $O | tee $O/summary.txt | cut -c1-250
