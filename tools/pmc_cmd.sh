set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc8_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc8_$C -o run -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $R/gpurun_out/pmc8_$C.log 2>&1
  tail -1 $R/gpurun_out/pmc8_$C.log | cut -c1-200
done
ls $R/gpurun_out/pmc8_FETCH_SIZE
