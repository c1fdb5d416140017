R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_merged8
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_merged8 -o run -- python $R/bench.py --steps 20 --warmup 3 --method merged --no-cpu-baseline > $R/gpurun_out/prof_merged8.log 2>&1
ls $R/gpurun_out/prof_merged8 | head -3
