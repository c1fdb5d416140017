set -x
R=$GRAFT_REPO_ROOT
python $R/bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err
tail -1 $R/gpurun_out/bench_final.json | cut -c1-1500
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fast8 -o run -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_fast8.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_merged8 -o run -- python $R/bench.py --steps 20 --warmup 3 --method merged --no-cpu-baseline > $R/gpurun_out/prof_merged8.log 2>&1
ls $R/gpurun_out/prof_fast8 $R/gpurun_out/prof_merged8
