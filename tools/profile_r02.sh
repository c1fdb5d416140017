# Round-2 profiles: default bench line, rocprofv3 kernel stats of the fast and the merged bench (gpurun, repo root).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
rm -rf $O; mkdir -p $O
timeout 600 python $R/bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
for M in fast merged; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$M -o run -- python $R/bench.py --steps 20 --warmup 3 --method $M --no-cpu-baseline --no-secondary --no-oracle-count > $O/$M.log 2>&1
  cd $R; python tools/summarize_rocprof.py $O/$M "python bench.py --steps 20 --warmup 3 --method $M --no-cpu-baseline --no-secondary --no-oracle-count" $O/$M.log > $O/${M}_kernel_stats.txt 2>&1; cd /tmp
done
head -c 1500 $O/bench_default.json; echo; cat $O/fast_kernel_stats.txt | cut -c1-80,100-150
