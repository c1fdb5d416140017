# Round 4, first GPU call (repo root on the GPU box):  gpurun --timeout 1200 -- 'bash tools/call1_r04.sh'
#   1. parity tests of the early-out paths            2. the default bench line (short line + full record)
#   3. rocprofv3 kernel stats of the headline command at HEAD
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_exact_early_out_gpu.py -m gpu -q -x -k "early_out or ordered_phases or benched or tag_wrap or pipeline" --durations=5 2>&1 | tail -12
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 300 $O/bench.err
wc -c $O/bench_line.json; head -c 4200 $O/bench_line.json
cp profiles/bench_full_r04.json $O/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast -o run -- python $R/bench.py $BENCH > $O/fast.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/fast "python bench.py $BENCH" $O/fast.log > $O/fast_kernel_stats.txt 2>&1)
find $O -name "*.csv" -size +2M -delete
cut -c1-150 $O/fast_kernel_stats.txt | head -30
