# First gpurun call of the next round (repo root on the GPU box): everything the last CPU-only stretch of round 3 left unmeasured, in
# ONE call:  /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/first_call_r04.sh'
#   1. the parity tests of the kernels that changed (k_test: live sub-runs, overlapped rounds; both switches back)
#   2. the default bench line with its *-switches A/B records  -> gpurun_out/first_r04/bench_default.json, switch table
#   3. rocprofv3 kernel stats of the headline command at HEAD  -> gpurun_out/first_r04/fast_kernel_stats.txt
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/first_r04
rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_exact_early_out_gpu.py -m gpu -q -x -k "early_out or ordered_phases or sub_runs or benched or golden" --durations=5 2>&1 | tail -12
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
python tools/switch_report.py $O/bench_default.json | tee $O/switches.txt
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast -o run -- python $R/bench.py $BENCH > $O/fast.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/fast "python bench.py $BENCH" $O/fast.log > $O/fast_kernel_stats.txt 2>&1)
find $O -name "*.csv" -size +2M -delete
cut -c1-70,100-150 $O/fast_kernel_stats.txt | head -30
