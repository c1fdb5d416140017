# Final round-3 measurement pass (gpurun, repo root): touched-since-last-full-run tests, the default bench line, the
# merged kernel stats at HEAD.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r03
rm -rf $O; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_reduce_multiprocess_gpu.py tests/test_host_adapter_gpu.py tests/test_frame_source.py tests/test_parity_gpu.py -m gpu -q -k "reduce or adapter or bag_replay or pipelin or batched or benched" --durations=5 2>&1 | tail -12
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 300 $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/merged -o run -- python $R/bench.py $BENCH --method merged > $O/merged.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/merged "python bench.py $BENCH --method merged" $O/merged.log > $O/merged_kernel_stats.txt 2>&1)
find $O -name "*.csv" -size +2M -delete
cut -c1-70,100-150 $O/merged_kernel_stats.txt | head -30
