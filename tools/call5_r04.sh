# Round 4: the whole GPU tier + the default bench line at HEAD
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04e
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=10 2>&1 | tail -40 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench.err
tail -c 400 $O/bench.err
wc -c $O/bench_line.json; cat $O/bench_line.json
cp profiles/bench_full_r04.json $O/ 2>/dev/null
