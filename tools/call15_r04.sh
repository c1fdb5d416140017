# Round 4: frames per stage-B batch / lag of the pipeline (headline workload, default mode)
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-oracle-count"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['value'], json.load(open(d['full_record']))['timing']['ms_per_step_all_regions'])"; }
run KS_BENCH_PIPE=8 KS_BENCH_FULL=/tmp/f1.json
run KS_BENCH_PIPE=16 KS_BENCH_FULL=/tmp/f2.json
run KS_BENCH_PIPE=16 KS_BATCH=4 KS_MARCH_STREAMS=3 KS_BENCH_FULL=/tmp/f3.json
run KS_BENCH_PIPE=16 KS_BATCH=8 KS_MARCH_STREAMS=3 KS_BENCH_FULL=/tmp/f4.json
run KS_BENCH_PIPE=12 KS_BATCH=4 KS_MARCH_STREAMS=2 KS_BENCH_FULL=/tmp/f5.json
