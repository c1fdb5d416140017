cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 120 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_all_regions'])"; }
run KS_BENCH_PIPE=4
run KS_BENCH_PIPE=8
run KS_BENCH_PIPE=4 KS_BATCH=1
run KS_BENCH_PIPE=4 GPU_MAX_HW_QUEUES=5
run KS_BENCH_PIPE=8 GPU_MAX_HW_QUEUES=5
