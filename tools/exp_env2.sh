cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 120 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_all_regions'])"; }
run KS_BENCH_PIPE=4
run KS_BENCH_PIPE=4
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "early_out or benched" 2>&1 | tail -3
