cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 200 python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count --method merged 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['timing']['ms_per_step_all_regions'])"; }
run A=1
run KS_MERGED_OWN_MARCH=1
run KS_BENCH_PIPE=2
run KS_BENCH_PIPE=8
run KS_TAIL_ON_MAIN=1
timeout 300 python -m pytest tests/test_parity_gpu.py -m gpu -q -x -k "pipelin and not soak" 2>&1 | tail -2
