# Round 5, GPU call (gpurun, repo root:  bash tools/call16_r05.sh): pipeline_frames = 16 — batches of EIGHT frames per launch sequence
# (24 frame slots, 16 early-out tables) against the default 8 (batches of four), both ring sizes; the tests of the new sizes first.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call16_r05
rm -rf $O; mkdir -p $O
cd $R
timeout 500 python -m pytest tests/test_parity_gpu.py tests/test_exact_early_out_gpu.py -m gpu -q -x -k "batched_stage_b or tag_wrap or early_out_pipelined" 2>&1 | tail -4 | tee $O/pytest_batch8.txt
run() { steps=$1; shift; echo "== steps $steps $*"; env "$@" python bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))['timing']
print(d['ms_per_step'], d['value'], f['ms_per_step_all_regions'], [(r.get('rounds'), r.get('fallbacks')) for r in f.get('early_out_all_regions', [])][:2])"; }
for v in "KS_BENCH_PIPE=16" "KS_BENCH_PIPE=8" "KS_BENCH_PIPE=16 KS_BATCH=6" "KS_BENCH_PIPE=12 KS_BATCH=6"; do
  run 40 $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | grep -v "^+\|^import\|^d=\|^print" | tee -a $O/c2_ab.txt
  run 20 $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | grep -v "^+\|^import\|^d=\|^print" | tee -a $O/c2_ab.txt
done
