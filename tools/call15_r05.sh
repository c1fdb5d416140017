# Round 5, GPU call (gpurun, repo root:  bash tools/call15_r05.sh): how much do the two march streams overlap?  One stream (no overlap
# between batches) against two (default) and four; batches of two.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call15_r05
rm -rf $O; mkdir -p $O
cd $R
run() { steps=$1; shift; echo "== steps $steps $*"; env "$@" python bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))['timing']
print(d['ms_per_step'], d['value'], f['ms_per_step_all_regions'], [(r.get('rounds'), r.get('fallbacks')) for r in f.get('early_out_all_regions', [])][:2])"; }
for v in "KS_MARCH_STREAMS=1" "KS_MARCH_STREAMS=4" "KS_BATCH=2" "KS_BATCH=3" "GPU_MAX_HW_QUEUES=4"; do
  run 40 $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | grep -v "^+\|^import\|^d=\|^print" | tee -a $O/c2_ab.txt
done
