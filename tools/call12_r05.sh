# Round 5, GPU call (gpurun, repo root:  bash tools/call12_r05.sh): the seed's phase growth between 18 and 26 at both ring sizes
# (call 11: 32 -> 0.575, 28 -> 0.559, 24 -> 0.541, 20 -> 0.525 ms/frame at K = 40; 0.487 / 0.461 (24) / 0.484 (20) at K = 20).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call12_r05
rm -rf $O; mkdir -p $O
cd $R
run() { steps=$1; shift; echo "== steps $steps $*"; env "$@" python bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))['timing']
print(d['ms_per_step'], d['value'], f['ms_per_step_all_regions'], [(r.get('rounds'), r.get('fallbacks')) for r in f.get('early_out_all_regions', [])][:2])"; }
for v in KS_EXACT_SEED_GROWTH=18 KS_EXACT_SEED_GROWTH=22 KS_EXACT_SEED_GROWTH=26 "KS_EXACT_SEED_GROWTH=20 KS_EXACT_BULK_ROUNDS=14" "KS_EXACT_SEED_GROWTH=24 KS_EXACT_BULK_ROUNDS=14"; do
  run 40 $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | grep -v "^+" | tee -a $O/c2_ab.txt
  run 20 $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | grep -v "^+" | tee -a $O/c2_ab.txt
done
