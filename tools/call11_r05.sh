# Round 5, GPU call (gpurun, repo root:  bash tools/call11_r05.sh): result-neutral knobs of the C2 chain — FINER seeds
# (KS_EXACT_SEED_GROWTH 28 / 24 / 20: more k_test phases, fewer marks and dirty rays; the coarser ones lost in call 2) and a
# third march stream.  The fixed point is unique: none of these changes a map.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call11_r05
rm -rf $O; mkdir -p $O
cd $R
run() { steps=$1; shift; echo "== steps $steps $*"; env "$@" python bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))['timing']
print(d['ms_per_step'], d['value'], f['ms_per_step_all_regions'], [r.get('rounds') for r in f.get('early_out_all_regions', [])][:2])"; }
for v in KS_X=0 KS_EXACT_SEED_GROWTH=28 KS_EXACT_SEED_GROWTH=24 KS_EXACT_SEED_GROWTH=20 KS_MARCH_STREAMS=3; do
  run 40 $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | tee -a $O/c2_ab.txt
done
for v in KS_X=0 KS_EXACT_SEED_GROWTH=24 KS_EXACT_SEED_GROWTH=20; do
  run 20 $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | tee -a $O/c2_ab.txt
done
