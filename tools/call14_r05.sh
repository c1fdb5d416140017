# (HISTORIC: KS_EXACT_SEED_LONGEST existed only in the build this script measured; no cap helped and the knob was removed.)
# Round 5, GPU call (gpurun, repo root:  bash tools/call14_r05.sh): the seed's schedule with a longest phase (KS_EXACT_SEED_LONGEST
# generations): geometric at first, then equal steps — the late phases are where a ray and the row above it share a phase.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call14_r05
rm -rf $O; mkdir -p $O
cd $R
run() { steps=$1; shift; echo "== steps $steps $*"; env "$@" python bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))['timing']
print(d['ms_per_step'], d['value'], f['ms_per_step_all_regions'], [(r.get('rounds'), r.get('fallbacks')) for r in f.get('early_out_all_regions', [])][:2])"; }
for v in "KS_X=0" "KS_EXACT_SEED_GROWTH=32 KS_EXACT_SEED_LONGEST=64" "KS_EXACT_SEED_GROWTH=32 KS_EXACT_SEED_LONGEST=128" "KS_EXACT_SEED_GROWTH=24 KS_EXACT_SEED_LONGEST=96" "KS_EXACT_SEED_GROWTH=22 KS_EXACT_SEED_LONGEST=128" "KS_EXACT_SEED_GROWTH=22 KS_EXACT_SEED_LONGEST=64"; do
  run 40 $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | grep -v "^+\|^import\|^d=\|^print" | tee -a $O/c2_ab.txt
  run 20 $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | grep -v "^+\|^import\|^d=\|^print" | tee -a $O/c2_ab.txt
done
cd /tmp && export TMPDIR=/tmp
BENCH="--steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast -o run -- python $R/bench.py $BENCH > $O/fast.log 2>&1
(cd $R; python tools/summarize_rocprof.py $O/fast "python bench.py $BENCH" $O/fast.log > $O/fast_kernel_stats.txt 2>&1)
find $O -name "*.csv" -size +1M -delete
cut -c1-150 $O/fast_kernel_stats.txt | head -40
