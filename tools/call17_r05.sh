# Round 5, GPU call (gpurun, repo root:  bash tools/call17_r05.sh): `merged` (C3) with pipeline_frames 8 against 16 (batches of eight).
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call17_r05
rm -rf $O; mkdir -p $O
cd $R
run() { echo "== $*"; env "$@" timeout 100 python bench.py --method merged --steps 40 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))['timing']
print(d['ms_per_step'], d['value'], f['ms_per_step_all_regions'])"; }
for v in "KS_BENCH_PIPE=16" "KS_BENCH_PIPE=8"; do
  run $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | grep -v "^+\|^import\|^d=\|^print" | tee -a $O/c3_ab.txt
done
