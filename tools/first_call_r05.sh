# First GPU call of the next round (gpurun, repo root:  bash tools/first_call_r05.sh): what round 4 could no longer measure.
#   1. the whole GPU tier + smoke at HEAD
#   2. the default bench line (the record now carries the fix-point rounds of every timed region)
#   3. the headline run queue by queue WITHOUT kernel tracing distorting it: tools/host_block.py (which calls wait, for how
#      long) on the bench trajectory's frames, then the traced view (tools/pipe_view.py) for the kernel durations
#   4. A/B of the launch strategies that change nothing but the clock, on the bench line (median of the regions):
#      KS_EXACT_BULK_ROUNDS 10 / 12 / 16, KS_EXACT_SEED_GROWTH 64, KS_XLONG 0
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/first_r05
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -12 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | grep -v amdgpu.ids | tail -6
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err; wc -c $O/bench_line.json; head -c 1500 $O/bench_line.json; echo
cp profiles/bench_full_r04.json $O/bench_full.json 2>/dev/null
python tools/host_block.py C2 8 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/host_block.txt
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary --no-oracle-count"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))['timing']
print(d['ms_per_step'], d['value'], f['ms_per_step_all_regions'], [r.get('rounds') for r in f.get('early_out_all_regions', [])])"; }
for v in KS_EXACT_BULK_ROUNDS=10 KS_EXACT_BULK_ROUNDS=12 KS_EXACT_BULK_ROUNDS=16 KS_EXACT_SEED_GROWTH=64 KS_XLONG=0; do
  run $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | tee -a $O/ab.txt
done
