# Round 5, ninth GPU call: why does the adapter (C++ process, 80 host clouds, on-demand) integrate the same 40-frame ring at 0.32 ms/frame
# when bench.py's regions (40 frames, flush, synchronize) take 0.68?
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call9_r05
rm -rf $O; mkdir -p $O
cd $R
B="python bench.py --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
run() { echo "== $*"; env "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))
print(d['ms_per_step'], d['value'], f['timing']['ms_per_step_all_regions'], [r.get('rounds') for r in f['timing'].get('early_out_all_regions', [])][:2], f['host_ms_per_frame'])"; }
run KS_BENCH_FULL=/tmp/ab.json $B --steps 40 2>&1 | tee -a $O/ab.txt
run KS_BENCH_FULL=/tmp/ab.json KS_EXACT_BULK_ROUNDS=14 $B --steps 40 2>&1 | tee -a $O/ab.txt
run KS_BENCH_FULL=/tmp/ab.json $B --steps 120 2>&1 | tee -a $O/ab.txt
run KS_BENCH_FULL=/tmp/ab.json KS_EXACT_BULK_ROUNDS=14 $B --steps 120 2>&1 | tee -a $O/ab.txt
python tools/host_block.py C2 8 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-900 | tee $O/host_block.txt
