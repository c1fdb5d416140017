# Round 5, GPU call (gpurun, repo root:  bash tools/call18_r05.sh): the regions of the driver's shape alternate between 0.463 and 0.49
# ms/frame — five batches per region, so the two march streams get 3 + 2 and 2 + 3 of them in turn: is one of the two sharing a hardware
# queue?  GPU_MAX_HW_QUEUES 12 / 16 against the 8 bench.py sets.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call18_r05
rm -rf $O; mkdir -p $O
cd $R
run() { steps=$1; shift; echo "== steps $steps $*"; env "$@" timeout 60 python bench.py --steps $steps --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))['timing']
print(d['ms_per_step'], d['value'], f['ms_per_step_all_regions'])"; }
run 20 GPU_MAX_HW_QUEUES=12 KS_BENCH_FULL=/tmp/ab.json 2>&1 | tee -a $O/c2_ab.txt
run 40 GPU_MAX_HW_QUEUES=12 KS_BENCH_FULL=/tmp/ab.json 2>&1 | tee -a $O/c2_ab.txt
run 20 GPU_MAX_HW_QUEUES=16 KS_BENCH_FULL=/tmp/ab.json 2>&1 | tee -a $O/c2_ab.txt
