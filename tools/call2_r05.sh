# (HISTORIC: a script of the round-5 search for the C4 device loop, kept as the provenance of profiles/r05_c4_fast_device_path.txt.  The switches it sets — KS_EXACT_EPOCHS, KS_EXACT_DENSE — existed only in the commits of that search; HEAD has KS_EXACT_SWEEPS / KS_EXACT_SWEEP_ORDER, see tools/call8_r05.sh, tools/call10_r05.sh.)
# Round 5, second GPU call (gpurun, repo root:  bash tools/call2_r05.sh): the default `fast` mode at C4 geometry ON THE DEVICE
# (views + dense iterations), and the C2 chain's launch-shape knobs.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call2_r05
rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x --durations=5 2>&1 | tail -12 | tee $O/pytest_exact.txt
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_golden_ref.py -m gpu -q -x -k "benched or early_out or golden" 2>&1 | tail -4 | tee $O/pytest_parity_subset.txt
for v in "KS_EXACT_EPOCHS=3 KS_EXACT_DENSE=5" "KS_EXACT_EPOCHS=2 KS_EXACT_DENSE=4" "KS_EXACT_EPOCHS=2 KS_EXACT_DENSE=7" "KS_EXACT_EPOCHS=3 KS_EXACT_DENSE=3" "KS_EXACT_EPOCHS=1 KS_EXACT_DENSE=8"; do
  env $v timeout 300 python tools/c4_fast_ab.py 8 0 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/c4_fast_ab.txt
done
env KS_EXACT_EPOCHS=3 KS_EXACT_DENSE=5 timeout 300 python tools/c4_fast_ab.py 8 8 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/c4_fast_ab.txt
sh tools/frame_trace.sh C4-fast > $O/c4_frame.log 2>&1; cp gpurun_out/frame_trace_C4-fast/one_frame.txt $O/c4_fast_one_frame.txt
B="python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-oracle-count"
run() { echo "== $*"; env "$@" $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); f=json.load(open(d['full_record']))['timing']
print(d['ms_per_step'], d['value'], f['ms_per_step_all_regions'], [r.get('rounds') for r in f.get('early_out_all_regions', [])][:2])"; }
for v in KS_X=0 KS_EXACT_SEED_GROWTH=64 KS_EXACT_SEED_GROWTH=128 KS_EXACT_SEED_GROWTH=256 KS_EXACT_BULK_ROUNDS=6; do
  run $v KS_BENCH_FULL=/tmp/ab.json 2>&1 | tee -a $O/c2_ab.txt
done
tail -30 $O/c4_fast_one_frame.txt
