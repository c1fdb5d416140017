cd $GRAFT_REPO_ROOT
O=gpurun_out/call13; rm -rf $O; mkdir -p $O
for D in 4 6; do
KS_DEBUG=1 KS_LANES_DEPTH=$D sh tools/ring_trace.sh C4-merged $O; cp $O/last_frame_C4-merged.txt $O/last_frame_C4-merged_d$D.txt
echo == depth $D; grep "k_apply\|# frame" $O/last_frame_C4-merged_d$D.txt
KS_DEBUG=1 KS_LANES_DEPTH=$D timeout 900 python bench.py --only-secondary C4-merged --no-cpu-baseline --steps 20 > $O/bench_d$D.json 2> $O/bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/call13/bench_d$D.json").read())
print("depth $D", d["value"], d["ms_per_step"])
for r in d.get("secondary", []): print(r)
PY
done
timeout 2400 python -m pytest tests -m gpu -q -x -n 4 --durations=6 2>&1 | tail -12 > $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt
