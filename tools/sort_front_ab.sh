cd $GRAFT_REPO_ROOT
O=gpurun_out/call_sf; rm -rf $O; mkdir -p $O
for SF in 1 0; do
KS_DEBUG=1 KS_SORT_FRONT=$SF timeout 900 python bench.py --only-secondary C4-merged,C3,C4-fast-ordered-phases --no-cpu-baseline --no-oracle-count > $O/bench_sf$SF.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
python - <<PY
import json
d = json.loads(open("gpurun_out/call_sf/bench_sf$SF.json").read())
print("sort front $SF", d["value"], d["ms_per_step"])
for r in d.get("secondary", []): print("   ", r)
PY
done
KS_DEBUG=1 KS_SORT_FRONT=1 timeout 1200 python -m pytest tests -m gpu -q -x -n 4 -k "merged or pipelin or frames_in_flight or no_early_out or full_size or lane_per_run or sensor" 2>&1 | tail -4
