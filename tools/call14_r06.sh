cd $GRAFT_REPO_ROOT
O=gpurun_out/call14; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --only-secondary C4-merged,C3,C4-fast-ordered-phases --no-cpu-baseline > $O/bench_on.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
KS_DEBUG=1 KS_APPLY_STREAM=0 timeout 900 python bench.py --only-secondary C4-merged,C3,C4-fast-ordered-phases --no-cpu-baseline > $O/bench_off.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
python - <<'PY'
import json
for t in ("on", "off"):
    d = json.loads(open(f"gpurun_out/call14/bench_{t}.json").read())
    print("apply stream", t, d["value"], d["ms_per_step"])
    for r in d.get("secondary", []): print("   ", r)
PY
timeout 2400 python -m pytest tests -m gpu -q -x -n 4 -k "pipelin or benched or full_size or frames_in_flight or batched or c4_geometry or lane_per_run or sensor" 2>&1 | tail -6 > $O/pytest_gpu.txt; tail -3 $O/pytest_gpu.txt
