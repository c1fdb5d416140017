# Round 6: does the runtime's hardware-queue count (GPU_MAX_HW_QUEUES, default 4) decide the side-stream experiments?   bash tools/hwq_ab.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/call_hwq; rm -rf $O; mkdir -p $O
run() {  # tag, env...
  tag=$1; shift
  env KS_DEBUG=1 "$@" timeout 900 python bench.py --only-secondary C3,C4-merged --no-cpu-baseline --no-oracle-count > $O/bench_$tag.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
  python - <<PY
import json
d = json.loads(open("gpurun_out/call_hwq/bench_$tag.json").read())
print("$tag", d["value"], d["ms_per_step"], [(r["config"], r["ms_per_step"]) for r in d.get("secondary", [])])
PY
}
run q4_bs0 KS_BUNDLE_STREAM=0
run q8_bs0 GPU_MAX_HW_QUEUES=8 KS_BUNDLE_STREAM=0
run q8_bs1 GPU_MAX_HW_QUEUES=8 KS_BUNDLE_STREAM=1
run q8_bs1_as1 GPU_MAX_HW_QUEUES=8 KS_BUNDLE_STREAM=1 KS_APPLY_STREAM=1
run q12_bs1 GPU_MAX_HW_QUEUES=12 KS_BUNDLE_STREAM=1
run q2_bs0 GPU_MAX_HW_QUEUES=2 KS_BUNDLE_STREAM=0
