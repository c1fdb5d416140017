# Round 6: what leaves the front stream of `merged` — k_bundles_long to the long-run stream, stage B to the tail stream.   bash tools/front_ab.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/call_front; rm -rf $O; mkdir -p $O
run() {  # tag, env...
  tag=$1; shift
  env KS_DEBUG=1 "$@" timeout 900 python bench.py --only-secondary C3,C4-merged --no-cpu-baseline --no-oracle-count > $O/bench_$tag.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
  python - <<PY
import json
d = json.loads(open("gpurun_out/call_front/bench_$tag.json").read())
print("$tag", d["value"], d["ms_per_step"], [(r["config"], r["ms_per_step"]) for r in d.get("secondary", [])])
PY
}
run bs0_et0 KS_BUNDLE_STREAM=0 KS_EMIT_ON_TAIL=0
run bs2_et0 KS_BUNDLE_STREAM=2 KS_EMIT_ON_TAIL=0
run bs0_etd KS_BUNDLE_STREAM=0
run bs2_etd KS_BUNDLE_STREAM=2
run bs2_et1 KS_BUNDLE_STREAM=2 KS_EMIT_ON_TAIL=1
run default
timeout 1500 python -m pytest tests -m gpu -q -x -n 4 -k "merged or bundle or pipelin" 2>&1 | tail -4
