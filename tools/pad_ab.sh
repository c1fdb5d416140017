cd $GRAFT_REPO_ROOT
O=gpurun_out/call_pad; rm -rf $O; mkdir -p $O
run() {  # tag, env...
  tag=$1; shift
  env KS_DEBUG=1 "$@" timeout 900 python bench.py --only-secondary C3 --no-cpu-baseline --no-oracle-count > $O/bench_$tag.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
  python - <<PY
import json
d = json.loads(open("gpurun_out/call_pad/bench_$tag.json").read())
print("$tag", d["value"], d["ms_per_step"], [(r["config"], r["ms_per_step"]) for r in d.get("secondary", [])])
PY
}
for P in 0 1 2 3 4 5; do run pad$P KS_BUNDLE_STREAM=1 KS_EMIT_ON_TAIL=0 KS_STREAM_PAD=$P; done
run inline KS_BUNDLE_STREAM=0 KS_EMIT_ON_TAIL=0
