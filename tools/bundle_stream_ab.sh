# Round 6: k_bundles_long on a stream of its own beside the bundle order (default) vs in line (KS_BUNDLE_STREAM=0).   bash tools/bundle_stream_ab.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/call_bs; rm -rf $O; mkdir -p $O
for BS in 1 0 1 0; do
KS_DEBUG=1 KS_BUNDLE_STREAM=$BS timeout 900 python bench.py --only-secondary C3,C4-merged --no-cpu-baseline --no-oracle-count > $O/bench_bs$BS.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
python - <<PY
import json
d = json.loads(open("gpurun_out/call_bs/bench_bs$BS.json").read())
print("bundle stream $BS", d["value"], d["ms_per_step"])
for r in d.get("secondary", []): print("   ", r)
PY
done
timeout 1500 python -m pytest tests -m gpu -q -x -n 4 -k "merged or bundle" 2>&1 | tail -4
