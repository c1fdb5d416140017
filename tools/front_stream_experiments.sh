# Round 6: the A/B runs behind profiles/r06_front_stream_experiments.txt, one section each.  On the GPU box, repo root:
#   bash tools/front_stream_experiments.sh <section>      sections: bundle_stream | pad | hwq | apply_pad | prio | front | steady | keys | hint
# Every switch is read under KS_DEBUG=1 only and none of them changes a result (the parity tests run with them where noted).
cd $GRAFT_REPO_ROOT
S=${1:-steady}
O=gpurun_out/front_$S; rm -rf $O; mkdir -p $O
bench_rec() {  # tag, env... : C3 and C4-merged records of bench.py (regions of 40 / 12 frames)
  tag=$1; shift
  env KS_DEBUG=1 "$@" timeout 900 python bench.py --only-secondary C3,C4-merged --no-cpu-baseline --no-oracle-count > $O/bench_$tag.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
  python - <<PY
import json
d = json.loads(open("$O/bench_$tag.json").read())
print("$tag", d["value"], d["ms_per_step"], [(r["config"], r["ms_per_step"]) for r in d.get("secondary", [])])
PY
}
steady() {  # env... : steady state (one region of 400 / 72 frames)
  echo "== $*"
  env KS_DEBUG=1 "$@" timeout 300 python tools/steady_probe.py C3 400 2>&1 | grep -v amdgpu | tail -1
  env KS_DEBUG=1 "$@" timeout 300 python tools/steady_probe.py C4-merged 72 2>&1 | grep -v amdgpu | tail -1
}
case $S in
bundle_stream)   # k_bundles_long on a stream of its own beside k_bo_* (1) vs in line (0)
  for BS in 1 0 1 0; do bench_rec bs$BS KS_BUNDLE_STREAM=$BS; done
  timeout 1500 python -m pytest tests -m gpu -q -x -n 4 -k "merged or bundle" 2>&1 | tail -4 ;;
pad)             # ... with never-used streams created in front of it: which hardware queue / pipe it lands on
  for P in 0 1 2 3 4 5; do bench_rec pad$P KS_BUNDLE_STREAM=1 KS_EMIT_ON_TAIL=0 KS_STREAM_PAD=$P; done
  bench_rec inline KS_BUNDLE_STREAM=0 KS_EMIT_ON_TAIL=0 ;;
hwq)             # the runtime's hardware-queue count (bench.py's default is 8)
  bench_rec q8_bs0 KS_BUNDLE_STREAM=0; bench_rec q12_bs1 GPU_MAX_HW_QUEUES=12 KS_BUNDLE_STREAM=1; bench_rec q2_bs0 GPU_MAX_HW_QUEUES=2 KS_BUNDLE_STREAM=0 ;;
apply_pad)       # k_apply_runs on its own stream, with the padding
  steady KS_APPLY_STREAM=0; for P in 0 1 2 3; do steady KS_APPLY_STREAM=1 KS_STREAM_PAD=$P; done ;;
prio)            # stream priorities
  for E in "KS_TAIL_PRIO=0" "KS_TAIL_PRIO=1" "KS_TAIL_PRIO=2" "KS_FRONT_PRIO=1" "KS_FRONT_PRIO=1 KS_TAIL_PRIO=1" "KS_FRONT_PRIO=2"; do
    steady $E; env KS_DEBUG=1 $E timeout 300 python tools/steady_probe.py C2 400 2>&1 | grep -v amdgpu | tail -1
  done ;;
front)           # the long bundles on the long-run stream (2), stage B on the tail stream (1)
  bench_rec bs0_et0 KS_BUNDLE_STREAM=0 KS_EMIT_ON_TAIL=0; bench_rec bs2_et0 KS_BUNDLE_STREAM=2 KS_EMIT_ON_TAIL=0
  bench_rec bs0_et1 KS_BUNDLE_STREAM=0 KS_EMIT_ON_TAIL=1; bench_rec bs2_et1 KS_BUNDLE_STREAM=2 KS_EMIT_ON_TAIL=1
  timeout 1500 python -m pytest tests -m gpu -q -x -n 4 -k "merged or bundle or pipelin" 2>&1 | tail -4 ;;
steady)
  steady KS_NONE=1; steady KS_BUNDLE_STREAM=1 KS_STREAM_PAD=2; steady KS_BUNDLE_STREAM=2 ;;
keys)            # 32-bit grouping keys (default) vs the sorted 64-bit end-voxel keys
  timeout 900 python -m pytest tests/test_merged_keys_gpu.py -m gpu -q -x 2>&1 | tail -3
  steady KS_KEY_WINDOW_BITS=0; steady KS_NONE=1 ;;
hint)            # the bundle order's epochs for the bundles of the frames before (default) / for n points (0) / all through k_bo_rest (1)
  steady KS_NONE=1; steady KS_BO_HINT=0; steady KS_BO_HINT=1 ;;
esac
