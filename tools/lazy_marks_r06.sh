# Round 6: the long-ray mark buffers allocated by the first integrate call instead of ks_create.   bash tools/lazy_marks_r06.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/lazy_marks; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_exact_early_out_gpu.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python bench.py --only-secondary C4-fast --no-cpu-baseline --no-oracle-count > $O/bench_c4fast.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
python - <<PY
import json
d = json.loads(open("gpurun_out/lazy_marks/bench_c4fast.json").read())
print(d["value"], d["ms_per_step"])
for r in d.get("secondary", []): print("   ", r)
PY
python - <<PY
import torch, numpy as np
from kimera_semantics_amd import binding as B
g = lambda: torch.cuda.mem_get_info()[0] / 2**30
torch.zeros(1, device="cuda"); f0 = g()
h = B.HipIntegrator(B.default_config(device_id=0, max_tiles=1 << 13, max_points=1280 * 720, voxel_size=0.02, max_ray_length_m=10.0))
f1 = g()
n = 1280 * 720
rng = np.random.default_rng(0)
xyz = rng.random((n, 3), dtype=np.float32) * np.float32(4.0) - np.float32(2.0); xyz[:, 2] = 3.0; rgba = np.zeros((n, 4), np.uint8); lab = np.zeros(n, np.uint8)
h.integrate(np.eye(4, dtype=np.float32), xyz, rgba, lab)
f2 = g()
print("free GiB: before create %.2f, after create %.2f (create holds %.2f), after the first frame %.2f (the frame added %.2f)" % (f0, f1, f0 - f1, f2, f1 - f2))
h.close()
PY
