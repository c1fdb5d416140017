# Round 6: the long-ray mark buffers allocated by the first integrate call instead of ks_create.   bash tools/lazy_marks_r06.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/lazy_marks; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_exact_early_out_gpu.py tests/test_c4_fast_device_gpu.py -m gpu -q -x 2>&1 | tail -4
timeout 900 python bench.py --only-secondary C4-fast --no-cpu-baseline --no-oracle-count > $O/bench_c4fast.json 2> $O/bench.err; tail -c 300 $O/bench.err | grep -v amdgpu
python - <<PY
import json
d = json.loads(open("gpurun_out/lazy_marks/bench_c4fast.json").read())
print(d["value"], d["ms_per_step"])
for r in d.get("secondary", []): print("   ", r)
PY
python - <<PY
import torch, time
from kimera_semantics_amd import binding as B
import inspect
free0 = torch.cuda.mem_get_info()[0]
print("free before create", free0 / 2**30)
PY
