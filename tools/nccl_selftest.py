#!/usr/bin/env python3
"""Exercises the N>1 code path of bench.py on ONE GPU: a world_size-1 RCCL process group,
parallel.warm_up and parallel.reduce_maps over a HipTileStore (API / dtype / stream handling;
the multi-rank protocol itself is covered by the gloo tests)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from kimera_semantics_amd import binding as B  # noqa: E402
from kimera_semantics_amd import parallel as PAR  # noqa: E402
from kimera_semantics_amd import synth  # noqa: E402

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
h = B.HipIntegrator(B.default_config(method=0, max_tiles=4096, max_points=1 << 16, pipeline_frames=1,
                                     semantic_measurement_probability=0.8, label_rgba=synth.default_label_colors()))
sc = synth.make_scene("room")
for k in range(3):
    f = synth.render_frame(sc, synth.trajectory_pose(k), 160, 120, seed=k)
    h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
h.flush()
print("warm_up ->", PAR.warm_up(dev))
dist.barrier()
print("reduce_maps ->", PAR.reduce_maps(PAR.HipTileStore(h, dev)))
t = torch.tensor([1.0], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
print("all_reduce ok", float(t.item()), "tiles", len(h.tile_keys()))
h.close()
dist.destroy_process_group()
print("nccl selftest OK")
