import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from kimera_semantics_amd import binding as B, synth, parallel as PAR
sc = synth.make_scene("room")
h = B.HipIntegrator(B.default_config(method=0, max_tiles=1 << 13, max_points=640*480, pipeline_frames=1,
                                     semantic_measurement_probability=0.8, dynamic_labels=[20], label_rgba=synth.default_label_colors()))
g = B.HipIntegrator(B.default_config(method=0, max_tiles=1 << 13, max_points=640*480, pipeline_frames=1,
                                     semantic_measurement_probability=0.8, dynamic_labels=[20], label_rgba=synth.default_label_colors()))
for k in range(0, 110, 4):
    f = synth.render_frame(sc, synth.trajectory_pose(k), 640, 480, seed=k)
    h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    f = synth.render_frame(sc, synth.trajectory_pose(k + 2), 640, 480, seed=k + 2)
    g.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
h.flush(); g.flush(); h.synchronize(); g.synchronize()
dev = torch.device("cuda", 0)
for rep in range(3):
    t0 = time.perf_counter(); keys = h.tile_keys(); t1 = time.perf_counter()
    own = PAR.owner_of(keys, 8); slots = np.nonzero(own != 0)[0].astype(np.uint32); t2 = time.perf_counter()
    buf = torch.empty((len(slots), PAR.TILE_WORDS), dtype=torch.int32, device=dev)
    h.export_tiles(slots, buf.data_ptr()); t3 = time.perf_counter()
    kk = keys[slots.astype(np.int64)]
    torch.cuda.synchronize(); t4 = time.perf_counter()
    g.merge_tiles(kk, buf.data_ptr()); t5 = time.perf_counter()
    print(f"tiles {len(keys)} sent {len(slots)} ({len(slots)*65536/1e6:.0f} MB): tile_keys {1e3*(t1-t0):.2f} ms, owners {1e3*(t2-t1):.2f}, "
          f"alloc+export {1e3*(t3-t2):.2f}, sync {1e3*(t4-t3):.2f}, merge {1e3*(t5-t4):.2f}  total {1e3*(t5-t0):.2f} ms")
