"""One rank of tests/test_reduce_multiprocess_gpu.py: integrates its share of two batches of overlapping frames,
calls ks_reduce (C ABI) after each batch through the communicator library named by KS_RCCL_LIB (the test double
tests/mock_rccl: several ranks share the one GPU of a development box), and writes the tiles it then holds."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def frames_of(rank, world, batch, n_per_rank=2):
    from kimera_semantics_amd import synth
    sc = synth.make_scene("room")
    n = world * n_per_rank
    out = []
    for j in range(n_per_rank):
        k = rank + world * j
        out.append(synth.render_frame(sc, synth.arc_pose(k, n, spacing=0.3 + 0.1 * batch), 160, 120, seed=500 + 10 * batch + k))
    return out


def config_kw():
    from kimera_semantics_amd import synth
    return dict(semantic_measurement_probability=0.8, dynamic_labels=[20], label_rgba=synth.default_label_colors(),
                method=1, voxels_per_side=8)


def export_all(h, torch=None):
    """Every resident tile as (keys, [n, 512, 32] u32 records).  Device staging through the HIP runtime directly
    (ctypes), so that a worker process does not have to import torch."""
    keys = h.tile_keys()
    n = len(keys)
    rec = np.zeros((n, 512, 32), dtype=np.uint32)
    if n and os.environ.get("KS_HIP_LIB", "").endswith("libks_hip_emu.so"):
        # the host functional model (tools/emu): "device" memory is host memory
        h.export_tiles(np.arange(n, dtype=np.uint32), rec.ctypes.data)
    elif n:
        hip = C.CDLL("libamdhip64.so")
        d = C.c_void_p()
        assert hip.hipMalloc(C.byref(d), C.c_size_t(n * 65536)) == 0
        h.export_tiles(np.arange(n, dtype=np.uint32), d.value)
        assert hip.hipMemcpy(C.c_void_p(rec.ctypes.data), d, C.c_size_t(n * 65536), C.c_int(2)) == 0   # hipMemcpyDeviceToHost
        hip.hipFree(d)
    return keys, rec


def round_config_kw():
    from kimera_semantics_amd import synth
    return dict(semantic_measurement_probability=0.8, dynamic_labels=[20], label_rgba=synth.default_label_colors(),
                method=0, voxels_per_side=8)   # `fast`, the reference's default early-out: the library's default mode


def round_frames(n_frames, w=160, h=120):
    """The frames of tests/test_reduce_multiprocess_gpu.py::test_exact_round_*: overlapping views of one wall, by GLOBAL frame number."""
    from kimera_semantics_amd import synth
    if os.environ.get("KS_ROUND_WH"):   # (the CPU tier runs the same test on the functional model, on small frames)
        w, h = (int(x) for x in os.environ["KS_ROUND_WH"].split("x"))
    sc = synth.make_scene("room")
    return [synth.render_frame(sc, synth.arc_pose(k, n_frames, spacing=0.25), w, h, seed=800 + k) for k in range(n_frames)]


def main_round(rank, world, comm, out, n_rounds):
    """ks_integrate_round_exact: this rank marches frame round * world + rank of every round and applies, per round, what the
    tiles it owns receive from every rank, in frame order."""
    from kimera_semantics_amd import binding as B
    frames = round_frames(world * n_rounds)
    marcher = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **round_config_kw()))
    owner = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **round_config_kw()))
    stats = []
    for r in range(n_rounds):
        f = frames[r * world + rank]
        stats.append(owner.integrate_round_exact(marcher, comm, rank, world, r * world, f.T_G_C, f.xyz, f.rgba, f.labels))
    keys, rec = export_all(owner)
    np.savez(os.path.join(out, f"round_rank{rank}.npz"), keys=keys, rec=rec[:, :, :25],
             marched=np.array([s["updates_marched"] for s in stats]), applied=np.array([s["updates_applied"] for s in stats]),
             sent=np.array([s["bytes_sent"] for s in stats]), origin=np.array([int(s["origin_voxel_touched"]) for s in stats]))
    marcher.close()
    owner.close()
    print("round worker", rank, "ok", stats)


def main_seq(out, n_frames):
    """All frames in order on ONE context (what the owners' tiles of the exact rounds must equal)."""
    from kimera_semantics_amd import binding as B
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **round_config_kw()))
    for f in round_frames(n_frames):
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    keys, rec = export_all(h)
    np.savez(os.path.join(out, "round_seq.npz"), keys=keys, rec=rec[:, :, :25])
    h.close()
    print("sequential ok", len(keys), "tiles")


def main():
    import time
    t0 = time.time()
    if sys.argv[1] == "seq":
        return main_seq(sys.argv[2], int(sys.argv[3]))
    rank, world, uid_hex, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    from kimera_semantics_amd import binding as B
    lib = C.CDLL(os.environ["KS_RCCL_LIB"])

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_byte * 128)]
    uid = UniqueId()
    C.memmove(C.byref(uid), bytes.fromhex(uid_hex), 128)
    comm = C.c_void_p()
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert lib.ncclCommInitRank(C.byref(comm), world, uid, rank) == 0
    if len(sys.argv) > 5 and sys.argv[5].startswith("round"):
        return main_round(rank, world, comm.value, out, int(sys.argv[5].split(":")[1]))
    t1 = time.time()
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **config_kw()))   # small pool: it has to grow (frames and received tiles)
    t2 = time.time()
    stats, times = [], []
    for batch in range(2):
        ta = time.time()
        fr = frames_of(rank, world, batch)
        tb = time.time()
        for f in fr:
            h.integrate(f.T_G_C, f.xyz, None, f.labels)
        h.synchronize()
        tc = time.time()
        stats.append(h.reduce(comm.value, rank, world))
        times.append((round(tb - ta, 2), round(tc - tb, 2), round(time.time() - tc, 2)))
    keys, rec = export_all(h)
    np.savez(os.path.join(out, f"rank{rank}.npz"), keys=keys, rec=rec[:, :, :25],
             sent=np.array([s["tiles_sent"] for s in stats]), received=np.array([s["tiles_received"] for s in stats]))
    h.close()
    print("worker", rank, "ok", stats, "seconds: imports", round(t1 - t0, 2), "create", round(t2 - t1, 2),
          "per batch (render, integrate, reduce)", times, "total", round(time.time() - t0, 2))


if __name__ == "__main__":
    main()
