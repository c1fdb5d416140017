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
    if n:
        hip = C.CDLL("libamdhip64.so")
        d = C.c_void_p()
        assert hip.hipMalloc(C.byref(d), C.c_size_t(n * 65536)) == 0
        h.export_tiles(np.arange(n, dtype=np.uint32), d.value)
        assert hip.hipMemcpy(C.c_void_p(rec.ctypes.data), d, C.c_size_t(n * 65536), C.c_int(2)) == 0   # hipMemcpyDeviceToHost
        hip.hipFree(d)
    return keys, rec


def main():
    import time
    t0 = time.time()
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
