"""N>1 path on CPU: two gloo ranks run the exchange protocol of kimera_semantics_amd.parallel
over numpy tile stores; the result must equal a serial merge in the documented order
(owner's own state first, then the other ranks ascending)."""
import os
import socket

import numpy as np
import pytest

from kimera_semantics_amd import parallel as PAR
from kimera_semantics_amd import synth
from tests import merge_ref as M


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_rank_tiles(rank, n_tiles=12, seed=0):
    """Random but valid tile records; keys overlap across ranks (same key universe)."""
    rng = np.random.default_rng(seed * 100 + rank)
    universe = (np.arange(40, dtype=np.uint64) * np.uint64(7919) + np.uint64(12345)) << np.uint64(3)
    keys = rng.choice(universe, size=n_tiles, replace=False)
    tiles = {}
    for k in keys.tolist():
        t = M.empty_tile()
        touched = rng.random(512) < 0.6
        t[touched, 0] = rng.uniform(-0.2, 0.2, touched.sum()).astype(np.float32).view(np.uint32)
        t[touched, 1] = rng.uniform(0.01, 30.0, touched.sum()).astype(np.float32).view(np.uint32)
        pri = (M.PRIOR_INIT + rng.uniform(-20, 0, (touched.sum(), 21))).astype(np.float32)
        t[touched, 4:25] = pri.view(np.uint32)
        t[touched, 3] = np.argmax(pri, axis=1).astype(np.uint32)
        tiles[k] = t
    return tiles


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lut = synth.default_label_colors()
    import torch
    assert PAR.warm_up(torch.device("cpu")) == world   # the bench's untimed connection warm-up
    store = M.NumpyTileStore(lut)
    for k, t in _make_rank_tiles(rank).items():
        store.add(k, t.copy())
    stats = PAR.reduce_maps(store)
    # a second reduce without new integration must not count anything twice: what was sent has
    # started over as an empty delta on the sender
    PAR.reduce_maps(store)
    keys = store.tile_keys()
    mine = PAR.owned_tile_mask(keys, rank, world)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), keys=keys[mine],
             recs=np.stack([store.tiles[int(k)] for k in keys[mine]]) if mine.any() else np.zeros((0, 512, 32), np.uint32),
             sent=stats["tiles_sent"], recv=stats["tiles_received"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_maps_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    lut = synth.default_label_colors()
    per_rank = [_make_rank_tiles(r) for r in range(world)]
    all_keys = sorted({k for t in per_rank for k in t})
    owners = PAR.owner_of(np.array(all_keys, dtype=np.uint64), world)
    got = {}
    total_sent = total_recv = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        total_sent += int(z["sent"]); total_recv += int(z["recv"])
        for k, rec in zip(z["keys"].tolist(), z["recs"]):
            assert k not in got, "a tile must have exactly one owner"
            got[k] = rec
    assert total_sent == total_recv
    assert sorted(got) == all_keys
    for k, owner in zip(all_keys, owners.tolist()):
        # documented order: the owner's own state first, then the other ranks ascending
        exp = per_rank[owner][k].copy() if k in per_rank[owner] else M.empty_tile()
        for src in range(world):
            if src != owner and k in per_rank[src]:
                M.merge_records(exp, per_rank[src][k], 10000.0, 1, lut)
        assert np.array_equal(got[k], exp), f"tile {k} (owner {owner})"


def test_owner_partition_is_total_and_balanced():
    keys = (np.arange(20000, dtype=np.uint64) * np.uint64(2654435761)) ^ np.uint64(0xABCDEF)
    for world in (1, 2, 4, 8):
        own = PAR.owner_of(keys, world)
        assert own.min() >= 0 and own.max() < world
        counts = np.bincount(own, minlength=world)
        assert counts.min() > 0.8 * len(keys) / world
    assert np.array_equal(PAR.owner_of(keys, 8), PAR.owner_of(keys.copy(), 8))  # deterministic


def test_merge_rule_properties():
    lut = synth.default_label_colors()
    a, b = M.empty_tile(), M.empty_tile()
    a[0, 0] = np.float32(0.1).view(np.uint32); a[0, 1] = np.float32(2.0).view(np.uint32); a[0, 3] = 5
    a[0, 4 + 5] = np.float32(-0.9).view(np.uint32)
    b[0, 0] = np.float32(-0.1).view(np.uint32); b[0, 1] = np.float32(2.0).view(np.uint32); b[0, 3] = 7
    b[0, 4 + 7] = np.float32(-0.8).view(np.uint32)
    out = M.merge_records(b.copy(), a, 10000.0, 1, lut)
    assert out[0, 0].view(np.float32) == 0.0 and out[0, 1].view(np.float32) == 4.0
    pri = out[0, 4:25].view(np.float32)
    assert pri[5] == np.float32(M.PRIOR_INIT + (np.float32(-0.9) - M.PRIOR_INIT))
    assert out[0, 3] == int(np.argmax(pri))
    # merging an untouched tile changes nothing
    assert np.array_equal(M.merge_records(b.copy(), M.empty_tile(), 10000.0, 1, lut), b)
    # weight clamp
    big = b.copy(); big[0, 1] = np.float32(9999.5).view(np.uint32)
    assert M.merge_records(big, a, 10000.0, 1, lut)[0, 1].view(np.float32) == 10000.0


def test_exact_round_two_ranks_on_the_functional_model_is_the_sequential_map(tmp_path):
    """ks_integrate_round_exact with world = 2, WITHOUT a GPU: two processes drive the host functional model of the library
    (tools/emu: the device code compiled for the CPU) and exchange their update records through the librccl test double built
    against the same stand-in runtime; a third process integrates the four frames in order on one context.  The tiles a rank
    owns must be the sequential ones, bit for bit (the GPU tier runs the same worker with 2 and 3 ranks on the real device)."""
    import ctypes as C
    import subprocess
    import sys
    import numpy as np
    from kimera_semantics_amd import parallel as PAR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        pytest.skip("host clang++ of the ROCm toolchain not found")
    r = subprocess.run(["bash", os.path.join(root, "tools", "emu", "build_emu.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    emu = os.path.join(root, "tools", "emu", "_build", "libks_hip_emu.so")
    mock = str(tmp_path / "libmock_rccl_emu.so")
    r = subprocess.run([cxx, "-std=c++17", "-O1", "-fPIC", "-shared", "-DKS_EMU_BUILD", "-Wno-unknown-attributes", "-I", os.path.join(root, "tools", "emu"),
                        "-o", mock, os.path.join(root, "tests", "mock_rccl", "mock_rccl.cpp"), "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = C.CDLL(mock)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_byte * 128)]
    uid = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    world, n_rounds = 2, 2
    env = dict(os.environ, KS_HIP_LIB=emu, KS_RCCL_LIB=mock, KS_ROUND_WH="48x36")
    worker = os.path.join(root, "tests", "reduce_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(rk), str(world), bytes(uid).hex(), str(tmp_path), f"round:{n_rounds}"], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for rk in range(world)]
    procs.append(subprocess.Popen([sys.executable, worker, "seq", str(tmp_path), str(world * n_rounds)], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    with np.load(os.path.join(str(tmp_path), "round_seq.npz")) as npz:
        want = dict(zip(npz["keys"].tolist(), npz["rec"]))
    owners = PAR.owner_of(np.array(sorted(want), dtype=np.uint64), world)
    total = 0
    for rk in range(world):
        with np.load(os.path.join(str(tmp_path), f"round_rank{rk}.npz")) as npz:
            got = {k: npz[k] for k in npz.files}
        assert not got["origin"].any()
        mine = {k for k, ow in zip(sorted(want), owners.tolist()) if ow == rk}
        assert set(got["keys"].tolist()) == mine
        for i, k in enumerate(got["keys"].tolist()):
            assert np.array_equal(got["rec"][i], want[k]), f"rank {rk} tile {k}"
        total += int(got["applied"].sum())
        assert int(got["sent"].sum()) > 0
    assert total > 0
