"""-m gpu: the MULTI-RANK path of ks_reduce (include/ks_hip.h) — count exchange, per-peer offsets, grouped
send / receive of keys and raw tile records, owner merge in ascending source-rank order, sender reset, repeated
reduce — with 2 and 3 ranks.  A development box has one GPU and RCCL refuses two ranks on one device, so the ranks
are PROCESSES SHARING THE GPU and the communicator is the test double tests/mock_rccl (same seven entry points,
messages through /dev/shm); the real RCCL path is exercised with one rank in test_parallel_gpu.py and with N ranks
by the driver's multi-GPU bench.  Expected result: the same exchange emulated inside one process with the
tile primitives (ks_export_tiles_device / ks_merge_tiles_device / ks_reset_tiles), bit for bit."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from kimera_semantics_amd import binding as B
from kimera_semantics_amd import parallel as PAR
from tests.reduce_worker import config_kw, export_all, frames_of

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_rccl", "libmock_rccl.so")


def _emulate(world):
    """The protocol with in-process 'ranks': owner r receives, in ascending source-rank order, every tile another
    rank holds that r owns (untouched tiles are empty deltas: merging them is a no-op); the senders reset them."""
    import torch
    hs = [B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **config_kw())) for _ in range(world)]
    for batch in range(2):
        for r in range(world):
            for f in frames_of(r, world, batch):
                hs[r].integrate(f.T_G_C, f.xyz, None, f.labels)
        exported = []
        for r in range(world):
            keys = hs[r].tile_keys()
            buf = torch.empty((len(keys), 16384), dtype=torch.int32, device="cuda")
            hs[r].export_tiles(np.arange(len(keys), dtype=np.uint32), buf.data_ptr())
            exported.append((keys, buf, PAR.owner_of(keys, world)))
        torch.cuda.synchronize()
        for dst in range(world):
            ks, bufs = [], []
            for src in range(world):
                if src == dst:
                    continue
                keys, buf, own = exported[src]
                sel = np.nonzero(own == dst)[0]
                ks.append(keys[sel])
                bufs.append(buf[torch.from_numpy(sel.astype(np.int64)).cuda()])
            k = np.concatenate(ks)
            if len(k):
                hs[dst].merge_tiles(k, torch.cat(bufs, dim=0).contiguous().data_ptr())
        for src in range(world):
            keys, _, own = exported[src]
            hs[src].reset_tiles(np.nonzero(own != src)[0].astype(np.uint32))
    out = []
    for r in range(world):
        keys, rec = export_all(hs[r], torch)
        out.append(dict(zip(keys.tolist(), rec[:, :, :25])))
        hs[r].close()
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_ks_reduce_multi_rank_equals_emulation(tmp_path, world):
    if not os.path.exists(MOCK):
        pytest.fail("tests/mock_rccl/libmock_rccl.so not built: run __graft_entry__.build()")
    lib = C.CDLL(MOCK)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_byte * 128)]
    uid = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    env = dict(os.environ, KS_RCCL_LIB=MOCK)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "reduce_worker.py"), str(r), str(world), bytes(uid).hex(), str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    print("\n".join(o.strip().splitlines()[-1] for o in outs))
    import time
    t_em = time.time()
    want = _emulate(world)
    print("emulation seconds", round(time.time() - t_em, 2))
    total_sent = total_recv = 0
    for r in range(world):
        with np.load(os.path.join(str(tmp_path), f"rank{r}.npz")) as npz:
            got = {k: npz[k] for k in npz.files}   # (an NpzFile re-reads the array on every access)
        total_sent += int(got["sent"].sum())
        total_recv += int(got["received"].sum())
        assert int(got["sent"][0]) > 0 and int(got["sent"][1]) > 0, "both reduces must move tiles"
        gk = got["keys"].tolist()
        own = PAR.owner_of(got["keys"], world)
        owned = {k: got["rec"][i] for i, k in enumerate(gk) if own[i] == r}
        want_owned = {k: v for k, v in want[r].items() if PAR.owner_of(np.array([k], dtype=np.uint64), world)[0] == r}
        assert sorted(owned) == sorted(want_owned)
        for k in owned:
            assert np.array_equal(owned[k], want_owned[k]), f"rank {r} tile {k}"
        # what a rank sent away is an empty delta again
        for i, k in enumerate(gk):
            if own[i] != r:
                assert (got["rec"][i][:, 3] == 255).all(), f"rank {r} kept content of tile {k} it does not own"
    assert total_sent == total_recv


def _sequential_tiles(n_frames):
    """All frames in order on ONE context: tile key -> [512, 25] u32 records."""
    from tests.reduce_worker import round_config_kw, round_frames
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **round_config_kw()))
    for f in round_frames(n_frames):
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    keys, rec = export_all(h)
    h.close()
    return dict(zip(keys.tolist(), rec[:, :, :25]))


def test_exact_round_on_one_rank_equals_sequential_integration():
    """ks_integrate_round_exact with world = 1: the frame is marched by one context (records: voxel, position, sdf, weight),
    applied by another — the map must be the plain integration's, bit for bit, and the serial oracle's."""
    from kimera_semantics_amd import synth
    from oracle import oracle_py as O
    from tests.reduce_worker import round_config_kw, round_frames
    from tests.util import compare_maps
    frames = round_frames(4)
    marcher = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **round_config_kw()))
    owner = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **round_config_kw()))
    o = O.Oracle(O.default_config(integrator_threads=1, **round_config_kw()))
    for k, f in enumerate(frames):
        so = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        st = owner.integrate_round_exact(marcher, None, 0, 1, k, f.T_G_C, f.xyz, f.rgba, f.labels)
        assert st["updates_marched"] == st["updates_applied"] == so.n_voxel_updates and not st["origin_voxel_touched"], st
    compare_maps(o, owner, exact=True)
    want = _sequential_tiles(4)
    keys, rec = export_all(owner)
    assert sorted(keys.tolist()) == sorted(want)
    for i, k in enumerate(keys.tolist()):
        assert np.array_equal(rec[i][:, :25], want[k]), k


@pytest.mark.parametrize("world", [2, 3])
def test_exact_round_multi_rank_is_the_sequential_map_bit_for_bit(tmp_path, world):
    """The frames of two rounds sharded over 2 / 3 ranks (processes sharing the GPU, the librccl test double): every rank marches
    its frames, every update travels to the owner of its tile as a 20-byte record, the owner applies the frames in frame order —
    the tiles a rank owns are EXACTLY those of the one-GPU sequential integration of all frames (SURVEY.md par. 8e asked for a
    reduce of overlapping blocks; merging maps cannot be exact, shipping the updates is)."""
    if not os.path.exists(MOCK):
        pytest.fail("tests/mock_rccl/libmock_rccl.so not built: run __graft_entry__.build()")
    lib = C.CDLL(MOCK)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_byte * 128)]
    uid = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    env = dict(os.environ, KS_RCCL_LIB=MOCK)
    n_rounds = 2
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "reduce_worker.py"), str(r), str(world), bytes(uid).hex(), str(tmp_path),
                               f"round:{n_rounds}"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("a rank hung")
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    want = _sequential_tiles(world * n_rounds)
    want_owner = PAR.owner_of(np.array(sorted(want), dtype=np.uint64), world)
    want_by_rank = {r: {k for k, ow in zip(sorted(want), want_owner.tolist()) if ow == r} for r in range(world)}
    marched = applied = 0
    for r in range(world):
        with np.load(os.path.join(str(tmp_path), f"round_rank{r}.npz")) as npz:
            got = {k: npz[k] for k in npz.files}
        assert not got["origin"].any(), "the test scene must stay away from the world origin's voxel (see ks_k_shard.h)"
        marched += int(got["marched"].sum())
        applied += int(got["applied"].sum())
        assert int(got["sent"].sum()) > 0
        gk = got["keys"].tolist()
        assert set(gk) == want_by_rank[r], f"rank {r}: the tiles it holds are not the tiles it owns of the sequential map"
        for i, k in enumerate(gk):
            assert np.array_equal(got["rec"][i], want[k]), f"rank {r} tile {k}"
    assert marched == applied > 0


def test_bench_gpus_2_spawns_its_ranks_and_reports_the_exchange(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: bench.py re-executes itself under torch.distributed.run with two
    ranks (here: two processes sharing the one GPU, torch.distributed over gloo, ks_reduce through the librccl test double —
    KS_BENCH_SHARED_GPU=1), the line says n_gpus = 2, carries the per-region exchange and the C5 record, and what ks_reduce
    reports as sent is whole tiles (the 64 KiB record + the 8-byte key of every dirty tile)."""
    import json
    if not os.path.exists(MOCK):
        pytest.fail("tests/mock_rccl/libmock_rccl.so not built: run __graft_entry__.build()")
    full = str(tmp_path / "full.json")
    env = dict(os.environ, KS_BENCH_SHARED_GPU="1", KS_RCCL_LIB=MOCK, KS_BENCH_FULL=full)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--no-cpu-baseline",
                        "--no-secondary", "--width", "320", "--height", "240"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["reduce"]["tiles_sent"] > 0 and d["reduce"]["bytes_sent"] == d["reduce"]["tiles_sent"] * (65536 + 8)
    c5 = [s for s in d["secondary"] if s["config"] == "C5"]
    assert len(c5) == 1 and "error" not in c5[0], c5
    assert c5[0]["reduce"]["tiles_sent"] > 0 and c5[0]["reduce"]["bytes_sent"] == c5[0]["reduce"]["tiles_sent"] * (65536 + 8)
    # the exact split (ks_integrate_round_exact through the same communicator): the owners' tiles ARE the sequential map
    assert c5[0]["bit_exact_vs_sequential"] is True and c5[0]["exchange"]["bytes_sent"] > 0, c5
    fullrec = json.load(open(full))
    assert fullrec["n_gpus"] == 2 and "ks_reduce" in fullrec["config"]["parallelism"]
