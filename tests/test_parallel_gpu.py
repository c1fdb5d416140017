"""-m gpu: the device side of the multi-GPU reduce on one GPU.  Several integrators stand in
for ranks; ks_export_tiles_device / ks_merge_tiles_device must agree bit-for-bit with the
numpy restatement of the merge rule (tests/merge_ref.py)."""
import os

import numpy as np
import pytest

from kimera_semantics_amd import binding as B
from kimera_semantics_amd import synth
from oracle import oracle_py as O
from tests import merge_ref as M
from tests.util import COMMON, NO_EARLY_OUT

pytestmark = pytest.mark.gpu


def _export_all(h):
    import torch
    keys = h.tile_keys()
    buf = torch.empty((len(keys), M.TILE_WORDS), dtype=torch.int32, device="cuda")
    h.export_tiles(np.arange(len(keys), dtype=np.uint32), buf.data_ptr())
    torch.cuda.synchronize()
    return keys, buf


@pytest.mark.parametrize("method,color_mode", [(0, 1), (1, 1), (0, 0)])
def test_merge_kernel_matches_numpy_rule(method, color_mode):
    import torch
    sc = synth.make_scene("room")
    kw = dict(COMMON, method=method, color_mode=color_mode, max_consecutive_ray_collisions=NO_EARLY_OUT)
    ranks = []
    for r in range(3):
        h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
        f = synth.render_frame(sc, synth.arc_pose(r, 3), 128, 96, seed=200 + r)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        ranks.append(h)
    raw = []
    for h in ranks:
        k, buf = _export_all(h)
        raw.append((k, buf.cpu().numpy().view(np.uint32).reshape(len(k), 512, 32).copy(), buf))
    # expected: rank 0's map, then ranks 1 and 2 merged into it in order
    lut = synth.default_label_colors()
    exp = {int(k): rec.copy() for k, rec in zip(raw[0][0], raw[0][1])}
    for src in (1, 2):
        for k, rec in zip(raw[src][0].tolist(), raw[src][1]):
            if k not in exp:
                exp[k] = M.empty_tile()
            M.merge_records(exp[k], rec, 10000.0, color_mode, lut)
    # device: merge into rank 0
    for src in (1, 2):
        ranks[0].merge_tiles(raw[src][0], raw[src][2].data_ptr())
    k0, buf0 = _export_all(ranks[0])
    got = buf0.cpu().numpy().view(np.uint32).reshape(len(k0), 512, 32)
    assert sorted(k0.tolist()) == sorted(exp)
    overlap = len(set(raw[0][0].tolist()) & set(raw[1][0].tolist()))
    assert overlap > 10, "poses must overlap for the test to mean anything"
    for k, rec in zip(k0.tolist(), got):
        assert np.array_equal(rec[:, :25], exp[k][:, :25]), f"tile {k}"


def test_reduced_map_is_close_to_sequential_integration():
    """Merging per-frame maps is not the same arithmetic as integrating the frames one after the
    other (the +-truncation clamp and the weight clamp act per update there): labels must agree
    except near ties, TSDF within a loose tolerance."""
    sc = synth.make_scene("room")
    kw = dict(COMMON, method=1)
    frames = [synth.render_frame(sc, synth.arc_pose(r, 3), 128, 96, seed=300 + r) for r in range(3)]
    seq = O.Oracle(O.default_config(**kw))
    for f in frames:
        seq.integrate(f.T_G_C, f.xyz, None, f.labels)
    hs = []
    for f in frames:
        h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
        h.integrate(f.T_G_C, f.xyz, None, f.labels)
        hs.append(h)
    for src in (1, 2):
        k, buf = _export_all(hs[src])
        hs[0].merge_tiles(k, buf.data_ptr())
    idx = seq.block_indices()
    assert np.array_equal(idx, hs[0].block_indices())
    _, ot, os_ = seq.download(idx)
    _, ht, hs_ = hs[0].download(idx)
    touched = ot["weight"] > 0
    assert np.array_equal(touched, ht["weight"] > 0)
    agree = (os_["label"] == hs_["label"])[touched].mean()
    assert agree > 0.99, agree
    dd = np.abs(ot["distance"] - ht["distance"])[touched]
    assert np.median(dd) < 1e-4 and (dd > 0.05).mean() < 0.05


def test_clear_resets_the_map():
    sc = synth.make_scene("room")
    h = B.HipIntegrator(B.default_config(max_tiles=2048, max_points=1 << 16, **dict(COMMON, method=0)))
    f = synth.render_frame(sc, synth.single_pose(), 96, 72, seed=1)
    s1 = h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    i1, t1, _ = h.download()
    h.clear()
    assert len(h.block_indices()) == 0 and len(h.tile_keys()) == 0
    h2 = B.HipIntegrator(B.default_config(max_tiles=2048, max_points=1 << 16, **dict(COMMON, method=0)))
    s2 = h2.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    assert s1.n_rays_cast == s2.n_rays_cast


@pytest.mark.parametrize("color_mode", [1, 0])
def test_one_merge_call_folds_duplicate_keys_in_order(color_mode):
    """reduce_maps hands the owner everything it received in ONE ks_merge_tiles_device call, the
    buffer ordered by source rank: tiles with the same key must be folded in buffer order, i.e.
    the result equals merging the sources one call after the other (bit for bit)."""
    import torch
    sc = synth.make_scene("room")
    kw = dict(COMMON, method=1, color_mode=color_mode, max_consecutive_ray_collisions=NO_EARLY_OUT)

    def rank(r):
        h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 16, **kw))
        f = synth.render_frame(sc, synth.arc_pose(r, 4), 128, 96, seed=400 + r)
        h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
        return h

    srcs = [rank(r) for r in (1, 2, 3)]
    exported = [_export_all(h) for h in srcs]
    a, b = rank(0), rank(0)
    for k, buf in exported:                       # one call per source
        a.merge_tiles(k, buf.data_ptr())
    keys = np.concatenate([k for k, _ in exported])
    payload = torch.cat([buf for _, buf in exported], dim=0).contiguous()
    assert len(set(keys.tolist())) < len(keys), "sources must share tiles"
    b.merge_tiles(keys, payload.data_ptr())       # one call for all sources
    ka, bufa = _export_all(a)
    kb, bufb = _export_all(b)
    da = dict(zip(ka.tolist(), bufa.cpu().numpy().view(np.uint32).reshape(len(ka), 512, 32)))
    db = dict(zip(kb.tolist(), bufb.cpu().numpy().view(np.uint32).reshape(len(kb), 512, 32)))
    assert sorted(da) == sorted(db)
    for k in da:
        assert np.array_equal(da[k][:, :25], db[k][:, :25]), f"tile {k}"


def _rccl_comm_world1():
    """An ncclComm_t of one rank, created with the librccl that ks_reduce loads."""
    import ctypes as C
    lib = C.CDLL(os.environ.get("KS_RCCL_LIB", "librccl.so.1"))

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert lib.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    return lib, comm


def test_ks_reduce_c_abi_single_rank_and_repeatable():
    """ks_reduce through the C ABI with a real RCCL communicator (world size 1 is all one GPU allows here):
    nothing travels, the map is untouched, the call is repeatable; ks_reset_tiles empties a tile;
    ks_tile_owner equals the Python protocol's owner function."""
    from kimera_semantics_amd import parallel as PAR
    h = B.HipIntegrator(B.default_config(max_tiles=4096, max_points=1 << 15, **dict(COMMON, method=1)))
    sc = synth.make_scene("room")
    f = synth.render_frame(sc, synth.single_pose(), 160, 120, seed=0)
    h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    idx, t0, s0 = h.download()
    lib, comm = _rccl_comm_world1()
    for _ in range(2):
        st = h.reduce(comm.value, 0, 1)
        assert st["tiles_sent"] == 0 and st["tiles_received"] == 0 and st["tiles_local"] == len(h.tile_keys())
    _, t1, s1 = h.download(idx)
    assert t0.tobytes() == t1.tobytes() and s0.tobytes() == s1.tobytes()
    keys = h.tile_keys()
    for w in (2, 3, 8):
        assert [B.lib().ks_tile_owner(int(k), w) for k in keys[:64]] == PAR.owner_of(keys[:64], w).tolist()
    h.reset_tiles(np.arange(len(keys), dtype=np.uint32))
    _, t2, s2 = h.download(idx)
    assert not (t2["weight"] > 0).any() and (s2["label"] == 0).all()
    lib.ncclCommDestroy(comm)
