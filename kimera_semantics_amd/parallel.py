"""Frame-sharded multi-GPU integration: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests), each rank integrates its share of
the frames into its own GPU-resident map with no data-path collective; `reduce_maps` is the one
exchange step that turns the per-rank partial maps into one owner-sharded global map.

New functionality: the reference is a single process.  Protocol:
  1. every tile (8^3 voxels) has an owner rank = splitmix64(tile key) % world;
  2. ranks exchange per-destination tile counts, then keys and raw 64 KiB tile records with a
     single all-to-all each — on a fully connected xGMI node every rank talks to its 7 peers
     at once, which a ring all-reduce (per-link bound) cannot do;
  3. the owner merges everything it received with ONE ks_merge_tiles_device call; tiles that
     several ranks sent for the same key are folded in ascending source-rank order
     (deterministic): weight-averaged TSDF (Voxblox's layer-merge rule), additive class
     log-likelihoods, argmax + colour.
After the reduce, rank r holds the authoritative state of the tiles it owns; its copies of tiles it sent
away start over as empty deltas, so the reduce can be repeated batch after batch.  (The same exchange
for a C/C++ host calling RCCL directly: ks_reduce in include/ks_hip.h.)

The exchange logic is backend-agnostic: `store` only needs tile_keys() / export(slots) /
merge(keys, payload); tests drive it on CPU with a numpy store over gloo.
"""
from __future__ import annotations

import numpy as np

TILE_WORDS = 16384  # 65536 B as int32


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def owner_of(keys: np.ndarray, world: int) -> np.ndarray:
    """Owner rank of each packed tile key."""
    return (splitmix64(np.asarray(keys, dtype=np.uint64)) % np.uint64(world)).astype(np.int64)


class HipTileStore:
    """Adapts a binding.HipIntegrator to the exchange protocol; payloads are torch CUDA tensors."""

    def __init__(self, integrator, device):
        import torch
        self.torch = torch
        self.integ = integrator
        self.device = device

    def tile_keys(self) -> np.ndarray:
        return self.integ.tile_keys()

    def export(self, slots: np.ndarray):
        buf = self.torch.empty((len(slots), TILE_WORDS), dtype=self.torch.int32, device=self.device)
        if len(slots):
            self.integ.export_tiles(slots, buf.data_ptr())
        return buf

    def empty(self, n: int):
        return self.torch.empty((n, TILE_WORDS), dtype=self.torch.int32, device=self.device)

    def merge(self, keys: np.ndarray, payload):
        if len(keys):
            # the payload was produced on torch's stream (RCCL); the merge kernel runs on the
            # integrator's own stream
            self.torch.cuda.synchronize(self.device)
            self.integ.merge_tiles(keys, payload.data_ptr())

    def reset(self, slots: np.ndarray):
        if len(slots):
            self.integ.reset_tiles(slots)


def reduce_maps(store, group=None) -> dict:
    """All-to-all reduce of per-rank partial maps to tile owners.  Collective: every rank of
    `group` must call it.  Returns traffic statistics."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    keys = store.tile_keys()
    own = owner_of(keys, world)
    send_slots = [np.nonzero(own == dst)[0].astype(np.uint32) if dst != rank else np.zeros(0, np.uint32)
                  for dst in range(world)]
    send_counts = [len(s) for s in send_slots]
    slots_cat = np.concatenate(send_slots) if world > 1 else np.zeros(0, np.uint32)
    payload = store.export(slots_cat)
    dev = payload.device
    # 1) counts
    t_send = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    t_recv = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(t_recv, t_send, group=group)
    recv_counts = [int(x) for x in t_recv.cpu().tolist()]
    n_recv = sum(recv_counts)
    # 2) keys (as int64 bit patterns) and 3) payload
    k_send = torch.from_numpy(keys[slots_cat.astype(np.int64)].view(np.int64).copy()).to(dev)
    k_recv = torch.empty(n_recv, dtype=torch.int64, device=dev)
    dist.all_to_all_single(k_recv, k_send, recv_counts, send_counts, group=group)
    p_recv = store.empty(n_recv)
    dist.all_to_all_single(p_recv, payload, recv_counts, send_counts, group=group)
    # 4) deterministic merge: the receive buffer is ordered by source rank, and the store folds
    #    tiles of the same key in buffer order
    k_host = k_recv.cpu().numpy().view(np.uint64)
    if n_recv:
        store.merge(k_host, p_recv)
    # 5) what was sent starts over as an empty delta on the sender: a second reduce counts nothing twice
    if hasattr(store, "reset"):
        store.reset(slots_cat)
    return {"tiles_sent": int(sum(send_counts)), "tiles_received": n_recv, "tiles_local": int(len(keys)),
            "bytes_sent": int(sum(send_counts)) * TILE_WORDS * 4}


def warm_up(device, group=None):
    """Establishes the point-to-point connections the all-to-alls of reduce_maps use (RCCL sets
    them up lazily on first use); call once outside any timed region."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t_send = torch.ones(world, dtype=torch.int64, device=device)
    t_recv = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(t_recv, t_send, group=group)
    counts = [1] * world
    p_send = torch.zeros((world, TILE_WORDS), dtype=torch.int32, device=device)
    p_recv = torch.empty((world, TILE_WORDS), dtype=torch.int32, device=device)
    dist.all_to_all_single(p_recv, p_send, counts, counts, group=group)
    return int(t_recv.sum().item())


def owned_tile_mask(keys: np.ndarray, rank: int, world: int) -> np.ndarray:
    return owner_of(keys, world) == rank


def rccl_comm(rank: int, world: int, device, timeout: float = 120.0):
    """An ncclComm_t (as an int) over the ranks of the default process group, created with the SAME librccl that
    ks_reduce loads (KS_RCCL_LIB is pointed at torch's copy so the process holds one RCCL): rank 0 draws the
    unique id, torch.distributed broadcasts its 128 bytes, every rank calls ncclCommInitRank.  One rank per GPU."""
    import ctypes as C
    import os

    import torch
    import torch.distributed as dist
    path = os.environ.get("KS_RCCL_LIB")
    if not path:
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        path = cand if os.path.exists(cand) else "librccl.so.1"
        os.environ["KS_RCCL_LIB"] = path
    lib = C.CDLL(path)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_byte * 128)]
    uid = UniqueId()
    if rank == 0:
        rc = lib.ncclGetUniqueId(C.byref(uid))
        if rc != 0:
            raise RuntimeError(f"ncclGetUniqueId: {rc}")
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).to(device)
    if world > 1:
        dist.broadcast(t, src=0)
    raw = bytes(t.cpu().numpy().tobytes())
    C.memmove(C.byref(uid), raw, 128)
    comm = C.c_void_p()
    lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    on_gpu = getattr(device, "type", str(device)) == "cuda" or str(device).startswith("cuda")
    if on_gpu:
        torch.cuda.set_device(device)
    # the collective initialisation runs on a helper thread with a deadline: a bootstrap that does not complete raises
    # RcclBootstrapStuck instead of hanging the caller for ever
    import threading
    box = {}

    def init():
        if on_gpu:
            torch.cuda.set_device(device)
        box["rc"] = lib.ncclCommInitRank(C.byref(comm), world, uid, rank)
    th = threading.Thread(target=init, daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        # The bootstrap is stuck inside the library with the unique id consumed: this process cannot take part in a later
        # collective cleanly (a fallback exchange would hang on its next barrier while other ranks may still complete their
        # side of the bootstrap).  What to do about that is the caller's decision — a library helper does not end the process.
        raise RcclBootstrapStuck(f"ncclCommInitRank did not return within {timeout} s on rank {rank}; the helper thread is still inside "
                                 "librccl and the unique id is consumed: do not start another collective from this process")
    if box.get("rc", -1) != 0:
        raise RuntimeError(f"ncclCommInitRank: {box.get('rc')}")
    return comm.value


class RcclBootstrapStuck(RuntimeError):
    """rccl_comm: ncclCommInitRank did not return before the deadline (see there)."""


def rccl_abort(comm: int) -> None:
    """Tears down a communicator made by rccl_comm without the collective handshake of ncclCommDestroy."""
    import ctypes as C
    import os
    lib = C.CDLL(os.environ.get("KS_RCCL_LIB") or "librccl.so.1")
    fn = getattr(lib, "ncclCommAbort", None)
    if fn is not None and comm:
        fn.argtypes = [C.c_void_p]
        fn(C.c_void_p(comm))
