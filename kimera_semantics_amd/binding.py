"""ctypes binding of the C ABI in include/ks_hip.h (libks_hip.so).

Plumbing for tests and bench.py; the product boundary is the C ABI itself and the C++
adapter in kimera_semantics_amd/host/.  There is NO CPU fallback: if the HIP library is
missing or no GPU is present, constructing an integrator raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KS_HIP_LIB") or os.path.join(_HERE, "libks_hip.so")   # KS_HIP_LIB: a diagnostics build
NUM_LABELS = 21

KS_METHOD_FAST, KS_METHOD_MERGED = 0, 1
KS_COLOR_MODE_COLOR, KS_COLOR_MODE_SEMANTIC, KS_COLOR_MODE_SEMANTIC_PROBABILITY = 0, 1, 2
KS_ORDER_MIXED, KS_ORDER_SORTED, KS_ORDER_MIXED_1024_GROUPS = 0, 1, 2   # include/ks_hip.h
KS_BUNDLE_ORDER_REFERENCE, KS_BUNDLE_ORDER_CANONICAL = 0, 1
KS_EARLY_OUT_EXACT = 1   # value of KsConfig.early_out_phase_growth: the reference's serial early-out result
KS_ERR_LABEL_RANGE, KS_ERR_PROBABILITY, KS_ERR_POOL_FULL, KS_ERR_NO_DEVICE, KS_ERR_UNSUPPORTED = -2, -3, -5, -7, -8

STAGES = ["points", "sort_points", "rays", "march", "emit", "sort_pairs", "apply", "apply_long"]

TSDF_DTYPE = np.dtype([("distance", "<f4"), ("weight", "<f4"), ("color", "u1", (4,))])
SEM_DTYPE = np.dtype([("label", "u1"), ("pad", "u1", (3,)), ("priors", "<f4", (NUM_LABELS,)),
                      ("color", "u1", (4,))])

# Every symbol include/ks_hip.h declares (checked by tests/test_abi.py).
ABI_SYMBOLS = [
    "ks_default_config", "ks_create", "ks_destroy", "ks_last_error", "ks_set_color_to_label",
    "ks_integrate_points", "ks_integrate_points_device", "ks_integrate_depth", "ks_integrate_depth_device", "ks_num_blocks", "ks_get_block_indices",
    "ks_get_updated_block_indices", "ks_count_updated_voxels", "ks_download_updated_voxels", "ks_download_blocks", "ks_upload_blocks", "ks_host_alloc", "ks_host_free", "ks_get_tile_keys", "ks_export_tiles_device", "ks_merge_tiles_device", "ks_clear", "ks_clear_voxels", "ks_reset_tiles", "ks_tile_owner", "ks_reduce",
    "ks_debug_radix_sort", "ks_synchronize", "ks_flush", "ks_stream",
    "ks_profile_enable", "ks_profile_get", "ks_early_out_iterations", "ks_early_out_stats", "ks_pipeline_shape", "ks_update_stats", "ks_integrate_round_exact",
]


class KsConfig(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_float), ("voxels_per_side", C.c_int32),
        ("truncation_distance", C.c_float), ("max_weight", C.c_float),
        ("min_ray_length_m", C.c_float), ("max_ray_length_m", C.c_float),
        ("voxel_carving_enabled", C.c_int32), ("use_const_weight", C.c_int32),
        ("allow_clear", C.c_int32), ("use_weight_dropoff", C.c_int32),
        ("use_sparsity_compensation_factor", C.c_int32), ("sparsity_compensation_factor", C.c_float),
        ("enable_anti_grazing", C.c_int32), ("start_voxel_subsampling_factor", C.c_float),
        ("max_consecutive_ray_collisions", C.c_int32), ("clear_checks_every_n_frames", C.c_int32),
        ("integration_order_mode", C.c_int32), ("integrator_threads", C.c_int32),
        ("method", C.c_int32), ("bundle_order", C.c_int32),
        ("semantic_measurement_probability", C.c_float), ("color_mode", C.c_int32),
        ("n_dynamic_labels", C.c_int32), ("dynamic_labels", C.c_uint8 * 32),
        ("label_rgba", (C.c_uint8 * 4) * 256),
        ("early_out_phase_growth", C.c_int32),
        ("device_id", C.c_int32), ("max_tiles", C.c_uint32), ("max_points", C.c_uint32), ("pipeline_frames", C.c_int32),
    ]


class KsRoundStats(C.Structure):
    _fields_ = [("updates_marched", C.c_uint64), ("updates_applied", C.c_uint64), ("bytes_sent", C.c_uint64),
                ("origin_voxel_touched", C.c_uint64), ("rays_cast", C.c_uint64)]


class KsFrameStats(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("n_valid_points", C.c_uint64), ("n_rays_cast", C.c_uint64),
                ("n_voxel_updates", C.c_uint64), ("n_blocks_allocated", C.c_uint64)]


class KsReduceStats(C.Structure):
    _fields_ = [("tiles_sent", C.c_uint64), ("tiles_received", C.c_uint64), ("tiles_local", C.c_uint64),
                ("bytes_sent", C.c_uint64)]


class KsProfile(C.Structure):
    _fields_ = [("ms", C.c_double * 8), ("launches", C.c_uint64 * 8), ("frames", C.c_uint64),
                ("updates", C.c_uint64), ("points", C.c_uint64), ("apply_kernel_ms", C.c_double),
                ("apply_kernel_launches", C.c_uint64), ("apply_kernel_updates", C.c_uint64),
                ("host_ms", C.c_double), ("host_wait_ms", C.c_double)]


def build(force: bool = False) -> str:
    """Compile libks_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(src_dir, f) for f in ("ks_hip.hip", "ks_types.h", "ks_k_rays.h", "ks_k_bundle_order.h", "ks_k_march.h", "ks_k_exact.h", "ks_k_apply.h",
                                                "ks_k_io.h", "ks_device_math.h", "ks_radix_sort.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "ks_hip.h"))
    stale = (not os.path.exists(LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", src_dir])
    return LIB_PATH


_lib = None


def lib():
    """Loads libks_hip.so.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the HIP path has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.ks_default_config.argtypes = [C.POINTER(KsConfig)]
        L.ks_create.argtypes = [C.POINTER(KsConfig), C.POINTER(vp)]
        L.ks_destroy.argtypes = [vp]
        L.ks_destroy.restype = None
        L.ks_last_error.argtypes = [vp]
        L.ks_last_error.restype = C.c_char_p
        L.ks_set_color_to_label.argtypes = [vp, vp, vp, C.c_size_t]
        L.ks_integrate_points.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.c_int, C.POINTER(KsFrameStats)]
        L.ks_integrate_points_device.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, C.c_int, C.POINTER(KsFrameStats)]
        L.ks_integrate_depth.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(KsFrameStats)]
        L.ks_integrate_depth_device.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, vp, C.c_int, C.POINTER(KsFrameStats)]
        L.ks_num_blocks.argtypes = [vp, C.POINTER(C.c_size_t)]
        L.ks_get_block_indices.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ks_get_updated_block_indices.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), C.c_int]
        L.ks_download_blocks.argtypes = [vp, vp, C.c_size_t, vp, vp]
        L.ks_count_updated_voxels.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.ks_download_updated_voxels.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t), vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ks_upload_blocks.argtypes = [vp, vp, C.c_size_t, vp, vp]
        L.ks_host_alloc.argtypes = [C.c_size_t]
        L.ks_host_alloc.restype = C.c_void_p
        L.ks_host_free.argtypes = [vp]
        L.ks_host_free.restype = None
        L.ks_get_tile_keys.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.ks_export_tiles_device.argtypes = [vp, vp, C.c_size_t, vp]
        L.ks_merge_tiles_device.argtypes = [vp, vp, C.c_size_t, vp]
        L.ks_clear.argtypes = [vp]
        L.ks_clear_voxels.argtypes = [vp]
        L.ks_reset_tiles.argtypes = [vp, vp, C.c_size_t]
        L.ks_tile_owner.argtypes = [C.c_uint64, C.c_int]
        L.ks_reduce.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(KsReduceStats)]
        L.ks_debug_radix_sort.argtypes = [vp, vp, vp, C.c_size_t, C.c_int, C.c_uint]
        L.ks_synchronize.argtypes = [vp]
        L.ks_flush.argtypes = [vp, C.POINTER(KsFrameStats)]
        L.ks_stream.argtypes = [vp]
        L.ks_stream.restype = vp
        L.ks_profile_enable.argtypes = [vp, C.c_int]
        L.ks_profile_get.argtypes = [vp, C.POINTER(KsProfile), C.c_int]
        L.ks_early_out_iterations.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.ks_early_out_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.ks_pipeline_shape.argtypes = [vp, C.POINTER(C.c_int32)]
        L.ks_update_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
        L.ks_integrate_round_exact.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_uint64, vp, vp, vp, vp, C.c_size_t, C.c_int, C.POINTER(KsRoundStats)]
        _lib = L
    return _lib


def default_config(**overrides) -> KsConfig:
    cfg = KsConfig()
    lib().ks_default_config(C.byref(cfg))
    apply_overrides(cfg, **overrides)
    return cfg


def apply_overrides(cfg, **overrides):
    for k, v in overrides.items():
        if k == "dynamic_labels":
            cfg.n_dynamic_labels = len(v)
            for i, lab in enumerate(v):
                cfg.dynamic_labels[i] = int(lab)
        elif k == "label_rgba":
            arr = np.ascontiguousarray(np.asarray(v, dtype=np.uint8).reshape(256, 4))
            C.memmove(cfg.label_rgba, arr.ctypes.data, 1024)
        else:
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
    return cfg


class KsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ks error {code}: {msg}")
        self.code = code


def _ptr(a):
    return None if a is None else a.ctypes.data


class HipIntegrator:
    """Thin RAII wrapper of ks_ctx."""

    def __init__(self, cfg: KsConfig):
        self.cfg = cfg
        self.vps = cfg.voxels_per_side
        self._h = C.c_void_p()
        rc = lib().ks_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise KsError(rc, lib().ks_last_error(None).decode())

    def _chk(self, rc):
        if rc != 0:
            raise KsError(rc, lib().ks_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            lib().ks_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_color_to_label(self, rgba_keys, labels):
        k = np.ascontiguousarray(rgba_keys, dtype=np.uint8).reshape(-1, 4)
        l = np.ascontiguousarray(labels, dtype=np.uint8)
        self._chk(lib().ks_set_color_to_label(self._h, _ptr(k), _ptr(l), len(l)))

    def integrate(self, T_G_C, xyz, rgba, labels, freespace=False) -> KsFrameStats:
        T = np.ascontiguousarray(T_G_C, dtype=np.float32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        labels = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint8)
        rgba = None if rgba is None else np.ascontiguousarray(rgba, dtype=np.uint8)
        st = KsFrameStats()
        self._chk(lib().ks_integrate_points(self._h, _ptr(T), _ptr(xyz), _ptr(rgba), _ptr(labels), xyz.shape[0],
                                            int(freespace), C.byref(st)))
        return st

    def integrate_depth(self, T_G_C, depth, K, label_img=None, rgba_img=None, freespace=False) -> KsFrameStats:
        """depth: [H,W] float32 metres or uint16 millimetres; K = (fx, fy, cx, cy)."""
        T = np.ascontiguousarray(T_G_C, dtype=np.float32)
        depth = np.ascontiguousarray(depth)
        fmt = 0 if depth.dtype == np.float32 else 1
        assert depth.dtype in (np.float32, np.uint16)
        Kc = np.ascontiguousarray(K, dtype=np.float32)
        lab = None if label_img is None else np.ascontiguousarray(label_img, dtype=np.uint8)
        col = None if rgba_img is None else np.ascontiguousarray(rgba_img, dtype=np.uint8)
        st = KsFrameStats()
        self._chk(lib().ks_integrate_depth(self._h, _ptr(T), _ptr(depth), fmt, _ptr(lab), _ptr(col), depth.shape[1],
                                           depth.shape[0], _ptr(Kc), int(freespace), C.byref(st)))
        return st

    def integrate_depth_device(self, T_G_C, d_depth: int, fmt: int, d_label_img: int, d_rgba_img: int, width: int,
                               height: int, K, freespace=False) -> KsFrameStats:
        """Raw device addresses (e.g. torch.Tensor.data_ptr()); fmt 0 = f32 metres, 1 = u16 mm."""
        T = np.ascontiguousarray(T_G_C, dtype=np.float32)
        Kc = np.ascontiguousarray(K, dtype=np.float32)
        st = KsFrameStats()
        self._chk(lib().ks_integrate_depth_device(self._h, _ptr(T), d_depth or None, fmt, d_label_img or None,
                                                  d_rgba_img or None, width, height, _ptr(Kc), int(freespace), C.byref(st)))
        return st

    def integrate_device(self, T_G_C, d_xyz: int, d_rgba: int, d_labels: int, n: int, freespace=False) -> KsFrameStats:
        """d_* are raw device addresses (e.g. torch.Tensor.data_ptr()); 0/None = NULL."""
        T = np.ascontiguousarray(T_G_C, dtype=np.float32)
        st = KsFrameStats()
        self._chk(lib().ks_integrate_points_device(self._h, _ptr(T), d_xyz or None, d_rgba or None, d_labels or None, n,
                                                   int(freespace), C.byref(st)))
        return st

    def block_indices(self) -> np.ndarray:
        n = C.c_size_t()
        self._chk(lib().ks_get_block_indices(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 3), dtype=np.int32)
        if n.value:
            self._chk(lib().ks_get_block_indices(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def updated_block_indices(self, reset=True) -> np.ndarray:
        n = C.c_size_t()
        self._chk(lib().ks_get_updated_block_indices(self._h, None, 0, C.byref(n), 0))
        out = np.zeros((n.value, 3), dtype=np.int32)
        self._chk(lib().ks_get_updated_block_indices(self._h, _ptr(out), n.value, C.byref(n), int(reset)))
        return out

    def download(self, indices=None):
        if indices is None:
            indices = self.block_indices()
        indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
        nv = self.vps ** 3
        t = np.zeros((len(indices), nv), dtype=TSDF_DTYPE)
        s = np.zeros((len(indices), nv), dtype=SEM_DTYPE)
        if len(indices):
            self._chk(lib().ks_download_blocks(self._h, _ptr(indices), len(indices), _ptr(t), _ptr(s)))
        return indices, t, s

    def upload(self, indices, tsdf=None, sem=None):
        """Seed / overwrite host-layout blocks in the GPU map (inverse of download)."""
        indices = np.ascontiguousarray(indices, dtype=np.int32).reshape(-1, 3)
        nv = self.vps ** 3
        tp = sp = None
        if tsdf is not None:
            tsdf = np.ascontiguousarray(tsdf, dtype=TSDF_DTYPE).reshape(len(indices), nv)
            tp = _ptr(tsdf)
        if sem is not None:
            sem = np.ascontiguousarray(sem, dtype=SEM_DTYPE).reshape(len(indices), nv)
            sp = _ptr(sem)
        self._chk(lib().ks_upload_blocks(self._h, _ptr(indices), len(indices), tp, sp))

    # ---- multi-GPU exchange primitives (used by kimera_semantics_amd.parallel) ----
    TILE_BYTES = 65536

    def tile_keys(self) -> np.ndarray:
        n = C.c_size_t()
        self._chk(lib().ks_get_tile_keys(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint64)
        if n.value:
            self._chk(lib().ks_get_tile_keys(self._h, _ptr(out), n.value, C.byref(n)))
        return out

    def export_tiles(self, slots: np.ndarray, d_payload: int):
        """Gathers the tiles at `slots` into the device buffer at address d_payload (len(slots) x 64 KiB)."""
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        self._chk(lib().ks_export_tiles_device(self._h, _ptr(s), len(s), d_payload or None))

    def merge_tiles(self, keys: np.ndarray, d_payload: int):
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        self._chk(lib().ks_merge_tiles_device(self._h, _ptr(k), len(k), d_payload or None))

    def download_updated_voxels(self):
        """Voxels written since the previous call: structured array (block [3] i32, linear u32, tsdf, sem)."""
        n, nr = C.c_size_t(), C.c_size_t()
        self._chk(lib().ks_count_updated_voxels(self._h, C.byref(n), C.byref(nr)))
        dt = np.dtype([("block", "<i4", (3,)), ("linear", "<u4"), ("tsdf", TSDF_DTYPE), ("sem", SEM_DTYPE)])
        rdt = np.dtype([("block", "<i4", (3,)), ("first", "<u4"), ("count", "<u4")])
        assert dt.itemsize == 120 and rdt.itemsize == 20
        out = np.zeros(n.value, dtype=dt)
        runs = np.zeros(nr.value, dtype=rdt)
        if n.value:
            m, mr = C.c_size_t(), C.c_size_t()
            self._chk(lib().ks_download_updated_voxels(self._h, _ptr(out), n.value, C.byref(m), _ptr(runs), nr.value, C.byref(mr)))
            assert m.value == n.value and mr.value == nr.value
            # every record belongs to exactly one run, and carries its run's block index
            assert int(runs["count"].sum()) == n.value
            for r in runs[:: max(1, len(runs) // 16)]:
                if r["count"]:
                    assert (out["block"][r["first"]:r["first"] + r["count"]] == r["block"]).all()
        return out

    def clear(self):
        self._chk(lib().ks_clear(self._h))

    def clear_voxels(self):
        """The map goes, the integrator's sets and counters stay (ks_clear_voxels)."""
        self._chk(lib().ks_clear_voxels(self._h))

    def reset_tiles(self, slots: np.ndarray):
        s = np.ascontiguousarray(slots, dtype=np.uint32)
        self._chk(lib().ks_reset_tiles(self._h, _ptr(s), len(s)))

    def reduce(self, rccl_comm, rank: int, world: int) -> dict:
        """ks_reduce: rccl_comm is an ncclComm_t (int / c_void_p) created with the same librccl."""
        st = KsReduceStats()
        self._chk(lib().ks_reduce(self._h, C.c_void_p(rccl_comm) if rccl_comm else None, rank, world, C.byref(st)))
        return {"tiles_sent": st.tiles_sent, "tiles_received": st.tiles_received, "tiles_local": st.tiles_local,
                "bytes_sent": st.bytes_sent}

    def debug_radix_sort(self, keys: np.ndarray, vals=None, end_bit=None):
        keys = np.ascontiguousarray(keys).copy()
        bits = keys.dtype.itemsize * 8
        vals = None if vals is None else np.ascontiguousarray(vals, dtype=np.uint32).copy()
        self._chk(lib().ks_debug_radix_sort(self._h, _ptr(keys), _ptr(vals), keys.shape[0], bits,
                                            bits if end_bit is None else end_bit))
        return keys, vals

    def synchronize(self):
        self._chk(lib().ks_synchronize(self._h))

    @property
    def stream(self) -> int:
        return lib().ks_stream(self._h) or 0

    def profile_enable(self, level=1):
        """0 off, 1 all stages, 2 sampled k_apply dispatches only (see ks_hip.h)."""
        self._chk(lib().ks_profile_enable(self._h, int(level)))

    def early_out_stats(self):
        """Exact early-out: dict(frames, rounds, fallbacks, event_driven, pipelined)."""
        out = (C.c_uint64 * 5)()
        self._chk(lib().ks_early_out_stats(self._h, out))
        return dict(frames=int(out[0]), rounds=int(out[1]), fallbacks=int(out[2]), event_driven=bool(out[3]), pipelined=bool(out[4]))

    def integrate_round_exact(self, marcher: "HipIntegrator", rccl_comm, rank: int, world: int, first_frame: int, T_G_C, xyz, rgba, labels,
                              freespace=False) -> dict:
        """self = the OWNER context of this rank; `marcher` casts this rank's frame of the round (ks_integrate_round_exact)."""
        T = np.ascontiguousarray(T_G_C, dtype=np.float32)
        n = 0 if xyz is None else len(xyz)
        x = None if xyz is None else np.ascontiguousarray(xyz, dtype=np.float32)
        c = None if rgba is None else np.ascontiguousarray(rgba, dtype=np.uint8)
        l = None if labels is None else np.ascontiguousarray(labels, dtype=np.uint8)
        st = KsRoundStats()
        self._chk(lib().ks_integrate_round_exact(marcher._h, self._h, rccl_comm, int(rank), int(world), int(first_frame), _ptr(T),
                                                 _ptr(x) if x is not None else None, _ptr(c) if c is not None else None,
                                                 _ptr(l) if l is not None else None, n, int(bool(freespace)), C.byref(st)))
        return dict(updates_marched=int(st.updates_marched), updates_applied=int(st.updates_applied), bytes_sent=int(st.bytes_sent),
                    origin_voxel_touched=bool(st.origin_voxel_touched), rays_cast=int(st.rays_cast))

    def update_stats(self):
        """Runs of more than 1024 updates: dict(walked, serial, chunks, replayed) — ks_update_stats."""
        out = (C.c_uint64 * 4)()
        self._chk(lib().ks_update_stats(self._h, out))
        return dict(walked=int(out[0]), serial=int(out[1]), chunks=int(out[2]), replayed=int(out[3]))

    def pipeline_shape(self):
        """dict(lag, slots, batch, march_streams): what ks_create made of ks_config.pipeline_frames."""
        out = (C.c_int32 * 4)()
        self._chk(lib().ks_pipeline_shape(self._h, out))
        return dict(lag=int(out[0]), slots=int(out[1]), batch=int(out[2]), march_streams=int(out[3]))

    def early_out_iterations(self):
        """(frames, fix-point iterations) of a KS_EARLY_OUT_EXACT context."""
        f, it = C.c_uint64(), C.c_uint64()
        self._chk(lib().ks_early_out_iterations(self._h, C.byref(f), C.byref(it)))
        return f.value, it.value

    def flush(self) -> KsFrameStats:
        """Finish the frames a pipelined context still holds; returns their statistics (summed)."""
        st = KsFrameStats()
        self._chk(lib().ks_flush(self._h, C.byref(st)))
        return st

    def profile(self, reset=False) -> dict:
        p = KsProfile()
        self._chk(lib().ks_profile_get(self._h, C.byref(p), int(reset)))
        return {"ms": {STAGES[i]: p.ms[i] for i in range(8)}, "launches": {STAGES[i]: p.launches[i] for i in range(8)},
                "frames": p.frames, "updates": p.updates, "points": p.points,
                "apply_kernel_ms": p.apply_kernel_ms, "apply_kernel_launches": p.apply_kernel_launches,
                "apply_kernel_updates": p.apply_kernel_updates, "host_ms": p.host_ms, "host_wait_ms": p.host_wait_ms}
