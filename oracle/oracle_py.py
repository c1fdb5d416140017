"""ctypes wrapper of the CPU oracle (oracle/libks_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under kimera_semantics_amd/ imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NUM_LABELS = 21

TSDF_DTYPE = np.dtype([("distance", "<f4"), ("weight", "<f4"), ("color", "u1", (4,))])
SEM_DTYPE = np.dtype([("label", "u1"), ("pad", "u1", (3,)), ("priors", "<f4", (NUM_LABELS,)),
                      ("color", "u1", (4,))])
assert TSDF_DTYPE.itemsize == 12 and SEM_DTYPE.itemsize == 92


class KoConfig(C.Structure):
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
    ]


class KoFrameStats(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("n_valid_points", C.c_uint64), ("n_rays_cast", C.c_uint64),
                ("n_voxel_updates", C.c_uint64), ("n_blocks_allocated", C.c_uint64)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libks_oracle.so")
    src = os.path.join(_HERE, "ks_oracle.cpp")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "libks_oracle.so"], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.ko_default_config.argtypes = [C.POINTER(KoConfig)]
        L.ko_create.argtypes = [C.POINTER(KoConfig), C.POINTER(C.c_void_p)]
        L.ko_create.restype = C.c_int
        L.ko_destroy.argtypes = [C.c_void_p]
        L.ko_last_error.argtypes = [C.c_void_p]
        L.ko_last_error.restype = C.c_char_p
        L.ko_integrate_points.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_int, C.POINTER(KoFrameStats)]
        L.ko_integrate_points.restype = C.c_int
        L.ko_num_blocks.argtypes = [C.c_void_p]
        L.ko_num_blocks.restype = C.c_size_t
        L.ko_num_semantic_blocks.argtypes = [C.c_void_p]
        L.ko_num_semantic_blocks.restype = C.c_size_t
        L.ko_get_block_indices.argtypes = [C.c_void_p, C.c_void_p]
        L.ko_get_semantic_block_indices.argtypes = [C.c_void_p, C.c_void_p]
        L.ko_get_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ko_get_block.restype = C.c_int
        L.ko_transform_point.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ko_grid_index_from_point.argtypes = [C.c_void_p, C.c_float, C.c_void_p]
        L.ko_cast_ray.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                  C.c_int, C.c_void_p, C.c_size_t]
        L.ko_cast_ray.restype = C.c_size_t
        L.ko_log_likelihood.argtypes = [C.c_float, C.c_void_p]
        L.ko_long_index_hash.argtypes = [C.c_void_p]
        L.ko_long_index_hash.restype = C.c_uint32
        L.ko_mixed_index.argtypes = [C.c_size_t, C.c_size_t, C.c_int]
        L.ko_mixed_index.restype = C.c_size_t
        L.ko_mixed_chains.argtypes = [C.c_size_t, C.c_int]
        L.ko_mixed_chains.restype = C.c_uint32
        L.ko_update_tsdf_voxel.argtypes = [C.POINTER(KoConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ko_blend_two_colors.argtypes = [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p]
        L.ko_rainbow_color_map.argtypes = [C.c_double, C.c_void_p]
        _lib = L
    return _lib


def default_config(**overrides) -> KoConfig:
    cfg = KoConfig()
    lib().ko_default_config(C.byref(cfg))
    apply_overrides(cfg, **overrides)
    return cfg


def apply_overrides(cfg, **overrides):
    """Shared by the oracle and the HIP binding: both config structs use the same field
    names for the shared reference knobs."""
    for k, v in overrides.items():
        if k == "dynamic_labels":
            cfg.n_dynamic_labels = len(v)
            for i, lab in enumerate(v):
                cfg.dynamic_labels[i] = int(lab)
        elif k == "label_rgba":
            arr = np.asarray(v, dtype=np.uint8).reshape(256, 4)
            C.memmove(cfg.label_rgba, arr.ctypes.data, 1024)
        else:
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
    return cfg


def _ptr(a):
    return None if a is None else a.ctypes.data


class Oracle:
    def __init__(self, cfg: KoConfig):
        self.cfg = cfg
        self.vps = cfg.voxels_per_side
        self._h = C.c_void_p()
        rc = lib().ko_create(C.byref(cfg), C.byref(self._h))
        if rc != 0:
            raise ValueError(f"ko_create failed: {rc}")

    def close(self):
        if self._h:
            lib().ko_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def integrate(self, T_G_C, xyz, rgba, labels, freespace=False) -> KoFrameStats:
        T = np.ascontiguousarray(T_G_C, dtype=np.float32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        labels = np.ascontiguousarray(labels, dtype=np.uint8)
        rgba = None if rgba is None else np.ascontiguousarray(rgba, dtype=np.uint8)
        st = KoFrameStats()
        rc = lib().ko_integrate_points(self._h, _ptr(T), _ptr(xyz), _ptr(rgba), _ptr(labels),
                                       xyz.shape[0], int(freespace), C.byref(st))
        if rc != 0:
            raise RuntimeError(f"ko_integrate_points rc={rc}: {lib().ko_last_error(self._h).decode()}")
        return st

    def block_indices(self) -> np.ndarray:
        n = lib().ko_num_blocks(self._h)
        out = np.zeros((n, 3), dtype=np.int32)
        if n:
            lib().ko_get_block_indices(self._h, _ptr(out))
        return out

    def semantic_block_indices(self) -> np.ndarray:
        n = lib().ko_num_semantic_blocks(self._h)
        out = np.zeros((n, 3), dtype=np.int32)
        if n:
            lib().ko_get_semantic_block_indices(self._h, _ptr(out))
        return out

    def get_block(self, idx):
        nv = self.vps ** 3
        t = np.zeros(nv, dtype=TSDF_DTYPE)
        s = np.zeros(nv, dtype=SEM_DTYPE)
        i = np.ascontiguousarray(idx, dtype=np.int32)
        absent = lib().ko_get_block(self._h, _ptr(i), _ptr(t), _ptr(s))
        return t, s, bool(absent)

    def download(self, indices=None):
        """Returns (indices [n,3], tsdf [n, vps^3], sem [n, vps^3])."""
        if indices is None:
            indices = self.block_indices()
        nv = self.vps ** 3
        t = np.zeros((len(indices), nv), dtype=TSDF_DTYPE)
        s = np.zeros((len(indices), nv), dtype=SEM_DTYPE)
        for k, idx in enumerate(indices):
            i = np.ascontiguousarray(idx, dtype=np.int32)
            lib().ko_get_block(self._h, _ptr(i), _ptr(t[k]), _ptr(s[k]))
        return indices, t, s


# ---- pure-function helpers for KATs ----
def transform_point(T, p):
    T = np.ascontiguousarray(T, dtype=np.float32)
    p = np.ascontiguousarray(p, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().ko_transform_point(_ptr(T), _ptr(p), _ptr(out))
    return out


def grid_index_from_point(p, inv):
    p = np.ascontiguousarray(p, dtype=np.float32)
    out = np.zeros(3, dtype=np.int64)
    lib().ko_grid_index_from_point(_ptr(p), float(inv), _ptr(out))
    return out


def cast_ray(origin, point_G, is_clearing=False, carving=True, max_ray_length_m=5.0, voxel_size_inv=20.0,
             truncation=0.2, cast_from_origin=True, cap=100000):
    o = np.ascontiguousarray(origin, dtype=np.float32)
    p = np.ascontiguousarray(point_G, dtype=np.float32)
    out = np.zeros((cap, 3), dtype=np.int64)
    n = lib().ko_cast_ray(_ptr(o), _ptr(p), int(is_clearing), int(carving), max_ray_length_m, voxel_size_inv,
                          truncation, int(cast_from_origin), _ptr(out), cap)
    assert n <= cap
    return out[:n].copy()


def log_likelihood(p_match):
    out = np.zeros((NUM_LABELS, NUM_LABELS), dtype=np.float32)
    lib().ko_log_likelihood(float(p_match), _ptr(out))
    return out


def long_index_hash(idx):
    i = np.ascontiguousarray(idx, dtype=np.int64)
    return int(lib().ko_long_index_hash(_ptr(i)))


ORDER_MIXED, ORDER_SORTED, ORDER_MIXED_1024_GROUPS = 0, 1, 2   # ks_oracle.h KO_ORDER_*


def mixed_index(s, n, mode=ORDER_MIXED):
    return int(lib().ko_mixed_index(s, n, mode))


def mixed_chains(n, mode=ORDER_MIXED):
    return int(lib().ko_mixed_chains(n, mode))


def update_tsdf_voxel(cfg, origin, point_G, voxel_idx, rgba, weight, distance, voxel_weight, voxel_rgba):
    o = np.ascontiguousarray(origin, dtype=np.float32)
    p = np.ascontiguousarray(point_G, dtype=np.float32)
    vi = np.ascontiguousarray(voxel_idx, dtype=np.int64)
    c = np.ascontiguousarray(rgba, dtype=np.uint8)
    d = np.array([distance], dtype=np.float32)
    w = np.array([voxel_weight], dtype=np.float32)
    vc = np.ascontiguousarray(voxel_rgba, dtype=np.uint8).copy()
    lib().ko_update_tsdf_voxel(C.byref(cfg), _ptr(o), _ptr(p), _ptr(vi), _ptr(c), float(weight), _ptr(d), _ptr(w),
                               _ptr(vc))
    return float(d[0]), float(w[0]), vc


def blend_two_colors(c1, w1, c2, w2):
    a = np.ascontiguousarray(c1, dtype=np.uint8)
    b = np.ascontiguousarray(c2, dtype=np.uint8)
    out = np.zeros(4, dtype=np.uint8)
    lib().ko_blend_two_colors(_ptr(a), float(w1), _ptr(b), float(w2), _ptr(out))
    return out


def rainbow_color_map(h):
    out = np.zeros(4, dtype=np.uint8)
    lib().ko_rainbow_color_map(float(h), _ptr(out))
    return out
