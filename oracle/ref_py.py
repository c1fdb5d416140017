"""ctypes wrapper of oracle/_ref/libks_ref.so — the REAL Kimera-Semantics integrator sources
(/root/reference/kimera_semantics/src/*.cpp) compiled against the header shims.  Test
infrastructure only.  The library is built in this container (oracle/Makefile `ref` target)
and travels to the GPU box as a prebuilt file; /root/reference is never read at run time."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .oracle_py import SEM_DTYPE, TSDF_DTYPE

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libks_ref.so")


def available() -> bool:
    return os.path.exists(LIB_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        L.kr_create.restype = C.c_void_p
        L.kr_create.argtypes = [C.c_char_p, C.c_float, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
        L.kr_destroy.argtypes = [C.c_void_p]
        L.kr_clear_tsdf_layer.argtypes = [C.c_void_p]
        L.kr_integrate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.kr_num_blocks.argtypes = [C.c_void_p]
        L.kr_num_blocks.restype = C.c_size_t
        L.kr_num_semantic_blocks.argtypes = [C.c_void_p]
        L.kr_num_semantic_blocks.restype = C.c_size_t
        L.kr_block_indices.argtypes = [C.c_void_p, C.c_void_p]
        L.kr_get_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kr_get_block.restype = C.c_int
        L.kr_set_mixed_order_form.argtypes = [C.c_int]
        L.kr_get_mixed_order_form.restype = C.c_int
        L.kr_mixed_sequence.argtypes = [C.c_size_t, C.c_void_p]
        _lib = L
    return _lib


def write_label_csv(path: str, label_rgba: np.ndarray, n_labels: int = 21):
    with open(path, "w") as fh:
        fh.write("name,red,green,blue,alpha,id\n")
        for i in range(n_labels):
            r, g, b, a = (int(x) for x in label_rgba[i])
            fh.write(f"label{i},{r},{g},{b},{a},{i}\n")


def set_mixed_order_form(form: int):
    """0 = upstream Voxblox as published (the default), 1 = 1024 groups of N/1024 (what this repository assumed until
    round 4): which permutation the shim's MixedThreadSafeIndex behind the real Kimera sources produces.  Process-wide."""
    lib().kr_set_mixed_order_form(int(form))


def mixed_sequence(n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint64)
    lib().kr_mixed_sequence(n, out.ctypes.data)
    return out


class Reference:
    """order_mode: "mixed" | "sorted" | "mixed_1024_groups" (= "mixed" with the shim switched to form 1 for the lifetime
    of every integrate call of this object)."""

    def __init__(self, method: str, label_csv: str, voxel_size=0.05, vps=16, truncation=0.2, max_ray=5.0, p_match=0.8,
                 color_mode=1, dynamic_labels=(20,), threads=1, max_consecutive_ray_collisions=2, order_mode="mixed",
                 **extra):
        self.vps = vps
        self._form = 1 if order_mode == "mixed_1024_groups" else 0
        if order_mode == "mixed_1024_groups":
            order_mode = "mixed"
        dyn = np.array(list(dynamic_labels), dtype=np.uint8)
        self._h = lib().kr_create(method.encode(), voxel_size, vps, truncation, max_ray, p_match, color_mode,
                                  dyn.ctypes.data if len(dyn) else None, len(dyn), threads,
                                  max_consecutive_ray_collisions, order_mode.encode(), label_csv.encode(),
                                  ";".join(f"{k}={float(v)}" for k, v in extra.items()).encode())

    def clear_tsdf_layer(self):
        """vxb::TsdfServer::clear(): the TSDF blocks go, the semantic layer and the integrator stay."""
        lib().kr_clear_tsdf_layer(self._h)

    def close(self):
        if self._h:
            lib().kr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def integrate(self, T_G_C, xyz, rgba, freespace=False):
        T = np.ascontiguousarray(T_G_C, dtype=np.float32)
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        lib().kr_set_mixed_order_form(self._form)
        lib().kr_integrate(self._h, T.ctypes.data, xyz.ctypes.data, rgba.ctypes.data, xyz.shape[0], int(freespace))

    def block_indices(self) -> np.ndarray:
        n = lib().kr_num_blocks(self._h)
        out = np.zeros((n, 3), dtype=np.int32)
        if n:
            lib().kr_block_indices(self._h, out.ctypes.data)
        order = np.lexsort((out[:, 2], out[:, 1], out[:, 0]))
        return out[order]

    def n_semantic_blocks(self) -> int:
        return lib().kr_num_semantic_blocks(self._h)

    def download(self, indices=None):
        if indices is None:
            indices = self.block_indices()
        nv = self.vps ** 3
        t = np.zeros((len(indices), nv), dtype=TSDF_DTYPE)
        s = np.zeros((len(indices), nv), dtype=SEM_DTYPE)
        for k, idx in enumerate(indices):
            i = np.ascontiguousarray(idx, dtype=np.int32)
            rc = lib().kr_get_block(self._h, i.ctypes.data, t[k].ctypes.data, s[k].ctypes.data)
            assert rc == 0
        return indices, t, s
