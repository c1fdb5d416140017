#!/usr/bin/env python3
"""Pipelined frame rate of one workload under environment toggles, without torch (device buffers through
libamdhip64 directly) so that a run costs seconds:  exp_pipe.py prepare | exp_pipe.py run [frames] | exp_pipe.py matrix.
Diagnostics only; needs a GPU."""
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

CACHE = "/tmp/ks_exp_frames.npz"
NF = 16


def prepare(name="C2"):
    import bench
    wl = bench.WORKLOADS[name]
    frames = bench.make_frames(wl, range(NF))
    d = {}
    for i, f in enumerate(frames):
        d[f"T{i}"], d[f"x{i}"], d[f"c{i}"], d[f"l{i}"] = f.T_G_C, f.xyz, f.rgba, f.labels
    np.savez(CACHE, name=name, **d)


def run(K=160):
    import bench
    from kimera_semantics_amd import binding as B
    z = np.load(CACHE)
    name = str(z["name"])
    wl = bench.WORKLOADS[name]
    hip = C.CDLL("/opt/rocm/lib/libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def up(a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), a.nbytes) == 0
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        return p.value

    fr = [(z[f"T{i}"], up(z[f"x{i}"]), up(z[f"c{i}"]), up(z[f"l{i}"]), len(z[f"l{i}"])) for i in range(NF)]
    cfg = B.default_config(max_tiles=1 << 16 if name.startswith("C4") else 1 << 13, max_points=wl["w"] * wl["h"],
                           pipeline_frames=int(os.environ.get("KS_EXP_PIPE", "4")), **bench.integ_cfg(wl))
    h = B.HipIntegrator(cfg)
    for i in range(NF):
        h.integrate_device(*fr[i])
    h.flush()
    h.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        h.integrate_device(*fr[i % NF])
    h.flush()
    h.synchronize()
    dt = time.perf_counter() - t0
    print(f"{1e3 * dt / K:.4f} ms/frame", flush=True)
    h.close()


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "matrix"
    if mode == "prepare":
        prepare(sys.argv[2] if len(sys.argv) > 2 else "C2")
    elif mode == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 160)
    else:
        if not os.path.exists(CACHE):
            prepare()
        for cfg in sys.argv[2:] or [""]:
            env = dict(os.environ)
            for kv in cfg.split(","):
                if kv:
                    k, v = kv.split("=")
                    env[k] = v
            r = subprocess.run([sys.executable, __file__, "run"], env=env, capture_output=True, text=True, timeout=120)
            print(f"{cfg or 'default':50s} {r.stdout.strip()} {r.stderr.strip()[-200:]}", flush=True)
