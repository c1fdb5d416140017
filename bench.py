#!/usr/bin/env python3
"""bench.py — semantic TSDF integration throughput on MI355X.

Workload (BASELINE.json configs[1], the config the metric is quoted on): replay of a
640x480 depth+label trajectory ("kimera_semantics_demo.bag" stand-in, synthetic — the bag
is not in the reference repository), `fast` integrator with the reference's default
parameters (5 cm voxels, 5 m rays, truncation 4 voxels, early-out after 2 consecutive
already-observed voxels, p=0.8, dynamic label 20).  One step = one frame integrated into
the GPU-resident map through the C ABI, inputs already resident in HBM.

Prints ONE JSON line (rank 0).  value = voxel updates/s over the whole job (all ranks),
where a voxel update is one (ray, voxel) pair for which the reference runs
updateTsdfVoxel + updateSemanticVoxel (semantic_tsdf_integrator_fast.cpp:128-140).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_UPDATE = 208         # R+W of TsdfVoxel (12 B) + SemanticVoxel (92 B), SURVEY.md §8d
BYTES_PER_POINT = 17           # xyz 12 B + rgba 4 B + label 1 B
try:
    METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
except Exception:
    METRIC = "Mvoxel-updates/s + frames/s, 640x480 @5cm voxels, 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--method", default="fast", choices=["fast", "merged"])
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=12)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="ks_config.pipeline_frames=0: every call completes its own frame (host wait not overlapped)")
    return ap.parse_args()


def common_cfg(method):
    from kimera_semantics_amd import synth
    return dict(method=0 if method == "fast" else 1, voxel_size=0.05, voxels_per_side=16,
                truncation_distance=0.2, max_ray_length_m=5.0, semantic_measurement_probability=0.8,
                dynamic_labels=[20], label_rgba=synth.default_label_colors())


def cpu_baseline(args, frames):
    """CPU baseline on this host's cores, on a bounded sample of the same workload.
    kind "reference": oracle/_ref/libks_ref.so = the REAL Kimera-Semantics integrator sources
    compiled (in the build container) against the Voxblox header shims; it has no update counter,
    so its voxel-update count is the one the bit-identical port (oracle/) reports for the same
    frames in single-thread order.  kind "port" (the oracle itself) when the prebuilt library is
    not there.  The reference defaults to integrator_threads = hardware_concurrency(); on many-core
    hosts its per-voxel mutexes make that slower than a few threads, so several thread counts are
    tried and the best one is the reported baseline."""
    import tempfile
    from oracle import oracle_py as O
    from oracle import ref_py as R
    from kimera_semantics_amd import synth
    cores = os.cpu_count() or 1
    n = min(args.cpu_frames, len(frames))
    thread_counts = sorted({1, min(8, cores), cores})
    # the port: timing + the update counts
    port = {}
    upd_serial = None
    for threads in thread_counts:
        nf = n if threads > 1 else max(1, n // 2)
        o = O.Oracle(O.default_config(integrator_threads=threads, **common_cfg(args.method)))
        upd = 0
        per_frame = []
        t0 = time.perf_counter()
        for f in frames[:nf]:
            st = o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
            upd += st.n_voxel_updates
            per_frame.append(st.n_voxel_updates)
        dt = time.perf_counter() - t0
        port[threads] = (upd / dt / 1e6, nf / dt, nf)
        if threads == 1:
            upd_serial = per_frame
        o.close()
    if upd_serial is None or len(upd_serial) < n:
        # single-thread counts for every sampled frame (the reference is credited with these)
        o = O.Oracle(O.default_config(integrator_threads=1, **common_cfg(args.method)))
        upd_serial = [o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates for f in frames[:n]]
        o.close()
    kind, tried = "port", port
    if R.available():
        try:
            tmp = tempfile.mkdtemp(prefix="ks_bench_")
            csv = os.path.join(tmp, "labels.csv")
            R.write_label_csv(csv, synth.default_label_colors())
            ref = {}
            for threads in thread_counts:
                nf = n if threads > 1 else max(1, n // 2)
                r = R.Reference(args.method, csv, voxel_size=0.05, vps=16, truncation=0.2, max_ray=5.0, p_match=0.8,
                                color_mode=1, dynamic_labels=(20,), threads=threads)
                t0 = time.perf_counter()
                for f in frames[:nf]:
                    r.integrate(f.T_G_C, f.xyz, f.rgba)
                dt = time.perf_counter() - t0
                ref[threads] = (sum(upd_serial[:nf]) / dt / 1e6, nf / dt, nf)
                r.close()
            kind, tried = "reference", ref
        except Exception as e:  # a stale or missing prebuilt library must not take the bench down
            sys.stderr.write(f"cpu_baseline: reference library unusable ({e}); using the port\n")
    best = max(tried, key=lambda t: tried[t][0])
    what = ("the real Kimera-Semantics integrator sources (oracle/_ref, Voxblox half restated)" if kind == "reference"
            else "the CPU oracle (restatement, bit-identical to the real reference sources for the Kimera half)")
    return {"value": round(tried[best][0], 4), "unit": "Mvoxel-updates/s", "cores": best, "kind": kind,
            "frames_per_s": round(tried[best][1], 3), "host_cores": cores,
            "by_threads": {str(t): round(v[0], 4) for t, v in tried.items()},
            "port_by_threads": {str(t): round(v[0], 4) for t, v in port.items()},
            "sample": f"first {tried[best][2]} frames of the same trajectory through {what}, "
                      f"'mixed' order, reference defaults; best of integrator_threads in {sorted(tried)}"
                      + ("; updates counted by the port in single-thread order" if kind == "reference" else "")}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from kimera_semantics_amd import binding as B
    from kimera_semantics_amd import synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    K, W = args.steps, args.warmup
    # frame-sharded: rank r integrates trajectory frames r, r+world, ... (weak scaling: K+W frames per GPU)
    scene = synth.make_scene("room")
    frames = []
    for k in range(K + W):
        gk = rank + world * k
        frames.append(synth.render_frame(scene, synth.trajectory_pose(gk), args.width, args.height, seed=gk))
    d_frames = []
    for f in frames:
        d_frames.append((torch.from_numpy(f.xyz).to(dev), torch.from_numpy(f.rgba).to(dev),
                         torch.from_numpy(f.labels).to(dev)))
    # bag replay = a stream of frames: frame pipelining on (the host's one wait per frame overlaps
    # the next frame's GPU work; results are identical, see tests/test_parity_gpu.py)
    cfg = B.default_config(device_id=local_rank, max_tiles=1 << 13, max_points=args.width * args.height,
                           pipeline_frames=0 if args.no_pipeline else 2, **common_cfg(args.method))
    integ = B.HipIntegrator(cfg)

    def step_on(h, i):
        x, c, l = d_frames[i]
        return h.integrate_device(frames[i].T_G_C, x.data_ptr(), c.data_ptr(), l.data_ptr(), x.shape[0])

    def step(i):
        return step_on(integ, i)

    for i in range(W):
        step(i)
    integ.flush()             # completes the warm-up frames AND hands their statistics over (discarded):
    integ.synchronize()       # nothing is pending or owed at t0
    torch.cuda.synchronize()
    if world > 1:
        from kimera_semantics_amd import parallel as PAR
        PAR.warm_up(dev)   # RCCL connects peers lazily: not part of the steady state being timed
        dist.barrier()
    # level 2: only the k_apply dispatch of every 4th frame carries HIP events (per-stage events
    # would put ~50 us of stream bubbles into every timed frame)
    integ.profile_enable(2)
    integ.profile(reset=True)
    t0 = time.perf_counter()
    updates = 0
    points = 0
    for i in range(W, W + K):
        st = step(i)          # pipelined: statistics of the frame(s) completed by this call
        updates += st.n_voxel_updates
        points += st.n_points
    st = integ.flush()        # the tails of the last two frames, inside the timed region
    updates += st.n_voxel_updates
    points += st.n_points
    integ.synchronize()
    reduce_stats = None
    if world > 1:
        # the one exchange step of the frame-sharded path: per-rank partial maps -> owner-sharded
        # global map (all-to-all of touched tiles over RCCL/xGMI + deterministic owner merge)
        from kimera_semantics_amd import parallel as PAR
        reduce_stats = PAR.reduce_maps(PAR.HipTileStore(integ, dev))
        integ.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = integ.profile(reset=True)
    # per-stage breakdown: separate untimed pass over the last frames with events around every stage
    integ.profile_enable(1)
    for i in range(max(W, W + K - 10), W + K):
        step(i)
    integ.flush()
    stage_prof = integ.profile()
    integ.profile_enable(0)
    # k_apply ALONE on the GPU (no other stage overlapping it): a second, unpipelined context over
    # the same frames; reported next to the timed-region figure as roofline.isolated
    iso = None
    if rank == 0 and world == 1 and not args.no_pipeline:
        cfg0 = B.default_config(device_id=local_rank, max_tiles=1 << 13, max_points=args.width * args.height,
                                pipeline_frames=0, **common_cfg(args.method))
        solo = B.HipIntegrator(cfg0)
        for i in range(min(W + K, 30)):
            if i == min(W + K, 30) - 10:
                solo.synchronize()
                solo.profile_enable(1)
            step_on(solo, i)
        iso = solo.profile()
        solo.close()

    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        u = torch.tensor([updates, points], device=dev, dtype=torch.float64)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        updates_all, points_all = int(u[0].item()), int(u[1].item())
    else:
        updates_all, points_all = updates, points

    if rank == 0:
        # roofline of the dominant kernel (k_apply: the per-voxel TSDF + semantic RMW), from
        # HIP events recorded on the integrator's own stream inside the timed region.
        # k_apply dispatch begin->end (events attached to the dispatch itself, on the integrator's stream)
        apply_ms = prof["apply_kernel_ms"] / max(1, prof["apply_kernel_launches"])
        upd_per_launch = prof["apply_kernel_updates"] / max(1, prof["apply_kernel_launches"])
        pts_per_launch = prof["points"] / max(1, prof["frames"])
        alg_bytes = BYTES_PER_UPDATE * upd_per_launch
        achieved = alg_bytes / (apply_ms * 1e-3) / 1e9 if apply_ms > 0 else 0.0
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_apply.json")
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        stage_ms = {k: round(v / max(1, stage_prof["frames"]), 4) for k, v in stage_prof["ms"].items()}
        whole_frame_alg = (BYTES_PER_UPDATE * updates + BYTES_PER_POINT * points) / K
        isolated = None
        if iso and iso["apply_kernel_launches"]:
            i_ms = iso["apply_kernel_ms"] / iso["apply_kernel_launches"]
            i_bytes = BYTES_PER_UPDATE * iso["apply_kernel_updates"] / iso["apply_kernel_launches"]
            i_gbs = i_bytes / (i_ms * 1e-3) / 1e9
            isolated = {"achieved": round(i_gbs, 2), "frac": round(i_gbs / HBM_PEAK_GBS, 5),
                        "avg_launch_ms": round(i_ms, 5), "launches": iso["apply_kernel_launches"],
                        "note": "same kernel with nothing else on the GPU (unpipelined context, untimed pass)"}
        out = {
            "metric": METRIC,
            "value": round(updates_all / dt / 1e6, 3),
            "unit": "Mvoxel-updates/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "frames_per_s": round(world * K / dt, 2),
            "config": {"workload": f"bag-replay stand-in: {args.width}x{args.height} depth+label trajectory, "
                                   f"'{args.method}' integrator, 5 cm voxels, 5 m rays, trunc 0.2 m, p=0.8",
                       "frames_per_gpu": K, "pipeline_frames": 0 if args.no_pipeline else 2, "points_per_frame": int(points_all / max(1, world * K)),
                       "updates_per_frame": int(updates_all / max(1, world * K)),
                       "parallelism": f"frame-sharded x{world}" + (
                           " + one all-to-all tile reduce to hash-owners at the end (inside the timed region)"
                           if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": "k_apply", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(alg_bytes), "avg_launch_ms": round(apply_ms, 5),
                         "timed_launches": prof["apply_kernel_launches"],
                         "overlapped": not args.no_pipeline, "isolated": isolated,
                         "whole_frame_frac": round(whole_frame_alg / (dt / K) / 1e9 / HBM_PEAK_GBS, 5)},
            "stage_ms_per_frame": stage_ms,
            "host_ms_per_frame": {"in_call": round(prof["host_ms"] / max(1, prof["frames"]), 4),
                                  "of_which_waiting_for_snapshot": round(prof["host_wait_ms"] / max(1, prof["frames"]), 4)},
        }
        if reduce_stats is not None:
            out["reduce"] = reduce_stats
        if not args.no_cpu_baseline and world == 1:   # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(args, frames[W:])
        print(json.dumps(out), flush=True)
    integ.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
