#!/usr/bin/env python3
"""bench.py — semantic TSDF integration throughput on MI355X.

Workload (BASELINE.json configs[1], the config the metric is quoted on): replay of a
640x480 depth+label trajectory ("kimera_semantics_demo.bag" stand-in, synthetic — the bag
is not in the reference repository), `fast` integrator with the reference's default
parameters (5 cm voxels, 5 m rays, truncation 4 voxels, early-out after 2 consecutive
already-observed voxels, p=0.8, dynamic label 20) in the library's default mode: the map the
reference produces at integrator_threads = 1, bit for bit (the serial early-out reproduced by an
event-driven fix point on the device, csrc/ks_k_exact.h).  One step = one frame integrated into
the GPU-resident map through the C ABI, inputs already resident in HBM.

Timing.  The trajectory is a ring of EXACTLY K distinct frames.  Every context first integrates >= PRIME + W untimed
frames (whole turns of the ring: one frame per frame slot and pipeline stage — per-slot graph capture and buffer growth
happen there), then R >= 9 timed regions, each ONE turn of the ring = the same K frames in the same order, every region
bracketed by barrier + synchronize.  The early-out sets are per frame and no update count depends on the map, so every
region does exactly the same work: the spread over the regions measures the code and the machine, not the trajectory.
The line reports the MEDIAN region (value, ms_per_step) and the spread.  All 640x480 sub-records replay the same ring.

Prints ONE short JSON line (rank 0, < 4 KB); the full record (every sub-record, stage table) goes to
profiles/bench_full_r05.json.
  value = voxel updates/s over the whole job, where a voxel update is one (ray, voxel) pair for which
          the reference runs updateTsdfVoxel + updateSemanticVoxel (semantic_tsdf_integrator_fast.cpp:128-140)
          and N_updates is the count the SERIAL REFERENCE ORDER (CPU oracle, one thread) gives for the timed
          frames — in the default mode the GPU performs exactly those.
  roofline = the whole frame against the HBM roofline (lead figure), every stage's share of the frame, and
          the per-voxel update kernel (k_apply) on its own, all from HIP events of this run; traffic = HBM
          bytes per k_apply launch from the committed PMC pass of this same command (profiles/), or null.
  early_out_fidelity = the benched mode against the serial reference order, MEASURED in this run on the first timed
          frames (outside the timed regions): bit-exact, or the touched-voxel Jaccard / label agreement if not.
  secondary (full record; the line carries value / ms / frac of each) = the ordered-phase schedule alone
          (C2-ordered-phases: the throughput option, not the reference's map), the unpipelined context, C3 (`merged`,
          reference bundle order), C4 (1280x720, 2 cm, 10 m; `fast` in both modes and `merged`), the host-pointer entry
          (H2D inside the call: SURVEY.md §8d's frames/s), the unmodified-server adapter path.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the HIP runtime's hardware queues (default 4; streams that share one run one after the other): ks_create asks for 8, but
# the setting only takes effect before the process first touches the runtime — here, before torch does.  Scheduling only.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_UPDATE = 208         # R+W of TsdfVoxel (12 B) + SemanticVoxel (92 B), SURVEY.md §8d
BYTES_PER_POINT = 17           # xyz 12 B + rgba 4 B + label 1 B
PRIME = 20                     # untimed frames per context before the warm-up: >= frame slots (12) + pipeline_frames (8); with 12 or 16 calls of lag (24 slots) the whole untimed turns of the ring before t0 cover the rest
MIN_REPEATS = 5
MIN_REPEATS_PRIMARY = 9        # timed regions of the headline: the stretches of the trajectory differ by +-30 % (fix-point rounds), the median of nine moves less between runs than the median of five
MIN_TIMED_FRAMES = 100
C4_STEPS = 12                  # frames per C4 region = distinct C4 frames (ring), 1280x720 each
try:
    METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
except Exception:
    METRIC = "Mvoxel-updates/s + frames/s, 640x480 @5cm voxels, 1/2/4/8 MI355X"

WORKLOADS = {
    "C2": dict(scene="room", w=640, h=480, hfov=90.0, voxel=0.05, max_ray=5.0, radius=1.5, method="fast"),
    "C3": dict(scene="room", w=640, h=480, hfov=90.0, voxel=0.05, max_ray=5.0, radius=1.5, method="merged"),
    "C4-fast": dict(scene="hall", w=1280, h=720, hfov=75.0, voxel=0.02, max_ray=10.0, radius=3.0, method="fast"),
    "C4-merged": dict(scene="hall", w=1280, h=720, hfov=75.0, voxel=0.02, max_ray=10.0, radius=3.0, method="merged"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--method", default="fast", choices=["fast", "merged"])
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=10)
    ap.add_argument("--no-secondary", action="store_true", help="skip the sub-records")
    ap.add_argument("--only-secondary", default="", help="comma list of sub-records to run (default: all)")
    ap.add_argument("--no-oracle-count", action="store_true",
                    help="value falls back to the GPU's own update count (marked in the output)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="ks_config.pipeline_frames=0: every call completes its own frame (host wait not overlapped)")
    return ap.parse_args()


def integ_cfg(wl, method=None, **extra):
    from kimera_semantics_amd import synth
    method = method or wl["method"]
    kw = dict(method=0 if method == "fast" else 1, voxel_size=wl["voxel"], voxels_per_side=16,
              truncation_distance=4 * wl["voxel"], max_ray_length_m=wl["max_ray"], semantic_measurement_probability=0.8,
              dynamic_labels=[20], label_rgba=synth.default_label_colors(),
              # experiments only (the default, 0, is the library's default schedule)
              early_out_phase_growth=int(os.environ.get("KS_BENCH_GROWTH", "0")))
    kw.update(extra)
    return kw


def make_frames(wl, indices):
    from kimera_semantics_amd import synth
    scene = synth.make_scene(wl["scene"])
    return [synth.render_frame(scene, synth.trajectory_pose(k, radius=wl["radius"]), wl["w"], wl["h"], hfov_deg=wl["hfov"], seed=k)
            for k in indices]


def oracle_counts(wl, frames):
    """Voxel updates per frame in the SERIAL REFERENCE ORDER (CPU oracle, one thread, reference defaults).
    Neither integrator's update count depends on the map, and `fast`'s early-out sets are per frame, so the
    counts of the timed frames do not need the frames before them."""
    from oracle import oracle_py as O
    o = O.Oracle(O.default_config(integrator_threads=1, **integ_cfg(wl, early_out_phase_growth=0)))
    out = [int(o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates) for f in frames]
    o.close()
    return out


def median_spread(xs):
    med = statistics.median(xs)
    return med, (max(xs) - min(xs)) / med if med else 0.0


def cpu_baseline(args, wl, frames, upd_serial):
    """CPU baseline on this host's cores, on a bounded sample of the same workload: for every thread count one
    warm-up pass and 5 timed passes over the sample (a fresh integrator each), median reported.
    kind "reference": oracle/_ref/libks_ref.so = the REAL Kimera-Semantics integrator sources compiled (in the
    build container) against the Voxblox header shims; it has no update counter, so its voxel-update count is the
    one the bit-identical port (oracle/) reports for the same frames in single-thread order.  kind "port" (the
    oracle itself) when the prebuilt library is not there.  The reference defaults to integrator_threads =
    hardware_concurrency(); on many-core hosts its per-voxel mutexes make that slower than a few threads, so
    several thread counts are tried and the best one is the reported baseline."""
    import tempfile
    from oracle import oracle_py as O
    from oracle import ref_py as R
    from kimera_semantics_amd import synth
    cores = os.cpu_count() or 1
    n = min(args.cpu_frames, len(frames))
    sample, upd = frames[:n], sum(upd_serial[:n])
    thread_counts = sorted({1, min(8, cores), cores})
    passes = 1 + 5

    def timed(make, run):
        ts = []
        for p in range(passes):
            inst = make()
            t0 = time.perf_counter()
            for f in sample:
                run(inst, f)
            ts.append(time.perf_counter() - t0)
            inst.close()
        return statistics.median(ts[1:]), (max(ts[1:]) - min(ts[1:])) / statistics.median(ts[1:])

    kind, tried = "port", {}
    if R.available():
        try:
            tmp = tempfile.mkdtemp(prefix="ks_bench_")
            csv = os.path.join(tmp, "labels.csv")
            R.write_label_csv(csv, synth.default_label_colors())
            for threads in thread_counts:
                dt, spread = timed(lambda: R.Reference(wl["method"], csv, voxel_size=wl["voxel"], vps=16, truncation=4 * wl["voxel"],
                                                       max_ray=wl["max_ray"], p_match=0.8, color_mode=1, dynamic_labels=(20,), threads=threads),
                                   lambda r, f: r.integrate(f.T_G_C, f.xyz, f.rgba))
                tried[threads] = (upd / dt / 1e6, n / dt, spread)
            kind = "reference"
        except Exception as e:  # a stale or missing prebuilt library must not take the bench down
            sys.stderr.write(f"cpu_baseline: reference library unusable ({e}); using the port\n")
            tried = {}
    if not tried:
        for threads in thread_counts:
            dt, spread = timed(lambda: O.Oracle(O.default_config(integrator_threads=threads, **integ_cfg(wl, early_out_phase_growth=0))),
                               lambda o, f: o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels))
            tried[threads] = (upd / dt / 1e6, n / dt, spread)
    best = max(tried, key=lambda t: tried[t][0])
    what = ("the real Kimera-Semantics integrator sources (oracle/_ref, Voxblox half restated)" if kind == "reference"
            else "the CPU oracle (restatement, bit-identical to the real reference sources for the Kimera half)")
    return {"value": round(tried[best][0], 4), "unit": "Mvoxel-updates/s", "cores": best, "kind": kind,
            "frames_per_s": round(tried[best][1], 3), "host_cores": cores, "spread": round(tried[best][2], 4),
            "by_threads": {str(t): round(v[0], 4) for t, v in tried.items()},
            # the reference's own default is integrator_threads = hardware_concurrency() (all cores); the best count is the reported baseline
            "reference_default_all_cores_value": round(tried[cores][0], 4),
            "sample": f"first {n} timed frames of the same trajectory through {what}, 'mixed' order, reference defaults; "
                      f"per thread count 1 warm-up + 5 timed passes (fresh integrator each), median; best of "
                      f"integrator_threads in {sorted(tried)}"
                      + ("; updates counted by the port in single-thread order" if kind == "reference" else "")}


class FrameRing:
    """frames[i] for any i: the distinct frames replayed cyclically (device copies made once)."""

    def __init__(self, frames, torch=None, dev=None):
        self.frames = frames
        self.dev_frames = None
        if torch is not None:
            self.dev_frames = [(torch.from_numpy(f.xyz).to(dev), torch.from_numpy(f.rgba).to(dev), torch.from_numpy(f.labels).to(dev))
                               for f in frames]

    def __len__(self):
        return len(self.frames)

    def host(self, i):
        return self.frames[i % len(self.frames)]

    def dev(self, i):
        return self.dev_frames[i % len(self.frames)]


def measure(B, torch, dist, dev, wl, ring, W, K, R, pipeline, max_tiles, world, reduce_fn=None, entry="device", prime=None, **cfg_extra):
    """Integrates the ring as a stream: PRIME + W untimed frames, then R regions of K timed frames.  Returns the
    regions' wall times, the GPU's statistics per region (checked to cover EXACTLY its frames), HIP-event profiles."""
    cfg = B.default_config(device_id=dev.index or 0, max_tiles=max_tiles, max_points=max(f.xyz.shape[0] for f in ring.frames),
                           pipeline_frames=pipeline, **integ_cfg(wl, **cfg_extra))
    integ = B.HipIntegrator(cfg)
    pinned = None
    if entry == "host":
        # page-locked host buffers (ks_host_alloc), one set per distinct frame: the call itself moves them (H2D inside)
        import ctypes
        import numpy as np
        pinned = []
        for f in ring.frames:
            n = f.xyz.shape[0]
            bufs = []
            for arr in (f.xyz, f.rgba, f.labels):
                p = B.lib().ks_host_alloc(arr.nbytes)
                v = np.frombuffer((ctypes.c_uint8 * arr.nbytes).from_address(p), dtype=arr.dtype).reshape(arr.shape)
                v[...] = arr
                bufs.append((p, v))
            pinned.append((n, bufs))

    if entry == "depth":
        # the depth + label IMAGES from page-locked host buffers (4 + 1 bytes per pixel instead of 17 per point);
        # the back-projection runs on the GPU, the result is the same cloud
        import ctypes
        import numpy as np
        pinned = []
        for f in ring.frames:
            bufs = []
            for arr in (np.ascontiguousarray(f.depth, dtype=np.float32), np.ascontiguousarray(f.label_img, dtype=np.uint8)):
                p = B.lib().ks_host_alloc(arr.nbytes)
                v = np.frombuffer((ctypes.c_uint8 * arr.nbytes).from_address(p), dtype=arr.dtype).reshape(arr.shape)
                v[...] = arr
                bufs.append((p, v))
            pinned.append((0, bufs))

    def step(i):
        if entry == "depth":
            f = ring.host(i)
            _, bufs = pinned[i % len(ring)]
            return integ.integrate_depth(f.T_G_C, bufs[0][1], f.K, label_img=bufs[1][1])
        if entry == "host":
            f = ring.host(i)
            n, bufs = pinned[i % len(ring)]
            return integ.integrate(f.T_G_C, bufs[0][1], bufs[1][1], bufs[2][1])
        x, c, l = ring.dev(i)
        return integ.integrate_device(ring.host(i).T_G_C, x.data_ptr(), c.data_ptr(), l.data_ptr(), x.shape[0])

    prime = PRIME if prime is None else prime
    # whole turns of the ring before t0, so that every timed region is the ring's frames 0 .. K-1 in order
    base = -(-(prime + W) // len(ring)) * len(ring) if K == len(ring) else prime + W
    for i in range(base):
        step(i)
    integ.flush()             # completes the untimed frames AND hands their statistics over (discarded):
    integ.synchronize()       # nothing is pending or owed at t0
    # level 2: only the k_apply dispatch of every 4th frame carries HIP events (per-stage events
    # would put stream bubbles into every timed frame)
    integ.profile_enable(2)
    integ.profile(reset=True)
    regions = []

    def eo_stats():   # (counters of the library: no synchronisation)
        try:
            return integ.early_out_stats()
        except Exception:
            return {}

    # one UNTIMED region first, run exactly like the timed ones (K steps, flush, synchronize): the first region after the priming
    # turns was 10 - 15 % slower than the rest in every run (clocks, the first replay of the graphs with the k_apply events attached)
    for r in range(-1, R):
        eo0 = eo_stats()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        updates = points = rays = 0
        first = base + max(r, 0) * K
        for i in range(first, first + K):
            st = step(i)          # pipelined: statistics of the frame(s) completed by this call
            updates += st.n_voxel_updates
            points += st.n_points
            rays += st.n_rays_cast
        st = integ.flush()        # the tails of the last frames, inside the timed region
        updates += st.n_voxel_updates
        points += st.n_points
        rays += st.n_rays_cast
        integ.synchronize()
        reduce_stats = reduce_fn(integ) if reduce_fn else None
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        want_points = sum(int(ring.host(i).xyz.shape[0]) for i in range(first, first + K))
        assert points == want_points, f"statistics cover {points} points, the timed frames hold {want_points}"
        if r < 0:
            continue
        eo1 = eo_stats()
        regions.append(dict(dt=dt, updates=updates, points=points, rays=rays, reduce=reduce_stats,
                            frames=list(range(first, first + K)),
                            early_out={k: eo1[k] - eo0[k] for k in ("frames", "rounds", "fallbacks") if k in eo0 and k in eo1}))
    prof = integ.profile(reset=True)
    # per-stage breakdown: separate untimed pass over a few frames with events around every stage
    integ.profile_enable(1)
    for i in range(base, base + min(10, K)):
        step(i)
    integ.flush()
    stage_prof = integ.profile()
    integ.profile_enable(0)
    n_tiles = len(integ.tile_keys())
    eo = integ.early_out_iterations()
    integ.close()
    if pinned:
        for _, bufs in pinned:
            for p, v in bufs:
                del v
                B.lib().ks_host_free(p)
    return dict(regions=regions, prof=prof, stage_prof=stage_prof, tiles=n_tiles, early_out_iterations=eo)


def gpu_counts(B, dev, wl, frames, max_tiles, **cfg_extra):
    """The GPU's own per-frame update counts for a few frames (unpipelined; for the oracle/GPU ratio)."""
    cfg = B.default_config(device_id=dev.index or 0, max_tiles=max_tiles, max_points=max(f.xyz.shape[0] for f in frames),
                           pipeline_frames=0, **integ_cfg(wl, **cfg_extra))
    integ = B.HipIntegrator(cfg)
    out = [int(integ.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates) for f in frames]
    integ.close()
    return out


def early_out_fidelity(B, dev, wl, frames, max_tiles):
    """The benched `fast` schedule against the SERIAL reference order, measured here and now: both integrate the
    same frames into fresh maps; touched-voxel Jaccard, label agreement on the common voxels, update ratio."""
    import numpy as np
    from oracle import oracle_py as O
    o = O.Oracle(O.default_config(integrator_threads=1, **integ_cfg(wl, early_out_phase_growth=0)))
    h = B.HipIntegrator(B.default_config(device_id=dev.index or 0, max_tiles=max_tiles, max_points=max(f.xyz.shape[0] for f in frames),
                                         pipeline_frames=0, **integ_cfg(wl)))
    uo = uh = 0
    for f in frames:
        uo += int(o.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates)
        uh += int(h.integrate(f.T_G_C, f.xyz, f.rgba, f.labels).n_voxel_updates)
    so = {tuple(x) for x in o.block_indices().tolist()}
    sh = {tuple(x) for x in h.block_indices().tolist()}
    common = np.array(sorted(so & sh), dtype=np.int32).reshape(-1, 3)
    _, ot, osem = o.download(common)
    _, ht, hsem = h.download(common)
    to, th = ot["weight"] > 0, ht["weight"] > 0
    both = to & th
    # voxels of blocks only one side allocated count against the union
    extra = 0
    for only, integ in ((so - sh, o), (sh - so, h)):
        if only:
            _, t, _ = integ.download(np.array(sorted(only), dtype=np.int32).reshape(-1, 3))
            extra += int((t["weight"] > 0).sum())
    union = int((to | th).sum()) + extra
    o.close()
    h.close()
    same = (so == sh and np.array_equal(osem["label"], hsem["label"]) and np.array_equal(osem["priors"].view(np.uint32), hsem["priors"].view(np.uint32))
            and np.array_equal(ot["distance"].view(np.uint32), ht["distance"].view(np.uint32)) and np.array_equal(ot["weight"].view(np.uint32), ht["weight"].view(np.uint32))
            and np.array_equal(ot["color"], ht["color"]) and uo == uh)
    return {"frames": len(frames), "bit_exact_vs_serial_reference": bool(same), "touched_jaccard": round(float(both.sum()) / max(1, union), 5),
            "block_jaccard": round(len(so & sh) / max(1, len(so | sh)), 5),
            "label_agreement_common_voxels": round(float((osem["label"] == hsem["label"])[both].mean()) if both.any() else 1.0, 5),
            "updates_gpu_over_serial": round(uh / max(1, uo), 4),
            "how": "this run: the benched mode (HIP) vs the serial reference order (CPU oracle, 1 thread), same frames, fresh maps; bit_exact = same "
                   "blocks, every voxel's label / priors / distance / weight / colour identical"}


def pmc_traffic(name):
    """HBM bytes per k_apply launch from the committed PMC pass of this command (profiles/r03_pmc_<name>.json,
    written by tools/pmc_bench.sh: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 unit correction applied)."""
    for tag in ("r06", "r05", "r04", "r03"):
        try:
            return json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc_{name}.json")))
        except Exception:
            continue
    return None


def roofline_of(m, region, K, upd_counted, world=1, pmc_name=None):
    """Whole frame + per stage + k_apply, all against the HBM roofline (algorithmic bytes)."""
    prof, sp = m["prof"], m["stage_prof"]
    frame_s = region["dt"] / K
    whole_bytes = (BYTES_PER_UPDATE * upd_counted + BYTES_PER_POINT * region["points"] / world) / K   # per GPU
    whole_gbs = whole_bytes / frame_s / 1e9
    nfr = max(1, sp["frames"])
    stage_ms = {k: v / nfr for k, v in sp["ms"].items()}
    tot = sum(stage_ms.values()) or 1.0
    upd_pf, pts_pf = sp["updates"] / nfr, sp["points"] / nfr
    stage_bytes = {"points": BYTES_PER_POINT * pts_pf, "apply": BYTES_PER_UPDATE * upd_pf}  # the stages that move the algorithmic bytes
    stages = {}
    for k, v in stage_ms.items():
        e = {"ms_per_frame": round(v, 4), "share_of_kernel_time": round(v / tot, 4)}
        if k in stage_bytes and v > 0:
            gbs = stage_bytes[k] / (v * 1e-3) / 1e9
            e.update({"achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 5)})
        stages[k] = e
    dominant = max(stage_ms, key=lambda k: stage_ms[k])
    apply_ms = prof["apply_kernel_ms"] / max(1, prof["apply_kernel_launches"])
    upd_per_launch = prof["apply_kernel_updates"] / max(1, prof["apply_kernel_launches"])
    a_gbs = BYTES_PER_UPDATE * upd_per_launch / (apply_ms * 1e-3) / 1e9 if apply_ms > 0 else 0.0
    pmc = pmc_traffic(pmc_name) if pmc_name else None
    # round 6: `traffic` is the WHOLE frame's HBM bytes (every kernel, per frame) — the scope of `kernel` / `achieved`; the update
    # kernel's own counter bytes sit with its figures under k_apply
    traffic = (pmc.get("whole_frame_hbm_bytes_per_frame") or None) if pmc else None
    k_apply_traffic = pmc.get("k_apply_hbm_bytes_per_launch") if pmc else None
    return {
        "bound": "hbm", "kernel": "whole frame (all stages, wall clock of the median timed region)",
        "achieved": round(whole_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(whole_gbs / HBM_PEAK_GBS, 5),
        "traffic": traffic,
        "traffic_note": (f"whole frame: HBM bytes of every kernel per frame (FETCH_SIZE + WRITE_SIZE, separate --pmc passes of this command, "
                         f"gfx950 correction applied; null = the committed pass predates round 6): {pmc.get('source', 'profiles/')}" if pmc else
                         "no committed PMC pass for this workload (profiles/r0N_pmc_*.json)"),
        "algorithmic_bytes_per_frame": int(whole_bytes),
        "dominant_stage": dominant,
        "stages": stages,
        "stage_note": "HIP events around every stage: separate untimed pass over 10 frames (stages of one frame back to "
                      "back); march = early-out phases + scan + pair emission, sort_* = radix sorts, apply(+_long) = per-voxel update",
        "k_apply": {"achieved": round(a_gbs, 2), "frac": round(a_gbs / HBM_PEAK_GBS, 5), "avg_launch_ms": round(apply_ms, 5),
                    "algorithmic_bytes_per_launch": int(BYTES_PER_UPDATE * upd_per_launch),
                    "traffic": k_apply_traffic, "kernel": (pmc or {}).get("update_kernel", "k_apply"),
                    "timed_launches": prof["apply_kernel_launches"],
                    "note": "events attached to the k_apply dispatch of every 4th timed frame (GPU's own update count); other "
                            "stages of neighbouring frames share the GPU (pipelined)"},
    }



def steady_state(B, torch, dist, dev, wl, ring, n_frames, pipeline, max_tiles, **cfg_extra):
    """The same stream in ONE long region: a region of K frames pays the pipeline's fill and drain once per K (flush + synchronize are
    inside it, as the measurement contract wants), which at 12 frames per region is 3 % of a C4 frame."""
    m = measure(B, torch, dist, dev, wl, ring, 0, n_frames, 2, pipeline, max_tiles, 1, **cfg_extra)
    dts = sorted(r["dt"] for r in m["regions"])
    reg = min(m["regions"], key=lambda r: r["dt"])
    return {"frames_in_one_region": n_frames, "ms_per_step": round(dts[0] / n_frames * 1e3, 4),
            "ms_per_step_all_regions": [round(d / n_frames * 1e3, 4) for d in dts],
            "gpu_counted_value": round(reg["updates"] / reg["dt"] / 1e6, 3), "unit": "Mvoxel-updates/s",
            "note": "one region of that many frames (flush + synchronize inside): the stream's steady-state rate; never `value`"}


def record(name, wl, m, K, count_of_frame, how, world=1, pmc_name=None, note=None, credit_scale=1.0):
    """One result record from the MEDIAN timed region (the primary line's core fields; also the secondary configs).
    count_of_frame(i) = reference-order update count of frame i, or None: the GPU's own count (x credit_scale <= 1)."""
    rates, per_region = [], []
    for reg in m["regions"]:
        gpu_upd = reg["updates"]
        counted = gpu_upd * min(1.0, credit_scale)
        if count_of_frame is not None:
            oc = sum(count_of_frame(i) for i in reg["frames"])
            counted = min(oc, gpu_upd)     # never credit updates the GPU skipped
        rates.append(counted / reg["dt"])
        per_region.append((reg, counted))
    order = sorted(range(len(rates)), key=lambda i: rates[i])
    mid = order[len(order) // 2]
    reg, counted = per_region[mid]
    dt, gpu_upd = reg["dt"], reg["updates"]
    _, spread = median_spread(rates)
    credit = how
    if count_of_frame is not None and sum(count_of_frame(i) for i in reg["frames"]) > gpu_upd:
        credit = how + "; the GPU performed fewer updates than that, so its own count is credited"
    out = {
        "config": name,
        "workload": f"{wl['w']}x{wl['h']} depth+label trajectory ({wl['scene']}), '{wl['method']}' integrator, "
                    f"{wl['voxel'] * 100:g} cm voxels, {wl['max_ray']:g} m rays, trunc {4 * wl['voxel']:g} m, p=0.8",
        "value": round(counted / dt / 1e6, 3), "unit": "Mvoxel-updates/s", "updates_counted_by": credit,
        "gpu_counted_value": round(gpu_upd / dt / 1e6, 3),
        "ms_per_step": round(dt / (K / world) * 1e3, 4), "frames_per_s": round(K / dt, 2), "steps": K // world,
        "repeats": len(rates), "spread": round(spread, 4),
        "ms_per_step_all_regions": [round(r["dt"] / (K / world) * 1e3, 4) for r in m["regions"]],
        "early_out_all_regions": [r.get("early_out") for r in m["regions"]],   # fix-point rounds / frames the host-driven loop repeated, per region
        "points_per_frame": int(reg["points"] / K), "rays_per_frame": int(reg["rays"] / K),
        "updates_per_frame": int(counted / K), "gpu_updates_per_frame": int(gpu_upd / K), "tiles": m["tiles"],
        "roofline": roofline_of(m, reg, K // world, counted / world, world, pmc_name),
    }
    if note:
        out["note"] = note
    return out, reg


def adapter_record(wl_frames):
    """The C++ drop-in adapter behind the reference's virtual, as an UNMODIFIED SemanticTsdfServer would drive it:
    host clouds in, host Layers current after every integratePointCloud (SyncPolicy::kEveryFrame), and the
    on-demand policy with pipelined frames beside it (kimera_semantics_amd/host/adapter_demo)."""
    import re
    import struct
    import subprocess
    import tempfile
    from kimera_semantics_amd import synth
    demo = os.path.join(ROOT, "kimera_semantics_amd", "host", "adapter_demo")
    if not os.path.exists(demo):
        return {"config": "adapter", "error": "adapter_demo not built"}
    tmp = tempfile.mkdtemp(prefix="ks_adapter_")
    fin, fout, csv = (os.path.join(tmp, x) for x in ("in.bin", "out.bin", "labels.csv"))
    with open(csv, "w") as fh:
        fh.write("name,red,green,blue,alpha,id\n")
        for i, (r, g, b, a) in enumerate(synth.default_label_colors()[:21]):
            fh.write(f"label{i},{int(r)},{int(g)},{int(b)},{int(a)},{i}\n")
    with open(fin, "wb") as fh:
        fh.write(struct.pack("<I", len(wl_frames)))
        for f in wl_frames:
            fh.write(f.T_G_C.astype("<f4").tobytes())
            fh.write(struct.pack("<I", len(f.xyz)))
            fh.write(f.xyz.astype("<f4").tobytes())
            fh.write(f.rgba.tobytes())
    out = {"config": "adapter", "workload": f"{len(wl_frames)} 640x480 host clouds through kimera::HipSemanticTsdfIntegrator "
                                            "(TsdfIntegratorBase virtual), steady state = last third of the frames; every_frame_sync = what an "
                                            "unmodified SemanticTsdfServer gets, on_demand = the sequence of integration/server.patch"}
    real = os.path.join(ROOT, "integration", "_build", "adapter_demo_real")
    runs = [(demo, m, pipe, f"{m}_{key}") for m in ("fast", "merged") for pipe, key in (("0", "every_frame_sync"), ("1", "on_demand_sync_pipelined"))]
    if os.path.exists(real):
        # the reference's OWN factory (its source + integration/factory.patch) hands the integrator out with default options,
        # then the server-side patch's sequence: setSyncPolicy(kOnDemand), syncLayers() where the Layers are read
        runs.append((real, "fast_hip", "1", "fast_hip_real_factory_patched_server_sequence"))
    for exe, method, pipe, key in runs:
        res = subprocess.run([exe, method, csv, fin, fout, "1", "2", "-1", pipe], capture_output=True, text=True, timeout=600)
        mm = re.search(r"integratePointCloud ([0-9.]+) ms/frame over (\d+) frames, ([0-9.]+) ms/frame over the last (\d+)", res.stdout)
        out[f"{key}_ms_per_frame"] = float(mm.group(3)) if mm else None
        if not mm:
            out[f"{key}_error"] = (res.stdout + res.stderr)[-300:]
    for p in (fin, fout):
        try:
            os.remove(p)
        except OSError:
            pass
    return out


def c5_record(B, torch, dist, dev, rank, world, comm, ddev=None):
    """BASELINE.json configs[4]: a batch of 8 overlapping 640x480 frames (arc poses looking at the same wall), frame-sharded over
    the ranks.  EXACT split (ks_integrate_round_exact): rank r marches frames r, r + world, ... — ray casting, early-out, emission —
    and every voxel update travels, as a 20-byte record, to the rank that owns the voxel's tile, which applies the frames in frame
    order: the tiles a rank owns are compared, record by record, with the same frames integrated sequentially on one GPU
    (`bit_exact_vs_sequential`).  The timed region is the whole batch (march + exchange + apply on every rank).  The older
    map-merging exchange (ks_reduce: a different arithmetic, labels agree to ~99 %) is kept as `tile_merge_reduce` beside it."""
    import numpy as np
    from kimera_semantics_amd import parallel as PAR
    from kimera_semantics_amd import synth
    wl = WORKLOADS["C2"]
    n_frames = 8 if 8 % world == 0 else world
    scene = synth.make_scene("room")
    frames = [synth.render_frame(scene, synth.arc_pose(k, n=n_frames), wl["w"], wl["h"], seed=100 + k) for k in range(n_frames)]
    kw = dict(integ_cfg(wl), voxels_per_side=8)   # host block = device tile: ownership is per block
    mk = lambda: B.HipIntegrator(B.default_config(device_id=dev.index or 0, max_tiles=1 << 13, max_points=wl["w"] * wl["h"], **kw))   # noqa: E731
    rec = {"config": "C5", "workload": f"{n_frames} arc-pose 640x480 frames looking at the same wall, frame-sharded x{world}: every rank marches "
                                       "its frames, updates travel to the tile owners as 20-byte records (RCCL send / recv, all peers at once), "
                                       "owners apply in frame order (ks_integrate_round_exact)",
           "frames": n_frames, "unit": "Mvoxel-updates/s"}
    # the same batch sequentially on this GPU (untimed): what the owners' tiles must be
    seq = mk()
    for f in frames:
        seq.integrate(f.T_G_C, f.xyz, f.rgba, f.labels)
    seq.synchronize()

    def tiles_of(h):
        keys = h.tile_keys()
        buf = torch.empty((len(keys), 16384), dtype=torch.int32, device=dev)
        if len(keys):
            h.export_tiles(np.arange(len(keys), dtype=np.uint32), buf.data_ptr())
        torch.cuda.synchronize()
        return keys, buf.cpu().numpy().view(np.uint32).reshape(len(keys), 512, 32)[:, :, :25]

    if comm is not None or world == 1:
        marcher, owner = mk(), mk()
        # one untimed round on throw-away contexts (buffers, communicator paths), then the timed batch
        wm, wo = mk(), mk()
        wo.integrate_round_exact(wm, comm, rank, world, 0, frames[rank].T_G_C, frames[rank].xyz, frames[rank].rgba, frames[rank].labels)
        wm.close()
        wo.close()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        stats = []
        for r0 in range(0, n_frames, world):
            f = frames[r0 + rank]
            stats.append(owner.integrate_round_exact(marcher, comm, rank, world, r0, f.T_G_C, f.xyz, f.rgba, f.labels))
        owner.synchronize()
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        sk, srec = tiles_of(seq)
        want = {int(k): srec[i] for i, k in enumerate(sk.tolist()) if PAR.owner_of(np.array([k], dtype=np.uint64), world)[0] == rank}
        gk, grec = tiles_of(owner)
        same_set = sorted(int(k) for k in gk.tolist()) == sorted(want)
        diff = 0 if same_set else -1
        if same_set:
            for i, k in enumerate(gk.tolist()):
                diff += int((grec[i] != want[int(k)]).any(axis=1).sum())
        agg = torch.tensor([float(sum(s["updates_marched"] for s in stats)), float(sum(s["bytes_sent"] for s in stats)),
                            float(len(want)), float(0 if (same_set and diff == 0) else 1), float(max(diff, 0)),
                            float(any(s["origin_voxel_touched"] for s in stats))], device=ddev or dev, dtype=torch.float64)
        tmax = torch.tensor([dt], device=ddev or dev, dtype=torch.float64)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        a, dtm = agg.tolist(), float(tmax.item())
        marcher.close()
        owner.close()
        rec.update({"batch_ms": round(dtm * 1e3, 3), "frames_per_s": round(n_frames / dtm, 2), "gpu_counted_value": round(a[0] / dtm / 1e6, 3),
                    "exchange": {"bytes_sent": int(a[1]), "bytes_per_update": 20},
                    "bit_exact_vs_sequential": bool(a[3] == 0), "owned_tiles_compared": int(a[2]), "voxel_records_differing": int(a[4]),
                    "origin_voxel_touched": bool(a[5] > 0)})
    else:
        rec["skipped_exact_split"] = "no ctypes RCCL communicator in this run: ks_integrate_round_exact needs one"
    # ---- the map-merging exchange (rounds 2-5), for comparison: per-rank maps, ONE ks_reduce of the dirty tiles ----
    h = mk()
    mine = list(range(rank, n_frames, world))
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for k in mine:
        h.integrate(frames[k].T_G_C, frames[k].xyz, frames[k].rgba, frames[k].labels)
    h.synchronize()
    rstats = h.reduce(comm, rank, world) if comm is not None else PAR.reduce_maps(PAR.HipTileStore(h, dev))
    h.synchronize()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    keys = h.tile_keys()
    own = keys[PAR.owned_tile_mask(keys, rank, world)]
    bias = 1 << 17
    idx = np.stack([((own >> np.uint64(36)) & np.uint64(0x3ffff)).astype(np.int64) - bias,
                    ((own >> np.uint64(18)) & np.uint64(0x3ffff)).astype(np.int64) - bias,
                    (own & np.uint64(0x3ffff)).astype(np.int64) - bias], axis=1).astype(np.int32) if len(own) else np.zeros((0, 3), np.int32)
    _, ht, hs = h.download(idx)
    _, st, ss = seq.download(idx)
    touched = st["weight"] > 0
    agg = torch.tensor([float(touched.sum()), float(((hs["label"] == ss["label"]) & touched).sum()),
                        float(np.abs(ht["distance"] - st["distance"])[touched].sum()), dt,
                        float(rstats["tiles_sent"]), float(rstats["bytes_sent"])], device=ddev or dev, dtype=torch.float64)
    tmax = torch.tensor([dt], device=ddev or dev, dtype=torch.float64)
    dist.all_reduce(agg, op=dist.ReduceOp.SUM)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    h.close()
    seq.close()
    a = agg.tolist()
    rec["reduce"] = {"tiles_sent": int(a[4]), "bytes_sent": int(a[5])}
    rec["tile_merge_reduce"] = {"batch_ms": round(float(tmax.item()) * 1e3, 3), "label_agreement_vs_sequential": round(a[1] / max(1.0, a[0]), 6),
                                "mean_abs_distance_diff_vs_sequential": a[2] / max(1.0, a[0]), "voxels_compared": int(a[0])}
    return rec


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher around it: re-executes itself under torch.distributed.run with N
    ranks on this node (one per GPU, RCCL) and passes the children's stdout (rank 0's JSON line) through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), KS_BENCH_LAUNCHED="1")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def mapped_library():
    """The libks_hip.so this process has actually mapped (from /proc/self/maps)."""
    try:
        for ln in open("/proc/self/maps"):
            if "libks_hip" in ln:
                return ln.split()[-1]
    except OSError:
        pass
    return None


LINE_LIMIT = 4000   # bytes of the ONE stdout line; everything else goes to the full record (profiles/bench_full_r06.json)


def compact_line(out, full_path):
    """The ONE line printed to stdout: the contract's fields + roofline + cpu_baseline + fidelity, <= LINE_LIMIT bytes.
    `out` is the full record (written to `full_path`).  Pure function of its arguments (tests/test_host_logic.py)."""
    def pick(d, keys):
        return {k: d[k] for k in keys if isinstance(d, dict) and k in d}
    rf = out.get("roofline") or {}
    line = pick(out, ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data", "frames_per_s", "frames_per_s_device_resident", "gpu_counted_value"])
    cfg = out.get("config") or {}
    line["config"] = pick(cfg, ["workload", "early_out", "pipeline_frames", "parallelism", "points_per_frame", "rays_per_frame",
                                "updates_per_frame"])
    line["roofline"] = pick(rf, ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "dominant_stage"])
    if "k_apply" in rf:
        line["roofline"]["k_apply"] = pick(rf["k_apply"], ["achieved", "frac", "avg_launch_ms", "algorithmic_bytes_per_launch", "traffic", "kernel"])
    if "stages" in rf:
        line["roofline"]["stage_ms"] = {k: v.get("ms_per_frame") for k, v in rf["stages"].items()}
    if "timing" in out:
        line["spread"] = out["timing"].get("spread_max_minus_min_over_median")
    if "cpu_baseline" in out:
        line["cpu_baseline"] = pick(out["cpu_baseline"], ["value", "unit", "cores", "kind", "host_cores", "frames_per_s",
                                                          "reference_default_all_cores_value", "by_threads", "sample"])
    if "early_out_fidelity" in out:
        line["early_out_fidelity"] = pick(out["early_out_fidelity"], ["frames", "touched_jaccard", "label_agreement_common_voxels",
                                                                      "updates_gpu_over_serial", "bit_exact_vs_serial_reference", "error"])
    if "host_inputs_h2d_inside" in out:
        line["host_inputs_h2d_inside"] = pick(out["host_inputs_h2d_inside"], ["value", "ms_per_step", "frames_per_s", "spread"])
    if "reduce" in out:
        line["reduce"] = out["reduce"]
    sec = []
    for r in out.get("secondary", []):
        if not isinstance(r, dict):
            continue
        e = pick(r, ["config", "value", "ms_per_step", "error", "batch_ms", "frames", "skipped"])
        if isinstance(r.get("roofline"), dict):
            e["frac"] = r["roofline"].get("frac")
        if r.get("config") == "adapter":
            e.update({k: v for k, v in r.items() if k.endswith("_ms_per_frame")})
        if r.get("config") == "C5":
            e.update(pick(r, ["reduce", "gpu_counted_value", "bit_exact_vs_sequential", "exchange"]))
        sec.append(e)
    if sec:
        line["secondary"] = sec
    line["library"] = out.get("library")
    line["full_record"] = full_path
    # never longer than the limit: optional parts go first
    for drop in (("secondary",), ("roofline", "stage_ms"), ("cpu_baseline", "sample"), ("cpu_baseline", "by_threads"),
                 ("config", "early_out"), ("early_out_fidelity",), ("config", "parallelism")):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k, {})
        d.pop(drop[-1], None)
    return json.dumps(line)


def main():
    t_start = time.time()
    args = parse()
    if os.environ.get("KS_HIP_LIB"):
        raise SystemExit("bench.py measures the in-tree kimera_semantics_amd/libks_hip.so only: unset KS_HIP_LIB "
                         f"(= {os.environ['KS_HIP_LIB']})")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)
    # stdout carries exactly ONE line, the JSON record: libraries that print to the C stdout (RCCL's version
    # banner, flushed at exit) are sent to stderr for the rest of the process
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    from kimera_semantics_amd import binding as B

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the launcher and --gpus disagree")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # KS_BENCH_SHARED_GPU=1 (tests only: a development box has ONE GPU): every rank uses cuda:0, torch.distributed runs
    # over gloo on host tensors, and ks_reduce's communicator is whatever KS_RCCL_LIB names (the test double
    # tests/mock_rccl: RCCL itself refuses two ranks on one device)
    shared_gpu = os.environ.get("KS_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ddev = torch.device("cpu") if shared_gpu else dev   # where tensors of torch.distributed collectives live
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus

    K, W = args.steps, args.warmup
    R = max(MIN_REPEATS_PRIMARY, -(-MIN_TIMED_FRAMES // max(1, K)))
    wl = dict(WORKLOADS["C2"], w=args.width, h=args.height, method=args.method)
    # frame-sharded: rank r integrates trajectory frames r, r+world, ... (weak scaling: the same number of frames per GPU);
    # a ring of exactly K frames per rank: every timed region is one turn of it
    n_distinct = K
    frames = make_frames(wl, [rank + world * k for k in range(n_distinct)])
    ring = FrameRing(frames, torch, dev)
    # bag replay = a stream of frames: frame pipelining on.  `fast` (the headline): 12 calls of lag = THREE batches of four frames in
    # flight (with 8 — rounds 4 / 5 — the call that needs frame f - 8's snapshot finds its batch just finishing and the GPU's queues run
    # dry: 0.52 vs 0.47 ms/frame; sub-record C2-pipeline-8); `merged` is indifferent (0.506 / 0.505 / 0.495 at 8 / 12 / 16): 8
    PIPE_FAST, PIPE_MERGED = 12, 8
    pipeline = 0 if args.no_pipeline else int(os.environ.get("KS_BENCH_PIPE", str(PIPE_FAST if args.method == "fast" else PIPE_MERGED)))

    reduce_fn = None
    comm = None
    if world > 1 or os.environ.get("KS_BENCH_C5") == "1":
        from kimera_semantics_amd import parallel as PAR
        if world == 1 and not dist.is_initialized():   # single-GPU self-test of this code path
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        try:
            comm = PAR.rccl_comm(rank, world, ddev)   # an ncclComm_t of the librccl ks_reduce loads (id broadcast over torch.distributed)
        except PAR.RcclBootstrapStuck as e:
            # a thread of this process is stuck inside librccl with the unique id consumed: no collective can be started from
            # here any more (the other ranks would wait for it) — the run ends HERE, loudly, and that is bench.py's decision
            sys.stderr.write(f"bench: {e}\n")
            os._exit(3)
        except Exception as e:   # (never take the line down: the same protocol also runs over torch.distributed)
            sys.stderr.write(f"bench: ctypes RCCL communicator unavailable ({type(e).__name__}: {e}); exchange through torch.distributed\n")
            comm = None
        if world > 1:
            # every rank must take the same path
            ok = torch.tensor([1 if comm is not None else 0], device=ddev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if comm is not None:   # a rank whose communicator DID come up must not keep a half-connected one around
                    PAR.rccl_abort(comm)
                comm = None
        if comm is None and shared_gpu:
            raise SystemExit("KS_BENCH_SHARED_GPU=1 needs the ks_reduce communicator (KS_RCCL_LIB): the torch.distributed exchange needs one GPU per rank")
    exchange = "ks_reduce (C ABI, RCCL all-to-all of the dirty tiles to hash-owners)" if comm is not None else \
        "parallel.reduce_maps (the same protocol over torch.distributed)"
    if world > 1:
        if comm is None:
            PAR.warm_up(dev)

        def reduce_fn(integ):
            # the one exchange step of the frame-sharded path: dirty tiles -> their owner ranks, all peers at once over
            # RCCL/xGMI, deterministic owner merge
            st = integ.reduce(comm, rank, world) if comm is not None else PAR.reduce_maps(PAR.HipTileStore(integ, dev))
            integ.synchronize()
            return st

    m = measure(B, torch, dist, dev, wl, ring, W, K, R, pipeline, 1 << 13, world, reduce_fn=reduce_fn)

    # N_updates in the serial reference order for this rank's frames (outside the timed regions)
    upd_serial = None if args.no_oracle_count else oracle_counts(wl, frames)
    count_of_frame = (lambda i: upd_serial[i % n_distinct]) if upd_serial is not None else None
    how = ("serial reference order (CPU oracle, 1 thread), every timed frame" if upd_serial is not None
           else "GPU's own count (oracle count skipped)")

    if world > 1:
        # whole-job regions: MAX of the ranks' wall times, SUM of their work
        for r, reg in enumerate(m["regions"]):
            t = torch.tensor([reg["dt"]], device=ddev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            oc = sum(count_of_frame(i) for i in reg["frames"]) if count_of_frame else 0
            u = torch.tensor([reg["updates"], reg["points"], reg["rays"], oc], device=ddev, dtype=torch.float64)
            dist.all_reduce(u, op=dist.ReduceOp.SUM)
            reg["dt"] = float(t.item())
            reg["updates"], reg["points"], reg["rays"] = int(u[0].item()), int(u[1].item()), int(u[2].item())
            reg["oracle_sum"] = int(u[3].item())
        if count_of_frame is not None:
            sums = {tuple(reg["frames"]): reg["oracle_sum"] for reg in m["regions"]}
            per_frame = {}
            for fr, s in sums.items():
                for i in fr:
                    per_frame[i] = s / len(fr)
            count_of_frame = lambda i: per_frame[i]   # noqa: E731  (only sums over whole regions are used)

    c5 = None
    if world > 1 or os.environ.get("KS_BENCH_C5") == "1":
        try:
            c5 = c5_record(B, torch, dist, dev, rank, world, comm, ddev)
        except Exception as e:   # collective calls above are symmetric; a local failure must not take the line down
            c5 = {"config": "C5", "error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        Kall = K * world
        rec, reg = record("C2" if args.method == "fast" else "C3", wl, m, Kall, count_of_frame, how, world,
                          pmc_name="c2" if args.method == "fast" else "c3")
        out = {
            "metric": METRIC,
            "value": rec["value"], "unit": rec["unit"],
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": rec["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "frames_per_s": rec["frames_per_s"],
            "gpu_counted_value": rec["gpu_counted_value"],
            "updates_counted_by": rec["updates_counted_by"],
            "timing": {"untimed_frames_before_t0": -(-(PRIME + W) // K) * K + K, "timed_regions": R, "steps_per_region": K,
                       "untimed": "whole turns of the ring covering PRIME + warm-up frames, then one more turn run exactly like a timed region",
                       "every_region_integrates": "the same K frames, in the same order (one turn of the ring)",
                       "reported": "median region", "spread_max_minus_min_over_median": rec["spread"],
                       "ms_per_step_all_regions": rec["ms_per_step_all_regions"],
                       "early_out_all_regions": rec["early_out_all_regions"]},
            "config": {"workload": "bag-replay stand-in: " + rec["workload"],
                       "frames_per_gpu": K, "pipeline_frames": pipeline, "distinct_frames_replayed_cyclically": n_distinct,
                       "points_per_frame": rec["points_per_frame"], "rays_per_frame": rec["rays_per_frame"],
                       "updates_per_frame": rec["updates_per_frame"], "gpu_updates_per_frame": rec["gpu_updates_per_frame"],
                       "early_out": ("the reference's serial result (library default): event-driven fix point on the device, pipelined; "
                                     "early_out_fidelity = checked in this run; the ordered-phase schedule alone: secondary C2-ordered-phases"
                                     if int(os.environ.get("KS_BENCH_GROWTH", "0")) == 0 else
                                     "ordered-phase schedule (KS_BENCH_GROWTH): not the reference's map") if args.method == "fast" else "n/a (merged)",
                       "bundle_order": "reference (std::unordered_map iteration order, computed on the GPU)" if args.method == "merged" else "n/a (fast)",
                       "parallelism": f"frame-sharded x{world}" + (
                           f" + one exchange at the end of every timed region: {exchange}"
                           if world > 1 else "")},
            "roofline": rec["roofline"],
            "host_ms_per_frame": {"in_call": round(m["prof"]["host_ms"] / max(1, m["prof"]["frames"]), 4),
                                  "of_which_waiting_for_snapshot": round(m["prof"]["host_wait_ms"] / max(1, m["prof"]["frames"]), 4)},
        }
        if world == 1 and not args.no_secondary:
            try:
                out["steady_state"] = steady_state(B, torch, dist, dev, wl, ring, 10 * len(ring), pipeline, 1 << 13)
            except Exception as e:
                out["steady_state"] = {"error": f"{type(e).__name__}: {e}"}
        if reg["reduce"] is not None:
            out["reduce"] = reg["reduce"]
        if c5 is not None:
            out.setdefault("secondary", []).append(c5)
        if world == 1 and args.method == "fast" and upd_serial is not None:
            try:
                out["early_out_fidelity"] = early_out_fidelity(B, dev, wl, [ring.host(i) for i in range(2)], 1 << 13)
            except Exception as e:
                out["early_out_fidelity"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_cpu_baseline and world == 1 and upd_serial is not None:   # reported on rank 0 at N=1 only
            first = [i % n_distinct for i in range(min(args.cpu_frames, n_distinct))]
            out["cpu_baseline"] = cpu_baseline(args, wl, [frames[i] for i in first], [upd_serial[i] for i in first])
        if world == 1 and not args.no_secondary:
            sec = out.get("secondary", [])
            only = set(filter(None, args.only_secondary.split(",")))

            def want(name):
                return not only or name in only

            # ---- the exact serial early-out mode, the merged integrator, the host-pointer entry (640x480) ----
            n_sub = n_distinct              # the same ring as the headline: the records are comparable with it and with each other
            sub_ring = ring
            for name, swl, kw in (("C2-ordered-phases", WORKLOADS["C2"], dict(cfg=dict(early_out_phase_growth=32), pipe=pipeline)),
                                  ("C2-unpipelined", WORKLOADS["C2"], dict(cfg={}, pipe=0)),
                                  ("C2-pipeline-8", WORKLOADS["C2"], dict(cfg={}, pipe=8)),
                                  ("C2-pipeline-16", WORKLOADS["C2"], dict(cfg={}, pipe=16)),
                                  ("C3", WORKLOADS["C3"], dict(cfg={}, pipe=min(pipeline, PIPE_MERGED))),
                                  ("C2-host-inputs", WORKLOADS["C2"], dict(cfg={}, pipe=pipeline, entry="host")),
                                  ("C2-depth-host-inputs", WORKLOADS["C2"], dict(cfg={}, pipe=pipeline, entry="depth"))):
                if not want(name) or args.method != "fast" or (args.width, args.height) != (640, 480):
                    continue
                try:
                    sK, sR = K, MIN_REPEATS
                    sm = measure(B, torch, dist, dev, swl, sub_ring, 2, sK, sR, kw["pipe"], 1 << 13, 1,
                                 entry=kw.get("entry", "device"), **kw["cfg"])
                    cof, show = None, "GPU's own count (oracle count skipped)"
                    if not args.no_oracle_count:
                        oc = upd_serial[:n_sub] if swl["method"] == "fast" else oracle_counts(swl, sub_ring.frames)
                        cof = lambda i, oc=oc: oc[i % len(oc)]   # noqa: E731
                        show = "serial reference order (CPU oracle, 1 thread), every timed frame"
                    note = None
                    if name == "C2-ordered-phases":
                        note = ("early_out_phase_growth = 32: the ordered-phase schedule alone (deterministic, bit-exact vs its CPU restatement, "
                                "NOT the reference's map: touched-voxel Jaccard ~0.98 against the serial order)")
                    elif name == "C2-unpipelined":
                        note = "pipeline_frames = 0: every call completes its own frame (the latency of one frame, host wait included)"
                    elif name == "C2-pipeline-8":
                        note = "pipeline_frames = 8: the headline's configuration of rounds 4 and 5 (two batches of four frames in flight); the same map"
                    elif name == "C2-pipeline-16":
                        note = ("pipeline_frames = 16: stage B of EIGHT frames per launch sequence, sixteen calls of lag (the headline runs with 12: "
                                "four per sequence, twelve calls of lag); the same map")
                    elif name == "C2-host-inputs":
                        note = ("ks_integrate_points on page-locked HOST buffers: the H2D copy of every frame is inside the call "
                                "(SURVEY.md §8d's frames/s definition); never the headline value")
                    elif name == "C2-depth-host-inputs":
                        note = ("ks_integrate_depth on page-locked HOST images (f32 depth + u8 labels: 5 bytes per pixel instead of 17 per point), "
                                "H2D copy and back-projection inside the call; the same frames, the same cloud")
                    elif name == "C3":
                        note = "bundles integrated in the reference's std::unordered_map iteration order (bit-exact vs the real sources)"
                    srec, _ = record(name, swl, sm, sK, cof, show, note=note, pmc_name="c3" if name == "C3" else None)
                    if name in ("C3", "C2-pipeline-8", "C2-pipeline-16"):
                        srec["steady_state"] = steady_state(B, torch, dist, dev, swl, sub_ring, 10 * len(sub_ring), kw["pipe"], 1 << 13)
                    if name == "C2-unpipelined" and cof is not None:
                        srec["gpu_count_equals_serial_reference_count"] = all(
                            r["updates"] == sum(cof(i) for i in r["frames"]) for r in sm["regions"])
                    sec.append(srec)
                except Exception as e:   # a secondary record must never take the primary line down
                    sec.append({"config": name, "error": f"{type(e).__name__}: {e}"})
            # ---- C4: 1280x720, 2 cm voxels, 10 m rays (both integrators on the same frames) ----
            c4_ring = None
            # (C4-fast in the default mode: the reference's serial result from the device loop for long rays — whole-ray marks
            # sorted once, then sweeps along the chains of the integration order, DESIGN.md 3.8 — one frame at a time)
            for name, steps, tiles, c4cfg in (("C4-fast", C4_STEPS, 1 << 16, {}), ("C4-fast-ordered-phases", C4_STEPS, 1 << 16, dict(early_out_phase_growth=32)),
                                              ("C4-merged", C4_STEPS, 1 << 16, {})):
                if not want(name):
                    continue
                try:
                    swl = WORKLOADS["C4-merged" if name == "C4-merged" else "C4-fast"]
                    if c4_ring is None:   # a ring of C4_STEPS distinct frames: every region is one turn of it
                        c4_ring = FrameRing(make_frames(swl, range(C4_STEPS)), torch, dev)
                    if time.time() - t_start > 600.0:
                        sec.append({"config": name, "skipped": f"the run is {time.time() - t_start:.0f} s old"})
                        continue
                    light = name == "C4-fast"
                    c4pipe = min(pipeline, PIPE_MERGED)   # (1280x720: 8 calls of lag measured best for `merged` and for the ordered phases; the default `fast` mode runs one frame at a time there)
                    sm = measure(B, torch, dist, dev, swl, c4_ring, 2, steps, MIN_REPEATS, c4pipe, tiles, 1,
                                 prime=steps if light else None, **c4cfg)
                    scale, show = 1.0, "GPU's own count (oracle count skipped)"
                    if light:
                        show = ("GPU's own count = the serial reference's: the default mode is the reference's result bit for bit "
                                "(tests/test_exact_early_out_gpu.py::test_three_full_size_c4_frames_on_the_device_vs_real_reference)")
                    elif not args.no_oracle_count:
                        # the serial oracle needs ~10 s per C4 frame: count ONE timed frame, compare with the GPU's count of it
                        i0 = 2
                        oc = oracle_counts(swl, [c4_ring.host(i0)])[0]
                        g = gpu_counts(B, dev, swl, [c4_ring.host(i0)], tiles, **c4cfg)[0]
                        ratio = oc / max(1, g)
                        if ratio >= 1.0:
                            show = (f"GPU's own count: on the first timed frame the serial reference order performs x{ratio:.4f} the GPU's "
                                    f"updates (oracle {oc}, GPU {g}); work the GPU skipped is not credited")
                        else:
                            scale = ratio
                            show = (f"GPU count x (serial-oracle / GPU) measured on the first timed frame: x{ratio:.4f} "
                                    f"(oracle {oc}, GPU {g})")
                    srec, _ = record(name, swl, sm, steps, None, show, credit_scale=scale,
                                     pmc_name={"C4-fast": "c4_fast", "C4-merged": "c4_merged"}.get(name))
                    if name == "C4-merged":
                        srec["steady_state"] = steady_state(B, torch, dist, dev, swl, c4_ring, 6 * steps, c4pipe, tiles, **c4cfg)
                    sec.append(srec)
                    torch.cuda.empty_cache()
                except Exception as e:
                    sec.append({"config": name, "error": f"{type(e).__name__}: {e}"})
            del c4_ring
            torch.cuda.empty_cache()
            if want("adapter") and args.method == "fast":
                try:
                    sec.append(adapter_record([ring.host(i) for i in range(2 * len(ring))]))   # two turns of the ring: the last third is steady state
                except Exception as e:
                    sec.append({"config": "adapter", "error": f"{type(e).__name__}: {e}"})
            out["secondary"] = sec
            # SURVEY.md §8(d) defines frames/s as the call "including H2D of the frame": the same K frames through the
            # host-pointer entry (page-locked buffers, the copy inside the call), beside the device-resident headline
            for r in sec:
                if isinstance(r, dict) and r.get("config") == "C2-host-inputs" and "value" in r:
                    out["host_inputs_h2d_inside"] = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
                                                     "frames_per_s": r["frames_per_s"], "spread": r["spread"],
                                                     "note": "same K frames as the headline, ks_integrate_points on page-locked host buffers"}
                    # `value` stays the device-resident rate (the measurement contract: inputs in HBM when the region starts, the
                    # PCIe-inclusive rate is never `value`); `frames_per_s` is what SURVEY.md 8(d) defines — H2D of the frame inside
                    out["frames_per_s_device_resident"] = out["frames_per_s"]
                    out["frames_per_s"] = r["frames_per_s"]
                    out["frames_per_s_definition"] = ("SURVEY.md 8(d): the call including the H2D copy of the frame (host-pointer entry, page-locked "
                                                      "buffers); frames_per_s_device_resident / value / ms_per_step: inputs resident in HBM")
        out["library"] = mapped_library()
        out["bench_seconds"] = round(time.time() - t_start, 1)
        # the full record (every sub-record, stage table, A/B record) goes to a file; stdout carries ONE short line
        full_path = os.environ.get("KS_BENCH_FULL") or os.path.join("profiles", "bench_full_r06.json" if world == 1 else f"bench_full_r06_n{world}.json")
        try:
            with open(os.path.join(ROOT, full_path) if not os.path.isabs(full_path) else full_path, "w") as fh:
                json.dump(out, fh, indent=1)
                fh.write("\n")
        except OSError as e:
            sys.stderr.write(f"bench: full record not written ({e})\n")
            full_path = None
        os.write(json_fd, (compact_line(out, full_path) + "\n").encode())
    if world > 1 or (dist.is_available() and dist.is_initialized()):
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
