"""Synthetic depth+label frame source (stand-in for the absent kimera_semantics_demo.bag).

The reference feeds the integrator with XYZRGB clouds built from a depth image and a
segmentation image used as the colour channel
(kimera_semantics_ros/include/kimera_semantics_ros/depth_map_to_pointcloud.h:213-275; launch
wiring kimera_semantics_ros/launch/kimera_semantics.launch:72-85).  The demo bag is not in
the repository (README.md:113) and there is no network, so this module renders analytic
scenes in the style of kimera_semantics_ros/src/semantic_simulation_eval.cpp:16-34
(room box + sphere + cube + cylinder + one "human" cylinder with the dynamic label 20)
and back-projects them with the same pinhole formula (depth_map_to_pointcloud.h:263-265).

Everything is seeded and deterministic (numpy PCG64).  Non-finite points are dropped, as
voxblox_ros' TsdfServer does before calling integratePointCloud.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

NUM_LABELS = 21

# label ids used by the generator
L_UNKNOWN, L_FLOOR, L_CEILING, L_WALL = 0, 3, 4, 19
L_SPHERE, L_CUBE, L_CYLINDER, L_HUMAN = 5, 7, 16, 20


def default_label_colors() -> np.ndarray:
    """A 256x4 label->RGBA table (ids 0..20 populated, alpha 255; id 0 = White as
    SemanticLabel2Color forces, kimera_semantics/src/color.cpp:64-66).  Distinct colours,
    generated arithmetically (not copied from the reference's CSV assets)."""
    t = np.zeros((256, 4), dtype=np.uint8)
    for i in range(1, NUM_LABELS):
        t[i] = ((37 * i + 11) % 256, (91 * i + 53) % 256, (173 * i + 29) % 256, 255)
    t[0] = (255, 255, 255, 255)
    return t


@dataclass
class Scene:
    room_min: np.ndarray
    room_max: np.ndarray
    sphere_c: np.ndarray
    sphere_r: float
    cube_min: np.ndarray
    cube_max: np.ndarray
    cyl_c: np.ndarray  # (x, y) centre, z from floor
    cyl_r: float
    cyl_h: float
    human_c: np.ndarray
    human_r: float = 0.25
    human_h: float = 1.7


def make_scene(kind: str = "room") -> Scene:
    """kind='room': 8x6x3 m (configs C1-C3, C5); kind='hall': 16x12x4 m (C4)."""
    if kind == "room":
        lo, hi = np.array([-4.0, -3.0, 0.0]), np.array([4.0, 3.0, 3.0])
        return Scene(lo, hi, np.array([2.0, 1.0, 0.8]), 0.8,
                     np.array([1.5, -2.2, 0.0]), np.array([2.5, -1.2, 1.0]),
                     np.array([-2.0, 1.5]), 0.4, 2.0, np.array([-1.0, -1.5]))
    if kind == "hall":
        lo, hi = np.array([-8.0, -6.0, 0.0]), np.array([8.0, 6.0, 4.0])
        return Scene(lo, hi, np.array([4.0, 2.0, 0.8]), 0.8,
                     np.array([3.0, -4.0, 0.0]), np.array([4.0, -3.0, 1.0]),
                     np.array([-4.0, 3.0]), 0.4, 2.5, np.array([-2.0, -3.0]))
    raise ValueError(kind)


def _rot_base() -> np.ndarray:
    # camera optical frame (x right, y down, z forward) -> body (x forward, y left, z up)
    return np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])


def pose_to_T(position, yaw_rad: float, pitch_rad: float = 0.0) -> np.ndarray:
    """Returns T_G_C as float32 [qw, qx, qy, qz, tx, ty, tz] (unit quaternion)."""
    cy, sy = math.cos(yaw_rad), math.sin(yaw_rad)
    cp, sp = math.cos(pitch_rad), math.sin(pitch_rad)
    Rz = np.array([[cy, -sy, 0.0], [sy, cy, 0.0], [0.0, 0.0, 1.0]])
    Ry = np.array([[cp, 0.0, sp], [0.0, 1.0, 0.0], [-sp, 0.0, cp]])
    R = Rz @ Ry @ _rot_base()
    # rotation matrix -> quaternion (Shepperd)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        w, x, y, z = 0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w, x, y, z = (R[2, 1] - R[1, 2]) / s, 0.25 * s, (R[0, 1] + R[1, 0]) / s, (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w, x, y, z = (R[0, 2] - R[2, 0]) / s, (R[0, 1] + R[1, 0]) / s, 0.25 * s, (R[1, 2] + R[2, 1]) / s
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w, x, y, z = (R[1, 0] - R[0, 1]) / s, (R[0, 2] + R[2, 0]) / s, (R[1, 2] + R[2, 1]) / s, 0.25 * s
    q = np.array([w, x, y, z], dtype=np.float64)
    q /= np.linalg.norm(q)
    return np.concatenate([q, np.asarray(position, dtype=np.float64)]).astype(np.float32)


def quat_to_R(T: np.ndarray) -> np.ndarray:
    w, x, y, z = [float(v) for v in T[:4]]
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _intersect(scene: Scene, o: np.ndarray, d: np.ndarray):
    """o: (3,), d: (N,3) world directions with camera-z component = 1 (so t == z-depth).
    Returns (t, label) of the nearest hit."""
    n = d.shape[0]
    big = 1e30
    with np.errstate(divide="ignore", invalid="ignore"):
        # room box (we are inside): exit distance per axis
        t_axes = np.where(d > 0, (scene.room_max - o) / d, np.where(d < 0, (scene.room_min - o) / d, big))
    axis = np.argmin(t_axes, axis=1)
    t_best = t_axes[np.arange(n), axis]
    label = np.full(n, L_WALL, dtype=np.uint8)
    label[(axis == 2) & (d[:, 2] < 0)] = L_FLOOR
    label[(axis == 2) & (d[:, 2] > 0)] = L_CEILING

    def take(t, lab):
        nonlocal t_best, label
        m = (t > 1e-6) & (t < t_best)
        t_best = np.where(m, t, t_best)
        label = np.where(m, np.uint8(lab), label)

    # sphere
    oc = o - scene.sphere_c
    a = np.sum(d * d, axis=1)
    b = 2 * (d @ oc)
    c = float(oc @ oc) - scene.sphere_r ** 2
    disc = b * b - 4 * a * c
    with np.errstate(invalid="ignore"):
        ts = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), big)
    take(ts, L_SPHERE)

    # cube (AABB slabs)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (scene.cube_min - o) / d
        t2 = (scene.cube_max - o) / d
    tn = np.nanmax(np.minimum(t1, t2), axis=1)
    tf = np.nanmin(np.maximum(t1, t2), axis=1)
    take(np.where((tn <= tf) & (tf > 0), tn, big), L_CUBE)

    # vertical cylinders (side surface + top cap)
    def cyl(cxy, r, h, lab):
        oc2 = o[:2] - cxy
        a2 = d[:, 0] ** 2 + d[:, 1] ** 2
        b2 = 2 * (d[:, 0] * oc2[0] + d[:, 1] * oc2[1])
        c2 = float(oc2 @ oc2) - r * r
        disc2 = b2 * b2 - 4 * a2 * c2
        with np.errstate(divide="ignore", invalid="ignore"):
            t = np.where((disc2 >= 0) & (a2 > 0), (-b2 - np.sqrt(np.maximum(disc2, 0))) / (2 * a2), big)
            z = o[2] + t * d[:, 2]
            t = np.where((z >= 0) & (z <= h), t, big)
            take(t, lab)
            tt = (h - o[2]) / d[:, 2]
            px = o[0] + tt * d[:, 0] - cxy[0]
            py = o[1] + tt * d[:, 1] - cxy[1]
            take(np.where((px * px + py * py <= r * r) & (tt > 0), tt, big), lab)

    cyl(scene.cyl_c, scene.cyl_r, scene.cyl_h, L_CYLINDER)
    cyl(scene.human_c, scene.human_r, scene.human_h, L_HUMAN)
    return t_best, label


@dataclass
class Frame:
    T_G_C: np.ndarray          # float32 [7]
    xyz: np.ndarray            # float32 [N,3] points in camera frame (finite only)
    rgba: np.ndarray           # uint8 [N,4] label colour, alpha 255
    labels: np.ndarray         # uint8 [N]
    depth: np.ndarray          # float32 [H,W] metres, NaN = invalid
    label_img: np.ndarray      # uint8 [H,W]
    K: tuple = field(default=(0.0, 0.0, 0.0, 0.0))  # fx, fy, cx, cy


def backproject(depth: np.ndarray, K) -> np.ndarray:
    """depth_map_to_pointcloud.h:263-265, float32, same evaluation order:
    x = (u - cx) * depth * (1/fx);  y = (v - cy) * depth * (1/fy);  z = depth."""
    fx, fy, cx, cy = [np.float32(v) for v in K]
    H, W = depth.shape
    u = np.arange(W, dtype=np.float32)[None, :]
    v = np.arange(H, dtype=np.float32)[:, None]
    constant_x = np.float32(np.float64(1.0) / np.float64(fx))
    constant_y = np.float32(np.float64(1.0) / np.float64(fy))
    x = ((u - cx) * depth) * constant_x
    y = ((v - cy) * depth) * constant_y
    return np.stack([x, y, depth], axis=-1).astype(np.float32)


def render_frame(scene: Scene, T_G_C: np.ndarray, width: int = 640, height: int = 480,
                 hfov_deg: float = 90.0, seed: int = 0, nan_fraction: float = 0.005,
                 label_noise: float = 0.01, label_colors: np.ndarray | None = None) -> Frame:
    rng = np.random.default_rng(seed)
    fx = fy = width / (2.0 * math.tan(math.radians(hfov_deg) / 2.0))
    cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
    K = (np.float32(fx), np.float32(fy), np.float32(cx), np.float32(cy))
    u, v = np.meshgrid(np.arange(width, dtype=np.float64), np.arange(height, dtype=np.float64))
    d_c = np.stack([(u - cx) / fx, (v - cy) / fy, np.ones_like(u)], axis=-1).reshape(-1, 3)
    R = quat_to_R(T_G_C)
    o = T_G_C[4:7].astype(np.float64)
    d_w = d_c @ R.T
    t, lab = _intersect(scene, o, d_w)
    depth = t.astype(np.float32).reshape(height, width)
    lab = lab.reshape(height, width)
    # invalid pixels and label salt noise
    inval = rng.random((height, width)) < nan_fraction
    depth = np.where(inval, np.float32(np.nan), depth).astype(np.float32)
    noise = rng.random((height, width)) < label_noise
    lab = np.where(noise, rng.integers(0, NUM_LABELS, size=(height, width), dtype=np.uint8), lab).astype(np.uint8)
    pts = backproject(depth, K).reshape(-1, 3)
    labs = lab.reshape(-1)
    ok = np.isfinite(pts).all(axis=1)
    pts, labs = np.ascontiguousarray(pts[ok]), np.ascontiguousarray(labs[ok])
    colors = default_label_colors() if label_colors is None else label_colors
    rgba = np.ascontiguousarray(colors[labs])
    return Frame(T_G_C=T_G_C.astype(np.float32), xyz=pts, rgba=rgba, labels=labs, depth=depth,
                 label_img=lab, K=K)


def trajectory_pose(k: int, radius: float = 1.5, step_m: float = 0.05, height: float = 1.5,
                    center=(0.0, 0.0)) -> np.ndarray:
    """200-frame 'bag stand-in' (SURVEY.md §8d): circle of radius 1.5 m, 0.05 m per frame,
    looking along the tangent, plus a slow extra yaw sweep of 1 deg per frame."""
    th = k * step_m / radius
    pos = (center[0] + radius * math.cos(th), center[1] + radius * math.sin(th), height)
    yaw = th + math.pi / 2 + math.radians(1.0) * k
    return pose_to_T(pos, yaw)


def arc_pose(k: int, n: int = 8, spacing: float = 0.5) -> np.ndarray:
    """C5: n poses on a 0.5 m-spaced arc looking at the same wall (>=60% frustum overlap)."""
    y = (k - (n - 1) / 2.0) * spacing
    return pose_to_T((-1.5, y, 1.5), math.radians(-2.0 * (k - (n - 1) / 2.0)))


def single_pose() -> np.ndarray:
    """C1: room centre, yaw 30 deg."""
    return pose_to_T((0.0, 0.0, 1.5), math.radians(30.0))
