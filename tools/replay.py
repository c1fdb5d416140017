#!/usr/bin/env python3
"""Offline replay through the MI355X integrator — the role of the reference's kimera_semantics_rosbag executable
(kimera_semantics_ros/src/kimera_semantics_rosbag.cpp:83-141) for the path this repository accelerates: read a ROS1
bag (or generate the synthetic stand-in), compose T_G_C = T_G_B * T_B_C per depth image, integrate depth + labels on
the GPU (ks_integrate_depth), report frames/s and voxel updates/s.  Meshing / ESDF / map saving stay on the host side
of the drop-in boundary (out of scope).
  python tools/replay.py --synthetic 50 [--method merged]
  python tools/replay.py --bag demo.bag --depth-topic /tesse/depth --semantic-topic /tesse/segmentation \\
      --camera-info-topic /tesse/left_cam/camera_info --sensor-frame left_cam --label-csv cfg/tesse_multiscene_office1_segmentation_mapping.csv"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kimera_semantics_amd import binding as B  # noqa: E402
from kimera_semantics_amd import frame_source as FS  # noqa: E402
from kimera_semantics_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bag")
    ap.add_argument("--synthetic", type=int, default=0, help="number of frames of the synthetic stand-in")
    ap.add_argument("--method", default="fast", choices=["fast", "merged"])
    ap.add_argument("--depth-topic", default="/depth")
    ap.add_argument("--semantic-topic", default="/semantic")
    ap.add_argument("--camera-info-topic", default="/camera_info")
    ap.add_argument("--sensor-frame", default="left_cam")
    ap.add_argument("--base-link-frame", default="base_link_gt")
    ap.add_argument("--world-frame", default="world")
    ap.add_argument("--label-csv", help="the reference's label CSV (name,red,green,blue,alpha,id); default: the synthetic palette")
    ap.add_argument("--voxel-size", type=float, default=0.05)
    ap.add_argument("--pipeline-frames", type=int, default=4)
    a = ap.parse_args()
    if a.bag:
        seq = FS.read_rosbag(a.bag, a.depth_topic, a.semantic_topic, a.camera_info_topic, a.sensor_frame, a.base_link_frame, a.world_frame)
    else:
        seq = FS.synthetic_sequence(a.synthetic or 20)
    lut = synth.default_label_colors()
    if a.label_csv:
        import csv
        lut = np.zeros((256, 4), np.uint8)
        for row in csv.reader(open(a.label_csv)):
            try:
                lut[int(row[5])] = [int(row[1]), int(row[2]), int(row[3]), int(row[4])]
            except (ValueError, IndexError):
                continue
        lut[0] = [255, 255, 255, 255]
    h0, w0 = seq.frames[0].depth.shape
    cfg = B.default_config(method=0 if a.method == "fast" else 1, voxel_size=a.voxel_size, truncation_distance=4 * a.voxel_size,
                           semantic_measurement_probability=0.8, dynamic_labels=[20], label_rgba=lut, max_points=h0 * w0,
                           pipeline_frames=a.pipeline_frames)
    integ = B.HipIntegrator(cfg)
    integ.set_color_to_label(lut[:21], np.arange(21, dtype=np.uint8))
    upd = 0

    def acc(fr, T, st):
        nonlocal upd
        upd += st.n_voxel_updates
    t0 = time.perf_counter()
    out = FS.replay(seq, integ, use_label_img=not a.bag, on_frame=acc)
    upd += integ.flush().n_voxel_updates
    integ.synchronize()
    dt = time.perf_counter() - t0
    print(f"{out['integrated']} frames integrated ({out['skipped_no_tf']} skipped: no tf) in {dt:.3f} s: "
          f"{out['integrated'] / dt:.1f} frames/s incl. H2D of the images, {upd / dt / 1e6:.1f} M voxel updates/s, "
          f"{len(integ.block_indices())} blocks")


if __name__ == "__main__":
    main()
