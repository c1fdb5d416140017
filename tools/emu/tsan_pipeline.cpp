// tools/emu/tsan_pipeline.cpp — the frame pipeline (caller thread: stages A and B; helper thread: stage T) of the library
// under ThreadSanitizer, on the host functional model: streams and events are modelled as the happens-before edges they
// stand for (hip/hip_runtime.h), so a buffer that travels between the two host threads' launches without such an edge is
// reported.  Build + run: tools/emu/run_tsan.sh      Usage: tsan_pipeline [method 0|1] [pipeline_frames] [frames] [w] [h]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ks_hip.h"

int main(int argc, char** argv) {
  const int method = argc > 1 ? atoi(argv[1]) : 0;
  const int pipe = argc > 2 ? atoi(argv[2]) : 4;
  const int frames = argc > 3 ? atoi(argv[3]) : 12;
  const int w = argc > 4 ? atoi(argv[4]) : 64, h = argc > 5 ? atoi(argv[5]) : 48;
  ks_config cfg;
  ks_default_config(&cfg);
  cfg.method = method;
  cfg.voxel_size = 0.05f;
  cfg.voxels_per_side = 16;
  cfg.truncation_distance = 0.2f;
  cfg.max_ray_length_m = 5.0f;
  cfg.max_tiles = 4096;
  cfg.max_points = (size_t)w * h;
  cfg.pipeline_frames = pipe;
  ks_ctx* c = nullptr;
  int rc = ks_create(&cfg, &c);
  if (rc) { fprintf(stderr, "ks_create: %d\n", rc); return 2; }
  const size_t n = (size_t)w * h;
  std::vector<float> xyz(3 * n);
  std::vector<uint8_t> rgba(4 * n), labels(n);
  unsigned long long updates = 0;
  for (int k = 0; k < frames; ++k) {
    // a wall seen from a sensor that moves sideways; points in the sensor frame
    for (int v = 0; v < h; ++v)
      for (int u = 0; u < w; ++u) {
        const size_t i = (size_t)v * w + u;
        const float ax = ((float)u - 0.5f * w) / (0.6f * w), ay = ((float)v - 0.5f * h) / (0.6f * w);
        const float depth = 2.0f + 0.8f * sinf(0.11f * u + 0.3f * k) + 0.4f * cosf(0.17f * v);
        xyz[3 * i] = ax * depth; xyz[3 * i + 1] = ay * depth; xyz[3 * i + 2] = depth;
        labels[i] = (uint8_t)((u / 8 + v / 8 + k) % 21);
        rgba[4 * i] = (uint8_t)(10 * labels[i]); rgba[4 * i + 1] = 7; rgba[4 * i + 2] = 99; rgba[4 * i + 3] = 255;
      }
    const float T[7] = {1.f, 0.f, 0.f, 0.f, 0.05f * k, 0.f, 0.f};   // (w, x, y, z, tx, ty, tz)
    ks_frame_stats st;
    rc = ks_integrate_points(c, T, xyz.data(), rgba.data(), labels.data(), n, 0, &st);
    if (rc) { fprintf(stderr, "frame %d: %d %s\n", k, rc, ks_last_error(c)); return 3; }
    updates += st.n_voxel_updates;
  }
  ks_frame_stats st;
  rc = ks_flush(c, &st);
  if (rc) { fprintf(stderr, "flush: %d\n", rc); return 4; }
  updates += st.n_voxel_updates;
  ks_destroy(c);
  printf("method %d pipeline %d: %d frames of %dx%d, %llu voxel updates\n", method, pipe, frames, w, h, updates);
  return 0;
}
