// adapter_demo — drives the C++ adapter exactly as SemanticTsdfServer would: build Layers,
// create the integrator through the factory, feed colour-encoded clouds through the
// TsdfIntegratorBase virtual, then dump the host Layers.  Used by tests/test_host_adapter_gpu.py.
//   adapter_demo <method> <labels.csv> <in.bin> <out.bin> [color_mode] [max_consecutive_ray_collisions] [restart_after] [pipeline]
// pipeline = 1: the patched server's sequence — factory with default options, THEN setSyncPolicy(kOnDemand) — frames
// overlap on the GPU, the host Layers are filled by one syncLayers() at the end, as a mesh timer would.
// restart_after = k: after frame k the integrator is destroyed and a new one is created on the
// same, now non-empty, Layers (the loadMap / re-configure case): it must pick the map up from the host.
// in.bin : u32 n_frames, then per frame { f32 T[7]; u32 n; f32 xyz[3n]; u8 rgba[4n] }
// out.bin: u32 n_blocks, u32 vps, then per block { i32 idx[3]; tsdf vps^3*12 B; semantic vps^3*92 B }
#include <algorithm>
#include <chrono>
#include <voxblox/utils/timing.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "hip_semantic_tsdf_integrator.h"
#ifdef KS_DEMO_REAL_FACTORY
#include <kimera_semantics/semantic_tsdf_integrator_factory.h>
#endif

namespace vxb = voxblox;

int main(int argc, char** argv) {
#ifdef KS_DEMO_REAL_FACTORY
  // (tests) which of its two permutations the reference-side Voxblox shim's "mixed" index produces — the adapter must
  // FIND OUT, it is not told (oracle/ref_shim/voxblox/integrator/integrator_utils.h)
  if (const char* form = std::getenv("KS_DEMO_SHIM_MIXED_FORM")) vxb::shim_mixed_order_form() = std::atoi(form);
#endif
  if (argc >= 2 && std::string(argv[1]) == "--probe-order") {
    // what HipSemanticTsdfIntegrator::probeMixedOrder() reads from the ThreadSafeIndexFactory of this build (no GPU needed)
    std::printf("probeMixedOrder: %d\n", kimera::HipSemanticTsdfIntegrator::probeMixedOrder());
    return 0;
  }
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s method labels.csv in.bin out.bin [color_mode] [max_collisions]\n", argv[0]);
    return 2;
  }
  const std::string method = argv[1];
  vxb::TsdfIntegratorBase::Config cfg;
  cfg.default_truncation_distance = 0.2f;  // voxblox_ros: 4 x voxel size
  cfg.max_ray_length_m = 5.0f;
  cfg.integrator_threads = 1;  // (only the reference's CPU integrators read it: one thread = their deterministic order)
  if (argc > 6) cfg.max_consecutive_ray_collisions = std::atoi(argv[6]);
  if (const char* b = std::getenv("KS_DEMO_MAX_INTEGRATION_TIME_S")) cfg.max_integration_time_s = (float)std::atof(b);  // (tests)
  kimera::SemanticIntegratorBase::SemanticConfig sc;
  sc.semantic_measurement_probability_ = 0.8f;
  sc.color_mode = static_cast<kimera::ColorMode>(argc > 5 ? std::atoi(argv[5]) : 1);
  sc.semantic_label_to_color_ = std::make_shared<kimera::SemanticLabel2Color>(argv[2]);
  sc.dynamic_labels_.push_back(20);
  vxb::Layer<vxb::TsdfVoxel> tsdf_layer(0.05f, 16);
  vxb::Layer<kimera::SemanticVoxel> semantic_layer(0.05f, 16);
  kimera::HipSemanticTsdfIntegrator::DeviceOptions opt;
  opt.max_tiles = 4096;
  opt.max_points = 1u << 18;
  if (const char* pf = std::getenv("KS_DEMO_PIPELINE_FRAMES")) opt.pipeline_frames = std::atoi(pf);  // (tests: DeviceOptions::pipeline_frames, 0 .. 16)
  // pipeline != 0: the sequence of a server with integration/server.patch — the factory hands the integrator out with its
  // default options (strict policy), THEN the server selects kOnDemand and syncs where it reads the Layers
  const bool pipeline = argc > 8 && std::atoi(argv[8]) != 0;
#ifdef KS_DEMO_REAL_FACTORY
  // integration/build_real_kimera.sh: the REAL kimera::SemanticTsdfIntegratorFactory (reference source +
  // integration/factory.patch) hands the integrator out, as SemanticTsdfServer's constructor gets it
  // (kimera_semantics_ros/src/semantic_tsdf_server.cpp:71-78); "enum:<n>" goes through the enum overload.
  auto make = [&]() -> std::unique_ptr<vxb::TsdfIntegratorBase> {
    if (method.rfind("enum:", 0) == 0)
      return kimera::SemanticTsdfIntegratorFactory::create(static_cast<kimera::SemanticTsdfIntegratorType>(std::atoi(method.c_str() + 5)),
                                                          cfg, sc, &tsdf_layer, &semantic_layer);
    return kimera::SemanticTsdfIntegratorFactory::create(method, cfg, sc, &tsdf_layer, &semantic_layer);
  };
#else
  auto make = [&]() -> std::unique_ptr<vxb::TsdfIntegratorBase> {
    if (method.rfind("enum:", 0) == 0)
      return kimera::HipSemanticTsdfIntegratorFactory::create(std::atoi(method.c_str() + 5), cfg, sc, &tsdf_layer, &semantic_layer, opt);
    return kimera::HipSemanticTsdfIntegratorFactory::create(method, cfg, sc, &tsdf_layer, &semantic_layer, opt);
  };
#endif
  std::unique_ptr<vxb::TsdfIntegratorBase> integrator = make();
  auto on_demand = [&]() {
    if (!pipeline) return;
    if (auto* hip = dynamic_cast<kimera::HipSemanticTsdfIntegrator*>(integrator.get()))
      hip->setSyncPolicy(kimera::HipSemanticTsdfIntegrator::SyncPolicy::kOnDemand);
  };
  on_demand();
  if (auto* hip = dynamic_cast<kimera::HipSemanticTsdfIntegrator*>(integrator.get())) {
    int32_t shape[4] = {0, 0, 0, 0};
    ks_pipeline_shape(hip->context(), shape);
    std::printf("adapter_demo: pipeline shape: lag %d slots %d batch %d march streams %d\n", shape[0], shape[1], shape[2], shape[3]);
  }

  FILE* in = std::fopen(argv[3], "rb");
  if (!in) return 3;
  uint32_t n_frames = 0;
  if (std::fread(&n_frames, 4, 1, in) != 1) return 3;
  const int restart_after = argc > 7 ? std::atoi(argv[7]) : -1;
  double integrate_ms = 0.0, tail_ms = 0.0;
  uint32_t tail_frames = 0;
  for (uint32_t f = 0; f < n_frames; ++f) {
    if ((int)f == restart_after) {
      if (pipeline)   // (a server syncs before it lets go of the integrator: the Layers are what the next one starts from)
        if (auto* hip = dynamic_cast<kimera::HipSemanticTsdfIntegrator*>(integrator.get())) hip->syncLayers();
      integrator.reset();
      integrator = make();
      on_demand();
    }
    if (const char* ca = std::getenv("KS_DEMO_CLEAR_AFTER")) {
      if ((int)f == std::atoi(ca)) {
        // SemanticTsdfServer::clear() with integration/server.patch: sync, the base class removes the TSDF blocks (the
        // semantic layer and the integrator survive), the GPU map follows
        auto* hip = dynamic_cast<kimera::HipSemanticTsdfIntegrator*>(integrator.get());
        if (!hip) return 7;
        hip->syncLayers();
        tsdf_layer.removeAllBlocks();   // vxb::TsdfServer::clear()
        hip->clearDeviceMap(/*keep_integrator_state=*/true);
        hip->uploadLayers();
      }
    }
    float T[7];
    uint32_t n;
    if (std::fread(T, 4, 7, in) != 7 || std::fread(&n, 4, 1, in) != 1) return 3;
    vxb::Pointcloud pts(n);
    vxb::Colors cols(n);
    std::vector<float> xyz(3 * size_t(n));
    std::vector<uint8_t> rgba(4 * size_t(n));
    if (std::fread(xyz.data(), 4, xyz.size(), in) != xyz.size() || std::fread(rgba.data(), 1, rgba.size(), in) != rgba.size()) return 3;
    for (uint32_t i = 0; i < n; ++i) {
      pts[i] = vxb::Point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
      cols[i] = vxb::Color(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]);
    }
    const auto t0 = std::chrono::steady_clock::now();
    integrator->integratePointCloud(vxb::Transformation(T[0], T[1], T[2], T[3], vxb::Point(T[4], T[5], T[6])), pts, cols, false);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    integrate_ms += ms;
    if (std::getenv("KS_DEMO_TRACE")) std::fprintf(stderr, "frame %u done (%.3f ms, %zu blocks)\n", f, ms, tsdf_layer.getNumberOfAllocatedBlocks());
    if (3 * f >= 2 * n_frames) {  // steady state: the last third (staging buffers and most blocks exist)
      tail_ms += ms;
      ++tail_frames;
    }
  }
  std::fclose(in);
  if (pipeline) {
    auto* hip = dynamic_cast<kimera::HipSemanticTsdfIntegrator*>(integrator.get());
    if (!hip) return 7;
    const auto t0 = std::chrono::steady_clock::now();
    hip->syncLayers();
    std::printf("adapter_demo: final syncLayers %.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  if (n_frames)
    std::printf("adapter_demo: integratePointCloud %.3f ms/frame over %u frames, %.3f ms/frame over the last %u (host clouds, %s)\n",
                integrate_ms / n_frames, n_frames, tail_frames ? tail_ms / tail_frames : 0.0, tail_frames,
                pipeline ? "kOnDemand + pipeline_frames" : "kEveryFrame layer sync");

  vxb::BlockIndexList blocks;
  tsdf_layer.getAllAllocatedBlocks(&blocks);
  std::sort(blocks.begin(), blocks.end(), [](const vxb::BlockIndex& a, const vxb::BlockIndex& b) {
    if (a.x() != b.x()) return a.x() < b.x();
    if (a.y() != b.y()) return a.y() < b.y();
    return a.z() < b.z();
  });
  FILE* out = std::fopen(argv[4], "wb");
  if (!out) return 4;
  const uint32_t nb = blocks.size(), vps = 16;
  std::fwrite(&nb, 4, 1, out);
  std::fwrite(&vps, 4, 1, out);
  size_t updated = 0;
  for (const auto& b : blocks) {
    const int32_t idx[3] = {b.x(), b.y(), b.z()};
    std::fwrite(idx, 4, 3, out);
    auto tb = tsdf_layer.getBlockPtrByIndex(b);
    auto sb = semantic_layer.getBlockPtrByIndex(b);
    if (!sb) return 5;
    updated += tb->updated() && sb->updated();
    for (size_t i = 0; i < tb->num_voxels(); ++i) {
      const vxb::TsdfVoxel& v = tb->getVoxelByLinearIndex(i);
      uint8_t rec[12];
      std::memcpy(rec, &v.distance, 4);
      std::memcpy(rec + 4, &v.weight, 4);
      rec[8] = v.color.r; rec[9] = v.color.g; rec[10] = v.color.b; rec[11] = v.color.a;
      std::fwrite(rec, 1, 12, out);
    }
    for (size_t i = 0; i < sb->num_voxels(); ++i) {
      const kimera::SemanticVoxel& s = sb->getVoxelByLinearIndex(i);
      uint8_t rec[92];
      std::memset(rec, 0, 92);
      rec[0] = s.semantic_label;
      for (int l = 0; l < 21; ++l) {
        const float p = s.semantic_priors[l];
        std::memcpy(rec + 4 + 4 * l, &p, 4);
      }
      rec[88] = s.color.r; rec[89] = s.color.g; rec[90] = s.color.b; rec[91] = s.color.a;
      std::fwrite(rec, 1, 92, out);
    }
  }
  std::fclose(out);
  std::printf("adapter_demo: %u frames, %u blocks (%zu flagged updated)\n", n_frames, nb, updated);
  // what SemanticTsdfServer prints when verbose (voxblox::timing::Timing::Print): the integrator's scopes
  std::printf("timing:\n%s", vxb::timing::Timing::Print().c_str());
  return updated == nb ? 0 : 6;
}
