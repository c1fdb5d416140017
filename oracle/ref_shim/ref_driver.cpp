// C driver around the REAL reference classes (compiled from /root/reference by build_ref.sh):
// kimera::SemanticTsdfIntegratorFactory::create(...) -> integratePointCloud(...), exposing the
// resulting Layers so tests can compare them with the oracle's restatement bit for bit.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include <kimera_semantics/color.h>
#include <kimera_semantics/semantic_tsdf_integrator_factory.h>
#include <kimera_semantics/semantic_tsdf_integrator_merged.h>
#include <kimera_semantics/semantic_voxel.h>

namespace vxb = voxblox;

struct kr_ctx {
  std::unique_ptr<vxb::Layer<vxb::TsdfVoxel>> tsdf_layer;
  std::unique_ptr<vxb::Layer<kimera::SemanticVoxel>> semantic_layer;
  std::unique_ptr<vxb::TsdfIntegratorBase> integrator;
  int vps;
  bool merged;
};

namespace {
// FastSemanticTsdfIntegrator::integratePointCloud counts frames in a FUNCTION-STATIC `reset_counter`
// ([K:src/semantic_tsdf_integrator_fast.cpp:165-170]): it outlives every integrator of the process, so what a kr_ctx with
// clear_checks_every_n_frames > 1 computes would depend on how many frames earlier contexts of the same process integrated
// (k mod n of them are still on the counter).  A checker must not depend on the order its tests run in: before a context is
// handed out, ONE empty frame through a throw-away `fast` integrator with clear_checks_every_n_frames = 1 takes the branch
// `++reset_counter >= 1` and leaves the counter at 0 — the state of a fresh process.
void realign_static_reset_counter(const std::shared_ptr<kimera::SemanticLabel2Color>& label_to_color) {
  vxb::Layer<vxb::TsdfVoxel> tsdf_layer(0.1f, 8);
  vxb::Layer<kimera::SemanticVoxel> semantic_layer(0.1f, 8);
  vxb::TsdfIntegratorBase::Config cfg;
  cfg.integrator_threads = 1;
  cfg.clear_checks_every_n_frames = 1;
  kimera::SemanticIntegratorBase::SemanticConfig sc;
  sc.semantic_label_to_color_ = label_to_color;
  auto scratch = kimera::SemanticTsdfIntegratorFactory::create(std::string("fast"), cfg, sc, &tsdf_layer, &semantic_layer);
  scratch->integratePointCloud(vxb::Transformation(), vxb::Pointcloud(), vxb::Colors(), false);
}
}  // namespace

extern "C" {

// label_csv: path of a "name,red,green,blue,alpha,id" file (SemanticLabel2Color input)
kr_ctx* kr_create(const char* method, float voxel_size, int vps, float truncation, float max_ray, float p_match,
                  int color_mode, const unsigned char* dynamic_labels, int n_dynamic, int threads,
                  int max_consecutive_ray_collisions, const char* order_mode, const char* label_csv,
                  const char* extra /* "key=value;key=value" overrides of TsdfIntegratorBase::Config */) {
  auto* c = new kr_ctx();
  c->vps = vps;
  c->merged = std::string(method) == "merged";
  c->tsdf_layer.reset(new vxb::Layer<vxb::TsdfVoxel>(voxel_size, vps));
  c->semantic_layer.reset(new vxb::Layer<kimera::SemanticVoxel>(voxel_size, vps));
  vxb::TsdfIntegratorBase::Config cfg;
  cfg.default_truncation_distance = truncation;
  cfg.max_ray_length_m = max_ray;
  cfg.integrator_threads = threads;
  cfg.max_consecutive_ray_collisions = max_consecutive_ray_collisions;
  cfg.integration_order_mode = order_mode;
  if (extra) {
    std::string e(extra);
    size_t pos = 0;
    while (pos < e.size()) {
      size_t semi = e.find(';', pos);
      if (semi == std::string::npos) semi = e.size();
      const std::string kv = e.substr(pos, semi - pos);
      pos = semi + 1;
      const size_t eq = kv.find('=');
      if (eq == std::string::npos) continue;
      const std::string k = kv.substr(0, eq);
      const double v = std::atof(kv.substr(eq + 1).c_str());
      if (k == "max_weight") cfg.max_weight = v;
      else if (k == "voxel_carving_enabled") cfg.voxel_carving_enabled = v != 0;
      else if (k == "min_ray_length_m") cfg.min_ray_length_m = v;
      else if (k == "use_const_weight") cfg.use_const_weight = v != 0;
      else if (k == "allow_clear") cfg.allow_clear = v != 0;
      else if (k == "use_weight_dropoff") cfg.use_weight_dropoff = v != 0;
      else if (k == "use_sparsity_compensation_factor") cfg.use_sparsity_compensation_factor = v != 0;
      else if (k == "sparsity_compensation_factor") cfg.sparsity_compensation_factor = v;
      else if (k == "enable_anti_grazing") cfg.enable_anti_grazing = v != 0;
      else if (k == "start_voxel_subsampling_factor") cfg.start_voxel_subsampling_factor = v;
      else if (k == "clear_checks_every_n_frames") cfg.clear_checks_every_n_frames = (int)v;
      else LOG(FATAL) << "unknown config key " << k;
    }
  }
  kimera::SemanticIntegratorBase::SemanticConfig sc;
  sc.semantic_measurement_probability_ = p_match;
  sc.color_mode = static_cast<kimera::ColorMode>(color_mode);
  sc.semantic_label_to_color_ = std::make_shared<kimera::SemanticLabel2Color>(std::string(label_csv));
  for (int i = 0; i < n_dynamic; ++i) sc.dynamic_labels_.push_back(dynamic_labels[i]);
  realign_static_reset_counter(sc.semantic_label_to_color_);
  c->integrator = kimera::SemanticTsdfIntegratorFactory::create(std::string(method), cfg, sc, c->tsdf_layer.get(),
                                                                c->semantic_layer.get());
  return c;
}

void kr_destroy(kr_ctx* c) { delete c; }
// What vxb::TsdfServer::clear() does to the map the integrator writes: the TSDF blocks go, the semantic layer and the
// integrator stay (tests of integration/server.patch's clear()).
void kr_clear_tsdf_layer(kr_ctx* c) { c->tsdf_layer->removeAllBlocks(); }

// Which permutation the shim's MixedThreadSafeIndex produces (0 = upstream as published, 1 = 1024 groups; see
// voxblox/integrator/integrator_utils.h).  Process-wide; read when an index is constructed, i.e. per frame.
void kr_set_mixed_order_form(int form) { voxblox::shim_mixed_order_form() = form; }
int kr_get_mixed_order_form() { return voxblox::shim_mixed_order_form(); }
// The sequence ThreadSafeIndexFactory::get("mixed", cloud of n points) hands out (what the adapter's probe reads).
void kr_mixed_sequence(size_t n, size_t* out) {
  vxb::Pointcloud pts(n, vxb::Point(0.f, 0.f, 1.f));
  std::unique_ptr<vxb::ThreadSafeIndex> index_getter(vxb::ThreadSafeIndexFactory::get("mixed", pts));
  size_t idx, k = 0;
  while (index_getter->getNextIndex(&idx)) out[k++] = idx;
}

// The virtual the server calls: labels come from the colours through the CSV map.
void kr_integrate(kr_ctx* c, const float* T, const float* xyz, const unsigned char* rgba, size_t n, int freespace) {
  vxb::Transformation T_G_C(T[0], T[1], T[2], T[3], vxb::Point(T[4], T[5], T[6]));
  vxb::Pointcloud pts(n);
  vxb::Colors cols(n);
  for (size_t i = 0; i < n; ++i) {
    pts[i] = vxb::Point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    cols[i] = vxb::Color(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2], rgba[4 * i + 3]);
  }
  c->integrator->integratePointCloud(T_G_C, pts, cols, freespace != 0);
}

size_t kr_num_blocks(kr_ctx* c) { return c->tsdf_layer->getNumberOfAllocatedBlocks(); }
size_t kr_num_semantic_blocks(kr_ctx* c) { return c->semantic_layer->getNumberOfAllocatedBlocks(); }

void kr_block_indices(kr_ctx* c, int* out) {
  vxb::BlockIndexList l;
  c->tsdf_layer->getAllAllocatedBlocks(&l);
  size_t k = 0;
  for (const auto& b : l) {
    out[3 * k] = b.x(); out[3 * k + 1] = b.y(); out[3 * k + 2] = b.z();
    ++k;
  }
}

// tsdf_out: vps^3 * 12 B, sem_out: vps^3 * 92 B (label, 3 pad, priors[21], rgba)
int kr_get_block(kr_ctx* c, const int* idx, unsigned char* tsdf_out, unsigned char* sem_out) {
  const vxb::BlockIndex b(idx[0], idx[1], idx[2]);
  auto tb = c->tsdf_layer->getBlockPtrByIndex(b);
  auto sb = c->semantic_layer->getBlockPtrByIndex(b);
  if (!tb || !sb) return 1;
  const size_t nv = static_cast<size_t>(c->vps) * c->vps * c->vps;
  for (size_t i = 0; i < nv; ++i) {
    const vxb::TsdfVoxel& v = tb->getVoxelByLinearIndex(i);
    std::memcpy(tsdf_out + 12 * i, &v.distance, 4);
    std::memcpy(tsdf_out + 12 * i + 4, &v.weight, 4);
    tsdf_out[12 * i + 8] = v.color.r; tsdf_out[12 * i + 9] = v.color.g; tsdf_out[12 * i + 10] = v.color.b; tsdf_out[12 * i + 11] = v.color.a;
    const kimera::SemanticVoxel& s = sb->getVoxelByLinearIndex(i);
    unsigned char* o = sem_out + 92 * i;
    std::memset(o, 0, 92);
    o[0] = s.semantic_label;
    for (int l = 0; l < 21; ++l) {
      const float p = s.semantic_priors[l];
      std::memcpy(o + 4 + 4 * l, &p, 4);
    }
    o[88] = s.color.r; o[89] = s.color.g; o[90] = s.color.b; o[91] = s.color.a;
  }
  return 0;
}

}  // extern "C"
