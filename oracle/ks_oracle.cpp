// ks_oracle.cpp — CPU ORACLE (test infrastructure; see ks_oracle.h for the contract).
//
// Dependency-free C++17 restatement of the semantic TSDF integration hot path.
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fPIC -shared (oracle/Makefile).
// Every function cites what it follows:
//   [K:file:line]  = a file under /root/reference/kimera_semantics/
//   [V:...]        = upstream ethz-asl/voxblox (NOT in /root/reference; un-pinned)
//   [M:...], [E:...] = minkindr / Eigen 3.3 (likewise external)
// Float arithmetic is f32 without FMA contraction, evaluated in the association order the
// upstream expression templates produce (Eigen's 3-element redux is  a0 + (a1 + a2) ).

#include "ks_oracle.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

// ---------------------------------------------------------------------------------------
// Types [V:core/common.h], [V:core/voxel.h], [K:semantic_voxel.h:14-27]
// ---------------------------------------------------------------------------------------
struct V3 { float x, y, z; };
struct I3 {
  int64_t x, y, z;
  bool operator==(const I3& o) const { return x == o.x && y == o.y && z == o.z; }
  bool operator!=(const I3& o) const { return !(*this == o); }
};
struct B3 {
  int32_t x, y, z;
  bool operator==(const B3& o) const { return x == o.x && y == o.y && z == o.z; }
  bool operator!=(const B3& o) const { return !(*this == o); }
};
struct Rgba { uint8_t r, g, b, a; };

constexpr float kEpsilon = 1e-6f;            // [V:core/common.h] kEpsilon
constexpr float kCoordinateEpsilon = 1e-6f;  // [V:core/common.h] kCoordinateEpsilon
constexpr float kFloatEpsilon = 1e-6f;       // [V:core/common.h] kFloatEpsilon
constexpr int kNumLabels = KO_NUM_LABELS;    // [K:common.h:26]

#pragma pack(push, 1)
struct TsdfVoxel {  // 12 B [V:core/voxel.h]
  float distance = 0.0f;
  float weight = 0.0f;
  Rgba color = {0, 0, 0, 0};
};
struct SemanticVoxel {  // 92 B [K:semantic_voxel.h:14-27]
  uint8_t semantic_label = 0;
  uint8_t pad[3] = {0, 0, 0};
  float semantic_priors[kNumLabels];
  Rgba color = {127, 127, 127, 255};  // HashableColor::Gray()
  SemanticVoxel() {
    for (int i = 0; i < kNumLabels; ++i) semantic_priors[i] = -0.60205999132f;  // [K:semantic_voxel.h:23]
  }
};
#pragma pack(pop)
static_assert(sizeof(TsdfVoxel) == 12, "TsdfVoxel layout");
static_assert(sizeof(SemanticVoxel) == 92, "SemanticVoxel layout");

// [V:core/block_hash.h] AnyIndexHash / LongIndexHash: truncated to 32 bits.
inline uint32_t index_hash(int64_t x, int64_t y, int64_t z) {
  constexpr uint64_t sl = 17191;
  constexpr uint64_t sl2 = sl * sl;
  return static_cast<uint32_t>(static_cast<uint64_t>(x) + static_cast<uint64_t>(y) * sl +
                               static_cast<uint64_t>(z) * sl2);
}
struct LongIndexHash {
  size_t operator()(const I3& i) const { return index_hash(i.x, i.y, i.z); }
};
struct AnyIndexHash {
  size_t operator()(const B3& i) const { return index_hash(i.x, i.y, i.z); }
};

// ---------------------------------------------------------------------------------------
// Small vector math in Eigen's evaluation order
// ---------------------------------------------------------------------------------------
inline V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mul(const V3& a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// [E:Core/Redux.h] redux_novec_unroller<.,.,0,3>: func(c0, func(c1, c2))
inline float dot(const V3& a, const V3& b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
inline float squared_norm(const V3& a) { return dot(a, a); }
inline float norm(const V3& a) { return std::sqrt(squared_norm(a)); }
// [E:Core/Dot.h] MatrixBase::normalized(): z>0 ? n / sqrt(z) : n
inline V3 normalized(const V3& a) {
  const float z = squared_norm(a);
  if (z > 0.0f) {
    const float s = std::sqrt(z);
    return {a.x / s, a.y / s, a.z / s};
  }
  return a;
}
inline V3 cross(const V3& a, const V3& b) {  // [E:Geometry/OrthoMethods.h]
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct Transform {  // [M:quat-transformation.h] q_A_B (w,x,y,z) + A_t_A_B
  float w;
  V3 v;
  V3 t;
};
// [M:QuatTransformationTemplate::transform] = q.rotate(p) + t ;
// [E:Geometry/Quaternion.h] _transformVector: uv = q.vec x p; uv += uv; p + w*uv + q.vec x uv
inline V3 transform_point(const Transform& T, const V3& p) {
  V3 uv = cross(T.v, p);
  uv = add(uv, uv);
  const V3 c2 = cross(T.v, uv);
  V3 r;
  r.x = (p.x + T.w * uv.x) + c2.x;
  r.y = (p.y + T.w * uv.y) + c2.y;
  r.z = (p.z + T.w * uv.z) + c2.z;
  return add(r, T.t);
}

// [V:core/common.h] getGridIndexFromPoint(point, grid_size_inv)
inline I3 grid_index_from_point(const V3& p, float inv) {
  return {static_cast<int64_t>(std::floor(p.x * inv + kCoordinateEpsilon)),
          static_cast<int64_t>(std::floor(p.y * inv + kCoordinateEpsilon)),
          static_cast<int64_t>(std::floor(p.z * inv + kCoordinateEpsilon))};
}
// [V:core/common.h] getGridIndexFromPoint(scaled_point)
inline I3 grid_index_from_scaled_point(const V3& p) {
  return {static_cast<int64_t>(std::floor(p.x + kCoordinateEpsilon)),
          static_cast<int64_t>(std::floor(p.y + kCoordinateEpsilon)),
          static_cast<int64_t>(std::floor(p.z + kCoordinateEpsilon))};
}
// [V:core/common.h] getCenterPointFromGridIndex: (float(i) + 0.5) * grid_size, evaluated in
// double (0.5 is a double literal) and rounded to float on construction of the Point.
inline V3 center_point_from_grid_index(const I3& i, float grid_size) {
  return {static_cast<float>((static_cast<double>(static_cast<float>(i.x)) + 0.5) * static_cast<double>(grid_size)),
          static_cast<float>((static_cast<double>(static_cast<float>(i.y)) + 0.5) * static_cast<double>(grid_size)),
          static_cast<float>((static_cast<double>(static_cast<float>(i.z)) + 0.5) * static_cast<double>(grid_size))};
}
// [V:core/common.h] getBlockIndexFromGlobalVoxelIndex: floor(float(v) * vps_inv)
inline B3 block_index_from_global_voxel_index(const I3& v, float vps_inv) {
  return {static_cast<int32_t>(std::floor(static_cast<float>(v.x) * vps_inv)),
          static_cast<int32_t>(std::floor(static_cast<float>(v.y) * vps_inv)),
          static_cast<int32_t>(std::floor(static_cast<float>(v.z) * vps_inv))};
}
// [V:core/common.h] getLocalFromGlobalVoxelIndex: v & (vps-1) per axis (vps power of two)
inline size_t local_linear_index(const I3& v, int vps) {
  const int64_t m = vps - 1;
  const int64_t lx = v.x & m, ly = v.y & m, lz = v.z & m;
  return static_cast<size_t>(lx + vps * (ly + vps * lz));  // [V:core/block_inl.h]
}

// [V:core/color.h] Color::blendTwoColors
inline Rgba blend_two_colors(const Rgba& c1, float w1, const Rgba& c2, float w2) {
  const float total = w1 + w2;
  w1 /= total;
  w2 /= total;
  Rgba o;
  o.r = static_cast<uint8_t>(std::round(c1.r * w1 + c2.r * w2));
  o.g = static_cast<uint8_t>(std::round(c1.g * w1 + c2.g * w2));
  o.b = static_cast<uint8_t>(std::round(c1.b * w1 + c2.b * w2));
  o.a = static_cast<uint8_t>(std::round(c1.a * w1 + c2.a * w2));
  return o;
}

// [V:core/color.h] rainbowColorMap(double h)
inline Rgba rainbow_color_map(double h) {
  Rgba c;
  c.a = 255;
  const double s = 1.0, v = 1.0;
  h -= std::floor(h);
  h *= 6;
  const int i = static_cast<int>(std::floor(h));
  double f = h - i;
  if (!(i & 1)) f = 1 - f;
  const double m = v * (1 - s);
  const double n = v * (1 - s * f);
  switch (i) {
    case 6:
    case 0: c.r = static_cast<uint8_t>(255 * v); c.g = static_cast<uint8_t>(255 * n); c.b = static_cast<uint8_t>(255 * m); break;
    case 1: c.r = static_cast<uint8_t>(255 * n); c.g = static_cast<uint8_t>(255 * v); c.b = static_cast<uint8_t>(255 * m); break;
    case 2: c.r = static_cast<uint8_t>(255 * m); c.g = static_cast<uint8_t>(255 * v); c.b = static_cast<uint8_t>(255 * n); break;
    case 3: c.r = static_cast<uint8_t>(255 * m); c.g = static_cast<uint8_t>(255 * n); c.b = static_cast<uint8_t>(255 * v); break;
    case 4: c.r = static_cast<uint8_t>(255 * n); c.g = static_cast<uint8_t>(255 * m); c.b = static_cast<uint8_t>(255 * v); break;
    case 5: c.r = static_cast<uint8_t>(255 * v); c.g = static_cast<uint8_t>(255 * m); c.b = static_cast<uint8_t>(255 * n); break;
    default: c.r = 255; c.g = 127; c.b = 127; break;
  }
  return c;
}

// ---------------------------------------------------------------------------------------
// RayCaster [V:integrator/integrator_utils.{h,cc}]
// ---------------------------------------------------------------------------------------
inline int signum(float v) { return (0.0f < v) - (v < 0.0f); }

struct RayCaster {
  I3 curr;
  int sign[3];
  float t_to_next[3];
  float t_step[3];
  int64_t current_step = 0;
  int64_t length_in_steps = 0;

  RayCaster(const V3& origin, const V3& point_G, bool is_clearing, bool carving,
            float max_ray_length_m, float voxel_size_inv, float truncation,
            bool cast_from_origin = true) {
    const V3 d = sub(point_G, origin);
    const V3 unit_ray = normalized(d);
    V3 ray_start, ray_end;
    if (is_clearing) {
      float ray_length = norm(d);
      ray_length = std::min(std::max(ray_length - truncation, 0.0f), max_ray_length_m);
      ray_end = add(origin, mul(unit_ray, ray_length));
      ray_start = carving ? origin : ray_end;
    } else {
      ray_end = add(point_G, mul(unit_ray, truncation));
      ray_start = carving ? origin : sub(point_G, mul(unit_ray, truncation));
    }
    const V3 start_scaled = mul(ray_start, voxel_size_inv);
    const V3 end_scaled = mul(ray_end, voxel_size_inv);
    if (cast_from_origin) setup(start_scaled, end_scaled);
    else setup(end_scaled, start_scaled);
  }

  void setup(const V3& start_scaled, const V3& end_scaled) {
    if (std::isnan(start_scaled.x) || std::isnan(start_scaled.y) || std::isnan(start_scaled.z) ||
        std::isnan(end_scaled.x) || std::isnan(end_scaled.y) || std::isnan(end_scaled.z)) {
      length_in_steps = 0;
      // upstream leaves curr_index_ uninitialised here and still emits one index; the
      // oracle pins it to 0 (callers never pass NaN points: they are dropped upstream of
      // integratePointCloud).
      curr = {0, 0, 0};
      sign[0] = sign[1] = sign[2] = 0;
      t_to_next[0] = t_to_next[1] = t_to_next[2] = 0.f;
      t_step[0] = t_step[1] = t_step[2] = 0.f;
      return;
    }
    curr = grid_index_from_scaled_point(start_scaled);
    const I3 end_index = grid_index_from_scaled_point(end_scaled);
    const int64_t dx = end_index.x - curr.x, dy = end_index.y - curr.y, dz = end_index.z - curr.z;
    current_step = 0;
    length_in_steps = std::abs(dx) + std::abs(dy) + std::abs(dz);
    const V3 ray_scaled = sub(end_scaled, start_scaled);
    sign[0] = signum(ray_scaled.x);
    sign[1] = signum(ray_scaled.y);
    sign[2] = signum(ray_scaled.z);
    const float corr[3] = {static_cast<float>(std::max(0, sign[0])), static_cast<float>(std::max(0, sign[1])),
                           static_cast<float>(std::max(0, sign[2]))};
    const V3 shifted = {start_scaled.x - static_cast<float>(curr.x), start_scaled.y - static_cast<float>(curr.y),
                        start_scaled.z - static_cast<float>(curr.z)};
    const float dist[3] = {corr[0] - shifted.x, corr[1] - shifted.y, corr[2] - shifted.z};
    const float rs[3] = {ray_scaled.x, ray_scaled.y, ray_scaled.z};
    for (int k = 0; k < 3; ++k) {
      // upstream: (std::abs(r) < 0.0) ? 2.0 : dist / r  — the guard is dead code, so a zero
      // component divides by zero (inf / NaN) exactly like upstream.
      t_to_next[k] = dist[k] / rs[k];
      t_step[k] = static_cast<float>(sign[k]) / rs[k];
    }
  }

  bool next(I3* out) {
    if (current_step++ > length_in_steps) return false;
    *out = curr;
    // [E:Core/Visitor.h] minCoeff(&idx): first strict minimum, NaN never replaces.
    int k = 0;
    float m = t_to_next[0];
    if (t_to_next[1] < m) { k = 1; m = t_to_next[1]; }
    if (t_to_next[2] < m) { k = 2; }
    if (k == 0) curr.x += sign[0];
    else if (k == 1) curr.y += sign[1];
    else curr.z += sign[2];
    t_to_next[k] += t_step[k];
    return true;
  }
};

// ---------------------------------------------------------------------------------------
// ApproxHashSet<20, 10000, GlobalIndex, LongIndexHash> [V:utils/approx_hash_array.h]
// ---------------------------------------------------------------------------------------
struct ApproxHashSet {
  static constexpr size_t kBits = 20;            // [K:semantic_tsdf_integrator_fast.h:102]
  static constexpr size_t kFullReset = 10000;    // [K:semantic_tsdf_integrator_fast.h:107]
  static constexpr size_t kMask = (size_t(1) << kBits) - 1;
  std::vector<std::atomic<size_t>> slots;
  size_t offset = 0;
  ApproxHashSet() : slots(size_t(1) << kBits) {
    for (auto& s : slots) s.store(0, std::memory_order_relaxed);
    slots[offset].store(std::numeric_limits<size_t>::max(), std::memory_order_relaxed);
  }
  // returns true if the hash was NOT present (and is now stored).  The slot index is
  // offset-dependent, the stored value is the bare hash: an entry left by an earlier frame
  // (different offset) can therefore never match, which is what makes the cheap reset
  // (++offset) equivalent to clearing.  Side effect kept as upstream: a
  // zero-initialised slot matches hash 0 once offset > 0.
  bool replace_hash(size_t hash) {
    const size_t idx = (hash + offset) & kMask;
    if (slots[idx].load(std::memory_order_relaxed) == hash) return false;
    slots[idx].store(hash, std::memory_order_relaxed);
    return true;
  }
  void reset() {
    if (++offset >= kFullReset) {
      for (auto& s : slots) s.store(0, std::memory_order_relaxed);
      offset = 0;
      slots[offset].store(std::numeric_limits<size_t>::max(), std::memory_order_relaxed);
    }
  }
};

// ---------------------------------------------------------------------------------------
// Layer / Block [V:core/layer.h, core/block.h]
// ---------------------------------------------------------------------------------------
template <typename Voxel>
struct Block {
  std::vector<Voxel> voxels;
  bool updated = false;
  explicit Block(int vps) : voxels(static_cast<size_t>(vps) * vps * vps) {}
};
template <typename Voxel>
struct Layer {
  using BlockPtr = std::shared_ptr<Block<Voxel>>;
  using Map = std::unordered_map<B3, BlockPtr, AnyIndexHash>;
  Map blocks;
  BlockPtr get(const B3& idx) const {
    auto it = blocks.find(idx);
    return it == blocks.end() ? nullptr : it->second;
  }
};

// [V:integrator/integrator_utils.h] ThreadSafeIndex ("mixed" / "sorted")
struct IndexGetter {
  std::atomic<size_t> atomic_idx{0};
  size_t n = 0;
  bool sorted = false;
  std::vector<size_t> sorted_indices;
  int mode_ = KO_ORDER_MIXED;
  static constexpr size_t kStep = 1024;

  // "mixed": Voxblox's MixedThreadSafeIndex::getNextIndexImpl.  Voxblox is not in /root/reference (un-pinned upstream),
  // so both readings of it exist (ks_oracle.h: KO_ORDER_MIXED = upstream as published, KO_ORDER_MIXED_1024_GROUPS = what
  // this repository assumed until round 4); the shim behind the real Kimera sources has the same switch.
  static size_t mixed_index(size_t s, size_t n, int mode) {
    const size_t q = n / kStep;
    if (q * kStep <= s) return s;
    if (mode == KO_ORDER_MIXED_1024_GROUPS) return (s % kStep) * q + s / kStep;
    return (s % q) * kStep + s / q;   // group_num * step_size_ + position_in_group, number_of_groups_ = q
  }
  // the groups of the order = the chains of the ordered-phase schedule (position s: chain s % chains, generation
  // s / chains); a frame of fewer than 1024 points (identity order) and the sorted order are cut into 1024 chains
  static uint32_t mixed_chains(size_t n, int mode) {
    const size_t q = n / kStep;
    return (mode == KO_ORDER_MIXED && q >= 1) ? (uint32_t)q : (uint32_t)kStep;
  }
  void init(int mode, const float* xyz, size_t num) {
    n = num;
    mode_ = mode;
    sorted = (mode == KO_ORDER_SORTED);
    if (sorted) {
      // upstream uses std::sort on (idx, squaredNorm) by norm; ties are unspecified
      // there — the oracle breaks ties by ascending index (stable).
      std::vector<std::pair<float, size_t>> v(num);
      for (size_t i = 0; i < num; ++i) {
        const V3 p = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        v[i] = {squared_norm(p), i};
      }
      std::stable_sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
      sorted_indices.resize(num);
      for (size_t i = 0; i < num; ++i) sorted_indices[i] = v[i].second;
    }
  }
  bool next(size_t* idx) {
    const size_t s = atomic_idx.fetch_add(1);
    if (s >= n) return false;
    *idx = sorted ? sorted_indices[s] : mixed_index(s, n, mode_);
    return true;
  }
};

}  // namespace

// ---------------------------------------------------------------------------------------
// The integrator context
// ---------------------------------------------------------------------------------------
struct ko_ctx {
  ko_config cfg;
  std::string err;

  // cached geometry [V:TsdfIntegratorBase::setLayer], [K:semantic_integrator_base.cpp:78-91]
  float voxel_size, block_size, voxel_size_inv, vps_inv;
  int vps;

  Layer<TsdfVoxel> tsdf_layer;
  Layer<SemanticVoxel> semantic_layer;
  Layer<TsdfVoxel>::Map temp_tsdf;
  Layer<SemanticVoxel>::Map temp_sem;
  std::mutex temp_tsdf_mutex, temp_sem_mutex;
  std::vector<std::mutex> mutexes{4096};  // ApproxHashArray<12, std::mutex, ...> [K:semantic_integrator_base.h:64-66]

  float log_match, log_non_match;
  float L[kNumLabels][kNumLabels];  // L[i][j]: row i = voxel label, column j = measured label

  ApproxHashSet start_voxel_set, voxel_observed_set;
  int64_t reset_counter = 0;  // function-static in the reference (fast.cpp:165); per-context here

  std::atomic<uint64_t> n_updates{0};
  std::atomic<uint64_t> n_rays{0};
  std::atomic<uint64_t> n_valid{0};

  // ---- [K:semantic_integrator_base.cpp:93-128] setSemanticProbabilities ----
  bool set_semantic_probabilities() {
    const float match = cfg.semantic_measurement_probability;
    const float non_match = 1.0f - cfg.semantic_measurement_probability;
    if (!(match > 0.0f) || !(non_match > 0.0f) || !(match < 1.0f) || !(non_match < 1.0f)) {
      err = "semantic_measurement_probability must be in (0,1)";
      return false;
    }
    log_match = std::log(match);          // std::log(float) -> logf
    log_non_match = std::log(non_match);
    if (!(log_match > log_non_match)) {
      err = "Your probabilities do not make sense (log p <= log(1-p))";
      return false;
    }
    for (int i = 0; i < kNumLabels; ++i)
      for (int j = 0; j < kNumLabels; ++j) L[i][j] = (i == j) ? log_match : log_non_match;
    for (int i = 0; i < kNumLabels; ++i) L[i][0] = 0.0f;  // .col(kUnknownSemanticLabelId).setZero()
    return true;
  }

  // ---- [V:tsdf_integrator.cc] ----
  bool is_point_valid(const V3& p, bool freespace, bool* is_clearing) const {
    const float ray_distance = norm(p);
    if (ray_distance < cfg.min_ray_length_m) return false;
    if (ray_distance > cfg.max_ray_length_m) {
      if (cfg.allow_clear || freespace) {
        *is_clearing = true;
        return true;
      }
      return false;
    }
    *is_clearing = freespace;
    return true;
  }
  float get_voxel_weight(const V3& p) const {
    if (cfg.use_const_weight) return 1.0f;
    const float dist_z = std::abs(p.z);
    if (dist_z > kEpsilon) return 1.0f / (dist_z * dist_z);
    return 0.0f;
  }
  bool is_semantic_label_valid(uint8_t label) const {  // [K:semantic_integrator_base.h:170-175]
    for (int i = 0; i < cfg.n_dynamic_labels; ++i)
      if (cfg.dynamic_labels[i] == label) return false;
    return true;
  }
  std::mutex& mutex_for(const I3& v) { return mutexes[index_hash(v.x, v.y, v.z) & 4095]; }

  static float compute_distance(const V3& origin, const V3& point_G, const V3& voxel_center) {
    const V3 v_voxel_origin = sub(voxel_center, origin);
    const V3 v_point_origin = sub(point_G, origin);
    const float dist_G = norm(v_point_origin);
    const float dist_G_V = dot(v_voxel_origin, v_point_origin) / dist_G;
    return dist_G - dist_G_V;
  }

  // [V:tsdf_integrator.cc] TsdfIntegratorBase::updateTsdfVoxel (lock taken by caller)
  static void update_tsdf_voxel_nolock(const ko_config& c, float voxel_size, const V3& origin,
                                       const V3& point_G, const I3& v, const Rgba& color,
                                       float weight, TsdfVoxel* voxel) {
    const V3 center = center_point_from_grid_index(v, voxel_size);
    const float sdf = compute_distance(origin, point_G, center);
    float updated_weight = weight;
    const float dropoff_epsilon = voxel_size;
    if (c.use_weight_dropoff && sdf < -dropoff_epsilon) {
      updated_weight = weight * (c.truncation_distance + sdf) / (c.truncation_distance - dropoff_epsilon);
      updated_weight = std::max(updated_weight, 0.0f);
    }
    if (c.use_sparsity_compensation_factor) {
      if (std::abs(sdf) < c.truncation_distance) updated_weight *= c.sparsity_compensation_factor;
    }
    const float new_weight = voxel->weight + updated_weight;
    if (new_weight < kFloatEpsilon) return;
    const float new_sdf = (sdf * updated_weight + voxel->distance * voxel->weight) / new_weight;
    if (std::abs(sdf) < c.truncation_distance) {
      voxel->color = blend_two_colors(voxel->color, voxel->weight, color, updated_weight);
    }
    voxel->distance = (new_sdf > 0.0f) ? std::min(c.truncation_distance, new_sdf)
                                        : std::max(-c.truncation_distance, new_sdf);
    voxel->weight = std::min(c.max_weight, new_weight);
  }

  // [K:semantic_integrator_base.cpp:136-194] updateSemanticVoxel (lock taken by caller)
  void update_semantic_voxel_nolock(const float* freq, TsdfVoxel* tsdf_voxel, SemanticVoxel* sv) const {
    // [K:...:283-314] priors += L * freq.  Eigen's fixed-size GEMV association is
    // version-dependent; the oracle pins it: j ascending, no FMA,
    // product evaluated into a temporary and then added.
    for (int i = 0; i < kNumLabels; ++i) {
      float acc = 0.0f;
      for (int j = 0; j < kNumLabels; ++j) acc += L[i][j] * freq[j];
      sv->semantic_priors[i] += acc;
    }
    // [K:...:352-367] maxCoeff: first strict maximum
    int best = 0;
    float m = sv->semantic_priors[0];
    for (int i = 1; i < kNumLabels; ++i)
      if (sv->semantic_priors[i] > m) { m = sv->semantic_priors[i]; best = i; }
    sv->semantic_label = static_cast<uint8_t>(best);
    // [K:...:370-380] label -> colour
    const uint8_t* c = cfg.label_rgba[sv->semantic_label];
    sv->color = {c[0], c[1], c[2], c[3]};
    switch (cfg.color_mode) {  // [K:...:174-191]
      case KO_COLOR_MODE_COLOR: break;
      case KO_COLOR_MODE_SEMANTIC: tsdf_voxel->color = sv->color; break;
      case KO_COLOR_MODE_SEMANTIC_PROBABILITY:
        tsdf_voxel->color = rainbow_color_map(std::exp(sv->semantic_priors[sv->semantic_label]));
        break;
      default: break;
    }
  }

  // [V:tsdf_integrator.cc allocateStorageAndGetVoxelPtr] / [K:semantic_integrator_base.cpp:205-254]
  template <typename Voxel>
  Voxel* allocate_and_get(const I3& v, Layer<Voxel>& layer, typename Layer<Voxel>::Map& temp,
                          std::mutex& temp_mutex, std::shared_ptr<Block<Voxel>>* last_block,
                          B3* last_idx) {
    const B3 block_idx = block_index_from_global_voxel_index(v, vps_inv);
    if (block_idx != *last_idx || *last_block == nullptr) {
      *last_block = layer.get(block_idx);
      *last_idx = block_idx;
    }
    if (*last_block == nullptr) {
      std::lock_guard<std::mutex> lock(temp_mutex);
      auto it = temp.find(block_idx);
      if (it != temp.end()) {
        *last_block = it->second;
      } else {
        auto ins = temp.emplace(block_idx, std::make_shared<Block<Voxel>>(vps));
        *last_block = ins.first->second;
      }
    }
    (*last_block)->updated = true;
    return &(*last_block)->voxels[local_linear_index(v, vps)];
  }
  uint64_t insert_temp_blocks() {  // updateLayerWithStoredBlocks + updateSemanticLayerWithStoredBlocks
    const uint64_t n = temp_tsdf.size();
    for (auto& kv : temp_tsdf) tsdf_layer.blocks.insert(kv);
    temp_tsdf.clear();
    for (auto& kv : temp_sem) semantic_layer.blocks.insert(kv);
    temp_sem.clear();
    return n;
  }

  // per-(ray,voxel) body shared by fast and merged:
  // [K:fast.cpp:124-140], [K:merged.cpp:315-327]
  struct BlockCache {
    std::shared_ptr<Block<TsdfVoxel>> block;
    B3 block_idx{0, 0, 0};
    std::shared_ptr<Block<SemanticVoxel>> sem_block;
    B3 sem_block_idx{0, 0, 0};
  };
  void update_voxel(const V3& origin, const V3& point_G, const I3& v, const Rgba& color, float weight,
                    const float* freq, BlockCache* bc) {
    TsdfVoxel* voxel = allocate_and_get<TsdfVoxel>(v, tsdf_layer, temp_tsdf, temp_tsdf_mutex, &bc->block, &bc->block_idx);
    {
      std::lock_guard<std::mutex> lock(mutex_for(v));
      update_tsdf_voxel_nolock(cfg, voxel_size, origin, point_G, v, color, weight, voxel);
    }
    SemanticVoxel* sv = allocate_and_get<SemanticVoxel>(v, semantic_layer, temp_sem, temp_sem_mutex, &bc->sem_block, &bc->sem_block_idx);
    {
      std::lock_guard<std::mutex> lock(mutex_for(v));
      update_semantic_voxel_nolock(freq, voxel, sv);
    }
  }

  // ---- fast: [K:semantic_tsdf_integrator_fast.cpp:57-143] ----
  void integrate_semantic_function(const Transform& T, const float* xyz, const uint8_t* rgba,
                                   const uint8_t* labels, bool freespace, IndexGetter* getter) {
    size_t point_idx;
    uint64_t updates = 0, rays = 0, valid = 0;
    while (getter->next(&point_idx)) {
      const V3 point_C = {xyz[3 * point_idx], xyz[3 * point_idx + 1], xyz[3 * point_idx + 2]};
      const Rgba color = rgba ? Rgba{rgba[4 * point_idx], rgba[4 * point_idx + 1], rgba[4 * point_idx + 2], rgba[4 * point_idx + 3]}
                              : Rgba{0, 0, 0, 0};
      const uint8_t label = labels[point_idx];
      bool is_clearing;
      if (!is_point_valid(point_C, freespace, &is_clearing) || !is_semantic_label_valid(label)) continue;
      ++valid;
      const V3 origin = T.t;
      const V3 point_G = transform_point(T, point_C);
      I3 gvi = grid_index_from_point(point_G, cfg.start_voxel_subsampling_factor * voxel_size_inv);
      if (!start_voxel_set.replace_hash(LongIndexHash()(gvi))) continue;
      ++rays;
      RayCaster caster(origin, point_G, is_clearing, cfg.voxel_carving_enabled != 0, cfg.max_ray_length_m,
                       voxel_size_inv, cfg.truncation_distance, /*cast_from_origin=*/false);
      int64_t consecutive = 0;
      BlockCache bc;
      while (caster.next(&gvi)) {
        if (!voxel_observed_set.replace_hash(LongIndexHash()(gvi))) ++consecutive;
        else consecutive = 0;
        if (consecutive > cfg.max_consecutive_ray_collisions) break;
        const float weight = get_voxel_weight(point_C);
        float freq[kNumLabels];
        for (int i = 0; i < kNumLabels; ++i) freq[i] = 0.0f;
        freq[label] += 1.0f;
        update_voxel(origin, point_G, gvi, color, weight, freq, &bc);
        ++updates;
      }
    }
    n_updates += updates;
    n_rays += rays;
    n_valid += valid;
  }

  // ---- fast, ORDERED-PHASE schedule (cfg.early_out_phase_growth >= 16) -------------------------
  // NOT reference code: the restatement of the schedule the HIP kernels run when the early-out of
  // [K:semantic_tsdf_integrator_fast.cpp:110-122] is enabled, so that the GPU can be checked bit for
  // bit against a CPU.  The reference's loop is inherently serial (ray k stops on marks rays 1..k-1
  // left in voxel_observed_approx_set_); the schedule keeps the dependencies that matter and cuts the rest:
  //   * integration position s -> chain c = s % chains (the "mixed" order's group, i.e. a run of
  //     neighbouring points; chains = N / 1024 in the upstream form of the order, 1024 in the other form, in sorted
  //     order and for frames of fewer than 1024 points: IndexGetter::mixed_chains) and generation g = s / chains;
  //   * generations are cut into phases [B_j, B_j+1), B_0 = 0, B_j+1 = B_j + max(1, B_j (growth-16)/16);
  //   * within a phase the chains are independent, and a chain's LIVE rays of the phase (those that
  //     survived the start-voxel dedup), taken in generation order, are cut into sub-runs of 16 that are
  //     independent too; a sub-run walks its rays in generation order;
  //   * a ray tests every voxel of its path against: the marks the PREVIOUS rays of its own sub-run made
  //     (private direct-mapped set of 1024 entries, newest (generation, step) wins an entry), else the
  //     shared set as it stood when the phase began;
  //   * when a phase ends its marks enter the shared set: per slot the mark of the highest
  //     (position, hash) wins (the reference: the last writer in serial order);
  //   * a slot nothing has been written to never matches (the reference's zero-initialised slots
  //     "contain" hash 0: that one-voxel artefact is not reproduced; marks are stored with a flag bit).
  // Start-voxel dedup, the ray caster, the consecutive-collision rule and the per-voxel update order
  // (integration position) are the reference's.  growth 16 = one generation per phase.
  static constexpr uint32_t kPrivSlots = 1024, kSubRun = 16;
  static constexpr size_t kMarkFlag = size_t(1) << 40;  // set in every mark the phased schedule stores in the shared set
  static std::vector<uint32_t> phase_bounds(uint32_t n_gen, int growth) {
    std::vector<uint32_t> b{0};
    for (;;) {
      const uint64_t inc = std::max<uint64_t>(1, (uint64_t)b.back() * (uint64_t)(growth - 16) / 16);
      if (b.back() + inc >= n_gen) break;
      b.push_back((uint32_t)(b.back() + inc));
    }
    return b;
  }
  void integrate_fast_phased(const Transform& T, const float* xyz, const uint8_t* rgba, const uint8_t* labels,
                             size_t n, bool freespace) {
    struct Ray { size_t idx; uint32_t pos; V3 pg; bool clearing; uint32_t cnt; };
    std::vector<Ray> rays;
    IndexGetter getter;
    getter.init(cfg.integration_order_mode, xyz, n);
    size_t point_idx;
    uint32_t pos = 0;
    uint64_t valid = 0;
    while (getter.next(&point_idx)) {
      const uint32_t p = pos++;
      const V3 point_C = {xyz[3 * point_idx], xyz[3 * point_idx + 1], xyz[3 * point_idx + 2]};
      bool is_clearing;
      if (!is_point_valid(point_C, freespace, &is_clearing) || !is_semantic_label_valid(labels[point_idx])) continue;
      ++valid;
      const V3 point_G = transform_point(T, point_C);
      const I3 g = grid_index_from_point(point_G, cfg.start_voxel_subsampling_factor * voxel_size_inv);
      if (!start_voxel_set.replace_hash(LongIndexHash()(g))) continue;
      rays.push_back({point_idx, p, point_G, is_clearing, 0});
    }
    const uint32_t kChains = IndexGetter::mixed_chains(n, cfg.integration_order_mode);
    const uint32_t n_gen = (uint32_t)((n + kChains - 1) / kChains);
    const std::vector<uint32_t> B = phase_bounds(n_gen, cfg.early_out_phase_growth);
    ApproxHashSet& S = voxel_observed_set;
    const int64_t lim = cfg.max_consecutive_ray_collisions;
    struct Mark { size_t slot; uint32_t pos; uint32_t hash; };
    std::vector<Mark> marks;
    std::vector<uint64_t> priv((size_t)kChains * kPrivSlots);
    std::vector<uint32_t> priv_sub(kChains);  // sub-run the chain's private set currently belongs to
    std::vector<uint32_t> live_seen(kChains);  // live rays of the chain the phase has handled so far
    const bool wave_stats = getenv("KO_WAVE_STATS") != nullptr;   // schedule research: per phase, the work of the (chain, sub-run) wavefronts -> stderr
    std::unordered_map<uint64_t, std::array<uint64_t, 3>> wave_work;
    std::vector<std::pair<uint32_t, uint64_t>> own;
    size_t r0 = 0;
    for (size_t j = 0; j < B.size() && r0 < rays.size(); ++j) {
      const uint64_t end_pos = (j + 1 < B.size()) ? (uint64_t)B[j + 1] * kChains : ~0ull;
      marks.clear();
      std::fill(priv.begin(), priv.end(), 0ull);
      std::fill(priv_sub.begin(), priv_sub.end(), 0u);
      std::fill(live_seen.begin(), live_seen.end(), 0u);
      size_t r1 = r0;
      // rays are in position order = generation-major; a chain's rays therefore appear in generation order
      for (; r1 < rays.size() && rays[r1].pos < end_pos; ++r1) {
        Ray& r = rays[r1];
        const uint32_t chain = r.pos % kChains, gen = r.pos / kChains;
        uint64_t* pv = &priv[(size_t)chain * kPrivSlots];
        const uint32_t sub = live_seen[chain]++ / kSubRun;
        if (sub != priv_sub[chain]) {  // a new sub-run starts with an empty private set
          std::fill(pv, pv + kPrivSlots, 0ull);
          priv_sub[chain] = sub;
        }
        RayCaster caster(T.t, r.pg, r.clearing, cfg.voxel_carving_enabled != 0, cfg.max_ray_length_m, voxel_size_inv,
                         cfg.truncation_distance, /*cast_from_origin=*/false);
        I3 v;
        int64_t consecutive = 0;
        uint32_t step = 0;
        own.clear();
        while (caster.next(&v)) {
          const uint32_t h = (uint32_t)LongIndexHash()(v);
          const size_t slot = ((size_t)h + S.offset) & ApproxHashSet::kMask;
          const uint64_t pe = pv[slot & (kPrivSlots - 1)];
          size_t content;
          if (pe != 0 && ((pe >> 32) & 1023u) == (slot >> 10)) content = (size_t)(pe & 0xffffffffull) | kMarkFlag;
          else content = S.slots[slot].load(std::memory_order_relaxed);
          if (content == ((size_t)h | kMarkFlag)) ++consecutive;
          else consecutive = 0;
          own.push_back({(uint32_t)(slot & (kPrivSlots - 1)),
                         ((uint64_t)gen << 52) | ((uint64_t)std::min<uint32_t>(step, 1023u) << 42) |
                             ((uint64_t)(slot >> 10) << 32) | (uint64_t)h});
          marks.push_back({slot, r.pos, h});
          ++step;
          if (consecutive > lim) break;
          ++r.cnt;
        }
        for (const auto& o : own) pv[o.first] = std::max(pv[o.first], o.second);
        if (wave_stats) {  // schedule research (KO_WAVE_STATS=1): what the (chain, sub-run) wavefront of this ray has to do
          auto& w = wave_work[((uint64_t)chain << 32) | sub];
          w[0] += 1;                                               // rays
          if (step > 16) { w[1] += 1; w[2] += (step - 16 + 63) / 64; }   // rays past their first 16 voxels, their 64-voxel rounds
        }
      }
      if (wave_stats) {
        uint64_t rays_max = 0, rounds = 0, rounds_max = 0, longs = 0, cost_max = 0;
        for (const auto& kv : wave_work) {
          rays_max = std::max(rays_max, kv.second[0]);
          longs += kv.second[1];
          rounds += kv.second[2];
          rounds_max = std::max(rounds_max, kv.second[2]);
          cost_max = std::max(cost_max, 30 + 4 * kv.second[0] + 15 * kv.second[2]);   // ~0.1 us units: fixed + per ray + per round
        }
        fprintf(stderr, "KO_WAVE_STATS phase [%u,%u): wavefronts with work %zu, rays %zu (max %llu per wavefront), long rays %llu, rounds %llu (max %llu per wavefront), "
                "slowest wavefront ~%.1f us (model: 3 + 0.4/ray + 1.5/round)\n", B[j], (j + 1 < B.size()) ? B[j + 1] : n_gen, wave_work.size(), r1 - r0,
                (unsigned long long)rays_max, (unsigned long long)longs, (unsigned long long)rounds, (unsigned long long)rounds_max, cost_max / 10.0);
        wave_work.clear();
      }
      // the phase's marks enter the shared set: per slot the highest (position, hash)
      std::sort(marks.begin(), marks.end(), [](const Mark& a, const Mark& b) {
        if (a.slot != b.slot) return a.slot < b.slot;
        if (a.pos != b.pos) return a.pos < b.pos;
        return a.hash < b.hash;
      });
      for (size_t i = 0; i < marks.size(); ++i)
        if (i + 1 == marks.size() || marks[i + 1].slot != marks[i].slot)
          S.slots[marks[i].slot].store((size_t)marks[i].hash | kMarkFlag, std::memory_order_relaxed);
      r0 = r1;
    }
    // voxel updates: every ray's first cnt voxels, rays in integration order (the order the reference's
    // single thread applies them in)
    uint64_t updates = 0;
    for (const Ray& r : rays) {
      if (r.cnt == 0) continue;
      const V3 point_C = {xyz[3 * r.idx], xyz[3 * r.idx + 1], xyz[3 * r.idx + 2]};
      const Rgba color = rgba ? Rgba{rgba[4 * r.idx], rgba[4 * r.idx + 1], rgba[4 * r.idx + 2], rgba[4 * r.idx + 3]}
                              : Rgba{0, 0, 0, 0};
      const uint8_t label = labels[r.idx];
      RayCaster caster(T.t, r.pg, r.clearing, cfg.voxel_carving_enabled != 0, cfg.max_ray_length_m, voxel_size_inv,
                       cfg.truncation_distance, false);
      I3 v;
      BlockCache bc;
      const float weight = get_voxel_weight(point_C);
      float freq[kNumLabels];
      for (int i = 0; i < kNumLabels; ++i) freq[i] = 0.0f;
      freq[label] += 1.0f;
      for (uint32_t s2 = 0; s2 < r.cnt && caster.next(&v); ++s2) {
        update_voxel(T.t, r.pg, v, color, weight, freq, &bc);
        ++updates;
      }
    }
    n_updates += updates;
    n_rays += rays.size();
    n_valid += valid;
  }

  // [K:semantic_tsdf_integrator_fast.cpp:145-199]
  void integrate_fast(const Transform& T, const float* xyz, const uint8_t* rgba, const uint8_t* labels,
                      size_t n, bool freespace) {
    if ((++reset_counter) >= cfg.clear_checks_every_n_frames) {
      reset_counter = 0;
      start_voxel_set.reset();
      voxel_observed_set.reset();
    }
    if (cfg.early_out_phase_growth >= 16) {
      integrate_fast_phased(T, xyz, rgba, labels, n, freespace);
      return;
    }
    IndexGetter getter;
    getter.init(cfg.integration_order_mode, xyz, n);
    const int threads = std::max(1, cfg.integrator_threads);
    if (threads == 1) {
      integrate_semantic_function(T, xyz, rgba, labels, freespace, &getter);
    } else {
      std::list<std::thread> pool;
      for (int i = 0; i < threads; ++i)
        pool.emplace_back(&ko_ctx::integrate_semantic_function, this, T, xyz, rgba, labels, freespace, &getter);
      for (auto& t : pool) t.join();
    }
  }

  // ---- merged ----
  using VoxelMap = std::unordered_map<I3, std::vector<size_t>, LongIndexHash>;  // [K:common.h:37]
  struct Bundle {
    I3 key;
    const std::vector<size_t>* pts;
  };

  // [V:tsdf_integrator.cc MergedTsdfIntegrator::bundleRays]
  void bundle_rays(const Transform& T, const float* xyz, bool freespace, IndexGetter* getter,
                   VoxelMap* voxel_map, VoxelMap* clear_map, std::vector<I3>* voxel_order,
                   std::vector<I3>* clear_order) {
    size_t point_idx;
    uint64_t valid = 0;
    while (getter->next(&point_idx)) {
      const V3 point_C = {xyz[3 * point_idx], xyz[3 * point_idx + 1], xyz[3 * point_idx + 2]};
      bool is_clearing;
      if (!is_point_valid(point_C, freespace, &is_clearing)) continue;
      ++valid;
      const V3 point_G = transform_point(T, point_C);
      const I3 key = grid_index_from_point(point_G, voxel_size_inv);
      VoxelMap* m = is_clearing ? clear_map : voxel_map;
      std::vector<I3>* order = is_clearing ? clear_order : voxel_order;
      auto& vec = (*m)[key];
      if (vec.empty()) order->push_back(key);  // first insertion (CANONICAL order)
      vec.push_back(point_idx);
    }
    n_valid += valid;
  }

  // [K:semantic_tsdf_integrator_merged.cpp:235-329]
  void integrate_voxel(const Transform& T, const float* xyz, const uint8_t* rgba, const uint8_t* labels,
                       bool enable_anti_grazing, bool clearing_ray, const I3& key,
                       const std::vector<size_t>& pts, const VoxelMap& voxel_map) {
    if (pts.empty()) return;
    const V3 origin = T.t;
    Rgba merged_color = {0, 0, 0, 0};
    V3 merged_point_C = {0.f, 0.f, 0.f};
    float merged_weight = 0.0f;
    float freq[kNumLabels];
    for (int i = 0; i < kNumLabels; ++i) freq[i] = 0.0f;
    for (const size_t pt_idx : pts) {
      const V3 point_C = {xyz[3 * pt_idx], xyz[3 * pt_idx + 1], xyz[3 * pt_idx + 2]};
      const Rgba color = rgba ? Rgba{rgba[4 * pt_idx], rgba[4 * pt_idx + 1], rgba[4 * pt_idx + 2], rgba[4 * pt_idx + 3]}
                              : Rgba{0, 0, 0, 0};
      const float point_weight = get_voxel_weight(point_C);
      if (point_weight < kEpsilon) continue;
      const float denom = merged_weight + point_weight;
      merged_point_C.x = (merged_point_C.x * merged_weight + point_C.x * point_weight) / denom;
      merged_point_C.y = (merged_point_C.y * merged_weight + point_C.y * point_weight) / denom;
      merged_point_C.z = (merged_point_C.z * merged_weight + point_C.z * point_weight) / denom;
      merged_color = blend_two_colors(merged_color, merged_weight, color, point_weight);
      merged_weight += point_weight;
      freq[labels[pt_idx]] += 1.0f;
      if (clearing_ray) break;
    }
    const V3 merged_point_G = transform_point(T, merged_point_C);
    RayCaster caster(origin, merged_point_G, clearing_ray, cfg.voxel_carving_enabled != 0,
                     cfg.max_ray_length_m, voxel_size_inv, cfg.truncation_distance);
    I3 gvi;
    BlockCache bc;
    uint64_t updates = 0;
    while (caster.next(&gvi)) {
      if (enable_anti_grazing) {
        if ((clearing_ray || gvi != key) && voxel_map.find(gvi) != voxel_map.end()) continue;
      }
      update_voxel(origin, merged_point_G, gvi, merged_color, merged_weight, freq, &bc);
      ++updates;
    }
    n_updates += updates;
    ++n_rays;
  }

  // [K:semantic_tsdf_integrator_merged.cpp:151-232] integrateRays/integrateVoxels
  void integrate_rays(const Transform& T, const float* xyz, const uint8_t* rgba, const uint8_t* labels,
                      bool clearing_ray, const VoxelMap& voxel_map, const VoxelMap& clear_map,
                      const std::vector<I3>& voxel_order, const std::vector<I3>& clear_order) {
    const VoxelMap& m = clearing_ray ? clear_map : voxel_map;
    std::vector<Bundle> bundles;
    bundles.reserve(m.size());
    if (cfg.bundle_order == KO_BUNDLE_ORDER_REFERENCE) {
      for (const auto& kv : m) bundles.push_back({kv.first, &kv.second});
    } else {
      const std::vector<I3>& order = clearing_ray ? clear_order : voxel_order;
      for (const I3& k : order) bundles.push_back({k, &m.find(k)->second});
    }
    const size_t threads = static_cast<size_t>(std::max(1, cfg.integrator_threads));
    auto worker = [&](size_t thread_idx) {
      for (size_t i = 0; i < bundles.size(); ++i) {
        if (((i + thread_idx + 1) % threads) == 0u) {
          integrate_voxel(T, xyz, rgba, labels, cfg.enable_anti_grazing != 0, clearing_ray,
                          bundles[i].key, *bundles[i].pts, voxel_map);
        }
      }
    };
    if (threads == 1) {
      worker(0);
    } else {
      std::list<std::thread> pool;
      for (size_t i = 0; i < threads; ++i) pool.emplace_back(worker, i);
      for (auto& t : pool) t.join();
    }
  }

  // [K:semantic_tsdf_integrator_merged.cpp:97-149]
  uint64_t integrate_merged(const Transform& T, const float* xyz, const uint8_t* rgba, const uint8_t* labels,
                            size_t n, bool freespace) {
    VoxelMap voxel_map, clear_map;
    std::vector<I3> voxel_order, clear_order;
    IndexGetter getter;
    getter.init(cfg.integration_order_mode, xyz, n);
    bundle_rays(T, xyz, freespace, &getter, &voxel_map, &clear_map, &voxel_order, &clear_order);
    integrate_rays(T, xyz, rgba, labels, false, voxel_map, clear_map, voxel_order, clear_order);
    uint64_t new_blocks = insert_temp_blocks();
    integrate_rays(T, xyz, rgba, labels, true, voxel_map, clear_map, voxel_order, clear_order);
    new_blocks += insert_temp_blocks();
    return new_blocks;
  }
};

// ---------------------------------------------------------------------------------------
// C API
// ---------------------------------------------------------------------------------------
template <typename Map>
static void sorted_indices(const Map& m, int32_t* out) {
  std::vector<B3> v;
  v.reserve(m.size());
  for (const auto& kv : m) v.push_back(kv.first);
  std::sort(v.begin(), v.end(), [](const B3& a, const B3& b) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    return a.z < b.z;
  });
  for (size_t i = 0; i < v.size(); ++i) {
    out[3 * i] = v[i].x;
    out[3 * i + 1] = v[i].y;
    out[3 * i + 2] = v[i].z;
  }
}
extern "C" {

void ko_default_config(ko_config* c) {
  std::memset(c, 0, sizeof(*c));
  // [V:TsdfIntegratorBase::Config], [V:voxblox_ros/ros_params.h]
  c->voxel_size = 0.05f;
  c->voxels_per_side = 16;
  c->truncation_distance = 4 * 0.05f;
  c->max_weight = 10000.0f;
  c->min_ray_length_m = 0.1f;
  c->max_ray_length_m = 5.0f;
  c->voxel_carving_enabled = 1;
  c->use_const_weight = 0;
  c->allow_clear = 1;
  c->use_weight_dropoff = 1;
  c->use_sparsity_compensation_factor = 0;
  c->sparsity_compensation_factor = 1.0f;
  c->enable_anti_grazing = 0;
  c->start_voxel_subsampling_factor = 2.0f;
  c->max_consecutive_ray_collisions = 2;
  c->clear_checks_every_n_frames = 1;
  c->integration_order_mode = KO_ORDER_MIXED;
  c->integrator_threads = 1;
  c->method = KO_METHOD_FAST;
  c->bundle_order = KO_BUNDLE_ORDER_REFERENCE;
  c->semantic_measurement_probability = 0.9f;  // [K:semantic_integrator_base.h:77]
  c->color_mode = KO_COLOR_MODE_SEMANTIC;      // [K:semantic_integrator_base.h:80]
  c->n_dynamic_labels = 0;
}

int ko_create(const ko_config* cfg, ko_ctx** out) {
  if (!cfg || !out) return -1;
  auto* ctx = new ko_ctx();
  ctx->cfg = *cfg;
  const int vps = cfg->voxels_per_side;
  if (vps <= 0 || (vps & (vps - 1)) != 0) {
    delete ctx;
    return -1;
  }
  ctx->vps = vps;
  ctx->voxel_size = cfg->voxel_size;
  ctx->block_size = cfg->voxel_size * static_cast<float>(vps);        // [V:core/layer.h]
  ctx->voxel_size_inv = static_cast<float>(1.0 / cfg->voxel_size);    // [V:TsdfIntegratorBase::setLayer]
  ctx->vps_inv = static_cast<float>(1.0 / static_cast<double>(vps));
  if (!ctx->set_semantic_probabilities()) {
    delete ctx;
    return -3;
  }
  *out = ctx;
  return 0;
}

void ko_destroy(ko_ctx* ctx) { delete ctx; }
const char* ko_last_error(ko_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int ko_integrate_points(ko_ctx* ctx, const float Tq[7], const float* xyz, const uint8_t* rgba,
                        const uint8_t* labels, size_t n, int freespace, ko_frame_stats* stats) {
  if (!ctx || !Tq || (n && (!xyz || !labels))) return -1;
  for (size_t i = 0; i < n; ++i) {
    if (labels[i] >= kNumLabels) {  // CHECK_LT(label, 21) [K:fast.cpp:134], [K:merged.cpp:278]
      ctx->err = "semantic label >= 21";
      return -2;
    }
  }
  Transform T;
  T.w = Tq[0];
  T.v = {Tq[1], Tq[2], Tq[3]};
  T.t = {Tq[4], Tq[5], Tq[6]};
  ctx->n_updates = 0;
  ctx->n_rays = 0;
  ctx->n_valid = 0;
  uint64_t new_blocks = 0;
  if (ctx->cfg.method == KO_METHOD_FAST) {
    ctx->integrate_fast(T, xyz, rgba, labels, n, freespace != 0);
    new_blocks = ctx->insert_temp_blocks();
  } else {
    new_blocks = ctx->integrate_merged(T, xyz, rgba, labels, n, freespace != 0);
  }
  if (stats) {
    stats->n_points = n;
    stats->n_valid_points = ctx->n_valid;
    stats->n_rays_cast = ctx->n_rays;
    stats->n_voxel_updates = ctx->n_updates;
    stats->n_blocks_allocated = new_blocks;
  }
  return 0;
}


// Design-study / test diagnostic: the voxels the `fast` integrator UPDATES for one frame on fresh
// approximate sets (first frame of a context), without touching any voxel state.
//   n_bounds == 0 : the reference's serial order, [K:semantic_tsdf_integrator_fast.cpp:57-143]
//   n_bounds  > 0 : the ORDERED-WINDOW schedule of the GPU path: integration positions are cut into
//                   windows [bounds[i], bounds[i+1]); every ray of a window tests against the
//                   observed set as it stood when the window began, then the window's marks are
//                   applied in serial order (last writer of a slot wins).
// Writes the sorted unique packed voxel keys (21 bits per axis, biased) of the updated voxels to
// out_keys (up to cap) and returns their number; *n_updates = updates with multiplicity.
size_t ko_sim_early_out(const ko_config* cfg, const float Tq[7], const float* xyz, const uint8_t* labels, size_t n,
                        const uint32_t* bounds, size_t n_bounds, uint64_t* out_keys, size_t cap, uint64_t* n_updates) {
  ko_ctx* c = nullptr;
  if (ko_create(cfg, &c) != 0) return 0;
  Transform T;
  T.w = Tq[0];
  T.v = {Tq[1], Tq[2], Tq[3]};
  T.t = {Tq[4], Tq[5], Tq[6]};
  c->start_voxel_set.reset();
  c->voxel_observed_set.reset();
  IndexGetter getter;
  getter.init(cfg->integration_order_mode, xyz, n);
  struct Ray { V3 pg; bool clearing; uint32_t pos; };
  std::vector<Ray> rays;
  size_t idx;
  uint32_t pos = 0;
  while (getter.next(&idx)) {
    const uint32_t p = pos++;
    const V3 pc = {xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2]};
    bool clr;
    if (!c->is_point_valid(pc, false, &clr) || !c->is_semantic_label_valid(labels[idx])) continue;
    const V3 pg = transform_point(T, pc);
    const I3 g = grid_index_from_point(pg, cfg->start_voxel_subsampling_factor * c->voxel_size_inv);
    if (!c->start_voxel_set.replace_hash(LongIndexHash()(g))) continue;
    rays.push_back({pg, clr, p});
  }
  std::vector<uint64_t> keys;
  uint64_t updates = 0;
  auto pack = [](const I3& v) {
    return ((uint64_t)(v.x + (1 << 20)) << 42) | ((uint64_t)(v.y + (1 << 20)) << 21) | (uint64_t)(v.z + (1 << 20));
  };
  ApproxHashSet& S = c->voxel_observed_set;
  const int64_t lim = cfg->max_consecutive_ray_collisions;
  if (n_bounds == 0) {
    for (const Ray& r : rays) {
      RayCaster caster(T.t, r.pg, r.clearing, cfg->voxel_carving_enabled != 0, cfg->max_ray_length_m, c->voxel_size_inv,
                       cfg->truncation_distance, false);
      I3 v;
      int64_t consecutive = 0;
      while (caster.next(&v)) {
        if (!S.replace_hash(LongIndexHash()(v))) ++consecutive;
        else consecutive = 0;
        if (consecutive > lim) break;
        keys.push_back(pack(v));
        ++updates;
      }
    }
  } else if (n_bounds >= 2 && bounds[0] == 0xfffffffeu) {
    // JACOBI-2 chain schedule (design study): per phase every ray first walks against the phase-start set only
    // (provisional walk); then every ray is re-tested against the phase-start set plus the provisional marks
    // of the EARLIER rays of its own chain in the phase (earliest marker of a voxel counts).
    const uint32_t chains = bounds[1];
    auto phase_of = [&](uint32_t pos) -> uint32_t {
      const uint32_t g = pos / chains;
      uint32_t ph = 0;
      for (size_t i = 3; i < n_bounds; ++i) if (g >= bounds[i]) ph = (uint32_t)(i - 3);
      return ph;
    };
    const int iters = (int)bounds[2];  // number of refinement passes (1 = the scheme above)
    std::vector<std::pair<size_t, size_t>> marks;
    size_t r0 = 0;
    while (r0 < rays.size()) {
      const uint32_t phase = phase_of(rays[r0].pos);
      size_t r1 = r0;
      while (r1 < rays.size() && phase_of(rays[r1].pos) == phase) ++r1;
      // paths
      struct Step { size_t slot, h; I3 v; };
      std::vector<std::vector<Step>> path(r1 - r0);
      std::vector<uint32_t> visited(r1 - r0), upd(r1 - r0);
      for (size_t r = r0; r < r1; ++r) {
        RayCaster caster(T.t, rays[r].pg, rays[r].clearing, cfg->voxel_carving_enabled != 0, cfg->max_ray_length_m,
                         c->voxel_size_inv, cfg->truncation_distance, false);
        I3 v;
        while (caster.next(&v)) {
          const size_t h = LongIndexHash()(v);
          path[r - r0].push_back({(h + S.offset) & ApproxHashSet::kMask, h, v});
        }
      }
      // pass 0: against S only
      auto walk = [&](size_t i, const std::unordered_map<size_t, std::pair<uint32_t, size_t>>* pm, uint32_t gen) {
        int64_t consecutive = 0;
        uint32_t s2 = 0, u = 0;
        for (const Step& st : path[i]) {
          bool hit = S.slots[st.slot].load(std::memory_order_relaxed) == st.h;
          if (pm) {
            auto it = pm->find(st.slot);
            if (it != pm->end() && it->second.first < gen) hit = it->second.second == st.h;
          }
          consecutive = hit ? consecutive + 1 : 0;
          ++s2;
          if (consecutive > lim) break;
          ++u;
        }
        visited[i] = s2;
        upd[i] = u;
      };
      for (size_t i = 0; i < r1 - r0; ++i) walk(i, nullptr, 0);
      for (int it = 0; it < iters; ++it) {
        // earliest own-chain marker per slot from the current walks
        std::vector<std::unordered_map<size_t, std::pair<uint32_t, size_t>>> pm(chains);
        for (size_t i = 0; i < r1 - r0; ++i) {
          const uint32_t ch = rays[r0 + i].pos % chains, gen = rays[r0 + i].pos / chains;
          for (uint32_t s2 = 0; s2 < visited[i]; ++s2) {
            const Step& st = path[i][s2];
            auto f = pm[ch].find(st.slot);
            if (f == pm[ch].end() || gen < f->second.first) pm[ch][st.slot] = {gen, st.h};
          }
        }
        for (size_t i = 0; i < r1 - r0; ++i) walk(i, &pm[rays[r0 + i].pos % chains], rays[r0 + i].pos / chains);
      }
      for (size_t i = 0; i < r1 - r0; ++i) {
        for (uint32_t s2 = 0; s2 < visited[i]; ++s2) S.slots[path[i][s2].slot].store(path[i][s2].h, std::memory_order_relaxed);
        for (uint32_t s2 = 0; s2 < upd[i]; ++s2) keys.push_back(pack(path[i][s2].v));
        updates += upd[i];
      }
      r0 = r1;
    }
  } else if (n_bounds >= 2 && bounds[0] == 0xffffffffu) {
    // CHAIN schedule: bounds = {0xffffffff, chains, gens_per_phase}.  chain = pos % chains; a phase covers
    // gens_per_phase generations (pos / chains).  A ray sees the set as of the phase start plus the marks
    // its OWN chain made earlier in the phase.
    const uint32_t chains = bounds[1] & 0xffffu, sub_len = bounds[1] >> 16, gpp = bounds[2];
    // sub_len > 0: inside a phase a chain's generations are cut into sub-runs of sub_len with separate private sets
    // gpp == 0: explicit generation boundaries follow (bounds[3..], ascending, first = 0)
    auto phase_of = [&](uint32_t pos) -> uint32_t {
      const uint32_t g = pos / chains;
      if (gpp) return g / gpp;
      uint32_t ph = 0;
      for (size_t i = 3; i < n_bounds; ++i) if (g >= bounds[i]) ph = (uint32_t)(i - 3);
      return ph;
    };
    std::vector<std::pair<size_t, size_t>> marks;
    std::map<std::pair<uint32_t, uint32_t>, std::unordered_map<size_t, size_t>> priv;
    // optional: bounds[n_bounds-1] = 0x80000000 | direct-mapped private-set size (power of two)
    size_t dm = 0;
    if (n_bounds > 3 && (bounds[n_bounds - 1] & 0x80000000u)) { dm = bounds[n_bounds - 1] & 0x7fffffffu; --n_bounds; }
    size_t r0 = 0;
    while (r0 < rays.size()) {
      const uint32_t phase = phase_of(rays[r0].pos);
      marks.clear();
      priv.clear();
      uint32_t g_first = rays[r0].pos / chains;
      if (gpp) g_first = phase * gpp; else g_first = bounds[3 + phase];
      size_t r1 = r0;
      for (; r1 < rays.size() && phase_of(rays[r1].pos) == phase; ++r1) {
        const Ray& r = rays[r1];
        auto& pm = priv[{r.pos % chains, sub_len ? (r.pos / chains - g_first) / sub_len : 0u}];
        RayCaster caster(T.t, r.pg, r.clearing, cfg->voxel_carving_enabled != 0, cfg->max_ray_length_m, c->voxel_size_inv,
                         cfg->truncation_distance, false);
        I3 v;
        int64_t consecutive = 0;
        std::vector<std::pair<size_t, size_t>> own;
        struct Flush { std::unordered_map<size_t, size_t>& m; std::vector<std::pair<size_t, size_t>>& o;
                       ~Flush() { for (auto& kv : o) m[kv.first] = kv.second; } } flush{pm, own};
        while (caster.next(&v)) {
          const size_t h = LongIndexHash()(v);
          const size_t slot = (h + S.offset) & ApproxHashSet::kMask;
          size_t content;
          if (dm) {
            auto it = pm.find(slot & (dm - 1));
            content = (it != pm.end() && (it->second >> 32) == slot) ? (it->second & 0xffffffffu) : S.slots[slot].load(std::memory_order_relaxed);
            own.push_back({slot & (dm - 1), (slot << 32) | h});
          } else {
            auto it = pm.find(slot);
            content = (it != pm.end()) ? it->second : S.slots[slot].load(std::memory_order_relaxed);
            own.push_back({slot, h});
          }
          if (content == h) ++consecutive;
          else consecutive = 0;
          marks.push_back({slot, h});
          if (consecutive > lim) break;
          keys.push_back(pack(v));
          ++updates;
        }
      }
      for (const auto& m : marks) S.slots[m.first].store(m.second, std::memory_order_relaxed);
      r0 = r1;
    }
  } else {
    size_t r0 = 0;
    std::vector<std::pair<size_t, size_t>> marks;  // (slot, hash) in serial order
    for (size_t w = 0; w < n_bounds; ++w) {
      const uint32_t end = (w + 1 < n_bounds) ? bounds[w + 1] : 0xffffffffu;
      marks.clear();
      size_t r1 = r0;
      for (; r1 < rays.size() && rays[r1].pos < end; ++r1) {
        const Ray& r = rays[r1];
        RayCaster caster(T.t, r.pg, r.clearing, cfg->voxel_carving_enabled != 0, cfg->max_ray_length_m, c->voxel_size_inv,
                         cfg->truncation_distance, false);
        I3 v;
        int64_t consecutive = 0;
        while (caster.next(&v)) {
          const size_t h = LongIndexHash()(v);
          const size_t slot = (h + S.offset) & ApproxHashSet::kMask;
          if (S.slots[slot].load(std::memory_order_relaxed) == h) ++consecutive;
          else consecutive = 0;
          marks.push_back({slot, h});
          if (consecutive > lim) break;
          keys.push_back(pack(v));
          ++updates;
        }
      }
      for (const auto& m : marks) S.slots[m.first].store(m.second, std::memory_order_relaxed);
      r0 = r1;
    }
  }
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  if (n_updates) *n_updates = updates;
  const size_t m = std::min(cap, keys.size());
  if (out_keys) std::memcpy(out_keys, keys.data(), m * sizeof(uint64_t));
  ko_destroy(c);
  return keys.size();
}


// Design study (tools/fixpoint_study.py) of the GPU's EVENT-DRIVEN fix point for the serial early-out
// (kimera_semantics_amd/csrc/ks_k_exact.h): seed lengths L0 from the chain schedule with doubling phases, potential
// marks for the first min(full, L0 + pad) steps of every ray sorted by (slot, time); per round every dirty ray walks
// again (a mark is valid iff its step < the CURRENT length of its ray; lengths are updated in place, rays in a
// shuffled order), and every mark whose validity toggled dirties the owners of the marks that follow it in its slot
// up to and including the first valid one.  Prints per-round statistics to stderr and returns the number of rays whose
// final length differs from the serial reference's (0 = exact), or SIZE_MAX if a ray outgrew its potential marks.
size_t ko_sim_fixpoint(const ko_config* cfg, const float Tq[7], const float* xyz, const uint8_t* labels, size_t n,
                       uint32_t pad, uint32_t seed_mode, uint64_t* stats, size_t n_stats) {
  ko_ctx* c = nullptr;
  if (ko_create(cfg, &c) != 0) return 0;
  Transform T;
  T.w = Tq[0];
  T.v = {Tq[1], Tq[2], Tq[3]};
  T.t = {Tq[4], Tq[5], Tq[6]};
  c->start_voxel_set.reset();
  c->voxel_observed_set.reset();
  IndexGetter getter;
  getter.init(cfg->integration_order_mode, xyz, n);
  struct Step { uint32_t slot, h; };
  std::vector<std::vector<Step>> path;
  std::vector<uint32_t> posv;
  ApproxHashSet& S = c->voxel_observed_set;
  size_t idx;
  uint32_t pos = 0;
  while (getter.next(&idx)) {
    const uint32_t p = pos++;
    const V3 pc = {xyz[3 * idx], xyz[3 * idx + 1], xyz[3 * idx + 2]};
    bool clr;
    if (!c->is_point_valid(pc, false, &clr) || !c->is_semantic_label_valid(labels[idx])) continue;
    const V3 pg = transform_point(T, pc);
    const I3 g = grid_index_from_point(pg, cfg->start_voxel_subsampling_factor * c->voxel_size_inv);
    if (!c->start_voxel_set.replace_hash(LongIndexHash()(g))) continue;
    RayCaster caster(T.t, pg, clr, cfg->voxel_carving_enabled != 0, cfg->max_ray_length_m, c->voxel_size_inv, cfg->truncation_distance, false);
    std::vector<Step> st;
    I3 v;
    while (caster.next(&v)) {
      const size_t h = LongIndexHash()(v);
      st.push_back({(uint32_t)((h + S.offset) & ApproxHashSet::kMask), (uint32_t)h});
    }
    path.push_back(std::move(st));
    posv.push_back(p);
  }
  const size_t R = path.size();
  const int64_t lim = cfg->max_consecutive_ray_collisions;
  auto plain = [&](uint32_t slot) { return S.slots[slot].load(std::memory_order_relaxed); };
  // ---- the serial reference ----
  std::vector<uint32_t> Lref(R);
  {
    std::unordered_map<uint32_t, size_t> cur;
    for (size_t i = 0; i < R; ++i) {
      int64_t cc = 0;
      uint32_t vis = 0;
      for (const Step& st : path[i]) {
        auto it = cur.find(st.slot);
        const size_t content = it != cur.end() ? it->second : plain(st.slot);
        cc = content == st.h ? cc + 1 : 0;
        cur[st.slot] = st.h;
        ++vis;
        if (cc > lim) break;
      }
      Lref[i] = vis;
    }
  }
  // ---- seed ----
  std::vector<uint32_t> L(R);
  if (seed_mode == 0) {
    for (size_t i = 0; i < R; ++i) L[i] = (uint32_t)path[i].size();
  } else {
    // chain schedule, phases of generations doubling (seed_mode = growth in 1/16ths, 32 = doubling), private set per chain and phase
    std::vector<uint32_t> B{0};
    const uint32_t chains = IndexGetter::mixed_chains(n, cfg->integration_order_mode);
    const uint32_t n_gen = (uint32_t)((n + chains - 1) / chains);
    for (;;) {
      const uint64_t inc = std::max<uint64_t>(1, (uint64_t)B.back() * (seed_mode - 16) / 16);
      if (B.back() + inc >= n_gen) break;
      B.push_back((uint32_t)(B.back() + inc));
    }
    std::unordered_map<uint32_t, size_t> shared;
    size_t r0 = 0;
    for (size_t ph = 0; ph < B.size(); ++ph) {
      const uint32_t g1 = ph + 1 < B.size() ? B[ph + 1] : 0xffffffffu;
      std::map<uint32_t, std::unordered_map<uint32_t, size_t>> priv;
      std::vector<std::pair<uint32_t, size_t>> marks;
      size_t r1 = r0;
      for (; r1 < R && posv[r1] / chains < g1; ++r1) {
        auto& pm = priv[posv[r1] % chains];
        int64_t cc = 0;
        uint32_t vis = 0;
        std::vector<std::pair<uint32_t, size_t>> own;
        for (const Step& st : path[r1]) {
          auto it = pm.find(st.slot);
          size_t content;
          if (it != pm.end()) content = it->second;
          else {
            auto is = shared.find(st.slot);
            content = is != shared.end() ? is->second : plain(st.slot);
          }
          cc = content == st.h ? cc + 1 : 0;
          own.push_back({st.slot, st.h});
          ++vis;
          if (cc > lim) break;
        }
        for (auto& m : own) { pm[m.first] = m.second; marks.push_back(m); }
        L[r1] = vis;
      }
      for (auto& m : marks) shared[m.first] = m.second;
      r0 = r1;
    }
  }
  size_t seed_wrong = 0, seed_marks = 0;
  for (size_t i = 0; i < R; ++i) { seed_wrong += L[i] != Lref[i]; seed_marks += L[i]; }
  // ---- potential marks ----
  std::vector<uint32_t> U(R);
  struct Mark { uint32_t ray, step, h; };
  std::unordered_map<uint32_t, std::vector<Mark>> M;
  size_t n_pot = 0;
  for (size_t i = 0; i < R; ++i) {
    U[i] = (uint32_t)std::min<size_t>(path[i].size(), (size_t)L[i] + pad);
    for (uint32_t k = 0; k < U[i]; ++k) M[path[i][k].slot].push_back({(uint32_t)i, k, path[i][k].h});
    n_pot += U[i];
  }  // (rays are visited in serial order and steps ascending: every slot's marks are in time order)
  auto find_mark = [&](const std::vector<Mark>& v, uint32_t ray, uint32_t step) {   // first mark at or after (ray, step)
    size_t lo = 0, hi = v.size();
    while (lo < hi) {
      const size_t mid = (lo + hi) / 2;
      if (v[mid].ray < ray || (v[mid].ray == ray && v[mid].step < step)) lo = mid + 1;
      else hi = mid;
    }
    return lo;
  };
  // marks past a ray's potential ones (it outgrew its seed length + pad): appended per slot, unsorted ("X" marks)
  std::unordered_map<uint32_t, std::vector<Mark>> X;
  std::vector<uint32_t> UX(U);
  size_t n_x = 0;
  std::vector<uint8_t> dirty(R, 1);
  std::vector<uint32_t> work(R);
  for (size_t i = 0; i < R; ++i) work[i] = (uint32_t)i;
  uint64_t rng = 88172645463325252ull;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
  bool overflow = false;
  size_t round = 0, n_out = 0;
  uint64_t tot_steps = 0, tot_back = 0, tot_fwd = 0;
  auto before = [](uint32_t r1, uint32_t k1, uint32_t r2, uint32_t k2) { return r1 < r2 || (r1 == r2 && k1 < k2); };
  static const std::vector<Mark> kNone;
  const bool jacobi = getenv("KO_STUDY_JACOBI") != nullptr;   // rounds as the GPU runs them: evaluate all against the old lengths, then apply + propagate
  while (!work.empty() && round < 1000) {
    for (size_t i = work.size(); i > 1; --i) std::swap(work[i - 1], work[rnd() % i]);
    for (uint32_t i : work) dirty[i] = 0;
    std::vector<uint32_t> next;
    size_t changed = 0, steps = 0, back = 0, fwd = 0, toggled = 0, max_fwd = 0, max_back = 0, max_len = 0;
    std::vector<std::pair<uint32_t, uint32_t>> pending;   // (ray, new length) of the Jacobi round
    auto propagate = [&](uint32_t i, uint32_t old, uint32_t vis) {
      for (uint32_t k = std::min(old, vis); k < std::max(old, vis); ++k) {
        ++toggled;
        auto im = M.find(path[i][k].slot);
        if (im != M.end()) {
          const std::vector<Mark>& v = im->second;
          size_t j = find_mark(v, i, k);
          if (j < v.size() && v[j].ray == i && v[j].step == k) ++j;
          size_t one = 0;
          for (; j < v.size(); ++j) {
            ++fwd; ++one;
            const uint32_t o = v[j].ray;
            if (o != i && !dirty[o]) { dirty[o] = 1; next.push_back(o); }
            if (v[j].step < L[o]) break;
          }
          max_fwd = std::max(max_fwd, one);
        }
        auto ix = X.find(path[i][k].slot);
        if (ix != X.end())
          for (const Mark& m : ix->second) {
            ++fwd;
            if (!before(i, k, m.ray, m.step)) continue;
            if (m.ray != i && !dirty[m.ray]) { dirty[m.ray] = 1; next.push_back(m.ray); }
          }
      }
    };
    for (uint32_t i : work) {
      int64_t cc = 0;
      uint32_t vis = 0;
      for (uint32_t k = 0; k < path[i].size(); ++k) {
        const Step& st = path[i][k];
        if (k >= UX[i]) { X[st.slot].push_back({i, k, st.h}); ++n_x; }
        auto im = M.find(st.slot);
        const std::vector<Mark>& v = im != M.end() ? im->second : kNone;
        size_t j = find_mark(v, i, k);
        size_t content = plain(st.slot);
        bool have = false;
        uint32_t br = 0, bk = 0;
        size_t one = 0;
        while (j > 0) {
          --j;
          ++back; ++one;
          if (v[j].ray == i || v[j].step < L[v[j].ray]) { content = v[j].h; have = true; br = v[j].ray; bk = v[j].step; break; }
        }
        max_back = std::max(max_back, one);
        auto ix = X.find(st.slot);
        if (ix != X.end())
          for (const Mark& m : ix->second) {
            ++back;
            if (!before(m.ray, m.step, i, k)) continue;
            if (!(m.ray == i || m.step < L[m.ray])) continue;
            if (!have || before(br, bk, m.ray, m.step)) { content = m.h; have = true; br = m.ray; bk = m.step; }
          }
        cc = content == st.h ? cc + 1 : 0;
        ++vis;
        ++steps;
        if (cc > lim) break;
      }
      max_len = std::max<size_t>(max_len, vis);
      UX[i] = std::max(UX[i], vis);
      const uint32_t old = L[i];
      if (vis != old) {
        ++changed;
        if (jacobi) pending.push_back({i, vis});
        else {
          L[i] = vis;
          propagate(i, old, vis);
        }
      }
    }
    if (jacobi) {
      std::vector<uint32_t> olds;
      for (auto& pr : pending) { olds.push_back(L[pr.first]); L[pr.first] = pr.second; }
      for (size_t q = 0; q < pending.size(); ++q) propagate(pending[q].first, olds[q], pending[q].second);
    }
    if (round == 0 && getenv("KO_STUDY_COMPACT")) {   // as the GPU does after its first (full) iteration: invalid marks leave M
      for (auto& kv : M) {
        auto& v = kv.second;
        v.erase(std::remove_if(v.begin(), v.end(), [&](const Mark& m) { return m.step >= L[m.ray]; }), v.end());
      }
      for (size_t i = 0; i < R; ++i) UX[i] = std::max<uint32_t>(std::min(UX[i], L[i]), 0);
    }
    fprintf(stderr, "      longest single scans: forward %zu, backward %zu; longest walk %zu steps\n", max_fwd, max_back, max_len);
    fprintf(stderr, "  round %2zu: dirty %7zu changed %6zu toggled %7zu walk steps %8zu back-scan %8zu fwd-scan %7zu  X marks %zu\n", round, work.size(), changed,
            toggled, steps, back, fwd, n_x);
    if (n_out + 3 <= n_stats) { stats[n_out++] = work.size(); stats[n_out++] = changed; stats[n_out++] = steps; }
    tot_steps += steps; tot_back += back; tot_fwd += fwd;
    work.swap(next);
    ++round;
  }
  {
    size_t max_x = 0, max_m = 0, x_slots = 0;
    for (auto& kv : X) { max_x = std::max(max_x, kv.second.size()); ++x_slots; }
    for (auto& kv : M) max_m = std::max(max_m, kv.second.size());
    std::vector<size_t> xs;
    for (auto& kv : X) xs.push_back(kv.second.size());
    std::sort(xs.rbegin(), xs.rend());
    fprintf(stderr, "longest X chain %zu (slots with X marks %zu; top:", max_x, x_slots);
    for (size_t i = 0; i < std::min<size_t>(8, xs.size()); ++i) fprintf(stderr, " %zu", xs[i]);
    fprintf(stderr, "), longest M range %zu\n", max_m);
  }
  size_t wrong = 0, ref_marks = 0;
  for (size_t i = 0; i < R; ++i) { wrong += L[i] != Lref[i]; ref_marks += Lref[i]; }
  fprintf(stderr, "fixpoint study: rays %zu, serial marks %zu, seed marks %zu (seed wrong on %zu rays), potential marks %zu (pad %u), rounds %zu, "
                  "total walk steps %llu back %llu fwd %llu, overflow %d, WRONG %zu\n", R, ref_marks, seed_marks, seed_wrong, n_pot, pad, round,
          (unsigned long long)tot_steps, (unsigned long long)tot_back, (unsigned long long)tot_fwd, (int)overflow, wrong);
  ko_destroy(c);
  return overflow ? (size_t)-1 : wrong;
}

size_t ko_num_blocks(ko_ctx* ctx) { return ctx->tsdf_layer.blocks.size(); }
size_t ko_num_semantic_blocks(ko_ctx* ctx) { return ctx->semantic_layer.blocks.size(); }

void ko_get_block_indices(ko_ctx* ctx, int32_t* out) { sorted_indices(ctx->tsdf_layer.blocks, out); }
void ko_get_semantic_block_indices(ko_ctx* ctx, int32_t* out) { sorted_indices(ctx->semantic_layer.blocks, out); }

int ko_get_block(ko_ctx* ctx, const int32_t idx[3], void* tsdf_out, void* sem_out) {
  const B3 b = {idx[0], idx[1], idx[2]};
  const size_t nvox = static_cast<size_t>(ctx->vps) * ctx->vps * ctx->vps;
  int absent = 0;
  if (tsdf_out) {
    auto blk = ctx->tsdf_layer.get(b);
    if (blk) {
      std::memcpy(tsdf_out, blk->voxels.data(), nvox * sizeof(TsdfVoxel));
    } else {
      absent = 1;
      TsdfVoxel d;
      for (size_t i = 0; i < nvox; ++i) std::memcpy(static_cast<char*>(tsdf_out) + i * sizeof(TsdfVoxel), &d, sizeof(d));
    }
  }
  if (sem_out) {
    auto blk = ctx->semantic_layer.get(b);
    if (blk) {
      std::memcpy(sem_out, blk->voxels.data(), nvox * sizeof(SemanticVoxel));
    } else {
      absent = 1;
      SemanticVoxel d;
      for (size_t i = 0; i < nvox; ++i) std::memcpy(static_cast<char*>(sem_out) + i * sizeof(SemanticVoxel), &d, sizeof(d));
    }
  }
  return absent;
}

// ---- pure functions for KATs ----
void ko_transform_point(const float Tq[7], const float p[3], float out[3]) {
  Transform T;
  T.w = Tq[0];
  T.v = {Tq[1], Tq[2], Tq[3]};
  T.t = {Tq[4], Tq[5], Tq[6]};
  const V3 r = transform_point(T, {p[0], p[1], p[2]});
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void ko_grid_index_from_point(const float p[3], float inv, int64_t out[3]) {
  const I3 i = grid_index_from_point({p[0], p[1], p[2]}, inv);
  out[0] = i.x; out[1] = i.y; out[2] = i.z;
}
size_t ko_cast_ray(const float origin[3], const float point_G[3], int is_clearing, int carving,
                   float max_ray_length_m, float voxel_size_inv, float truncation, int cast_from_origin,
                   int64_t* out_xyz, size_t cap) {
  RayCaster c({origin[0], origin[1], origin[2]}, {point_G[0], point_G[1], point_G[2]}, is_clearing != 0,
              carving != 0, max_ray_length_m, voxel_size_inv, truncation, cast_from_origin != 0);
  size_t n = 0;
  I3 v;
  while (c.next(&v)) {
    if (n < cap) {
      out_xyz[3 * n] = v.x; out_xyz[3 * n + 1] = v.y; out_xyz[3 * n + 2] = v.z;
    }
    ++n;
  }
  return n;
}
void ko_log_likelihood(float p_match, float* out) {
  ko_config c;
  ko_default_config(&c);
  c.semantic_measurement_probability = p_match;
  ko_ctx tmp;
  tmp.cfg = c;
  if (!tmp.set_semantic_probabilities()) {
    for (int i = 0; i < kNumLabels * kNumLabels; ++i) out[i] = std::numeric_limits<float>::quiet_NaN();
    return;
  }
  for (int i = 0; i < kNumLabels; ++i)
    for (int j = 0; j < kNumLabels; ++j) out[i * kNumLabels + j] = tmp.L[i][j];
}
uint32_t ko_long_index_hash(const int64_t idx[3]) { return index_hash(idx[0], idx[1], idx[2]); }
size_t ko_mixed_index(size_t s, size_t n, int mode) { return IndexGetter::mixed_index(s, n, mode); }
uint32_t ko_mixed_chains(size_t n, int mode) { return IndexGetter::mixed_chains(n, mode); }
void ko_update_tsdf_voxel(const ko_config* cfg, const float origin[3], const float point_G[3],
                          const int64_t vi[3], const uint8_t rgba[4], float weight, float* distance,
                          float* voxel_weight, uint8_t voxel_rgba[4]) {
  TsdfVoxel v;
  v.distance = *distance;
  v.weight = *voxel_weight;
  v.color = {voxel_rgba[0], voxel_rgba[1], voxel_rgba[2], voxel_rgba[3]};
  ko_ctx::update_tsdf_voxel_nolock(*cfg, cfg->voxel_size, {origin[0], origin[1], origin[2]},
                                   {point_G[0], point_G[1], point_G[2]}, {vi[0], vi[1], vi[2]},
                                   {rgba[0], rgba[1], rgba[2], rgba[3]}, weight, &v);
  *distance = v.distance;
  *voxel_weight = v.weight;
  voxel_rgba[0] = v.color.r; voxel_rgba[1] = v.color.g; voxel_rgba[2] = v.color.b; voxel_rgba[3] = v.color.a;
}
void ko_blend_two_colors(const uint8_t c1[4], float w1, const uint8_t c2[4], float w2, uint8_t out[4]) {
  const Rgba o = blend_two_colors({c1[0], c1[1], c1[2], c1[3]}, w1, {c2[0], c2[1], c2[2], c2[3]}, w2);
  out[0] = o.r; out[1] = o.g; out[2] = o.b; out[3] = o.a;
}
void ko_rainbow_color_map(double h, uint8_t out[4]) {
  const Rgba o = rainbow_color_map(h);
  out[0] = o.r; out[1] = o.g; out[2] = o.b; out[3] = o.a;
}

}  // extern "C"
