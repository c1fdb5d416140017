// ks_device_math.h — device-side arithmetic of the semantic TSDF hot path (gfx950).
//
// Built with -ffp-contract=off (no FMA contraction) and hipcc's default correctly-rounded
// f32 divide/sqrt so that voxel indices are bit-identical to the x86 reference arithmetic.
// Each function cites the reference call site it serves ([K:...] = file under
// /root/reference/kimera_semantics/) and the upstream Voxblox/minkindr/Eigen routine whose
// published behaviour it implements ([V:...], [M:...], [E:...]; those libraries are not in
// /root/reference — see SURVEY.md Appendix A).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ksd {

constexpr int kNumLabels = 21;           // [K:include/kimera_semantics/common.h:26]
constexpr float kEps = 1e-6f;            // voxblox kEpsilon / kCoordinateEpsilon / kFloatEpsilon
constexpr int kTileEdge = 8;             // device tile = 8x8x8 voxels
constexpr int kTileVoxels = 512;
constexpr int kIndexLimit = 1 << 23;     // |voxel index| must stay below this (packing range)

struct f3 { float x, y, z; };

// std::min / std::max operand semantics (NaN behaviour differs from fminf/fmaxf)
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }

__device__ __forceinline__ f3 sub3(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 add3(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 mul3(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
// Eigen's fixed-size 3-element reduction associates as c0 + (c1 + c2).
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ float norm3(f3 a) { return sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 normalized3(f3 a) {
  const float z = dot3(a, a);
  if (z > 0.0f) {
    const float s = sqrtf(z);
    return {a.x / s, a.y / s, a.z / s};
  }
  return a;
}
__device__ __forceinline__ f3 cross3(f3 a, f3 b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

struct Pose {  // T_G_C: unit quaternion (w, v) + translation
  float w;
  f3 v;
  f3 t;
};

// T_G_C * point_C  — [K:src/semantic_tsdf_integrator_fast.cpp:81], [K:src/semantic_tsdf_integrator_merged.cpp:287]
// (minkindr transform = rotate + translate; Eigen quaternion _transformVector).
__device__ __forceinline__ f3 transform_point(const Pose& T, f3 p) {
  f3 uv = cross3(T.v, p);
  uv = add3(uv, uv);
  const f3 c2 = cross3(T.v, uv);
  f3 r;
  r.x = (p.x + T.w * uv.x) + c2.x;
  r.y = (p.y + T.w * uv.y) + c2.y;
  r.z = (p.z + T.w * uv.z) + c2.z;
  return add3(r, T.t);
}

// vxb::AnyIndexHash / LongIndexHash (truncated to 32 bits) — [K:semantic_integrator_base.h:64-66],
// [K:semantic_tsdf_integrator_fast.h:114-130].  Low 32 bits of x + 17191 y + 17191^2 z.
__device__ __forceinline__ uint32_t index_hash(int x, int y, int z) {
  return (uint32_t)x + (uint32_t)y * 17191u + (uint32_t)z * 295530481u;
}

// getGridIndexFromPoint(point, inv) — [K:src/semantic_tsdf_integrator_fast.cpp:88-89]
__device__ __forceinline__ float grid_coord(float p, float inv) { return floorf(p * inv + kEps); }
// getGridIndexFromPoint(scaled_point) — used by the ray caster set-up
__device__ __forceinline__ float grid_coord_scaled(float p) { return floorf(p + kEps); }

// isPointValid — [K:src/semantic_tsdf_integrator_fast.cpp:75]; returns 0 invalid, 1 valid, 2 valid+clearing
__device__ __forceinline__ int point_validity(f3 p, float min_ray, float max_ray, bool allow_clear, bool freespace) {
  const float d = norm3(p);
  if (d < min_ray) return 0;
  if (d > max_ray) return (allow_clear || freespace) ? 2 : 0;
  return freespace ? 2 : 1;
}
// getVoxelWeight — [K:src/semantic_tsdf_integrator_fast.cpp:126], [K:src/semantic_tsdf_integrator_merged.cpp:267]
__device__ __forceinline__ float voxel_weight(float z, bool use_const_weight) {
  if (use_const_weight) return 1.0f;
  const float dz = fabsf(z);
  if (dz > kEps) return 1.0f / (dz * dz);
  return 0.0f;
}

// vxb::RayCaster — [K:src/semantic_tsdf_integrator_fast.cpp:95-110] (cast_from_origin=false),
// [K:src/semantic_tsdf_integrator_merged.cpp:288-305] (default: origin -> surface).
struct Dda {
  int cx, cy, cz;
  int sx, sy, sz;
  float tx, ty, tz;
  float dx, dy, dz;
  int steps;      // ray_length_in_steps_: emits steps+1 indices
  bool in_range;  // all indices representable

  __device__ __forceinline__ void setup_scaled(f3 start, f3 end) {
    in_range = true;
    if (isnan(start.x) || isnan(start.y) || isnan(start.z) || isnan(end.x) || isnan(end.y) || isnan(end.z)) {
      cx = cy = cz = 0;
      sx = sy = sz = 0;
      tx = ty = tz = 0.f;
      dx = dy = dz = 0.f;
      steps = 0;
      return;
    }
    const float fcx = grid_coord_scaled(start.x), fcy = grid_coord_scaled(start.y), fcz = grid_coord_scaled(start.z);
    const float fex = grid_coord_scaled(end.x), fey = grid_coord_scaled(end.y), fez = grid_coord_scaled(end.z);
    const float lim = (float)kIndexLimit;
    if (!(fabsf(fcx) < lim && fabsf(fcy) < lim && fabsf(fcz) < lim && fabsf(fex) < lim && fabsf(fey) < lim &&
          fabsf(fez) < lim)) {
      in_range = false;
      cx = cy = cz = 0;
      sx = sy = sz = 0;
      tx = ty = tz = 0.f;
      dx = dy = dz = 0.f;
      steps = 0;
      return;
    }
    cx = (int)fcx; cy = (int)fcy; cz = (int)fcz;
    const int ex = (int)fex, ey = (int)fey, ez = (int)fez;
    steps = abs(ex - cx) + abs(ey - cy) + abs(ez - cz);
    const f3 ray = sub3(end, start);
    sx = (0.0f < ray.x) - (ray.x < 0.0f);
    sy = (0.0f < ray.y) - (ray.y < 0.0f);
    sz = (0.0f < ray.z) - (ray.z < 0.0f);
    const float shx = start.x - (float)cx, shy = start.y - (float)cy, shz = start.z - (float)cz;
    const float bx = (float)max(0, sx) - shx, by = (float)max(0, sy) - shy, bz = (float)max(0, sz) - shz;
    // upstream's "|r| < 0 ? 2 : ..." guard is dead code: zero components divide by zero.
    tx = bx / ray.x; ty = by / ray.y; tz = bz / ray.z;
    dx = (float)sx / ray.x; dy = (float)sy / ray.y; dz = (float)sz / ray.z;
  }

  __device__ __forceinline__ void setup(f3 origin, f3 point_G, bool is_clearing, bool carving, float max_ray_length,
                                        float voxel_size_inv, float trunc, bool cast_from_origin) {
    const f3 d = sub3(point_G, origin);
    const f3 unit = normalized3(d);
    f3 ray_start, ray_end;
    if (is_clearing) {
      float len = norm3(d);
      len = std_min(std_max(len - trunc, 0.0f), max_ray_length);
      ray_end = add3(origin, mul3(unit, len));
      ray_start = carving ? origin : ray_end;
    } else {
      ray_end = add3(point_G, mul3(unit, trunc));
      ray_start = carving ? origin : sub3(point_G, mul3(unit, trunc));
    }
    const f3 s = mul3(ray_start, voxel_size_inv);
    const f3 e = mul3(ray_end, voxel_size_inv);
    if (cast_from_origin) setup_scaled(s, e);
    else setup_scaled(e, s);
  }

  // advance to the next index (Eigen minCoeff: first strict minimum, NaN never wins):
  //   k = 0; m = tx; if (ty < m) { k = 1; m = ty; } if (tz < m) k = 2; cur[k] += sign[k]; t[k] += step[k];
  // written with selects only (lanes of a group replay different numbers of steps: a branch per step
  // would serialise the wave); `on` = false leaves the state untouched.
  __device__ __forceinline__ void advance(bool on = true) {
    const bool y_lt = ty < tx;
    const float m = y_lt ? ty : tx;
    const bool z_lt = tz < m;
    const bool sel_z = on && z_lt, sel_y = on && y_lt && !z_lt, sel_x = on && !y_lt && !z_lt;
    const float nx = tx + dx, ny = ty + dy, nz = tz + dz;
    cx += sel_x ? sx : 0;
    cy += sel_y ? sy : 0;
    cz += sel_z ? sz : 0;
    tx = sel_x ? nx : tx;
    ty = sel_y ? ny : ty;
    tz = sel_z ? nz : tz;
  }
  // advance(), carrying index_hash(cx, cy, cz) along: the hash is linear in the coordinates (mod 2^32), so a step along
  // an axis adds that axis' increment (hx = sx, hy = 17191 sy, hz = 17191^2 sz as uint32) — no multiplies per step
  __device__ __forceinline__ void advance_hashed(uint32_t& h, uint32_t hx, uint32_t hy, uint32_t hz) {
    const bool y_lt = ty < tx;
    const float m = y_lt ? ty : tx;
    const bool z_lt = tz < m;
    const bool sel_z = z_lt, sel_y = y_lt && !z_lt, sel_x = !y_lt && !z_lt;
    const float nx = tx + dx, ny = ty + dy, nz = tz + dz;
    cx += sel_x ? sx : 0;
    cy += sel_y ? sy : 0;
    cz += sel_z ? sz : 0;
    h += sel_z ? hz : sel_y ? hy : hx;
    tx = sel_x ? nx : tx;
    ty = sel_y ? ny : ty;
    tz = sel_z ? nz : tz;
  }
};

// vxb::Color::blendTwoColors — [K:src/semantic_tsdf_integrator_merged.cpp:273-274] and inside updateTsdfVoxel
__device__ __forceinline__ uint32_t blend_two_colors(uint32_t c1, float w1, uint32_t c2, float w2) {
  const float total = w1 + w2;
  w1 /= total;
  w2 /= total;
  uint32_t out = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float a = (float)((c1 >> (8 * k)) & 0xffu);
    const float b = (float)((c2 >> (8 * k)) & 0xffu);
    const float v = roundf(a * w1 + b * w2);
    out |= ((uint32_t)(uint8_t)(int)v) << (8 * k);
  }
  return out;
}

// vxb::rainbowColorMap(double) — [K:src/semantic_integrator_base.cpp:183]
__device__ __forceinline__ uint32_t rainbow_color_map(double h) {
  const double s = 1.0, v = 1.0;
  h -= floor(h);
  h *= 6;
  const int i = (int)floor(h);
  double f = h - i;
  if (!(i & 1)) f = 1 - f;
  const double m = v * (1 - s);
  const double n = v * (1 - s * f);
  uint32_t r, g, b;
  switch (i) {
    case 6:
    case 0: r = (uint8_t)(255 * v); g = (uint8_t)(255 * n); b = (uint8_t)(255 * m); break;
    case 1: r = (uint8_t)(255 * n); g = (uint8_t)(255 * v); b = (uint8_t)(255 * m); break;
    case 2: r = (uint8_t)(255 * m); g = (uint8_t)(255 * v); b = (uint8_t)(255 * n); break;
    case 3: r = (uint8_t)(255 * m); g = (uint8_t)(255 * n); b = (uint8_t)(255 * v); break;
    case 4: r = (uint8_t)(255 * n); g = (uint8_t)(255 * m); b = (uint8_t)(255 * v); break;
    case 5: r = (uint8_t)(255 * v); g = (uint8_t)(255 * m); b = (uint8_t)(255 * n); break;
    default: r = 255; g = 127; b = 127; break;
  }
  return r | (g << 8) | (b << 16) | (255u << 24);
}

// Parameters of TsdfIntegratorBase::updateTsdfVoxel that are constant per context.
struct TsdfParams {
  float voxel_size;
  float trunc;
  float max_weight;
  float dropoff_denominator;  // trunc - voxel_size
  float sparsity_factor;
  int use_dropoff;
  int use_sparsity;
};

// vxb::TsdfIntegratorBase::updateTsdfVoxel (+ computeDistance) —
// [K:src/semantic_tsdf_integrator_fast.cpp:128], [K:src/semantic_tsdf_integrator_merged.cpp:317-319],
// split in two halves so kernels can evaluate the half that does not depend on the voxel
// state (sdf, updated weight) for many updates in parallel:
//   tsdf_operands : computeDistance + weight drop-off + sparsity compensation
//   tsdf_combine  : the voxel-state recurrence (weighted mean, clamp, optional colour blend)
// The arithmetic and its order are exactly those of the single function.
__device__ __forceinline__ void tsdf_operands(const TsdfParams& P, f3 origin, f3 point_G, int vx, int vy, int vz,
                                              float weight, float& sdf, float& uw) {
  // getCenterPointFromGridIndex: (float(i) + 0.5) * voxel_size (exact in f32 == upstream's double evaluation)
  const f3 c = {((float)vx + 0.5f) * P.voxel_size, ((float)vy + 0.5f) * P.voxel_size, ((float)vz + 0.5f) * P.voxel_size};
  const f3 v_voxel_origin = sub3(c, origin);
  const f3 v_point_origin = sub3(point_G, origin);
  const float dist_G = norm3(v_point_origin);
  const float dist_G_V = dot3(v_voxel_origin, v_point_origin) / dist_G;
  sdf = dist_G - dist_G_V;
  uw = weight;
  if (P.use_dropoff && sdf < -P.voxel_size) {
    uw = weight * (P.trunc + sdf) / P.dropoff_denominator;
    uw = std_max(uw, 0.0f);
  }
  if (P.use_sparsity) {
    if (fabsf(sdf) < P.trunc) uw *= P.sparsity_factor;
  }
}

template <bool BLEND>
__device__ __forceinline__ void tsdf_combine(const TsdfParams& P, float sdf, float uw, uint32_t color, float& distance,
                                             float& vweight, uint32_t& vcolor) {
  const float new_weight = vweight + uw;
  if (new_weight < kEps) return;
  const float new_sdf = (sdf * uw + distance * vweight) / new_weight;
  if (BLEND) {
    if (fabsf(sdf) < P.trunc) vcolor = blend_two_colors(vcolor, vweight, color, uw);
  }
  distance = (new_sdf > 0.0f) ? std_min(P.trunc, new_sdf) : std_max(-P.trunc, new_sdf);
  vweight = std_min(P.max_weight, new_weight);
}

template <bool BLEND>
__device__ __forceinline__ void update_tsdf_voxel(const TsdfParams& P, f3 origin, f3 point_G, int vx, int vy, int vz,
                                                  uint32_t color, float weight, float& distance, float& vweight,
                                                  uint32_t& vcolor) {
  float sdf, uw;
  tsdf_operands(P, origin, point_G, vx, vy, vz, weight, sdf, uw);
  tsdf_combine<BLEND>(P, sdf, uw, color, distance, vweight, vcolor);
}

}  // namespace ksd
