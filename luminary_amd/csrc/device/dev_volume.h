// The fog volume: closed-form distance sampling along the path, light scattered into it (sun, ambient, bridges to emissive triangles) and
// the phase-function bounce. Reference: cuda/volume_utils.cuh (descriptor, path through the disk-box, sampling, transmittance),
// cuda/volume.cuh (the three kernels), cuda/light_bridges.cuh + light_common.cuh:17-32 (bridges), cuda/math.cuh:1169-1322 (phase functions),
// cuda/bsdf.cuh:302-318,:404-421,:458-474, cuda/direct_lighting.cuh:20-121,:385-403,:521-584, optix/optix_kernel_shadow_volume.cu.
// Two volume types: the fog (scalar scattering, no absorption, a disk-box around the camera) and the ocean's water (RGB Jerlov coefficients,
// everything below the surface). A path carries the stack of the volumes it is in (medium_stack.cuh:29-45) in the upper bits of its sample-id
// word: the fog from k_generate on, the water when the camera starts below the surface or the path refracts through it.
// Numerics contract as everywhere: expf := exp_det, logf := log2_det * ln 2, cbrtf := exp2_det(log2_det / 3).
#pragma once

#include "dev_ocean.h"

LUM_NS_BEGIN

constexpr uint32_t kHitInvalid = 0xFFFFFFFFu, kHitVolumeBase = 0xFFFE0000u, kHitVolumeMax = 0xFFFEFFFFu, kHitTriangleLimit = 0x7FFFFFFFu;  // cuda/utils.cuh:51-63, :84-85
enum VolumeType : uint32_t { kVolumeNone = 0, kVolumeFog = 1, kVolumeOcean = 2 };  // utils.h:39
LUM_DEV bool volume_is_hit(uint32_t instance_id) { return instance_id >= kHitVolumeBase && instance_id <= kHitVolumeMax; }
// The path's volume stack: four 2-bit ids above the 20-bit sample id (hit_id.w), newest lowest (medium_stack_volume_peek / _modify with 16-bit ids).
constexpr uint32_t kSampleIdMask = 0xFFFFFu;
LUM_DEV uint32_t path_sample_id(uint32_t w) { return w & kSampleIdMask; }
LUM_DEV uint32_t volume_stack_peek(uint32_t w, bool previous) { return (w >> (previous ? 22u : 20u)) & 3u; }
LUM_DEV uint32_t volume_stack_modify(uint32_t w, uint32_t id, bool push) {
  const uint32_t stack = (w >> 20) & 0xFFu;
  const uint32_t next = push ? ((stack << 2) | id) & 0xFFu : stack >> 2;
  return (w & kSampleIdMask) | (next << 20);
}
constexpr float kBridgesHgG = 0.85f, kBridgesForwardProb = 0.95f;                                              // light_common.cuh:21-22
constexpr uint32_t kBridgesMaxVertexCount = 15, kBridgeLengthStride = 8;                                       // light_common.cuh:23, device_utils.h:51
// random.cuh:24-66; the volume context draws from LIGHT_SUN<1>, LIGHT_GEO<1>, BSDF<0> (bounce) and BSDF<2> (ambient): material.cuh:76-81
constexpr uint32_t kRndVolumeIntersection = 59, kRndSunInitialVertex = 344, kRndGeoInitialVertex = 358;
constexpr uint32_t kRndVolSunBsdf = 347, kRndVolSunBsdfMethod = 350, kRndVolSunRay = 353, kRndVolSunResampling = 356;
constexpr uint32_t kRndVolGeoResampling = 385, kRndVolTreePrepass = 395, kRndVolTreePostpass = 412;
constexpr uint32_t kRndBridgeDistance = 421, kRndBridgePhase = 486, kRndBridgeLightPoint = 551, kRndBridgeVertexCount = 560;
constexpr uint32_t kRndVolGiDiffuse = 43, kRndVolGiResampling = 51, kRndVolAmbientDiffuse = 45, kRndVolAmbientResampling = 53;

LUM_DEV float log_det(float x) { return log2_det(x) * 0.693147181f; }
LUM_DEV float cbrt_det(float x) { return (x == 0.0f) ? 0.0f : copysignf(exp2_det(log2_det(fabsf(x)) * 0.333333333f), x); }
LUM_DEV float clampf(float x, float a, float b) { return fminf(b, fmaxf(a, x)); }

// ---- descriptors (volume_utils.cuh:8-57); `scattering` is the reference's max_scattering: what distances are sampled with ----
struct Volume { float scattering, dist, max_height, min_height; Col scat, absorb; float max_absorption; uint32_t type; };
LUM_DEV Volume volume_descriptor(const DeviceScene& sc, uint32_t type) {
  Volume v{0.0f, 0.0f, 0.0f, 0.0f, splat(0.0f), splat(0.0f), 0.0f, type};
  if (type == kVolumeFog) {
    v.scattering = 0.001f * sc.fog_density;
    v.scat = splat(v.scattering);
    v.dist = sc.fog_dist; v.max_height = sc.fog_height;
    v.min_height = sc.ocean_active ? ocean_max_height(sc) : -65535.0f;
  }
  else if (type == kVolumeOcean) {
    v.absorb = col(sc.ocean_absorption[0], sc.ocean_absorption[1], sc.ocean_absorption[2]);
    v.scat = col(sc.ocean_scattering[0], sc.ocean_scattering[1], sc.ocean_scattering[2]);
    v.dist = 10000.0f; v.max_height = 65535.0f; v.min_height = -65535.0f;
    v.max_absorption = importance(v.absorb);
    v.scattering = importance(v.scat);
  }
  return v;
}
LUM_DEV Volume fog_volume(const DeviceScene& sc) { return volume_descriptor(sc, kVolumeFog); }

struct VolumePath { float start, length; };  // start >= 0 iff the ray passes through the volume within the limit
LUM_DEV VolumePath volume_compute_path(const DeviceScene& sc, const Volume& vol, V3 origin, V3 ray, float limit, bool ocean_fast_path) {  // volume_utils.cuh:88-170
  const VolumePath none{-kFltMax, 0.0f};
  if (limit <= 0.0f) return none;
  if (vol.max_height <= vol.min_height) return none;
  if (vol.type == kVolumeNone) return none;
  float start_y, end_y;
  if (vol.type == kVolumeOcean) {
    start_y = 0.0f;
    end_y = ocean_fast_path ? limit : ocean_intersection_distance(sc, origin, ray, limit);
  }
  else if (fabsf(ray.y) < 0.005f) {
    if (origin.y >= vol.min_height && origin.y <= vol.max_height) { start_y = 0.0f; end_y = vol.dist; }
    else return none;
  }
  else {
    const float sy1 = (vol.min_height - origin.y) / ray.y, sy2 = (vol.max_height - origin.y) / ray.y;
    start_y = fmaxf(fminf(sy1, sy2), 0.0f);
    end_y = fmaxf(sy1, sy2);
  }
  const float rn = rsqrt_ieee(ray.x * ray.x + ray.z * ray.z);
  const float rx = ray.x * rn, rz = ray.z * rn;
  const float dx = origin.x - sc.cam_pos[0], dz = origin.z - sc.cam_pos[2];
  const float dt = dx * rx + dz * rz;
  const float r2 = vol.dist * vol.dist;
  const float c = (dx * dx + dz * dz) - r2;
  const float kx = dx - rx * dt, kz = dz - rz * dt;
  const float d = r2 - (kx * kx + kz * kz);
  if (d < 0.0f) return none;
  const float sd = sqrtf(d);
  const float q = -dt - copysignf(sd, dt);
  const float t0 = fmaxf(0.0f, c / q), t1 = fmaxf(0.0f, q);
  const float start_xz = fminf(t0, t1), end_xz = fmaxf(t0, t1);
  if (end_xz < start_xz || limit < start_xz) return none;
  const float start = fmaxf(start_xz, start_y);
  const float dist = fminf(fminf(end_xz, end_y) - start, limit - start);
  if (dist < 0.0f) return none;
  return VolumePath{start, dist};
}
// volume_utils.cuh:182-214
LUM_DEV float volume_sample_intersection(const Volume& v, float start, float max_length, float random) {
  const float t = (-log_det(random)) / v.scattering;
  return (t > max_length) ? kFltMax : start + t;
}
LUM_DEV float volume_sample_intersection_pdf(const Volume& v, float start, float t) { return v.scattering * exp_det(-v.scattering * (t - start)); }
LUM_DEV float volume_miss_probability(const Volume& v, float depth) { return exp_det(-v.scattering * depth); }
LUM_DEV float volume_sample_bounded(const Volume& v, float max_length, float random) {
  const float prob_hit_at_max = 1.0f - exp_det(-v.scattering * max_length);
  return -log_det(1.0f - random * prob_hit_at_max) / v.scattering;
}
LUM_DEV float volume_sample_bounded_pdf(const Volume& v, float max_length, float t) {
  const float prob_hit_at_max = 1.0f - exp_det(-v.scattering * max_length);
  return v.scattering * exp_det(-v.scattering * t) / prob_hit_at_max;
}
LUM_DEV Col volume_transmittance_length(const Volume& v, float length) {  // volume_utils.cuh:245-254
  return col(exp_det(-length * (v.absorb.r + v.scat.r)), exp_det(-length * (v.absorb.g + v.scat.g)), exp_det(-length * (v.absorb.b + v.scat.b)));
}
// volume_integrate_transmittance (volume_utils.cuh:292-308) of the volume a vertex is in: 1 without one
LUM_DEV Col volume_transmittance(const DeviceScene& sc, uint32_t type, V3 origin, V3 ray, float depth) {
  if (type == kVolumeNone) return splat(1.0f);
  const Volume v = volume_descriptor(sc, type);
  const VolumePath p = volume_compute_path(sc, v, origin, ray, depth, false);
  return (p.start >= 0.0f) ? volume_transmittance_length(v, p.length) : splat(1.0f);
}

// ---- phase functions (math.cuh:1169-1322) ----
LUM_DEV float draine_phase(float c, float g, float alpha) { return hg_phase(c, g) * ((1.0f + alpha * c * c) / (1.0f + (alpha / 3.0f) * (1.0f + 2.0f * g * g))); }
LUM_DEV float je_phase_function(const float* p, float c) {  // jendersie_eon_phase_function, math.cuh:1234-1239; p = g_hg, g_d, alpha, w_d
  return (1.0f - p[3]) * hg_phase(c, p[0]) + p[3] * draine_phase(c, p[1], p[2]);
}
LUM_DEV float fog_phase_function(const DeviceScene& sc, float c) { return je_phase_function(sc.fog_phase, c); }
LUM_DEV V3 phase_sample_basis(float alpha, float beta, V3 basis) {  // math.cuh:1249-1272
  V3 u1, u2;
  if (basis.z < -0.9999805689f) { u1 = v3(0.0f, -1.0f, 0.0f); u2 = v3(-1.0f, 0.0f, 0.0f); }
  else {
    const float a = 1.0f / (1.0f + basis.z);
    const float b = -basis.x * basis.y * a;
    u1 = v3(1.0f - basis.x * basis.x * a, b, -basis.x);
    u2 = v3(b, 1.0f - basis.y * basis.y * a, -basis.y);
  }
  const V3 sp = sample_ray_sphere(alpha, beta);
  return normalize(v3(sp.x * u1.x + sp.y * u2.x + sp.z * basis.x, sp.x * u1.y + sp.y * u2.y + sp.z * basis.y, sp.x * u1.z + sp.y * u2.z + sp.z * basis.z));
}
LUM_DEV float hg_phase_sample(float g, float r) {
  const float g2 = g * g;
  const float t = (1.0f - g2) / (1.0f - g + 2.0f * g * r);
  return (1.0f + g2 - t * t) / (2.0f * g);
}
LUM_DEV float draine_phase_sample(float g, float alpha, float r) {  // math.cuh:1283-1300
  const float g2 = g * g, g4 = g2 * g2;
  const float t0 = alpha - alpha * g2;
  const float t1 = alpha * g4 - alpha;
  const float t2 = -3.0f * (4.0f * (g4 - g2) + t1 * (1.0f + g2));
  const float t3 = g * (2.0f * r - 1.0f);
  const float t4 = 3.0f * g2 * (1.0f + t3) + alpha * (2.0f + g2 * (1.0f + (1.0f + 2.0f * g2) * t3));
  const float t5 = t0 * (t1 * t2 + t4 * t4) + t1 * t1 * t1;
  const float t6 = t0 * 4.0f * (g4 - g2);
  const float t7 = cbrt_det(t5 + sqrtf(t5 * t5 - t6 * t6 * t6));
  const float t8 = 2.0f * ((t1 + (t6 / t7) + t7) / t0);
  const float t9 = sqrtf(6.0f * (1.0f + g2) + t8);
  const float h = sqrtf(6.0f * (1.0f + g2) - t8 + 8.0f * t4 / (t0 * t9)) - t9;
  return 0.5f * g + ((1.0f / (2.0f * g)) - (1.0f / (8.0f * g)) * (h * h));
}
LUM_DEV V3 je_phase_sample(const float* p, V3 ray, F2 r_dir, float r_choice) {  // jendersie_eon_phase_sample, math.cuh:1311-1323
  const float cos_angle = (r_choice < p[3]) ? draine_phase_sample(p[1], p[2], r_dir.x) : hg_phase_sample(p[0], r_dir.x);
  return phase_sample_basis(cos_angle, r_dir.y, ray);
}
LUM_DEV V3 fog_phase_sample(const DeviceScene& sc, V3 ray, F2 r_dir, float r_choice) { return je_phase_sample(sc.fog_phase, ray, r_dir, r_choice); }

// the phase function is drawn from by volume type (bsdf.cuh:310-314; ocean_phase_sampling, ocean_utils.cuh:412-425)
LUM_DEV V3 volume_phase_sample(const DeviceScene& sc, uint32_t type, V3 ray, F2 r_dir, float r_choice) {
  if (type == kVolumeOcean) return phase_sample_basis(ocean_phase_sample_cos(sc, r_dir.x, r_choice), r_dir.y, ray);
  return fog_phase_sample(sc, ray, r_dir, r_choice);
}

// ---- the volume's shading context (material.cuh:76-89, volume_utils.cuh:310-321) ----
struct VolContext { Volume vol; V3 position, V; uint32_t state; float max_dist; };
template <> struct TreeTargets<VolContext> { static constexpr uint32_t kPrepass = kRndVolTreePrepass, kPostpass = kRndVolTreePostpass; };
LUM_DEV VolContext volume_context(const DeviceScene& sc, uint32_t type, V3 origin, V3 ray, uint32_t state, float max_dist) {
  return VolContext{volume_descriptor(sc, type), origin, ray * -1.0f, state, max_dist};
}
// bsdf_sample<MATERIAL_VOLUME> (bsdf.cuh:302-318); the weight is 1
LUM_DEV V3 volume_bsdf_sample(const DeviceScene& sc, const VolContext& c, const Sampler& smp, uint32_t rnd_resampling, uint32_t rnd_diffuse) {
  const float random_choice = smp.next1(rnd_resampling);
  const F2 random_dir = smp.next2(rnd_diffuse);
  return volume_phase_sample(sc, c.vol.type, c.V * -1.0f, random_dir, random_choice);
}
LUM_DEV float volume_phase_evaluate(const DeviceScene& sc, const VolContext& c, V3 L) {  // volume_utils.cuh:216-243
  const float cos_angle = -dot(c.V, L);
  return (c.vol.type == kVolumeOcean) ? ocean_phase(sc, cos_angle) : fog_phase_function(sc, cos_angle);
}

// volume_sample_sky_dl_initial_vertex (volume_utils.cuh:323-352): moves the context to a vertex on the ray, returns its weight
LUM_DEV Col volume_sky_initial_vertex(VolContext& c, const Sampler& smp) {
  const float dist = volume_sample_bounded(c.vol, c.max_dist, smp.next1(kRndSunInitialVertex));
  c.position = c.position + c.V * -dist;
  const Col w = col(exp_det(-dist * (c.vol.absorb.r + c.vol.scat.r)) * c.vol.scat.r, exp_det(-dist * (c.vol.absorb.g + c.vol.scat.g)) * c.vol.scat.g,
                    exp_det(-dist * (c.vol.absorb.b + c.vol.scat.b)) * c.vol.scat.b);
  return w * (1.0f / volume_sample_bounded_pdf(c.vol, c.max_dist, dist));
}

// direct_lighting_sun_create_task + direct_lighting_sun_direct for a volume vertex (direct_lighting.cuh:20-121, :352-383)
LUM_DEV bool volume_sun_sample(const DeviceScene& sc, const SkyView& sky, const VolContext& c, const Sampler& smp, Col& light_out, V3& dir_out) {
  const V3 sky_pos = world_to_sky(sky, c.position);
  const bool sun_below_horizon = sph_hit_p0(normalize(sky.sun_pos - sky_pos), sky_pos, kSkyEarthRadius);
  const bool inside_earth = length(sky_pos) < kSkyEarthRadius;
  if (sun_below_horizon || inside_earth) return false;
  const F2 random_dir = smp.next2(kRndVolSunBsdf);
  const float random_method = smp.next1(kRndVolSunBsdfMethod);
  const V3 dir_bsdf = volume_phase_sample(sc, c.vol.type, c.V * -1.0f, random_dir, random_method);
  Col light_bsdf = splat(0.0f);
  if (sphere_hit(dir_bsdf, sky_pos, sky.sun_pos, kSkySunRadius)) light_bsdf = sky_sun_color(sky, sky_pos, dir_bsdf) * splat(volume_phase_evaluate(sc, c, dir_bsdf) * 1.0f);
  float solid_angle;
  const V3 dir_sa = sample_sphere(sky.sun_pos, kSkySunRadius, sky_pos, smp.next2(kRndVolSunRay), solid_angle);
  const Col light_sa = sky_sun_color(sky, sky_pos, dir_sa) * splat(volume_phase_evaluate(sc, c, dir_sa) * 1.0f);
  const float target_bsdf = importance(light_bsdf), target_sa = importance(light_sa);
  const float mis_bsdf = solid_angle / (volume_phase_evaluate(sc, c, dir_bsdf) * solid_angle + 1.0f);
  const float mis_sa = solid_angle / (volume_phase_evaluate(sc, c, dir_sa) * solid_angle + 1.0f);
  const float weight_bsdf = target_bsdf * mis_bsdf, weight_sa = target_sa * mis_sa;
  const float sum_weights = weight_bsdf + weight_sa;
  if (sum_weights == 0.0f) return false;
  float target;
  Col light;
  if (smp.next1(kRndVolSunResampling) * sum_weights < weight_bsdf) { dir_out = dir_bsdf; target = target_bsdf; light = light_bsdf; }
  else { dir_out = dir_sa; target = target_sa; light = light_sa; }
  light = light * (sum_weights / target);
  if (target == 0.0f) return false;
  if (importance(light) == 0.0f) return false;
  light_out = light * volume_transmittance(sc, c.vol.type, c.position, dir_out, kFltMax);  // direct_lighting.cuh:104-108
  return true;
}

// sky_color_no_compute (sky.cuh:534-565)
LUM_DEV Col sky_color_no_compute(const DeviceScene& sc, V3 origin, V3 ray, uint32_t state) {
  if (sc.sky_mode == kSkyHdri) return sky_hdri_color(sc, origin, ray, state);
  if (sc.sky_mode == kSkyConstantColor) return col(sc.sky_constant_color[0], sc.sky_constant_color[1], sc.sky_constant_color[2]);
  return splat(0.0f);
}

// ---- bridges (light_bridges.cuh) ----
LUM_DEV V3 bridges_phase_sample(V3 ray, F2 r_dir) { return phase_sample_basis(hg_phase_sample(kBridgesHgG, r_dir.x), r_dir.y, ray); }
LUM_DEV Quat bridges_compute_rotation(V3 initial_vertex, V3 light_point, V3 end_vertex) {  // :16-52
  const V3 target_dir = normalize(light_point - initial_vertex), actual_dir = normalize(end_vertex - initial_vertex);
  const float d = dot(actual_dir, target_dir);
  if (d > 0.999f) return Quat{0.0f, 0.0f, 0.0f, 1.0f};
  if (d < -0.999f) return Quat{1.0f, 0.0f, 0.0f, 0.0f};
  const V3 cr = cross(actual_dir, target_dir);
  Quat r{cr.x, cr.y, cr.z, 1.0f + d};
  const float scale = rsqrt_ieee(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);  // normalize_quaternion, math.cuh:353-364
  r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
  return r;
}
// quaternion_pack then quaternion16_apply's unpacking (math.cuh:1687-1696, :415-423): the task carries 16-bit components of the inverse
LUM_DEV Quat quat_through_16_bits(Quat q) {
  const uint32_t x = (uint32_t) (((1.0f - q.x) * 0x7FFF) + 0.5f) & 0xFFFFu, y = (uint32_t) (((1.0f - q.y) * 0x7FFF) + 0.5f) & 0xFFFFu;
  const uint32_t z = (uint32_t) (((1.0f - q.z) * 0x7FFF) + 0.5f) & 0xFFFFu, w = (uint32_t) (((1.0f + q.w) * 0x7FFF) + 0.5f) & 0xFFFFu;
  return Quat{(x * (1.0f / 0x7FFF)) - 1.0f, (y * (1.0f / 0x7FFF)) - 1.0f, (z * (1.0f / 0x7FFF)) - 1.0f, (w * (1.0f / 0x7FFF)) - 1.0f};
}
LUM_DEV float bridges_log_factorial(uint32_t vertex_count) {  // :54-65
  if (vertex_count == 1) return 0.0f;
  const float n = (float) (vertex_count - 1);
  const float t0 = n * log_det(n);
  const float t1 = (1.0f / 6.0f) * log_det(n * (1.0f + 4.0f * n * (1.0f + 2.0f * n)));
  const float t2 = 0.5f * log_det(kPi);
  return t0 + t1 + t2 - n;
}
LUM_DEV float bridges_vertex_count_importance(const float* lut_all, uint32_t vertex_count, float effective_dist) {  // :67-108
  const float* lut = lut_all + (vertex_count - 1) * 21;
  const float min_dist = lut[0], center_dist = lut[1], max_dist = lut[2];
  if (effective_dist > max_dist) return 0.0f;
  if (effective_dist < min_dist) return lut[3] * effective_dist / min_dist;
  const bool low = effective_dist < center_dist;
  const float low_dist = low ? min_dist : center_dist, high_dist = low ? center_dist : max_dist;
  const float step = (high_dist - low_dist) * 0.25f;
  const uint32_t step_id = (uint32_t) ((effective_dist - low_dist) / step);
  const float floor_dist = low_dist + step_id * step;
  const uint32_t index = low ? (3 + 2 * step_id) : (3 + 2 * (step_id + 4));
  const float y0 = lut[index], dy0 = lut[index + 1], y1 = lut[index + 2], dy1 = lut[index + 3];
  const float t = saturate((effective_dist - floor_dist) / step);
  const float t2 = t * t, t3 = t2 * t;
  const float h00 = 2.0f * t3 - 3.0f * t2 + 1.0f, h10 = t3 - 2.0f * t2 + t, h01 = -2.0f * t3 + 3.0f * t2, h11 = t3 - t2;
  return h00 * y0 + h10 * step * dy0 + h01 * y1 + h11 * step * dy1;
}
LUM_DEV uint32_t bridges_sample_vertex_count(const DeviceScene& sc, const Volume& vol, float light_dist, uint32_t seed, const Sampler& smp, float& pdf) {  // :110-140
  const float effective_dist = light_dist * vol.scattering;
  Reservoir rv;
  rv.random = smp.next1(kRndBridgeVertexCount + seed);
  rv.reset();
  const uint32_t max_num_vertices = min(sc.bridge_max_num_vertices, kBridgesMaxVertexCount);
  uint32_t selected = max_num_vertices - 1;
  for (uint32_t vc = 0; vc < max_num_vertices; vc++) {
    const float imp = bridges_vertex_count_importance(sc.bridge_lut, vc + 1, effective_dist);
    if (rv.add(imp, 1.0f)) selected = vc;
  }
  pdf = (rv.sum_weight > 0.0f) ? rv.selected_target / rv.sum_weight : 1.0f;
  return 1 + selected;
}
// :142-222
LUM_DEV Col bridges_sample_bridge(const DeviceScene& sc, const VolContext& c, V3 light_point, V3 initial_vertex, uint32_t seed, const Sampler& smp, float& path_pdf, V3& end_vertex,
                                    float& scale) {
  const V3 light_vector = light_point - initial_vertex;
  const float target_scale = length(light_vector);
  float vertex_count_pdf;
  const uint32_t vertex_count = bridges_sample_vertex_count(sc, c.vol, target_scale, seed, smp, vertex_count_pdf);
  V3 current_vertex = initial_vertex, current_direction = normalize(light_vector);
  float sum_dist = 0.0f;
  {
    const float dist = -log_det(smp.next1(kRndBridgeDistance + seed * kBridgeLengthStride + 0));
    current_vertex = current_vertex + current_direction * dist;
    sum_dist += dist;
  }
  for (uint32_t i = 1; i < vertex_count; i++) {
    current_direction = bridges_phase_sample(current_direction, smp.next2(kRndBridgePhase + seed * kBridgeLengthStride + i));
    const float dist = -log_det(smp.next1(kRndBridgeDistance + seed * kBridgeLengthStride + i));
    current_vertex = current_vertex + current_direction * dist;
    sum_dist += dist;
  }
  const float actual_scale = length(current_vertex - initial_vertex);
  if (actual_scale == 0.0f) { path_pdf = 0.0f; return splat(0.0f); }
  scale = target_scale / actual_scale;
  sum_dist *= scale;
  end_vertex = current_vertex;
  const Col s = c.vol.scat, a = c.vol.absorb;
  const Col w = col(exp_det(vertex_count * log_det(s.r) - sum_dist * (s.r + a.r)), exp_det(vertex_count * log_det(s.g) - sum_dist * (s.g + a.g)),
                    exp_det(vertex_count * log_det(s.b) - sum_dist * (s.b + a.b)));
  const float log_path_pdf = bridges_log_factorial(vertex_count) - vertex_count * log_det(sum_dist);
  path_pdf = vertex_count_pdf * exp_det(log_path_pdf) * target_scale * target_scale * target_scale;
  return w;
}
LUM_DEV V3 bridges_sample_initial_vertex(const VolContext& c, V3 point_on_light, const Sampler& smp, uint32_t output_id, Col& attenuation, float& pdf) {  // :224-266
  float random_intersection = smp.next1(kRndGeoInitialVertex + output_id);
  const V3 PO = point_on_light - c.position;
  const float dist_to_light = fmaxf(-dot(PO, c.V), 0.0f);
  const float forward_prob = (dist_to_light < c.max_dist) ? kBridgesForwardProb : 1.0f;
  float max_dist, t_offset;
  if (random_intersection < forward_prob) {
    random_intersection = random_intersection / forward_prob;
    max_dist = clampf(dist_to_light, 0.0f, c.max_dist);
    t_offset = 0.0f;
    pdf = forward_prob;
  }
  else {
    random_intersection = (random_intersection - forward_prob) / (1.0f - forward_prob);
    max_dist = c.max_dist - dist_to_light;
    t_offset = dist_to_light;
    pdf = 1.0f - forward_prob;
  }
  const float t = t_offset + volume_sample_bounded(c.vol, max_dist, random_intersection);
  attenuation = col(exp_det(-t * (c.vol.absorb.r + c.vol.scat.r)) * c.vol.scat.r, exp_det(-t * (c.vol.absorb.g + c.vol.scat.g)) * c.vol.scat.g,
                    exp_det(-t * (c.vol.absorb.b + c.vol.scat.b)) * c.vol.scat.b);
  pdf *= volume_sample_bounded_pdf(c.vol, max_dist, t - t_offset);
  return c.position + c.V * -t;
}
// light_triangle.cuh:209-243
LUM_DEV V3 tri_light_sample_bridges(const TriLight& t, F2 random) {
  const float r1 = sqrtf(random.x), r2 = random.y;
  const float u = 1.0f - r1, v = r1 * r2;
  return t.vertex + (t.edge1 * u + t.edge2 * v);
}
LUM_DEV bool tri_light_finalize_bridges(const TriLight& t, V3 origin, V3 point_on_light, V3& ray, float& dist, float& area, F2& uv) {
  const V3 cr = cross(t.edge1, t.edge2);
  area = length(cr) * 0.5f;
  ray = point_on_light - origin;
  if (!t.bidirectional && dot(ray, cr) >= 0.0f) { dist = kFltMax; return false; }
  ray = normalize(ray);
  dist = intersect_triangle(t.vertex, t.edge1, t.edge2, origin, ray, uv);
  return dist != kFltMax;
}

struct BridgeSample { uint32_t light_id; Col color; uint32_t seed; Quat rotation; float scale; };  // LightSampleResult<VOLUME>, light_common.cuh:51-58

// bridges_sample (light_bridges.cuh:268-350): one candidate of the volume's light resampling
LUM_DEV BridgeSample bridges_sample(const DeviceScene& sc, const VolContext& c, const TriLight& light, uint32_t light_id, const Sampler& smp, uint32_t output_id, float& target,
                                    float& weight) {
  BridgeSample res{kLightIdInvalid, splat(0.0f), 0u, Quat{0.0f, 0.0f, 0.0f, 1.0f}, 0.0f};
  target = 0.0f; weight = 1.0f;
  const V3 point_on_light = tri_light_sample_bridges(light, smp.next2(kRndBridgeLightPoint + output_id));
  Col initial_attenuation; float initial_pdf;
  const V3 initial_vertex = bridges_sample_initial_vertex(c, point_on_light, smp, output_id, initial_attenuation, initial_pdf);
  if (initial_pdf == 0.0f || importance(initial_attenuation) == 0.0f) return res;
  V3 light_dir; float area, light_dist; F2 uv;
  tri_light_finalize_bridges(light, initial_vertex, point_on_light, light_dir, light_dist, area, uv);
  if (light_dist == kFltMax || area < kEps) return res;
  Col light_color = tri_light_color(sc, light, uv) * initial_attenuation;
  if (importance(light_color) == 0.0f) return res;
  const V3 light_point = initial_vertex + light_dir * light_dist;
  if (light_point.y < c.vol.min_height || light_point.y > c.vol.max_height) return res;
  float sample_weight = area / initial_pdf;
  float path_pdf, path_scale; V3 path_end;
  const Col path_w = bridges_sample_bridge(sc, c, light_point, initial_vertex, output_id, smp, path_pdf, path_end, path_scale);
  if (path_pdf == 0.0f) return res;
  sample_weight *= 1.0f / path_pdf;
  const Quat rot = bridges_compute_rotation(initial_vertex, light_point, path_end);
  const V3 rotated = qapply(rot, light_dir);
  const float cos_angle = -dot(rotated, c.V);
  light_color = light_color * hg_phase(cos_angle, kBridgesHgG);
  const Col path_weight = path_w * light_color;
  target = importance(path_weight); weight = sample_weight;
  return BridgeSample{light_id, path_weight, output_id, rot, path_scale};
}

// light_tree_importance<VOLUME> (light_tree.cuh:91-122); absorption 0, the factor exp(-0 * depth) is kept as a multiplication
LUM_DEV float tree_importance(const VolContext& c, float power, V3 mean, float std_dev) {
  const V3 PO = mean - c.position;
  const float dist_along_ray = -dot(PO, c.V);
  const float clamped = clampf(dist_along_ray, 0.0f, c.max_dist);
  const V3 perp = PO - c.V * clamped;
  const float perp_sq = dot(perp, perp);
  const float falloff = 1.0f / (perp_sq + std_dev);
  const float variance = std_dev * std_dev;
  const float transmittance_depth = fmaxf(perp_sq + clamped * clamped - variance, 0.0f);
  const float transmittance = exp_det(-c.vol.max_absorption * transmittance_depth);
  const float scattering = 1.0f - exp_det(-c.vol.scattering * (variance + clamped));
  return power * falloff * transmittance * scattering;
}

// light_sample<MATERIAL_VOLUME> (light.cuh:84-159): the eight tree outputs are bridge candidates
LUM_DEV BridgeSample volume_light_sample(const DeviceScene& sc, const VolContext& c, const Sampler& smp) {
  const TreeWork work = tree_prepass(sc, c, smp);
  BridgeSample res{kLightIdInvalid, splat(0.0f), 0u, Quat{0.0f, 0.0f, 0.0f, 1.0f}, 0.0f};
  Reservoir rv;
  rv.random = smp.next1(kRndVolGeoResampling);
  rv.reset();
#pragma nounroll
  for (uint32_t lane = 0; lane < kLightTreeOutputs; lane++) {
    const TreePick pick = tree_postpass(sc, c, smp, lane, work);
    if (pick.light_id == kLightIdInvalid) continue;
    const uint2 handle = sc.light_tri_handles[pick.light_id];
    const TriLight tl = load_tri_light(sc, handle.x, handle.y);
    float target, weight;
    const BridgeSample bs = bridges_sample(sc, c, tl, pick.light_id, smp, lane, target, weight);
    if (rv.add(target, weight * pick.weight)) res = bs;
  }
  res.color = res.color * rv.sampling_weight();
  return res;
}

// The segments of a sampled bridge (bridges_sample_apply_shadowing, light_bridges.cuh:356-446): the path is rebuilt from its seed, rotated by
// the task's 16-bit quaternion and scaled. Returns the number of segments; segment k runs from origin[k] along dir[k] for dist[k].
struct BridgeWalk {
  V3 vertex, dir_sampled, dir;
  float dist, scale;
  Quat rotation;
  uint32_t seed, vertex_count;
  uint2 light;
};
LUM_DEV BridgeWalk bridge_walk_begin(const DeviceScene& sc, const VolContext& c, const BridgeSample& task, const Sampler& smp) {
  BridgeWalk w;
  w.seed = task.seed;
  w.light = sc.light_tri_handles[task.light_id];
  const TriLight light = load_tri_light(sc, w.light.x, w.light.y);
  const V3 point_on_light = tri_light_sample_bridges(light, smp.next2(kRndBridgeLightPoint + w.seed));
  Col att; float ipdf;
  const V3 initial_vertex = bridges_sample_initial_vertex(c, point_on_light, smp, w.seed, att, ipdf);
  V3 light_dir; float area, light_dist; F2 uv;
  tri_light_finalize_bridges(light, initial_vertex, point_on_light, light_dir, light_dist, area, uv);
  const V3 light_vector = light_dir * light_dist;
  float vc_pdf;
  w.vertex_count = bridges_sample_vertex_count(sc, c.vol, length(light_vector), w.seed, smp, vc_pdf);
  w.rotation = quat_through_16_bits(task.rotation);
  w.scale = task.scale;
  w.vertex = initial_vertex;
  w.dir_sampled = normalize(light_vector);
  w.dir = qapply(w.rotation, w.dir_sampled);
  w.dist = -log_det(smp.next1(kRndBridgeDistance + w.seed * kBridgeLengthStride + 0)) * w.scale;
  return w;
}
LUM_DEV void bridge_walk_next(BridgeWalk& w, const Sampler& smp, uint32_t vertex_id) {
  w.vertex = w.vertex + w.dir * w.dist;
  w.dir_sampled = bridges_phase_sample(w.dir_sampled, smp.next2(kRndBridgePhase + w.seed * kBridgeLengthStride + vertex_id));
  w.dir = qapply(w.rotation, w.dir_sampled);
  w.dist = -log_det(smp.next1(kRndBridgeDistance + w.seed * kBridgeLengthStride + vertex_id)) * w.scale;
}

LUM_NS_END
