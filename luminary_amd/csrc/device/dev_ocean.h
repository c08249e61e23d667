// The ocean surface: a procedural height field (after TDM's "Seascape") that rays march with an approximate Lipschitz bound, its normal, the
// water's phase function and the Fresnel reflection coefficient of the interface. The water below it is the second volume type (dev_volume.h).
// Reference: cuda/ocean_utils.cuh:26-111 (height), :117-159 (normals), :161-287 (ray marcher + bracketing solver), :387-449 (phase function),
// :457-475 (Fresnel), :477-523 (surface context, origin shift); the Jerlov coefficients of the water type arrive with the scene.
// Numerics contract: sincosf := sincos_det, asinf := atan2_det(x, sqrt(1 - x^2)), otherwise IEEE + - x / sqrt.
#pragma once

#include "dev_sky.h"

LUM_NS_BEGIN

constexpr uint32_t kHitOcean = 0xFFFFFFFDu;  // cuda/utils.cuh:53
constexpr int kOceanIterations = 8;         // OCEAN_ITERATIONS_INTERSECTION / _NORMAL / _NORMAL_CAUSTICS

LUM_DEV float ocean_max_height(const DeviceScene& sc) { return sc.ocean_height + 1.33f * sc.ocean_amplitude; }
LUM_DEV float ocean_min_height(const DeviceScene& sc) { return sc.ocean_height; }
LUM_DEV float ocean_lipschitz(const DeviceScene& sc) { return sc.ocean_amplitude * 2.0f; }

// white_noise_offset: the 16-bit Squares generator (random.cuh:196-211, :297-307, :150-154)
LUM_DEV float white_noise_offset(uint32_t offset) {
  const uint32_t key = 0xfcbd6e15u, counter = offset;
  uint32_t x = counter * key, y = counter * key, z = y + key;
  x = x * x + y; x = swap_halves(x);
  x = x * x + z; x = swap_halves(x);
  const uint32_t v = ((x * x + y) >> 16) & 0xFFFFu;
  return bitsf(0x3F800000u | (v << 7)) - 1.0f;
}
LUM_DEV float ocean_hash(float px, float py) {
  const float x = fabsf(px + py * (311.7f / 127.1f));
  return white_noise_offset((x < 4294967040.0f) ? (uint32_t) x : 0xFFFFFFFFu);  // saturating conversion
}
LUM_DEV float ocean_noise(float px, float py) {
  float ix = floorf(px), iy = floorf(py);
  float fx = px - ix, fy = py - iy;
  fx = fx * fx * (3.0f - 2.0f * fx);
  fy = fy * fy * (3.0f - 2.0f * fy);
  const float hash1 = ocean_hash(ix, iy);
  ix += 1.0f;
  const float hash2 = ocean_hash(ix, iy);
  iy += 1.0f;
  const float hash4 = ocean_hash(ix, iy);
  ix -= 1.0f;
  const float hash3 = ocean_hash(ix, iy);
  const float a = lerpf(hash1, hash2, fx), b = lerpf(hash3, hash4, fx);
  return -1.0f + 2.0f * lerpf(a, b, fy);
}
LUM_DEV float ocean_octave(float px, float py) {
  const float offset = ocean_noise(px, py);
  px += offset; py += offset;
  float sin_x, cos_x, sin_y, cos_y;
  sincos_det(px, sin_x, cos_x);
  sincos_det(py, sin_y, cos_y);
  float w1x = 1.0f - fabsf(sin_x), w1y = 1.0f - fabsf(sin_y);
  const float w2x = fabsf(cos_x), w2y = fabsf(cos_y);
  w1x = lerpf(w1x, w2x, w1x);
  w1y = lerpf(w1y, w2y, w1y);
  float octave = 1.0f - sqrtf(w1x * w1y);
  octave *= octave;
  return octave;
}
LUM_DEV float ocean_get_height(const DeviceScene& sc, V3 p, int steps) {
  float amplitude = 1.0f, frequency = sc.ocean_frequency;
  float qx = p.x * 0.75f, qy = p.z;
  float h = 0.0f;
#pragma nounroll
  for (int i = 0; i < steps; i++) {
    h += ocean_octave(qx * frequency, qy * frequency) * amplitude;
    const float u = qx, v = qy;
    qx = 1.6f * u - 1.2f * v;
    qy = 1.2f * u + 1.6f * v;
    frequency *= 1.9f;
    amplitude *= 0.22f;
  }
  return h * sc.ocean_amplitude;
}
LUM_DEV float ocean_relative_height(const DeviceScene& sc, V3 p, int steps) { return p.y - (sc.ocean_height + ocean_get_height(sc, p, steps)); }
LUM_DEV bool ocean_is_underwater(const DeviceScene& sc, V3 p) { return ocean_relative_height(sc, p, kOceanIterations) < 0.0f; }

LUM_DEV V3 ocean_get_normal(const DeviceScene& sc, V3 p) {  // :117-140, Sobel filter
  if (sc.ocean_amplitude == 0.0f) return v3(0.0f, 1.0f, 0.0f);
  const float d = (ocean_lipschitz(sc) + length(p) + 1.0f) * kEps * 16.0f;
  const float h0 = ocean_get_height(sc, p + v3(-d, 0.0f, d), kOceanIterations), h1 = ocean_get_height(sc, p + v3(0.0f, 0.0f, d), kOceanIterations);
  const float h2 = ocean_get_height(sc, p + v3(d, 0.0f, d), kOceanIterations), h3 = ocean_get_height(sc, p + v3(-d, 0.0f, 0.0f), kOceanIterations);
  const float h4 = ocean_get_height(sc, p + v3(d, 0.0f, 0.0f), kOceanIterations), h5 = ocean_get_height(sc, p + v3(-d, 0.0f, -d), kOceanIterations);
  const float h6 = ocean_get_height(sc, p + v3(0.0f, 0.0f, -d), kOceanIterations), h7 = ocean_get_height(sc, p + v3(d, 0.0f, -d), kOceanIterations);
  V3 n;
  n.x = ((h5 + 2.0f * h3 + h0) - (h7 + 2.0f * h4 + h2)) * (1.0f / 8.0f);
  n.y = d;
  n.z = ((h5 + 2.0f * h6 + h7) - (h0 + 2.0f * h1 + h2)) * (1.0f / 8.0f);
  return normalize(n);
}
LUM_DEV V3 ocean_get_normal_fast(const DeviceScene& sc, V3 p) {  // :142-159
  if (sc.ocean_amplitude == 0.0f) return v3(0.0f, 1.0f, 0.0f);
  const float d = (ocean_lipschitz(sc) + length(p) + 1.0f) * kEps * 16.0f;
  const float h0 = ocean_get_height(sc, p + v3(0.0f, 0.0f, d), kOceanIterations), h1 = ocean_get_height(sc, p + v3(-d, 0.0f, 0.0f), kOceanIterations);
  const float h2 = ocean_get_height(sc, p + v3(d, 0.0f, 0.0f), kOceanIterations), h3 = ocean_get_height(sc, p + v3(0.0f, 0.0f, -d), kOceanIterations);
  return normalize(v3((h1 - h2) * (1.0f / 4.0f), d, (h3 - h0) * (1.0f / 4.0f)));
}

LUM_DEV float ocean_shell_radius(const DeviceScene& sc, const SkyView& sky) {  // world_to_sky_scale(OCEAN_MAX_HEIGHT) + SKY_WORLD_REFERENCE_HEIGHT
  return ocean_max_height(sc) * 0.001f + length(world_to_sky(sky, v3(0.0f, 0.0f, 0.0f)));
}
LUM_DEV float ocean_far_distance(const DeviceScene& sc, const SkyView& sky, V3 origin, V3 ray) {  // :161-181
  if (!sph_hit_p0(ray, world_to_sky(sky, origin), ocean_shell_radius(sc, sky))) return kFltMax;
  if (fabsf(ray.y) < kEps) return kFltMax;
  const float d1 = ocean_min_height(sc) - origin.y, d2 = ocean_max_height(sc) - origin.y;
  const float inv_ray = 1.0f / ray.y;
  const float t = fmaxf(d1 * inv_ray, d2 * inv_ray);
  return (t >= kEps) ? t : kFltMax;
}
LUM_DEV float ocean_short_distance(const DeviceScene& sc, const SkyView& sky, V3 origin, V3 ray) {  // :183-205
  if (!sph_hit_p0(ray, world_to_sky(sky, origin), ocean_shell_radius(sc, sky))) return kFltMax;
  if (fabsf(ray.y) < kEps) return (origin.y >= ocean_min_height(sc) && origin.y <= ocean_max_height(sc)) ? 0.0f : kFltMax;
  const float d1 = ocean_min_height(sc) - origin.y, d2 = ocean_max_height(sc) - origin.y;
  const float inv_ray = 1.0f / ray.y;
  const float s1 = d1 * inv_ray, s2 = d2 * inv_ray;
  if (s1 < 0.0f && s2 < 0.0f) return kFltMax;
  return (s1 * s2 < 0.0f) ? fmaxf(s1, s2) : fminf(s1, s2);
}
LUM_DEV float ocean_intersection_solver(const DeviceScene& sc, V3 origin, V3 ray, float start, float limit) {  // :207-267
  if (start >= limit) return kFltMax;
  const float target_residual = 1e-4f;
  float lo = start, hi = limit;
  float residual_at_max = kFltMax, residual_at_min = 0.0f;
  const int step_count = (int) ((sc.ocean_amplitude * sc.ocean_amplitude - 0.0f) / (1.0f - 0.0f) * (16.0f - 4.0f) + 4.0f);  // remap(a^2, 0, 1, 4, 16)
  float t = start, last_residual = 0.0f;
  const float slope_confidence_factor = fminf(8.0f / ocean_lipschitz(sc), (limit - start) * (1.0f / step_count));
#pragma nounroll
  for (int i = 0; i < step_count; i++) {
    const float residual_at_t = ocean_relative_height(sc, origin + ray * t, kOceanIterations);
    if (last_residual * residual_at_t < 0.0f) { hi = t; residual_at_max = residual_at_t; break; }
    last_residual = residual_at_t;
    lo = t; residual_at_min = residual_at_t;
    t += fabsf(residual_at_t) * slope_confidence_factor;
  }
  if (residual_at_max == kFltMax) residual_at_max = ocean_relative_height(sc, origin + ray * limit, kOceanIterations);
#pragma nounroll
  for (int i = 0; i < step_count; i++) {
    const float step = residual_at_min / (residual_at_min - residual_at_max);
    const float mid = lerpf(lo, hi, fminf(0.95f, fmaxf(0.05f, step)));
    const float residual_at_mid = ocean_relative_height(sc, origin + ray * mid, kOceanIterations);
    if (fabsf(residual_at_mid) < target_residual) return (mid >= start) ? mid : kFltMax;
    if (residual_at_mid * residual_at_min < 0.0f) { hi = mid; residual_at_max = residual_at_mid; }
    else { lo = mid; residual_at_min = residual_at_mid; }
  }
  if (residual_at_max * residual_at_min < 0.0f) return 0.5f * (lo + hi);
  return kFltMax;
}
LUM_DEV float ocean_intersection_distance(const DeviceScene& sc, V3 origin, V3 ray, float limit) {  // :269-287
  const SkyView sky = sky_view(sc);
  float start = 0.0f;
  if (origin.y < ocean_min_height(sc) || origin.y > ocean_max_height(sc)) {
    const float short_distance = ocean_short_distance(sc, sky, origin, ray);
    if (short_distance == kFltMax) return kFltMax;
    start = short_distance;
  }
  if (sc.ocean_amplitude == 0.0f) return start;
  const float end = fminf(limit, ocean_far_distance(sc, sky, origin, ray));
  return ocean_intersection_solver(sc, origin, ray, start, end);
}

// ---- the water's phase function (:387-449) ----
LUM_DEV float ocean_phase(const DeviceScene& sc, float cos_angle) {
  const float w = sc.ocean_molecular_weight;
  return hg_phase(cos_angle, 0.0f) * w + hg_phase(cos_angle, 0.924f) * (1.0f - w);
}
LUM_DEV float ocean_phase_sample_cos(const DeviceScene& sc, float r_dir, float r_choice) {
  if (r_choice < sc.ocean_molecular_weight) return 2.0f * r_dir - 1.0f;
  const float g = 0.924f;
  float denom = (1.0f - g + 2.0f * g * r_dir);
  if (fabsf(denom) < kEps) denom = copysignf(kEps, denom);
  const float sq = (1.0f - g * g) / denom;
  return (1.0f + g * g - sq * sq) / (2.0f * g);
}

LUM_DEV float ocean_reflection_coefficient(V3 normal, V3 ray, V3 refraction, float index_in_over_out) {  // :457-475
  const float NdotV = -dot(ray, normal), NdotT = -dot(refraction, normal);
  const float s1 = index_in_over_out * NdotV, s2 = 1.0f * NdotT;
  const float p1 = index_in_over_out * NdotT, p2 = 1.0f * NdotV;
  float rs = (s1 - s2) / (s1 + s2), rp = (p1 - p2) / (p1 + p2);
  rs *= rs; rp *= rp;
  return saturate(0.5f * (rs + rp));
}

LUM_NS_END
