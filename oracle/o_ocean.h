/*
 * ORACLE (test infrastructure, not product): the ocean surface.
 * Follows /root/reference/src/luminary/device/cuda/ocean_utils.cuh: the height function (:26-111, after TDM's "Seascape"), its normal (:117-159), the
 * ray marcher with an approximate Lipschitz bound and the bracketing solver (:161-287), the water's phase function (:387-449), the Fresnel reflection
 * coefficient (:457-475), the surface's shading context (:477-517). The Jerlov coefficients of the water arrive with the scene.
 * Numerics contract as everywhere: sincosf := o_sincos, asinf := o_asin, IEEE + - x / sqrt otherwise.
 */
#ifndef ORACLE_O_OCEAN_H
#define ORACLE_O_OCEAN_H

#include "o_sky.h"

#define HIT_TYPE_OCEAN 0xFFFFFFFDu
#define OCEAN_ITERATIONS 8

static inline float ocean_max_height(const OracleScene* s) { return s->ocean_height + 1.33f * s->ocean_amplitude; }
static inline float ocean_min_height(const OracleScene* s) { return s->ocean_height; }
static inline float ocean_lipschitz(const OracleScene* s) { return s->ocean_amplitude * 2.0f; }

/* white_noise_offset: the 16-bit Squares generator (random.cuh:196-211, :297-307, :150-154) */
static inline float white_noise_offset(uint32_t offset) {
  const uint32_t key = 0xfcbd6e15u, counter = offset;
  uint32_t x = counter * key, y = counter * key, z = y + key;
  x = x * x + y; x = swap16(x);
  x = x * x + z; x = swap16(x);
  const uint32_t v = ((x * x + y) >> 16) & 0xFFFFu;
  return u2f(0x3F800000u | (v << 7)) - 1.0f;
}
static inline float ocean_hash(float px, float py) {
  const float x = fabsf(px + py * (311.7f / 127.1f));
  return white_noise_offset((x < 4294967040.0f) ? (uint32_t) x : 0xFFFFFFFFu); /* the device's conversion saturates */
}
static inline float ocean_noise(float px, float py) {
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
  const float a = o_lerp(hash1, hash2, fx), b = o_lerp(hash3, hash4, fx);
  return -1.0f + 2.0f * o_lerp(a, b, fy);
}
static inline float ocean_octave(float px, float py) {
  const float offset = ocean_noise(px, py);
  px += offset; py += offset;
  float sin_x, cos_x, sin_y, cos_y;
  o_sincos(px, &sin_x, &cos_x);
  o_sincos(py, &sin_y, &cos_y);
  float w1x = 1.0f - fabsf(sin_x), w1y = 1.0f - fabsf(sin_y);
  const float w2x = fabsf(cos_x), w2y = fabsf(cos_y);
  w1x = o_lerp(w1x, w2x, w1x);
  w1y = o_lerp(w1y, w2y, w1y);
  float octave = 1.0f - sqrtf(w1x * w1y);
  octave *= octave;
  return octave;
}
static inline float ocean_get_height(const OracleScene* s, vec3 p, int steps) {
  float amplitude = 1.0f, frequency = s->ocean_frequency;
  float qx = p.x * 0.75f, qy = p.z;
  float h = 0.0f;
  for (int i = 0; i < steps; i++) {
    h += ocean_octave(qx * frequency, qy * frequency) * amplitude;
    const float u = qx, v = qy;
    qx = 1.6f * u - 1.2f * v;
    qy = 1.2f * u + 1.6f * v;
    frequency *= 1.9f;
    amplitude *= 0.22f;
  }
  return h * s->ocean_amplitude;
}
static inline float ocean_relative_height(const OracleScene* s, vec3 p, int steps) { return p.y - (s->ocean_height + ocean_get_height(s, p, steps)); }
static inline bool ocean_is_underwater(const OracleScene* s, vec3 p) { return ocean_relative_height(s, p, OCEAN_ITERATIONS) < 0.0f; }

static inline vec3 ocean_get_normal(const OracleScene* s, vec3 p) { /* :117-140, Sobel filter */
  if (s->ocean_amplitude == 0.0f) return v3(0.0f, 1.0f, 0.0f);
  const float d = (ocean_lipschitz(s) + v_len(p) + 1.0f) * O_EPS * 16.0f;
  const float h0 = ocean_get_height(s, v_add(p, v3(-d, 0.0f, d)), OCEAN_ITERATIONS), h1 = ocean_get_height(s, v_add(p, v3(0.0f, 0.0f, d)), OCEAN_ITERATIONS);
  const float h2 = ocean_get_height(s, v_add(p, v3(d, 0.0f, d)), OCEAN_ITERATIONS), h3 = ocean_get_height(s, v_add(p, v3(-d, 0.0f, 0.0f)), OCEAN_ITERATIONS);
  const float h4 = ocean_get_height(s, v_add(p, v3(d, 0.0f, 0.0f)), OCEAN_ITERATIONS), h5 = ocean_get_height(s, v_add(p, v3(-d, 0.0f, -d)), OCEAN_ITERATIONS);
  const float h6 = ocean_get_height(s, v_add(p, v3(0.0f, 0.0f, -d)), OCEAN_ITERATIONS), h7 = ocean_get_height(s, v_add(p, v3(d, 0.0f, -d)), OCEAN_ITERATIONS);
  vec3 n;
  n.x = ((h5 + 2.0f * h3 + h0) - (h7 + 2.0f * h4 + h2)) * (1.0f / 8.0f);
  n.y = d;
  n.z = ((h5 + 2.0f * h6 + h7) - (h0 + 2.0f * h1 + h2)) * (1.0f / 8.0f);
  return v_norm(n);
}
static inline vec3 ocean_get_normal_fast(const OracleScene* s, vec3 p) { /* :142-159 */
  if (s->ocean_amplitude == 0.0f) return v3(0.0f, 1.0f, 0.0f);
  const float d = (ocean_lipschitz(s) + v_len(p) + 1.0f) * O_EPS * 16.0f;
  const float h0 = ocean_get_height(s, v_add(p, v3(0.0f, 0.0f, d)), OCEAN_ITERATIONS), h1 = ocean_get_height(s, v_add(p, v3(-d, 0.0f, 0.0f)), OCEAN_ITERATIONS);
  const float h2 = ocean_get_height(s, v_add(p, v3(d, 0.0f, 0.0f)), OCEAN_ITERATIONS), h3 = ocean_get_height(s, v_add(p, v3(0.0f, 0.0f, -d)), OCEAN_ITERATIONS);
  return v_norm(v3((h1 - h2) * (1.0f / 4.0f), d, (h3 - h0) * (1.0f / 4.0f)));
}

static inline float ocean_shell_radius(const OracleScene* s, const OSky* sky) { /* world_to_sky_scale(OCEAN_MAX_HEIGHT) + SKY_WORLD_REFERENCE_HEIGHT */
  return ocean_max_height(s) * 0.001f + v_len(world_to_sky(sky, v3(0.0f, 0.0f, 0.0f)));
}
static inline float ocean_far_distance(const OracleScene* s, const OSky* sky, vec3 origin, vec3 ray) { /* :161-181 */
  if (!sph_hit_p0(ray, world_to_sky(sky, origin), ocean_shell_radius(s, sky))) return FLT_MAX;
  if (fabsf(ray.y) < O_EPS) return FLT_MAX;
  const float d1 = ocean_min_height(s) - origin.y, d2 = ocean_max_height(s) - origin.y;
  const float inv_ray = 1.0f / ray.y;
  const float t = fmaxf(d1 * inv_ray, d2 * inv_ray);
  return (t >= O_EPS) ? t : FLT_MAX;
}
static inline float ocean_short_distance(const OracleScene* s, const OSky* sky, vec3 origin, vec3 ray) { /* :183-205 */
  if (!sph_hit_p0(ray, world_to_sky(sky, origin), ocean_shell_radius(s, sky))) return FLT_MAX;
  if (fabsf(ray.y) < O_EPS) return (origin.y >= ocean_min_height(s) && origin.y <= ocean_max_height(s)) ? 0.0f : FLT_MAX;
  const float d1 = ocean_min_height(s) - origin.y, d2 = ocean_max_height(s) - origin.y;
  const float inv_ray = 1.0f / ray.y;
  const float s1 = d1 * inv_ray, s2 = d2 * inv_ray;
  if (s1 < 0.0f && s2 < 0.0f) return FLT_MAX;
  return (s1 * s2 < 0.0f) ? fmaxf(s1, s2) : fminf(s1, s2);
}
static inline float ocean_intersection_solver(const OracleScene* s, vec3 origin, vec3 ray, float start, float limit) { /* :207-267 */
  if (start >= limit) return FLT_MAX;
  const float target_residual = 1e-4f;
  float min = start, max = limit;
  float residual_at_max = FLT_MAX, residual_at_min = 0.0f; /* the reference leaves residual_at_min unset when its first loop does not run (step_count < 1) */
  const int32_t step_count = (int32_t) ((s->ocean_amplitude * s->ocean_amplitude - 0.0f) / (1.0f - 0.0f) * (16.0f - 4.0f) + 4.0f); /* remap(a^2, 0, 1, 4, 16) */
  float t = start, last_residual = 0.0f;
  const float slope_confidence_factor = fminf(8.0f / ocean_lipschitz(s), (limit - start) * (1.0f / step_count));
  for (int i = 0; i < step_count; i++) {
    const float residual_at_t = ocean_relative_height(s, v_add(origin, v_scale(ray, t)), OCEAN_ITERATIONS);
    if (last_residual * residual_at_t < 0.0f) { max = t; residual_at_max = residual_at_t; break; }
    last_residual = residual_at_t;
    min = t; residual_at_min = residual_at_t;
    t += fabsf(residual_at_t) * slope_confidence_factor;
  }
  if (residual_at_max == FLT_MAX) residual_at_max = ocean_relative_height(s, v_add(origin, v_scale(ray, limit)), OCEAN_ITERATIONS);
  for (int i = 0; i < step_count; i++) {
    const float step = residual_at_min / (residual_at_min - residual_at_max);
    const float mid = o_lerp(min, max, fminf(0.95f, fmaxf(0.05f, step)));
    const float residual_at_mid = ocean_relative_height(s, v_add(origin, v_scale(ray, mid)), OCEAN_ITERATIONS);
    if (fabsf(residual_at_mid) < target_residual) return (mid >= start) ? mid : FLT_MAX;
    if (residual_at_mid * residual_at_min < 0.0f) { max = mid; residual_at_max = residual_at_mid; }
    else { min = mid; residual_at_min = residual_at_mid; }
  }
  if (residual_at_max * residual_at_min < 0.0f) return 0.5f * (min + max);
  return FLT_MAX;
}
static inline float ocean_intersection_distance(const OracleScene* s, vec3 origin, vec3 ray, float limit) { /* :269-287 */
  const OSky sky = osky_view(s);
  float start = 0.0f;
  if (origin.y < ocean_min_height(s) || origin.y > ocean_max_height(s)) {
    const float short_distance = ocean_short_distance(s, &sky, origin, ray);
    if (short_distance == FLT_MAX) return FLT_MAX;
    start = short_distance;
  }
  if (s->ocean_amplitude == 0.0f) return start;
  const float end = fminf(limit, ocean_far_distance(s, &sky, origin, ray));
  return ocean_intersection_solver(s, origin, ray, start, end);
}

/* ---- the water's phase function (:387-449): Henyey-Greenstein with g = 0 (molecules) and g = 0.924 (particles), mixed by the water type ---- */
static inline float ocean_phase(const OracleScene* s, float cos_angle) {
  const float w = s->ocean_molecular_weight;
  return hg_phase(cos_angle, 0.0f) * w + hg_phase(cos_angle, 0.924f) * (1.0f - w);
}
static inline float ocean_phase_sample_cos(const OracleScene* s, float r_dir, float r_choice) {
  if (r_choice < s->ocean_molecular_weight) return 2.0f * r_dir - 1.0f;
  const float g = 0.924f;
  float denom = (1.0f - g + 2.0f * g * r_dir);
  if (fabsf(denom) < O_EPS) denom = copysignf(O_EPS, denom);
  const float sq = (1.0f - g * g) / denom;
  return (1.0f + g * g - sq * sq) / (2.0f * g);
}

/* :457-475 */
static inline float ocean_reflection_coefficient(vec3 normal, vec3 ray, vec3 refraction, float index_in_over_out) {
  const float NdotV = -v_dot(ray, normal), NdotT = -v_dot(refraction, normal);
  const float s1 = index_in_over_out * NdotV, s2 = 1.0f * NdotT;
  const float p1 = index_in_over_out * NdotT, p2 = 1.0f * NdotV;
  float rs = (s1 - s2) / (s1 + s2), rp = (p1 - p2) / (p1 + p2);
  rs *= rs; rp *= rp;
  return o_saturate(0.5f * (rs + rp));
}

#endif
