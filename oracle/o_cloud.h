/*
 * ORACLE (test infrastructure, not product): clouds.
 * Follows /root/reference/src/luminary/device/cuda/cloud_noise.cuh (the three noise textures: tiling Perlin and inverted Worley octaves),
 * cloud_utils.cuh (layers, weather map, density function), cloud.cuh (the ray march with sun and ambient light and its multi-octave approximation of
 * multiple scattering; the ordering of the three layers along a ray) and cloud_shadow.cuh (the binary shadow the layers cast into the sky's
 * in-scattering march). Marched per path in sky mode DEFAULT only (device_manager.c:474); in HDRI mode they are baked into the panorama (o_sky.h).
 * Textures: RGBA8, normalised coordinates, wrap addressing, linear filter; the reference creates them without mip levels (texture.c:85,
 * device_texture.c:92-95), so the LOD bias its lookups pass has no effect and is left out. Filter weights are exact floats (as for the 2-D textures,
 * o_light.h), trilinear in the order x, y, z.
 * Numerics contract as everywhere: sinf := o_sincos, powf := o_pow, expf := o_exp, rsqrtf := o_rsqrt; fmodf is exact in IEEE arithmetic.
 */
#ifndef ORACLE_O_CLOUD_H
#define ORACLE_O_CLOUD_H

#include "o_volume.h"

#define CLOUD_SHAPE_RES 128
#define CLOUD_DETAIL_RES 32
#define CLOUD_WEATHER_RES 1024
#define RT_CLOUD_STEP_OFFSET 67u
#define RT_CLOUD_STEP_COUNT 71u
#define RT_CLOUD_DIR 75u
#define CLOUD_SCATTERING_DENSITY (1000.0f * 0.1f * 0.9f)
#define CLOUD_EXTINCTION_DENSITY (1000.0f * 0.1f)
#define CLOUD_WEATHER_CUTOFF 0.05f

/* ---- math.cuh:33-72 ---- */
static inline float c_fract(float x) { return x - floorf(x); }
static inline float c_remap(float value, float src_low, float src_high, float dst_low, float dst_high) { return (value - src_low) / (src_high - src_low) * (dst_high - dst_low) + dst_low; }
static inline float c_remap01(float value, float src_low, float src_high) { return o_saturate(c_remap(value, src_low, src_high, 0.0f, 1.0f)); }
static inline float c_step(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
static inline float c_smoothstep(float x, float edge0, float edge1) { const float t = c_remap01(x, edge0, edge1); return t * t * (3.0f - 2.0f * t); }
static inline float c_sin(float x) { float s, c; o_sincos(x, &s, &c); return s; }

/* ---- noise (cloud_noise.cuh) ---- */
static inline float interp_cubic_d2(float x) { return x * x * x * (x * (x * 6.0f - 15.0f) + 10.0f); }
static inline void perlin_hash(vec3 grid, float scale, bool tile, float low0[4], float low1[4], float low2[4], float high0[4], float high1[4], float high2[4]) {
  const float offset_x = 50.0f, offset_y = 161.0f, domain = 69.0f;
  const float largef[3] = {635.298681f, 682.357502f, 668.926525f}, z_inc[3] = {48.500388f, 65.294118f, 63.934599f};
  grid.x -= floorf(grid.x / domain) * domain;
  grid.y -= floorf(grid.y / domain) * domain;
  grid.z -= floorf(grid.z / domain) * domain;
  const float d = domain - 1.5f;
  float inc_x = c_step(grid.x, d) * (grid.x + 1.0f), inc_y = c_step(grid.y, d) * (grid.y + 1.0f), inc_z = c_step(grid.z, d) * (grid.z + 1.0f);
  if (tile) { inc_x = fmodf(inc_x, scale); inc_y = fmodf(inc_y, scale); inc_z = fmodf(inc_z, scale); }
  float p[4] = {grid.x + offset_x, grid.y + offset_y, inc_x + offset_x, inc_y + offset_y};
  for (int k = 0; k < 4; k++) p[k] *= p[k];
  const float q[4] = {p[0] * p[1], p[2] * p[1], p[0] * p[3], p[2] * p[3]};
  float low[3], high[3];
  for (int k = 0; k < 3; k++) { low[k] = 1.0f / (largef[k] + grid.z * z_inc[k]); high[k] = 1.0f / (largef[k] + inc_z * z_inc[k]); }
  for (int k = 0; k < 4; k++) {
    low0[k] = c_fract(q[k] * low[0]); low1[k] = c_fract(q[k] * low[1]); low2[k] = c_fract(q[k] * low[2]);
    high0[k] = c_fract(q[k] * high[0]); high1[k] = c_fract(q[k] * high[1]); high2[k] = c_fract(q[k] * high[2]);
  }
}
static inline float perlin(vec3 p, float scale, bool tile) {
  p = v_scale(p, scale);
  const vec3 p1 = v3(floorf(p.x), floorf(p.y), floorf(p.z));
  const vec3 pf = v_sub(p, p1);
  const vec3 pm = v3(pf.x + -1.0f, pf.y + -1.0f, pf.z + -1.0f);
  float hx0[4], hy0[4], hz0[4], hx1[4], hy1[4], hz1[4];
  perlin_hash(p1, scale, tile, hx0, hy0, hz0, hx1, hy1, hz1);
  float gx0[4], gy0[4], gz0[4], gx1[4], gy1[4], gz1[4];
  for (int k = 0; k < 4; k++) {
    gx0[k] = hx0[k] - 0.49999f; gy0[k] = hy0[k] - 0.49999f; gz0[k] = hz0[k] - 0.49999f;
    gx1[k] = hx1[k] - 0.49999f; gy1[k] = hy1[k] - 0.49999f; gz1[k] = hz1[k] - 0.49999f;
  }
  const float fx[4] = {pf.x, pm.x, pf.x, pm.x}, fy[4] = {pf.y, pf.y, pm.y, pm.y};
  float grad0[4], grad1[4];
  for (int k = 0; k < 4; k++) {
    grad0[k] = o_rsqrt(gx0[k] * gx0[k] + gy0[k] * gy0[k] + gz0[k] * gz0[k]) * (fx[k] * gx0[k] + fy[k] * gy0[k] + pf.z * gz0[k]);
    grad1[k] = o_rsqrt(gx1[k] * gx1[k] + gy1[k] * gy1[k] + gz1[k] * gz1[k]) * (fx[k] * gx1[k] + fy[k] * gy1[k] + pm.z * gz1[k]);
  }
  const float bx = interp_cubic_d2(pf.x), by = interp_cubic_d2(pf.y), bz = interp_cubic_d2(pf.z);
  float res[4];
  for (int k = 0; k < 4; k++) res[k] = o_lerp(grad0[k], grad1[k], bz);
  const float b2z = 1.0f - bx, b2w = 1.0f - by;
  float final = res[0] * b2z * b2w + res[1] * bx * b2w + res[2] * b2z * by + res[3] * bx * by;
  final /= sqrtf(0.75f);
  return ((final * 1.5f) + 1.0f) * 0.5f;
}
static inline float perlin_octaves(vec3 p, float scale, int octaves, bool tile) {
  float frequency = 1.0f, persistence = 1.0f, value = 0.0f;
  for (int i = 0; i < octaves; i++) {
    value += persistence * perlin(p, scale * frequency, tile);
    persistence *= 0.5f;
    frequency *= 2.0f;
  }
  return value;
}
static inline vec3 voronoi_hash(vec3 x, float scale) {
  x.x = fmodf(x.x, scale); x.y = fmodf(x.y, scale); x.z = fmodf(x.z, scale);
  x = v3(v_dot(x, v3(127.1f, 311.7f, 74.7f)), v_dot(x, v3(269.5f, 183.3f, 246.1f)), v_dot(x, v3(113.5f, 271.9f, 124.6f)));
  const float h = 43758.5453123f;
  return v3(c_fract(c_sin(x.x) * h), c_fract(c_sin(x.y) * h), c_fract(c_sin(x.z) * h));
}
static inline float voronoi_x(vec3 x, float scale, float seed, bool inverted) { /* the callers use the nearest distance only */
  x = v_scale(x, scale);
  x = v3(x.x + 0.5f, x.y + 0.5f, x.z + 0.5f);
  const vec3 p = v3(floorf(x.x), floorf(x.y), floorf(x.z));
  const vec3 f = v3(c_fract(x.x), c_fract(x.y), c_fract(x.z));
  float res_x = 1.0f;
  for (int k = -1; k <= 1; k++)
    for (int j = -1; j <= 1; j++)
      for (int i = -1; i <= 1; i++) {
        const vec3 b = v3((float) i, (float) j, (float) k);
        const vec3 pb = v_add(p, b);
        const vec3 r = v_add(v_sub(b, f), voronoi_hash(v3(pb.x + seed * 10.0f, pb.y + seed * 10.0f, pb.z + seed * 10.0f), scale));
        const float d = v_dot(r, r);
        if (d < res_x) res_x = d;
      }
  return inverted ? 1.0f - res_x : res_x;
}
static inline float worley_octaves(vec3 p, float scale, int octaves, float seed, float persistence) {
  float value = o_saturate(voronoi_x(p, scale, seed, true));
  float frequency = 2.0f;
  for (int i = 1; i < octaves; i++) {
    value -= persistence * o_saturate(voronoi_x(p, scale * frequency, seed, false));
    frequency *= 2.0f;
  }
  return value;
}
static inline float dilate_perlin_worley(float p, float w, float x) {
  const float curve = 0.75f;
  if (x < 0.5f) {
    x *= 2.0f;
    const float n = p + w * x;
    return n * o_lerp(1.0f, 0.5f, o_pow(x, curve));
  }
  x = 2.0f * (x - 0.5f);
  const float n = w + p * (1.0f - x);
  return n * o_lerp(0.5f, 1.0f, o_pow(x, 1.0f / curve));
}
static inline uint32_t cloud_pack(float a, float b, float c, float d) { /* make_uchar4 of float products: conversion truncates */
  return (uint32_t) (uint8_t) (a) | ((uint32_t) (uint8_t) (b) << 8) | ((uint32_t) (uint8_t) (c) << 16) | ((uint32_t) (uint8_t) (d) << 24);
}
static uint32_t cloud_shape_texel(uint32_t x, uint32_t y, uint32_t z, uint32_t dim) {
  const float sc = 1.0f / dim;
  const vec3 s = v3(x * sc, y * sc, z * sc);
  const float size_scale = 1.0f;
  float perlin_dilate = perlin_octaves(s, 4.0f * size_scale, 7, true);
  float worley_dilate = worley_octaves(s, 6.0f * size_scale, 3, 0.0f, 0.3f);
  float worley_large = worley_octaves(s, 6.0f * size_scale, 3, 0.0f, 0.3f);
  float worley_medium = worley_octaves(s, 12.0f * size_scale, 3, 0.0f, 0.3f);
  float worley_small = worley_octaves(s, 24.0f * size_scale, 3, 0.0f, 0.3f);
  perlin_dilate = c_remap01(perlin_dilate, 0.3f, 1.4f);
  worley_dilate = c_remap01(worley_dilate, -0.3f, 1.3f);
  worley_large = c_remap01(worley_large, -0.4f, 1.0f);
  worley_medium = c_remap01(worley_medium, -0.4f, 1.0f);
  worley_small = c_remap01(worley_small, -0.4f, 1.0f);
  const float perlin_worley = dilate_perlin_worley(perlin_dilate, worley_dilate, 0.3f);
  return cloud_pack(o_saturate(perlin_worley) * 255.0f, o_saturate(worley_large) * 255.0f, o_saturate(worley_medium) * 255.0f, o_saturate(worley_small) * 255.0f);
}
static uint32_t cloud_detail_texel(uint32_t x, uint32_t y, uint32_t z, uint32_t dim) {
  const float sc = 1.0f / dim;
  const vec3 s = v3(x * sc, y * sc, z * sc);
  const float size_scale = 0.5f;
  float worley_large = worley_octaves(s, 10.0f * size_scale, 3, 0.0f, 0.3f);
  float worley_medium = worley_octaves(s, 15.0f * size_scale, 3, 0.0f, 0.3f);
  float worley_small = worley_octaves(s, 20.0f * size_scale, 3, 0.0f, 0.3f);
  worley_large = c_remap01(worley_large, -1.0f, 1.0f);
  worley_medium = c_remap01(worley_medium, -1.0f, 1.0f);
  worley_small = c_remap01(worley_small, -1.0f, 1.0f);
  return cloud_pack(o_saturate(worley_large) * 255.0f, o_saturate(worley_medium) * 255.0f, o_saturate(worley_small) * 255.0f, 255.0f);
}
static uint32_t cloud_weather_texel(uint32_t x, uint32_t y, uint32_t dim, float seed) {
  const float sc = 1.0f / dim;
  const float sx = x * sc, sy = y * sc;
  const float size_scale = 3.0f, coverage_perlin_worley_diff = 0.4f, remap_low = 0.5f, remap_high = 1.3f;
  float perlin1 = perlin_octaves(v3(sx, sy, 0.0f), 2.0f * size_scale, 7, true);
  float worley1 = worley_octaves(v3(sx, sy, 0.0f), 3.0f * size_scale, 2, seed, 0.25f);
  float perlin2 = perlin_octaves(v3(sx, sy, 500.0f), 4.0f * size_scale, 7, true);
  float perlin3 = perlin_octaves(v3(sx, sy, 100.0f), 2.0f * size_scale, 7, true);
  float perlin4 = perlin_octaves(v3(sx, sy, 200.0f), 3.0f * size_scale, 7, true);
  perlin1 = c_remap01(perlin1, remap_low, remap_high);
  worley1 = c_remap01(worley1, remap_low, remap_high);
  perlin2 = c_remap01(perlin2, remap_low, remap_high);
  perlin3 = c_remap01(perlin3, remap_low, remap_high);
  perlin4 = c_remap01(perlin4, remap_low, remap_high);
  perlin1 = o_pow(perlin1, 1.0f);
  worley1 = o_pow(worley1, 0.75f);
  perlin2 = o_pow(perlin2, 2.0f);
  perlin3 = o_pow(perlin3, 3.0f);
  perlin4 = o_pow(perlin4, 1.0f);
  perlin1 = o_saturate(perlin1 * 1.2f) * 0.4f + 0.1f;
  worley1 = o_saturate(1.0f - worley1 * 2.0f);
  perlin2 = o_saturate(perlin2) * 0.5f;
  perlin3 = o_saturate(1.0f - perlin3 * 3.0f);
  perlin4 = o_saturate(1.0f - perlin4 * 1.5f);
  perlin4 = dilate_perlin_worley(worley1, perlin4, coverage_perlin_worley_diff);
  perlin1 -= perlin4;
  perlin2 -= perlin4 * perlin4;
  perlin1 = c_remap01(2.0f * perlin1, 0.05f, 1.0f);
  return cloud_pack(o_saturate(perlin1) * 255.0f, o_saturate(perlin2) * 255.0f, o_saturate(perlin3) * 255.0f, o_saturate(perlin4) * 255.0f);
}

/* ---- texture lookups ---- */
static inline float4_t cloud_tex3d(const uint32_t* tex, int n, float u, float v, float w) {
  const float xb = (u - floorf(u)) * (float) n - 0.5f, yb = (v - floorf(v)) * (float) n - 0.5f, zb = (w - floorf(w)) * (float) n - 0.5f;
  const float xf = floorf(xb), yf = floorf(yb), zf = floorf(zb);
  const float ax = xb - xf, ay = yb - yf, az = zb - zf;
  int x0 = (int) xf, y0 = (int) yf, z0 = (int) zf, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
  if (x0 < 0) x0 += n;
  if (y0 < 0) y0 += n;
  if (z0 < 0) z0 += n;
  if (x0 >= n) x0 -= n; /* (u - floor(u)) can round to 1 */
  if (y0 >= n) y0 -= n;
  if (z0 >= n) z0 -= n;
  if (x1 >= n) x1 -= n;
  if (y1 >= n) y1 -= n;
  if (z1 >= n) z1 -= n;
  float4_t c[8];
  for (int k = 0; k < 8; k++) c[k] = texel_unpack(tex[((k & 1) ? x1 : x0) + n * (((k & 2) ? y1 : y0) + n * ((k & 4) ? z1 : z0))]);
  float r[4];
  for (int ch = 0; ch < 4; ch++) {
    float q[8];
    for (int k = 0; k < 8; k++) q[k] = (ch == 0) ? c[k].x : (ch == 1) ? c[k].y : (ch == 2) ? c[k].z : c[k].w;
    const float a0 = q[0] + ax * (q[1] - q[0]), a1 = q[2] + ax * (q[3] - q[2]), a2 = q[4] + ax * (q[5] - q[4]), a3 = q[6] + ax * (q[7] - q[6]);
    const float b0 = a0 + ay * (a1 - a0), b1 = a2 + ay * (a3 - a2);
    r[ch] = b0 + az * (b1 - b0);
  }
  return f4(r[0], r[1], r[2], r[3]);
}
static inline float4_t cloud_tex2d(const uint32_t* tex, int n, float u, float v) { /* texture_load without flip and gamma (cloud_utils.cuh:77-79) */
  const float xb = (u - floorf(u)) * (float) n - 0.5f, yb = (v - floorf(v)) * (float) n - 0.5f;
  const float xf = floorf(xb), yf = floorf(yb);
  const float ax = xb - xf, ay = yb - yf;
  int x0 = (int) xf, y0 = (int) yf, x1 = x0 + 1, y1 = y0 + 1;
  if (x0 < 0) x0 += n;
  if (y0 < 0) y0 += n;
  if (x0 >= n) x0 -= n;
  if (y0 >= n) y0 -= n;
  if (x1 >= n) x1 -= n;
  if (y1 >= n) y1 -= n;
  const float4_t c00 = texel_unpack(tex[x0 + y0 * n]), c10 = texel_unpack(tex[x1 + y0 * n]), c01 = texel_unpack(tex[x0 + y1 * n]), c11 = texel_unpack(tex[x1 + y1 * n]);
  float4_t r;
  { const float top = c00.x + ax * (c10.x - c00.x), bot = c01.x + ax * (c11.x - c01.x); r.x = top + ay * (bot - top); }
  { const float top = c00.y + ax * (c10.y - c00.y), bot = c01.y + ax * (c11.y - c01.y); r.y = top + ay * (bot - top); }
  { const float top = c00.z + ax * (c10.z - c00.z), bot = c01.z + ax * (c11.z - c01.z); r.z = top + ay * (bot - top); }
  { const float top = c00.w + ax * (c10.w - c00.w), bot = c01.w + ax * (c11.w - c01.w); r.w = top + ay * (bot - top); }
  return r;
}

/* ---- layers, weather, density (cloud_utils.cuh) ---- */
enum { CLOUD_LAYER_LOW = 0, CLOUD_LAYER_MID = 1, CLOUD_LAYER_TOP = 2 };
enum { CL_ACTIVE = 0, CL_HEIGHT_MAX, CL_HEIGHT_MIN, CL_COVERAGE, CL_COVERAGE_MIN, CL_TYPE, CL_TYPE_MIN, CL_WIND_SPEED, CL_WIND_COS, CL_WIND_SIN };
typedef struct { float coverage, type, coverage1, coverage2; } CloudWeather;
static const float CLOUD_GRADIENT_STRATUS[4] = {0.01f, 0.15f, 0.17f, 0.3f}, CLOUD_GRADIENT_STRATOCUMULUS[4] = {0.01f, 0.12f, 0.45f, 0.6f};
static const float CLOUD_GRADIENT_CUMULUS[4] = {0.01f, 0.06f, 0.8f, 0.99f}, CLOUD_GRADIENT_ALTOSTRATUS[4] = {0.01f, 0.5f, 0.5f, 0.95f};
static const float CLOUD_GRADIENT_ALTOCUMULUS[4] = {0.25f, 0.30f, 0.60f, 0.75f}, CLOUD_GRADIENT_TOPLAYER[4] = {0.01f, 0.20f, 0.80f, 0.95f};
static inline float cloud_gradient(const float g[4], float height) { return c_smoothstep(height, g[0], g[1]) - c_smoothstep(height, g[2], g[3]); }
static inline float cloud_height(const OracleScene* s, vec3 pos, int layer) {
  const float* L = s->cloud_layers[layer];
  return (sky_height(pos) - L[CL_HEIGHT_MIN]) / (L[CL_HEIGHT_MAX] - L[CL_HEIGHT_MIN]);
}
static inline CloudWeather cloud_weather(const OracleScene* s, vec3 pos, float height, int layer) {
  const float* L = s->cloud_layers[layer];
  pos.x += s->cloud_offset_x;
  pos.z += s->cloud_offset_z;
  vec3 wp = pos;
  wp.x = wp.x + L[CL_WIND_SPEED] * height * L[CL_WIND_COS];
  wp.z = wp.z + L[CL_WIND_SPEED] * height * L[CL_WIND_SIN];
  const float k = (layer == CLOUD_LAYER_LOW) ? 0.012f : (layer == CLOUD_LAYER_MID) ? 0.01f : 0.004f;
  wp = v_scale(wp, k * s->cloud_noise_weather_scale);
  const float4_t tex = cloud_tex2d((const uint32_t*) s->cloud_noise_weather, CLOUD_WEATHER_RES, wp.x, wp.z);
  CloudWeather w = {0.0f, 0.0f, 0.0f, 0.0f};
  if (layer == CLOUD_LAYER_LOW) {
    w.coverage = o_saturate(c_remap(tex.x * L[CL_COVERAGE], 0.0f, 1.0f, L[CL_COVERAGE_MIN], 1.0f));
    w.type = o_saturate(c_remap(tex.y * L[CL_TYPE], 0.0f, 1.0f, L[CL_TYPE_MIN], 1.0f));
  }
  else if (layer == CLOUD_LAYER_MID) {
    w.coverage = o_saturate(c_remap(tex.z * L[CL_COVERAGE], 0.0f, 1.0f, L[CL_COVERAGE_MIN], 1.0f));
    w.type = o_saturate(c_remap(tex.w * L[CL_TYPE], 0.0f, 1.0f, L[CL_TYPE_MIN], 1.0f));
  }
  else {
    w.coverage = o_saturate(c_remap(tex.x * L[CL_COVERAGE], 0.0f, 1.0f, L[CL_COVERAGE_MIN], 1.0f));
    w.coverage1 = o_saturate(c_remap(tex.y * L[CL_COVERAGE], 0.0f, 1.0f, L[CL_COVERAGE_MIN], 1.0f));
    w.coverage2 = o_saturate(c_remap(tex.z * L[CL_COVERAGE], 0.0f, 1.0f, L[CL_COVERAGE_MIN], 1.0f));
  }
  return w;
}
static inline void cloud_gradient_type(int layer, const CloudWeather* w, float out[4]) {
  if (layer == CLOUD_LAYER_LOW) {
    const float stratus = 1.0f - o_saturate(w->type * 2.0f), stratocumulus = 1.0f - fabsf(2.0f * w->type - 1.0f), cumulus = o_saturate(2.0f * w->type - 1.0f);
    for (int k = 0; k < 4; k++) out[k] = stratus * CLOUD_GRADIENT_STRATUS[k] + stratocumulus * CLOUD_GRADIENT_STRATOCUMULUS[k] + cumulus * CLOUD_GRADIENT_CUMULUS[k];
  }
  else if (layer == CLOUD_LAYER_MID) {
    const float altostratus = 1.0f - o_saturate(w->type), altocumulus = o_saturate(w->type);
    for (int k = 0; k < 4; k++) out[k] = altostratus * CLOUD_GRADIENT_ALTOSTRATUS[k] + altocumulus * CLOUD_GRADIENT_ALTOCUMULUS[k];
  }
  else for (int k = 0; k < 4; k++) out[k] = CLOUD_GRADIENT_TOPLAYER[k];
}
static inline bool cloud_significant_point(float height, const CloudWeather* w, int layer) {
  float type[4];
  cloud_gradient_type(layer, w, type);
  const bool covered = (layer == CLOUD_LAYER_TOP) ? (w->coverage > CLOUD_WEATHER_CUTOFF || w->coverage1 > CLOUD_WEATHER_CUTOFF || w->coverage2 > CLOUD_WEATHER_CUTOFF)
                                                  : (w->coverage > CLOUD_WEATHER_CUTOFF);
  return covered && (type[0] < height) && (type[3] > height);
}
static inline void cloud_layer_intersection(const OracleScene* s, vec3 origin, vec3 ray, float limit, int layer, float* start_out, float* dist_out) { /* :214-277 */
  const float* L = s->cloud_layers[layer];
  if (L[CL_ACTIVE] == 0.0f) { *start_out = FLT_MAX; *dist_out = 0.0f; return; }
  const float hmin = L[CL_HEIGHT_MIN] + SKY_EARTH_RADIUS, hmax = L[CL_HEIGHT_MAX] + SKY_EARTH_RADIUS;
  const float height = v_len(origin);
  const float dist_hmax = sph_int_p0(ray, origin, hmax), dist_hmin = sph_int_p0(ray, origin, hmin);
  float start;
  if (height > hmax) start = dist_hmax;
  else if (height < hmin) start = dist_hmin;
  else start = 0.0f;
  const float end_1 = (height < hmin) ? dist_hmax : dist_hmin;
  const float end_2 = (height > hmax) ? sph_int_back_p0(ray, origin, hmax) : dist_hmax;
  const float end_dist = fminf(end_1, end_2);
  const float earth_hit = sph_int_p0(ray, origin, SKY_EARTH_RADIUS);
  const float distance = fminf(earth_hit, fminf(limit, end_dist)) - start;
  if (distance < 0.0f) start = FLT_MAX;
  *start_out = start; *dist_out = distance;
}
static inline float cloud_density(const OracleScene* s, vec3 pos, float height, const CloudWeather* w, int layer) { /* :283-403 */
  const float* L = s->cloud_layers[layer];
  pos.x += s->cloud_offset_x;
  pos.z += s->cloud_offset_z;
  float density;
  float gradient_type[4];
  cloud_gradient_type(layer, w, gradient_type);
  const float density_gradient = cloud_gradient(gradient_type, height);
  if (layer == CLOUD_LAYER_LOW) {
    vec3 sp = pos;
    sp.x = sp.x + L[CL_WIND_SPEED] * height * L[CL_WIND_COS] * 0.33f;
    sp.z = sp.z + L[CL_WIND_SPEED] * height * L[CL_WIND_SIN] * 0.33f;
    sp = v_scale(sp, 0.4f * s->cloud_noise_shape_scale);
    const float4_t shape = cloud_tex3d((const uint32_t*) s->cloud_noise_shape, CLOUD_SHAPE_RES, sp.x, sp.y, sp.z);
    float shape_sum = shape.x * 5.0f;
    shape_sum += shape.y * cloud_gradient(CLOUD_GRADIENT_STRATUS, height);
    shape_sum += shape.z * cloud_gradient(CLOUD_GRADIENT_STRATOCUMULUS, height);
    shape_sum += shape.w * cloud_gradient(CLOUD_GRADIENT_CUMULUS, height);
    shape_sum *= 0.16f;
    density = fabsf(shape_sum * density_gradient);
    density = o_pow(density, o_saturate(height * 6.0f));
    density = c_smoothstep(density, 0.25f, 1.1f);
    density = o_saturate(density - (1.0f - w->coverage)) * w->coverage;
  }
  else {
    const vec3 sp = v_scale(pos, 0.2f * s->cloud_noise_shape_scale);
    const float4_t shape = cloud_tex3d((const uint32_t*) s->cloud_noise_shape, CLOUD_SHAPE_RES, sp.x, sp.y, sp.z);
    if (layer == CLOUD_LAYER_MID) {
      const float d0 = (shape.x * 0.5f + shape.y * 0.25f + shape.w * 0.125f + shape.z * 0.125f) * sqrtf(w->coverage);
      const float d1 = c_smoothstep(shape.x * 0.1f + shape.y * 0.7f + shape.w * 0.1f + shape.z * 0.1f, 0.50f, 1.0f) * w->coverage;
      const float interp = c_smoothstep(w->type, 0.1f, 0.5f);
      density = c_remap01(density_gradient * (d0 * (1.0f - interp) + d1 * interp), 0.05f, 1.0f);
    }
    else {
      const float d1 = (shape.x * 0.3f + shape.y * 0.3f + shape.z * 0.2f + shape.w * 0.2f) * sqrtf(w->coverage1);
      const float d2 = c_smoothstep(shape.x * 0.2f + shape.y * 0.4f + shape.w * 0.2f + shape.z * 0.2f, 0.50f, 1.0f) * w->coverage2;
      density = c_remap01(density_gradient * (d1 + d2) * 0.25f, 0.05f, 1.0f);
    }
  }
  if (layer != CLOUD_LAYER_TOP && density > 0.0f) { /* cloud_erode_density */
    const vec3 dp = v_scale(pos, 2.0f * s->cloud_noise_detail_scale);
    const float4_t detail = cloud_tex3d((const uint32_t*) s->cloud_noise_detail, CLOUD_DETAIL_RES, dp.x, dp.y, dp.z);
    const float detail_fbm = o_saturate(detail.x * 0.625f + detail.y * 0.25f + detail.z * 0.125f);
    const float noise_modifier = o_lerp(1.0f - detail_fbm, detail_fbm, o_saturate(height * 10.0f));
    density = c_remap(density, noise_modifier * 0.2f, 1.0f, 0.0f, 1.0f);
  }
  return fmaxf(density * s->cloud_density, 0.0f);
}

/* ---- the shadow the layers cast into the sky march (cloud_shadow.cuh) ---- */
static inline bool cloud_shadow_layer(const OracleScene* s, vec3 origin, vec3 ray, int step_count, int layer) {
  const float* L = s->cloud_layers[layer];
  float start, d;
  cloud_layer_intersection(s, origin, ray, FLT_MAX, layer, &start, &d);
  const float max_dist = 6.0f * (L[CL_HEIGHT_MAX] - L[CL_HEIGHT_MIN]);
  const float dist = fminf(d, max_dist);
  if (start != FLT_MAX && dist > 0.0f) {
    const float step_size = dist / step_count;
    float reach = start + 0.1f * step_size;
    for (int i = 0; i < step_count; i++) {
      const vec3 pos = v_add(origin, v_scale(ray, reach));
      const float height = cloud_height(s, pos, layer);
      if (height < 0.0f || height > 1.0f) break;
      const CloudWeather w = cloud_weather(s, pos, height, layer);
      if (cloud_significant_point(height, &w, layer)) {
        if (cloud_density(s, pos, height, &w, layer) > 0.0f) return true;
      }
      reach += step_size;
    }
  }
  return false;
}
static float cloud_shadow(const OracleScene* s, vec3 origin, vec3 ray) {
  if (!s->cloud_active || !s->cloud_atmosphere_scattering || !s->cloud_noise_shape) return 1.0f;
  if (s->cloud_layers[0][CL_ACTIVE] != 0.0f && cloud_shadow_layer(s, origin, ray, (int) s->cloud_steps / 3, CLOUD_LAYER_LOW)) return 0.0f;
  if (s->cloud_layers[1][CL_ACTIVE] != 0.0f && cloud_shadow_layer(s, origin, ray, (int) s->cloud_steps / 16, CLOUD_LAYER_MID)) return 0.1f;
  if (s->cloud_layers[2][CL_ACTIVE] != 0.0f && cloud_shadow_layer(s, origin, ray, (int) s->cloud_steps / 32, CLOUD_LAYER_TOP)) return 0.5f;
  return 1.0f;
}

/* ---- the march (cloud.cuh) ---- */
static inline float cloud_extinction(const OracleScene* s, vec3 origin, vec3 ray, int layer) { /* :49-81 */
  const float iter_step = 1.0f / (float) (int) s->cloud_shadow_steps;
  float optical_depth = 0.0f;
  for (float i = 0.0f; i < 1.0f; i += iter_step) {
    float t0 = i, t1 = i + iter_step;
    t0 = t0 * t0;
    t1 = t1 * t1;
    const float step_size = t1 - t0;
    const float reach = t0 + step_size * 0.5f;
    const vec3 pos = v_add(origin, v_scale(ray, reach));
    const float height = cloud_height(s, pos, layer);
    if (height > 1.0f || height < 0.0f) break;
    const CloudWeather w = cloud_weather(s, pos, height, layer);
    if (cloud_significant_point(height, &w, layer)) optical_depth -= cloud_density(s, pos, height, &w, layer) * step_size;
  }
  optical_depth *= CLOUD_EXTINCTION_DENSITY;
  return o_exp(optical_depth);
}
typedef struct { RGBF scattered_light; float transmittance, hit_dist; } CloudResult;
static inline float je_phase_function_ms(const float p[4], float c, float ms_factor) { /* math.cuh:1234-1239 with the octave's factor on both asymmetries */
  return (1.0f - p[3]) * hg_phase(c, p[0] * ms_factor) + p[3] * draine_phase(c, p[1] * ms_factor, p[2]);
}
static CloudResult clouds_compute(const OracleScene* s, const OSky* sky, const Sampler* smp, vec3 origin, vec3 ray, float start, float dist, int layer) { /* :86-262 */
  CloudResult result;
  result.scattered_light = c_splat(0.0f); result.transmittance = 1.0f; result.hit_dist = start;
  if (dist < 0.0f || start == FLT_MAX) return result;
  const float* L = s->cloud_layers[layer];
  const float span = L[CL_HEIGHT_MAX] - L[CL_HEIGHT_MIN];
  dist = fminf(6.0f * span, dist);
  const int base_steps = (layer == CLOUD_LAYER_LOW) ? (int) s->cloud_steps : (layer == CLOUD_LAYER_MID) ? (int) s->cloud_steps / 4 : (int) s->cloud_steps / 8;
  int step_count = (int) ((float) base_steps * o_saturate(dist / (6.0f * span)));
  step_count = (int) ((float) step_count + 8.0f * rnd1(smp, RT_CLOUD_STEP_COUNT + layer));
  start = fmaxf(0.0f, start);
  const float step_size = dist / (float) step_count;
  const float random_offset = rnd1(smp, RT_CLOUD_STEP_OFFSET + layer);
  float reach = start + (0.1f + random_offset * 0.9f) * step_size;
  const float sun_solid_angle = sphere_solid_angle(sky->sun_pos, SKY_SUN_RADIUS, v_add(origin, v_scale(ray, reach)));
  float transmittance = 1.0f;
  RGBF scattered_light = c_splat(0.0f);
  float hit_dist = start;
  bool hit = false;
  const float2_t ambient_r = rnd2(smp, RT_CLOUD_DIR);
  const vec3 ambient_ray = sample_ray_sphere(2.0f * ambient_r.x - 1.0f, ambient_r.y);
  const float ambient_cos_angle = v_dot(ray, ambient_ray);
  for (int i = 0; i < step_count; i++) {
    const vec3 pos = v_add(origin, v_scale(ray, reach));
    if (!hit) hit_dist = reach;
    const float height = cloud_height(s, pos, layer);
    if (height < 0.0f || height > 1.0f) break;
    const CloudWeather w = cloud_weather(s, pos, height, layer);
    if (!cloud_significant_point(height, &w, layer)) { reach += step_size; continue; }
    const float density = cloud_density(s, pos, height, &w, layer);
    if (density > 0.0f) {
      hit = true;
      const RGBF ambient_color = sky_get_color(sky, pos, ambient_ray, FLT_MAX, false, (int) (sky->steps / 2u), rnd1(smp, RANDOM_TARGET_SKY_STEP_OFFSET));
      float ambient_extinction = cloud_extinction(s, pos, ambient_ray, layer);
      RGBF sun_color;
      float sun_extinction, sun_cos_angle;
      const vec3 sun_ray = v_norm(v_sub(sky->sun_pos, pos));
      if (!sph_hit_p0(sun_ray, pos, SKY_EARTH_RADIUS)) {
        sun_color = sky_sun_color_ex(sky, pos, sun_ray, false);
        sun_cos_angle = v_dot(ray, sun_ray);
        sun_extinction = cloud_extinction(s, pos, sun_ray, layer);
      }
      else { sun_color = c_splat(0.0f); sun_extinction = 1.0f; sun_cos_angle = 0.0f; }
      float scattering = density * CLOUD_SCATTERING_DENSITY;
      float extinction = fmaxf(density * CLOUD_EXTINCTION_DENSITY, 0.0001f);
      float phase_factor = 1.0f;
      for (uint32_t o = 0; o < s->cloud_octaves; o++) {
        scattering *= 0.5f;
        extinction *= 0.5f;
        const float sun_phase = je_phase_function_ms(s->cloud_phase, sun_cos_angle, phase_factor);
        const float ambient_phase = je_phase_function_ms(s->cloud_phase, ambient_cos_angle, phase_factor);
        phase_factor *= 0.5f;
        const RGBF sun_color_i = c_scale(sun_color, sun_extinction * sun_phase * sun_solid_angle);
        const RGBF ambient_color_i = c_scale(ambient_color, ambient_extinction * ambient_phase * 4.0f * REF_PI);
        sun_extinction = sqrtf(sun_extinction);
        ambient_extinction = sqrtf(ambient_extinction);
        RGBF S = c_add(sun_color_i, ambient_color_i);
        S = c_scale(S, scattering);
        const float step_trans = o_exp(-extinction * step_size);
        S = c_scale(c_sub(S, c_scale(S, step_trans)), 1.0f / extinction);
        scattered_light = c_add(scattered_light, c_scale(S, transmittance));
      }
      transmittance *= o_exp(-density * CLOUD_EXTINCTION_DENSITY * step_size);
      if (transmittance < 0.1f) { transmittance = 0.0f; break; }
    }
    reach += step_size;
  }
  result.scattered_light = scattered_light; result.transmittance = transmittance; result.hit_dist = hit_dist;
  return result;
}
/* clouds_render (:268-334): the layers in the order a ray enters them; with atmosphere_scattering the air between them is marched as well */
static float clouds_render(const OracleScene* s, const OSky* sky, const Sampler* smp, vec3 origin, vec3 ray, float limit, RGBF* color, RGBF* transmittance,
                           float* transmittance_cloud_only) {
  float starts[3], dists[3];
  CloudResult results[3];
  for (int l = 0; l < 3; l++) {
    cloud_layer_intersection(s, origin, ray, limit, l, &starts[l], &dists[l]);
    results[l] = clouds_compute(s, sky, smp, origin, ray, starts[l], dists[l], l);
  }
  const bool less01 = starts[0] <= starts[1], less02 = starts[0] <= starts[2], less12 = starts[1] <= starts[2];
  int order[3];
  if (less01) {
    if (less02) { order[0] = 0; order[1] = less12 ? 1 : 2; order[2] = less12 ? 2 : 1; }
    else { order[0] = 2; order[1] = less01 ? 0 : 1; order[2] = less01 ? 1 : 0; }
  }
  else if (less12) { order[0] = 1; order[1] = less02 ? 0 : 2; order[2] = less02 ? 2 : 0; }
  else { order[0] = 2; order[1] = less01 ? 0 : 1; order[2] = less01 ? 1 : 0; }
  float prev_start = 0.0f;
  for (int i = 0; i < 3; i++) {
    const CloudResult r = results[order[i]];
    if (r.hit_dist == FLT_MAX) break;
    if (s->cloud_atmosphere_scattering) {
      *color = c_add(*color, sky_trace_inscattering(sky, origin, ray, r.hit_dist - prev_start, transmittance, smp->depth == 0, rnd1(smp, RANDOM_TARGET_SKY_INSCATTERING_STEP),
                                                    rnd1(smp, RANDOM_TARGET_SKY_STEP_OFFSET)));
      origin = v_add(origin, v_scale(ray, r.hit_dist - prev_start));
    }
    *color = c_add(*color, c_mul(r.scattered_light, *transmittance));
    *transmittance = c_scale(*transmittance, r.transmittance);
    *transmittance_cloud_only *= r.transmittance;
    prev_start = r.hit_dist;
  }
  return prev_start;
}

#endif
