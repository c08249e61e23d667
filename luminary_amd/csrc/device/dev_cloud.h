// Clouds, first part: the noise textures, the layers' density function and the shadow they cast into the sky's in-scattering march (included by
// dev_sky.h ahead of sky_compute_atmosphere; the march itself is dev_cloud_march.h).
// Reference: cuda/cloud_noise.cuh (tiling Perlin and inverted Worley octaves; texture sizes device_cloud.c:9-11), cuda/cloud_utils.cuh (layers, weather
// map, density), cuda/cloud_shadow.cuh. The textures are RGBA8 with wrap addressing and a linear filter and have no mip levels (texture.c:85), so the LOD
// bias of the reference's lookups has no effect. Filter weights are exact floats, trilinear in the order x, y, z (as the oracle defines it).
#pragma once

LUM_NS_BEGIN

constexpr int kCloudShapeRes = 128, kCloudDetailRes = 32, kCloudWeatherRes = 1024;
constexpr uint32_t kRndCloudStepOffset = 67, kRndCloudStepCount = 71, kRndCloudDir = 75;
constexpr float kCloudScatteringDensity = 1000.0f * 0.1f * 0.9f, kCloudExtinctionDensity = 1000.0f * 0.1f, kCloudWeatherCutoff = 0.05f;
enum CloudLayer : int { kCloudLow = 0, kCloudMid = 1, kCloudTop = 2 };
enum CloudLayerField : int { kClActive = 0, kClHeightMax, kClHeightMin, kClCoverage, kClCoverageMin, kClType, kClTypeMin, kClWindSpeed, kClWindCos, kClWindSin };

// math.cuh:33-72
LUM_DEV float c_fract(float x) { return x - floorf(x); }
LUM_DEV float c_remap(float value, float src_low, float src_high, float dst_low, float dst_high) { return (value - src_low) / (src_high - src_low) * (dst_high - dst_low) + dst_low; }
LUM_DEV float c_remap01(float value, float src_low, float src_high) { return saturate(c_remap(value, src_low, src_high, 0.0f, 1.0f)); }
LUM_DEV float c_step(float edge, float x) { return (x < edge) ? 0.0f : 1.0f; }
LUM_DEV float c_smoothstep(float x, float edge0, float edge1) { const float t = c_remap01(x, edge0, edge1); return t * t * (3.0f - 2.0f * t); }
LUM_DEV float c_sin(float x) { float s, c; sincos_det(x, s, c); return s; }

// ---- noise (cloud_noise.cuh) ----
LUM_DEV float interp_cubic_d2(float x) { return x * x * x * (x * (x * 6.0f - 15.0f) + 10.0f); }
LUM_DEV void perlin_hash(V3 grid, float scale, bool tile, float* low0, float* low1, float* low2, float* high0, float* high1, float* high2) {
  const float offset_x = 50.0f, offset_y = 161.0f, domain = 69.0f;
  const float largef[3] = {635.298681f, 682.357502f, 668.926525f}, z_inc[3] = {48.500388f, 65.294118f, 63.934599f};
  grid.x -= floorf(grid.x / domain) * domain;
  grid.y -= floorf(grid.y / domain) * domain;
  grid.z -= floorf(grid.z / domain) * domain;
  const float d = domain - 1.5f;
  float inc_x = c_step(grid.x, d) * (grid.x + 1.0f), inc_y = c_step(grid.y, d) * (grid.y + 1.0f), inc_z = c_step(grid.z, d) * (grid.z + 1.0f);
  if (tile) { inc_x = fmodf(inc_x, scale); inc_y = fmodf(inc_y, scale); inc_z = fmodf(inc_z, scale); }
  float p[4] = {grid.x + offset_x, grid.y + offset_y, inc_x + offset_x, inc_y + offset_y};
#pragma unroll
  for (int k = 0; k < 4; k++) p[k] *= p[k];
  const float q[4] = {p[0] * p[1], p[2] * p[1], p[0] * p[3], p[2] * p[3]};
  float low[3], high[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { low[k] = 1.0f / (largef[k] + grid.z * z_inc[k]); high[k] = 1.0f / (largef[k] + inc_z * z_inc[k]); }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    low0[k] = c_fract(q[k] * low[0]); low1[k] = c_fract(q[k] * low[1]); low2[k] = c_fract(q[k] * low[2]);
    high0[k] = c_fract(q[k] * high[0]); high1[k] = c_fract(q[k] * high[1]); high2[k] = c_fract(q[k] * high[2]);
  }
}
LUM_DEV float perlin(V3 p, float scale, bool tile) {
  p = p * scale;
  const V3 p1 = v3(floorf(p.x), floorf(p.y), floorf(p.z));
  const V3 pf = p - p1;
  const V3 pm = v3(pf.x + -1.0f, pf.y + -1.0f, pf.z + -1.0f);
  float hx0[4], hy0[4], hz0[4], hx1[4], hy1[4], hz1[4];
  perlin_hash(p1, scale, tile, hx0, hy0, hz0, hx1, hy1, hz1);
  const float fx[4] = {pf.x, pm.x, pf.x, pm.x}, fy[4] = {pf.y, pf.y, pm.y, pm.y};
  float res[4];
  const float bx = interp_cubic_d2(pf.x), by = interp_cubic_d2(pf.y), bz = interp_cubic_d2(pf.z);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float gx0 = hx0[k] - 0.49999f, gy0 = hy0[k] - 0.49999f, gz0 = hz0[k] - 0.49999f;
    const float gx1 = hx1[k] - 0.49999f, gy1 = hy1[k] - 0.49999f, gz1 = hz1[k] - 0.49999f;
    const float grad0 = rsqrt_ieee(gx0 * gx0 + gy0 * gy0 + gz0 * gz0) * (fx[k] * gx0 + fy[k] * gy0 + pf.z * gz0);
    const float grad1 = rsqrt_ieee(gx1 * gx1 + gy1 * gy1 + gz1 * gz1) * (fx[k] * gx1 + fy[k] * gy1 + pm.z * gz1);
    res[k] = lerpf(grad0, grad1, bz);
  }
  const float b2z = 1.0f - bx, b2w = 1.0f - by;
  float final = res[0] * b2z * b2w + res[1] * bx * b2w + res[2] * b2z * by + res[3] * bx * by;
  final /= sqrtf(0.75f);
  return ((final * 1.5f) + 1.0f) * 0.5f;
}
LUM_DEV float perlin_octaves(V3 p, float scale, int octaves, bool tile) {
  float frequency = 1.0f, persistence = 1.0f, value = 0.0f;
  for (int i = 0; i < octaves; i++) {
    value += persistence * perlin(p, scale * frequency, tile);
    persistence *= 0.5f;
    frequency *= 2.0f;
  }
  return value;
}
LUM_DEV V3 voronoi_hash(V3 x, float scale) {
  x.x = fmodf(x.x, scale); x.y = fmodf(x.y, scale); x.z = fmodf(x.z, scale);
  x = v3(dot(x, v3(127.1f, 311.7f, 74.7f)), dot(x, v3(269.5f, 183.3f, 246.1f)), dot(x, v3(113.5f, 271.9f, 124.6f)));
  const float h = 43758.5453123f;
  return v3(c_fract(c_sin(x.x) * h), c_fract(c_sin(x.y) * h), c_fract(c_sin(x.z) * h));
}
LUM_DEV float voronoi_x(V3 x, float scale, float seed, bool inverted) {  // the callers use the nearest distance only
  x = x * scale;
  x = v3(x.x + 0.5f, x.y + 0.5f, x.z + 0.5f);
  const V3 p = v3(floorf(x.x), floorf(x.y), floorf(x.z));
  const V3 f = v3(c_fract(x.x), c_fract(x.y), c_fract(x.z));
  float res_x = 1.0f;
  for (int k = -1; k <= 1; k++)
    for (int j = -1; j <= 1; j++)
      for (int i = -1; i <= 1; i++) {
        const V3 b = v3((float) i, (float) j, (float) k);
        const V3 pb = p + b;
        const V3 r = (b - f) + voronoi_hash(v3(pb.x + seed * 10.0f, pb.y + seed * 10.0f, pb.z + seed * 10.0f), scale);
        const float d = dot(r, r);
        if (d < res_x) res_x = d;
      }
  return inverted ? 1.0f - res_x : res_x;
}
LUM_DEV float worley_octaves(V3 p, float scale, int octaves, float seed, float persistence) {
  float value = saturate(voronoi_x(p, scale, seed, true));
  float frequency = 2.0f;
  for (int i = 1; i < octaves; i++) {
    value -= persistence * saturate(voronoi_x(p, scale * frequency, seed, false));
    frequency *= 2.0f;
  }
  return value;
}
LUM_DEV float dilate_perlin_worley(float p, float w, float x) {
  const float curve = 0.75f;
  if (x < 0.5f) {
    x *= 2.0f;
    const float n = p + w * x;
    return n * lerpf(1.0f, 0.5f, pow_det(x, curve));
  }
  x = 2.0f * (x - 0.5f);
  const float n = w + p * (1.0f - x);
  return n * lerpf(0.5f, 1.0f, pow_det(x, 1.0f / curve));
}
LUM_DEV uint32_t cloud_pack(float a, float b, float c, float d) {  // make_uchar4 of float products: the conversion truncates
  return ((uint32_t) a & 0xFFu) | (((uint32_t) b & 0xFFu) << 8) | (((uint32_t) c & 0xFFu) << 16) | (((uint32_t) d & 0xFFu) << 24);
}
__global__ void k_cloud_noise_shape(uint32_t* dst, uint32_t dim) {
  const uint32_t amount = dim * dim * dim;
  const float sc = 1.0f / dim;
  for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < amount; id += gridDim.x * blockDim.x) {
    const uint32_t z = id / (dim * dim), y = (id - z * (dim * dim)) / dim, x = id - y * dim - z * dim * dim;
    const V3 s = v3(x * sc, y * sc, z * sc);
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
    dst[id] = cloud_pack(saturate(perlin_worley) * 255.0f, saturate(worley_large) * 255.0f, saturate(worley_medium) * 255.0f, saturate(worley_small) * 255.0f);
  }
}
__global__ void k_cloud_noise_detail(uint32_t* dst, uint32_t dim) {
  const uint32_t amount = dim * dim * dim;
  const float sc = 1.0f / dim;
  for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < amount; id += gridDim.x * blockDim.x) {
    const uint32_t z = id / (dim * dim), y = (id - z * (dim * dim)) / dim, x = id - y * dim - z * dim * dim;
    const V3 s = v3(x * sc, y * sc, z * sc);
    const float size_scale = 0.5f;
    float worley_large = worley_octaves(s, 10.0f * size_scale, 3, 0.0f, 0.3f);
    float worley_medium = worley_octaves(s, 15.0f * size_scale, 3, 0.0f, 0.3f);
    float worley_small = worley_octaves(s, 20.0f * size_scale, 3, 0.0f, 0.3f);
    worley_large = c_remap01(worley_large, -1.0f, 1.0f);
    worley_medium = c_remap01(worley_medium, -1.0f, 1.0f);
    worley_small = c_remap01(worley_small, -1.0f, 1.0f);
    dst[id] = cloud_pack(saturate(worley_large) * 255.0f, saturate(worley_medium) * 255.0f, saturate(worley_small) * 255.0f, 255.0f);
  }
}
__global__ void k_cloud_noise_weather(uint32_t* dst, uint32_t dim, float seed) {
  const uint32_t amount = dim * dim;
  const float sc = 1.0f / dim;
  for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < amount; id += gridDim.x * blockDim.x) {
    const uint32_t y = id / dim, x = id - y * dim;
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
    perlin1 = pow_det(perlin1, 1.0f);
    worley1 = pow_det(worley1, 0.75f);
    perlin2 = pow_det(perlin2, 2.0f);
    perlin3 = pow_det(perlin3, 3.0f);
    perlin4 = pow_det(perlin4, 1.0f);
    perlin1 = saturate(perlin1 * 1.2f) * 0.4f + 0.1f;
    worley1 = saturate(1.0f - worley1 * 2.0f);
    perlin2 = saturate(perlin2) * 0.5f;
    perlin3 = saturate(1.0f - perlin3 * 3.0f);
    perlin4 = saturate(1.0f - perlin4 * 1.5f);
    perlin4 = dilate_perlin_worley(worley1, perlin4, coverage_perlin_worley_diff);
    perlin1 -= perlin4;
    perlin2 -= perlin4 * perlin4;
    perlin1 = c_remap01(2.0f * perlin1, 0.05f, 1.0f);
    dst[id] = cloud_pack(saturate(perlin1) * 255.0f, saturate(perlin2) * 255.0f, saturate(perlin3) * 255.0f, saturate(perlin4) * 255.0f);
  }
}

// ---- texture lookups ----
LUM_DEV float4 cloud_texel(uint32_t t) {
  return make_float4((t & 0xFFu) * (1.0f / 255.0f), ((t >> 8) & 0xFFu) * (1.0f / 255.0f), ((t >> 16) & 0xFFu) * (1.0f / 255.0f), (t >> 24) * (1.0f / 255.0f));
}
LUM_DEV float4 cloud_tex3d(const uint32_t* __restrict__ tex, int n, float u, float v, float w) {
  const float xb = (u - floorf(u)) * (float) n - 0.5f, yb = (v - floorf(v)) * (float) n - 0.5f, zb = (w - floorf(w)) * (float) n - 0.5f;
  const float xf = floorf(xb), yf = floorf(yb), zf = floorf(zb);
  const float ax = xb - xf, ay = yb - yf, az = zb - zf;
  int x0 = (int) xf, y0 = (int) yf, z0 = (int) zf, x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
  if (x0 < 0) x0 += n;
  if (y0 < 0) y0 += n;
  if (z0 < 0) z0 += n;
  if (x0 >= n) x0 -= n;  // (u - floor(u)) can round to 1
  if (y0 >= n) y0 -= n;
  if (z0 >= n) z0 -= n;
  if (x1 >= n) x1 -= n;
  if (y1 >= n) y1 -= n;
  if (z1 >= n) z1 -= n;
  const float4 c0 = cloud_texel(tex[x0 + n * (y0 + n * z0)]), c1 = cloud_texel(tex[x1 + n * (y0 + n * z0)]);
  const float4 c2 = cloud_texel(tex[x0 + n * (y1 + n * z0)]), c3 = cloud_texel(tex[x1 + n * (y1 + n * z0)]);
  const float4 c4 = cloud_texel(tex[x0 + n * (y0 + n * z1)]), c5 = cloud_texel(tex[x1 + n * (y0 + n * z1)]);
  const float4 c6 = cloud_texel(tex[x0 + n * (y1 + n * z1)]), c7 = cloud_texel(tex[x1 + n * (y1 + n * z1)]);
  auto tri = [&](float q0, float q1, float q2, float q3, float q4, float q5, float q6, float q7) {
    const float a0 = q0 + ax * (q1 - q0), a1 = q2 + ax * (q3 - q2), a2 = q4 + ax * (q5 - q4), a3 = q6 + ax * (q7 - q6);
    const float b0 = a0 + ay * (a1 - a0), b1 = a2 + ay * (a3 - a2);
    return b0 + az * (b1 - b0);
  };
  return make_float4(tri(c0.x, c1.x, c2.x, c3.x, c4.x, c5.x, c6.x, c7.x), tri(c0.y, c1.y, c2.y, c3.y, c4.y, c5.y, c6.y, c7.y),
                     tri(c0.z, c1.z, c2.z, c3.z, c4.z, c5.z, c6.z, c7.z), tri(c0.w, c1.w, c2.w, c3.w, c4.w, c5.w, c6.w, c7.w));
}
LUM_DEV float4 cloud_tex2d(const uint32_t* __restrict__ tex, int n, float u, float v) {  // texture_load without flip and gamma (cloud_utils.cuh:77-79)
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
  const float4 c00 = cloud_texel(tex[x0 + y0 * n]), c10 = cloud_texel(tex[x1 + y0 * n]), c01 = cloud_texel(tex[x0 + y1 * n]), c11 = cloud_texel(tex[x1 + y1 * n]);
  float4 r;
  { const float top = c00.x + ax * (c10.x - c00.x), bot = c01.x + ax * (c11.x - c01.x); r.x = top + ay * (bot - top); }
  { const float top = c00.y + ax * (c10.y - c00.y), bot = c01.y + ax * (c11.y - c01.y); r.y = top + ay * (bot - top); }
  { const float top = c00.z + ax * (c10.z - c00.z), bot = c01.z + ax * (c11.z - c01.z); r.z = top + ay * (bot - top); }
  { const float top = c00.w + ax * (c10.w - c00.w), bot = c01.w + ax * (c11.w - c01.w); r.w = top + ay * (bot - top); }
  return r;
}

// ---- layers, weather, density (cloud_utils.cuh) ----
struct CloudWeather { float coverage, type, coverage1, coverage2; };
struct CloudGradient { float g[4]; };
LUM_DEV float cloud_gradient(const CloudGradient& c, float height) { return c_smoothstep(height, c.g[0], c.g[1]) - c_smoothstep(height, c.g[2], c.g[3]); }
LUM_DEV CloudGradient cloud_gradient_stratus() { return CloudGradient{{0.01f, 0.15f, 0.17f, 0.3f}}; }
LUM_DEV CloudGradient cloud_gradient_stratocumulus() { return CloudGradient{{0.01f, 0.12f, 0.45f, 0.6f}}; }
LUM_DEV CloudGradient cloud_gradient_cumulus() { return CloudGradient{{0.01f, 0.06f, 0.8f, 0.99f}}; }
LUM_DEV float cloud_height(const DeviceScene& sc, V3 pos, int layer) {
  const float* L = sc.cloud_layers[layer];
  return (sky_height(pos) - L[kClHeightMin]) / (L[kClHeightMax] - L[kClHeightMin]);
}
LUM_DEV CloudWeather cloud_weather(const DeviceScene& sc, V3 pos, float height, int layer) {
  const float* L = sc.cloud_layers[layer];
  pos.x += sc.cloud_offset_x;
  pos.z += sc.cloud_offset_z;
  V3 wp = pos;
  wp.x = wp.x + L[kClWindSpeed] * height * L[kClWindCos];
  wp.z = wp.z + L[kClWindSpeed] * height * L[kClWindSin];
  const float k = (layer == kCloudLow) ? 0.012f : (layer == kCloudMid) ? 0.01f : 0.004f;
  wp = wp * (k * sc.cloud_noise_weather_scale);
  const float4 tex = cloud_tex2d(sc.cloud_noise_weather, kCloudWeatherRes, wp.x, wp.z);
  CloudWeather w{0.0f, 0.0f, 0.0f, 0.0f};
  if (layer == kCloudLow) {
    w.coverage = saturate(c_remap(tex.x * L[kClCoverage], 0.0f, 1.0f, L[kClCoverageMin], 1.0f));
    w.type = saturate(c_remap(tex.y * L[kClType], 0.0f, 1.0f, L[kClTypeMin], 1.0f));
  }
  else if (layer == kCloudMid) {
    w.coverage = saturate(c_remap(tex.z * L[kClCoverage], 0.0f, 1.0f, L[kClCoverageMin], 1.0f));
    w.type = saturate(c_remap(tex.w * L[kClType], 0.0f, 1.0f, L[kClTypeMin], 1.0f));
  }
  else {
    w.coverage = saturate(c_remap(tex.x * L[kClCoverage], 0.0f, 1.0f, L[kClCoverageMin], 1.0f));
    w.coverage1 = saturate(c_remap(tex.y * L[kClCoverage], 0.0f, 1.0f, L[kClCoverageMin], 1.0f));
    w.coverage2 = saturate(c_remap(tex.z * L[kClCoverage], 0.0f, 1.0f, L[kClCoverageMin], 1.0f));
  }
  return w;
}
LUM_DEV CloudGradient cloud_gradient_type(int layer, const CloudWeather& w) {
  CloudGradient out;
  if (layer == kCloudLow) {
    const float stratus = 1.0f - saturate(w.type * 2.0f), stratocumulus = 1.0f - fabsf(2.0f * w.type - 1.0f), cumulus = saturate(2.0f * w.type - 1.0f);
    const CloudGradient a = cloud_gradient_stratus(), b = cloud_gradient_stratocumulus(), c = cloud_gradient_cumulus();
#pragma unroll
    for (int k = 0; k < 4; k++) out.g[k] = stratus * a.g[k] + stratocumulus * b.g[k] + cumulus * c.g[k];
  }
  else if (layer == kCloudMid) {
    const float altostratus = 1.0f - saturate(w.type), altocumulus = saturate(w.type);
    const CloudGradient a{{0.01f, 0.5f, 0.5f, 0.95f}}, b{{0.25f, 0.30f, 0.60f, 0.75f}};
#pragma unroll
    for (int k = 0; k < 4; k++) out.g[k] = altostratus * a.g[k] + altocumulus * b.g[k];
  }
  else out = CloudGradient{{0.01f, 0.20f, 0.80f, 0.95f}};
  return out;
}
LUM_DEV bool cloud_significant_point(float height, const CloudWeather& w, int layer) {
  const CloudGradient type = cloud_gradient_type(layer, w);
  const bool covered = (layer == kCloudTop) ? (w.coverage > kCloudWeatherCutoff || w.coverage1 > kCloudWeatherCutoff || w.coverage2 > kCloudWeatherCutoff)
                                            : (w.coverage > kCloudWeatherCutoff);
  return covered && (type.g[0] < height) && (type.g[3] > height);
}
LUM_DEV F2 cloud_layer_intersection(const DeviceScene& sc, V3 origin, V3 ray, float limit, int layer) {  // cloud_utils.cuh:214-277: (start, distance)
  const float* L = sc.cloud_layers[layer];
  if (L[kClActive] == 0.0f) return F2{kFltMax, 0.0f};
  const float hmin = L[kClHeightMin] + kSkyEarthRadius, hmax = L[kClHeightMax] + kSkyEarthRadius;
  const float height = length(origin);
  const float dist_hmax = sph_int_p0(ray, origin, hmax), dist_hmin = sph_int_p0(ray, origin, hmin);
  float start;
  if (height > hmax) start = dist_hmax;
  else if (height < hmin) start = dist_hmin;
  else start = 0.0f;
  const float end_1 = (height < hmin) ? dist_hmax : dist_hmin;
  const float end_2 = (height > hmax) ? sph_int_back_p0(ray, origin, hmax) : dist_hmax;
  const float end_dist = fminf(end_1, end_2);
  const float earth_hit = sph_int_p0(ray, origin, kSkyEarthRadius);
  const float distance = fminf(earth_hit, fminf(limit, end_dist)) - start;
  if (distance < 0.0f) start = kFltMax;
  return F2{start, distance};
}
LUM_DEV float cloud_density(const DeviceScene& sc, V3 pos, float height, const CloudWeather& w, int layer) {  // cloud_utils.cuh:283-403
  const float* L = sc.cloud_layers[layer];
  pos.x += sc.cloud_offset_x;
  pos.z += sc.cloud_offset_z;
  float density;
  const float density_gradient = cloud_gradient(cloud_gradient_type(layer, w), height);
  if (layer == kCloudLow) {
    V3 sp = pos;
    sp.x = sp.x + L[kClWindSpeed] * height * L[kClWindCos] * 0.33f;
    sp.z = sp.z + L[kClWindSpeed] * height * L[kClWindSin] * 0.33f;
    sp = sp * (0.4f * sc.cloud_noise_shape_scale);
    const float4 shape = cloud_tex3d(sc.cloud_noise_shape, kCloudShapeRes, sp.x, sp.y, sp.z);
    float shape_sum = shape.x * 5.0f;
    shape_sum += shape.y * cloud_gradient(cloud_gradient_stratus(), height);
    shape_sum += shape.z * cloud_gradient(cloud_gradient_stratocumulus(), height);
    shape_sum += shape.w * cloud_gradient(cloud_gradient_cumulus(), height);
    shape_sum *= 0.16f;
    density = fabsf(shape_sum * density_gradient);
    density = pow_det(density, saturate(height * 6.0f));
    density = c_smoothstep(density, 0.25f, 1.1f);
    density = saturate(density - (1.0f - w.coverage)) * w.coverage;
  }
  else {
    const V3 sp = pos * (0.2f * sc.cloud_noise_shape_scale);
    const float4 shape = cloud_tex3d(sc.cloud_noise_shape, kCloudShapeRes, sp.x, sp.y, sp.z);
    if (layer == kCloudMid) {
      const float d0 = (shape.x * 0.5f + shape.y * 0.25f + shape.w * 0.125f + shape.z * 0.125f) * sqrtf(w.coverage);
      const float d1 = c_smoothstep(shape.x * 0.1f + shape.y * 0.7f + shape.w * 0.1f + shape.z * 0.1f, 0.50f, 1.0f) * w.coverage;
      const float interp = c_smoothstep(w.type, 0.1f, 0.5f);
      density = c_remap01(density_gradient * (d0 * (1.0f - interp) + d1 * interp), 0.05f, 1.0f);
    }
    else {
      const float d1 = (shape.x * 0.3f + shape.y * 0.3f + shape.z * 0.2f + shape.w * 0.2f) * sqrtf(w.coverage1);
      const float d2 = c_smoothstep(shape.x * 0.2f + shape.y * 0.4f + shape.w * 0.2f + shape.z * 0.2f, 0.50f, 1.0f) * w.coverage2;
      density = c_remap01(density_gradient * (d1 + d2) * 0.25f, 0.05f, 1.0f);
    }
  }
  if (layer != kCloudTop && density > 0.0f) {  // cloud_erode_density
    const V3 dp = pos * (2.0f * sc.cloud_noise_detail_scale);
    const float4 detail = cloud_tex3d(sc.cloud_noise_detail, kCloudDetailRes, dp.x, dp.y, dp.z);
    const float detail_fbm = saturate(detail.x * 0.625f + detail.y * 0.25f + detail.z * 0.125f);
    const float noise_modifier = lerpf(1.0f - detail_fbm, detail_fbm, saturate(height * 10.0f));
    density = c_remap(density, noise_modifier * 0.2f, 1.0f, 0.0f, 1.0f);
  }
  return fmaxf(density * sc.cloud_density, 0.0f);
}

// ---- the shadow the layers cast into the sky march (cloud_shadow.cuh) ----
LUM_DEV bool cloud_shadow_layer(const DeviceScene& sc, V3 origin, V3 ray, int step_count, int layer) {
  const float* L = sc.cloud_layers[layer];
  const F2 isect = cloud_layer_intersection(sc, origin, ray, kFltMax, layer);
  const float max_dist = 6.0f * (L[kClHeightMax] - L[kClHeightMin]);
  const float start = isect.x, dist = fminf(isect.y, max_dist);
  if (start != kFltMax && dist > 0.0f) {
    const float step_size = dist / step_count;
    float reach = start + 0.1f * step_size;
#pragma nounroll
    for (int i = 0; i < step_count; i++) {
      const V3 pos = origin + ray * reach;
      const float height = cloud_height(sc, pos, layer);
      if (height < 0.0f || height > 1.0f) break;
      const CloudWeather w = cloud_weather(sc, pos, height, layer);
      if (cloud_significant_point(height, w, layer)) {
        if (cloud_density(sc, pos, height, w, layer) > 0.0f) return true;
      }
      reach += step_size;
    }
  }
  return false;
}
LUM_DEV float cloud_shadow(const DeviceScene& sc, V3 origin, V3 ray) {
  if (!sc.cloud_active || !sc.cloud_atmosphere_scattering || !sc.cloud_noise_shape) return 1.0f;
  if (sc.cloud_layers[0][kClActive] != 0.0f && cloud_shadow_layer(sc, origin, ray, (int) sc.cloud_steps / 3, kCloudLow)) return 0.0f;
  if (sc.cloud_layers[1][kClActive] != 0.0f && cloud_shadow_layer(sc, origin, ray, (int) sc.cloud_steps / 16, kCloudMid)) return 0.1f;
  if (sc.cloud_layers[2][kClActive] != 0.0f && cloud_shadow_layer(sc, origin, ray, (int) sc.cloud_steps / 32, kCloudTop)) return 0.5f;
  return 1.0f;
}

LUM_NS_END
