// Procedural sky (first part): the atmosphere seen by rays that leave the scene when sky.mode = DEFAULT, its two
// look-up tables, and the sun disk.
// Reference: cuda/sky.cuh:47-108 (densities, path through the atmosphere), :110-176 (transmittance LUT, [Bru17]), :186-332
// (multiscattering LUT, [Hil20]), :338-446 (ray-marched single scattering + multiscattering), :508-515, :567-577 (sky_color_main),
// cuda/sky_utils.cuh (spectrum of 8 wavelengths, LUT parametrisation, spectrum -> RGB), cuda/math.cuh:620-779 (sphere tests),
// :1162-1239 (Henyey-Greenstein / Draine / Jendersie-Eon phase functions), :1429-1439 (solid angle of the sun).
// Kept out: cloud shadows, aerial perspective (off by default), HDRI mode.
// Numerics contract as everywhere: IEEE + - x / sqrt only; expf := exp2_det(x * log2 e); asinf(x) := atan2_det(x, sqrt(1 - x^2));
// the LUTs are float4 pairs filtered in software (clamp addressing, exact lerps) instead of by the texture unit.
#pragma once

#include "dev_light.h"
#include "dev_sampler.h"
#include "dev_scene.h"

LUM_NS_BEGIN

constexpr float kSkyEarthRadius = 6371.0f, kSkySunRadius = 696340.0f, kSkySunDistance = 149597870.0f, kSkyAtmoHeight = 100.0f;  // sky_defines.h
constexpr float kSkyMoonRadius = 1737.4f;
constexpr float kRefPi = 3.141592653589f;  // the reference's PI (utils.h:14), where it enters texture coordinates and grid cells
constexpr float kSkyAtmoRadius = kSkyAtmoHeight + kSkyEarthRadius;
constexpr float kSkyHeightOffset = 0.0005f;
constexpr int kSkyTmWidth = 256, kSkyTmHeight = 64, kSkyMsSize = 32, kSkyMsBase = 16, kSkyMsIter = 256;
// RANDOM_TARGET_SKY_STEP_OFFSET and RandomSet::LIGHT_SUN<0> (geometry, material.cuh:61) by the allocation rule of random.cuh:24-66
constexpr uint32_t kRndSkyStepOffset = 77, kRndSkyInscatteringStep = 79, kRndSunBsdf = 346, kRndSunBsdfMethod = 349, kRndSunRay = 352, kRndSunResampling = 355;
constexpr float kSkyMieScattering = 3.996f * 0.001f, kSkyMieExtinction = 4.440f * 0.001f;

struct Spectrum { float v[8]; };
#define LUM_SPECTRUM_OP(expr) Spectrum r; _Pragma("unroll") for (int i = 0; i < 8; i++) r.v[i] = (expr); return r;
LUM_DEV Spectrum sp_set1(float x) { LUM_SPECTRUM_OP(x) }
LUM_DEV Spectrum sp_add(const Spectrum& a, const Spectrum& b) { LUM_SPECTRUM_OP(a.v[i] + b.v[i]) }
LUM_DEV Spectrum sp_sub(const Spectrum& a, const Spectrum& b) { LUM_SPECTRUM_OP(a.v[i] - b.v[i]) }
LUM_DEV Spectrum sp_mul(const Spectrum& a, const Spectrum& b) { LUM_SPECTRUM_OP(a.v[i] * b.v[i]) }
LUM_DEV Spectrum sp_scale(const Spectrum& a, float b) { LUM_SPECTRUM_OP(a.v[i] * b) }
LUM_DEV Spectrum sp_inv(const Spectrum& a) { LUM_SPECTRUM_OP(1.0f / a.v[i]) }
LUM_DEV float exp_det(float x) { return exp2_det(x * 1.44269504f); }
LUM_DEV Spectrum sp_exp(const Spectrum& a) { LUM_SPECTRUM_OP(exp_det(a.v[i])) }
LUM_DEV Spectrum sp_merge(float4 lo, float4 hi) { Spectrum r; r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w; r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w; return r; }
// sky_utils.cuh:112-127, :251-269
LUM_DEV Spectrum sp_ident() { return Spectrum{{8.4205e-03f, 2.6449e-01f, 4.0273e-01f, 1.6624e-01f, 2.4324e-01f, 3.5849e-01f, 3.6342e-01f, 2.4177e-01f}}; }
LUM_DEV Spectrum sky_sun_radiance() { return Spectrum{{2.463170e+04f, 2.888721e+04f, 2.795153e+04f, 2.629836e+04f, 2.667237e+04f, 2.638737e+04f, 2.490630e+04f, 2.338930e+04f}}; }
LUM_DEV Spectrum sky_rayleigh_scattering() { return Spectrum{{3.945800e-02f, 2.939289e-02f, 2.235060e-02f, 1.730112e-02f, 1.360286e-02f, 1.084340e-02f, 8.750306e-03f, 7.139216e-03f}}; }
LUM_DEV Spectrum sky_ozone_extinction() { return Spectrum{{1.484836e-05f, 8.501668e-05f, 2.646158e-04f, 7.953520e-04f, 1.661103e-03f, 2.510733e-03f, 2.697211e-03f, 1.727741e-03f}}; }

// sky_utils.cuh:289-316
LUM_DEV Col sky_color_from_spectrum(const Spectrum& s) {
  const float r = 0.00640271f * s.v[0] + 0.179441f * s.v[1] + 0.04852f * s.v[2] - 0.43822f * s.v[3] - 0.920721f * s.v[4] - 0.0226871f * s.v[5] + 1.83443f * s.v[6] + 2.36265f * s.v[7];
  const float g = -0.00550232f * s.v[0] - 0.164f * s.v[1] - 0.119836f * s.v[2] + 0.365423f * s.v[3] + 1.28952f * s.v[4] + 1.41809f * s.v[5] + 0.629138f * s.v[6] - 0.0816028f * s.v[7];
  const float b = 0.0386558f * s.v[0] + 1.21426f * s.v[1] + 1.80395f * s.v[2] + 0.475181f * s.v[3] - 0.0638328f * s.v[4] - 0.169502f * s.v[5] - 0.114583f * s.v[6] - 0.0374822f * s.v[7];
  return col(fmaxf(r, 0.0f), fmaxf(g, 0.0f), fmaxf(b, 0.0f));
}

// ---- sphere tests about the origin / about p (math.cuh:620-779) ----
LUM_DEV float sph_int_p0(V3 ray, V3 origin, float r) {
  const float d0 = dot(origin, ray), r2 = r * r;
  const V3 k = origin - ray * d0;
  const float d = r2 - dot(k, k);
  if (d < 0.0f) return kFltMax;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  const float c = dot(origin, origin) - r2;
  const float t0 = c / q;
  if (t0 >= 0.0f) return t0;
  return (q >= 0.0f) ? q : kFltMax;
}
LUM_DEV float sph_int_back_p0(V3 ray, V3 origin, float r) {
  const float d0 = dot(origin, ray), r2 = r * r;
  const V3 k = origin - ray * d0;
  const float d = r2 - dot(k, k);
  if (d < 0.0f) return kFltMax;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  const float c = dot(origin, origin) - r2;
  if (q >= 0.0f) return q;
  const float t0 = c / q;
  return (t0 >= 0.0f) ? t0 : kFltMax;
}
LUM_DEV bool sph_hit_p0(V3 ray, V3 origin, float r) {
  const float d0 = dot(origin, ray), r2 = r * r;
  const V3 k = origin - ray * d0;
  const float d = r2 - dot(k, k);
  if (d < 0.0f) return false;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  const float c = dot(origin, origin) - r2;
  return (c / q) >= 0.0f;
}
LUM_DEV float sphere_int(V3 ray, V3 origin, V3 p, float r) {
  const V3 diff = origin - p;
  const float d0 = dot(diff, ray), r2 = r * r;
  const float c = dot(diff, diff) - r2;
  const V3 k = diff - ray * d0;
  const float d = r2 - dot(k, k);
  if (d < 0.0f) return kFltMax;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  const float t0 = c / q;
  if (t0 >= 0.0f) return t0;
  return (q >= 0.0f) ? q : kFltMax;
}
LUM_DEV bool sphere_hit(V3 ray, V3 origin, V3 p, float r);
LUM_DEV float asin_det(float x) { return atan2_det(x, sqrtf(fmaxf(1.0f - x * x, 0.0f))); }
// math.cuh:1429-1439
LUM_DEV float sphere_solid_angle(V3 p, float r, V3 origin) {
  const float d = length(p - origin);
  if (d < r) return 2.0f * kPi;
  const float a = asin_det(r / d);
  return 2.0f * kPi * a * a;
}

// ---- the atmosphere's parameters as the kernels see them (DeviceSky, device_structs.h:101-124) ----
struct SkyView {
  uint32_t steps, ozone_absorption;
  V3 geometry_offset, sun_pos;
  float sun_strength, base_density, rayleigh_density, mie_density, ozone_density, rayleigh_falloff, mie_falloff, ground_visibility, ozone_layer_thickness,
    multiscattering_factor;
  float g_hg, g_d, alpha, w_d;  // Jendersie-Eon parameters of mie_diameter (math.cuh:1189-1232), evaluated once by the host layer
  const float4* tm;             // transmittance LUT: low plane [64][256], then high plane
  const float4* ms;             // multiscattering LUT: low plane [32][32], then high plane
  V3 moon_pos;
  float moon_tex_offset, stars_intensity;
  uint32_t moon_albedo_tex, moon_normal_tex, stars_count;
  const float4* stars;
  const uint32_t* stars_offsets;
  const float4* cloud_hdri;  // the panorama when its fourth channel (the baked clouds' transmittance) dims the sun: HDRI mode with active clouds
  uint32_t cloud_hdri_dim;
};

LUM_DEV SkyView sky_view(const DeviceScene& sc) {
  SkyView s;
  s.steps = sc.sky_steps; s.ozone_absorption = sc.sky_ozone_absorption;
  s.geometry_offset = v3(sc.sky_geometry_offset[0], sc.sky_geometry_offset[1], sc.sky_geometry_offset[2]);
  s.sun_pos = v3(sc.sky_sun_pos[0], sc.sky_sun_pos[1], sc.sky_sun_pos[2]);
  s.sun_strength = sc.sky_sun_strength; s.base_density = sc.sky_base_density; s.rayleigh_density = sc.sky_rayleigh_density; s.mie_density = sc.sky_mie_density;
  s.ozone_density = sc.sky_ozone_density; s.rayleigh_falloff = sc.sky_rayleigh_falloff; s.mie_falloff = sc.sky_mie_falloff;
  s.ground_visibility = sc.sky_ground_visibility; s.ozone_layer_thickness = sc.sky_ozone_layer_thickness; s.multiscattering_factor = sc.sky_multiscattering_factor;
  s.g_hg = sc.sky_mie_phase[0]; s.g_d = sc.sky_mie_phase[1]; s.alpha = sc.sky_mie_phase[2]; s.w_d = sc.sky_mie_phase[3];
  s.tm = sc.sky_lut_transmittance; s.ms = sc.sky_lut_multiscattering;
  s.moon_pos = v3(sc.sky_moon_pos[0], sc.sky_moon_pos[1], sc.sky_moon_pos[2]);
  s.moon_tex_offset = sc.sky_moon_tex_offset; s.stars_intensity = sc.sky_stars_intensity;
  s.moon_albedo_tex = sc.sky_moon_albedo_tex; s.moon_normal_tex = sc.sky_moon_normal_tex; s.stars_count = sc.sky_stars_count;
  s.stars = sc.sky_stars; s.stars_offsets = sc.sky_stars_offsets;
  const bool cloud_hdri = sc.cloud_active && sc.sky_mode == kSkyHdri && sc.sky_hdri != nullptr && sc.sky_hdri_dim != 0u;
  s.cloud_hdri = cloud_hdri ? sc.sky_hdri : nullptr; s.cloud_hdri_dim = cloud_hdri ? sc.sky_hdri_dim : 0u;
  return s;
}

// sky_utils.cuh:9-31, :82-101; sky.cuh:47-75
LUM_DEV float sky_height(V3 p) { return length(p) - kSkyEarthRadius; }
LUM_DEV V3 world_to_sky(const SkyView& s, V3 p) { return v3(p.x * 0.001f, p.y * 0.001f + kSkyEarthRadius, p.z * 0.001f) + s.geometry_offset; }
LUM_DEV float sky_sub_to_unit_uv(float u, float res) { return (u - 0.5f / res) * (res / (res - 1.0f)); }
LUM_DEV float sky_rayleigh_phase(float c) { return 3.0f * (1.0f + c * c) / (16.0f * 3.1415926535f); }
LUM_DEV float sky_rayleigh_density(const SkyView& s, float h) { return 2.5f * s.base_density * exp_det(-h * (1.0f / s.rayleigh_falloff)); }
LUM_DEV float sky_mie_density(const SkyView& s, float h) {
  const float inso = exp_det(-h * (1.0f / s.mie_falloff));
  float waso = 0.0f;
  if (h < 2.0f) waso = 1.0f + 0.125f * (2.0f - h);
  else if (h < 3.0f) waso = 3.0f - h;
  waso *= 60.0f / s.ground_visibility;
  return s.base_density * (inso + waso);
}
LUM_DEV float sky_ozone_density(const SkyView& s, float h) {
  if (!s.ozone_absorption) return 0.0f;
  const float min_val = (h > 25.0f) ? 0.0f : 0.1f;
  return s.base_density * fmaxf(min_val, 1.0f - fabsf(h - 25.0f) / s.ozone_layer_thickness);
}
// math.cuh:1162-1239
LUM_DEV float hg_phase(float c, float g) {
  const float g2 = g * g;
  const float den = 1.0f + g2 - 2.0f * g * c;
  return (1.0f - g * g) / (4.0f * kPi * (den * sqrtf(den)));
}
LUM_DEV float sky_mie_phase(const SkyView& s, float c) {
  const float hg = hg_phase(c, s.g_hg);
  const float dr = hg_phase(c, s.g_d) * ((1.0f + s.alpha * c * c) / (1.0f + (s.alpha / 3.0f) * (1.0f + 2.0f * s.g_d * s.g_d)));
  return (1.0f - s.w_d) * hg + s.w_d * dr;
}
struct SkyMedium { Spectrum scattering_rayleigh, scattering, extinction; float scattering_mie; };
LUM_DEV SkyMedium sky_medium(const SkyView& s, float height) {
  const float dr = sky_rayleigh_density(s, height) * s.rayleigh_density, dm = sky_mie_density(s, height) * s.mie_density, doz = sky_ozone_density(s, height) * s.ozone_density;
  SkyMedium m;
  m.scattering_rayleigh = sp_scale(sky_rayleigh_scattering(), dr);
  m.scattering_mie = kSkyMieScattering * dm;
  const Spectrum ext_r = sp_scale(sky_rayleigh_scattering(), dr);  // SKY_RAYLEIGH_EXTINCTION = SKY_RAYLEIGH_SCATTERING
  const float ext_m = kSkyMieExtinction * dm;
  const Spectrum ext_o = sp_scale(sky_ozone_extinction(), doz);
  m.scattering = sp_add(m.scattering_rayleigh, sp_set1(m.scattering_mie));
  m.extinction = sp_add(sp_add(ext_r, sp_set1(ext_m)), ext_o);
  return m;
}

// sky.cuh:78-108: start and length of the ray's path through the shell [min_height, max_height]
LUM_DEV F2 sky_compute_path(V3 origin, V3 ray, float min_height, float max_height) {
  const float height = length(origin);
  if (height <= min_height) return F2{0.0f, -kFltMax};
  float distance, start = 0.0f;
  if (height > max_height) {
    const float earth = sph_int_p0(ray, origin, min_height), atmo = sph_int_p0(ray, origin, max_height), atmo2 = sph_int_back_p0(ray, origin, max_height);
    distance = fminf(earth - atmo, atmo2 - atmo);
    start = atmo;
  }
  else {
    const float earth = sph_int_p0(ray, origin, min_height), atmo = sph_int_p0(ray, origin, max_height);
    distance = fminf(earth, atmo);
  }
  return F2{start, distance};
}

// ---- LUT fetch: normalised coordinates, linear filter, clamp addressing (device_sky.c:46-60), exact float lerps ----
LUM_DEV Spectrum sky_lut_fetch(const float4* __restrict__ lut, int w, int h, float u, float v) {
  const float x = u * (float) w - 0.5f, y = v * (float) h - 0.5f;
  const float fx = floorf(x), fy = floorf(y);
  const float tx = x - fx, ty = y - fy;
  const int x0 = min(max((int) fx, 0), w - 1), x1 = min(max((int) fx + 1, 0), w - 1);
  const int y0 = min(max((int) fy, 0), h - 1), y1 = min(max((int) fy + 1, 0), h - 1);
  Spectrum r;
#pragma unroll
  for (int plane = 0; plane < 2; plane++) {
    const float4* __restrict__ p = lut + plane * w * h;
    const float4 a = p[y0 * w + x0], b = p[y0 * w + x1], c = p[y1 * w + x0], d = p[y1 * w + x1];
    const float ax[4] = {a.x, a.y, a.z, a.w}, bx[4] = {b.x, b.y, b.z, b.w}, cx[4] = {c.x, c.y, c.z, c.w}, dx[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float top = ax[k] + tx * (bx[k] - ax[k]), bot = cx[k] + tx * (dx[k] - cx[k]);
      r.v[plane * 4 + k] = top + ty * (bot - top);
    }
  }
  return r;
}
// sky_utils.cuh:273-287 [Hil20]
LUM_DEV F2 sky_transmittance_uv(float height, float zenith_cos) {
  height += kSkyEarthRadius;
  const float H = sqrtf(fmaxf(0.0f, kSkyAtmoRadius * kSkyAtmoRadius - kSkyEarthRadius * kSkyEarthRadius));
  const float rho = sqrtf(fmaxf(0.0f, height * height - kSkyEarthRadius * kSkyEarthRadius));
  const float disc = height * height * (zenith_cos * zenith_cos - 1.0f) + kSkyAtmoRadius * kSkyAtmoRadius;
  const float d = fmaxf(0.0f, (-height * zenith_cos + sqrtf(disc)));
  const float d_min = kSkyAtmoRadius - height, d_max = rho + H;
  return F2{(d - d_min) / (d_max - d_min), rho / H};
}

// ---- LUT generation ----
// sky_compute_transmittance_optical_depth + sky_compute_transmittance_lut (sky.cuh:110-176), one texel per thread
LUM_DEV Spectrum sky_optical_depth(const SkyView& s, float r, float mu) {
  const int steps = 2500;
  const float disc = r * r * (mu * mu - 1.0f) + kSkyAtmoRadius * kSkyAtmoRadius;
  const float dist = fmaxf(-r * mu + sqrtf(fmaxf(0.0f, disc)), 0.0f);
  const float step_size = dist / steps;
  Spectrum depth = sp_set1(0.0f);
  for (int i = 0; i <= steps; i++) {
    const float reach = i * step_size;
    const float height = sqrtf(reach * reach + 2.0f * r * mu * reach + r * r) - kSkyEarthRadius;
    const SkyMedium m = sky_medium(s, height);
    const float w = (i == 0 || i == steps) ? 0.5f : 1.0f;
    depth = sp_add(depth, sp_scale(m.extinction, w * step_size));
  }
  return depth;
}
#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
__global__ __launch_bounds__(64) void k_sky_transmittance_lut(DeviceScene sc, float4* __restrict__ dst) {
  const int id = blockIdx.x * 64 + threadIdx.x;
  if (id >= kSkyTmWidth * kSkyTmHeight) return;
  const SkyView s = sky_view(sc);
  const int y = id / kSkyTmWidth, x = id - y * kSkyTmWidth;
  float fx = ((float) x + 0.5f) / kSkyTmWidth, fy = ((float) y + 0.5f) / kSkyTmHeight;
  fx = sky_sub_to_unit_uv(fx, kSkyTmWidth); fy = sky_sub_to_unit_uv(fy, kSkyTmHeight);
  const float H = sqrtf(kSkyAtmoRadius * kSkyAtmoRadius - kSkyEarthRadius * kSkyEarthRadius);
  const float rho = H * fy;
  const float r = sqrtf(rho * rho + kSkyEarthRadius * kSkyEarthRadius);
  const float d_min = kSkyAtmoRadius - r, d_max = rho + H;
  const float d = d_min + fx * (d_max - d_min);
  float mu = (d == 0.0f) ? 1.0f : (H * H - rho * rho - d * d) / (2.0f * r * d);
  mu = fminf(1.0f, fmaxf(-1.0f, mu));
  const Spectrum t = sp_exp(sp_scale(sky_optical_depth(s, r, mu), -1.0f));
  dst[id] = make_float4(t.v[0], t.v[1], t.v[2], t.v[3]);
  dst[kSkyTmWidth * kSkyTmHeight + id] = make_float4(t.v[4], t.v[5], t.v[6], t.v[7]);
}
#endif

struct SkyMsResult { Spectrum L, ms_as_1; };
// sky_compute_multiscattering_integration, sky.cuh:186-273
LUM_DEV SkyMsResult sky_multiscattering_integration(const SkyView& s, V3 origin, V3 ray, V3 sun_pos) {
  SkyMsResult res;
  res.L = sp_set1(0.0f); res.ms_as_1 = sp_set1(0.0f);
  const F2 path = sky_compute_path(origin, ray, kSkyEarthRadius, kSkyAtmoRadius);
  if (path.y == -kFltMax) return res;
  const float start = path.x, distance = path.y;
  if (distance > 0.0f) {
    const int steps = 500;
    float reach = start;
    const float light_angle = sphere_solid_angle(sun_pos, kSkySunRadius, origin);
    Spectrum transmittance = sp_set1(1.0f);
    for (int i = 0; i < steps; i++) {
      const float new_reach = start + distance * (i + 0.3f) / steps;
      const float step_size = new_reach - reach;
      reach = new_reach;
      const V3 pos = origin + ray * reach;
      const float height = sky_height(pos);
      const V3 ray_scatter = normalize(sun_pos - pos);
      const float cos_angle = dot(ray, ray_scatter);
      const float phase_r = sky_rayleigh_phase(cos_angle), phase_m = sky_mie_phase(s, cos_angle);
      const float zenith_cos = dot(normalize(pos), ray_scatter);
      const F2 uv = sky_transmittance_uv(height, zenith_cos);
      const Spectrum extinction_sun = sky_lut_fetch(s.tm, kSkyTmWidth, kSkyTmHeight, uv.x, uv.y);
      const SkyMedium m = sky_medium(s, height);
      const Spectrum phase_times_scattering = sp_add(sp_scale(m.scattering_rayleigh, phase_r), sp_set1(m.scattering_mie * phase_m));
      const float shadow = sph_hit_p0(ray_scatter, pos, kSkyEarthRadius) ? 0.0f : 1.0f;
      const Spectrum S = sp_scale(sp_mul(extinction_sun, phase_times_scattering), shadow * light_angle);
      const Spectrum step_t = sp_exp(sp_scale(m.extinction, -step_size));
      const Spectrum inv_ext = sp_inv(m.extinction);
      const Spectrum ss_int = sp_mul(sp_sub(S, sp_mul(S, step_t)), inv_ext);
      const Spectrum ms_int = sp_mul(sp_sub(m.scattering, sp_mul(m.scattering, step_t)), inv_ext);
      res.L = sp_add(res.L, sp_mul(ss_int, transmittance));
      res.ms_as_1 = sp_add(res.ms_as_1, sp_mul(ms_int, transmittance));
      transmittance = sp_mul(transmittance, step_t);
    }
  }
  return res;
}
#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// sky_compute_multiscattering_lut, sky.cuh:276-332: one workgroup of 256 directions per texel, shared-memory tree reduction
__global__ __launch_bounds__(256) void k_sky_multiscattering_lut(DeviceScene sc, float4* __restrict__ dst) {
  __shared__ Spectrum lum_shared[kSkyMsIter], ms_shared[kSkyMsIter];
  const SkyView s = sky_view(sc);
  const int x = blockIdx.x, y = blockIdx.y;
  float fx = ((float) x + 0.5f) / kSkyMsSize, fy = ((float) y + 0.5f) / kSkyMsSize;
  fx = sky_sub_to_unit_uv(fx, kSkyMsSize); fy = sky_sub_to_unit_uv(fy, kSkyMsSize);
  const float cos_angle = fx * 2.0f - 1.0f;
  const V3 sun_dir = v3(0.0f, cos_angle, sqrtf(saturate(1.0f - cos_angle * cos_angle)));
  const float height = kSkyEarthRadius + saturate(fy + kSkyHeightOffset) * (kSkyAtmoHeight - kSkyHeightOffset);
  const V3 pos = v3(0.0f, height, 0.0f), sun_pos = sun_dir * kSkySunDistance;
  const float sqrt_sample = (float) kSkyMsBase;
  const float a = (float) (threadIdx.x / kSkyMsBase), b = (float) (threadIdx.x - (threadIdx.x / kSkyMsBase) * kSkyMsBase);
  const V3 ray = sample_ray_sphere(2.0f * (a / sqrt_sample) - 1.0f, b / sqrt_sample);
  const SkyMsResult r = sky_multiscattering_integration(s, pos, ray, sun_pos);
  lum_shared[threadIdx.x] = r.L; ms_shared[threadIdx.x] = r.ms_as_1;
  for (int i = kSkyMsIter >> 1; i > 0; i >>= 1) {
    __syncthreads();
    if ((int) threadIdx.x < i) {
      lum_shared[threadIdx.x] = sp_add(lum_shared[threadIdx.x], lum_shared[threadIdx.x + i]);
      ms_shared[threadIdx.x] = sp_add(ms_shared[threadIdx.x], ms_shared[threadIdx.x + i]);
    }
  }
  if (threadIdx.x > 0) return;
  const Spectrum luminance = sp_scale(lum_shared[0], 1.0f / (sqrt_sample * sqrt_sample));
  const Spectrum multiscattering = sp_scale(ms_shared[0], 1.0f / (sqrt_sample * sqrt_sample));
  const Spectrum contribution = sp_inv(sp_sub(sp_set1(1.0f), multiscattering));
  const Spectrum L = sp_scale(sp_mul(luminance, contribution), s.multiscattering_factor);
  const int id = x + y * kSkyMsSize;
  dst[id] = make_float4(L.v[0], L.v[1], L.v[2], L.v[3]);
  dst[kSkyMsSize * kSkyMsSize + id] = make_float4(L.v[4], L.v[5], L.v[6], L.v[7]);
}
#endif

LUM_DEV Spectrum sky_moon_solar_flux() { return Spectrum{{1.7f, 1.8f, 2.0f, 1.9f, 1.87f, 1.7f, 1.65f, 1.55f}}; }  // sky_utils.cuh:272
// math.cuh:781-789
LUM_DEV V3 angles_to_direction(float altitude, float azimuth) {
  float sa, ca, sz, cz;
  sincos_det(altitude, sa, ca); sincos_det(azimuth, sz, cz);
  return v3(cz * ca, sa, sz * ca);
}

LUM_NS_END
#include "dev_cloud.h"
LUM_NS_BEGIN

// ---- sky_compute_atmosphere (sky.cuh:338-505): ray-marched atmosphere, then sun disk, moon and stars. `cloud_shadows`: the sun's single scattering
// is shadowed by the cloud layers (dev_cloud.h). `transmittance_out` is multiplied by the transmittance of the marched segment. ----
LUM_DEV Spectrum sky_compute_atmosphere(const DeviceScene& sc, const SkyView& s, Spectrum& transmittance_out, V3 origin, V3 ray, float limit, bool celestials, bool cloud_shadows,
                                        int steps, float random_offset) {
  Spectrum result = sp_set1(0.0f);
  const F2 path = sky_compute_path(origin, ray, kSkyEarthRadius, kSkyAtmoRadius);
  const float start = path.x, distance = fminf(path.y, limit - start);
  Spectrum transmittance = sp_ident();
  if (distance > 0.0f) {
    float reach = start;
    const float light_angle = sphere_solid_angle(s.sun_pos, kSkySunRadius, origin);
    for (int i = 0; i < steps; i++) {
      const float new_reach = start + distance * (i + random_offset) / steps;
      const float step_size = new_reach - reach;
      reach = new_reach;
      const V3 pos = origin + ray * reach;
      const float height = sky_height(pos);
      const V3 ray_scatter = normalize(s.sun_pos - pos);
      const float cos_angle = dot(ray, ray_scatter);
      const float zenith_cos = dot(normalize(pos), ray_scatter);
      const float phase_r = sky_rayleigh_phase(cos_angle), phase_m = sky_mie_phase(s, cos_angle);
      const float shadow = sph_hit_p0(ray_scatter, pos, kSkyEarthRadius) ? 0.0f : (cloud_shadows ? cloud_shadow(sc, pos, ray_scatter) : 1.0f);
      const F2 uv = sky_transmittance_uv(height, zenith_cos);
      const Spectrum extinction_sun = sky_lut_fetch(s.tm, kSkyTmWidth, kSkyTmHeight, uv.x, uv.y);
      const SkyMedium m = sky_medium(s, height);
      const Spectrum phase_times_scattering = sp_add(sp_scale(m.scattering_rayleigh, phase_r), sp_set1(m.scattering_mie * phase_m));
      const Spectrum ss_radiance = sp_scale(sp_mul(extinction_sun, phase_times_scattering), shadow * light_angle);
      const Spectrum ms_tex = sky_lut_fetch(s.ms, kSkyMsSize, kSkyMsSize, zenith_cos * 0.5f + 0.5f, height / kSkyAtmoHeight);
      const Spectrum S = sp_add(ss_radiance, sp_mul(ms_tex, m.scattering));
      const Spectrum step_t = sp_exp(sp_scale(m.extinction, -step_size));
      const Spectrum s_int = sp_mul(sp_sub(S, sp_mul(S, step_t)), sp_inv(m.extinction));
      result = sp_add(result, sp_mul(s_int, transmittance));
      transmittance = sp_mul(transmittance, step_t);
    }
    result = sp_mul(result, sp_scale(sky_sun_radiance(), s.sun_strength));
  }
  if (celestials) {
    const float sun_hit = sphere_int(ray, origin, s.sun_pos, kSkySunRadius);
    const float earth_hit = sph_int_p0(ray, origin, kSkyEarthRadius);
    const bool has_moon = s.moon_albedo_tex != 0xFFFFFFFFu;
    const float moon_hit = has_moon ? sphere_int(ray, origin, s.moon_pos, kSkyMoonRadius) : kFltMax;
    if (earth_hit > sun_hit && moon_hit > sun_hit) result = sp_add(result, sp_mul(transmittance, sp_scale(sky_sun_radiance(), s.sun_strength)));
    else if (earth_hit > moon_hit) {
      const V3 moon_point = origin + ray * moon_hit;
      const V3 bounce_ray = normalize(s.sun_pos - moon_point);
      if (!sphere_hit(bounce_ray, moon_point, v3(0.0f, 0.0f, 0.0f), kSkyEarthRadius)) {
        V3 normal = normalize(moon_point - s.moon_pos);
        const float tex_u = 0.5f + s.moon_tex_offset + atan2_det(normal.z, normal.x) * (1.0f / (2.0f * kRefPi));
        const float tex_v = 0.5f + asin_det(normal.y) * (1.0f / kRefPi);
        const F2 uv = F2{tex_u, tex_v};
        // create_basis + transform_vec3, math.cuh:301-321, :445-453
        const float sign = copysignf(1.0f, normal.z);
        const float a = -1.0f / (sign + normal.z);
        const float b = normal.x * normal.y * a;
        const V3 u1 = v3(1.0f + sign * normal.x * normal.x * a, sign * b, -sign * normal.x);
        const V3 u2 = v3(b, sign + normal.y * normal.y * a, -normal.y);
        const float4 nv = texture_load(sc, s.moon_normal_tex, uv, true, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        const V3 mn = v3(nv.x * 2.0f - 1.0f, nv.y * 2.0f - 1.0f, nv.z * 2.0f - 1.0f);
        normal = normalize(v3(u1.x * mn.x + u2.x * mn.y + normal.x * mn.z, u1.y * mn.x + u2.y * mn.y + normal.y * mn.z, u1.z * mn.x + u2.z * mn.y + normal.z * mn.z));
        const float NdotL = dot(normal, bounce_ray);
        if (NdotL > 0.0f) {
          const float albedo = texture_load(sc, s.moon_albedo_tex, uv, true, make_float4(0.0f, 0.0f, 0.0f, 0.0f)).x;
          const float light_angle = sphere_solid_angle(s.sun_pos, kSkySunRadius, moon_point);
          const float weight = albedo * s.sun_strength * NdotL * light_angle / (2.0f * kRefPi);
          result = sp_add(result, sp_mul(transmittance, sp_mul(sky_moon_solar_flux(), sp_scale(sky_sun_radiance(), weight))));
        }
      }
    }
    if (s.stars != nullptr && sun_hit == kFltMax && earth_hit == kFltMax && moon_hit == kFltMax) {
      const float ray_altitude = asin_det(ray.y);
      const float ray_azimuth = atan2_det(-ray.z, -ray.x) + kRefPi;
      const uint32_t x = f2u_sat(ray_azimuth * 10.0f), y = f2u_sat((ray_altitude + kRefPi * 0.5f) * 10.0f);
      const uint32_t grid = min(x, 63u) + min(y, 31u) * 64u;  // the reference does not clamp; x = 63 and y = 31 are the last cells the generator fills
      const uint32_t first = s.stars_offsets[grid], last = s.stars_offsets[grid + 1u];
      for (uint32_t i = first; i < last; i++) {
        const float4 star = s.stars[i];
        const V3 star_pos = angles_to_direction(star.x, star.y);
        if (sphere_hit(ray, v3(0.0f, 0.0f, 0.0f), star_pos, star.z)) result = sp_add(result, sp_scale(transmittance, star.w * s.stars_intensity));
      }
    }
  }
  transmittance_out = sp_mul(transmittance_out, transmittance);
  return result;
}
// sky_get_color, sky.cuh:508-515
LUM_DEV Col sky_get_color(const DeviceScene& sc, const SkyView& s, V3 origin, V3 ray, float limit, bool celestials, int steps, float random_offset) {
  Spectrum unused = sp_set1(0.0f);
  return sky_color_from_spectrum(sky_compute_atmosphere(sc, s, unused, origin, ray, limit, celestials, false, steps, random_offset));
}
// Aerial perspective, sky_trace_inscattering (sky.cuh:517-532): the air between a ray's origin and its hit scatters sun light towards the
// viewer and dims what lies behind. `limit` in sky units (km); returns the in-scattered colour times `record`, and dims `record`.
LUM_DEV Col sky_trace_inscattering(const DeviceScene& sc, const SkyView& s, V3 origin, V3 ray, float limit, Col& record, bool primary_ray, float step_random, float random_offset) {
  Spectrum transmittance = sp_set1(1.0f);
  const float base_range = primary_ray ? 40.0f : 80.0f;
  const int steps = (int) (fminf(fmaxf(0.5f, limit / base_range), 2.0f) * (float) (s.steps / 6u) + step_random - 0.5f);
  const Spectrum radiance = sky_compute_atmosphere(sc, s, transmittance, origin, ray, limit, false, true, steps, random_offset);
  const Col inscattering = sky_color_from_spectrum(radiance) * record;
  record = record * sky_color_from_spectrum(transmittance);
  return inscattering;
}

// ---- HDRI bake (cuda/sky_hdri.cuh:13-160, device/device_sky.c:283-316): the sky without celestial bodies seen from `origin`, as an
// equirectangular dim x dim image. 32 lanes per texel share its samples; their means go through the reference's trimmed mean. ----
LUM_DEV float sky_hdri_median_of_means(float* buckets, uint32_t num_buckets) {  // sky_hdri.cuh:13-56, on this group's 32 LDS slots
  for (uint32_t i = 1; i < num_buckets; i++) {
    const float x = buckets[i];
    uint32_t j = i;
    while (j > 0 && buckets[j - 1] > x) { buckets[j] = buckets[j - 1]; j--; }
    buckets[j] = x;
  }
  float num = 0.0f, denom = 0.0f;
  for (uint32_t b = 0; b < num_buckets; b++) { num += (float) b * buckets[b]; denom += buckets[b]; }
  num *= 2.0f;
  denom *= (float) num_buckets;
  const float G = saturate((num / denom) - ((float) num_buckets + 1.0f) / (float) num_buckets);
  const uint32_t k = num_buckets >> 1;
  const uint32_t c = f2u_sat((float) k - (1.0f - G) * (float) k);
  float output = 0.0f;
  for (uint32_t b = c; b < num_buckets - c; b++) output += buckets[b];
  return output / (float) (num_buckets - 2u * c);
}
// (the bake kernel k_sky_hdri is in kernels.h: it marches the clouds, dev_cloud_march.h)

// ---- sun next-event estimation (cuda/direct_lighting.cuh:21-119, :352-383; cuda/bsdf.cuh:355-458) ----
LUM_DEV bool sphere_hit(V3 ray, V3 origin, V3 p, float r) {  // math.cuh:679-696
  const V3 diff = origin - p;
  const float d0 = dot(diff, ray), r2 = r * r;
  const float c = dot(diff, diff) - r2;
  const V3 k = diff - ray * d0;
  const float d = r2 - dot(k, k);
  if (d < 0.0f) return false;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  return (c / q) >= 0.0f;
}
// math.cuh:277-299
LUM_DEV V3 sample_hemisphere_basis(float altitude, float azimuth, V3 basis) {
  const float sign = copysignf(1.0f, basis.z);
  const float a = -1.0f / (sign + basis.z);
  const float b = basis.x * basis.y * a;
  const V3 u1 = v3(1.0f + sign * basis.x * basis.x * a, sign * b, -sign * basis.x);
  const V3 u2 = v3(b, sign + basis.y * basis.y * a, -basis.y);
  float sa, ca, sz, cz;
  sincos_det(altitude, sa, ca); sincos_det(azimuth, sz, cz);
  const float c1 = sa * cz, c2 = sa * sz, c3 = ca;
  return normalize(v3(c1 * u1.x + c2 * u2.x + c3 * basis.x, c1 * u1.y + c2 * u2.y + c3 * basis.y, c1 * u1.z + c2 * u2.z + c3 * basis.z));
}
// math.cuh:1393-1419
LUM_DEV V3 sample_sphere(V3 p, float r, V3 origin, F2 random, float& area) {
  float r1 = random.x, r2 = random.y;
  V3 dir = p - origin;
  const float d = length(dir);
  if (d < r) { area = 4.0f * kPi; return normalize(sample_ray_sphere(2.0f * r1 - 1.0f, r2)); }
  r1 = 0.999f * r1; r2 = 0.999f * r2;
  dir = dir * (1.0f / d);
  const float angle = asin_det(saturate(r / d));
  area = 2.0f * kPi * angle * angle;
  const float u = sqrtf(r1) * angle, v = 2.0f * kPi * r2;
  return normalize(sample_hemisphere_basis(u, v, dir));
}
// sky_utils.cuh:318-347 (no clouds, no HDRI)
// sky_get_sun_color (sky_utils.cuh:318-347); `include_cloud_hdri`: in HDRI mode with active clouds the panorama's fourth channel dims the sun
LUM_DEV Col sky_sun_color(const SkyView& s, V3 origin, V3 ray, bool include_cloud_hdri = true) {
  const float height = sky_height(origin);
  const float zenith_cos = dot(normalize(origin), ray);
  const F2 uv = sky_transmittance_uv(height, zenith_cos);
  const Spectrum extinction_sun = sp_mul(sp_ident(), sky_lut_fetch(s.tm, kSkyTmWidth, kSkyTmHeight, uv.x, uv.y));
  Col sun_color = sky_color_from_spectrum(sp_mul(extinction_sun, sp_scale(sky_sun_radiance(), s.sun_strength)));
  if (include_cloud_hdri && s.cloud_hdri != nullptr) {
    const float theta = atan2_det(ray.z, ray.x), phi = asin_det(ray.y);
    const float u = (theta + kRefPi) / (2.0f * kRefPi);
    const float v = 1.0f - ((phi + 0.5f * kRefPi) / kRefPi);
    const float dim = (float) s.cloud_hdri_dim;
    const uint32_t x = (uint32_t) ((u - floorf(u)) * dim) % s.cloud_hdri_dim, y = (uint32_t) ((v - floorf(v)) * dim) % s.cloud_hdri_dim;
    sun_color = sun_color * s.cloud_hdri[x + (size_t) y * s.cloud_hdri_dim].w;
  }
  return sun_color;
}
// ---- baked panorama as the sky (sky mode HDRI) ----
// sky_hdri_sample, sky_utils.cuh:49-63: equirectangular lookup, nearest texel (device_sky.c:352), wrap addressing (texture_create's
// default), no gamma. The texel of a normalised coordinate u is floor(frac(u) * dim), as the texture unit's point filter picks it.
LUM_DEV Col sky_hdri_sample(const DeviceScene& sc, V3 ray) {
  if (sc.sky_hdri == nullptr) return splat(0.0f);
  const float theta = atan2_det(ray.z, ray.x), phi = asin_det(ray.y);
  const float u = (theta + kRefPi) / (2.0f * kRefPi);
  const float v = 1.0f - ((phi + 0.5f * kRefPi) / kRefPi);
  const float dim = (float) sc.sky_hdri_dim;
  const uint32_t x = (uint32_t) ((u - floorf(u)) * dim) % sc.sky_hdri_dim, y = (uint32_t) ((v - floorf(v)) * dim) % sc.sky_hdri_dim;
  const float4 t = sc.sky_hdri[x + (size_t) y * sc.sky_hdri_dim];
  return col(t.x, t.y, t.z);
}
// sky_color_main / sky_color_no_compute in HDRI mode (sky.cuh:534-606): the panorama holds no sun disk, so rays that may still see
// emitters add it when they hit the disk above the horizon.
LUM_DEV Col sky_hdri_color(const DeviceScene& sc, V3 origin, V3 ray, uint32_t state) {
  Col c = sky_hdri_sample(sc, ray);
  if (state & (kStCameraDirection | kStAllowEmission)) {
    const SkyView s = sky_view(sc);
    const V3 sky_origin = world_to_sky(s, origin);
    if (sphere_hit(ray, sky_origin, s.sun_pos, kSkySunRadius) && !sph_hit_p0(ray, sky_origin, kSkyEarthRadius)) c = c + sky_sun_color(s, sky_origin, ray);
  }
  return c;
}
// bsdf_sample_for_sun_pdf<GEOMETRY>, bsdf.cuh:438-458. The reference hands the WORLD-space view vector to the bounded-VNDF density,
// which reads it as a local one; kept as it is.
LUM_DEV float sun_bsdf_pdf(const GeoContext& g, V3 L, float reflection_prob, float refraction_prob) {
  const RayTerms c = analyze_direction(g.params, g.normal, g.V, L);
  const float roughness = g.params.roughness();
  if (c.is_refraction) return refraction_prob * pdf_refraction(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, g.params.ior());
  return reflection_prob * pdf_vndf_bounded(g.V, roughness, c.NdotH, c.NdotV);
}
// direct_lighting_sun_create_task + direct_lighting_sun_direct: two candidate directions (BSDF sample, sun solid angle), one kept by
// resampling. Returns false when there is nothing to trace.
template <class Smp>
LUM_DEV bool sample_sun(const DeviceScene& sc, const SkyView& sky, const LocalFrame& lf, const GeoContext& g, const Smp& smp, Col& light_out, V3& dir_out) {
  const Energy energy = energy_terms(sc, g.params, world_ndotv(g));  // bsdf_evaluate analyses the direction in world space, like sample_light
  const V3 sky_pos = world_to_sky(sky, g.position);
  const bool sun_below_horizon = sph_hit_p0(normalize(sky.sun_pos - sky_pos), sky_pos, kSkyEarthRadius);
  const bool inside_earth = length(sky_pos) < kSkyEarthRadius;
  if (sun_below_horizon || inside_earth) return false;
  const MatParams& p = g.params;
  // bsdf_sample_for_light_probabilities, bsdf.cuh:355-374
  const bool translucent = (p.flags & kMatSubstrateMask) == kMatTranslucent;
  const float w_refl = 1.0f, w_refr = translucent ? 1.0f : 0.0f;
  const float reflection_prob = w_refl / (w_refl + w_refr), refraction_prob = w_refr / (w_refl + w_refr);
  // bsdf_sample_for_sun<GEOMETRY>, bsdf.cuh:380-403
  const float roughness = p.roughness();
  V3 ray_local;
  if (smp.next1(kRndSunBsdfMethod) < reflection_prob) ray_local = reflect(lf.V, sample_vndf_bounded(lf.V, roughness, smp.next2(kRndSunBsdf)));
  else { bool total_reflection; ray_local = refract(lf.V, sample_vndf_caps(lf.V, roughness, smp.next2(kRndSunBsdf)), p.ior(), total_reflection); }
  const V3 dir_bsdf = normalize(qapply(qinv(lf.to_z), ray_local));
  Col light_bsdf = splat(0.0f);
  bool is_refraction;
  if (sphere_hit(dir_bsdf, sky_pos, sky.sun_pos, kSkySunRadius)) light_bsdf = sky_sun_color(sky, sky_pos, dir_bsdf) * eval_bsdf(energy, g, dir_bsdf, kHintGeneral, is_refraction, 1.0f);
  float solid_angle;
  const V3 dir_sa = sample_sphere(sky.sun_pos, kSkySunRadius, sky_pos, smp.next2(kRndSunRay), solid_angle);
  const Col light_sa = sky_sun_color(sky, sky_pos, dir_sa) * eval_bsdf(energy, g, dir_sa, kHintGeneral, is_refraction, 1.0f);
  const float target_bsdf = importance(light_bsdf), target_sa = importance(light_sa);
  const float mis_bsdf = solid_angle / (sun_bsdf_pdf(g, dir_bsdf, reflection_prob, refraction_prob) * solid_angle + 1.0f);
  const float mis_sa = solid_angle / (sun_bsdf_pdf(g, dir_sa, reflection_prob, refraction_prob) * solid_angle + 1.0f);
  const float weight_bsdf = target_bsdf * mis_bsdf, weight_sa = target_sa * mis_sa;
  const float sum_weights = weight_bsdf + weight_sa;
  if (sum_weights == 0.0f) return false;
  float target;
  Col light;
  if (smp.next1(kRndSunResampling) * sum_weights < weight_bsdf) { dir_out = dir_bsdf; target = target_bsdf; light = light_bsdf; }
  else { dir_out = dir_sa; target = target_sa; light = light_sa; }
  light = light * (sum_weights / target);
  if (target == 0.0f) return false;
  if (importance(light) == 0.0f) return false;
  light_out = light;
  return true;
}

LUM_NS_END
