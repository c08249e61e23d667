/*
 * ORACLE (test infrastructure, not product): procedural sky, first part of SURVEY §8 f4.
 *
 * Restated from cuda/sky.cuh:47-108 (densities, path through the atmosphere), :110-176 (transmittance LUT), :186-332 (multiscattering
 * LUT), :338-446 (sky_compute_atmosphere), :508-515, :567-577 (sky_get_color / sky_color_main, DEFAULT branch), cuda/sky_utils.cuh
 * (8-wavelength spectrum, LUT parametrisation, spectrum -> RGB), cuda/math.cuh:620-779 (sphere tests), :1162-1239 (phase functions),
 * :1429-1439 (solid angle of the sun). Out: cloud shadows, aerial perspective, HDRI mode.
 * expf := o_exp2(x * log2 e); asinf(x) := o_atan2(x, sqrt(1 - x^2)); the LUTs are filtered in software (clamp addressing, exact lerps)
 * where the reference uses the texture unit. The Jendersie-Eon parameters of the droplet diameter arrive with the scene.
 * Parity unpinned, like the rest of the oracle.
 */
#ifndef ORACLE_O_SKY_H
#define ORACLE_O_SKY_H

#include "o_light.h"
#include "o_rng.h"
#include "oracle.h"

#define SKY_EARTH_RADIUS 6371.0f
#define SKY_SUN_RADIUS 696340.0f
#define SKY_SUN_DISTANCE 149597870.0f
#define SKY_MOON_RADIUS 1737.4f
#define REF_PI 3.141592653589f /* utils.h:14: the reference's PI where it enters texture coordinates and grid cells */
#define SKY_ATMO_HEIGHT 100.0f
#define SKY_ATMO_RADIUS (SKY_ATMO_HEIGHT + SKY_EARTH_RADIUS)
#define SKY_HEIGHT_OFFSET 0.0005f
#define SKY_TM_W 256
#define SKY_TM_H 64
#define SKY_MS_SIZE 32
#define SKY_MS_BASE 16
#define SKY_MS_ITER 256
#define RANDOM_TARGET_SKY_INSCATTERING_STEP 79u
#define RANDOM_TARGET_SKY_STEP_OFFSET 77u /* allocation rule of random.cuh:24-66; the sun targets are RandomSet::LIGHT_SUN<0> (material.cuh:61) */
#define RT_SUN_BSDF 346u
#define RT_SUN_BSDF_METHOD 349u
#define RT_SUN_RAY 352u
#define RT_SUN_RESAMPLING 355u
#define SKY_MIE_SCATTERING (3.996f * 0.001f)
#define SKY_MIE_EXTINCTION (4.440f * 0.001f)

typedef struct { float v[8]; } Spectrum;
static inline Spectrum sp_set1(float x) { Spectrum r; for (int i = 0; i < 8; i++) r.v[i] = x; return r; }
static inline Spectrum sp_add(Spectrum a, Spectrum b) { Spectrum r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
static inline Spectrum sp_sub(Spectrum a, Spectrum b) { Spectrum r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
static inline Spectrum sp_mul(Spectrum a, Spectrum b) { Spectrum r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i] * b.v[i]; return r; }
static inline Spectrum sp_scale(Spectrum a, float b) { Spectrum r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i] * b; return r; }
static inline Spectrum sp_inv(Spectrum a) { Spectrum r; for (int i = 0; i < 8; i++) r.v[i] = 1.0f / a.v[i]; return r; }
static inline float o_exp(float x) { return o_exp2(x * 1.44269504f); }
static inline Spectrum sp_exp(Spectrum a) { Spectrum r; for (int i = 0; i < 8; i++) r.v[i] = o_exp(a.v[i]); return r; }
static const Spectrum SP_IDENT = {{8.4205e-03f, 2.6449e-01f, 4.0273e-01f, 1.6624e-01f, 2.4324e-01f, 3.5849e-01f, 3.6342e-01f, 2.4177e-01f}};
static const Spectrum SKY_SUN_RADIANCE = {{2.463170e+04f, 2.888721e+04f, 2.795153e+04f, 2.629836e+04f, 2.667237e+04f, 2.638737e+04f, 2.490630e+04f, 2.338930e+04f}};
static const Spectrum SKY_RAYLEIGH_SCATTERING = {{3.945800e-02f, 2.939289e-02f, 2.235060e-02f, 1.730112e-02f, 1.360286e-02f, 1.084340e-02f, 8.750306e-03f, 7.139216e-03f}};
static const Spectrum SKY_OZONE_EXTINCTION = {{1.484836e-05f, 8.501668e-05f, 2.646158e-04f, 7.953520e-04f, 1.661103e-03f, 2.510733e-03f, 2.697211e-03f, 1.727741e-03f}};

/* sky_utils.cuh:289-316 */
static inline RGBF sky_color_from_spectrum(Spectrum s) {
  const float r = 0.00640271f * s.v[0] + 0.179441f * s.v[1] + 0.04852f * s.v[2] - 0.43822f * s.v[3] - 0.920721f * s.v[4] - 0.0226871f * s.v[5] + 1.83443f * s.v[6] + 2.36265f * s.v[7];
  const float g = -0.00550232f * s.v[0] - 0.164f * s.v[1] - 0.119836f * s.v[2] + 0.365423f * s.v[3] + 1.28952f * s.v[4] + 1.41809f * s.v[5] + 0.629138f * s.v[6] - 0.0816028f * s.v[7];
  const float b = 0.0386558f * s.v[0] + 1.21426f * s.v[1] + 1.80395f * s.v[2] + 0.475181f * s.v[3] - 0.0638328f * s.v[4] - 0.169502f * s.v[5] - 0.114583f * s.v[6] - 0.0374822f * s.v[7];
  return c3(fmaxf(r, 0.0f), fmaxf(g, 0.0f), fmaxf(b, 0.0f));
}

/* math.cuh:620-779 */
static inline float sph_int_p0(vec3 ray, vec3 origin, float r) {
  const float d0 = v_dot(origin, ray), r2 = r * r;
  const vec3 k = v_sub(origin, v_scale(ray, d0));
  const float d = r2 - v_dot(k, k);
  if (d < 0.0f) return FLT_MAX;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  const float c = v_dot(origin, origin) - r2;
  const float t0 = c / q;
  if (t0 >= 0.0f) return t0;
  return (q >= 0.0f) ? q : FLT_MAX;
}
static inline float sph_int_back_p0(vec3 ray, vec3 origin, float r) {
  const float d0 = v_dot(origin, ray), r2 = r * r;
  const vec3 k = v_sub(origin, v_scale(ray, d0));
  const float d = r2 - v_dot(k, k);
  if (d < 0.0f) return FLT_MAX;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  const float c = v_dot(origin, origin) - r2;
  if (q >= 0.0f) return q;
  const float t0 = c / q;
  return (t0 >= 0.0f) ? t0 : FLT_MAX;
}
static inline bool sph_hit_p0(vec3 ray, vec3 origin, float r) {
  const float d0 = v_dot(origin, ray), r2 = r * r;
  const vec3 k = v_sub(origin, v_scale(ray, d0));
  const float d = r2 - v_dot(k, k);
  if (d < 0.0f) return false;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  const float c = v_dot(origin, origin) - r2;
  return (c / q) >= 0.0f;
}
static inline float sphere_int(vec3 ray, vec3 origin, vec3 p, float r) {
  const vec3 diff = v_sub(origin, p);
  const float d0 = v_dot(diff, ray), r2 = r * r;
  const float c = v_dot(diff, diff) - r2;
  const vec3 k = v_sub(diff, v_scale(ray, d0));
  const float d = r2 - v_dot(k, k);
  if (d < 0.0f) return FLT_MAX;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  const float t0 = c / q;
  if (t0 >= 0.0f) return t0;
  return (q >= 0.0f) ? q : FLT_MAX;
}
static inline float o_asin(float x) { return o_atan2(x, sqrtf(fmaxf(1.0f - x * x, 0.0f))); }
static inline float sphere_solid_angle(vec3 p, float r, vec3 origin) { /* math.cuh:1429-1439 */
  const float d = v_len(v_sub(p, origin));
  if (d < r) return 2.0f * O_PI;
  const float a = o_asin(r / d);
  return 2.0f * O_PI * a * a;
}

typedef struct {
  uint32_t steps, ozone_absorption;
  vec3 geometry_offset, sun_pos;
  float sun_strength, base_density, rayleigh_density, mie_density, ozone_density, rayleigh_falloff, mie_falloff, ground_visibility, ozone_layer_thickness,
    multiscattering_factor;
  float g_hg, g_d, alpha, w_d;
  const float* tm; /* low plane [64][256][4], high plane */
  const float* ms; /* low plane [32][32][4], high plane */
  vec3 moon_pos;
  float moon_tex_offset, stars_intensity;
  uint32_t moon_albedo_tex, moon_normal_tex, stars_count;
  const float* stars;
  const uint32_t* stars_offsets;
  const OracleScene* scene; /* for the moon's textures */
} OSky;

static inline OSky osky_view(const OracleScene* sc) {
  OSky s;
  s.steps = sc->sky_steps; s.ozone_absorption = sc->sky_ozone_absorption;
  s.geometry_offset = v3(sc->sky_geometry_offset[0], sc->sky_geometry_offset[1], sc->sky_geometry_offset[2]);
  s.sun_pos = v3(sc->sky_sun_pos[0], sc->sky_sun_pos[1], sc->sky_sun_pos[2]);
  s.sun_strength = sc->sky_sun_strength; s.base_density = sc->sky_base_density; s.rayleigh_density = sc->sky_rayleigh_density; s.mie_density = sc->sky_mie_density;
  s.ozone_density = sc->sky_ozone_density; s.rayleigh_falloff = sc->sky_rayleigh_falloff; s.mie_falloff = sc->sky_mie_falloff;
  s.ground_visibility = sc->sky_ground_visibility; s.ozone_layer_thickness = sc->sky_ozone_layer_thickness; s.multiscattering_factor = sc->sky_multiscattering_factor;
  s.g_hg = sc->sky_mie_phase[0]; s.g_d = sc->sky_mie_phase[1]; s.alpha = sc->sky_mie_phase[2]; s.w_d = sc->sky_mie_phase[3];
  s.tm = sc->sky_lut_transmittance; s.ms = sc->sky_lut_multiscattering;
  s.moon_pos = v3(sc->sky_moon_pos[0], sc->sky_moon_pos[1], sc->sky_moon_pos[2]);
  s.moon_tex_offset = sc->sky_moon_tex_offset; s.stars_intensity = sc->sky_stars_intensity;
  s.moon_albedo_tex = sc->sky_moon_albedo_tex; s.moon_normal_tex = sc->sky_moon_normal_tex; s.stars_count = sc->sky_stars_count;
  s.stars = sc->sky_stars; s.stars_offsets = sc->sky_stars_offsets;
  s.scene = sc;
  return s;
}

static inline float sky_height(vec3 p) { return v_len(p) - SKY_EARTH_RADIUS; }
static inline vec3 world_to_sky(const OSky* s, vec3 p) { return v_add(v3(p.x * 0.001f, p.y * 0.001f + SKY_EARTH_RADIUS, p.z * 0.001f), s->geometry_offset); }
static inline float sky_sub_to_unit_uv(float u, float res) { return (u - 0.5f / res) * (res / (res - 1.0f)); }
static inline float sky_rayleigh_phase(float c) { return 3.0f * (1.0f + c * c) / (16.0f * 3.1415926535f); }
static inline float sky_rayleigh_density(const OSky* s, float h) { return 2.5f * s->base_density * o_exp(-h * (1.0f / s->rayleigh_falloff)); }
static inline float sky_mie_density(const OSky* s, float h) {
  const float inso = o_exp(-h * (1.0f / s->mie_falloff));
  float waso = 0.0f;
  if (h < 2.0f) waso = 1.0f + 0.125f * (2.0f - h);
  else if (h < 3.0f) waso = 3.0f - h;
  waso *= 60.0f / s->ground_visibility;
  return s->base_density * (inso + waso);
}
static inline float sky_ozone_density(const OSky* s, float h) {
  if (!s->ozone_absorption) return 0.0f;
  const float min_val = (h > 25.0f) ? 0.0f : 0.1f;
  return s->base_density * fmaxf(min_val, 1.0f - fabsf(h - 25.0f) / s->ozone_layer_thickness);
}
static inline float hg_phase(float c, float g) {
  const float g2 = g * g;
  const float den = 1.0f + g2 - 2.0f * g * c;
  return (1.0f - g * g) / (4.0f * O_PI * (den * sqrtf(den)));
}
static inline float sky_mie_phase(const OSky* s, float c) {
  const float hg = hg_phase(c, s->g_hg);
  const float dr = hg_phase(c, s->g_d) * ((1.0f + s->alpha * c * c) / (1.0f + (s->alpha / 3.0f) * (1.0f + 2.0f * s->g_d * s->g_d)));
  return (1.0f - s->w_d) * hg + s->w_d * dr;
}
typedef struct { Spectrum scattering_rayleigh, scattering, extinction; float scattering_mie; } SkyMedium;
static inline SkyMedium sky_medium(const OSky* s, float height) {
  const float dr = sky_rayleigh_density(s, height) * s->rayleigh_density, dm = sky_mie_density(s, height) * s->mie_density, doz = sky_ozone_density(s, height) * s->ozone_density;
  SkyMedium m;
  m.scattering_rayleigh = sp_scale(SKY_RAYLEIGH_SCATTERING, dr);
  m.scattering_mie = SKY_MIE_SCATTERING * dm;
  const Spectrum ext_r = sp_scale(SKY_RAYLEIGH_SCATTERING, dr);
  const float ext_m = SKY_MIE_EXTINCTION * dm;
  const Spectrum ext_o = sp_scale(SKY_OZONE_EXTINCTION, doz);
  m.scattering = sp_add(m.scattering_rayleigh, sp_set1(m.scattering_mie));
  m.extinction = sp_add(sp_add(ext_r, sp_set1(ext_m)), ext_o);
  return m;
}

/* sky.cuh:78-108 */
static inline float2_t sky_compute_path(vec3 origin, vec3 ray, float min_height, float max_height) {
  const float height = v_len(origin);
  float2_t r;
  if (height <= min_height) { r.x = 0.0f; r.y = -FLT_MAX; return r; }
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
  r.x = start; r.y = distance;
  return r;
}

static inline Spectrum sky_lut_fetch(const float* lut, int w, int h, float u, float v) {
  const float x = u * (float) w - 0.5f, y = v * (float) h - 0.5f;
  const float fx = floorf(x), fy = floorf(y);
  const float tx = x - fx, ty = y - fy;
  int x0 = (int) fx, x1 = (int) fx + 1, y0 = (int) fy, y1 = (int) fy + 1;
  x0 = x0 < 0 ? 0 : (x0 > w - 1 ? w - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > w - 1 ? w - 1 : x1);
  y0 = y0 < 0 ? 0 : (y0 > h - 1 ? h - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > h - 1 ? h - 1 : y1);
  Spectrum r;
  for (int plane = 0; plane < 2; plane++) {
    const float* p = lut + (size_t) plane * w * h * 4;
    const float* a = p + 4 * (y0 * w + x0); const float* b = p + 4 * (y0 * w + x1);
    const float* c = p + 4 * (y1 * w + x0); const float* d = p + 4 * (y1 * w + x1);
    for (int k = 0; k < 4; k++) {
      const float top = a[k] + tx * (b[k] - a[k]), bot = c[k] + tx * (d[k] - c[k]);
      r.v[plane * 4 + k] = top + ty * (bot - top);
    }
  }
  return r;
}
static inline float2_t sky_transmittance_uv(float height, float zenith_cos) { /* sky_utils.cuh:273-287 */
  height += SKY_EARTH_RADIUS;
  const float H = sqrtf(fmaxf(0.0f, SKY_ATMO_RADIUS * SKY_ATMO_RADIUS - SKY_EARTH_RADIUS * SKY_EARTH_RADIUS));
  const float rho = sqrtf(fmaxf(0.0f, height * height - SKY_EARTH_RADIUS * SKY_EARTH_RADIUS));
  const float disc = height * height * (zenith_cos * zenith_cos - 1.0f) + SKY_ATMO_RADIUS * SKY_ATMO_RADIUS;
  const float d = fmaxf(0.0f, (-height * zenith_cos + sqrtf(disc)));
  const float d_min = SKY_ATMO_RADIUS - height, d_max = rho + H;
  float2_t r;
  r.x = (d - d_min) / (d_max - d_min); r.y = rho / H;
  return r;
}

/* sky.cuh:110-176 */
static Spectrum sky_optical_depth(const OSky* s, float r, float mu) {
  const int steps = 2500;
  const float disc = r * r * (mu * mu - 1.0f) + SKY_ATMO_RADIUS * SKY_ATMO_RADIUS;
  const float dist = fmaxf(-r * mu + sqrtf(fmaxf(0.0f, disc)), 0.0f);
  const float step_size = dist / steps;
  Spectrum depth = sp_set1(0.0f);
  for (int i = 0; i <= steps; i++) {
    const float reach = i * step_size;
    const float height = sqrtf(reach * reach + 2.0f * r * mu * reach + r * r) - SKY_EARTH_RADIUS;
    const SkyMedium m = sky_medium(s, height);
    const float w = (i == 0 || i == steps) ? 0.5f : 1.0f;
    depth = sp_add(depth, sp_scale(m.extinction, w * step_size));
  }
  return depth;
}
static void sky_transmittance_lut(const OSky* s, float* dst) {
#pragma omp parallel for schedule(dynamic, 16)
  for (int id = 0; id < SKY_TM_W * SKY_TM_H; id++) {
    const int y = id / SKY_TM_W, x = id - y * SKY_TM_W;
    float fx = ((float) x + 0.5f) / SKY_TM_W, fy = ((float) y + 0.5f) / SKY_TM_H;
    fx = sky_sub_to_unit_uv(fx, SKY_TM_W); fy = sky_sub_to_unit_uv(fy, SKY_TM_H);
    const float H = sqrtf(SKY_ATMO_RADIUS * SKY_ATMO_RADIUS - SKY_EARTH_RADIUS * SKY_EARTH_RADIUS);
    const float rho = H * fy;
    const float r = sqrtf(rho * rho + SKY_EARTH_RADIUS * SKY_EARTH_RADIUS);
    const float d_min = SKY_ATMO_RADIUS - r, d_max = rho + H;
    const float d = d_min + fx * (d_max - d_min);
    float mu = (d == 0.0f) ? 1.0f : (H * H - rho * rho - d * d) / (2.0f * r * d);
    mu = fminf(1.0f, fmaxf(-1.0f, mu));
    const Spectrum t = sp_exp(sp_scale(sky_optical_depth(s, r, mu), -1.0f));
    for (int k = 0; k < 4; k++) { dst[4 * id + k] = t.v[k]; dst[4 * (SKY_TM_W * SKY_TM_H + id) + k] = t.v[4 + k]; }
  }
}

typedef struct { Spectrum L, ms_as_1; } SkyMsResult;
/* sky.cuh:186-273 */
static SkyMsResult sky_multiscattering_integration(const OSky* s, vec3 origin, vec3 ray, vec3 sun_pos) {
  SkyMsResult res;
  res.L = sp_set1(0.0f); res.ms_as_1 = sp_set1(0.0f);
  const float2_t path = sky_compute_path(origin, ray, SKY_EARTH_RADIUS, SKY_ATMO_RADIUS);
  if (path.y == -FLT_MAX) return res;
  const float start = path.x, distance = path.y;
  if (distance > 0.0f) {
    const int steps = 500;
    float reach = start;
    const float light_angle = sphere_solid_angle(sun_pos, SKY_SUN_RADIUS, origin);
    Spectrum transmittance = sp_set1(1.0f);
    for (int i = 0; i < steps; i++) {
      const float new_reach = start + distance * (i + 0.3f) / steps;
      const float step_size = new_reach - reach;
      reach = new_reach;
      const vec3 pos = v_add(origin, v_scale(ray, reach));
      const float height = sky_height(pos);
      const vec3 ray_scatter = v_norm(v_sub(sun_pos, pos));
      const float cos_angle = v_dot(ray, ray_scatter);
      const float phase_r = sky_rayleigh_phase(cos_angle), phase_m = sky_mie_phase(s, cos_angle);
      const float zenith_cos = v_dot(v_norm(pos), ray_scatter);
      const float2_t uv = sky_transmittance_uv(height, zenith_cos);
      const Spectrum extinction_sun = sky_lut_fetch(s->tm, SKY_TM_W, SKY_TM_H, uv.x, uv.y);
      const SkyMedium m = sky_medium(s, height);
      const Spectrum phase_times_scattering = sp_add(sp_scale(m.scattering_rayleigh, phase_r), sp_set1(m.scattering_mie * phase_m));
      const float shadow = sph_hit_p0(ray_scatter, pos, SKY_EARTH_RADIUS) ? 0.0f : 1.0f;
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
/* sky.cuh:276-332: 256 directions per texel, added by the kernel's shared-memory tree (i = 128, 64, ... 1: s[t] += s[t + i]) */
static void sky_multiscattering_lut(const OSky* s, float* dst) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int id = 0; id < SKY_MS_SIZE * SKY_MS_SIZE; id++) {
    const int y = id / SKY_MS_SIZE, x = id - y * SKY_MS_SIZE;
    float fx = ((float) x + 0.5f) / SKY_MS_SIZE, fy = ((float) y + 0.5f) / SKY_MS_SIZE;
    fx = sky_sub_to_unit_uv(fx, SKY_MS_SIZE); fy = sky_sub_to_unit_uv(fy, SKY_MS_SIZE);
    const float cos_angle = fx * 2.0f - 1.0f;
    const vec3 sun_dir = v3(0.0f, cos_angle, sqrtf(o_saturate(1.0f - cos_angle * cos_angle)));
    const float height = SKY_EARTH_RADIUS + o_saturate(fy + SKY_HEIGHT_OFFSET) * (SKY_ATMO_HEIGHT - SKY_HEIGHT_OFFSET);
    const vec3 pos = v3(0.0f, height, 0.0f), sun_pos = v_scale(sun_dir, SKY_SUN_DISTANCE);
    const float sqrt_sample = (float) SKY_MS_BASE;
    Spectrum lum[SKY_MS_ITER], ms[SKY_MS_ITER];
    for (int t = 0; t < SKY_MS_ITER; t++) {
      const float a = (float) (t / SKY_MS_BASE), b = (float) (t - (t / SKY_MS_BASE) * SKY_MS_BASE);
      const vec3 ray = sample_ray_sphere(2.0f * (a / sqrt_sample) - 1.0f, b / sqrt_sample);
      const SkyMsResult r = sky_multiscattering_integration(s, pos, ray, sun_pos);
      lum[t] = r.L; ms[t] = r.ms_as_1;
    }
    for (int i = SKY_MS_ITER >> 1; i > 0; i >>= 1)
      for (int t = 0; t < i; t++) { lum[t] = sp_add(lum[t], lum[t + i]); ms[t] = sp_add(ms[t], ms[t + i]); }
    const Spectrum luminance = sp_scale(lum[0], 1.0f / (sqrt_sample * sqrt_sample));
    const Spectrum multiscattering = sp_scale(ms[0], 1.0f / (sqrt_sample * sqrt_sample));
    const Spectrum contribution = sp_inv(sp_sub(sp_set1(1.0f), multiscattering));
    const Spectrum L = sp_scale(sp_mul(luminance, contribution), s->multiscattering_factor);
    for (int k = 0; k < 4; k++) { dst[4 * id + k] = L.v[k]; dst[4 * (SKY_MS_SIZE * SKY_MS_SIZE + id) + k] = L.v[4 + k]; }
  }
}

static const Spectrum SKY_MOON_SOLAR_FLUX = {{1.7f, 1.8f, 2.0f, 1.9f, 1.87f, 1.7f, 1.65f, 1.55f}}; /* sky_utils.cuh:272 */
static inline bool sphere_hit(vec3 ray, vec3 origin, vec3 p, float r);
static inline vec3 angles_to_direction(float altitude, float azimuth) { /* math.cuh:781-789 */
  float sa, ca, sz, cz;
  o_sincos(altitude, &sa, &ca); o_sincos(azimuth, &sz, &cz);
  return v3(cz * ca, sa, sz * ca);
}

/* sky_compute_atmosphere (sky.cuh:338-505): ray-marched atmosphere, then sun disk, moon and stars; `cloud_shadows`: the sun's single scattering is
 * shadowed by the cloud layers (o_cloud.h); *transmittance_out is multiplied by the transmittance of the marched segment */
static float cloud_shadow(const OracleScene* s, vec3 origin, vec3 ray);
static Spectrum sky_compute_atmosphere(const OSky* s, Spectrum* transmittance_out, vec3 origin, vec3 ray, float limit, bool celestials, bool cloud_shadows, int steps,
                                       float random_offset) {
  Spectrum result = sp_set1(0.0f);
  const float2_t path = sky_compute_path(origin, ray, SKY_EARTH_RADIUS, SKY_ATMO_RADIUS);
  const float start = path.x, distance = fminf(path.y, limit - start);
  Spectrum transmittance = SP_IDENT;
  if (distance > 0.0f) {
    float reach = start;
    const float light_angle = sphere_solid_angle(s->sun_pos, SKY_SUN_RADIUS, origin);
    for (int i = 0; i < steps; i++) {
      const float new_reach = start + distance * (i + random_offset) / steps;
      const float step_size = new_reach - reach;
      reach = new_reach;
      const vec3 pos = v_add(origin, v_scale(ray, reach));
      const float height = sky_height(pos);
      const vec3 ray_scatter = v_norm(v_sub(s->sun_pos, pos));
      const float cos_angle = v_dot(ray, ray_scatter);
      const float zenith_cos = v_dot(v_norm(pos), ray_scatter);
      const float phase_r = sky_rayleigh_phase(cos_angle), phase_m = sky_mie_phase(s, cos_angle);
      const float shadow = sph_hit_p0(ray_scatter, pos, SKY_EARTH_RADIUS) ? 0.0f : (cloud_shadows ? cloud_shadow(s->scene, pos, ray_scatter) : 1.0f);
      const float2_t uv = sky_transmittance_uv(height, zenith_cos);
      const Spectrum extinction_sun = sky_lut_fetch(s->tm, SKY_TM_W, SKY_TM_H, uv.x, uv.y);
      const SkyMedium m = sky_medium(s, height);
      const Spectrum phase_times_scattering = sp_add(sp_scale(m.scattering_rayleigh, phase_r), sp_set1(m.scattering_mie * phase_m));
      const Spectrum ss_radiance = sp_scale(sp_mul(extinction_sun, phase_times_scattering), shadow * light_angle);
      const Spectrum ms_tex = sky_lut_fetch(s->ms, SKY_MS_SIZE, SKY_MS_SIZE, zenith_cos * 0.5f + 0.5f, height / SKY_ATMO_HEIGHT);
      const Spectrum S = sp_add(ss_radiance, sp_mul(ms_tex, m.scattering));
      const Spectrum step_t = sp_exp(sp_scale(m.extinction, -step_size));
      const Spectrum s_int = sp_mul(sp_sub(S, sp_mul(S, step_t)), sp_inv(m.extinction));
      result = sp_add(result, sp_mul(s_int, transmittance));
      transmittance = sp_mul(transmittance, step_t);
    }
    result = sp_mul(result, sp_scale(SKY_SUN_RADIANCE, s->sun_strength));
  }
  if (celestials) {
    const float sun_hit = sphere_int(ray, origin, s->sun_pos, SKY_SUN_RADIUS);
    const float earth_hit = sph_int_p0(ray, origin, SKY_EARTH_RADIUS);
    const bool has_moon = s->moon_albedo_tex != 0xFFFFFFFFu;
    const float moon_hit = has_moon ? sphere_int(ray, origin, s->moon_pos, SKY_MOON_RADIUS) : FLT_MAX;
    if (earth_hit > sun_hit && moon_hit > sun_hit) result = sp_add(result, sp_mul(transmittance, sp_scale(SKY_SUN_RADIANCE, s->sun_strength)));
    else if (earth_hit > moon_hit) {
      const vec3 moon_point = v_add(origin, v_scale(ray, moon_hit));
      const vec3 bounce_ray = v_norm(v_sub(s->sun_pos, moon_point));
      if (!sphere_hit(bounce_ray, moon_point, v3(0.0f, 0.0f, 0.0f), SKY_EARTH_RADIUS)) {
        vec3 normal = v_norm(v_sub(moon_point, s->moon_pos));
        UV uv;
        uv.u = 0.5f + s->moon_tex_offset + o_atan2(normal.z, normal.x) * (1.0f / (2.0f * REF_PI));
        uv.v = 0.5f + o_asin(normal.y) * (1.0f / REF_PI);
        const float sign = copysignf(1.0f, normal.z);
        const float a = -1.0f / (sign + normal.z);
        const float b = normal.x * normal.y * a;
        const vec3 u1 = v3(1.0f + sign * normal.x * normal.x * a, sign * b, -sign * normal.x);
        const vec3 u2 = v3(b, sign + normal.y * normal.y * a, -normal.y);
        const float4_t nv = texture_load(s->scene, s->moon_normal_tex, uv, true, f4(0.0f, 0.0f, 0.0f, 0.0f));
        const vec3 mn = v3(nv.x * 2.0f - 1.0f, nv.y * 2.0f - 1.0f, nv.z * 2.0f - 1.0f);
        normal = v_norm(v3(u1.x * mn.x + u2.x * mn.y + normal.x * mn.z, u1.y * mn.x + u2.y * mn.y + normal.y * mn.z, u1.z * mn.x + u2.z * mn.y + normal.z * mn.z));
        const float NdotL = v_dot(normal, bounce_ray);
        if (NdotL > 0.0f) {
          const float albedo = texture_load(s->scene, s->moon_albedo_tex, uv, true, f4(0.0f, 0.0f, 0.0f, 0.0f)).x;
          const float light_angle = sphere_solid_angle(s->sun_pos, SKY_SUN_RADIUS, moon_point);
          const float weight = albedo * s->sun_strength * NdotL * light_angle / (2.0f * REF_PI);
          result = sp_add(result, sp_mul(transmittance, sp_mul(SKY_MOON_SOLAR_FLUX, sp_scale(SKY_SUN_RADIANCE, weight))));
        }
      }
    }
    if (s->stars != NULL && sun_hit == FLT_MAX && earth_hit == FLT_MAX && moon_hit == FLT_MAX) {
      const float ray_altitude = o_asin(ray.y);
      const float ray_azimuth = o_atan2(-ray.z, -ray.x) + REF_PI;
      uint32_t x = f2u_sat(ray_azimuth * 10.0f), y = f2u_sat((ray_altitude + REF_PI * 0.5f) * 10.0f);
      if (x > 63u) x = 63u;
      if (y > 31u) y = 31u;
      const uint32_t grid = x + y * 64u;
      const uint32_t first = s->stars_offsets[grid], last = s->stars_offsets[grid + 1u];
      for (uint32_t i = first; i < last; i++) {
        const float* star = s->stars + 4 * (size_t) i;
        const vec3 star_pos = angles_to_direction(star[0], star[1]);
        if (sphere_hit(ray, v3(0.0f, 0.0f, 0.0f), star_pos, star[2])) result = sp_add(result, sp_scale(transmittance, star[3] * s->stars_intensity));
      }
    }
  }
  *transmittance_out = sp_mul(*transmittance_out, transmittance);
  return result;
}
/* sky_get_color, sky.cuh:508-515 */
static RGBF sky_get_color(const OSky* s, vec3 origin, vec3 ray, float limit, bool celestials, int steps, float random_offset) {
  Spectrum unused = sp_set1(0.0f);
  return sky_color_from_spectrum(sky_compute_atmosphere(s, &unused, origin, ray, limit, celestials, false, steps, random_offset));
}
/* aerial perspective: sky_trace_inscattering, sky.cuh:517-532 (limit in sky units; IS_PRIMARY_RAY = the depth constant is 0) */
static RGBF sky_trace_inscattering(const OSky* s, vec3 origin, vec3 ray, float limit, RGBF* record, bool primary_ray, float step_random, float random_offset) {
  Spectrum transmittance = sp_set1(1.0f);
  const float base_range = primary_ray ? 40.0f : 80.0f;
  const int steps = (int) (fminf(fmaxf(0.5f, limit / base_range), 2.0f) * (float) (s->steps / 6u) + step_random - 0.5f);
  const Spectrum radiance = sky_compute_atmosphere(s, &transmittance, origin, ray, limit, false, true, steps, random_offset);
  const RGBF inscattering = c_mul(sky_color_from_spectrum(radiance), *record);
  *record = c_mul(*record, sky_color_from_spectrum(transmittance));
  return inscattering;
}

/* ---- HDRI bake (cuda/sky_hdri.cuh:13-160) ---- */
static float sky_hdri_median_of_means(float* buckets, uint32_t num_buckets) {
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
  const float G = o_saturate((num / denom) - ((float) num_buckets + 1.0f) / (float) num_buckets);
  const uint32_t k = num_buckets >> 1;
  const uint32_t c = f2u_sat((float) k - (1.0f - G) * (float) k);
  float output = 0.0f;
  for (uint32_t b = c; b < num_buckets - c; b++) output += buckets[b];
  return output / (float) (num_buckets - 2u * c);
}
static float clouds_render(const OracleScene* s, const OSky* sky, const Sampler* smp, vec3 origin, vec3 ray, float limit, RGBF* color, RGBF* transmittance,
                           float* transmittance_cloud_only); /* o_cloud.h */
/* sky_compute_hdri (sky_hdri.cuh:58-159): with active clouds the ray is marched through them first and the sky behind is dimmed by their transmittance;
 * the fourth channel is the clouds' own transmittance (the reference's separate "shadow" texture, same size and filter), 1 without clouds */
static void sky_hdri_bake(const OracleScene* sc, vec3 origin_world, uint32_t dim, uint32_t sample_count, float* dst /* dim*dim*4 */) {
  const OSky sky = osky_view(sc);
  const bool clouds = sc->cloud_active && sc->cloud_noise_shape && sc->cloud_noise_detail && sc->cloud_noise_weather;
  const float step_size = 1.0f / (float) (dim - 1u);
  const uint32_t buckets = sample_count < 32u ? sample_count : 32u;
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t pixel = 0; pixel < (int64_t) dim * dim; pixel++) {
    const uint32_t y = (uint32_t) pixel / dim, x = (uint32_t) pixel - y * dim;
    float mean[4][32];
    for (uint32_t lane = 0; lane < 32; lane++) {
      RGBF color = c_splat(0.0f);
      float alpha = 0.0f;
      uint32_t num_samples = 0;
      for (uint32_t sample_id = lane; sample_id < sample_count; sample_id += 32u) {
        const Sampler smp = {sc->bluenoise_2d, x, y, sample_id, 0};
        const float2_t jitter = rnd2(&smp, RT_CAMERA_JITTER);
        const float u = ((float) x + jitter.x) * step_size, v = 1.0f - ((float) y + jitter.y) * step_size;
        const float altitude = O_PI * v - 0.5f * O_PI, azimuth = 2.0f * O_PI * u - O_PI;
        const vec3 ray = angles_to_direction(altitude, azimuth);
        RGBF sky_color = c_splat(0.0f), transmittance = c_splat(1.0f);
        float cloud_transmittance = 1.0f;
        vec3 sky_origin = world_to_sky(&sky, origin_world);
        if (clouds) {
          const float offset = clouds_render(sc, &sky, &smp, sky_origin, ray, FLT_MAX, &sky_color, &transmittance, &cloud_transmittance);
          sky_origin = v_add(sky_origin, v_scale(ray, offset));
        }
        const RGBF behind = sky_get_color(&sky, sky_origin, ray, FLT_MAX, false, (int) sky.steps, rnd1(&smp, RANDOM_TARGET_SKY_STEP_OFFSET));
        sky_color = c_add(sky_color, c_mul(behind, transmittance));
        color = c_add(color, sky_color);
        alpha += cloud_transmittance;
        num_samples++;
      }
      mean[0][lane] = num_samples ? color.r / (float) num_samples : 0.0f;
      mean[1][lane] = num_samples ? color.g / (float) num_samples : 0.0f;
      mean[2][lane] = num_samples ? color.b / (float) num_samples : 0.0f;
      mean[3][lane] = num_samples ? alpha / (float) num_samples : 0.0f;
    }
    for (int ch = 0; ch < 4; ch++) dst[4 * pixel + ch] = sky_hdri_median_of_means(mean[ch], buckets);
  }
}

/* ---- sun next-event estimation (cuda/direct_lighting.cuh:21-119, :352-383; cuda/bsdf.cuh:355-458) ---- */
static inline bool sphere_hit(vec3 ray, vec3 origin, vec3 p, float r) { /* math.cuh:679-696 */
  const vec3 diff = v_sub(origin, p);
  const float d0 = v_dot(diff, ray), r2 = r * r;
  const float c = v_dot(diff, diff) - r2;
  const vec3 k = v_sub(diff, v_scale(ray, d0));
  const float d = r2 - v_dot(k, k);
  if (d < 0.0f) return false;
  const float sd = sqrtf(d);
  const float q = -d0 - copysignf(sd, d0);
  return (c / q) >= 0.0f;
}
static inline vec3 sample_hemisphere_basis(float altitude, float azimuth, vec3 basis) { /* math.cuh:277-299 */
  const float sign = copysignf(1.0f, basis.z);
  const float a = -1.0f / (sign + basis.z);
  const float b = basis.x * basis.y * a;
  const vec3 u1 = v3(1.0f + sign * basis.x * basis.x * a, sign * b, -sign * basis.x);
  const vec3 u2 = v3(b, sign + basis.y * basis.y * a, -basis.y);
  float sa, ca, sz, cz;
  o_sincos(altitude, &sa, &ca); o_sincos(azimuth, &sz, &cz);
  const float c1 = sa * cz, c2 = sa * sz, c3 = ca;
  return v_norm(v3(c1 * u1.x + c2 * u2.x + c3 * basis.x, c1 * u1.y + c2 * u2.y + c3 * basis.y, c1 * u1.z + c2 * u2.z + c3 * basis.z));
}
static inline vec3 sample_sphere(vec3 p, float r, vec3 origin, float2_t random, float* area) { /* math.cuh:1393-1419 */
  float r1 = random.x, r2 = random.y;
  vec3 dir = v_sub(p, origin);
  const float d = v_len(dir);
  if (d < r) { *area = 4.0f * O_PI; return v_norm(sample_ray_sphere(2.0f * r1 - 1.0f, r2)); }
  r1 = 0.999f * r1; r2 = 0.999f * r2;
  dir = v_scale(dir, 1.0f / d);
  const float angle = o_asin(o_saturate(r / d));
  *area = 2.0f * O_PI * angle * angle;
  const float u = sqrtf(r1) * angle, v = 2.0f * O_PI * r2;
  return v_norm(sample_hemisphere_basis(u, v, dir));
}
/* sky_get_sun_color, sky_utils.cuh:318-347; `include_cloud_hdri`: in HDRI mode with active clouds the panorama's fourth channel (the clouds' transmittance,
 * point filter like the colour) dims the sun */
static inline RGBF sky_sun_color_ex(const OSky* s, vec3 origin, vec3 ray, bool include_cloud_hdri) {
  const float height = sky_height(origin);
  const float zenith_cos = v_dot(v_norm(origin), ray);
  const float2_t uv = sky_transmittance_uv(height, zenith_cos);
  const Spectrum extinction_sun = sp_mul(SP_IDENT, sky_lut_fetch(s->tm, SKY_TM_W, SKY_TM_H, uv.x, uv.y));
  RGBF sun_color = sky_color_from_spectrum(sp_mul(extinction_sun, sp_scale(SKY_SUN_RADIANCE, s->sun_strength)));
  const OracleScene* sc = s->scene;
  if (include_cloud_hdri && sc->cloud_active && sc->sky_mode == 1u /* HDRI */ && sc->sky_hdri && sc->sky_hdri_dim) {
    const float theta = o_atan2(ray.z, ray.x), phi = o_asin(ray.y);
    const float u = (theta + REF_PI) / (2.0f * REF_PI);
    const float v = 1.0f - ((phi + 0.5f * REF_PI) / REF_PI);
    const float dim = (float) sc->sky_hdri_dim;
    const uint32_t x = (uint32_t) ((u - floorf(u)) * dim) % sc->sky_hdri_dim, y = (uint32_t) ((v - floorf(v)) * dim) % sc->sky_hdri_dim;
    sun_color = c_scale(sun_color, sc->sky_hdri[4 * ((size_t) x + (size_t) y * sc->sky_hdri_dim) + 3]);
  }
  return sun_color;
}
static inline RGBF sky_sun_color(const OSky* s, vec3 origin, vec3 ray) { return sky_sun_color_ex(s, origin, ray, true); }
/* bsdf_sample_for_sun_pdf<GEOMETRY>, bsdf.cuh:438-458: the world-space V goes into the bounded-VNDF density as it does there */
/* ---- baked panorama as the sky (sky mode HDRI) ----
 * sky_hdri_sample, sky_utils.cuh:49-63: equirectangular lookup, point filter (device_sky.c:352), wrap addressing (texture_create's default),
 * no gamma; the texel of a normalised coordinate u is floor(frac(u) * dim). */
static inline RGBF sky_hdri_sample(const OracleScene* sc, vec3 ray) {
  if (!sc->sky_hdri || !sc->sky_hdri_dim) return c_splat(0.0f);
  const float theta = o_atan2(ray.z, ray.x), phi = o_asin(ray.y);
  const float u = (theta + REF_PI) / (2.0f * REF_PI);
  const float v = 1.0f - ((phi + 0.5f * REF_PI) / REF_PI);
  const float dim = (float) sc->sky_hdri_dim;
  const uint32_t x = (uint32_t) ((u - floorf(u)) * dim) % sc->sky_hdri_dim, y = (uint32_t) ((v - floorf(v)) * dim) % sc->sky_hdri_dim;
  const float* t = sc->sky_hdri + 4 * ((size_t) x + (size_t) y * sc->sky_hdri_dim);
  return c3(t[0], t[1], t[2]);
}
/* sky_color_main / sky_color_no_compute, HDRI branch (sky.cuh:534-606): the sun disk is not part of the panorama */
static inline RGBF sky_hdri_color(const OracleScene* sc, vec3 origin, vec3 ray, uint32_t state) {
  RGBF c = sky_hdri_sample(sc, ray);
  if (state & (ST_CAMERA_DIRECTION | ST_ALLOW_EMISSION)) {
    const OSky s = osky_view(sc);
    const vec3 sky_origin = world_to_sky(&s, origin);
    if (sphere_hit(ray, sky_origin, s.sun_pos, SKY_SUN_RADIUS) && !sph_hit_p0(ray, sky_origin, SKY_EARTH_RADIUS)) c = c_add(c, sky_sun_color(&s, sky_origin, ray));
  }
  return c;
}
static inline float sun_bsdf_pdf(const GeoCtx* g, vec3 L, float reflection_prob, float refraction_prob) {
  const BSDFRayCtx c = bsdf_evaluate_analyze(&g->params, g->normal, g->V, L);
  const float roughness = mp_roughness(&g->params);
  if (c.is_refraction) return refraction_prob * microfacet_refraction_pdf(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, mp_ior(&g->params));
  return reflection_prob * microfacet_pdf(g->V, roughness, c.NdotH, c.NdotV);
}
/* direct_lighting_sun_create_task + direct_lighting_sun_direct */
static bool sun_sample(const OSky* sky, const OLuts* l, const GeoCtx* g, const Sampler* smp, RGBF* light_out, vec3* dir_out) {
  const vec3 sky_pos = world_to_sky(sky, g->position);
  const bool sun_below_horizon = sph_hit_p0(v_norm(v_sub(sky->sun_pos, sky_pos)), sky_pos, SKY_EARTH_RADIUS);
  const bool inside_earth = v_len(sky_pos) < SKY_EARTH_RADIUS;
  if (sun_below_horizon || inside_earth) return false;
  const MatParams* p = &g->params;
  const bool translucent = (p->flags & MAT_SUBSTRATE_MASK) == MAT_TRANSLUCENT;
  const float w_refl = 1.0f, w_refr = translucent ? 1.0f : 0.0f;
  const float reflection_prob = w_refl / (w_refl + w_refr), refraction_prob = w_refr / (w_refl + w_refr);
  const Quat rot = q_rotation_to_z(g->normal);
  const vec3 Vl = q_apply(rot, g->V);
  const float roughness = mp_roughness(p);
  vec3 ray_local;
  if (rnd1(smp, RT_SUN_BSDF_METHOD) < reflection_prob) ray_local = v_reflect(Vl, microfacet_sample_normal(Vl, roughness, rnd2(smp, RT_SUN_BSDF)));
  else { bool tot; ray_local = refract_vector(Vl, microfacet_refraction_sample_normal(Vl, roughness, rnd2(smp, RT_SUN_BSDF)), mp_ior(p), &tot); }
  const vec3 dir_bsdf = v_norm(q_apply(q_inverse(rot), ray_local));
  RGBF light_bsdf = c_splat(0.0f);
  bool is_refraction;
  if (sphere_hit(dir_bsdf, sky_pos, sky->sun_pos, SKY_SUN_RADIUS)) light_bsdf = c_mul(sky_sun_color(sky, sky_pos, dir_bsdf), bsdf_evaluate(l, g, dir_bsdf, HINT_GENERAL, &is_refraction, 1.0f));
  float solid_angle;
  const vec3 dir_sa = sample_sphere(sky->sun_pos, SKY_SUN_RADIUS, sky_pos, rnd2(smp, RT_SUN_RAY), &solid_angle);
  const RGBF light_sa = c_mul(sky_sun_color(sky, sky_pos, dir_sa), bsdf_evaluate(l, g, dir_sa, HINT_GENERAL, &is_refraction, 1.0f));
  const float target_bsdf = c_importance(light_bsdf), target_sa = c_importance(light_sa);
  const float mis_bsdf = solid_angle / (sun_bsdf_pdf(g, dir_bsdf, reflection_prob, refraction_prob) * solid_angle + 1.0f);
  const float mis_sa = solid_angle / (sun_bsdf_pdf(g, dir_sa, reflection_prob, refraction_prob) * solid_angle + 1.0f);
  const float weight_bsdf = target_bsdf * mis_bsdf, weight_sa = target_sa * mis_sa;
  const float sum_weights = weight_bsdf + weight_sa;
  if (sum_weights == 0.0f) return false;
  float target;
  RGBF light;
  if (rnd1(smp, RT_SUN_RESAMPLING) * sum_weights < weight_bsdf) { *dir_out = dir_bsdf; target = target_bsdf; light = light_bsdf; }
  else { *dir_out = dir_sa; target = target_sa; light = light_sa; }
  light = c_scale(light, sum_weights / target);
  if (target == 0.0f) return false;
  if (c_importance(light) == 0.0f) return false;
  *light_out = light;
  return true;
}

#endif
