/*
 * ORACLE (test infrastructure, not product): BSDF evaluation and sampling.
 * Follows /root/reference/src/luminary/device/cuda/bsdf_utils.cuh and cuda/bsdf.cuh:11-301.
 * Bug-compatible where the reference is: the dielectric lobe reads ROUGHNESS as IOR
 * (bsdf_utils.cuh:517), conductor/glossy do the same under the refraction hint (:396, :447) and the
 * DIFFUSE case of the dielectric reflection switch falls through (:547-552).
 *
 * LUT fetch: the reference samples u16-normalised CUDA textures with hardware bilinear/trilinear filtering
 * (bsdf_utils.cuh:379-381, :439-441, :499-507; clamp addressing, device_bsdf.c:38-52). The restatement uses an
 * exact-float software filter (texel centres at (i+0.5)/32) so that it is reproducible; see DESIGN.md.
 */
#ifndef ORACLE_O_BSDF_H
#define ORACLE_O_BSDF_H

#include "o_material.h"
#include "o_rng.h"

#define BSDF_LUT_SIZE 32

typedef struct {
  const uint16_t* conductor;      /* 32*32 */
  const uint16_t* glossy;         /* 32*32 */
  const uint16_t* dielectric;     /* 32*32*32 */
  const uint16_t* dielectric_inv; /* 32*32*32 */
} OLuts;

static inline void lut_axis(float coord, int* i0, int* i1, float* f) {
  const float x  = coord * (float) BSDF_LUT_SIZE - 0.5f;
  const float fl = floorf(x);
  *f             = x - fl;
  int a          = (int) fl;
  int b          = a + 1;
  a = a < 0 ? 0 : (a > BSDF_LUT_SIZE - 1 ? BSDF_LUT_SIZE - 1 : a);
  b = b < 0 ? 0 : (b > BSDF_LUT_SIZE - 1 ? BSDF_LUT_SIZE - 1 : b);
  *i0 = a; *i1 = b;
}
static inline float lut_texel(const uint16_t* t, int idx) { return t[idx] * (1.0f / 65535.0f); }
static inline float lut2d_slice(const uint16_t* t, int x0, int x1, float fx, int y0, int y1, float fy) {
  const float a = lut_texel(t, y0 * BSDF_LUT_SIZE + x0), b = lut_texel(t, y0 * BSDF_LUT_SIZE + x1);
  const float c = lut_texel(t, y1 * BSDF_LUT_SIZE + x0), d = lut_texel(t, y1 * BSDF_LUT_SIZE + x1);
  const float top = a + fx * (b - a), bot = c + fx * (d - c);
  return top + fy * (bot - top);
}
static inline float lut2d(const uint16_t* t, float u, float v) {
  int x0, x1, y0, y1; float fx, fy;
  lut_axis(u, &x0, &x1, &fx); lut_axis(v, &y0, &y1, &fy);
  return lut2d_slice(t, x0, x1, fx, y0, y1, fy);
}
static inline float lut3d(const uint16_t* t, float u, float v, float w) {
  int x0, x1, y0, y1, z0, z1; float fx, fy, fz;
  lut_axis(u, &x0, &x1, &fx); lut_axis(v, &y0, &y1, &fy); lut_axis(w, &z0, &z1, &fz);
  const float lo = lut2d_slice(t + z0 * BSDF_LUT_SIZE * BSDF_LUT_SIZE, x0, x1, fx, y0, y1, fy);
  const float hi = lut2d_slice(t + z1 * BSDF_LUT_SIZE * BSDF_LUT_SIZE, x0, x1, fx, y0, y1, fy);
  return lo + fz * (hi - lo);
}

typedef struct { vec3 V; float fresnel_dielectric, NdotH, NdotL, NdotV, HdotL, HdotV; bool is_refraction; } BSDFRayCtx;
enum { HINT_GENERAL = 0, HINT_MICROFACET = 1, HINT_DIFFUSE = 2, HINT_MICROFACET_REFRACTION = 3 };

/* bsdf_utils.cuh:79-96 */
static inline float bsdf_fresnel(vec3 n, vec3 V, vec3 refr, float ior) {
  const float NdotV = v_dot(V, n), NdotT = -v_dot(refr, n);
  const float s1 = ior * NdotV, s2 = 1.0f * NdotT, p1 = ior * NdotT, p2 = 1.0f * NdotV;
  float rs = (s1 - s2) / (s1 + s2), rp = (p1 - p2) / (p1 + p2);
  rs *= rs; rp *= rp;
  return o_saturate(0.5f * (rs + rp));
}
/* bsdf_utils.cuh:105-118 */
static inline RGBF bsdf_fresnel_schlick(RGBF f0, float f90, float HdotV) {
  const float om = 1.0f - fabsf(HdotV), p2 = om * om, t = p2 * p2 * om;
  const RGBF diff = c_sub(c3(f90, f90, f90), f0);
  return c_add(f0, c_scale(diff, t));
}
/* bsdf_utils.cuh:120-123 */
static inline float bsdf_shadowed_F90(RGBF f0) { return fminf(1.0f, (1.0f / 0.04f) * c_luminance(f0)); }
/* bsdf_utils.cuh:137-144 */
static inline vec3 bsdf_normal_from_pair(vec3 L, vec3 V, float ior) {
  const vec3 n = v_add(L, v_scale(V, ior));
  const float l = v_len(n);
  return (l > 0.0f) ? v_scale(n, 1.0f / l) : V;
}
/* bsdf_utils.cuh:151-176 */
static inline float ggx_G1(float r4, float NdotS) {
  const float n2 = fmaxf(0.0001f, NdotS * NdotS);
  return 2.0f / (sqrtf(((r4 * (1.0f - n2)) + n2) / n2) + 1.0f);
}
static inline float ggx_G2(float r4, float NdotL, float NdotV) {
  const float a = NdotV * sqrtf(r4 + NdotL * (NdotL - r4 * NdotL));
  const float b = NdotL * sqrtf(r4 + NdotV * (NdotV - r4 * NdotV));
  return 0.5f / (a + b);
}
static inline float ggx_G2_over_G1(float r4, float NdotL, float NdotV) {
  const float g1v = ggx_G1(r4, NdotV), g1l = ggx_G1(r4, NdotL);
  return g1l / (g1v + g1l - g1v * g1l);
}
static inline float ggx_D(float NdotH, float r4) {
  const float n2 = fminf(NdotH * NdotH, 1.0f);
  const float a  = 1.0f - n2 + r4 * n2;
  return r4 / (O_PI * a * a);
}
/* bsdf_utils.cuh:185-204: bounded VNDF sampling (Eto & Tokuyoshi 2023) */
static inline vec3 microfacet_sample_normal(vec3 V, float roughness, float2_t rnd) {
  const float r2 = roughness * roughness, r4 = r2 * r2;
  const vec3 v = v_norm(v3(r2 * V.x, r2 * V.y, V.z));
  const float phi = 2.0f * O_PI * rnd.x;
  const float s = 1.0f + sqrtf(V.x * V.x + V.y * V.y), s2 = s * s;
  const float k = (1.0f - r4) * s2 / (s2 + r4 * V.z * V.z);
  const float b = k * v.z;
  const float z = (1.0f - rnd.y) * (1.0f + b) - b;
  const float st = sqrtf(o_saturate(1.0f - z * z));
  float sp, cp; o_sincos(phi, &sp, &cp);
  const vec3 smp = v_add(v3(st * cp, st * sp, z), v);
  return v_norm(v3(smp.x * r2, smp.y * r2, smp.z));
}
static inline float vndf_k_term(vec3 V, float r4, float* t_out) {
  const float len2 = r4 * (V.x * V.x + V.y * V.y);
  *t_out = sqrtf(len2 + V.z * V.z);
  const float s = 1.0f + sqrtf(V.x * V.x + V.y * V.y), s2 = s * s;
  return (1.0f - r4) * s2 / (s2 + r4 * V.z * V.z);
}
/* bsdf_utils.cuh:206-221 */
static inline float microfacet_pdf(vec3 V, float roughness, float NdotH, float NdotV) {
  const float r2 = roughness * roughness, r4 = r2 * r2;
  const float D = ggx_D(NdotH, r4);
  float t; const float k = vndf_k_term(V, r4, &t);
  return D / (2.0f * (k * NdotV + t));
}
/* bsdf_utils.cuh:228-239 */
static inline float microfacet_eval(float roughness, float NdotH, float NdotL, float NdotV) {
  const float r2 = roughness * roughness, r4 = r2 * r2;
  return ggx_D(NdotH, r4) * ggx_G2(r4, NdotL, NdotV) * NdotL;
}
/* bsdf_utils.cuh:241-259 */
static inline float microfacet_eval_sampled_microfacet(vec3 V, float roughness, float NdotL, float NdotV) {
  const float r2 = roughness * roughness, r4 = r2 * r2;
  const float G2 = ggx_G2(r4, NdotL, NdotV);
  float t; const float k = vndf_k_term(V, r4, &t);
  return 2.0f * (k * NdotV + t) * G2 * NdotL;
}
/* bsdf_utils.cuh:261-272 */
static inline float microfacet_eval_sampled_diffuse(float roughness, float NdotH, float NdotL, float NdotV) {
  const float r2 = roughness * roughness, r4 = r2 * r2;
  return ggx_D(NdotH, r4) * ggx_G2(r4, NdotL, NdotV) * O_PI;
}
/* bsdf_utils.cuh:279-292: spherical-cap VNDF (Dupuy & Benyoub 2023) */
static inline vec3 microfacet_refraction_sample_normal(vec3 V, float roughness, float2_t rnd) {
  const float r2 = roughness * roughness;
  const vec3 v = v_norm(v3(r2 * V.x, r2 * V.y, V.z));
  const float phi = 2.0f * O_PI * rnd.x;
  const float z = (1.0f - rnd.y) * (1.0f + v.z) - v.z;
  const float st = sqrtf(o_saturate(1.0f - z * z));
  float sp, cp; o_sincos(phi, &sp, &cp);
  const vec3 smp = v_add(v3(st * cp, st * sp, z), v);
  return v_norm(v3(smp.x * r2, smp.y * r2, smp.z));
}
/* bsdf_utils.cuh:294-309 */
static inline float microfacet_refraction_pdf(float roughness, float NdotH, float NdotV, float HdotV, float HdotL, float ior) {
  const float r2 = roughness * roughness, r4 = r2 * r2;
  const float D = ggx_D(NdotH, r4), G1 = ggx_G1(r4, NdotV);
  float den = ior * HdotV + HdotL;
  den = den * den;
  return D * G1 * (HdotV / NdotV) * (HdotL / den);
}
/* bsdf_utils.cuh:317-333 */
static inline float microfacet_refraction_eval(float roughness, float HdotL, float HdotV, float NdotH, float NdotL, float NdotV, float ior) {
  const float r2 = roughness * roughness, r4 = r2 * r2;
  const float D = ggx_D(NdotH, r4), G2 = ggx_G2(r4, NdotL, NdotV);
  float den = ior * HdotV + HdotL;
  den = den * den;
  return 4.0f * NdotL * HdotV * HdotL * D * G2 / den;
}
/* bsdf_utils.cuh:352-373 */
static inline float diffuse_pdf(float NdotL) { return o_saturate(NdotL) * (1.0f / O_PI); }
static inline float diffuse_eval_sampled_microfacet(vec3 V, float roughness, float NdotL, float NdotH, float NdotV) {
  const float r2 = roughness * roughness, r4 = r2 * r2;
  const float D = ggx_D(NdotH, r4);
  float t; const float k = vndf_k_term(V, r4, &t);
  return NdotL * (2.0f * (k * NdotV + t)) / (O_PI * D);
}

/* Single-scatter term shared by conductor and glossy (bsdf_utils.cuh:398-414, 449-465). */
static inline float bsdf_ss_term(const BSDFRayCtx* c, int hint, float roughness, float ior_bug, float inv_pdf) {
  switch (hint) {
    case HINT_GENERAL: return microfacet_eval(roughness, c->NdotH, c->NdotL, c->NdotV) * inv_pdf;
    case HINT_MICROFACET: return microfacet_eval_sampled_microfacet(c->V, roughness, c->NdotL, c->NdotV);
    case HINT_DIFFUSE: return microfacet_eval_sampled_diffuse(roughness, c->NdotH, c->NdotL, c->NdotV);
    default:
      return microfacet_eval(roughness, c->NdotH, c->NdotL, c->NdotV)
             / microfacet_refraction_pdf(roughness, c->NdotH, c->NdotV, c->HdotV, c->HdotL, ior_bug);
  }
}
/* bsdf_utils.cuh:383-427 */
static inline RGBF bsdf_conductor(const OLuts* l, const MatParams* p, const BSDFRayCtx* c, int hint, float inv_pdf) {
  if (c->NdotL <= 0.0f || c->NdotV <= 0.0f) return c_splat(0.0f);
  if ((p->flags & MAT_SUBSTRATE_MASK) != 0) return c_splat(0.0f);
  if ((p->flags & MAT_METALLIC) == 0) return c_splat(0.0f);
  const float roughness = mp_roughness(p);
  const float ior = (hint == HINT_MICROFACET_REFRACTION) ? roughness : 1.0f;
  const float ss = bsdf_ss_term(c, hint, roughness, ior, inv_pdf);
  const RGBF albedo = mp_albedo(p);
  const float da = lut2d(l->conductor, c->NdotV, roughness);
  const RGBF fres = bsdf_fresnel_schlick(albedo, bsdf_shadowed_F90(albedo), c->HdotV);
  const RGBF ssf = c_scale(fres, ss);
  const RGBF msf = c_mul(albedo, c_scale(fres, ((1.0f / da) - 1.0f) * ss));
  return c_add(ssf, msf);
}
/* bsdf_utils.cuh:433-497 */
static inline RGBF bsdf_glossy(const OLuts* l, const MatParams* p, const BSDFRayCtx* c, int hint, float inv_pdf) {
  if (c->NdotL <= 0.0f || c->NdotV <= 0.0f) return c_splat(0.0f);
  if ((p->flags & MAT_SUBSTRATE_MASK) != 0) return c_splat(0.0f);
  if ((p->flags & MAT_METALLIC) != 0) return c_splat(0.0f);
  const float roughness = mp_roughness(p);
  const float ior = (hint == HINT_MICROFACET_REFRACTION) ? roughness : 1.0f;
  const float ss = bsdf_ss_term(c, hint, roughness, ior, inv_pdf);
  float diff;
  switch (hint) {
    case HINT_GENERAL: diff = diffuse_pdf(c->NdotL) * inv_pdf; break;
    case HINT_DIFFUSE: diff = 1.0f; break;
    case HINT_MICROFACET: diff = diffuse_eval_sampled_microfacet(c->V, roughness, c->NdotL, c->NdotH, c->NdotV); break;
    default: diff = diffuse_pdf(c->NdotL) / microfacet_refraction_pdf(roughness, c->NdotH, c->NdotV, c->HdotV, c->HdotL, ior); break;
  }
  const RGBF albedo = mp_albedo(p);
  const float cda = lut2d(l->conductor, c->NdotV, roughness), gda = lut2d(l->glossy, c->NdotV, roughness);
  const RGBF f0 = c3(0.04f, 0.04f, 0.04f);
  const RGBF fres = bsdf_fresnel_schlick(f0, bsdf_shadowed_F90(f0), c->HdotV);
  return c_add(c_scale(fres, ss / cda), c_scale(albedo, diff * (1.0f - gda)));
}
/* bsdf_utils.cuh:499-507 */
static inline float bsdf_dielectric_da(const OLuts* l, float NdotV, float roughness, float ior) {
  const bool use_inv = (ior > 1.0f);
  const float w = use_inv ? (ior - 1.0f) * 0.5f : (1.0f / ior - 1.0f) * 0.5f;
  return lut3d(use_inv ? l->dielectric_inv : l->dielectric, NdotV, roughness, w);
}
/* bsdf_utils.cuh:509-568 */
static inline RGBF bsdf_dielectric(const OLuts* l, const MatParams* p, const BSDFRayCtx* c, int hint, float inv_pdf) {
  if (c->NdotL <= 0.0f || c->NdotV <= 0.0f) return c_splat(0.0f);
  if ((p->flags & MAT_SUBSTRATE_MASK) != MAT_TRANSLUCENT) return c_splat(0.0f);
  const float ior = mp_roughness(p); /* sic, :517 */
  const float roughness = mp_roughness(p);
  float term;
  if (c->is_refraction) {
    switch (hint) {
      case HINT_GENERAL: term = microfacet_refraction_eval(roughness, c->HdotL, c->HdotV, c->NdotH, c->NdotL, c->NdotV, ior) * inv_pdf; break;
      case HINT_MICROFACET_REFRACTION: { const float r2 = roughness * roughness; term = ggx_G2_over_G1(r2 * r2, c->NdotL, c->NdotV); } break;
      default: term = 0.0f; break;
    }
    term *= (1.0f - c->fresnel_dielectric);
  }
  else {
    switch (hint) {
      case HINT_GENERAL: term = microfacet_eval(roughness, c->NdotH, c->NdotL, c->NdotV) * inv_pdf; break;
      case HINT_MICROFACET: term = microfacet_eval_sampled_microfacet(c->V, roughness, c->NdotL, c->NdotV); break;
      default: /* DIFFUSE falls through into MICROFACET_REFRACTION, :547-552 */
        term = microfacet_eval(roughness, c->NdotH, c->NdotL, c->NdotV)
               / microfacet_refraction_pdf(roughness, c->NdotH, c->NdotV, c->HdotV, c->HdotL, ior);
        break;
    }
    term *= c->fresnel_dielectric;
  }
  const RGBF albedo = mp_albedo(p);
  term /= bsdf_dielectric_da(l, c->NdotV, roughness, ior);
  if (ior == 1.0f && c->is_refraction) term = (hint == HINT_MICROFACET_REFRACTION) ? 1.0f : 0.0f;
  return c_scale(albedo, term);
}
/* bsdf_utils.cuh:574-587 */
static inline RGBF bsdf_multiscattering_evaluate(const OLuts* l, const MatParams* p, const BSDFRayCtx* c, int hint, float inv_pdf) {
  const float opacity = mp_opacity(p);
  if (c->is_refraction) return c_scale(bsdf_dielectric(l, p, c, hint, inv_pdf), opacity);
  const RGBF a = bsdf_conductor(l, p, c, hint, inv_pdf), b = bsdf_glossy(l, p, c, hint, inv_pdf), d = bsdf_dielectric(l, p, c, hint, inv_pdf);
  return c_scale(c_add(c_add(a, b), d), opacity);
}

/* bsdf.cuh:11-50 */
static inline BSDFRayCtx bsdf_evaluate_analyze(const MatParams* p, vec3 normal, vec3 V, vec3 L) {
  BSDFRayCtx c;
  c.NdotL = v_dot(normal, L);
  c.NdotV = o_saturate(v_dot(normal, V));
  c.is_refraction = (c.NdotL < 0.0f);
  c.NdotL = c.is_refraction ? -c.NdotL : c.NdotL;
  const float ior = mp_ior(p);
  vec3 refr, H;
  bool total_reflection;
  if (c.is_refraction) { total_reflection = false; H = bsdf_normal_from_pair(L, V, ior); refr = L; }
  else { H = bsdf_normal_from_pair(L, V, 1.0f); refr = refract_vector(V, H, ior, &total_reflection); }
  c.HdotV = fabsf(v_dot(H, V));
  c.HdotL = fabsf(v_dot(H, L));
  c.NdotH = v_dot(normal, H);
  if (c.NdotH < 0.0f) { H = v_scale(H, -1.0f); c.NdotH = -c.NdotH; }
  c.fresnel_dielectric = total_reflection ? 1.0f : bsdf_fresnel(H, V, refr, ior);
  c.V = V;
  return c;
}
/* bsdf.cuh:52-64 */
static inline RGBF bsdf_evaluate_core(const OLuts* l, const MatParams* p, const BSDFRayCtx* c, int hint, vec3 L, vec3 face_normal, float inv_pdf) {
  const float fl = v_dot(face_normal, L);
  const float flip = c->is_refraction ? -1.0f : 1.0f;
  if (fl * flip < O_EPS) return c_splat(0.0f);
  return bsdf_multiscattering_evaluate(l, p, c, hint, inv_pdf);
}
/* bsdf.cuh:103-133 */
static inline BSDFRayCtx bsdf_sample_context(const MatParams* p, vec3 normal, vec3 V, vec3 H, vec3 L, bool is_refraction) {
  BSDFRayCtx c;
  c.NdotL = v_dot(normal, L);
  c.NdotV = o_saturate(v_dot(normal, V));
  c.is_refraction = is_refraction;
  c.NdotL = is_refraction ? -c.NdotL : c.NdotL;
  const float ior = mp_ior(p);
  bool total_reflection = false;
  const vec3 refr = is_refraction ? L : refract_vector(V, H, ior, &total_reflection);
  c.HdotV = fabsf(v_dot(H, V));
  c.HdotL = fabsf(v_dot(H, L));
  c.NdotH = v_dot(normal, H);
  float flipH = 1.0f;
  if (c.NdotH < 0.0f) { flipH = -1.0f; c.NdotH = -c.NdotH; }
  c.fresnel_dielectric = total_reflection ? 1.0f : bsdf_fresnel(v_scale(H, flipH), V, refr, ior);
  c.V = V;
  return c;
}

/* Shading context of a triangle hit (material.cuh:65-80). */
typedef struct {
  uint32_t instance_id, tri_id;
  vec3 position, V, normal;
  uint32_t face_normal; /* packed, object space (geometry_utils.cuh:205) */
  uint16_t state;
  MatParams params;
} GeoCtx;

/* bsdf.cuh:73-83 */
static inline RGBF bsdf_evaluate(const OLuts* l, const GeoCtx* g, vec3 L, int hint, bool* is_refraction, float inv_pdf) {
  const BSDFRayCtx c = bsdf_evaluate_analyze(&g->params, g->normal, g->V, L);
  *is_refraction = c.is_refraction;
  return bsdf_evaluate_core(l, &g->params, &c, hint, L, normal_unpack(g->face_normal), inv_pdf);
}

typedef struct { vec3 ray; RGBF weight; bool is_transparent_pass, is_microfacet_based; } BSDFSample;

/* bsdf.cuh:138-301; `set` selects the RandomSet::BSDF<set> targets (random.cuh:120-129). */
static inline BSDFSample bsdf_sample(const OLuts* l, const GeoCtx* g, const Sampler* smp, uint32_t set) {
  const MatParams* p = &g->params;
  BSDFSample info;
  const float opacity = mp_opacity(p);
  if (opacity < 1.0f) {
    const float tr = rnd1(smp, RT_BSDF_OPACITY + set);
    if (tr > opacity) {
      info.ray = v_scale(g->V, -1.0f);
      info.weight = (p->flags & MAT_COLORED_TRANSPARENCY) ? mp_albedo(p) : c3(1.0f, 1.0f, 1.0f);
      info.is_microfacet_based = false;
      info.is_transparent_pass = true;
      return info;
    }
  }
  const Quat rot = q_rotation_to_z(g->normal);
  const vec3 Vl = q_apply(rot, g->V);
  const vec3 fnl = q_apply(rot, normal_unpack(g->face_normal));
  const vec3 up = v3(0.0f, 0.0f, 1.0f);
  info.is_transparent_pass = false;
  info.is_microfacet_based = false;
  vec3 ray_local;
  const uint32_t substrate = p->flags & MAT_SUBSTRATE_MASK;
  const bool include_diffuse = (substrate == 0) && ((p->flags & MAT_METALLIC) == 0);
  const bool include_refraction = (substrate == MAT_TRANSLUCENT);
  float sum_weights = 0.0f;
  RGBF selected = c3(0.0f, 0.0f, 0.0f);
  float rr = rnd1(smp, RT_BSDF_RESAMPLING + set);
  const float ior = mp_ior(p);
  const float roughness = mp_roughness(p);
  {
    const vec3 m = microfacet_sample_normal(Vl, roughness, rnd2(smp, RT_BSDF_REFLECTION + set));
    const vec3 ray = v_reflect(Vl, m);
    const BSDFRayCtx c = bsdf_sample_context(p, up, Vl, m, ray, false);
    const RGBF eval = bsdf_evaluate_core(l, p, &c, HINT_MICROFACET, ray, fnl, 1.0f);
    const float pdf = microfacet_pdf(Vl, roughness, c.NdotH, c.NdotV);
    const float dpdf = include_diffuse ? diffuse_pdf(c.NdotL) : 0.0f;
    const float rpdf = include_refraction ? microfacet_refraction_pdf(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior) : 0.0f;
    const float sum = pdf + dpdf + rpdf;
    const float mis = (sum > 0.0f) ? pdf / sum : 0.0f;
    const float w = c_importance(eval) * mis;
    ray_local = ray; sum_weights = w; selected = eval;
    info.is_transparent_pass = false; info.is_microfacet_based = true;
  }
  if (include_diffuse) {
    const float2_t r2 = rnd2(smp, RT_BSDF_DIFFUSE + set);
    const vec3 ray = sample_ray_sphere(r2.x, r2.y);
    const vec3 m = v_norm(v_add(Vl, ray));
    const BSDFRayCtx c = bsdf_sample_context(p, up, Vl, m, ray, false);
    const RGBF eval = bsdf_evaluate_core(l, p, &c, HINT_DIFFUSE, ray, fnl, 1.0f);
    const float pdf = diffuse_pdf(c.NdotL);
    const float mpdf = microfacet_pdf(Vl, roughness, c.NdotH, c.NdotV);
    const float rpdf = include_refraction ? microfacet_refraction_pdf(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior) : 0.0f;
    const float sum = pdf + mpdf + rpdf;
    const float mis = (sum > 0.0f) ? pdf / sum : 0.0f;
    const float w = c_importance(eval) * mis;
    sum_weights += w;
    const float prob = w / sum_weights;
    if (rr < prob) {
      ray_local = ray; selected = eval;
      info.is_transparent_pass = false; info.is_microfacet_based = false;
      rr = rng_saturate(rr / prob);
    }
    else rr = rng_saturate((rr - prob) / (1.0f - prob));
  }
  if (include_refraction) {
    bool total_reflection;
    const vec3 m = microfacet_refraction_sample_normal(Vl, roughness, rnd2(smp, RT_BSDF_REFRACTION + set));
    const vec3 ray = refract_vector(Vl, m, ior, &total_reflection);
    const BSDFRayCtx c = bsdf_sample_context(p, up, Vl, m, ray, !total_reflection);
    const RGBF eval = bsdf_evaluate_core(l, p, &c, HINT_MICROFACET_REFRACTION, ray, fnl, 1.0f);
    float mis = 1.0f;
    if (total_reflection) {
      const float pdf = microfacet_refraction_pdf(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior);
      const float refl = microfacet_pdf(Vl, roughness, c.NdotH, c.NdotV);
      const float dpdf = include_diffuse ? diffuse_pdf(c.NdotL) : 0.0f;
      const float sum = pdf + refl + dpdf;
      mis = (sum > 0.0f) ? pdf / sum : 0.0f;
    }
    const float w = c_importance(eval) * mis;
    sum_weights += w;
    const float prob = w / sum_weights;
    if (rr < prob) {
      ray_local = ray; selected = eval;
      info.is_transparent_pass = !total_reflection; info.is_microfacet_based = true;
      rr = rng_saturate(rr / prob);
    }
    else rr = rng_saturate((rr - prob) / (1.0f - prob));
  }
  info.weight = (sum_weights > 0.0f) ? c_scale(selected, sum_weights / c_importance(selected)) : c3(0.0f, 0.0f, 0.0f);
  info.ray = v_norm(q_apply(q_inverse(rot), ray_local));
  return info;
}

/* bsdf_utils.cuh:68-77 */
static inline bool bsdf_is_pass_through_ray(const GeoCtx* g, const BSDFSample* s) {
  const float ior = mp_ior(&g->params);
  return s->is_transparent_pass && ((ior == 1.0f) || (s->is_microfacet_based == false));
}

#endif
