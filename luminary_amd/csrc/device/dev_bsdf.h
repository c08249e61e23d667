// Material decode, per-vertex parameter quantisation and the layered BSDF.
// Reference: cuda/memory.cuh:442-474, cuda/material.cuh:36-325, cuda/bsdf_utils.cuh, cuda/bsdf.cuh:11-301.
// The reference's known quirks are kept on purpose (they change the image): the dielectric lobe reads the roughness
// parameter as IOR (bsdf_utils.cuh:517), conductor/glossy do so under the refraction hint (:396, :447), and the DIFFUSE
// case of the dielectric reflection switch falls through (:547-552).
// Energy LUTs are fetched with an exact-float bilinear/trilinear filter (texel centres at (i+0.5)/32, clamp addressing)
// instead of the texture unit, which keeps results reproducible.
#pragma once

#include "dev_sampler.h"
#include "dev_scene.h"

LUM_NS_BEGIN

enum DevMatFlag : uint32_t {  // device_structs.h:186-200
  kDMatSubstrateMask = 0x01, kDMatEmission = 0x02, kDMatMetallic = 0x08, kDMatColoredTransparency = 0x10,
  kDMatRoughnessAsSmoothness = 0x20, kDMatNormalMapCompressed = 0x40, kDMatBidirectionalEmission = 0x80
};
enum MatFlag : uint32_t { kMatTranslucent = 1, kMatSubstrateMask = 1, kMatRefractionInside = 2, kMatMetallic = 4, kMatColoredTransparency = 8 };
constexpr uint32_t kTextureNone = 0xFFFFu;

struct Material {
  uint32_t flags;
  float roughness_clamp, roughness, refraction_index;
  Col albedo; float alpha;
  Col emission;
  float emission_scale;  // as stored: the material's scale over the emission normalisation (device_structs.c:289-303); scales textured emission
  uint32_t metallic_tex, albedo_tex, luminance_tex, roughness_tex, normal_tex;
};

LUM_DEV Material load_material(const DeviceScene& sc, uint32_t id) {
  const uint4 a = sc.materials[2 * id], b = sc.materials[2 * id + 1];
  Material m;
  m.flags            = a.x & 0xFFu;
  m.roughness_clamp  = unorm16(a.x & 0xFF00u);
  m.metallic_tex     = a.x >> 16;
  m.roughness        = unorm16(a.y);
  m.refraction_index = unorm16(a.y >> 16) * 2.0f + 1.0f;
  m.albedo           = col(unorm16(a.z), unorm16(a.z >> 16), unorm16(a.w));
  m.alpha            = unorm16(a.w >> 16);
  const float scale  = bitsf((b.y >> 16) << 15);
  m.emission         = col(unorm16(b.x), unorm16(b.x >> 16), unorm16(b.y)) * scale;
  m.emission_scale   = scale;
  m.albedo_tex = b.z & 0xFFFFu; m.luminance_tex = b.z >> 16; m.roughness_tex = b.w & 0xFFFFu; m.normal_tex = b.w >> 16;
  return m;
}

// ---- textures (cuda/texture_utils.cuh:20-45; objects are created normalised, wrap-addressed, linearly filtered, mip level 0:
// texture.c:77-90, device_texture.c:247-268). The texture unit's 8-bit interpolation weights are replaced by exact float lerps, the
// gamma powf by pow_det, so that a fetch is reproducible. `def` is returned for an invalid handle. ----
LUM_DEV float4 unpack_texel(uint32_t t) {
  return make_float4((t & 0xFFu) * (1.0f / 255.0f), ((t >> 8) & 0xFFu) * (1.0f / 255.0f), ((t >> 16) & 0xFFu) * (1.0f / 255.0f), (t >> 24) * (1.0f / 255.0f));
}
LUM_DEV float4 texture_load(const DeviceScene& sc, uint32_t tex, F2 uv, bool apply_gamma, float4 def) {
  if (tex >= sc.num_textures) return def;
  const uint4 t = sc.texture_table[tex];
  const int w = (int) t.y, h = (int) t.z;
  const float u = uv.x, v = 1.0f - uv.y;  // flip_v
  const float xb = (u - floorf(u)) * (float) w - 0.5f, yb = (v - floorf(v)) * (float) h - 0.5f;
  const float xf = floorf(xb), yf = floorf(yb);
  const float ax = xb - xf, ay = yb - yf;
  int x0 = (int) xf, y0 = (int) yf, x1 = x0 + 1, y1 = y0 + 1;
  if (x0 < 0) x0 += w;
  if (y0 < 0) y0 += h;
  if (x1 >= w) x1 -= w;
  if (y1 >= h) y1 -= h;
  const uint32_t* __restrict__ base = sc.texels + t.x;
  const float4 c00 = unpack_texel(base[x0 + y0 * w]), c10 = unpack_texel(base[x1 + y0 * w]);
  const float4 c01 = unpack_texel(base[x0 + y1 * w]), c11 = unpack_texel(base[x1 + y1 * w]);
  float4 r;
  {
    const float top = c00.x + ax * (c10.x - c00.x), bot = c01.x + ax * (c11.x - c01.x);
    r.x = top + ay * (bot - top);
  }
  {
    const float top = c00.y + ax * (c10.y - c00.y), bot = c01.y + ax * (c11.y - c01.y);
    r.y = top + ay * (bot - top);
  }
  {
    const float top = c00.z + ax * (c10.z - c00.z), bot = c01.z + ax * (c11.z - c01.z);
    r.z = top + ay * (bot - top);
  }
  {
    const float top = c00.w + ax * (c10.w - c00.w), bot = c01.w + ax * (c11.w - c01.w);
    r.w = top + ay * (bot - top);
  }
  const float gamma = bitsf(t.w);
  if (apply_gamma && gamma != 1.0f) { r.x = pow_det(r.x, gamma); r.y = pow_det(r.y, gamma); r.z = pow_det(r.z, gamma); }  // never the alpha
  return r;
}
// cuda/math.cuh:246-253, :1706-1713 and cuda/memory.cuh:414-425
LUM_DEV F2 uv_unpack(uint32_t d) { return F2{bitsf(d & 0xFFFF0000u), bitsf(d << 16)}; }
LUM_DEV F2 triangle_uv(uint4 tri_tex, F2 coords) {
  const F2 a = uv_unpack(tri_tex.x), b = uv_unpack(tri_tex.y), c = uv_unpack(tri_tex.z);
  return F2{a.x + coords.x * (b.x - a.x) + coords.y * (c.x - a.x), a.y + coords.x * (b.y - a.y) + coords.y * (c.y - a.y)};
}

// material.cuh:36-53: EMISSION bits 0..31 | ALBEDO 32..61 | OPACITY 62..69 | ROUGHNESS 70..79 | IOR 80..87
struct MatParams {
  uint32_t d0, d1, d2, flags;

  LUM_DEV float opacity() const { return (((d1 >> 30) | ((d2 & 0x3Fu) << 2)) & 0xFFu) * (1.0f / 255); }
  LUM_DEV float roughness() const { return ((d2 >> 6) & 0x3FFu) * (1.0f / 1023); }
  LUM_DEV float ior() const { return ((d2 >> 16) & 0xFFu) * (1.0f / 255) * 3.0f; }
  LUM_DEV Col albedo() const {
    const uint32_t d = d1 & 0x3FFFFFFFu;
    return col((d & 0x3FF) * (1.0f / 0x3FF), ((d >> 10) & 0x3FF) * (1.0f / 0x3FF), (d >> 20) * (1.0f / 0x3FF));
  }
  LUM_DEV Col emission() const {
    const uint32_t dmax = d0 & 0x3FFF, dlo = (d0 >> 14) & 0xFF, dhi = (d0 >> 22) & 0xFF, comp = d0 >> 30;
    const float mx = (dmax > 0) ? bitsf((dmax << 14) | 0x30000000u) * (1023.0f / 2.0f) : 0.0f;
    const float lo = dlo * (1.0f / 0xFF) * mx, hi = dhi * (1.0f / 0xFF) * mx;
    return (comp == 0) ? col(mx, lo, hi) : (comp == 1) ? col(lo, mx, hi) : col(lo, hi, mx);
  }
  LUM_DEV void set(Col albedo, float opacity, float roughness, Col emission, float ior) {
    const uint32_t ar = (uint32_t) (saturate(albedo.r) * 0x3FF + 0.5f), ag = (uint32_t) (saturate(albedo.g) * 0x3FF + 0.5f),
                   ab = (uint32_t) (saturate(albedo.b) * 0x3FF + 0.5f);
    const uint32_t op = (uint32_t) (saturate(opacity) * 255 + 0.5f);
    const uint32_t ro = (uint32_t) (saturate(roughness) * 1023 + 0.5f);
    const uint32_t io = (uint32_t) (saturate(ior * (1.0f / 3.0f)) * 255 + 0.5f);
    uint32_t comp;
    float mx, lo, hi;
    if (emission.r > emission.g && emission.r > emission.b) { comp = 0; mx = emission.r; lo = emission.g; hi = emission.b; }
    else if (emission.g > emission.b) { comp = 1; mx = emission.g; lo = emission.r; hi = emission.b; }
    else { comp = 2; mx = emission.b; lo = emission.r; hi = emission.g; }
    mx = saturate(mx * (1.0f / 1023.0f)) * 2.0f;
    lo = saturate(lo * (2.0f / 1023.0f) * (1.0f / mx));
    hi = saturate(hi * (2.0f / 1023.0f) * (1.0f / mx));
    const uint32_t dmax = (fbits(mx) >= 0x30000000u) ? (fbits(mx) >> 14) & 0x3FFF : 0;
    d0 = dmax | (((uint32_t) (lo * 0xFF + 0.5f)) << 14) | (((uint32_t) (hi * 0xFF + 0.5f)) << 22) | (comp << 30);
    d1 = (ar | (ag << 10) | (ab << 20)) | (op << 30);
    d2 = (op >> 2) | (ro << 6) | (io << 16);
  }
};

struct GeoContext {  // material.cuh:65-80
  uint32_t instance_id, tri_id;
  V3 position, V, normal;
  uint32_t face_normal_packed;  // object space, geometry_utils.cuh:205
  uint32_t state;
  MatParams params;
};

// ---- LUT filter ----
struct LutAxis { int i0, i1; float f; };
LUM_DEV LutAxis lut_axis(float coord) {
  const float x  = coord * 32.0f - 0.5f;
  const float fl = floorf(x);
  LutAxis a;
  a.f  = x - fl;
  a.i0 = min(max((int) fl, 0), 31);
  a.i1 = min(max((int) fl + 1, 0), 31);
  return a;
}
LUM_DEV float lut_texel(const uint16_t* t, int idx) { return t[idx] * (1.0f / 65535.0f); }
LUM_DEV float lut_slice(const uint16_t* t, LutAxis x, LutAxis y) {
  const float a = lut_texel(t, y.i0 * 32 + x.i0), b = lut_texel(t, y.i0 * 32 + x.i1);
  const float c = lut_texel(t, y.i1 * 32 + x.i0), d = lut_texel(t, y.i1 * 32 + x.i1);
  const float top = a + x.f * (b - a), bot = c + x.f * (d - c);
  return top + y.f * (bot - top);
}
LUM_DEV float lut2d(const uint16_t* t, float u, float v) { return lut_slice(t, lut_axis(u), lut_axis(v)); }
LUM_DEV float lut3d(const uint16_t* t, float u, float v, float w) {
  const LutAxis x = lut_axis(u), y = lut_axis(v), z = lut_axis(w);
  const float lo = lut_slice(t + z.i0 * 1024, x, y), hi = lut_slice(t + z.i1 * 1024, x, y);
  return lo + z.f * (hi - lo);
}

// ---- microfacet terms (bsdf_utils.cuh:79-373) ----
struct RayTerms { V3 V; float fresnel_dielectric, NdotH, NdotL, NdotV, HdotL, HdotV; bool is_refraction; };
enum SamplingHint : int { kHintGeneral = 0, kHintMicrofacet = 1, kHintDiffuse = 2, kHintRefraction = 3 };

LUM_DEV float fresnel_dielectric(V3 n, V3 V, V3 refr, float ior) {
  const float NdotV = dot(V, n), NdotT = -dot(refr, n);
  const float s1 = ior * NdotV, s2 = 1.0f * NdotT, p1 = ior * NdotT, p2 = 1.0f * NdotV;
  float rs = (s1 - s2) / (s1 + s2), rp = (p1 - p2) / (p1 + p2);
  rs *= rs; rp *= rp;
  return saturate(0.5f * (rs + rp));
}
LUM_DEV Col fresnel_schlick(Col f0, float f90, float HdotV) {
  const float om = 1.0f - fabsf(HdotV), p2 = om * om, t = p2 * p2 * om;
  return f0 + (col(f90, f90, f90) - f0) * t;
}
LUM_DEV float shadowed_f90(Col f0) { return fminf(1.0f, (1.0f / 0.04f) * luminance(f0)); }
LUM_DEV V3 half_vector(V3 L, V3 V, float ior) {
  const V3 n = L + V * ior;
  const float l = length(n);
  return (l > 0.0f) ? n * (1.0f / l) : V;
}
LUM_DEV float ggx_g1(float r4, float NdotS) {
  const float n2 = fmaxf(0.0001f, NdotS * NdotS);
  return 2.0f / (sqrtf(((r4 * (1.0f - n2)) + n2) / n2) + 1.0f);
}
LUM_DEV float ggx_g2(float r4, float NdotL, float NdotV) {
  const float a = NdotV * sqrtf(r4 + NdotL * (NdotL - r4 * NdotL));
  const float b = NdotL * sqrtf(r4 + NdotV * (NdotV - r4 * NdotV));
  return 0.5f / (a + b);
}
LUM_DEV float ggx_g2_over_g1(float r4, float NdotL, float NdotV) {
  const float g1v = ggx_g1(r4, NdotV), g1l = ggx_g1(r4, NdotL);
  return g1l / (g1v + g1l - g1v * g1l);
}
LUM_DEV float ggx_d(float NdotH, float r4) {
  const float n2 = fminf(NdotH * NdotH, 1.0f);
  const float a  = 1.0f - n2 + r4 * n2;
  return r4 / (kPi * a * a);
}
LUM_DEV float pow4(float r) { const float r2 = r * r; return r2 * r2; }
// bounded VNDF (bsdf_utils.cuh:185-204)
LUM_DEV V3 sample_vndf_bounded(V3 V, float roughness, F2 rnd) {
  const float r2 = roughness * roughness, r4 = r2 * r2;
  const V3 v = normalize(v3(r2 * V.x, r2 * V.y, V.z));
  const float phi = 2.0f * kPi * rnd.x;
  const float s = 1.0f + sqrtf(V.x * V.x + V.y * V.y), s2 = s * s;
  const float k = (1.0f - r4) * s2 / (s2 + r4 * V.z * V.z);
  const float b = k * v.z;
  const float z = (1.0f - rnd.y) * (1.0f + b) - b;
  const float st = sqrtf(saturate(1.0f - z * z));
  float sp, cp; sincos_det(phi, sp, cp);
  const V3 m = v3(st * cp, st * sp, z) + v;
  return normalize(v3(m.x * r2, m.y * r2, m.z));
}
LUM_DEV float vndf_k(V3 V, float r4, float& t) {
  const float len2 = r4 * (V.x * V.x + V.y * V.y);
  t = sqrtf(len2 + V.z * V.z);
  const float s = 1.0f + sqrtf(V.x * V.x + V.y * V.y), s2 = s * s;
  return (1.0f - r4) * s2 / (s2 + r4 * V.z * V.z);
}
LUM_DEV float pdf_vndf_bounded(V3 V, float roughness, float NdotH, float NdotV) {
  const float r4 = pow4(roughness);
  const float D = ggx_d(NdotH, r4);
  float t; const float k = vndf_k(V, r4, t);
  return D / (2.0f * (k * NdotV + t));
}
LUM_DEV float eval_microfacet(float roughness, float NdotH, float NdotL, float NdotV) {
  const float r4 = pow4(roughness);
  return ggx_d(NdotH, r4) * ggx_g2(r4, NdotL, NdotV) * NdotL;
}
LUM_DEV float eval_microfacet_over_vndf(V3 V, float roughness, float NdotL, float NdotV) {
  const float r4 = pow4(roughness);
  const float G2 = ggx_g2(r4, NdotL, NdotV);
  float t; const float k = vndf_k(V, r4, t);
  return 2.0f * (k * NdotV + t) * G2 * NdotL;
}
LUM_DEV float eval_microfacet_over_diffuse(float roughness, float NdotH, float NdotL, float NdotV) {
  const float r4 = pow4(roughness);
  return ggx_d(NdotH, r4) * ggx_g2(r4, NdotL, NdotV) * kPi;
}
// spherical-cap VNDF (bsdf_utils.cuh:279-292)
LUM_DEV V3 sample_vndf_caps(V3 V, float roughness, F2 rnd) {
  const float r2 = roughness * roughness;
  const V3 v = normalize(v3(r2 * V.x, r2 * V.y, V.z));
  const float phi = 2.0f * kPi * rnd.x;
  const float z = (1.0f - rnd.y) * (1.0f + v.z) - v.z;
  const float st = sqrtf(saturate(1.0f - z * z));
  float sp, cp; sincos_det(phi, sp, cp);
  const V3 m = v3(st * cp, st * sp, z) + v;
  return normalize(v3(m.x * r2, m.y * r2, m.z));
}
LUM_DEV float pdf_refraction(float roughness, float NdotH, float NdotV, float HdotV, float HdotL, float ior) {
  const float r4 = pow4(roughness);
  const float D = ggx_d(NdotH, r4), G1 = ggx_g1(r4, NdotV);
  float den = ior * HdotV + HdotL;
  den = den * den;
  return D * G1 * (HdotV / NdotV) * (HdotL / den);
}
LUM_DEV float eval_refraction(float roughness, float HdotL, float HdotV, float NdotH, float NdotL, float NdotV, float ior) {
  const float r4 = pow4(roughness);
  const float D = ggx_d(NdotH, r4), G2 = ggx_g2(r4, NdotL, NdotV);
  float den = ior * HdotV + HdotL;
  den = den * den;
  return 4.0f * NdotL * HdotV * HdotL * D * G2 / den;
}
LUM_DEV float pdf_diffuse(float NdotL) { return saturate(NdotL) * (1.0f / kPi); }
LUM_DEV float eval_diffuse_over_vndf(V3 V, float roughness, float NdotL, float NdotH, float NdotV) {
  const float r4 = pow4(roughness);
  const float D = ggx_d(NdotH, r4);
  float t; const float k = vndf_k(V, r4, t);
  return NdotL * (2.0f * (k * NdotV + t)) / (kPi * D);
}

// ---- lobes (bsdf_utils.cuh:383-587) ----
// The energy-compensation lookups depend on (NdotV, roughness, material flags) only, i.e. on the vertex and not on the direction
// being evaluated, so they are fetched once per vertex and reused by every evaluation there (same values as fetching each time).
struct Energy { float conductor, glossy, dielectric; };
LUM_DEV Energy energy_terms(const DeviceScene& sc, const MatParams& p, float NdotV) {
  Energy e{1.0f, 0.0f, 1.0f};
  const float roughness = p.roughness();
  if ((p.flags & kMatSubstrateMask) == 0) {
    e.conductor = lut2d(sc.lut_conductor, NdotV, roughness);
    if ((p.flags & kMatMetallic) == 0) e.glossy = lut2d(sc.lut_glossy, NdotV, roughness);
  }
  else {
    const float ior = roughness;  // sic: bsdf_utils.cuh:517 reads the roughness parameter
    const bool use_inv = ior > 1.0f;
    const float w = use_inv ? (ior - 1.0f) * 0.5f : (1.0f / ior - 1.0f) * 0.5f;
    e.dielectric = lut3d(use_inv ? sc.lut_dielectric_inv : sc.lut_dielectric, NdotV, roughness, w);
  }
  return e;
}
LUM_DEV float single_scatter_term(const RayTerms& c, int hint, float roughness, float ior_quirk, float inv_pdf) {
  if (hint == kHintGeneral) return eval_microfacet(roughness, c.NdotH, c.NdotL, c.NdotV) * inv_pdf;
  if (hint == kHintMicrofacet) return eval_microfacet_over_vndf(c.V, roughness, c.NdotL, c.NdotV);
  if (hint == kHintDiffuse) return eval_microfacet_over_diffuse(roughness, c.NdotH, c.NdotL, c.NdotV);
  return eval_microfacet(roughness, c.NdotH, c.NdotL, c.NdotV) / pdf_refraction(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior_quirk);
}
LUM_DEV Col lobe_conductor(const Energy& en, const MatParams& p, const RayTerms& c, int hint, float inv_pdf) {
  if (c.NdotL <= 0.0f || c.NdotV <= 0.0f) return splat(0.0f);
  if ((p.flags & kMatSubstrateMask) != 0 || (p.flags & kMatMetallic) == 0) return splat(0.0f);
  const float roughness = p.roughness();
  const float ss = single_scatter_term(c, hint, roughness, (hint == kHintRefraction) ? roughness : 1.0f, inv_pdf);
  const Col albedo = p.albedo();
  const float da = en.conductor;
  const Col fres = fresnel_schlick(albedo, shadowed_f90(albedo), c.HdotV);
  return fres * ss + albedo * (fres * (((1.0f / da) - 1.0f) * ss));
}
LUM_DEV Col lobe_glossy(const Energy& en, const MatParams& p, const RayTerms& c, int hint, float inv_pdf) {
  if (c.NdotL <= 0.0f || c.NdotV <= 0.0f) return splat(0.0f);
  if ((p.flags & kMatSubstrateMask) != 0 || (p.flags & kMatMetallic) != 0) return splat(0.0f);
  const float roughness = p.roughness();
  const float iorq = (hint == kHintRefraction) ? roughness : 1.0f;
  const float ss = single_scatter_term(c, hint, roughness, iorq, inv_pdf);
  float diff;
  if (hint == kHintGeneral) diff = pdf_diffuse(c.NdotL) * inv_pdf;
  else if (hint == kHintDiffuse) diff = 1.0f;
  else if (hint == kHintMicrofacet) diff = eval_diffuse_over_vndf(c.V, roughness, c.NdotL, c.NdotH, c.NdotV);
  else diff = pdf_diffuse(c.NdotL) / pdf_refraction(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, iorq);
  const Col albedo = p.albedo();
  const float cda = en.conductor, gda = en.glossy;
  const Col f0 = col(0.04f, 0.04f, 0.04f);
  const Col fres = fresnel_schlick(f0, shadowed_f90(f0), c.HdotV);
  return fres * (ss / cda) + albedo * (diff * (1.0f - gda));
}
LUM_DEV Col lobe_dielectric(const Energy& en, const MatParams& p, const RayTerms& c, int hint, float inv_pdf) {
  if (c.NdotL <= 0.0f || c.NdotV <= 0.0f) return splat(0.0f);
  if ((p.flags & kMatSubstrateMask) != kMatTranslucent) return splat(0.0f);
  const float roughness = p.roughness();
  const float ior = roughness;  // sic: bsdf_utils.cuh:517 reads the roughness parameter
  float term;
  if (c.is_refraction) {
    if (hint == kHintGeneral) term = eval_refraction(roughness, c.HdotL, c.HdotV, c.NdotH, c.NdotL, c.NdotV, ior) * inv_pdf;
    else if (hint == kHintRefraction) term = ggx_g2_over_g1(pow4(roughness), c.NdotL, c.NdotV);
    else term = 0.0f;
    term *= (1.0f - c.fresnel_dielectric);
  }
  else {
    if (hint == kHintGeneral) term = eval_microfacet(roughness, c.NdotH, c.NdotL, c.NdotV) * inv_pdf;
    else if (hint == kHintMicrofacet) term = eval_microfacet_over_vndf(c.V, roughness, c.NdotL, c.NdotV);
    else term = eval_microfacet(roughness, c.NdotH, c.NdotL, c.NdotV) / pdf_refraction(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior);
    term *= c.fresnel_dielectric;
  }
  term /= en.dielectric;
  if (ior == 1.0f && c.is_refraction) term = (hint == kHintRefraction) ? 1.0f : 0.0f;
  return p.albedo() * term;
}
LUM_DEV Col eval_layers(const Energy& en, const MatParams& p, const RayTerms& c, int hint, float inv_pdf) {
  const float opacity = p.opacity();
  if (c.is_refraction) return lobe_dielectric(en, p, c, hint, inv_pdf) * opacity;
  return ((lobe_conductor(en, p, c, hint, inv_pdf) + lobe_glossy(en, p, c, hint, inv_pdf)) + lobe_dielectric(en, p, c, hint, inv_pdf)) * opacity;
}

// bsdf.cuh:11-50
LUM_DEV RayTerms analyze_direction(const MatParams& p, V3 normal, V3 V, V3 L) {
  RayTerms c;
  c.NdotL = dot(normal, L);
  c.NdotV = saturate(dot(normal, V));
  c.is_refraction = c.NdotL < 0.0f;
  c.NdotL = c.is_refraction ? -c.NdotL : c.NdotL;
  const float ior = p.ior();
  // the dielectric Fresnel term is read by the dielectric lobe only, which is empty unless the substrate is translucent
  const bool needs_fresnel = (p.flags & kMatSubstrateMask) == kMatTranslucent;
  V3 refr = L, H;
  bool total_reflection = false;
  if (c.is_refraction) H = half_vector(L, V, ior);
  else {
    H = half_vector(L, V, 1.0f);
    if (needs_fresnel) refr = refract(V, H, ior, total_reflection);
  }
  c.HdotV = fabsf(dot(H, V));
  c.HdotL = fabsf(dot(H, L));
  c.NdotH = dot(normal, H);
  if (c.NdotH < 0.0f) { H = H * -1.0f; c.NdotH = -c.NdotH; }
  c.fresnel_dielectric = 1.0f;
  if (needs_fresnel && !total_reflection) c.fresnel_dielectric = fresnel_dielectric(H, V, refr, ior);
  c.V = V;
  return c;
}
// bsdf.cuh:52-64
LUM_DEV Col eval_with_face_normal(const Energy& en, const MatParams& p, const RayTerms& c, int hint, V3 L, V3 face_normal, float inv_pdf) {
  const float fl = dot(face_normal, L);
  const float flip = c.is_refraction ? -1.0f : 1.0f;
  if (fl * flip < kEps) return splat(0.0f);
  return eval_layers(en, p, c, hint, inv_pdf);
}
// bsdf.cuh:103-133
LUM_DEV RayTerms sampled_direction_terms(const MatParams& p, V3 normal, V3 V, V3 H, V3 L, bool is_refraction) {
  RayTerms c;
  c.NdotL = dot(normal, L);
  c.NdotV = saturate(dot(normal, V));
  c.is_refraction = is_refraction;
  c.NdotL = is_refraction ? -c.NdotL : c.NdotL;
  const float ior = p.ior();
  const bool needs_fresnel = (p.flags & kMatSubstrateMask) == kMatTranslucent;  // see analyze_direction
  bool total_reflection = false;
  V3 refr = L;
  if (!is_refraction && needs_fresnel) refr = refract(V, H, ior, total_reflection);
  c.HdotV = fabsf(dot(H, V));
  c.HdotL = fabsf(dot(H, L));
  c.NdotH = dot(normal, H);
  float flip = 1.0f;
  if (c.NdotH < 0.0f) { flip = -1.0f; c.NdotH = -c.NdotH; }
  c.fresnel_dielectric = 1.0f;
  if (needs_fresnel && !total_reflection) c.fresnel_dielectric = fresnel_dielectric(H * flip, V, refr, ior);
  c.V = V;
  return c;
}
// bsdf.cuh:73-83; `en` = energy_terms(sc, g.params, world_ndotv(g))
LUM_DEV float world_ndotv(const GeoContext& g) { return saturate(dot(g.normal, g.V)); }
LUM_DEV Col eval_bsdf(const Energy& en, const GeoContext& g, V3 L, int hint, bool& is_refraction, float inv_pdf) {
  const RayTerms c = analyze_direction(g.params, g.normal, g.V, L);
  is_refraction = c.is_refraction;
  return eval_with_face_normal(en, g.params, c, hint, L, normal_unpack(g.face_normal_packed), inv_pdf);
}

// Shading frame with the normal on +z, shared by the bounce and the BSDF-driven light direction (bsdf.cuh:150-156, light_bsdf.cuh:30-36).
struct LocalFrame { Quat to_z; V3 V, face_normal; Energy energy; };
LUM_DEV LocalFrame local_frame(const DeviceScene& sc, const GeoContext& g) {
  LocalFrame f;
  f.to_z = rotation_to_z(g.normal);
  f.V = qapply(f.to_z, g.V);
  f.face_normal = qapply(f.to_z, normal_unpack(g.face_normal_packed));
  f.energy = energy_terms(sc, g.params, saturate(dot(v3(0.0f, 0.0f, 1.0f), f.V)));
  return f;
}

struct BounceSample { V3 ray; Col weight; bool transparent_pass, microfacet_based; };

// Three-technique resampled bounce (bsdf.cuh:138-301). `set` picks RandomSet::BSDF<set> (random.cuh:120-129).
template <class Smp>
LUM_DEV BounceSample sample_bounce(const LocalFrame& lf, const GeoContext& g, const Smp& smp, uint32_t set) {
  const MatParams& p = g.params;
  BounceSample out;
  const float opacity = p.opacity();
  if (opacity < 1.0f) {
    if (smp.next1(kRndBsdfOpacity + set) > opacity) {
      out.ray = g.V * -1.0f;
      out.weight = (p.flags & kMatColoredTransparency) ? p.albedo() : col(1.0f, 1.0f, 1.0f);
      out.microfacet_based = false;
      out.transparent_pass = true;
      return out;
    }
  }
  const Quat to_z = lf.to_z;
  const V3 Vl = lf.V, fnl = lf.face_normal;
  const V3 up = v3(0.0f, 0.0f, 1.0f);
  const uint32_t substrate = p.flags & kMatSubstrateMask;
  const bool with_diffuse = (substrate == 0) && ((p.flags & kMatMetallic) == 0);
  const bool with_refraction = substrate == kMatTranslucent;
  float pick = smp.next1(kRndBsdfResampling + set);
  const float ior = p.ior(), roughness = p.roughness();
  V3 chosen_ray;
  Col chosen_eval;
  float weight_sum;
  out.transparent_pass = false;
  {
    const V3 m = sample_vndf_bounded(Vl, roughness, smp.next2(kRndBsdfReflection + set));
    const V3 ray = reflect(Vl, m);
    const RayTerms c = sampled_direction_terms(p, up, Vl, m, ray, false);
    const Col eval = eval_with_face_normal(lf.energy, p, c, kHintMicrofacet, ray, fnl, 1.0f);
    const float pdf = pdf_vndf_bounded(Vl, roughness, c.NdotH, c.NdotV);
    const float dp = with_diffuse ? pdf_diffuse(c.NdotL) : 0.0f;
    const float rp = with_refraction ? pdf_refraction(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior) : 0.0f;
    const float sum = pdf + dp + rp;
    const float mis = (sum > 0.0f) ? pdf / sum : 0.0f;
    chosen_ray = ray; chosen_eval = eval; weight_sum = importance(eval) * mis;
    out.microfacet_based = true;
  }
  if (with_diffuse) {
    const F2 r2 = smp.next2(kRndBsdfDiffuse + set);
    const V3 ray = sample_ray_sphere(r2.x, r2.y);
    const V3 m = normalize(Vl + ray);
    const RayTerms c = sampled_direction_terms(p, up, Vl, m, ray, false);
    const Col eval = eval_with_face_normal(lf.energy, p, c, kHintDiffuse, ray, fnl, 1.0f);
    const float pdf = pdf_diffuse(c.NdotL);
    const float mp = pdf_vndf_bounded(Vl, roughness, c.NdotH, c.NdotV);
    const float rp = with_refraction ? pdf_refraction(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior) : 0.0f;
    const float sum = pdf + mp + rp;
    const float mis = (sum > 0.0f) ? pdf / sum : 0.0f;
    const float w = importance(eval) * mis;
    weight_sum += w;
    const float prob = w / weight_sum;
    if (pick < prob) {
      chosen_ray = ray; chosen_eval = eval; out.transparent_pass = false; out.microfacet_based = false;
      pick = clamp_random(pick / prob);
    }
    else pick = clamp_random((pick - prob) / (1.0f - prob));
  }
  if (with_refraction) {
    bool total_reflection;
    const V3 m = sample_vndf_caps(Vl, roughness, smp.next2(kRndBsdfRefraction + set));
    const V3 ray = refract(Vl, m, ior, total_reflection);
    const RayTerms c = sampled_direction_terms(p, up, Vl, m, ray, !total_reflection);
    const Col eval = eval_with_face_normal(lf.energy, p, c, kHintRefraction, ray, fnl, 1.0f);
    float mis = 1.0f;
    if (total_reflection) {
      const float pdf = pdf_refraction(roughness, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior);
      const float refl = pdf_vndf_bounded(Vl, roughness, c.NdotH, c.NdotV);
      const float dp = with_diffuse ? pdf_diffuse(c.NdotL) : 0.0f;
      const float sum = pdf + refl + dp;
      mis = (sum > 0.0f) ? pdf / sum : 0.0f;
    }
    const float w = importance(eval) * mis;
    weight_sum += w;
    const float prob = w / weight_sum;
    if (pick < prob) {
      chosen_ray = ray; chosen_eval = eval; out.transparent_pass = !total_reflection; out.microfacet_based = true;
      pick = clamp_random(pick / prob);
    }
    else pick = clamp_random((pick - prob) / (1.0f - prob));
  }
  out.weight = (weight_sum > 0.0f) ? chosen_eval * (weight_sum / importance(chosen_eval)) : col(0.0f, 0.0f, 0.0f);
  out.ray = normalize(qapply(qinv(to_z), chosen_ray));
  return out;
}

LUM_DEV bool is_pass_through(const GeoContext& g, const BounceSample& s) {  // bsdf_utils.cuh:68-77
  return s.transparent_pass && ((g.params.ior() == 1.0f) || !s.microfacet_based);
}

LUM_NS_END
