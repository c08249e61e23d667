/*
 * ORACLE (test infrastructure, not product): material decode + per-vertex parameter quantisation.
 * Follows /root/reference/src/luminary/device/cuda/memory.cuh:442-474 (load_material) and
 * cuda/material.cuh:36-53, 125-325 (MaterialParams bit allocation, get/set with re-quantisation).
 */
#ifndef ORACLE_O_MATERIAL_H
#define ORACLE_O_MATERIAL_H

#include "o_pack.h"

/* device_structs.h:186-200 */
enum {
  DMAT_SUBSTRATE_MASK = 0x01, DMAT_EMISSION = 0x02, DMAT_THIN_WALLED = 0x04, DMAT_METALLIC = 0x08,
  DMAT_COLORED_TRANSPARENCY = 0x10, DMAT_ROUGHNESS_AS_SMOOTHNESS = 0x20, DMAT_NORMAL_MAP_COMPRESSED = 0x40,
  DMAT_BIDIRECTIONAL_EMISSION = 0x80
};
/* device_utils.h:258-270 */
enum { MAT_TRANSLUCENT = 1, MAT_SUBSTRATE_MASK = 1, MAT_REFRACTION_IS_INSIDE = 2, MAT_METALLIC = 4, MAT_COLORED_TRANSPARENCY = 8 };

#define TEXTURE_NONE 0xFFFFu

/* device_structs.h:202-223: 32 bytes, 16 x u16 */
typedef struct { uint16_t w[16]; } OMaterialC;

typedef struct {
  uint8_t flags;
  float roughness_clamp, roughness, refraction_index;
  RGBAF albedo;
  RGBF emission;
  float emission_scale;
  uint16_t albedo_tex, luminance_tex, roughness_tex, metallic_tex, normal_tex;
} OMaterial;

/* memory.cuh:442-474. Word layout: w0 = flags | clamp<<8, w1 metallic_tex, w2 roughness, w3 ior, w4..7 albedo rgba,
 * w8..10 emission rgb, w11 emission_scale, w12 albedo_tex, w13 luminance_tex, w14 roughness_tex, w15 normal_tex. */
static inline OMaterial material_load(const OMaterialC* m) {
  OMaterial r;
  const uint16_t* w = m->w;
  r.flags            = (uint8_t) (w[0] & 0x00FF);
  r.roughness_clamp  = normed_u16(w[0] & 0xFF00);
  r.metallic_tex     = w[1];
  r.roughness        = normed_u16(w[2]);
  r.refraction_index = normed_u16(w[3]) * 2.0f + 1.0f;
  r.albedo.r = normed_u16(w[4]); r.albedo.g = normed_u16(w[5]); r.albedo.b = normed_u16(w[6]); r.albedo.a = normed_u16(w[7]);
  r.emission.r = normed_u16(w[8]); r.emission.g = normed_u16(w[9]); r.emission.b = normed_u16(w[10]);
  r.emission_scale = unsigned_float_unpack(w[11]);
  r.albedo_tex = w[12]; r.luminance_tex = w[13]; r.roughness_tex = w[14]; r.normal_tex = w[15];
  r.emission = c_scale(r.emission, r.emission_scale);
  return r;
}

/* material.cuh:36-53: bit offsets EMISSION 0..31, ALBEDO 32..61, OPACITY 62..69, ROUGHNESS 70..79, IOR 80..87 */
typedef struct { uint32_t data[3]; uint32_t flags; } MatParams;

static inline uint32_t mp_get(const MatParams* p, uint32_t off, uint32_t size) {
  const uint32_t idx = off >> 5, sh = off & 31u, mask = (size < 32u) ? ((1u << size) - 1u) : 0xFFFFFFFFu;
  uint32_t r = (p->data[idx] >> sh) & mask;
  if (sh + size > 32u) r |= (p->data[idx + 1] & ((1u << (sh + size - 32u)) - 1u)) << (32u - sh);
  return r;
}
static inline void mp_set(MatParams* p, uint32_t off, uint32_t size, uint32_t v) {
  const uint32_t idx = off >> 5, sh = off & 31u, mask = (size < 32u) ? ((1u << size) - 1u) : 0xFFFFFFFFu;
  p->data[idx] = (p->data[idx] & ~(mask << sh)) | ((v & mask) << sh);
  if (sh + size > 32u) {
    const uint32_t sh2 = 32u - sh;
    p->data[idx + 1] = (p->data[idx + 1] & ~(mask >> sh2)) | ((v & mask) >> sh2);
  }
}
/* material.cuh:125-153 / 210-241 */
static inline float mp_get_norm(const MatParams* p, uint32_t off, uint32_t size) { return mp_get(p, off, size) * (1.0f / ((1u << size) - 1)); }
static inline void mp_set_norm(MatParams* p, uint32_t off, uint32_t size, float v01) {
  mp_set(p, off, size, (uint32_t) (o_saturate(v01) * ((1u << size) - 1) + 0.5f));
}
static inline float mp_opacity(const MatParams* p) { return mp_get_norm(p, 62, 8); }
static inline float mp_roughness(const MatParams* p) { return mp_get_norm(p, 70, 10); }
static inline float mp_ior(const MatParams* p) { return mp_get_norm(p, 80, 8) * 3.0f; }
static inline void mp_set_opacity(MatParams* p, float v) { mp_set_norm(p, 62, 8, v); }
static inline void mp_set_roughness(MatParams* p, float v) { mp_set_norm(p, 70, 10, v); }
static inline void mp_set_ior(MatParams* p, float v) { mp_set_norm(p, 80, 8, v * (1.0f / 3.0f)); }
/* material.cuh:155-170 / 250-258: 3 x 10 bit */
static inline RGBF mp_albedo(const MatParams* p) {
  const uint32_t d = mp_get(p, 32, 30);
  return c3((d & 0x3FF) * (1.0f / 0x3FF), ((d >> 10) & 0x3FF) * (1.0f / 0x3FF), (d >> 20) * (1.0f / 0x3FF));
}
static inline void mp_set_albedo(MatParams* p, RGBF v) {
  const uint32_t r = (uint32_t) (o_saturate(v.r) * 0x3FF + 0.5f), g = (uint32_t) (o_saturate(v.g) * 0x3FF + 0.5f),
                 b = (uint32_t) (o_saturate(v.b) * 0x3FF + 0.5f);
  mp_set(p, 32, 30, r | (g << 10) | (b << 20));
}
/* material.cuh:171-199 / 259-299: shared-exponent emission, [0,1023]^3 */
static inline RGBF mp_emission(const MatParams* p) {
  const uint32_t d = mp_get(p, 0, 32);
  const uint32_t dmax = d & 0x3FFF, dlo = (d >> 14) & 0xFF, dhi = (d >> 22) & 0xFF, comp = d >> 30;
  const float mx = (dmax > 0) ? u2f((dmax << 14) | 0x30000000u) * (1023.0f / 2.0f) : 0.0f;
  const float lo = dlo * (1.0f / 0xFF) * mx, hi = dhi * (1.0f / 0xFF) * mx;
  if (comp == 0) return c3(mx, lo, hi);
  if (comp == 1) return c3(lo, mx, hi);
  return c3(lo, hi, mx);
}
static inline void mp_set_emission(MatParams* p, RGBF v) {
  uint32_t comp;
  float mx, lo, hi;
  if (v.r > v.g && v.r > v.b) { comp = 0; mx = v.r; lo = v.g; hi = v.b; }
  else if (v.g > v.b) { comp = 1; mx = v.g; lo = v.r; hi = v.b; }
  else { comp = 2; mx = v.b; lo = v.r; hi = v.g; }
  mx = o_saturate(mx * (1.0f / 1023.0f)) * 2.0f;
  lo = o_saturate(lo * (2.0f / 1023.0f) * (1.0f / mx));
  hi = o_saturate(hi * (2.0f / 1023.0f) * (1.0f / mx));
  const uint32_t dmax = (f2u(mx) >= 0x30000000u) ? (f2u(mx) >> 14) & 0x3FFF : 0;
  const uint32_t dlo = (uint32_t) (lo * 0xFF + 0.5f), dhi = (uint32_t) (hi * 0xFF + 0.5f);
  mp_set(p, 0, 32, dmax | (dlo << 14) | (dhi << 22) | (comp << 30));
}

#endif
