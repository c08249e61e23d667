/*
 * ORACLE (test infrastructure, not product): lossy encodings applied between kernels.
 * Follows /root/reference/src/luminary/device/cuda/math.cuh:1525-1768 and cuda/medium_stack.cuh.
 * Integer bit-ops on float bits -> exact.
 */
#ifndef ORACLE_O_PACK_H
#define ORACLE_O_PACK_H

#include "o_math.h"

/* math.cuh:1529-1545 */
static inline float bfloat_unpack(uint16_t v) { return u2f(((uint32_t) v) << 16); }
static inline uint16_t bfloat_pack(float v) { return (uint16_t) (f2u(v) >> 16); }

/* math.cuh:1547-1575: throughput keeps 21 bits per channel (sign, exponent, 12 mantissa bits). */
static inline RGBF record_unpack(uint2_t p) {
  const uint32_t red = p.x & 0x1FFFFFu, green = (p.x >> 21) | ((p.y & 0x3FFu) << 11), blue = p.y >> 10;
  return c3(u2f(red << 11), u2f(green << 11), u2f(blue << 11));
}
static inline uint2_t record_pack(RGBF r) {
  const uint32_t red = f2u(r.r) >> 11, green = f2u(r.g) >> 11, blue = f2u(r.b) >> 11;
  uint2_t p;
  p.x = red | (green << 21);
  p.y = (green >> 11) | (blue << 10);
  return p;
}
/* math.cuh:1577-1620: octahedral 2x32 bit direction */
static inline vec3 ray_unpack(uint2_t p) {
  float x = p.x * (1.0f / 0xFFFFFFFF), y = p.y * (1.0f / 0xFFFFFFFF);
  x = (x * 2.0f) - 1.0f; y = (y * 2.0f) - 1.0f;
  vec3 r = v3(x, y, 1.0f - fabsf(x) - fabsf(y));
  const float t = o_saturate(-r.z);
  r.x += (r.x >= 0.0f) ? -t : t;
  r.y += (r.y >= 0.0f) ? -t : t;
  return v_norm(r);
}
static inline uint2_t ray_pack(vec3 ray) {
  float x = ray.x, y = ray.y, z = ray.z;
  const float rn = 1.0f / (fabsf(x) + fabsf(y) + fabsf(z));
  x *= rn; y *= rn; z *= rn;
  const float t = o_saturate(-z);
  x += (x >= 0.0f) ? t : -t;
  y += (y >= 0.0f) ? t : -t;
  x = fminf(1.0f, fmaxf(-1.0f, x)); y = fminf(1.0f, fmaxf(-1.0f, y));
  x = (x + 1.0f) * 0.5f; y = (y + 1.0f) * 0.5f;
  uint2_t p;
  p.x = f2u_sat(x * 0xFFFFFFFF + 0.5f);
  p.y = f2u_sat(y * 0xFFFFFFFF + 0.5f);
  return p;
}
/* math.cuh:1677-1686 */
static inline float normed_u16(uint32_t d) { return ((uint16_t) d) * (1.0f / 0xFFFF); }
static inline float unsigned_float_unpack(uint32_t d) { return u2f(((uint32_t) (uint16_t) d) << 15); }
static inline UV uv_unpack(uint32_t d) { UV uv = {u2f(d & 0xFFFF0000u), u2f(d << 16)}; return uv; }
/* math.cuh:1697-1711 */
static inline vec3 normal_unpack(uint32_t d) {
  float x = (d & 0xFFFFu) * (1.0f / 0xFFFF), y = (d >> 16) * (1.0f / 0xFFFF);
  x = (x * 2.0f) - 1.0f; y = (y * 2.0f) - 1.0f;
  vec3 n = v3(x, y, 1.0f - fabsf(x) - fabsf(y));
  const float t = o_saturate(-n.z);
  n.x += (n.x >= 0.0f) ? -t : t;
  n.y += (n.y >= 0.0f) ? -t : t;
  return v_norm(n);
}
/* math.cuh:1713-1741 */
static inline uint32_t normal_pack(vec3 n) {
  float x = n.x, y = n.y, z = n.z;
  const float rn = 1.0f / (fabsf(x) + fabsf(y) + fabsf(z));
  x *= rn; y *= rn; z *= rn;
  const float t = fmaxf(fminf(-z, 1.0f), 0.0f);
  x += (x >= 0.0f) ? t : -t;
  y += (y >= 0.0f) ? t : -t;
  x = fmaxf(fminf(x, 1.0f), -1.0f); y = fmaxf(fminf(y, 1.0f), -1.0f);
  x = (x + 1.0f) * 0.5f; y = (y + 1.0f) * 0.5f;
  const uint32_t xu = f2u_sat(x * 0xFFFF + 0.5f), yu = f2u_sat(y * 0xFFFF + 0.5f);
  return (yu << 16) | xu;
}
/* math.cuh:1743-1768 */
static inline uint32_t ior_compress(float ior) { return (f2u((0.5f * (ior - 1.0f)) + 1.0f) >> 15) & 0xFFu; }
static inline float ior_decompress(uint32_t c) { return ((u2f(0x3F800000u | (c << 15)) - 1.0f) * 2.0f) + 1.0f; }

/* medium_stack.cuh:11-29 (the volume id words stay 0 on the triangle path) */
static inline float medium_ior_peek(uint32_t stack, bool previous) { return ior_decompress(((previous) ? stack >> 8 : stack) & 0xFFu); }
static inline uint32_t medium_ior_modify(uint32_t stack, float ior, bool push) {
  return push ? ((stack << 8) | ior_compress(ior)) : (stack >> 8);
}

#endif
