/*
 * ORACLE (test infrastructure, not product): CPU restatement of the reference's device math.
 * Follows /root/reference/src/luminary/device/cuda/math.cuh (file:line cited per function).
 *
 * Parity status: "parity unpinned" by the reference (it ships no tests/fixtures and cannot be built here,
 * SURVEY.md §0 F2/F8). The integer paths are pinned by the two known answers recorded in SURVEY.md §0 F9.
 *
 * Numerics contract shared with the HIP product (DESIGN.md "Determinism"):
 *   - only IEEE-754 binary32 +,-,*,/ and sqrt (all correctly rounded), compiled with -ffp-contract=off;
 *   - rsqrt(x) := 1/sqrt(x); saturate(x) := fmin(fmax(x,0),1);
 *   - sin/cos/atan2 are fixed polynomial sequences defined here (the reference builds with
 *     --use_fast_math, so its own transcendental bits are unspecified anyway);
 *   - float->uint conversions saturate (CUDA cvt semantics).
 * With that contract oracle and product agree bit-for-bit, which is what the parity tests assert.
 */
#ifndef ORACLE_O_MATH_H
#define ORACLE_O_MATH_H

#include <float.h>
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } vec3;
typedef struct { float r, g, b; } RGBF;
typedef struct { float r, g, b, a; } RGBAF;
typedef struct { float u, v; } UV;
typedef struct { float x, y; } float2_t;
typedef struct { float x, y, z, w; } Quat;
typedef struct { uint32_t x, y; } uint2_t;
typedef struct { uint16_t x, y, z, w; } Quat16;

#define O_PI 3.14159265358979323846f
#define O_EPS FLT_EPSILON

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* CUDA float->u32 conversion saturates; C leaves out-of-range undefined. */
static inline uint32_t f2u_sat(float v) {
  if (!(v >= 0.0f)) return 0u;
  if (v >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t) v;
}

static inline float o_saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float o_rsqrt(float x) { return 1.0f / sqrtf(x); }
static inline float o_exp2i(int e) { return ldexpf(1.0f, e); }

/* ---- deterministic transcendentals (contract above) ---- */

/* Cody-Waite reduction by pi/2 followed by degree-7/8 minimax polynomials on [-pi/4, pi/4]. */
static inline void o_sincos(float x, float* s_out, float* c_out) {
  const float fj = rintf(x * 0.636619772367581343f);
  const int j    = (int) fj;
  float y        = x - fj * 1.5703125f;
  y              = y - fj * 4.837512969970703125e-4f;
  y              = y - fj * 7.54978995489188e-8f;
  const float z  = y * y;

  float sp = -1.9515295891e-4f;
  sp       = sp * z + 8.3321608736e-3f;
  sp       = sp * z + -1.6666654611e-1f;
  const float s = y + y * (z * sp);

  float cp = 2.443315711809948e-5f;
  cp       = cp * z + -1.388731625493765e-3f;
  cp       = cp * z + 4.166664568298827e-2f;
  const float c = (1.0f - 0.5f * z) + (z * z) * cp;

  float sr, cr;
  switch (j & 3) {
    case 0: sr = s; cr = c; break;
    case 1: sr = c; cr = -s; break;
    case 2: sr = -s; cr = -c; break;
    default: sr = -c; cr = s; break;
  }
  *s_out = sr;
  *c_out = cr;
}

static inline float o_sin(float x) { float s, c; o_sincos(x, &s, &c); return s; }
static inline float o_cos(float x) { float s, c; o_sincos(x, &s, &c); return c; }

/* atan on [0, inf) by two-step range reduction + odd polynomial. */
static inline float o_atan_pos(float x) {
  float y0;
  if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; x = -1.0f / x; }
  else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
  else { y0 = 0.0f; }
  const float z = x * x;
  float p = 8.05374449538e-2f;
  p       = p * z - 1.38776856032e-1f;
  p       = p * z + 1.99777106478e-1f;
  p       = p * z - 3.33329491539e-1f;
  return y0 + (p * z * x + x);
}

static inline float o_atan2(float y, float x) {
  if (x != x || y != y) return x + y;
  if (y == 0.0f) return (x < 0.0f || (x == 0.0f && signbit(x))) ? copysignf(O_PI, y) : copysignf(0.0f, y);
  if (x == 0.0f) return copysignf(1.5707963267948966f, y);
  const float a = o_atan_pos(fabsf(y) / fabsf(x));
  const float r = (x < 0.0f) ? (O_PI - a) : a;
  return copysignf(r, y);
}

/* ---- vectors (math.cuh:19-218) ---- */
/* log2 / exp2 / pow as fixed sequences (relative error < 3e-7), used where the reference calls log2f / powf */
static inline float o_log2(float x) {
  const uint32_t bits = f2u(x);
  int e = (int) ((bits >> 23) & 0xFFu) - 127;
  float m = u2f((bits & 0x007FFFFFu) | 0x3F800000u);
  if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }
  const float s = (m - 1.0f) / (m + 1.0f);
  const float z = s * s;
  float p = 0.0909090909f;
  p = p * z + 0.111111111f;
  p = p * z + 0.142857143f;
  p = p * z + 0.2f;
  p = p * z + 0.333333333f;
  p = p * z;
  const float ln_m = 2.0f * s + (2.0f * s) * p;
  return (float) e + ln_m * 1.44269504f;
}
static inline float o_exp2(float x) {
  x = fminf(fmaxf(x, -126.0f), 127.0f);
  const float n = rintf(x);
  const float f = x - n;
  float p = 1.52527338e-5f;
  p = p * f + 1.54035304e-4f;
  p = p * f + 1.33335581e-3f;
  p = p * f + 9.61812911e-3f;
  p = p * f + 5.55041087e-2f;
  p = p * f + 2.40226507e-1f;
  p = p * f + 6.93147181e-1f;
  p = p * f + 1.0f;
  return ldexpf(p, (int) n);
}
static inline float o_pow(float x, float y) { return (x > 0.0f) ? o_exp2(y * o_log2(x)) : 0.0f; }

static inline vec3 v3(float x, float y, float z) { vec3 r = {x, y, z}; return r; }
static inline vec3 v_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v_sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v_mul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 v_scale(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 v_inv(vec3 a) { return v3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
static inline float v_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline vec3 v_cross(vec3 a, vec3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float v_len(vec3 a) { return sqrtf(v_dot(a, a)); }
/* math.cuh:178-186 (rsqrtf -> 1/sqrt by contract) */
static inline vec3 v_norm(vec3 a) { const float s = o_rsqrt(v_dot(a, a)); return v3(a.x * s, a.y * s, a.z * s); }
/* math.cuh:196-201 */
static inline vec3 v_reflect(vec3 V, vec3 n) {
  const float d = v_dot(V, n);
  return v_norm(v_sub(v_scale(n, 2.0f * d), V));
}
static inline float o_lerp(float a, float b, float t) { return a + t * (b - a); }
/* math.cuh:50-56 */
static inline float o_remap01(float v, float lo, float hi) { return o_saturate((v - lo) / (hi - lo)); }

/* ---- colours (math.cuh:800-1070) ---- */
static inline RGBF c3(float r, float g, float b) { RGBF c = {r, g, b}; return c; }
static inline RGBF c_splat(float v) { return c3(v, v, v); }
static inline RGBF c_add(RGBF a, RGBF b) { return c3(a.r + b.r, a.g + b.g, a.b + b.b); }
static inline RGBF c_sub(RGBF a, RGBF b) { return c3(a.r - b.r, a.g - b.g, a.b - b.b); }
static inline RGBF c_mul(RGBF a, RGBF b) { return c3(a.r * b.r, a.g * b.g, a.b * b.b); }
static inline RGBF c_scale(RGBF a, float s) { return c3(a.r * s, a.g * s, a.b * s); }
static inline int c_any(RGBF a) { return (a.r > 0.0f || a.g > 0.0f || a.b > 0.0f); }
static inline float c_luminance(RGBF v) { return 0.212655f * v.r + 0.715158f * v.g + 0.072187f * v.b; }
/* math.cuh:1066-1068 + intrinsics.cuh:91-127: a SIGNED-INTEGER max over the float bit patterns (bug-compatible, F10). */
static inline float c_importance(RGBF c) {
  int32_t a = (int32_t) f2u(c.r), b = (int32_t) f2u(c.g), d = (int32_t) f2u(c.b);
  int32_t m = b > d ? b : d;
  m         = a > m ? a : m;
  return u2f((uint32_t) m);
}

/* ---- quaternions / transforms (math.cuh:346-486) ---- */
static inline Quat q_inverse(Quat q) { Quat r = {-q.x, -q.y, -q.z, q.w}; return r; }
/* math.cuh:368-391 */
static inline Quat q_rotation_to_z(vec3 v) {
  Quat r;
  if (v.z < -1.0f + O_EPS) { r.x = 1.0f; r.y = 0.0f; r.z = 0.0f; r.w = 0.0f; return r; }
  r.x = v.y; r.y = -v.x; r.z = 0.0f; r.w = 1.0f + v.z;
  const float n = o_rsqrt(r.x * r.x + r.y * r.y + r.w * r.w);
  r.x *= n; r.y *= n; r.w *= n;
  return r;
}
/* math.cuh:393-409 */
static inline vec3 q_apply(Quat q, vec3 v) {
  const vec3 u  = v3(q.x, q.y, q.z);
  const float s = q.w;
  const float duv = v_dot(u, v), duu = v_dot(u, u);
  const vec3 cr = v_cross(u, v);
  vec3 r = v_scale(u, 2.0f * duv);
  r      = v_add(r, v_scale(v, s * s - duu));
  r      = v_add(r, v_scale(cr, 2.0f * s));
  return r;
}
/* math.cuh:411-433 */
static inline vec3 q16_apply(Quat16 q, vec3 v) {
  Quat f;
  f.x = (q.x * (1.0f / 0x7FFF)) - 1.0f; f.y = (q.y * (1.0f / 0x7FFF)) - 1.0f;
  f.z = (q.z * (1.0f / 0x7FFF)) - 1.0f; f.w = (q.w * (1.0f / 0x7FFF)) - 1.0f;
  return q_apply(f, v);
}
static inline vec3 q16_apply_inv(Quat16 q, vec3 v) {
  Quat f;
  f.x = 1.0f - (q.x * (1.0f / 0x7FFF)); f.y = 1.0f - (q.y * (1.0f / 0x7FFF));
  f.z = 1.0f - (q.z * (1.0f / 0x7FFF)); f.w = (q.w * (1.0f / 0x7FFF)) - 1.0f;
  return q_apply(f, v);
}

typedef struct { vec3 translation; vec3 scale; Quat16 rotation; } OTransform; /* device_structs.h:295-300, 32 bytes */

/* math.cuh:459-489 */
static inline vec3 t_rot(OTransform t, vec3 v) { return q16_apply(t.rotation, v); }
static inline vec3 t_rot_inv(OTransform t, vec3 v) { return q16_apply_inv(t.rotation, v); }
static inline vec3 t_rel(OTransform t, vec3 v) { return v_mul(t_rot(t, v), t.scale); }
static inline vec3 t_rel_inv(OTransform t, vec3 v) { return t_rot_inv(t, v_mul(v, v_inv(t.scale))); }
static inline vec3 t_apply(OTransform t, vec3 v) { return v_add(t_rel(t, v), t.translation); }
static inline vec3 t_apply_inv(OTransform t, vec3 v) { return t_rel_inv(t, v_sub(v, t.translation)); }

/* math.cuh:203-214 */
static inline float2_t tri_coords(vec3 vertex, vec3 e1, vec3 e2, vec3 p) {
  const vec3 d = v_sub(p, vertex);
  const float d00 = v_dot(e1, e1), d01 = v_dot(e1, e2), d11 = v_dot(e2, e2);
  const float d20 = v_dot(d, e1), d21 = v_dot(d, e2);
  const float den = 1.0f / (d00 * d11 - d01 * d01);
  float2_t r = {(d11 * d20 - d01 * d21) * den, (d00 * d21 - d01 * d20) * den};
  return r;
}
/* math.cuh:216-228 */
static inline vec3 lerp_normals(vec3 vn, vec3 e1n, vec3 e2n, float2_t c, vec3 face) {
  vec3 r;
  r.x = vn.x + c.x * e1n.x + c.y * e2n.x;
  r.y = vn.y + c.x * e1n.y + c.y * e2n.y;
  r.z = vn.z + c.x * e1n.z + c.y * e2n.z;
  const float l = v_len(r);
  return (l < O_EPS) ? face : v_scale(r, 1.0f / l);
}
/* math.cuh:326-344 */
static inline vec3 sample_ray_sphere(float alpha, float beta) {
  if (fabsf(alpha) > 1.0f - O_EPS) return v3(0.0f, 0.0f, copysignf(1.0f, alpha));
  const float a = sqrtf(1.0f - alpha * alpha);
  const float b = 2.0f * O_PI * beta;
  float s, c;
  o_sincos(b, &s, &c);
  return v3(a * c, a * s, alpha);
}
/* math.cuh:766-786 */
static inline vec3 refract_vector(vec3 V, vec3 n, float index_ratio, bool* total_reflection) {
  if (index_ratio < O_EPS) { *total_reflection = false; return v_scale(V, -1.0f); }
  const float d = fabsf(v_dot(n, V));
  const float b = 1.0f - index_ratio * index_ratio * (1.0f - d * d);
  *total_reflection = b < 0.0f;
  if (*total_reflection) return v_reflect(V, n);
  return v_norm(v_sub(v_scale(n, index_ratio * d - sqrtf(b)), v_scale(V, index_ratio)));
}
/* math.cuh:1337-1358 == light_triangle.cuh:10-31 (Moeller-Trumbore; t<0 or NaN -> FLT_MAX) */
static inline float tri_intersect(vec3 vertex, vec3 e1, vec3 e2, vec3 origin, vec3 ray, float2_t* coords) {
  const vec3 h  = v_cross(ray, e2);
  const float a = v_dot(e1, h);
  const float f = 1.0f / a;
  const vec3 s  = v_sub(origin, vertex);
  const float u = f * v_dot(s, h);
  const vec3 q  = v_cross(s, e1);
  const float v = f * v_dot(ray, q);
  coords->x = u; coords->y = v;
  if (v < 0.0f || u < 0.0f || !(u + v <= 1.0f)) return FLT_MAX;
  const float t = f * v_dot(e2, q);
  return (t >= 0.0f) ? t : FLT_MAX;
}
/* math.cuh:1498-1523 */
static inline vec3 normal_adaptation_apply(vec3 V, vec3 sn, vec3 gn) {
  if (v_dot(sn, gn) < 0.0f) sn = v_scale(sn, -1.0f);
  if (v_dot(V, sn) < 0.0f) {
    const vec3 proj = v_scale(V, v_dot(sn, V));
    return v_norm(v_sub(sn, v_scale(proj, 1.1f)));
  }
  return sn;
}

#endif
