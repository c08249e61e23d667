// Device-side math for the gfx950 path-tracing kernels.
//
// Numerics contract (DESIGN.md "Determinism"): IEEE binary32 +,-,*,/ and sqrt only, built with -ffp-contract=off;
// rsqrt := 1/sqrt; sin/cos/atan2 are the fixed polynomial sequences below. The reference is built with
// --use_fast_math, so its transcendental bits are unspecified; fixing ours makes every sample a pure, reproducible
// function of (scene, pixel, sample id) on any number of GPUs.
//
// Reference for the formulas: src/luminary/device/cuda/math.cuh (cited per function).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flavour.h"

// Parts of the fast flavour, individually switchable for diagnosis (tools/flavour_diff.py): all on by default.
#ifndef LUM_FAST_RSQ
#define LUM_FAST_RSQ LUM_FAST
#endif
#ifndef LUM_FAST_SINCOS
#define LUM_FAST_SINCOS LUM_FAST
#endif
#ifndef LUM_FAST_EXPLOG
#define LUM_FAST_EXPLOG LUM_FAST
#endif

#define LUM_DEV __device__ __forceinline__

// Diagnostic build (-DLUM_PHASE_STATS, tools/phase_stats.py): how many lanes are active where. LUM_STAT(i, l) adds one to
// g_phase[i] per wave-level execution and the number of active lanes to g_phase[l].
#ifdef LUM_PHASE_STATS
#if LUM_FAST
#define g_phase g_phase_fast  // one counter block per flavour (two translation units in one library)
#endif
__device__ unsigned long long g_phase[16];
#if LUM_FAST
#define g_phase_time g_phase_time_fast
#endif
// wave-level cycles (s_memtime) per kind of iteration of the ray kernels' phase loop, and how many there were:
//   0/1 node iterations that touch memory   2/3 node iterations on staged nodes only   4/5 triangle iterations   6/7 instance-entry iterations
//   8/9 refills (ray fetch)   10/11 whole kernel per wave
__device__ unsigned long long g_phase_time[16];
#if LUM_FAST
#define g_shade_time g_shade_time_fast
#endif
// wave-level cycles of k_shade's parts (per batch of 64 vertices):
//   0 queue words + surface context + local frame   1 light-tree root pass + energy terms   2 the resampling candidates   3 BSDF-driven light direction
//   4 bounce + ambient record   5 sun   6 NEE stores, classification, roulette   7 appends   8 collecting hits (input rounds)   9 batches   10 candidate: pick + triangle sample
//   11 candidate: colour + BSDF   12 candidate: MIS + reservoir
__device__ unsigned long long g_shade_time[16];
#if LUM_FAST
#define g_vis_stat g_vis_stat_fast
#endif
// visibility rays by kind (0 sampled light, 1 BSDF-sampled light, 2 ambient, 3 sun): [2k] rays, [2k+1] of them blocked by an opaque surface
__device__ unsigned long long g_vis_stat[8];
struct ShadeClock {
  unsigned long long t[16];
  unsigned long long last;
  LUM_DEV void start() { for (int k = 0; k < 16; k++) t[k] = 0; last = __builtin_readcyclecounter(); }
  LUM_DEV void lap(int k) { const unsigned long long now = __builtin_readcyclecounter(); t[k] += now - last; last = now; if (k == 8) t[9]++; }
  LUM_DEV void flush() { if ((threadIdx.x & 63u) == 0u) for (int k = 0; k < 16; k++) if (t[k]) atomicAdd(&g_shade_time[k], t[k]); }
};
#define LUM_LAP(clock, k) (clock).lap(k)
#define LUM_STAT(k_iter, k_lanes) do { const unsigned long long act_ = __ballot(true); if ((threadIdx.x & 63u) == (uint32_t) __builtin_ctzll(act_)) { \
  atomicAdd(&g_phase[k_iter], 1ull); atomicAdd(&g_phase[k_lanes], (unsigned long long) __popcll(act_)); } } while (0)
#else
#define LUM_STAT(k_iter, k_lanes) do {} while (0)
struct ShadeClock { LUM_DEV void start() {} LUM_DEV void lap(int) {} LUM_DEV void flush() {} };
#define LUM_LAP(clock, k) do {} while (0)
#endif

LUM_NS_BEGIN

struct V3 { float x, y, z; };
struct Col { float r, g, b; };
struct F2 { float x, y; };
struct U2 { uint32_t x, y; };
struct Quat { float x, y, z, w; };

constexpr float kPi  = 3.14159265358979323846f;
constexpr float kEps = 1.1920928955078125e-7f;  // FLT_EPSILON (cuda/utils.cuh:41-43)
constexpr float kFltMax = 3.402823466e+38f;

// Streaming accesses of the wavefront queues (experiment LUM_NT_STREAMS): every queue word is written once and read once or twice per depth,
// gigabytes per launch that flow through L2 and the 256 MB Infinity Cache between two ray kernels and evict the scene (nodes + triangles,
// 130 MB on the hall) those kernels gather from. `nt` marks them non-temporal.
#ifndef LUM_NT_STREAMS
#define LUM_NT_STREAMS 1  // measured: +1.7 % (hall), +1.0 % (scan), +1.2 % (example)
#endif
typedef float lum_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t lum_v4u __attribute__((ext_vector_type(4)));
LUM_DEV float4 ld_stream(const float4* p) {
#if LUM_NT_STREAMS
  const lum_v4f v = __builtin_nontemporal_load(reinterpret_cast<const lum_v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
LUM_DEV uint4 ld_stream(const uint4* p) {
#if LUM_NT_STREAMS
  const lum_v4u v = __builtin_nontemporal_load(reinterpret_cast<const lum_v4u*>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *p;
#endif
}
LUM_DEV void st_stream(float4* p, float4 v) {
#if LUM_NT_STREAMS
  lum_v4f w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<lum_v4f*>(p));
#else
  *p = v;
#endif
}
LUM_DEV void st_stream(uint4* p, uint4 v) {
#if LUM_NT_STREAMS
  lum_v4u w; w.x = v.x; w.y = v.y; w.z = v.z; w.w = v.w;
  __builtin_nontemporal_store(w, reinterpret_cast<lum_v4u*>(p));
#else
  *p = v;
#endif
}

LUM_DEV uint32_t fbits(float f) { return __float_as_uint(f); }
LUM_DEV float bitsf(uint32_t u) { return __uint_as_float(u); }

// CUDA's float->u32 conversion saturates; make that explicit.
LUM_DEV uint32_t f2u_sat(float v) {
  if (!(v >= 0.0f)) return 0u;
  if (v >= 4294967296.0f) return 0xFFFFFFFFu;
  return (uint32_t) v;
}
LUM_DEV float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
#if LUM_FAST_RSQ
LUM_DEV float rsqrt_ieee(float x) { return __builtin_amdgcn_rsqf(x); }  // v_rsq_f32, 1 ulp (the reference: rsqrtf under --use_fast_math)
#else
LUM_DEV float rsqrt_ieee(float x) { return 1.0f / sqrtf(x); }
#endif
LUM_DEV float exp2i(int e) { return ldexpf(1.0f, e); }

// sin/cos: Cody-Waite reduction by pi/2, minimax polynomials on [-pi/4, pi/4].
#if LUM_FAST_SINCOS
// fast flavour: the hardware's v_sin_f32 / v_cos_f32 on x / 2pi (what __sinf / __cosf are under --use_fast_math in the reference)
LUM_DEV void sincos_det(float x, float& s_out, float& c_out) {
  const float r = x * 0.15915494309189532f;
  s_out = __builtin_amdgcn_sinf(r);
  c_out = __builtin_amdgcn_cosf(r);
}
#else
LUM_DEV void sincos_det(float x, float& s_out, float& c_out) {
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
  const int q = j & 3;
  s_out = (q == 0) ? s : (q == 1) ? c : (q == 2) ? -s : -c;
  c_out = (q == 0) ? c : (q == 1) ? -s : (q == 2) ? -c : s;
}
#endif
LUM_DEV float atan_pos(float x) {
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
LUM_DEV float atan2_det(float y, float x) {
  if (x != x || y != y) return x + y;
  if (y == 0.0f) return (x < 0.0f || (x == 0.0f && signbit(x))) ? copysignf(kPi, y) : copysignf(0.0f, y);
  if (x == 0.0f) return copysignf(1.5707963267948966f, y);
  const float a = atan_pos(fabsf(y) / fabsf(x));
  const float r = (x < 0.0f) ? (kPi - a) : a;
  return copysignf(r, y);
}

// ---- log2 / exp2 / pow as fixed sequences (relative error < 3e-7), used where the reference calls log2f / powf ----
// log2 for positive normal floats: exponent + 2*atanh((m-1)/(m+1)) / ln 2 with m in [sqrt(1/2), sqrt(2)).
#if LUM_FAST_EXPLOG
LUM_DEV float log2_det(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32
LUM_DEV float exp2_det(float x) { return __builtin_amdgcn_exp2f(fminf(fmaxf(x, -126.0f), 127.0f)); }  // v_exp_f32
#else
LUM_DEV float log2_det(float x) {
  const uint32_t bits = fbits(x);
  int e = (int) ((bits >> 23) & 0xFFu) - 127;
  float m = bitsf((bits & 0x007FFFFFu) | 0x3F800000u);
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
// 2^x for x in [-126, 127]: nearest integer part exactly (ldexp), fraction by the degree-7 Taylor polynomial of exp(f ln 2).
LUM_DEV float exp2_det(float x) {
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
#endif
LUM_DEV float pow_det(float x, float y) { return (x > 0.0f) ? exp2_det(y * log2_det(x)) : 0.0f; }

// ---- vectors (math.cuh:19-218) ----
LUM_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
LUM_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
LUM_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
LUM_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
LUM_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
LUM_DEV V3 vinv(V3 a) { return V3{1.0f / a.x, 1.0f / a.y, 1.0f / a.z}; }
LUM_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
LUM_DEV V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
LUM_DEV float length(V3 a) { return sqrtf(dot(a, a)); }
LUM_DEV V3 normalize(V3 a) { const float s = rsqrt_ieee(dot(a, a)); return V3{a.x * s, a.y * s, a.z * s}; }  // math.cuh:178-186
LUM_DEV V3 reflect(V3 V, V3 n) { const float d = dot(V, n); return normalize(n * (2.0f * d) - V); }            // math.cuh:196-201
LUM_DEV float lerpf(float a, float b, float t) { return a + t * (b - a); }
LUM_DEV float remap01(float v, float lo, float hi) { return saturate((v - lo) / (hi - lo)); }                   // math.cuh:50-56

// ---- colours (math.cuh:800-1070) ----
LUM_DEV Col col(float r, float g, float b) { return Col{r, g, b}; }
LUM_DEV Col splat(float v) { return Col{v, v, v}; }
LUM_DEV Col operator+(Col a, Col b) { return Col{a.r + b.r, a.g + b.g, a.b + b.b}; }
LUM_DEV Col operator-(Col a, Col b) { return Col{a.r - b.r, a.g - b.g, a.b - b.b}; }
LUM_DEV Col operator*(Col a, Col b) { return Col{a.r * b.r, a.g * b.g, a.b * b.b}; }
LUM_DEV Col operator*(Col a, float s) { return Col{a.r * s, a.g * s, a.b * s}; }
LUM_DEV bool any_positive(Col a) { return a.r > 0.0f || a.g > 0.0f || a.b > 0.0f; }
LUM_DEV float luminance(Col v) { return 0.212655f * v.r + 0.715158f * v.g + 0.072187f * v.b; }
// math.cuh:1066-1068 via intrinsics.cuh:91-127: a signed-integer max over the float bit patterns (kept as is).
LUM_DEV float importance(Col c) {
  const int a = __float_as_int(c.r), b = __float_as_int(c.g), d = __float_as_int(c.b);
  return __int_as_float(max(a, max(b, d)));
}

// ---- quaternions and instance transforms (math.cuh:346-489) ----
LUM_DEV Quat qinv(Quat q) { return Quat{-q.x, -q.y, -q.z, q.w}; }
LUM_DEV Quat rotation_to_z(V3 v) {
  if (v.z < -1.0f + kEps) return Quat{1.0f, 0.0f, 0.0f, 0.0f};
  Quat r{v.y, -v.x, 0.0f, 1.0f + v.z};
  const float n = rsqrt_ieee(r.x * r.x + r.y * r.y + r.w * r.w);
  r.x *= n; r.y *= n; r.w *= n;
  return r;
}
LUM_DEV V3 qapply(Quat q, V3 v) {
  const V3 u = v3(q.x, q.y, q.z);
  const float s = q.w;
  const float duv = dot(u, v), duu = dot(u, u);
  const V3 cr = cross(u, v);
  V3 r = u * (2.0f * duv);
  r    = r + v * (s * s - duu);
  r    = r + cr * (2.0f * s);
  return r;
}

struct Transform { V3 translation, scale; uint32_t rot_xy, rot_zw; };  // device_structs.h:295-300 (32 B)

LUM_DEV Quat quat16(const Transform& t) {
  return Quat{((t.rot_xy & 0xFFFFu) * (1.0f / 0x7FFF)) - 1.0f, ((t.rot_xy >> 16) * (1.0f / 0x7FFF)) - 1.0f,
              ((t.rot_zw & 0xFFFFu) * (1.0f / 0x7FFF)) - 1.0f, ((t.rot_zw >> 16) * (1.0f / 0x7FFF)) - 1.0f};
}
LUM_DEV Quat quat16_inv(const Transform& t) {
  return Quat{1.0f - ((t.rot_xy & 0xFFFFu) * (1.0f / 0x7FFF)), 1.0f - ((t.rot_xy >> 16) * (1.0f / 0x7FFF)),
              1.0f - ((t.rot_zw & 0xFFFFu) * (1.0f / 0x7FFF)), ((t.rot_zw >> 16) * (1.0f / 0x7FFF)) - 1.0f};
}
LUM_DEV V3 xf_rot(const Transform& t, V3 v) { return qapply(quat16(t), v); }
LUM_DEV V3 xf_rot_inv(const Transform& t, V3 v) { return qapply(quat16_inv(t), v); }
LUM_DEV V3 xf_rel(const Transform& t, V3 v) { return xf_rot(t, v) * t.scale; }
LUM_DEV V3 xf_rel_inv(const Transform& t, V3 v) { return xf_rot_inv(t, v * vinv(t.scale)); }
LUM_DEV V3 xf_point(const Transform& t, V3 v) { return xf_rel(t, v) + t.translation; }
LUM_DEV V3 xf_point_inv(const Transform& t, V3 v) { return xf_rel_inv(t, v - t.translation); }

// Rows of the world->object matrix used by the ray queries: column j is the inverse-rotated unit vector e_j divided by the scale,
// i.e. the linear part of xf_rel_inv written out once per instance instead of once per ray (the reference hands OptiX a 3x4
// instance matrix for the same purpose, optix_bvh.c:16-66). The object ray is  M * (origin - T),  M * direction  with the fixed
// evaluation order of `mat_row_apply`.
struct Mat3 { V3 r0, r1, r2; };
__host__ __device__ inline float mat_row_apply(float a, float b, float c, float x, float y, float z) { return (a * x + b * y) + c * z; }

// math.cuh:203-214
LUM_DEV F2 barycentric_in_triangle(V3 vertex, V3 e1, V3 e2, V3 p) {
  const V3 d = p - vertex;
  const float d00 = dot(e1, e1), d01 = dot(e1, e2), d11 = dot(e2, e2), d20 = dot(d, e1), d21 = dot(d, e2);
  const float den = 1.0f / (d00 * d11 - d01 * d01);
  return F2{(d11 * d20 - d01 * d21) * den, (d00 * d21 - d01 * d20) * den};
}
// math.cuh:216-228
LUM_DEV V3 lerp_normals(V3 vn, V3 e1n, V3 e2n, F2 c, V3 face) {
  V3 r;
  r.x = vn.x + c.x * e1n.x + c.y * e2n.x;
  r.y = vn.y + c.x * e1n.y + c.y * e2n.y;
  r.z = vn.z + c.x * e1n.z + c.y * e2n.z;
  const float l = length(r);
  return (l < kEps) ? face : r * (1.0f / l);
}
// math.cuh:326-344
LUM_DEV V3 sample_ray_sphere(float alpha, float beta) {
  if (fabsf(alpha) > 1.0f - kEps) return v3(0.0f, 0.0f, copysignf(1.0f, alpha));
  const float a = sqrtf(1.0f - alpha * alpha);
  const float b = 2.0f * kPi * beta;
  float s, c;
  sincos_det(b, s, c);
  return v3(a * c, a * s, alpha);
}
// math.cuh:766-786
LUM_DEV V3 refract(V3 V, V3 n, float index_ratio, bool& total_reflection) {
  if (index_ratio < kEps) { total_reflection = false; return V * -1.0f; }
  const float d = fabsf(dot(n, V));
  const float b = 1.0f - index_ratio * index_ratio * (1.0f - d * d);
  total_reflection = b < 0.0f;
  if (total_reflection) return reflect(V, n);
  return normalize(n * (index_ratio * d - sqrtf(b)) - V * index_ratio);
}
// math.cuh:1337-1358 (Moeller-Trumbore; t < 0 or NaN -> FLT_MAX)
LUM_DEV float intersect_triangle(V3 vertex, V3 e1, V3 e2, V3 origin, V3 ray, F2& coords) {
  const V3 h = cross(ray, e2);
  const float a = dot(e1, h);
  const float f = 1.0f / a;
  const V3 s = origin - vertex;
  const float u = f * dot(s, h);
  const V3 q = cross(s, e1);
  const float v = f * dot(ray, q);
  coords = F2{u, v};
  if (v < 0.0f || u < 0.0f || !(u + v <= 1.0f)) return kFltMax;
  const float t = f * dot(e2, q);
  return (t >= 0.0f) ? t : kFltMax;
}
// math.cuh:1498-1523
LUM_DEV V3 adapt_normal(V3 V, V3 sn, V3 gn) {
  if (dot(sn, gn) < 0.0f) sn = sn * -1.0f;
  if (dot(V, sn) < 0.0f) {
    const V3 proj = V * dot(sn, V);
    return normalize(sn - proj * 1.1f);
  }
  return sn;
}

// ---- packing (math.cuh:1525-1768, medium_stack.cuh:11-29) ----
LUM_DEV float bfloat_unpack(uint32_t v16) { return bitsf((v16 & 0xFFFFu) << 16); }
LUM_DEV Col record_unpack(U2 p) {
  const uint32_t red = p.x & 0x1FFFFFu, green = (p.x >> 21) | ((p.y & 0x3FFu) << 11), blue = p.y >> 10;
  return Col{bitsf(red << 11), bitsf(green << 11), bitsf(blue << 11)};
}
LUM_DEV U2 record_pack(Col r) {
  const uint32_t red = fbits(r.r) >> 11, green = fbits(r.g) >> 11, blue = fbits(r.b) >> 11;
  return U2{red | (green << 21), (green >> 11) | (blue << 10)};
}
LUM_DEV V3 ray_unpack(U2 p) {
  float x = p.x * (1.0f / 4294967296.0f), y = p.y * (1.0f / 4294967296.0f);  // (1.0f / 0xFFFFFFFF) rounds to 2^-32
  x = (x * 2.0f) - 1.0f; y = (y * 2.0f) - 1.0f;
  V3 r = v3(x, y, 1.0f - fabsf(x) - fabsf(y));
  const float t = saturate(-r.z);
  r.x += (r.x >= 0.0f) ? -t : t;
  r.y += (r.y >= 0.0f) ? -t : t;
  return normalize(r);
}
LUM_DEV U2 ray_pack(V3 ray) {
  float x = ray.x, y = ray.y, z = ray.z;
  const float rn = 1.0f / (fabsf(x) + fabsf(y) + fabsf(z));
  x *= rn; y *= rn; z *= rn;
  const float t = saturate(-z);
  x += (x >= 0.0f) ? t : -t;
  y += (y >= 0.0f) ? t : -t;
  x = fminf(1.0f, fmaxf(-1.0f, x)); y = fminf(1.0f, fmaxf(-1.0f, y));
  x = (x + 1.0f) * 0.5f; y = (y + 1.0f) * 0.5f;
  return U2{f2u_sat(x * 4294967296.0f + 0.5f), f2u_sat(y * 4294967296.0f + 0.5f)};
}
LUM_DEV float unorm16(uint32_t d) { return (d & 0xFFFFu) * (1.0f / 0xFFFF); }
LUM_DEV V3 normal_unpack(uint32_t d) {
  float x = (d & 0xFFFFu) * (1.0f / 0xFFFF), y = (d >> 16) * (1.0f / 0xFFFF);
  x = (x * 2.0f) - 1.0f; y = (y * 2.0f) - 1.0f;
  V3 n = v3(x, y, 1.0f - fabsf(x) - fabsf(y));
  const float t = saturate(-n.z);
  n.x += (n.x >= 0.0f) ? -t : t;
  n.y += (n.y >= 0.0f) ? -t : t;
  return normalize(n);
}
LUM_DEV uint32_t normal_pack(V3 n) {
  float x = n.x, y = n.y, z = n.z;
  const float rn = 1.0f / (fabsf(x) + fabsf(y) + fabsf(z));
  x *= rn; y *= rn; z *= rn;
  const float t = fmaxf(fminf(-z, 1.0f), 0.0f);
  x += (x >= 0.0f) ? t : -t;
  y += (y >= 0.0f) ? t : -t;
  x = fmaxf(fminf(x, 1.0f), -1.0f); y = fmaxf(fminf(y, 1.0f), -1.0f);
  x = (x + 1.0f) * 0.5f; y = (y + 1.0f) * 0.5f;
  return (f2u_sat(y * 0xFFFF + 0.5f) << 16) | f2u_sat(x * 0xFFFF + 0.5f);
}
LUM_DEV uint32_t ior_compress(float ior) { return (fbits((0.5f * (ior - 1.0f)) + 1.0f) >> 15) & 0xFFu; }
LUM_DEV float ior_decompress(uint32_t c) { return ((bitsf(0x3F800000u | (c << 15)) - 1.0f) * 2.0f) + 1.0f; }
LUM_DEV float medium_ior_peek(uint32_t stack, bool previous) { return ior_decompress((previous ? stack >> 8 : stack) & 0xFFu); }
LUM_DEV uint32_t medium_ior_modify(uint32_t stack, float ior, bool push) { return push ? ((stack << 8) | ior_compress(ior)) : (stack >> 8); }

LUM_NS_END
