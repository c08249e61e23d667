/*
 * ORACLE (test infrastructure, not product): restatement of the sampler.
 * Follows /root/reference/src/luminary/device/cuda/random.cuh and cuda/utils.cuh:147-178 (PathID).
 * Integer-exact. Pinned by SURVEY.md §0 F9: squares32(0xfcbd6e15, 0..3) = c4dd8039 9a790012 681b4e66 a69b3786,
 * sobol(5,17) = 76a64ec1 aefefe9d (tests/test_oracle_known_answers.py).
 */
#ifndef ORACLE_O_RNG_H
#define ORACLE_O_RNG_H

#include "o_math.h"

/* random.cuh:24-66: START_next = END_prev + 1 with END = START + count * sets (one slot is skipped per allocation). */
enum {
  RT_LENS_METHOD = 0, RT_LENS = 33, RT_LENS_BLADE = 35, RT_LENS_WAVELENGTH = 37,
  RT_BSDF_REFLECTION = 39, RT_BSDF_DIFFUSE = 43, RT_BSDF_REFRACTION = 47, RT_BSDF_RESAMPLING = 51, RT_BSDF_OPACITY = 55,
  RT_VOLUME_INTERSECTION = 59, RT_RUSSIAN_ROULETTE = 61, RT_CAMERA_JITTER = 63, RT_CAMERA_TIME = 65,
  RT_LIGHT_SUN_INITIAL_VERTEX = 344,
  RT_LIGHT_GEO_INITIAL_VERTEX = 358, RT_LIGHT_GEO_RAY = 367, RT_LIGHT_GEO_RESAMPLING = 384,
  RT_LIGHT_GEO_TREE_PREPASS = 387, RT_LIGHT_GEO_TREE_POSTPASS = 404,
  RT_BRIDGE_DISTANCE = 421, RT_BRIDGE_PHASE = 486, RT_BRIDGE_LIGHT_POINT = 551, RT_BRIDGE_VERTEX_COUNT = 560,
  /* the volume context's random sets (material.cuh:76-81): LIGHT_SUN<1>, LIGHT_GEO<1>, BSDF<0> for the bounce, BSDF<2> for the ambient sample;
   * element of set k = base + k * allocation size (random.cuh:72) */
  RT_VOL_SUN_BSDF = 346 + 1, RT_VOL_SUN_BSDF_METHOD = 349 + 1, RT_VOL_SUN_RAY = 352 + 1, RT_VOL_SUN_RESAMPLING = 355 + 1,
  RT_VOL_GEO_RESAMPLING = 384 + 1, RT_VOL_TREE_PREPASS = 387 + 8, RT_VOL_TREE_POSTPASS = 404 + 8,
  RT_VOL_GI_DIFFUSE = 43, RT_VOL_GI_RESAMPLING = 51, RT_VOL_AMBIENT_DIFFUSE = 43 + 2, RT_VOL_AMBIENT_RESAMPLING = 51 + 2,
  RT_LIGHT_BSDF_CHOICE = 569, RT_LIGHT_BSDF_DIRECTION = 571, RT_LIGHT_BSDF_TRACE = 573, RT_LIGHT_BSDF_RR = 575,
  RT_COUNT = 577
};

#define PATH_SENSOR_BITS 14
#define PATH_SAMPLE_BITS 20
#define MAX_GLOBAL_SAMPLES (1u << PATH_SAMPLE_BITS)

typedef struct { uint16_t x, y, z; } PathID; /* device_utils.h:191 (ushort3) */

/* cuda/utils.cuh:147-155 */
static inline PathID path_id_make(uint32_t x, uint32_t y, uint32_t sample_id) {
  const uint32_t extra_bits = 16 - PATH_SENSOR_BITS, sensor_mask = (1u << PATH_SENSOR_BITS) - 1, extra_mask = (1u << extra_bits) - 1;
  PathID p;
  p.x = (uint16_t) ((x & sensor_mask) | (((sample_id >> (16 + extra_bits * 0)) & extra_mask) << PATH_SENSOR_BITS));
  p.y = (uint16_t) ((y & sensor_mask) | (((sample_id >> (16 + extra_bits * 1)) & extra_mask) << PATH_SENSOR_BITS));
  p.z = (uint16_t) (sample_id & 0xFFFFu);
  return p;
}
/* cuda/utils.cuh:157-178 */
static inline void path_id_pixel(PathID p, uint32_t* x, uint32_t* y) {
  *x = p.x & ((1u << PATH_SENSOR_BITS) - 1);
  *y = p.y & ((1u << PATH_SENSOR_BITS) - 1);
}
static inline uint32_t path_id_sample(PathID p) {
  const uint32_t extra_bits = 16 - PATH_SENSOR_BITS;
  uint32_t s = p.z;
  s |= ((uint32_t) p.x >> PATH_SENSOR_BITS) << (16 + extra_bits * 0);
  s |= ((uint32_t) p.y >> PATH_SENSOR_BITS) << (16 + extra_bits * 1);
  return s;
}

static inline uint32_t swap16(uint32_t a) { return (a >> 16) | (a << 16); }
static inline uint32_t brev32(uint32_t v) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
  v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
  return (v >> 16) | (v << 16);
}

/* random.cuh:172-194: Squares counter RNG (Widynski 2020), 4 rounds, 32-bit state variant used by the reference. */
static inline uint32_t squares32(uint32_t key, uint32_t counter) {
  uint32_t x = counter * key, y = counter * key, z = y + key;
  x = x * x + y; x = swap16(x);
  x = x * x + z; x = swap16(x);
  x = x * x + y; x = swap16(x);
  x = x * x + z; z = x; x = swap16(x);
  return z ^ (x * x + y);
}
/* random.cuh:232-239 */
static inline uint32_t lk_perm(uint32_t x, uint32_t seed) {
  x += seed;
  x ^= x * 0x6c50b47cu; x ^= x * 0xb82f1e52u; x ^= x * 0xc7afe638u; x ^= x * 0x8d22f6e6u;
  return x;
}
/* random.cuh:241-246 */
static inline uint32_t owen_scramble(uint32_t x, uint32_t seed) { return brev32(lk_perm(brev32(x), seed)); }
/* random.cuh:248-250 */
static inline uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
/* random.cuh:252-259 */
static inline uint32_t sobol_P(uint32_t v) {
  v ^= v << 16; v ^= (v & 0x00FF00FFu) << 8; v ^= (v & 0x0F0F0F0Fu) << 4; v ^= (v & 0x33333333u) << 2; v ^= (v & 0x55555555u) << 1;
  return v;
}
/* random.cuh:261-287 */
static inline uint2_t rng_sobol(uint32_t offset, uint32_t dimension) {
  const uint32_t seed = squares32(0xfcbd6e15u, dimension);
  const uint32_t J    = lk_perm(brev32(offset), seed);
  uint2_t r;
  r.x = owen_scramble(J, hash_combine(seed, 0));
  r.y = owen_scramble(sobol_P(J), hash_combine(seed, 1));
  return r;
}
/* random.cuh:144-154 */
static inline float u32_to_unit(uint32_t v) { return u2f(0x3F800000u | (v >> 9)) - 1.0f; }
/* random.cuh:163-166 */
static inline float rng_saturate(float r) { return fminf(fmaxf(r, 0.0f), u2f(0x3F7FFFFFu)); }

/* random.cuh:309-333: Sobol + R2-shifted blue-noise mask. `bn` = 256*256 u32. */
static inline uint2_t rng_2d_u32(const uint32_t* bn, uint32_t target, uint32_t px, uint32_t py, uint32_t sample_id, uint32_t depth) {
  const uint32_t dim = target + depth * RT_COUNT;
  uint2_t q = rng_sobol(sample_id, dim);
  const uint32_t ox = (1u + dim) * 3242174889u, oy = (1u + dim) * 2447445413u;
  const uint32_t x = px + (ox >> 24), y = py + (oy >> 24);
  const uint32_t texel = bn[(x & 0xFFu) + (y & 0xFFu) * 256u];
  q.x += texel & 0xFFFF0000u;
  q.y += texel << 16;
  return q;
}

/* Per-path sampler state: pixel, sample id and the depth constant the kernels see (device.state.depth). */
typedef struct { const uint32_t* bn; uint32_t px, py, sample_id, depth; } Sampler;

/* random.cuh:335-368 */
static inline float2_t rnd2(const Sampler* s, uint32_t target) {
  const uint2_t q = rng_2d_u32(s->bn, target, s->px, s->py, s->sample_id, s->depth);
  float2_t r = {u32_to_unit(q.x), u32_to_unit(q.y)};
  return r;
}
static inline float rnd1(const Sampler* s, uint32_t target) {
  return u32_to_unit(rng_2d_u32(s->bn, target, s->px, s->py, s->sample_id, s->depth).x);
}

#endif
