// Sampler: Owen-scrambled Sobol points + R2-shifted blue-noise mask, keyed by the Squares counter RNG.
// Integer-exact restatement of src/luminary/device/cuda/random.cuh (:24-66 target table, :172-194 Squares,
// :232-287 Sobol/Owen, :309-368 blue noise + wrappers) and cuda/utils.cuh:147-178 (PathID is kept unpacked here).
#pragma once

#include "dev_math.h"

LUM_NS_BEGIN

// random.cuh:24-66 - every allocation skips one slot: START_next = START + count * sets + 1.
enum RandomTarget : uint32_t {
  kRndLens = 33, kRndLensBlade = 35,
  kRndBsdfReflection = 39, kRndBsdfDiffuse = 43, kRndBsdfRefraction = 47, kRndBsdfResampling = 51, kRndBsdfOpacity = 55,
  kRndRussianRoulette = 61, kRndCameraJitter = 63,
  kRndLightGeoRay = 367, kRndLightGeoResampling = 384, kRndLightTreePrepass = 387, kRndLightTreePostpass = 404,
  kRndLightBsdfChoice = 569, kRndLightBsdfDirection = 571, kRndLightBsdfTrace = 573, kRndLightBsdfRR = 575,
  kRndTargetCount = 577
};
constexpr uint32_t kMaxGlobalSamples = 1u << 20;  // device_utils.h:37-39

LUM_DEV uint32_t swap_halves(uint32_t a) { return (a >> 16) | (a << 16); }

LUM_DEV uint32_t squares32(uint32_t key, uint32_t counter) {
  uint32_t x = counter * key, y = counter * key, z = y + key;
  x = x * x + y; x = swap_halves(x);
  x = x * x + z; x = swap_halves(x);
  x = x * x + y; x = swap_halves(x);
  x = x * x + z; z = x; x = swap_halves(x);
  return z ^ (x * x + y);
}
LUM_DEV uint32_t laine_karras(uint32_t x, uint32_t seed) {
  x += seed;
  x ^= x * 0x6c50b47cu; x ^= x * 0xb82f1e52u; x ^= x * 0xc7afe638u; x ^= x * 0x8d22f6e6u;
  return x;
}
LUM_DEV uint32_t owen_scramble(uint32_t x, uint32_t seed) { return __brev(laine_karras(__brev(x), seed)); }
LUM_DEV uint32_t hash_combine(uint32_t seed, uint32_t v) { return seed ^ (v + (seed << 6) + (seed >> 2)); }
LUM_DEV uint32_t sobol_second_dim(uint32_t v) {
  v ^= v << 16; v ^= (v & 0x00FF00FFu) << 8; v ^= (v & 0x0F0F0F0Fu) << 4; v ^= (v & 0x33333333u) << 2; v ^= (v & 0x55555555u) << 1;
  return v;
}
// squares32(key, dimension) for every dimension a path can ask for (target + depth * kRndTargetCount, depth < 64): a random number's dimension is
// the same in every lane of every wave, so its seed is one scalar load from this table instead of six 32-bit multiplies (quarter-rate instructions) and
// a dozen other vector instructions per random number and lane - the hash was a twelfth of the shading kernel. Filled per device and flavour by
// init_sampler_seeds() (wavefront_table_impl.h) with the same integer function.
constexpr uint32_t kSeedTableSize = 64u * kRndTargetCount;
__constant__ uint32_t g_sampler_seeds[kSeedTableSize];
LUM_DEV U2 sobol_owen(uint32_t index, uint32_t dimension) {
  const uint32_t seed = g_sampler_seeds[dimension];
  const uint32_t j    = laine_karras(__brev(index), seed);
  return U2{owen_scramble(j, hash_combine(seed, 0)), owen_scramble(sobol_second_dim(j), hash_combine(seed, 1))};
}
LUM_DEV float unit_float(uint32_t v) { return bitsf(0x3F800000u | (v >> 9)) - 1.0f; }
LUM_DEV float clamp_random(float r) { return fminf(fmaxf(r, 0.0f), bitsf(0x3F7FFFFFu)); }

// LUM_SCALAR_SOBOL (round 5): the Sobol / Owen part of a random number - three Laine-Karras hashes, twelve 32-bit multiplies at a quarter of the vector rate
// - depends on (sample id, dimension) only, and the dimension is the same in every lane. The queues stay nearly sorted by sample id through the depths
// (k_generate writes sample-major, the kernels walk and append in queue order), so the lanes of a wave mostly hold ONE sample id: then the whole hash is
// wave-uniform and runs on the scalar unit (s_mul_i32, s_brev_b32 ...), which k_shade leaves five sixths idle (SQ_INSTS_SALU = 0.17 x SQ_INSTS_VALU). A wave
// whose lanes differ takes the vector form. Same integer function either way: the random numbers do not change by a bit.
#ifndef LUM_SCALAR_SOBOL
#define LUM_SCALAR_SOBOL 0  // measured (profiles/r05_ab_experiments.txt): k_shade +0.5 ... +3 % - off
#endif
// The pass's Sobol table (round 5): the Sobol / Owen part of a random number is a function of (sample id, dimension) alone, and a pass of lumc_render holds a few
// dozen consecutive sample ids and (depths x kRndTargetCount) dimensions. k_sobol_table (kernels.h) writes all of them once per pass - a megabyte - and the
// k_shade<..., kTable = true> instances read the pair they want (the lanes of a wave mostly ask for one line) instead of running three Laine-Karras hashes per
// number. Same integer function, so not a bit changes; the exact flavour's parity tests run through the table. Passes without one (the adaptive sampler's, whose
// ids differ by pixel; the undersampling preview; more than kSobolTableMaxSamples ids; LUM_SOBOL_TABLE_RT=0) launch the kTable = false instances, which hold no
// table code. Measured (profiles/r05_ab_experiments.txt): k_shade -4.7 %, +2.2 % samples/s on the hall; one kernel with a run-time branch between the two forms
// gained half of that and lost 3 % when the hash ran.
constexpr uint32_t kSobolTableMaxSamples = 1024u;  // sample ids per pass a table is built for (42 MB at 8 bounces; a depth's slab - what one k_shade launch reads - 4.7 MB);
                                                  // larger passes hash. Round 6: 256 -> 1024, because bench.py's pass is 64 ids x the number of ranks (512 at 8 GPUs)
template <bool kTable>
struct SamplerT {
  const uint32_t* bluenoise;
  uint32_t px, py, sample_id, depth;
  bool uniform = false;  // every active lane of the wave holds this sample id (detect_uniform(); wave-uniform)
  const uint8_t* table = nullptr;  // kTable: this depth's slab of the pass's table, rows of table_row bytes per target (wave-uniform)
  uint32_t table_row = 0, table_lane = 0;  // bytes per row; this lane's byte offset in a row (8 x (sample id - the pass's first))
  LUM_DEV void use_table(const uint2* slab, uint32_t stride, uint32_t first_sample) {
    if (kTable) { table = (const uint8_t*) slab; table_row = stride * 8u; table_lane = (sample_id - first_sample) * 8u; }
  }
  LUM_DEV void detect_uniform() {
#if LUM_SCALAR_SOBOL
    uniform = __ballot(sample_id != (uint32_t) __builtin_amdgcn_readfirstlane((int) sample_id)) == 0ull;
#endif
  }

  LUM_DEV U2 raw2(uint32_t target) const { return raw2_at(target, px, py, depth); }
  LUM_DEV U2 raw2_at(uint32_t target, uint32_t x, uint32_t y, uint32_t d) const {
    const uint32_t dim = target + d * kRndTargetCount;
    U2 q;
#if LUM_SCALAR_SOBOL
    if (uniform) {
      // the asm statement pins the index in a scalar register: without it the compiler folds the two branches into one vector hash of
      // `uniform ? first lane's id : own id`
      uint32_t sid = (uint32_t) __builtin_amdgcn_readfirstlane((int) sample_id);
      asm volatile("" : "+s"(sid));
      q = sobol_owen(sid, (uint32_t) __builtin_amdgcn_readfirstlane((int) dim));  // (the dimension is wave-uniform by construction; say so where a loop counter hides it)
      asm volatile("" : "+s"(q.x), "+s"(q.y));  // ... and the result: the hash between the two statements can only be scalar code
    }
    else
#endif
    if (kTable && d == depth) {
      const uint2 t = *(const uint2*) (table + (size_t) target * table_row + table_lane);
      q = U2{t.x, t.y};
    }
    else
      q = sobol_owen(sample_id, dim);
    const uint32_t ox = (1u + dim) * 3242174889u, oy = (1u + dim) * 2447445413u;
#ifndef LUM_ABLATE_RNG
#define LUM_ABLATE_RNG 0  // measurement only (wrong images): 1 no blue-noise texel fetch, 2 no table / hash for the Sobol pair either - what the random numbers' gathers cost a kernel
#endif
#if LUM_ABLATE_RNG
    const uint32_t texel = (x * 0x9E3779B9u + y * 0x85EBCA6Bu + dim * 0xC2B2AE35u) | 1u;
    if (LUM_ABLATE_RNG & 2) { q.x = texel * 0x27D4EB2Fu; q.y = texel * 0x165667B1u; }
#else
    const uint32_t texel = bluenoise[((x + (ox >> 24)) & 0xFFu) + ((y + (oy >> 24)) & 0xFFu) * 256u];
#endif
    q.x += texel & 0xFFFF0000u;
    q.y += texel << 16;
    return q;
  }
  LUM_DEV F2 next2(uint32_t target) const { const U2 q = raw2(target); return F2{unit_float(q.x), unit_float(q.y)}; }
  LUM_DEV float next1(uint32_t target) const { return unit_float(raw2(target).x); }
};
using Sampler = SamplerT<false>;

LUM_NS_END
