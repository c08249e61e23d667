/*
 * ORACLE (test infrastructure, not product): adaptive sampling and the result image.
 *
 * Restated from cuda/adaptive_sampling.cuh:9-221 (sample bookkeeping, pixel and block variance, stage sample counts),
 * device/device_adaptive_sampler.c:60-215 (stage build), cuda/kernels.cuh:195-355 (which sample ids an execution takes),
 * cuda/accumulation.cuh:86-200 (accumulation_generate_result: beauty with optional local error minimisation, variance, error and
 * sample-distribution images).
 * An execution of stage 0 takes one sample of every pixel; an execution of stage s >= 1 takes count_s(block) samples of every pixel of
 * a 4x4 block (byte s-1 of stage_counts[block], plus one). Samples of a pixel are consecutive ids.
 * Fixed on purpose where the reference is timing- or order-dependent: the total of the block variances is added in chunks of 256
 * blocks and then over the chunk sums (the reference uses a float atomicAdd); when a stage is built is the caller's decision
 * (the product builds it after exactly update_interval << stage executions). Parity unpinned, like the rest of the oracle.
 */
#ifndef ORACLE_O_ADAPTIVE_H
#define ORACLE_O_ADAPTIVE_H

#include "o_output.h"

#define O_ADAPTIVE_BLOCK_LOG 2u /* device_utils.h:32 */
#define O_ADAPTIVE_STAGES 4u    /* device_utils.h:331 */
#define O_ADAPTIVE_MAX_RATE 256u
#define O_ADAPTIVE_SUM_CHUNK 256u

typedef struct {
  const uint32_t* stage_counts; /* per block; NULL = adaptive sampling off (result image only) */
  uint32_t blocks_x, blocks_y;
  uint32_t executions[O_ADAPTIVE_STAGES + 1]; /* completed executions per stage */
  uint32_t stage_id;
} OAdaptive;

static inline uint32_t oa_stage_count(uint32_t packed, uint32_t stage) { return ((packed >> ((stage - 1u) * 8u)) & 0xFFu) + 1u; }
/* adaptive_sampling.cuh:57-105 */
static inline uint32_t oa_pixel_samples(const OAdaptive* a, uint32_t packed) {
  uint32_t n = a->executions[0];
  for (uint32_t s = 1; s <= O_ADAPTIVE_STAGES; s++) n += a->executions[s] * oa_stage_count(packed, s);
  return n;
}
static inline uint32_t oa_block_of(const OAdaptive* a, uint32_t x, uint32_t y) { return (x >> O_ADAPTIVE_BLOCK_LOG) + (y >> O_ADAPTIVE_BLOCK_LOG) * a->blocks_x; }

/* adaptive_sampling.cuh:122-166 */
static inline float oa_pixel_variance(const float* fm, const float* sm, uint32_t num_pixels, uint32_t index, float inv_n, RGBF* mean) {
  const float r1 = fm[index] * inv_n, g1 = fm[num_pixels + index] * inv_n, b1 = fm[2 * (size_t) num_pixels + index] * inv_n;
  *mean = c3(r1, g1, b1);
  const float lum2 = sm[index] * inv_n;
  const float lum_sq = c_luminance(c3(r1 * r1, g1 * g1, b1 * b1));
  return fmaxf(lum2 - lum_sq, 0.0f);
}
/* adaptive_sampling.cuh:9-18 */
static inline float oa_tonemap_compression(const OracleOutputParams* op, RGBF color, float exposure) {
  const RGBF exposed = c_mul(color, c_splat(exposure));
  const RGBF mapped = o_tonemap_curve(op, exposed);
  const float ev = c_luminance(exposed), tv = c_luminance(mapped);
  return (ev > 0.0f) ? tv / ev : 1.0f;
}

/* adaptive_sampling_block_reduce_variance, adaptive_sampling.cuh:168-199 */
static void oa_block_variance(const OAdaptive* a, const OracleOutputParams* op, uint32_t width, uint32_t height, float exposure, const float* fm,
                              const float* sm, float* block_variance) {
  const uint32_t nb = a->blocks_x * a->blocks_y;
  for (uint32_t block = 0; block < nb; block++) {
    const uint32_t by = block / a->blocks_x, bx = block - by * a->blocks_x;
    float best = 0.0f;
    for (uint32_t l = 0; l < 16; l++) {
      const uint32_t x = (bx << O_ADAPTIVE_BLOCK_LOG) + (l & 3u), y = (by << O_ADAPTIVE_BLOCK_LOG) + (l >> 2);
      float variance = 0.0f;
      if (x < width && y < height) {
        const uint32_t n = oa_pixel_samples(a, a->stage_counts[block]);
        const float inv_n = 1.0f / (float) n;
        RGBF mean;
        variance = oa_pixel_variance(fm, sm, width * height, x + y * width, inv_n, &mean);
        if (exposure != 0.0f) {
          const float c = oa_tonemap_compression(op, mean, exposure);
          variance *= c * c;
        }
      }
      best = fmaxf(best, variance);
    }
    block_variance[block] = fabsf(best);
  }
}

static float oa_variance_total(const float* block_variance, uint32_t num_blocks) {
  float total = 0.0f;
  for (uint32_t first = 0; first < num_blocks; first += O_ADAPTIVE_SUM_CHUNK) {
    const uint32_t last = (first + O_ADAPTIVE_SUM_CHUNK < num_blocks) ? first + O_ADAPTIVE_SUM_CHUNK : num_blocks;
    float s = 0.0f;
    for (uint32_t i = first; i < last; i++) s += block_variance[i];
    total += s;
  }
  return total;
}

/* adaptive_sampling_compute_stage_sample_counts, adaptive_sampling.cuh:201-221 */
static void oa_stage_counts(const float* block_variance, float total, uint32_t num_blocks, uint32_t current_stage, uint32_t max_rate, uint32_t avg_rate,
                            uint32_t* stage_counts) {
  const float avg_variance = total / (float) num_blocks;
  for (uint32_t block = 0; block < num_blocks; block++) {
    uint32_t packed = stage_counts[block];
    packed &= (1u << (current_stage * 8u)) - 1u;
    const float mapped = block_variance[block] / avg_variance * (float) avg_rate; /* remap(v, 0, avg, 0, rate), math.cuh:54-56 */
    uint32_t rate = f2u_sat(mapped + 0.5f);
    if (rate < 1u) rate = 1u;
    if (rate > max_rate) rate = max_rate;
    packed |= (rate - 1u) << (current_stage * 8u);
    stage_counts[block] = packed;
  }
}

typedef struct {
  uint32_t width, height, mode, local_error_minimization, uniform_samples;
  float exposure;
} OResultParams;

static inline uint32_t oa_result_samples(const OAdaptive* a, const OResultParams* rp, uint32_t x, uint32_t y) {
  return a->stage_counts ? oa_pixel_samples(a, a->stage_counts[oa_block_of(a, x, y)]) : rp->uniform_samples;
}
static inline uint32_t oa_min(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline uint32_t oa_max(uint32_t a, uint32_t b) { return a > b ? a : b; }
static inline float oa_lerp(float a, float b, float t) { return a + t * (b - a); }
static inline float oa_remap01(float v, float lo, float hi) { return o_saturate((v - lo) / (hi - lo)); }

/* accumulation_generate_result, accumulation.cuh:86-200 */
static void oa_generate_result(const OAdaptive* a, const OResultParams* rp, const OracleOutputParams* op, const float* fm, const float* sm, float* frame_result) {
  const uint32_t n = rp->width * rp->height;
#pragma omp parallel for schedule(static)
  for (int64_t ii = 0; ii < (int64_t) n; ii++) {
    const uint32_t index = (uint32_t) ii;
    const uint32_t y = index / rp->width, x = index - y * rp->width;
    const uint32_t samples = oa_result_samples(a, rp, x, y);
    const float normalization = 1.0f / (float) samples;
    RGBF result;
    switch (rp->mode) {
      default:
      case 0: {
        if (rp->local_error_minimization) {
          RGBF center_mean;
          const float center_variance = oa_pixel_variance(fm, sm, n, index, normalization, &center_mean);
          const float center_error = center_variance * normalization;
          const uint32_t xi_start = oa_max(x, 1u) - 1u, xi_end = oa_min(x, rp->width - 1u) + 1u;
          const uint32_t yi_start = oa_max(y, 1u) - 1u, yi_end = oa_min(y, rp->height - 1u) + 1u;
          RGBF neighbour_mean = c_splat(0.0f);
          float neighbour_error = 0.0f;
          for (uint32_t yi = yi_start; yi <= yi_end; yi++) {
            for (uint32_t xi = xi_start; xi <= xi_end; xi++) {
              if (xi == x && yi == y) continue;
              RGBF m = c_splat(0.0f);
              float variance = 0.0f;
              /* the range runs one past the last row/column; pixels outside the frame contribute zero but count in the divisor */
              const uint32_t ns = oa_result_samples(a, rp, oa_min(xi, rp->width - 1u), oa_min(yi, rp->height - 1u));
              const float norm = 1.0f / (float) ns;
              if (xi < rp->width && yi < rp->height) variance = oa_pixel_variance(fm, sm, n, xi + yi * rp->width, norm, &m);
              neighbour_mean = c_add(neighbour_mean, m);
              neighbour_error += variance * norm;
            }
          }
          const float neighbour_norm = 1.0f / (float) ((xi_end - xi_start + 1u) * (yi_end - yi_start + 1u) - 1u);
          neighbour_mean = c_mul(neighbour_mean, c_splat(neighbour_norm));
          neighbour_error *= neighbour_norm;
          const float t = oa_remap01(center_error, 0.0f, 8.0f * neighbour_error);
          result = c3(oa_lerp(center_mean.r, neighbour_mean.r, t), oa_lerp(center_mean.g, neighbour_mean.g, t), oa_lerp(center_mean.b, neighbour_mean.b, t));
        }
        else result = c3(fm[index] * normalization, fm[n + index] * normalization, fm[2 * (size_t) n + index] * normalization);
      } break;
      case 1: {
        RGBF mean;
        result = c_splat(128.0f * oa_pixel_variance(fm, sm, n, index, normalization, &mean));
      } break;
      case 2: {
        RGBF mean;
        const float variance = oa_pixel_variance(fm, sm, n, index, normalization, &mean);
        const float compression = oa_tonemap_compression(op, mean, rp->exposure);
        const float mse = sqrtf(variance * normalization) * compression;
        const float value = 1024.0f * mse;
        result = c3(o_saturate(2.0f * value), o_saturate(2.0f * (value - 0.5f)),
                    o_saturate((value > 0.5f) ? 4.0f * (0.25f - fabsf(value - 1.0f)) : 4.0f * (0.25f - fabsf(value - 0.25f))));
      } break;
      case 3: {
        uint32_t per_pixel = 1;
        if (a->stage_counts && a->stage_id > 0) per_pixel = oa_stage_count(a->stage_counts[oa_block_of(a, x, y)], a->stage_id);
        result = c_splat((float) per_pixel / (float) O_ADAPTIVE_MAX_RATE);
      } break;
    }
    frame_result[index] = result.r; frame_result[n + index] = result.g; frame_result[2 * (size_t) n + index] = result.b;
  }
}

#endif
