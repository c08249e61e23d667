// Adaptive sampling: per 4x4-pixel block sample rates chosen in up to four refinement stages from the measured
// variance, and the result image with its diagnostic modes.
// Reference: cuda/adaptive_sampling.cuh:9-240 (sample bookkeeping, block variance, stage sample counts),
// device/device_adaptive_sampler.c:60-215 (stage build), device/device_renderer.c:350-375 (when a stage is built),
// cuda/kernels.cuh:195-355 (tasks_create_adaptive_sampling), cuda/accumulation.cuh:86-200 (accumulation_generate_result).
//
// Bookkeeping (DeviceSampleAllocation, device_utils.h:331-338): an *execution* of stage 0 takes one sample of every pixel; an execution
// of stage s >= 1 takes count_s(block) samples of every pixel of a block, count_s - 1 being byte s-1 of stage_counts[block]. A pixel's
// samples are consecutive ids, so the samples it has received so far are also the id of its next one.
//
// What differs from the reference, on purpose: the reference adds the block variances with a float atomicAdd (order unspecified) and
// builds a stage asynchronously (the number of executions per stage depends on timing). Here the sum is a fixed two-level sequential
// order (chunks of 256 blocks, then the chunk sums) and a stage is built exactly after `update_interval << stage` executions, so the
// oracle reproduces every count bit for bit. The reference's task-range machinery (prefix mips, tile block ranges) exists for its
// per-thread task layout and has no counterpart: tasks are addressed through one inclusive prefix sum over blocks.
#pragma once

#include "dev_output.h"
#include "kernels.h"

LUM_NS_BEGIN

constexpr uint32_t kAdaptiveBlockLog = 2;        // ADAPTIVE_SAMPLING_BLOCK_SIZE_LOG, device_utils.h:32
constexpr uint32_t kAdaptiveMaxRate = 256;       // ADAPTIVE_SAMPLING_MAX_SAMPLING_RATE, device_utils.h:35
constexpr uint32_t kAdaptiveSumChunk = 256;      // blocks per partial sum of the variance total

// samples per pixel and execution of `stage`: one in stage 0, the block's rate in stages 1..4
LUM_DEV uint32_t adaptive_stage_count(uint32_t packed, uint32_t stage) { return stage ? ((packed >> ((stage - 1u) * 8u)) & 0xFFu) + 1u : 1u; }
// adaptive_sampling.cuh:57-76 / :78-105: samples a pixel of this block has received = id of its next sample
LUM_DEV uint32_t adaptive_pixel_samples(const AdaptiveView& a, uint32_t packed) {
  uint32_t n = a.executions[0];
#pragma unroll
  for (uint32_t s = 1; s <= kAdaptiveStages; s++) n += a.executions[s] * (((packed >> ((s - 1u) * 8u)) & 0xFFu) + 1u);
  return n;
}
LUM_DEV uint32_t adaptive_block_of(const AdaptiveView& a, uint32_t x, uint32_t y) { return (x >> kAdaptiveBlockLog) + (y >> kAdaptiveBlockLog) * a.blocks_x; }

// adaptive_sampling.cuh:122-166
LUM_DEV float adaptive_pixel_variance(const float* __restrict__ fm, const float* __restrict__ sm, uint32_t num_pixels, uint32_t index, float inv_n, Col& mean) {
  const float r1 = fm[index] * inv_n, g1 = fm[num_pixels + index] * inv_n, b1 = fm[2 * num_pixels + index] * inv_n;
  mean = col(r1, g1, b1);
  const float lum2 = sm[index] * inv_n;
  const float lum_sq = luminance(col(r1 * r1, g1 * g1, b1 * b1));
  return fmaxf(lum2 - lum_sq, 0.0f);
}
// adaptive_sampling.cuh:9-18
LUM_DEV float adaptive_tonemap_compression(const OutputParams& op, Col color, float exposure) {
  const Col exposed = color * exposure;
  const Col mapped = tonemap_curve(op, exposed);
  const float ev = luminance(exposed), tv = luminance(mapped);
  return (ev > 0.0f) ? tv / ev : 1.0f;
}

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// adaptive_sampling_block_reduce_variance (adaptive_sampling.cuh:168-199): 16 lanes per block, four blocks per wave.
__global__ __launch_bounds__(256) void k_adaptive_block_variance(AdaptiveView a, OutputParams op, uint32_t width, uint32_t height, float exposure,
                                                                const float* __restrict__ fm, const float* __restrict__ sm, float* __restrict__ block_variance) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t block = t >> 4;
  float variance = 0.0f;
  if (block < a.num_blocks) {
    const uint32_t by = block / a.blocks_x, bx = block - by * a.blocks_x;
    const uint32_t x = (bx << kAdaptiveBlockLog) + (t & 3u), y = (by << kAdaptiveBlockLog) + ((t >> 2) & 3u);
    if (x < width && y < height) {
      const uint32_t n = adaptive_pixel_samples(a, a.stage_counts[block]);
      const float inv_n = 1.0f / (float) n;
      Col mean;
      variance = adaptive_pixel_variance(fm, sm, width * height, x + y * width, inv_n, mean);
      if (exposure != 0.0f) {
        const float c = adaptive_tonemap_compression(op, mean, exposure);
        variance *= c * c;
      }
    }
  }
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) variance = fmaxf(variance, __shfl_xor(variance, off, 16));
  if ((t & 15u) == 0u && block < a.num_blocks) block_variance[block] = fabsf(variance);
}
#endif

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// Total of the block variances in a fixed order (see the header): one thread per chunk, then one thread over the chunk sums.
__global__ __launch_bounds__(64) void k_adaptive_sum_chunks(const float* __restrict__ block_variance, uint32_t num_blocks, float* __restrict__ partial) {
  const uint32_t c = blockIdx.x * 64u + threadIdx.x;
  const uint32_t first = c * kAdaptiveSumChunk;
  if (first >= num_blocks) return;
  const uint32_t last = min(first + kAdaptiveSumChunk, num_blocks);
  float s = 0.0f;
  for (uint32_t i = first; i < last; i++) s += block_variance[i];
  partial[c] = s;
}
#endif
#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
__global__ void k_adaptive_sum_total(const float* __restrict__ partial, uint32_t num_chunks, float* __restrict__ total) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  float s = 0.0f;
  for (uint32_t i = 0; i < num_chunks; i++) s += partial[i];
  *total = s;
}
#endif

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// adaptive_sampling_compute_stage_sample_counts (adaptive_sampling.cuh:201-221): the rate of the stage after `current_stage`.
// Also writes the tasks of that stage per block (16 pixels x rate) for the prefix sum.
__global__ __launch_bounds__(256) void k_adaptive_stage_counts(const float* __restrict__ block_variance, const float* __restrict__ total, uint32_t num_blocks,
                                                              uint32_t current_stage, uint32_t max_rate, uint32_t avg_rate, uint32_t* __restrict__ stage_counts,
                                                              uint32_t* __restrict__ block_tasks, const uint8_t* __restrict__ block_mask) {
  const uint32_t block = blockIdx.x * 256u + threadIdx.x;
  if (block >= num_blocks) return;
  const float avg_variance = *total / (float) num_blocks;
  const float variance = block_variance[block];
  uint32_t packed = stage_counts[block];
  packed &= (1u << (current_stage * 8u)) - 1u;  // keep the bytes of the stages already run
  // remap(variance, 0, avg_variance, 0, avg_rate), math.cuh:54-56; a NaN (0/0) converts to 0 as on the reference's hardware
  const float mapped = variance / avg_variance * (float) avg_rate;
  uint32_t rate = f2u_sat(mapped + 0.5f);
  rate = max(rate, 1u);
  rate = min(rate, max_rate);
  packed |= (rate - 1u) << (current_stage * 8u);
  stage_counts[block] = packed;
  // image-tile partition over GPUs: every rank knows every block's rate, but only creates tasks for the blocks it owns
  block_tasks[block] = (!block_mask || block_mask[block]) ? rate << (2u * kAdaptiveBlockLog) : 0u;
}
#endif
#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// Tasks per block of a stage-0 execution under a partition (one sample per pixel of the owned blocks).
__global__ __launch_bounds__(256) void k_adaptive_uniform_tasks(const uint8_t* __restrict__ block_mask, uint32_t num_blocks, uint32_t* __restrict__ block_tasks) {
  const uint32_t block = blockIdx.x * 256u + threadIdx.x;
  if (block < num_blocks) block_tasks[block] = block_mask[block] ? 1u << (2u * kAdaptiveBlockLog) : 0u;
}
#endif

// tasks_create_adaptive_sampling (cuda/kernels.cuh:195-355): task -> (block, pixel of the block, sample of this execution).
// Result slot = task id; paths are appended compacted (tasks outside the frame or beyond the last sample id create nothing).
// One pass covers the tasks [task_begin, task_end) = all tasks of the blocks [block_begin, block_end); slots are relative to task_begin.
// `executions` consecutive executions of the stage share the pass: a pixel then takes executions * rate consecutive sample ids, which
// are added in the same order as one execution after the other would add them. Task numbers are in units of the merged pass
// (block_task_end * executions).
// (struct AdaptivePass: dev_scene.h)

__global__ __launch_bounds__(256) void k_generate_adaptive(DeviceScene sc, AdaptiveView a, AdaptivePass pass, PathQueue q, float4* results, uint32_t* count) {
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t pass_tasks = pass.task_end - pass.task_begin;
  const uint32_t rounds = (pass_tasks + gridDim.x * 256u - 1u) / (gridDim.x * 256u);
  for (uint32_t round = 0; round < rounds; round++) {
    const uint32_t slot = (round * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
    const uint32_t t = pass.task_begin + slot;
    bool valid = false;
    uint32_t x = 0, y = 0, sample_id = 0;
    if (slot < pass_tasks) {
      // adaptive_sampling_find_block (adaptive_sampling.cuh:24-47): first block whose end lies beyond the task
      uint32_t lo = pass.block_begin, hi = pass.block_end - 1u;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (t < a.block_task_end[mid] * pass.executions) hi = mid; else lo = mid + 1u;
      }
      const uint32_t block = lo;
      const uint32_t base = block ? a.block_task_end[block - 1u] * pass.executions : 0u;
      const uint32_t packed = a.stage_counts[block];
      const uint32_t per_pixel = adaptive_stage_count(packed, a.stage_id) * pass.executions;
      const uint32_t local = t - base;
      const uint32_t local_pixel = local / per_pixel, local_sample = local - local_pixel * per_pixel;
      const uint32_t by = block / a.blocks_x, bx = block - by * a.blocks_x;
      x = (bx << kAdaptiveBlockLog) + (local_pixel & 3u);
      y = (by << kAdaptiveBlockLog) + (local_pixel >> kAdaptiveBlockLog);
      sample_id = adaptive_pixel_samples(a, packed) + local_sample;
      valid = x < sc.width && y < sc.height && sample_id < kMaxGlobalSamples;
      results[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    const unsigned long long ballot = __ballot(valid);
    if (ballot) {
      uint32_t base = 0;
      if (lane == (uint32_t) __builtin_ctzll(ballot)) base = atomicAdd(count, (uint32_t) __popcll(ballot));
      base = __shfl(base, __builtin_ctzll(ballot));
      if (valid) {
        const uint32_t i = base + (uint32_t) __popcll(ballot & below);
        const Sampler smp{sc.bluenoise_2d, x, y, sample_id, 0};
        V3 o, d;
        camera_ray(sc, smp, o, d);
        const U2 rec = record_pack(splat(1.0f));
        q.origin_t[i] = make_float4(o.x, o.y, o.z, kFltMax);
        q.dir_slot[i] = make_float4(d.x, d.y, d.z, bitsf(slot));
        q.aux[i]      = make_uint4(rec.x, rec.y, initial_medium(sc, o), kStDeltaPath | kStCameraDirection | kStAllowEmission | kStAllowAmbient);
        q.hit_id[i]   = make_uint4(0u, 0u, x | (y << 16), initial_volumes(sc, o, sample_id));
      }
    }
  }
}

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// accumulation_collect_results for one adaptive execution: a pixel's samples of this execution are added in sample order.
__global__ __launch_bounds__(256) void k_accumulate_adaptive(AdaptiveView a, AdaptivePass pass, uint32_t width, uint32_t height, const float4* __restrict__ results,
                                                            float* first_moment, float* second_moment) {
  const uint32_t num_pixels = width * height;
  const uint32_t pass_pixels = (pass.block_end - pass.block_begin) << (2u * kAdaptiveBlockLog);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < pass_pixels; i += gridDim.x * 256u) {
    const uint32_t block = pass.block_begin + (i >> (2u * kAdaptiveBlockLog)), local = i & 15u;
    const uint32_t by = block / a.blocks_x, bx = block - by * a.blocks_x;
    const uint32_t x = (bx << kAdaptiveBlockLog) + (local & 3u), y = (by << kAdaptiveBlockLog) + (local >> kAdaptiveBlockLog);
    if (x >= width || y >= height) continue;
    const uint32_t p = x + y * width;
    const uint32_t packed = a.stage_counts[block];
    const uint32_t per_pixel = adaptive_stage_count(packed, a.stage_id) * pass.executions;
    const uint32_t first_id = adaptive_pixel_samples(a, packed);
    const uint32_t block_begin = block ? a.block_task_end[block - 1u] : 0u;
    if (a.block_task_end[block] == block_begin) continue;  // a block of another GPU's tiles: no tasks here
    const uint32_t base = block_begin * pass.executions - pass.task_begin + local * per_pixel;
    float r = first_moment[p], g = first_moment[num_pixels + p], b = first_moment[2 * num_pixels + p];
    float s = second_moment[p];
    for (uint32_t k = 0; k < per_pixel; k++) {
      if (first_id + k >= kMaxGlobalSamples) break;
      const float4 v = results[base + k];
      r += v.x; g += v.y; b += v.z;
      s += luminance(col(v.x * v.x, v.y * v.y, v.z * v.z));
    }
    first_moment[p] = r; first_moment[num_pixels + p] = g; first_moment[2 * num_pixels + p] = b;
    second_moment[p] = s;
  }
}
#endif

// accumulation_generate_result (cuda/accumulation.cuh:86-200): mean radiance (optionally with local error minimisation) or one of
// the diagnostic images. With a null stage_counts every pixel has `uniform_samples` samples (adaptive sampling off).
struct ResultParams {
  uint32_t width, height;
  uint32_t mode;            // LuminaryAdaptiveSamplingOutputMode: 0 beauty, 1 variance, 2 error, 3 sample distribution
  uint32_t local_error_minimization;
  uint32_t uniform_samples;
  float exposure;           // camera exposure (error mode)
};

LUM_DEV uint32_t result_pixel_samples(const AdaptiveView& a, const ResultParams& rp, uint32_t x, uint32_t y) {
  return a.stage_counts ? adaptive_pixel_samples(a, a.stage_counts[adaptive_block_of(a, x, y)]) : rp.uniform_samples;
}

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
__global__ __launch_bounds__(256) void k_generate_result(AdaptiveView a, ResultParams rp, OutputParams op, const float* __restrict__ fm, const float* __restrict__ sm,
                                                        float* __restrict__ frame_result) {
  const uint32_t n = rp.width * rp.height;
  for (uint32_t index = blockIdx.x * 256u + threadIdx.x; index < n; index += gridDim.x * 256u) {
    const uint32_t y = index / rp.width, x = index - y * rp.width;
    const uint32_t samples = result_pixel_samples(a, rp, x, y);
    const float normalization = 1.0f / (float) samples;
    Col result;
    switch (rp.mode) {
      default:
      case 0: {
        if (rp.local_error_minimization) {
          Col center_mean;
          const float center_variance = adaptive_pixel_variance(fm, sm, n, index, normalization, center_mean);
          const float center_error = center_variance * normalization;
          const uint32_t xi_start = max(x, 1u) - 1u, xi_end = min(x, rp.width - 1u) + 1u;
          const uint32_t yi_start = max(y, 1u) - 1u, yi_end = min(y, rp.height - 1u) + 1u;
          Col neighbour_mean = splat(0.0f);
          float neighbour_error = 0.0f;
          for (uint32_t yi = yi_start; yi <= yi_end; yi++) {
            for (uint32_t xi = xi_start; xi <= xi_end; xi++) {
              if (xi == x && yi == y) continue;
              Col m = splat(0.0f);
              float variance = 0.0f, norm = 0.0f;
              // the reference's range runs one past the last row/column; pixels outside the frame contribute zero
              // (adaptive_sampling.cuh:146-151) but still count in the divisor below
              const uint32_t ns = (xi < rp.width && yi < rp.height) ? result_pixel_samples(a, rp, xi, yi) : result_pixel_samples(a, rp, min(xi, rp.width - 1u), min(yi, rp.height - 1u));
              norm = 1.0f / (float) ns;
              if (xi < rp.width && yi < rp.height) variance = adaptive_pixel_variance(fm, sm, n, xi + yi * rp.width, norm, m);
              neighbour_mean = neighbour_mean + m;
              neighbour_error += variance * norm;
            }
          }
          const float neighbour_norm = 1.0f / (float) ((xi_end - xi_start + 1u) * (yi_end - yi_start + 1u) - 1u);
          neighbour_mean = neighbour_mean * neighbour_norm;
          neighbour_error *= neighbour_norm;
          const float t = remap01(center_error, 0.0f, 8.0f * neighbour_error);
          result = col(lerpf(center_mean.r, neighbour_mean.r, t), lerpf(center_mean.g, neighbour_mean.g, t), lerpf(center_mean.b, neighbour_mean.b, t));
        }
        else result = col(fm[index] * normalization, fm[n + index] * normalization, fm[2 * n + index] * normalization);
      } break;
      case 1: {
        Col mean;
        result = splat(128.0f * adaptive_pixel_variance(fm, sm, n, index, normalization, mean));
      } break;
      case 2: {
        Col mean;
        const float variance = adaptive_pixel_variance(fm, sm, n, index, normalization, mean);
        const float compression = adaptive_tonemap_compression(op, mean, rp.exposure);
        const float mse = sqrtf(variance * normalization) * compression;
        const float value = 1024.0f * mse;
        result = col(saturate(2.0f * value), saturate(2.0f * (value - 0.5f)),
                     saturate((value > 0.5f) ? 4.0f * (0.25f - fabsf(value - 1.0f)) : 4.0f * (0.25f - fabsf(value - 0.25f))));
      } break;
      case 3: {
        // adaptive_sampling_get_current_tasks_per_pixel (adaptive_sampling.cuh:107-120)
        uint32_t per_pixel = 1;
        if (a.stage_counts && a.stage_id > 0) per_pixel = adaptive_stage_count(a.stage_counts[adaptive_block_of(a, x, y)], a.stage_id);
        result = splat((float) per_pixel / (float) kAdaptiveMaxRate);
      } break;
    }
    frame_result[index] = result.r; frame_result[n + index] = result.g; frame_result[2 * n + index] = result.b;
  }
}
#endif

LUM_NS_END
