// Defines this translation unit's WavefrontKernels table (include once, after kernels.h and dev_adaptive.h).
#pragma once

#include <vector>

#include "dev_adaptive.h"
#include "wavefront_table.h"

LUM_NS_BEGIN
namespace table {

static int set_ray_kernel_lds(size_t bytes) {
  hipError_t e = hipFuncSetAttribute((const void*) k_trace, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*) k_shadow_rays, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*) k_trace_rays, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*) k_trace_particles, hipFuncAttributeMaxDynamicSharedMemorySize, (int) bytes);
  return (int) e;
}
static int init_sampler_seeds() {
  static std::vector<uint32_t> seeds;  // squares32(0xfcbd6e15, dimension), random.cuh:172-194 - the device function's integer arithmetic on the host
  if (seeds.empty()) {
    seeds.resize(kSeedTableSize);
    auto swap_h = [](uint32_t a) { return (a >> 16) | (a << 16); };
    const uint32_t key = 0xfcbd6e15u;
    for (uint32_t d = 0; d < kSeedTableSize; d++) {
      uint32_t x = d * key, y = d * key, z = y + key;
      x = x * x + y; x = swap_h(x);
      x = x * x + z; x = swap_h(x);
      x = x * x + y; x = swap_h(x);
      x = x * x + z; z = x; x = swap_h(x);
      seeds[d] = z ^ (x * x + y);
    }
  }
  return (int) hipMemcpyToSymbol(HIP_SYMBOL(g_sampler_seeds), seeds.data(), sizeof(uint32_t) * kSeedTableSize);
}
static void sobol_table(hipStream_t s, uint2* table, uint32_t first_sample, uint32_t count, uint32_t stride, uint32_t dims) {
  hipLaunchKernelGGL(k_sobol_table, dim3((dims * stride + 255u) / 256u), dim3(256), 0, s, table, first_sample, count, stride, dims);
}
static void generate(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PassParams& pp, const PathQueue& q, float4* results, uint32_t* count) {
  hipLaunchKernelGGL(k_generate, dim3(grid), dim3(kBlock), 0, s, sc, pp, q, results, count);
}
static void generate_adaptive(uint32_t grid, hipStream_t s, const DeviceScene& sc, const AdaptiveView& a, const AdaptivePass& pass, const PathQueue& q, float4* results,
                              uint32_t* count) {
  hipLaunchKernelGGL(k_generate_adaptive, dim3(grid), dim3(kBlock), 0, s, sc, a, pass, q, results, count);
}
static void trace(uint32_t grid, size_t lds, hipStream_t s, const DeviceScene& sc, const PathQueue& q, const uint32_t* order, uint32_t* ctrl, uint64_t* counters,
                  uint32_t lds_nodes) {
  hipLaunchKernelGGL(k_trace, dim3(grid), dim3(kTraceBlock), lds, s, sc, q, order, ctrl, counters, lds_nodes);
}
static void sky_inscattering(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, float4* results, const uint32_t* ctrl, uint32_t depth_const) {
  hipLaunchKernelGGL(k_sky_inscattering, dim3(grid), dim3(kBlock), 0, s, sc, in, results, ctrl, depth_const);
}
static void shade(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& out, const NeeQueue& nee, const ShadowQueue& sq, float4* results,
                  uint32_t* ctrl, uint32_t depth_const, uint64_t* counters, uint32_t ambient_reuse, const FusedResolve* fused_dev, uint32_t fused_flags) {
#if LUM_SHADE_STAGED
  auto* k1 = sc.sky_mode == kSkyDefault ? k_shade<kSkyDefault, false, 1> : sc.sky_mode == kSkyHdri ? k_shade<kSkyHdri, false, 1> : k_shade<kSkyConstantColor, false, 1>;
  auto* k2 = sc.sky_mode == kSkyDefault ? k_shade<kSkyDefault, false, 2> : sc.sky_mode == kSkyHdri ? k_shade<kSkyHdri, false, 2> : k_shade<kSkyConstantColor, false, 2>;
  if (sc.ocean_active) {
    k1 = sc.sky_mode == kSkyDefault ? k_shade<kSkyDefault, true, 1> : sc.sky_mode == kSkyHdri ? k_shade<kSkyHdri, true, 1> : k_shade<kSkyConstantColor, true, 1>;
    k2 = sc.sky_mode == kSkyDefault ? k_shade<kSkyDefault, true, 2> : sc.sky_mode == kSkyHdri ? k_shade<kSkyHdri, true, 2> : k_shade<kSkyConstantColor, true, 2>;
  }
  hipLaunchKernelGGL(k1, dim3(grid), dim3(kBlock), 0, s, sc, in, out, nee, sq, results, ctrl, depth_const, counters, ambient_reuse, fused_dev, fused_flags);
  hipLaunchKernelGGL(k2, dim3(grid), dim3(kBlock), 0, s, sc, in, out, nee, sq, results, ctrl, depth_const, counters, ambient_reuse, fused_dev, fused_flags);
#else
  auto* k = sc.sky_mode == kSkyDefault ? k_shade<kSkyDefault, false> : sc.sky_mode == kSkyHdri ? k_shade<kSkyHdri, false> : k_shade<kSkyConstantColor, false>;
  if (sc.ocean_active) k = sc.sky_mode == kSkyDefault ? k_shade<kSkyDefault, true> : sc.sky_mode == kSkyHdri ? k_shade<kSkyHdri, true> : k_shade<kSkyConstantColor, true>;
  if (sc.sobol_table) {  // the pass has a Sobol table (dev_sampler.h): the instances that read it instead of hashing
    k = sc.sky_mode == kSkyDefault ? k_shade<kSkyDefault, false, 0, true> : sc.sky_mode == kSkyHdri ? k_shade<kSkyHdri, false, 0, true> : k_shade<kSkyConstantColor, false, 0, true>;
    if (sc.ocean_active) k = sc.sky_mode == kSkyDefault ? k_shade<kSkyDefault, true, 0, true> : sc.sky_mode == kSkyHdri ? k_shade<kSkyHdri, true, 0, true> : k_shade<kSkyConstantColor, true, 0, true>;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(kBlock), 0, s, sc, in, out, nee, sq, results, ctrl, depth_const, counters, ambient_reuse, fused_dev, fused_flags);
#endif
}
static void shade_debug(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, float4* results, const uint32_t* ctrl) {
  hipLaunchKernelGGL(k_shade_debug, dim3(grid), dim3(kBlock), 0, s, sc, in, results, ctrl);
}
static void sky(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const ShadowQueue& sq, float4* results, const uint32_t* ctrl, uint32_t depth_const) {
  hipLaunchKernelGGL(k_sky, dim3(grid), dim3(kBlock), 0, s, sc, in, sq, results, ctrl, depth_const);
}
static void light_query(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, uint32_t* ctrl, uint32_t depth_const,
                        uint64_t* counters) {
  hipLaunchKernelGGL(k_light_query, dim3(grid), dim3(kBlock), 0, s, sc, in, nee, sq, ctrl, depth_const, counters);
}
static void shadow_rays(uint32_t grid, size_t lds, hipStream_t s, const DeviceScene& sc, const ShadowQueue& sq, const uint32_t* order, uint32_t* ctrl, uint64_t* counters,
                        uint32_t lds_nodes) {
  hipLaunchKernelGGL(k_shadow_rays, dim3(grid), dim3(kTraceBlock), lds, s, sc, sq, order, ctrl, counters, lds_nodes);
}
static void resolve(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, float4* results, const uint32_t* ctrl) {
  hipLaunchKernelGGL(k_resolve, dim3(grid), dim3(kBlock), 0, s, sc, in, nee, sq, results, ctrl);
}
static void resolve_reuse(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& next, const NeeQueue& nee, const ShadowQueue& sq, float4* results,
                          uint32_t* ctrl, uint64_t* counters) {
  hipLaunchKernelGGL(k_resolve_reuse, dim3(grid), dim3(kBlock), 0, s, sc, in, next, nee, sq, results, ctrl, counters);
}
static void resolve_listed(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, float4* results, const uint32_t* ctrl) {
  hipLaunchKernelGGL(k_resolve_listed, dim3(grid), dim3(kBlock), 0, s, sc, in, nee, sq, results, ctrl);
}
static void resolve_ended(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, float4* results, const uint32_t* ctrl,
                          const uint32_t* list) {
  hipLaunchKernelGGL(k_resolve_ended, dim3(grid), dim3(kBlock), 0, s, sc, in, nee, sq, results, ctrl, list);
}
static void volume_inscatter(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const VolumeQueue& vq, const ShadowQueue& sq, uint32_t* ctrl,
                             uint32_t depth_const) {
  hipLaunchKernelGGL(k_volume_inscatter, dim3(grid), dim3(kBlock), 0, s, sc, in, vq, sq, ctrl, depth_const);
}
static void volume_resolve(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const VolumeQueue& vq, const ShadowQueue& sq, float4* results,
                           const uint32_t* ctrl) {
  hipLaunchKernelGGL(k_volume_resolve, dim3(grid), dim3(kBlock), 0, s, sc, in, vq, sq, results, ctrl);
}
static void volume_events(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const VolumeQueue& vq, float4* results, uint32_t* ctrl,
                          uint32_t depth_const) {
  hipLaunchKernelGGL(k_volume_events, dim3(grid), dim3(kBlock), 0, s, sc, in, vq, results, ctrl, depth_const);
}
static void volume_bounce(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& out, const VolumeQueue& vq, uint32_t* ctrl,
                          uint32_t depth_const) {
  hipLaunchKernelGGL(k_volume_bounce, dim3(grid), dim3(kBlock), 0, s, sc, in, out, vq, ctrl, depth_const);
}
static void trace_particles(uint32_t grid, size_t lds, hipStream_t s, const DeviceScene& particle_tree, const PathQueue& q, uint32_t* ctrl, uint32_t lds_nodes) {
  hipLaunchKernelGGL(k_trace_particles, dim3(grid), dim3(kTraceBlock), lds, s, particle_tree, q, ctrl, lds_nodes);
}
static void particle_shade(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& out, const NeeQueue& nee, const ShadowQueue& sq,
                           uint32_t* ctrl, uint32_t depth_const) {
  hipLaunchKernelGGL(k_particle_shade, dim3(grid), dim3(kBlock), 0, s, sc, in, out, nee, sq, ctrl, depth_const);
}
static void trace_ocean(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& q, const uint32_t* ctrl) {
  hipLaunchKernelGGL(k_trace_ocean, dim3(grid), dim3(kBlock), 0, s, sc, q, ctrl);
}
static void ocean_shade(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& out, const NeeQueue& nee, const ShadowQueue& sq,
                        uint32_t* ctrl, uint32_t depth_const) {
  hipLaunchKernelGGL(k_ocean_shade, dim3(grid), dim3(kBlock), 0, s, sc, in, out, nee, sq, ctrl, depth_const);
}
static void clouds_list(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const CloudQueue& cq, uint32_t* ctrl) {
  hipLaunchKernelGGL(k_clouds_list, dim3(grid), dim3(kBlock), 0, s, sc, in, cq, ctrl);
}
static void clouds_march(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const CloudQueue& cq, uint32_t* ctrl, uint32_t depth_const) {
  hipLaunchKernelGGL(k_clouds_march, dim3(grid), dim3(kBlock), 0, s, sc, in, cq, ctrl, depth_const);
}
static void clouds(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const CloudQueue& cq, float4* results, const uint32_t* ctrl, uint32_t depth_const) {
  hipLaunchKernelGGL(k_clouds, dim3(grid), dim3(kBlock), 0, s, sc, in, cq, results, ctrl, depth_const);
}
static void trace_rays(uint32_t grid, size_t lds, hipStream_t s, const DeviceScene& sc, uint32_t n, const float* origins, const float* dirs, const uint32_t* ignore, uint32_t* out,
                       uint32_t* cursor, uint64_t* counters, uint32_t lds_nodes) {
  hipLaunchKernelGGL(k_trace_rays, dim3(grid), dim3(kTraceBlock), lds, s, sc, n, origins, dirs, ignore, out, cursor, counters, lds_nodes);
}

static const WavefrontKernels kTable = {LUM_FLAVOUR_NAME, (uint32_t) kTraceBlock, set_ray_kernel_lds, init_sampler_seeds, sobol_table, generate,    generate_adaptive, trace,  sky_inscattering, shade,
                                        shade_debug,      sky,              light_query,        shadow_rays, resolve, resolve_reuse, resolve_listed, resolve_ended, LUM_FAST && !LUM_SHADE_STAGED, volume_inscatter, volume_resolve, volume_events, volume_bounce, trace_particles, particle_shade, trace_ocean, ocean_shade, clouds_list, clouds_march, clouds, trace_rays};

}  // namespace table
LUM_NS_END

#if defined(LUM_PHASE_STATS) && LUM_FAST
// the fast flavour's own counter block (dev_math.h); the exact flavour's is read by lumc_debug_phase_stats (core.hip)
extern "C" int lumc_debug_phase_stats_fast(uint64_t out[16], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(uint64_t) * 16) != hipSuccess) return 1;
  if (reset) { const uint64_t zero[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof(zero)) != hipSuccess) return 1; }
  return 0;
}
extern "C" int lumc_debug_phase_times_fast(uint64_t out[16], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_time), sizeof(uint64_t) * 16) != hipSuccess) return 1;
  if (reset) { const uint64_t zero[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_time), zero, sizeof(zero)) != hipSuccess) return 1; }
  return 0;
}
extern "C" int lumc_debug_vis_stats_fast(uint64_t out[8], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vis_stat), sizeof(uint64_t) * 8) != hipSuccess) return 1;
  if (reset) { const uint64_t zero[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_vis_stat), zero, sizeof(zero)) != hipSuccess) return 1; }
  return 0;
}
extern "C" int lumc_debug_shade_times_fast(uint64_t out[16], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_shade_time), sizeof(uint64_t) * 16) != hipSuccess) return 1;
  if (reset) { const uint64_t zero[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_shade_time), zero, sizeof(zero)) != hipSuccess) return 1; }
  return 0;
}
#endif

namespace lum {
#if LUM_FAST
const WavefrontKernels* wavefront_kernels_fast() { return &fast::table::kTable; }
#else
const WavefrontKernels* wavefront_kernels_exact() { return &exact::table::kTable; }
#endif
}  // namespace lum
