// The wavefront kernels as a table of launchers, one table per arithmetic flavour (flavour.h). The host side (csrc/host/core.hip) picks a
// table per context (lumc_set_flavour) and launches through it; everything in the signatures is a plain layout from dev_scene.h.
// Kernels that produce data both flavours must agree on (BSDF / sky tables, panorama bake), bookkeeping (accumulation, adaptive rates) and
// the display chain exist once, in the exact flavour.
#pragma once

#include <hip/hip_runtime.h>

#include "dev_scene.h"

namespace lum {

struct WavefrontKernels {
  const char* flavour;
  uint32_t trace_block;  // threads per workgroup of the persistent ray kernels (one workgroup per CU)
  // dynamic LDS of the persistent ray kernels (the staged tree top); returns a hipError_t
  int (*set_ray_kernel_lds)(size_t bytes);
  // fills this flavour's table of sampler seeds on the current device (dev_sampler.h); returns a hipError_t
  int (*init_sampler_seeds)();
  // the Sobol / Owen pairs of a pass's sample ids for `dims` dimensions, rows of `stride` entries (dev_sampler.h LUM_SOBOL_TABLE)
  void (*sobol_table)(hipStream_t s, uint2* table, uint32_t first_sample, uint32_t count, uint32_t stride, uint32_t dims);
  void (*generate)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PassParams& pp, const PathQueue& q, float4* results, uint32_t* count);
  void (*generate_adaptive)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const AdaptiveView& a, const AdaptivePass& pass, const PathQueue& q, float4* results,
                            uint32_t* count);
  void (*trace)(uint32_t grid, size_t lds, hipStream_t s, const DeviceScene& sc, const PathQueue& q, const uint32_t* order, uint32_t* ctrl, uint64_t* counters,
                uint32_t lds_nodes);
  void (*sky_inscattering)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, float4* results, const uint32_t* ctrl, uint32_t depth_const);
  void (*shade)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& out, const NeeQueue& nee, const ShadowQueue& sq, float4* results,
                uint32_t* ctrl, uint32_t depth_const, uint64_t* counters, uint32_t ambient_reuse, const FusedResolve* fused_dev, uint32_t fused_flags);  // fused_dev: device memory; flags: 1 resolve the entries' parents, 2 announce the survivors' entries
  void (*shade_debug)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, float4* results, const uint32_t* ctrl);
  void (*sky)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const ShadowQueue& sq, float4* results, const uint32_t* ctrl, uint32_t depth_const);
  void (*light_query)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, uint32_t* ctrl, uint32_t depth_const,
                      uint64_t* counters);
  void (*shadow_rays)(uint32_t grid, size_t lds, hipStream_t s, const DeviceScene& sc, const ShadowQueue& sq, const uint32_t* order, uint32_t* ctrl, uint64_t* counters,
                      uint32_t lds_nodes);
  void (*resolve)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, float4* results, const uint32_t* ctrl);
  // ambient-visibility reuse (kernels.h, above TraceQuery): the resolve of a depth after the NEXT depth's closest-hit pass over `next`, and of the vertices it listed
  void (*resolve_reuse)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& next, const NeeQueue& nee, const ShadowQueue& sq, float4* results,
                        uint32_t* ctrl, uint64_t* counters);
  void (*resolve_listed)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, float4* results, const uint32_t* ctrl);
  void (*resolve_ended)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, float4* results, const uint32_t* ctrl,
                        const uint32_t* list);
  bool fused_resolve;  // k_shade resolves the previous depth's vertices itself when asked to (FusedResolve; not in the staged-shade experiment build)
  // fog (dev_volume.h): light scattered into the rays of a depth, its summation, the scattering events and their bounce
  void (*volume_inscatter)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const VolumeQueue& vq, const ShadowQueue& sq, uint32_t* ctrl,
                           uint32_t depth_const);
  void (*volume_resolve)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const VolumeQueue& vq, const ShadowQueue& sq, float4* results,
                         const uint32_t* ctrl);
  void (*volume_events)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const VolumeQueue& vq, float4* results, uint32_t* ctrl, uint32_t depth_const);
  void (*volume_bounce)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& out, const VolumeQueue& vq, uint32_t* ctrl,
                        uint32_t depth_const);
  // particles (dev_particle.h): the particle pass of the closest-hit kernel (`particle_tree`: the scene with the particle tree in place of the
  // surfaces' tree) and the shading of particle hits
  void (*trace_particles)(uint32_t grid, size_t lds, hipStream_t s, const DeviceScene& particle_tree, const PathQueue& q, uint32_t* ctrl, uint32_t lds_nodes);
  void (*particle_shade)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& out, const NeeQueue& nee, const ShadowQueue& sq,
                         uint32_t* ctrl, uint32_t depth_const);
  // ocean (dev_ocean.h, dev_water.h): the height-field pass of the closest-hit kernel and the shading of water-surface hits
  void (*trace_ocean)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& q, const uint32_t* ctrl);
  void (*ocean_shade)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const PathQueue& out, const NeeQueue& nee, const ShadowQueue& sq, uint32_t* ctrl,
                      uint32_t depth_const);
  // clouds (dev_cloud.h, dev_cloud_march.h): the march through the cloud layers, sky mode DEFAULT only
  void (*clouds_list)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const CloudQueue& cq, uint32_t* ctrl);
  void (*clouds_march)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const CloudQueue& cq, uint32_t* ctrl, uint32_t depth_const);
  void (*clouds)(uint32_t grid, hipStream_t s, const DeviceScene& sc, const PathQueue& in, const CloudQueue& cq, float4* results, const uint32_t* ctrl, uint32_t depth_const);
  void (*trace_rays)(uint32_t grid, size_t lds, hipStream_t s, const DeviceScene& sc, uint32_t n, const float* origins, const float* dirs, const uint32_t* ignore, uint32_t* out,
                     uint32_t* cursor, uint64_t* counters, uint32_t lds_nodes);
};

const WavefrontKernels* wavefront_kernels_exact();  // csrc/host/core.hip
const WavefrontKernels* wavefront_kernels_fast();   // csrc/device/wavefront_fast.hip

}  // namespace lum
