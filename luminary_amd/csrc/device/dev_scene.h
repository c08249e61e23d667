// Layout of the scene as the kernels see it (all pointers are HBM addresses).
// Encodings follow the reference's device formats so that host conversion code and kernels agree with it:
//   vertices       device_structs.h:270-273 (16 B: position + oct-packed normal)
//   tri_tex        device_structs.h:275-281 (16 B: 3 packed UVs + material id)
//   materials      device_structs.h:202-223 (32 B)
//   transforms     device_structs.h:295-300 (32 B)
//   light tree     device_utils.h:283-327
// Added by this implementation: the BVH4 arrays that replace the OptiX acceleration structures (optix_bvh.c).
#pragma once

#include <hip/hip_vector_types.h>
#include <stdint.h>

namespace lum {

// One 128-byte BVH4 node = one L2 cache line. Child boxes are stored SoA so that a lane reads them with 8 x 16-byte loads.
struct alignas(128) Bvh4Node {
  float lo_x[4], lo_y[4], lo_z[4];
  float hi_x[4], hi_y[4], hi_z[4];
  uint32_t child[4];  // kBvhEmpty | leaf: kBvhLeafBit | (count-1) << 28 | first primitive | inner: node index
  uint32_t pad[4];
};
static_assert(sizeof(Bvh4Node) == 128, "BVH4 node must be one cache line");

// 8-wide node with quantised child boxes, the format of the scene's and the particles' trees when LUM_BVH8 is on (the light tree keeps 4-wide
// nodes). Same 128-byte slot, same child words. Child j's box = origin + q * 2^(e - 127) per axis, lower corners rounded down, upper up.
// Experiment, measured negative and therefore off (profiles/r02_ab_experiments.txt): node visits per ray fall by 22 % (hall 16.4 -> 12.7), but a visit
// costs 2.8 x the VALU work (48 byte->float conversions, a 19-comparator sort) and the ray kernels are as much issue- as latency-limited at 4 waves
// per SIMD: closest-hit +9 %, visibility +10 % time on the hall. Parity tests pass with it on.
#ifndef LUM_BVH8
#define LUM_BVH8 0
#endif
struct alignas(128) Bvh8Node {
  float origin[3];
  uint8_t exp[3], pad0;   // biased exponents of the per-axis scale
  uint32_t child[8];
  uint8_t lo_x[8], lo_y[8], lo_z[8], hi_x[8], hi_y[8], hi_z[8];
  uint32_t pad[8];
};
static_assert(sizeof(Bvh8Node) == 128, "BVH8 node must be one cache line");

// 8-wide node with its children in OCTANT SLOTS (LUM_BVH8O; after Ylitie, Karras, Laine: "Efficient Incoherent Ray Traversal on GPUs Through Compressed
// Wide BVHs", HPG 2017 - the format of the reference's own, unused software traversal: node src/luminary/utils.h:123-138, slot assignment bvh.c:1093-1145,
// traversal order cuda/bvh.cuh:82-106). The builder puts the child whose centre lies towards octant direction s (bit a set: the +a side of the node's
// centre) into slot s; a ray whose direction signs are `oct` (bit a set: d_a < 0) meets the slots roughly front to back in the order s ^ oct = 0 ... 7, so
// a visit needs no distances, no sort and ONE stack entry (the group of children still to be walked) instead of up to seven. 80 bytes of a 128-byte slot:
//   16 B  origin xyz | biased exponents of the per-axis scale x, y, z | imask (bit s: slot s holds an inner node)
//   16 B  child_base (the inner children are consecutive nodes, in slot order) | leaf_base | meta[8]: leaf slot s = offset (5 bits) | count - 1 (2 bits):
//         its primitives are leaf_base + offset ... (the builder reorders the triangles / top-level leaf records node by node); other slots 0
//   48 B  lo_x[8] lo_y[8] | lo_z[8] hi_x[8] | hi_y[8] hi_z[8]: one byte per slot, child box = origin + q * 2^(e - 127), rounded outwards; empty slots hold
//         inverted boxes (lo 255, hi 0) and are never entered
#ifndef LUM_BVH8O
#define LUM_BVH8O 0
#endif
struct alignas(128) Bvh8oNode {
  float origin[3];
  uint8_t exp[3], imask;
  uint32_t child_base, leaf_base;
  uint8_t meta[8];
  uint8_t lo_x[8], lo_y[8], lo_z[8], hi_x[8], hi_y[8], hi_z[8];
  uint32_t pad[12];
};
static_assert(sizeof(Bvh8oNode) == 128, "one slot");

// 4-wide node with quantised child boxes in half a cache line (LUM_BVH4Q): same tree, same node indices, same child words as Bvh4Node; child j's box =
// origin + q * 2^(e - 127) per axis, lower corners rounded down, upper up (the boxes only grow). A visit fetches 64 instead of 112 bytes and the LDS
// holds twice the nodes; it pays with 24 byte->float conversions. Scene and particle trees; the light tree keeps float boxes.
#ifndef LUM_BVH4Q
#define LUM_BVH4Q 0
#endif
struct alignas(64) Bvh4QNode {
  float origin[3];
  uint8_t exp[3], pad0;   // biased exponents of the per-axis scale
  uint32_t child[4];
  uint8_t lo_x[4], lo_y[4], lo_z[4], hi_x[4], hi_y[4], hi_z[4];
  uint32_t pad[2];
};
static_assert(sizeof(Bvh4QNode) == 64, "quantised BVH4 node must be half a cache line");
constexpr uint32_t kNodeBytes = LUM_BVH4Q ? 64u : 128u, kNodeShift = LUM_BVH4Q ? 6u : 7u;

constexpr uint32_t kBvhEmpty      = 0xFFFFFFFFu;
constexpr uint32_t kBvhLeafBit    = 0x80000000u;
// Most triangles of a leaf: the builders' limit AND the size of the ray kernels' leaf registers (3 x float4 per slot). Rounds 1-5: 4. Round 6: 2 - the
// CPU model (tools/bvh_quality.cpp, hall) says 8.1 -> 4.6 triangle tests per closest-hit ray for 17.3 -> 18.2 node visits, and the kernels shed the 24
// registers of two slots: k_trace 128 VGPRs + 12 spilled -> 114, none spilled; k_shadow_rays 128 + 4 -> 110. With the optimal 4-wide collapse (bvh_build.cpp
// CollapsePlan), same box: hall k_trace -3.7 %, k_shadow_rays -3.2 %, +1.75 % samples/s; scan +1.7 %; Example-class +3.7 % (profiles/r06_ab_experiments.txt).
// Leaves of two with registers for four (LUM_BVH_MAX_LEAF=2 on the old build) gain nothing: the registers are what pays.
#ifndef LUM_LEAF_MAX
#define LUM_LEAF_MAX 2
#endif
constexpr uint32_t kBvhLeafMaxTri = LUM_LEAF_MAX;
static_assert(LUM_LEAF_MAX >= 1 && LUM_LEAF_MAX <= 8, "a leaf's triangle count travels in 3 bits");

// Triangle in traversal order, 48 bytes: v0.xyz + id | e1.xyz + scene index | e2.xyz (edges precomputed with the same float
// subtraction the reference's intersection code performs, so hit distances are identical). `id` is the triangle id inside its
// mesh (what hits report), `scene_index` = mesh_tri_offset[mesh] + id indexes vertices/tri_tex directly.
struct BvhTri { float p0[3]; uint32_t id; float e1[3]; uint32_t scene_index; float e2[3]; uint32_t albedo_tex; };  // albedo_tex: texture id of the
// triangle's material or kBvhTriNoTexture: the ray queries consult the texture's alpha without touching the material first
constexpr uint32_t kBvhTriNoTexture = 0xFFFFFFFFu;
// ... and this value marks a triangle whose material has no albedo texture and alpha 1: a visibility ray that crosses it is blocked, which the any-hit
// test then knows from the triangle's own 48 bytes (k_tri_opacity writes it at scene upload; without it every crossing costs two more dependent
// fetches - the triangle's material id, then the material - before the ray may stop)
constexpr uint32_t kBvhTriOpaque = 0xFFFFFFFEu;
static_assert(sizeof(BvhTri) == 48, "48 bytes per triangle");

#ifndef LUM_PHASE_QUEUES
#define LUM_PHASE_QUEUES 0  // 1: k_trace / k_shadow_rays keep their rays in per-wave LDS pools and regroup them by phase (dev_trace_pool.h)
#endif

struct DeviceScene {
  // geometry
  const uint32_t* mesh_tri_offset;
  const float4* vertices;
  const uint4* tri_tex;
  const uint32_t* instance_mesh_ids;
  const float4* instance_transforms;  // 2 x float4 per instance
  const float4* instance_rows;        // 3 x float4 per instance: the rows of its world->object matrix, translation in .w - what the top-level leaf records hold, by instance id
  const uint4* materials;             // 2 x uint4 per material
  // light tree
  const uint4* light_tree_root;   // header, then 3 x 16 B per section; nullptr without lights
  const float* light_root_children;  // the root's children dequantised (8 floats each: mean.xyz, sigma, power, 0, 0, 0), read with scalar loads (dev_light.h)
  const uint4* light_tree_nodes;  // 4 x 16 B per node
  const uint2* light_tri_handles;
  const float4* light_tri_table;     // 4 x 16 B per light: world-space vertex | material id, bidirectional << 16; edge1 | scene triangle; edge2 | area; colour | textured (k_light_table)
  // sampler and LUTs
  const uint32_t* bluenoise_2d;
  const uint16_t* lut_conductor;
  const uint16_t* lut_glossy;
  const uint16_t* lut_dielectric;
  const uint16_t* lut_dielectric_inv;
  // textures (reference: device_texture.c, cuda/texture_utils.cuh): RGBA8 texels of all textures back to back + a table
  const uint4* texture_table;     // first texel, width, height, gamma (float bits)
  const uint32_t* texels;         // r in the low byte
  // acceleration structures (node indices and leaf ranges are absolute, so one base pointer serves both levels)
  const Bvh4Node* bvh_nodes;       // [0, tlas_num_nodes): top level over instances (leaves index tlas_leaves); then every mesh's BVH
  const BvhTri* blas_tris;         // all meshes, traversal order
  const float4* tlas_leaves;       // 4 x float4 per top-level leaf (traversal order): rows of the instance's world->object matrix
                                   // (.w = translation component), then uint bits {instance id, root node of its mesh, 0, 0}
  const Bvh4Node* light_nodes;     // leaves index light_tris
  const BvhTri* light_tris;        // world space, id = light id
  uint32_t num_meshes, num_instances, num_materials, num_lights, num_textures;
  uint32_t tlas_num_nodes, light_num_nodes, tlas_num_leaves;
  // settings / camera / sky (device_structs.h:8-124)
  uint32_t width, height, max_ray_depth, shading_mode;
  float cam_pos[3];
  float cam_rotation[4];
  float cam_fov, cam_aperture_size, cam_object_distance, cam_scale, cam_rr_threshold;
  uint32_t cam_aperture_shape, cam_aperture_blade_count;
  uint32_t sky_mode;
  float sky_constant_color[3];
  // procedural sky (dev_sky.h), read when sky_mode == DEFAULT
  uint32_t sky_steps, sky_ozone_absorption;
  float sky_geometry_offset[3];
  float sky_sun_strength, sky_base_density, sky_rayleigh_density, sky_mie_density, sky_ozone_density, sky_rayleigh_falloff, sky_mie_falloff,
    sky_ground_visibility, sky_ozone_layer_thickness, sky_multiscattering_factor;
  float sky_sun_pos[3];
  float sky_mie_phase[4];
  const float4* sky_lut_transmittance;    // low plane [64][256], high plane
  const float4* sky_lut_multiscattering;  // low plane [32][32], high plane
  float sky_moon_pos[3];
  float sky_moon_tex_offset;
  uint32_t sky_moon_albedo_tex, sky_moon_normal_tex;
  float sky_stars_intensity;
  uint32_t sky_stars_count;
  const float4* sky_stars;            // altitude, azimuth, radius, intensity
  const uint32_t* sky_stars_offsets;  // 64 x 32 + 1
  const float4* sky_hdri;             // [dim][dim] baked panorama (k_sky_hdri), read when sky_mode == HDRI; alpha unused
  uint32_t sky_hdri_dim;
  uint32_t sky_aerial_perspective;    // the air between a ray's origin and its hit is marched too (k_sky_inscattering)
  // fog (dev_volume.h): homogeneous, scattering only; bounded by a disk of radius fog_dist around the camera and by y <= fog_height
  uint32_t fog_active;
  float fog_density, fog_dist, fog_height;
  float fog_phase[4];        // Jendersie-Eon g_hg, g_d, alpha, w_d of the droplet diameter
  const float* bridge_lut;   // 64 x 21 floats
  uint32_t bridge_max_num_vertices;
  // ocean (dev_ocean.h, dev_water.h): procedural height field at ocean_height, Jerlov water below it
  uint32_t ocean_active;
  float ocean_height, ocean_amplitude, ocean_frequency, ocean_refractive_index;
  float ocean_scattering[3], ocean_absorption[3];
  float ocean_molecular_weight;
  uint32_t ocean_caustics_active, ocean_caustics_ris_sample_count;
  float ocean_caustics_domain_scale;
  uint32_t ocean_multiscattering, ocean_triangle_light_contribution;
  // clouds (dev_cloud.h): layer rows = active, height_max, height_min, coverage, coverage_min, type, type_min, wind_speed, cos, sin of wind_angle;
  // RGBA8 noise: shape 128^3, detail 32^3, weather 1024^2
  uint32_t cloud_active, cloud_atmosphere_scattering, cloud_steps, cloud_shadow_steps, cloud_octaves;
  float cloud_offset_x, cloud_offset_z, cloud_density, cloud_noise_shape_scale, cloud_noise_detail_scale, cloud_noise_weather_scale;
  float cloud_phase[4];
  float cloud_layers[3][10];
  const uint32_t* cloud_noise_shape;
  const uint32_t* cloud_noise_detail;
  const uint32_t* cloud_noise_weather;
  // particles (dev_particle.h): quads of the unit cell, tiled 25^3 times in a space scaled by particles_scale; their own two-level tree
  uint32_t particles_active, particles_count;
  float particles_scale, particles_speed;
  float particles_albedo[3], particles_direction[3], particles_phase[4];
  const float4* particle_normals;
  const Bvh4Node* particle_bvh_nodes;
  const BvhTri* particle_tris;
  const float4* particle_leaves;
  uint32_t particle_tlas_num_nodes, particle_num_leaves;
#if LUM_PHASE_QUEUES
  // dev_trace_pool.h: per workgroup of the persistent ray kernels and per pool slot, what does not fit LDS - 4 x 16 bytes of query state, the stack
  // entries beyond the LDS ones. Only the variant build carries the fields (tools/check_variants.sh compiles it).
  uint4* pool_state;
  unsigned long long* pool_stack;
#endif
  // the pass's Sobol / Owen table (dev_sampler.h LUM_SOBOL_TABLE, k_sobol_table): entry (dimension, sample) at [dimension * sobol_stride + (sample - sobol_first)],
  // dimensions 0 .. (max_ray_depth + 1) * kRndTargetCount - 1; nullptr in passes without one (set per pass on the host's copy, wavefront_depths)
  const uint2* sobol_table;
  uint32_t sobol_first, sobol_count, sobol_stride;
};

// Path state, one entry per live path, structure-of-arrays of 16-byte words (coalesced 16 B/lane accesses).
struct PathQueue {
  float4* origin_t;   // origin.xyz, hit distance (written by the trace kernel)
  float4* dir_slot;   // direction.xyz, result slot (uint bits)
  uint4* aux;         // throughput record x,y | medium IOR stack | state flags
  uint4* hit_id;      // hit (or ignore) instance, triangle | pixel x | y << 16 | sample id
  uint32_t* hit_scene_tri;  // index of the hit triangle in the scene arrays (vertices, tri_tex): spares the shade kernel two dependent loads
  uint32_t* parent;         // fused resolve (kernels.h FusedResolve): index of the vertex this path continues, in the previous depth's queue | kParentDeferred
};
constexpr uint32_t kParentDeferred = 0x80000000u, kParentMask = 0x7FFFFFFFu;  // bit 31: that vertex left its ambient sample to this path's closest hit

// Next-event-estimation data of the vertices of one depth, indexed like the path queue they were shaded from.
struct NeeQueue {
  float4* geo_color_light;  // sampled light: radiance * weight rgb | light id (uint bits)
  float4* bsdf_ray_prob;    // BSDF-sampled light direction xyz | its probability (0 = none)
  float4* bsdf_weight_sum;  // shade: bsdf weight rgb | light-tree root sum; after the light query: light colour rgb | valid flag
  uint4* ambient;           // packed colour (record format) xy | packed ray zw
  uint4* sun;               // same for the sun sample (written and read unless the sky is a constant colour)
  uint32_t* amb_path;       // ambient-visibility reuse (kernels.h): the surviving path's index in the next queue - its closest hit answers the ambient sample - or none
  // only with an active ocean (dev_water.h): what lies between the two visibility segments of a sun / ambient sample taken under water
  float4* sun_water;        // 1 - Fresnel reflection at the surface | - | - | flags (uint bits: second segment, total reflection)
  float4* amb_t1;           // transmittance of the vertex's volume up to the surface rgb | 1 - Fresnel reflection
  float4* amb_t2;           // transmittance of the volume beyond the surface rgb | flags
};

// Visibility rays, compacted: every entry is one any-hit ray whose transparency goes to vis[out].
struct ShadowQueue {
  float4* origin_dist;  // origin.xyz | distance
  float4* dir_out;      // direction.xyz | output index (uint bits) = kind * capacity + path index
  uint4* ids;           // target instance, target triangle (the sampled light) | self instance, self triangle
  float4* vis;          // [kinds * capacity]: kind 0 sampled light, 1 BSDF-sampled light, 2 ambient, 3 sun; with an ocean 4 ambient's and 5 sun's second segment
  uint32_t* light_items;  // path indices that need a light-BVH query
  uint32_t capacity;
};

// Fused resolve (kernels.h, above k_shade): what k_shade of depth d needs to resolve the vertices of depth d - 1.
struct FusedResolve {
  PathQueue prev;        // the queue of depth d - 1 (intact: the queues rotate through three buffers)
  NeeQueue nee_prev;     // its NEE records
  ShadowQueue fallback;  // item arrays of the undecided samples' rays; vis = the visibility words of depth d - 1, light_items = the list of their vertices
  uint32_t* ended;       // out: the vertices of depth d that no entry of depth d + 1 continues (counted in kCtlSkyItems: the procedural sky's list does not exist in this mode)
  const uint32_t* ended_prev;  // in: that list of depth d - 1 (the other of two buffers): k_shade of depth d resolves them after its own entries (fused_flags & 4)
};

// What the fog scatters into the rays of one depth (k_volume_inscatter -> visibility rays -> k_volume_resolve), indexed like the path queue.
// The visibility rays of this pass use the ShadowQueue with 17 kinds per path: 0..14 the segments of the bridge to the sampled light, 15 the
// sun, 16 the ambient sample.
constexpr uint32_t kVolumeShadowKinds = 19, kVolumeKindSun = 15, kVolumeKindAmbient = 16, kVolumeKindSun2 = 17, kVolumeKindAmbient2 = 18;
constexpr uint32_t kShadowKindAmbient2 = 4, kShadowKindSun2 = 5, kSurfaceShadowKindsWater = 6;
// Clouds: the (path, layer) marches of a depth and their results (k_clouds_list -> k_clouds_march -> k_clouds)
struct CloudQueue {
  uint32_t* items;       // path index | layer << 30, for every layer a path's ray reaches before its hit
  float4* result;        // [layer * capacity + path]: scattered light rgb | transmittance
  float* hit_dist;       // [layer * capacity + path]
  uint32_t capacity;
};
struct VolumeQueue {
  float4* bridge;        // light colour rgb (already weighted) | number of segments (uint bits; 0 = no bridge)
  uint4* sky;            // sun colour (record format) xy | ambient colour zw; zero = no sample
  float4* weight;        // weight rgb of the vertex the sun and the ambient sample start from
  float4* sun_water;     // as in NeeQueue
  float4* amb_t1;
  float4* amb_t2;
  uint32_t* items;       // paths that scattered in a volume at this depth (k_volume_events), bounced by k_volume_bounce
};

// One wavefront pass over `batch` consecutive sample ids of `num_pixels` pixels (k_generate).
struct PassParams {
  const uint32_t* pixels;  // pixel index (x + y*width) per local pixel, or nullptr for identity
  uint32_t num_pixels, batch, first_sample;
};

// Adaptive sampling bookkeeping shared by host and kernels (dev_adaptive.h).
constexpr uint32_t kAdaptiveStages = 4;          // ADAPTIVE_SAMPLER_NUM_STAGES, device_utils.h:331
struct AdaptiveView {
  const uint32_t* stage_counts;    // per block
  const uint32_t* block_task_end;  // inclusive prefix sum over blocks of 16 * count(current stage); tasks of block b: [end[b-1], end[b])
  uint32_t blocks_x, blocks_y, num_blocks;
  uint32_t executions[kAdaptiveStages + 1];  // completed executions per stage (stage_sample_offsets)
  uint32_t stage_id;
};
struct AdaptivePass { uint32_t task_begin, task_end, block_begin, block_end, executions; };

enum PathState : uint32_t {  // cuda/utils.cuh:114-121
  kStDeltaPath = 1, kStCameraDirection = 2, kStVolumeScattered = 4, kStAllowEmission = 8, kStAllowAmbient = 16, kStUseIgnoreHandle = 32
};
enum SkyMode : uint32_t { kSkyDefault = 0, kSkyHdri = 1, kSkyConstantColor = 2 };

// Per-depth control words (zeroed once per pass).
// One row per depth. The three words k_shade's waves bump with an atomic per 64 vertices (survivors of the NEXT row, visibility items, light
// queries) sit on 128-byte lines of their own: a single word takes ~88 atomics per microsecond (MI355X_MICROARCH.md, "dequeue"), and with all
// three on one line the shade kernel was bound by that (measured: spreading them took 13 % off k_shade). The work cursors of the persistent
// ray kernels have a line each as well (8 words: LUM_XCD_RANGES experiment; word 0 alone otherwise).
#ifndef LUM_CTL_LINE
#define LUM_CTL_LINE 32u  // words between the hot counters (32 words = 128 bytes)
#endif
enum CtrlWord : uint32_t {
  kCtlPaths = 0, kCtlShadowItems = LUM_CTL_LINE, kCtlLightItems = 2u * LUM_CTL_LINE, kCtlSkyItems = 2u * LUM_CTL_LINE + 1u, kCtlTraceCursor = 3u * LUM_CTL_LINE,
  kCtlShadowCursor = 3u * LUM_CTL_LINE + 8u, kCtlStride = 4u * LUM_CTL_LINE,
  // the fog's own visibility pass runs k_shadow_rays on `ctrl + kCtlVolumeShift`: its item count and cursor are these two words
  kCtlVolumeShift = 16u, kCtlVolumeShadowItems = kCtlShadowItems + kCtlVolumeShift, kCtlVolumeShadowCursor = kCtlShadowCursor + kCtlVolumeShift,
  kCtlVolumeItems = 2u * LUM_CTL_LINE + 2u,
  kCtlParticleCursor = LUM_CTL_LINE + 8u,  // work cursor of the particle pass of the closest-hit kernel (8 words, like the other cursors)
  kCtlCloudItems = 2u * LUM_CTL_LINE + 3u, kCtlCloudCursor = LUM_CTL_LINE + 24u,  // the cloud marches of a depth: their number and the persistent kernel's cursor
  kCtlShadeCursor = 3u * LUM_CTL_LINE + 16u  // k_shade's input cursor (LUM_SHADE_DYNAMIC, kernels.h): its waves take the depth's queue entries in chunks
};
static_assert(LUM_CTL_LINE >= 32u, "the fog's control words sit in the second half of the 32-word lines");

#if LUM_PHASE_QUEUES
#ifndef LUM_TRACE_BLOCK
#define LUM_TRACE_BLOCK 1024
#endif
#ifndef LUM_POOL_SLOTS
#define LUM_POOL_SLOTS 128  // rays per wave (<= 256: list entries are bytes)
#endif
#ifndef LUM_POOL_STACK_ENTRIES
#define LUM_POOL_STACK_ENTRIES 2  // 8-byte stack entries per slot kept in LDS
#endif
// per slot: the LDS part of its stack, three 16-byte state words, one byte in each of the wave's four lists
#define LUM_LDS_STACK_BYTES ((LUM_POOL_STACK_ENTRIES * 8u + 48u + 4u) * LUM_POOL_SLOTS * (LUM_TRACE_BLOCK / 64u))
#endif
#ifndef LUM_LDS_STACK_BYTES
// Of a ray workgroup's LDS: bytes that hold the oldest entries of its lanes' traversal stacks instead of tree nodes (0: stacks in scratch). Rounds 2-5: 64 KB (8 closest-hit /
// 16 visibility entries per lane beside 704 staged nodes; round 3 measured 96 KB at 0 / +1 %). Round 6, with the deeper two-triangle-leaf trees: 96 KB (12 / 24 entries, 448 nodes) -
// same box, 48 / 64 / 80 / 96 / 112 KB: hall k_trace 418 / 409 / 409 / 409 / 411 ms per 3 steps (samples/s -0.1 % at 96 KB: noise), the 10 M-triangle scan 127 / 116.5 / 110.6 /
// 108.1 / 107.3 (+3.2 % samples/s at 96 KB), Example-class level up to 96 KB and -1.2 % at 112 (profiles/r06_ab_experiments.txt): what a big tree's rays push past the LDS entries
// goes to scratch, i.e. through the same memory path as the nodes they are looking for.
#define LUM_LDS_STACK_BYTES 98304u
#endif
#ifndef LUM_RAY_BLOCK_MAX
#define LUM_RAY_BLOCK_MAX 1024u
#endif
constexpr uint32_t kRayBlockMax = LUM_RAY_BLOCK_MAX;  // the ray kernels' largest workgroup: a lane's share of the LDS stack area is sized for it

enum Counter : uint32_t { kCntTrace = 0, kCntShadow, kCntLightBvh, kCntVertices, kCntNodes, kCntTris, kCntNodesShadow, kCntTrisShadow, kCntNodesLight, kCntTrisLight, kCntNodesLds,
                         kCntNodesLdsShadow,
                         kCntAmbientDeferred,  // ambient samples whose visibility ray was not queued: the path's next closest-hit ray answers (k_shade)
                         kCntAmbientFallback,  // ... of which the closest hit could not decide: traced by the second visibility pass (counted in kCntShadow too)
                         kCntSpare0, kCntSpare1, kCntCount };

}  // namespace lum
