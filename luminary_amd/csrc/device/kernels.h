// Wavefront path-tracing kernels for gfx950.
//
// One pass renders `batch` consecutive sample ids of `num_pixels` pixels: path slot = sample_in_batch * num_pixels + pixel.
// Live paths are kept compacted in a PathQueue (SoA of 16-byte words); every depth runs
//     trace (closest hit, persistent waves)
//     -> shade (context, NEE sampling, bounce, emission, roulette; appends survivors to the other queue and the visibility
//        rays / light queries it needs to compacted work lists, one wave-aggregated atomic per list)
//     -> light query (rare: BSDF-sampled directions against the light-only BVH; may append one more visibility ray)
//     -> shadow rays (any-hit, persistent waves, one ray per lane)
//     -> resolve (adds the visible light into the path's result slot in the fixed order sampled light, BSDF light, ambient)
// and the pass ends with an in-order accumulation into the frame moments, which reproduces the reference's
// one-sample-at-a-time sums bit for bit (cuda/accumulation.cuh:63-84).
// Reference schedule: device/device_renderer.c:53-134; per-kernel restatements cite their sources below.
#pragma once

#include "kernel_shadow.h"  // dev_trace.h (+ dev_trace_pool.h), ShadowQuery, k_shadow_rays
#include "dev_sky.h"
#include "dev_volume.h"
#include "dev_particle.h"
#include "dev_water.h"
#include "dev_cloud_march.h"

LUM_NS_BEGIN

constexpr int kBlock = 256;
#ifndef LUM_SHADE_WAVES
#define LUM_SHADE_WAVES 3  // minimum waves per SIMD the shade kernel is compiled for (register budget 512 / waves); round 4: 3 for every sky mode, ocean and flavour (2 before: procedural sky -3.6 % / -10 % of the kernel's time fast / exact, ocean scenes -6 %)
#endif
// (Round 2 compiled only the fast flavour's constant-colour-sky instantiation for 3 waves - the instantiations with sun sampling then spilled 56-63 registers.
// The kernel has shrunk since: now they spill 26-30, 73-112 with an ocean, and every instantiation is faster at 3 waves; profiles/r04_ab_experiments.txt.)
#ifndef LUM_CLOUD_WAVES
#define LUM_CLOUD_WAVES 4  // k_clouds: waves per SIMD it is compiled for (2 / 3 / 4 measured: 1602 / 1369 / 1283 ms, profiles/r02_ab_experiments.txt)
#endif
#ifndef LUM_CLOUD_PERSISTENT
#define LUM_CLOUD_PERSISTENT 1  // 0 (measurement only): one lane per path marches its three layers inside k_clouds, no list and no persistent lanes
#endif
#ifndef LUM_HIT_COMPACT
#define LUM_HIT_COMPACT 1  // 0 (measurement only): k_ocean_shade / k_particle_shade shade what every round finds, partial waves and all
#endif
#ifndef LUM_FEATURE_WAVES
#define LUM_FEATURE_WAVES 3  // the shading kernels of particles, ocean surface and volumes (without a bound k_particle_shade took 266 registers: one wave per SIMD)
#endif
#ifndef LUM_VOLUME_WAVES
#define LUM_VOLUME_WAVES 4  // k_volume_inscatter (round 4: 128 VGPRs with 67 spilled beat 168 with 21: the fog's kernels -5 %, a fogged scene +3 % samples/s; the ocean's kernels indifferent)
#endif
#ifndef LUM_SHADE_WAVES_CONSTANT_SKY
#define LUM_SHADE_WAVES_CONSTANT_SKY 3  // both flavours (round 4: the exact flavour's kernel at 3 waves - 168 registers, 8 spilled - instead of 2: -11.5 % of its time, +6 % samples/s)
#endif

// ---- tasks_create (cuda/kernels.cuh:45-193) + thin-lens camera (cuda/camera_thin_lens.cuh:8-86, cuda/camera.cuh:29-35) ----
LUM_DEV void camera_ray(const DeviceScene& sc, const Sampler& smp, V3& origin, V3& ray) {
  const U2 jq = smp.raw2_at(kRndCameraJitter, 0, 0, 0);  // same jitter for every pixel of a sample (camera_utils.cuh:23-27)
  const float jx = unit_float(jq.x), jy = unit_float(jq.y);
  const float step = 2.0f * (sc.cam_fov / sc.width);
  const float vfov = step * sc.height * 0.5f;
  const V3 sensor = v3(sc.cam_fov - step * (smp.px + jx), -vfov + step * (smp.py + jy), 1.0f);
  const V3 to_focal = normalize(v3(0.0f, 0.0f, 0.0f) - sensor);
  const float focal = fmaxf(sc.cam_object_distance * (1.0f / 0.001f), 0.01f);
  const V3 focal_point = to_focal * (-focal / to_focal.z);
  V3 aperture = v3(0.0f, 0.0f, 0.0f);
  if (sc.cam_aperture_size != 0.0f) {
    const F2 r = smp.next2(kRndLens);
    const float asz = sc.cam_aperture_size * (1.0f / 0.001f);
    if (sc.cam_aperture_shape == 1) {
      const int blade = (int) (smp.next1(kRndLensBlade) * sc.cam_aperture_blade_count);
      const float alpha = sqrtf(r.x), beta = r.y;
      const float u = 1.0f - alpha, v = alpha * beta;
      const float astep = (2.0f * kPi) / sc.cam_aperture_blade_count;
      float s1, c1, s2, c2;
      sincos_det(astep * blade, s1, c1); sincos_det(astep * (blade + 1), s2, c2);
      aperture = v3((s1 * u + s2 * v) * asz, (c1 * u + c2 * v) * asz, 0.0f);
    }
    else {
      const float alpha = r.x * 2.0f * kPi, beta = sqrtf(r.y) * asz;
      float sa, ca; sincos_det(alpha, sa, ca);
      aperture = v3(ca * beta, sa * beta, 0.0f);
    }
  }
  const Quat q{sc.cam_rotation[0], sc.cam_rotation[1], sc.cam_rotation[2], sc.cam_rotation[3]};
  V3 o = qapply(q, aperture);
  o = o * (sc.cam_scale * 0.001f);
  origin = o + v3(sc.cam_pos[0], sc.cam_pos[1], sc.cam_pos[2]);
  ray = qapply(q, normalize(focal_point - aperture));
}

// kernels.cuh:146, :172-186: the medium the camera is in (bsdf_refraction_index_ambient, bsdf_utils.cuh:128-133) and the volumes around it, in the
// upper bits of the sample-id word (dev_volume.h)
LUM_DEV uint32_t initial_medium(const DeviceScene& sc, V3 origin) {
  return medium_ior_modify(0u, (sc.ocean_active && origin.y < sc.ocean_height) ? sc.ocean_refractive_index : 1.0f, true);
}
LUM_DEV uint32_t initial_volumes(const DeviceScene& sc, V3 origin, uint32_t sample_id) {
  uint32_t w = sample_id;
  if (sc.fog_active) w = volume_stack_modify(w, kVolumeFog, true);
  if (sc.ocean_active && ocean_is_underwater(sc, origin)) w = volume_stack_modify(w, kVolumeOcean, true);
  return w;
}

__global__ __launch_bounds__(kBlock) void k_generate(DeviceScene sc, PassParams pp, PathQueue q, float4* results, uint32_t* count) {
  const uint32_t total = pp.num_pixels * pp.batch;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < total; i += gridDim.x * kBlock) {
    const uint32_t b = i / pp.num_pixels, p = i - b * pp.num_pixels;
    const uint32_t index = pp.pixels ? pp.pixels[p] : p;
    const uint32_t y = index / sc.width, x = index - y * sc.width;
    const Sampler smp{sc.bluenoise_2d, x, y, pp.first_sample + b, 0};
    V3 o, d;
    camera_ray(sc, smp, o, d);
    const U2 rec = record_pack(splat(1.0f));
    q.origin_t[i] = make_float4(o.x, o.y, o.z, kFltMax);
    q.dir_slot[i] = make_float4(d.x, d.y, d.z, bitsf(i));
    q.aux[i]      = make_uint4(rec.x, rec.y, initial_medium(sc, o), kStDeltaPath | kStCameraDirection | kStAllowEmission | kStAllowAmbient);
    q.hit_id[i]   = make_uint4(0u, 0u, x | (y << 16), initial_volumes(sc, o, pp.first_sample + b));
    results[i]    = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = total;
}

// ---- closest-hit pass (replaces optix/optix_kernel_raytrace.cu:147-183) ----
// `order` (optional): the queue is traced in the order of a sort by ray origin cell and direction octant (k_ray_sort_keys, core.hip), so
// that the lanes of a wave and the waves of a CU walk the same part of the tree; the queue itself is not moved.
// Ambient-visibility reuse (fast flavour, plain scenes - lumc_set_ambient_reuse, core.hip). In a constant-colour or panorama sky the ambient sample of a
// vertex runs along its bounce direction (direct_lighting.cuh:388-405): the closest-hit ray of the NEXT depth walks the same line. Where that ray's
// nearest hit is opaque, or it hits nothing, the ambient visibility is known without a second traversal (94 % of the hall's ambient rays end at an opaque
// surface). k_shade then queues no visibility ray for a surviving path but notes the path's new queue index next to the vertex (nee.amb_path); the
// closest-hit pass leaves two flag bits next to the hit's triangle index (below); k_resolve of the vertex's depth runs AFTER that pass and reads the
// answer from the path's hit. What the nearest hit cannot decide (transparent or textured first hit, a hit closer than eps, a skipped cut-out) is listed,
// traced by a small second visibility pass - the very ray k_shade would have queued - and resolved by k_resolve_listed. The reference traces the ambient
// ray along the direction after its 2 x 32-bit octahedral packing (ray_unpack(ray_pack(bounce))) and from the hit point, a last bit away from the ray the
// path continues along. The fast flavour does not promise the last bit and takes the closest hit's word for it; the exact flavour (kReuseProves) takes
// "blocked" only after re-testing the ambient ray itself against the triangle the closest hit found, and traces every sample whose ray found nothing:
// bit-identical to the oracle and still nine ambient rays in ten answered on a closed scene - but the re-test's scattered fetches cost more than those
// cheap rays (measured on the hall, exact flavour: visibility kernel -29 ms, resolve +61 ms per step), so the exact flavour only does it when asked to.
constexpr uint32_t kNoAmbientPath = 0xFFFFFFFFu;
constexpr bool kReuseProves = !LUM_FAST;  // the exact flavour takes only the answers it can prove for the ambient ray itself (k_resolve_reuse)
// hit_scene_tri: flag bits above the triangle index (28 bits, like the leaf ranges): the hit is opaque on its own | a cut-out was skipped | the hit lies beyond
// eps | something was hit - everything k_resolve_reuse needs to know about the path's closest hit, in the one word it reads
constexpr uint32_t kHitTriOpaque = 0x80000000u, kHitTriCutout = 0x40000000u, kHitTriBeyondEps = 0x20000000u, kHitTriHit = 0x10000000u, kHitTriMask = 0x0FFFFFFFu;

struct TraceQuery : ClosestState {
  PathQueue q;
  const uint32_t* order;
  uint32_t item;
  LUM_DEV bool load(const DeviceScene&, uint32_t j, V3& o, V3& d, float& tmax) {
    const uint32_t i = order ? order[j] : j;
    item = i;
    const float4 o4 = ld_stream(&q.origin_t[i]), d4 = ld_stream(&q.dir_slot[i]);
    const uint32_t state = q.aux[i].w;
    const uint2 ign = *reinterpret_cast<const uint2*>(&q.hit_id[i]);
    begin((state & kStUseIgnoreHandle) != 0, ign.x, ign.y);
    o = v3(o4.x, o4.y, o4.z); d = v3(d4.x, d4.y, d4.z); tmax = kFltMax;
    return true;
  }
  // LUM_PHASE_QUEUES (dev_trace_pool.h): the state a pool slot keeps outside LDS, and the world-space ray again
  static constexpr uint32_t kMutableVecs = 1;
  LUM_DEV void save_mutable(uint4* m) const { m[0] = make_uint4(best.instance_id, best.tri_id, best.scene_tri, cutout ? 1u : 0u); }
  LUM_DEV void load_mutable(const uint4* m, float tmax) { best = Hit{m[0].x, m[0].y, tmax, m[0].z}; cutout = m[0].w != 0u; }
  LUM_DEV void save_const(uint4& c) const { c = make_uint4(use_ignore ? 1u : 0u, ign_inst, ign_tri, 0u); }
  LUM_DEV void load_const(uint4 c) { use_ignore = c.x != 0u; ign_inst = c.y; ign_tri = c.z; }
  LUM_DEV void world_ray(const DeviceScene&, uint32_t i, V3& o, V3& d) const {
    const float4 o4 = q.origin_t[i], d4 = q.dir_slot[i];
    o = v3(o4.x, o4.y, o4.z); d = v3(d4.x, d4.y, d4.z);
  }
  LUM_DEV void finish(const DeviceScene&, uint32_t) {
    const uint32_t i = item;
    const Hit h = result();
    reinterpret_cast<float*>(&q.origin_t[i])[3] = h.t;
    *reinterpret_cast<uint2*>(&q.hit_id[i]) = make_uint2(h.instance_id, h.tri_id);
    static_assert(kOpaqueBit == kHitTriOpaque, "the opaque bit is stored where ClosestState keeps it");
    q.hit_scene_tri[i] = ((best.t == kFltMax) ? 0u : (best.scene_tri | kHitTriHit | (best.t > kEps ? kHitTriBeyondEps : 0u))) | (cutout ? kHitTriCutout : 0u);
  }
};

__global__ LUM_TRACE_BOUNDS void k_trace(DeviceScene sc, PathQueue q, const uint32_t* order, uint32_t* ctrl, uint64_t* counters, uint32_t lds_nodes) {
  RayStats st{0, 0, 0};
  uint32_t rays = 0;
  TraceQuery tq;
  tq.q = q;
  tq.order = order;
  tq.item = 0;
#if LUM_PHASE_QUEUES
  trace_items_pool(sc, ctrl[kCtlPaths], ctrl + kCtlTraceCursor, tq, st, rays, lds_nodes);
#else
  LUM_TRACE_ITEMS(sc, ctrl[kCtlPaths], ctrl + kCtlTraceCursor, tq, st, rays, lds_nodes);
#endif
  flush_stats(counters, st, rays, kCntTrace, kCntNodes, kCntTris, kCntNodesLds);
}

// ---- surface context (cuda/geometry_utils.cuh:13-221, untextured) ----
LUM_DEV GeoContext build_context(const DeviceScene& sc, V3 hit_origin, V3 ray_world, uint32_t state, uint32_t inst, uint32_t tri, uint32_t tbase, uint32_t medium) {
  const Transform tf = load_transform(sc, inst);
  const float4 a = sc.vertices[3 * tbase], b = sc.vertices[3 * tbase + 1], c = sc.vertices[3 * tbase + 2];
  const uint4 tt = sc.tri_tex[tbase];
  V3 position = xf_point_inv(tf, hit_origin);
  const V3 ray = xf_rot_inv(tf, ray_world);
  const V3 p0 = v3(a.x, a.y, a.z);
  const V3 e1 = v3(b.x, b.y, b.z) - p0, e2 = v3(c.x, c.y, c.z) - p0;
  V3 face = normalize(cross(e1, e2));
  const F2 co = barycentric_in_triangle(p0, e1, e2, position);
  position = p0 + (e1 * co.x + e2 * co.y);
  position = xf_point(tf, position);
  const Material mat = load_material(sc, tt.w & 0xFFFFu);
  const F2 tex_coords = triangle_uv(tt, co);
  const V3 n0 = normal_unpack(fbits(a.w)), n1 = normal_unpack(fbits(b.w)), n2 = normal_unpack(fbits(c.w));
  const bool inside = dot(face, ray) > 0.0f;
  if (inside) face = face * -1.0f;
  V3 normal = lerp_normals(n0, n1 - n0, n2 - n0, co, face);
  if (mat.normal_tex != kTextureNone) {  // geometry_utils.cuh:25-50
    const bool valid = mat.normal_tex < sc.num_textures;
    const float4 nf = texture_load(sc, mat.normal_tex, tex_coords, false, make_float4(0.0f, 0.0f, 1.0f, 0.0f));
    V3 mn = v3(nf.x, nf.y, nf.z);
    if ((mat.flags & kDMatNormalMapCompressed) && valid) mn = mn * 2.0f - v3(1.0f, 1.0f, 1.0f);
    mn = normalize(mn);
    normal = qapply(qinv(rotation_to_z(normal)), mn);
  }
  normal = adapt_normal(ray * -1.0f, normal, face);
  Col albedo = mat.albedo;
  float alpha = mat.alpha;
  if (mat.albedo_tex != kTextureNone) {  // geometry_utils.cuh:109-121
    const float4 af = texture_load(sc, mat.albedo_tex, tex_coords, true, make_float4(0.9f, 0.9f, 0.9f, 1.0f));
    albedo = col(af.x, af.y, af.z);
    alpha = af.w;
  }
  const bool emissive_side = !inside || (mat.flags & kDMatBidirectionalEmission);
  const bool emits = (mat.flags & kDMatEmission) && emissive_side && ((state & kStAllowEmission) != 0);
  Col emission = emits ? mat.emission : col(0.0f, 0.0f, 0.0f);
  if (emits && mat.luminance_tex != kTextureNone) {  // geometry_utils.cuh:130-137
    const float4 lf = texture_load(sc, mat.luminance_tex, tex_coords, true, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
    emission = col(lf.x, lf.y, lf.z) * (alpha * mat.emission_scale);
  }
  float roughness = mat.roughness;
  if (mat.roughness_tex != kTextureNone) roughness = texture_load(sc, mat.roughness_tex, tex_coords, true, make_float4(0.5f, 0.0f, 0.0f, 0.0f)).x;  // :140-150
  if (mat.flags & kDMatRoughnessAsSmoothness) roughness = 1.0f - roughness;
  roughness = fmaxf(roughness, 2e-2f);                                          // BSDF_ROUGHNESS_CLAMP, cuda/utils.cuh:46
  if ((state & kStDeltaPath) == 0) roughness = fmaxf(roughness, mat.roughness_clamp);
  uint32_t flags = mat.flags & kDMatSubstrateMask;
  if (mat.metallic_tex == kTextureNone && (mat.flags & kDMatMetallic)) flags |= kMatMetallic;
  if (mat.flags & kDMatColoredTransparency) flags |= kMatColoredTransparency;
  if (inside) flags |= kMatRefractionInside;
  const float other_ior = medium_ior_peek(medium, inside);
  const float ior_in = inside ? mat.refraction_index : other_ior;
  const float ior_out = inside ? other_ior : mat.refraction_index;
  if (((flags & kMatSubstrateMask) == kMatTranslucent) && (fabsf(1.0f - ior_in / ior_out) < 1e-4f)) {
    if ((flags & kMatColoredTransparency) == 0) albedo = col(lerpf(1.0f, albedo.r, alpha), lerpf(1.0f, albedo.g, alpha), lerpf(1.0f, albedo.b, alpha));
    alpha = 0.0f;
    flags |= kMatColoredTransparency;
  }
  GeoContext g;
  g.instance_id = inst; g.tri_id = tri;
  g.normal = xf_rot(tf, normal);
  g.face_normal_packed = normal_pack(face);
  g.position = position;
  g.V = ray_world * -1.0f;
  g.state = state;
  g.params.flags = flags;
  g.params.set(albedo, alpha, roughness, emission, ior_in / ior_out);
  return g;
}

LUM_DEV void add_to_result(float4* results, uint32_t slot, Col v) {  // write_beauty_buffer, cuda/memory.cuh:359-368
  if (!any_positive(v)) return;
  float4 r = results[slot];
  r.x += v.r; r.y += v.g; r.z += v.b;
  results[slot] = r;
}

// ---- the resolve of a vertex (optix/optix_kernel_shadow.cu:15-100: sampled light, BSDF-sampled light, (sun,) ambient, then the path's weight), used by the
// resolve kernels further down and - fused - by k_shade for the vertex a path comes from ----
// The loads come in two batches, each issued as a block before anything waits for it: the vertex's own records (state, slot, the four NEE records),
// then what those call for (the visibility words that matter, the path's result slot). Written branch by branch, as the reference's kernel reads, a lane
// made five dependent round trips; the kernel is a stream of 160 bytes per vertex with no arithmetic to speak of, so its time was those round trips.
// The sums are formed in the reference's order (sampled light, BSDF-sampled light, sun, ambient; then the path's weight).
struct ResolveRecords { uint4 aux; uint32_t slot; float4 cl, lc; uint4 amb; };
LUM_DEV ResolveRecords load_resolve_records(const PathQueue& in, const NeeQueue& nee, uint32_t i) {
  ResolveRecords r;
  r.aux = ld_stream(&in.aux[i]);
  r.slot = fbits(reinterpret_cast<const float*>(&in.dir_slot[i])[3]);
  r.cl = ld_stream(&nee.geo_color_light[i]);
  r.lc = ld_stream(&nee.bsdf_weight_sum[i]);
  r.amb = ld_stream(&nee.ambient[i]);
  return r;
}
// kAmbient: 0 = the ambient sample's visibility is the visibility pass's word; 1 = it is `ambient_vis` (decided by the caller from the path's next closest
// hit); 2 = the fast flavour's reuse: it comes from `*hit_word` (next.hit_scene_tri of the deferred sample's path), loaded HERE with the second batch and
// decoded afterwards - false is returned, and nothing written, when that hit does not decide the sample (the caller queues its ray).
template <int kAmbient>
LUM_DEV bool resolve_records(const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, float4* results, uint32_t i, bool lights_present, float ambient_vis,
                             const ResolveRecords& r, const uint32_t* hit_word = nullptr) {
  const uint4 aux = r.aux;
  const uint32_t slot = r.slot;
  const bool geo_allowed = lights_present && ((aux.w & kStVolumeScattered) == 0);
  const float4 cl = r.cl, lc = r.lc;
  const uint4 amb = r.amb;
  const bool deferred = kAmbient == 2 && hit_word != nullptr;
  const bool need_geo = fbits(cl.w) != kLightIdInvalid && geo_allowed, need_bsdf = lc.w != 0.0f, need_amb = kAmbient != 1 && !deferred && (amb.x != 0 || amb.y != 0);
  const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float4 vg = zero, vb = zero, va = zero;
  uint32_t word = 0u;
  if (deferred) word = *hit_word;
  if (need_geo) vg = ld_stream(&sq.vis[i]);
  if (need_bsdf) vb = ld_stream(&sq.vis[sq.capacity + i]);
  if (need_amb) va = ld_stream(&sq.vis[2u * sq.capacity + i]);
  const float4 before = results[slot];
  if (deferred) {  // nothing hit (and no cut-out skipped) = visible; an opaque nearest hit beyond eps = blocked; anything else is not decided here
    if (!(word & kHitTriHit)) { if (word & kHitTriCutout) return false; ambient_vis = 1.0f; }
    else { if ((word & (kHitTriBeyondEps | kHitTriOpaque)) != (kHitTriBeyondEps | kHitTriOpaque)) return false; ambient_vis = 0.0f; }
  }
  Col acc = splat(0.0f);
  acc = acc + col(cl.x, cl.y, cl.z) * col(vg.x, vg.y, vg.z);  // sampled light (direct_lighting.cuh:445-464)
  {  // BSDF-sampled direction (direct_lighting.cuh:586-667)
    Col seen = col(lc.x, lc.y, lc.z) * col(vb.x, vb.y, vb.z);
    if (need_bsdf && (sc.fog_active || sc.ocean_active)) { const float4 t = nee.bsdf_ray_prob[i]; seen = seen * col(t.x, t.y, t.z); }
    acc = acc + seen;
  }
  if (sc.sky_mode != kSkyConstantColor) {  // sun (direct_lighting.cuh:466-519)
    const uint4 sun = nee.sun[i];
    Col vis = splat(0.0f);
    if (sun.x != 0 || sun.y != 0) { const float4 v = sq.vis[3u * sq.capacity + i]; vis = col(v.x, v.y, v.z); }
    if (sc.ocean_active && (sun.x != 0 || sun.y != 0)) {
      const float4 w = nee.sun_water[i];
      Col vis2 = splat(1.0f);
      if ((fbits(w.w) & kSkyRaySecond) && !(fbits(w.w) & kSkyRayTotalReflection)) { const float4 v = sq.vis[kShadowKindSun2 * sq.capacity + i]; vis2 = col(v.x, v.y, v.z); }
      acc = acc + combine_sun_ray(record_unpack(U2{sun.x, sun.y}), vis, w.x, fbits(w.w), vis2);
    }
    else acc = acc + record_unpack(U2{sun.x, sun.y}) * vis;
  }
  {  // ambient (direct_lighting.cuh:521-584); zero in DEFAULT mode
    const Col vis = (kAmbient == 1 || deferred) ? splat(ambient_vis) : col(va.x, va.y, va.z);
    Col seen = record_unpack(U2{amb.x, amb.y}) * vis;
    if (sc.ocean_active) {
      if (amb.x != 0 || amb.y != 0) {
        const float4 t1 = nee.amb_t1[i], t2 = nee.amb_t2[i];
        Col vis2 = splat(1.0f);
        if ((fbits(t2.w) & kSkyRaySecond) && !(fbits(t2.w) & kSkyRayTotalReflection)) { const float4 v = sq.vis[kShadowKindAmbient2 * sq.capacity + i]; vis2 = col(v.x, v.y, v.z); }
        seen = combine_ambient_ray(record_unpack(U2{amb.x, amb.y}), vis, col(t1.x, t1.y, t1.z), t1.w, col(t2.x, t2.y, t2.z), fbits(t2.w), vis2);
      }
    }
    else if (sc.fog_active) {  // direct_lighting.cuh:561-563
      const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
      seen = seen * volume_transmittance(sc, kVolumeFog, v3(o4.x, o4.y, o4.z) + v3(d4.x, d4.y, d4.z) * o4.w, ray_unpack(U2{amb.z, amb.w}), kFltMax);
    }
    acc = acc + seen;
  }
  const Col v = acc * record_unpack(U2{aux.x, aux.y});  // write_beauty_buffer, cuda/memory.cuh:359-368 (add_to_result with the slot already read)
  if (any_positive(v)) results[slot] = make_float4(before.x + v.r, before.y + v.g, before.z + v.b, before.w);
  return true;
}
template <bool kAmbientKnown>
LUM_DEV void resolve_vertex(const DeviceScene& sc, const PathQueue& in, const NeeQueue& nee, const ShadowQueue& sq, float4* results, uint32_t i, bool lights_present, float ambient_vis) {
  resolve_records<kAmbientKnown ? 1 : 0>(sc, in, nee, sq, results, i, lights_present, ambient_vis, load_resolve_records(in, nee, i));
}

// ---- shading pass: cuda/geometry.cuh:11-180 (+ miss handling of cuda/sky.cuh:609-633, roulette cuda/directives.cuh:11-32) ----
// One instantiation per sky mode (kSkyMode == sc.sky_mode, chosen at launch): sun sampling is a third of the kernel's code again, which a
// constant-colour scene never runs but would pay for in registers and instruction cache. The modes differ in three places
// (direct_lighting.cuh:257-283, sky.cuh:534-606): what a path that left the scene adds (DEFAULT: ray-marched by k_sky; HDRI: panorama
// texel + sun disk; CONSTANT: the colour), whether the bounce direction is an ambient sample (not DEFAULT), and whether the sun is
// sampled (not CONSTANT).
// kWater (the scene has an ocean): a path's volume is read from its stack, a vertex under water takes the sun through the surface and its sun and
// ambient samples get a second visibility segment beyond it (dev_water.h); without an ocean none of that code exists in the kernel.
// kStage (experiment, LUM_SHADE_STAGED): 0 = the whole vertex in one kernel (the product); 1 = the light sampling alone - context, root pass, the eight
// candidates, the chosen light's visibility item, the light record and the root sum (parked in the fourth word of the BSDF-direction record) - and
// 2 = everything else, with the root sum read back: the split the reference's geometry kernel suggests (cuda/geometry.cuh:11-180 calls the light
// sampling as one block) so that each stage gets its own register budget.
#ifndef LUM_SHADE_STAGED
#define LUM_SHADE_STAGED 0
#endif
#ifndef LUM_SHADE_STAGE1_WAVES
#define LUM_SHADE_STAGE1_WAVES 3
#endif
#ifndef LUM_SHADE_STAGE2_WAVES
#define LUM_SHADE_STAGE2_WAVES 4
#endif
// Fused resolve (fast flavour with the ambient reuse, lumc_set_fused_resolve): k_resolve_reuse is a stream of 160 bytes per vertex with five loads and a
// dozen multiply-adds - a kernel that waits - while k_shade computes with idle memory pipes. So the resolve of depth d - 1 rides in k_shade of depth d: every
// path entry knows the vertex it continues (PathQueue::parent, written with the survivor) and, before anything else touches its result slot, forms that
// vertex's sum from the previous depth's records (`nee_prev`, `prev`: the queues of three and the NEE records of two depths are kept) and its own closest
// hit's word - the same loads, now under the other waves' candidate loops. The per-path order of the sums is the usual one (vertex d - 1, then emission or sky of
// depth d). Vertices whose path ended have no entry: k_shade lists them, and the next depth's k_shade takes the list as further input (fused_flags & 4; round 4
// ran k_resolve_ended after the depth's visibility pass, kept behind lumc_set_fused_resolve(2)). What stays outside: samples the hit
// cannot decide (`fallback`: their ambient ray is queued into its own item arrays, traced by a small second pass, resolved by k_resolve_listed - for those
// the vertex's sum lands after the next depth's emission: the fast flavour's rounding, not the exact flavour's, which never runs this).
// (struct FusedResolve: dev_scene.h)
#ifndef LUM_SHADE_DYNAMIC
#define LUM_SHADE_DYNAMIC (!LUM_SHADE_STAGED)  // k_shade's waves take their input through a cursor (the staged experiment's two kernels would share it: fixed shares there)
#endif
constexpr uint32_t kShadeChunkRounds = 16u;  // x 64 queue entries per bump of the cursor at most: ~45 000 atomics on the word per launch of 42 M entries
template <uint32_t kSkyMode, bool kWater, int kStage = 0, bool kTable = false>
__global__ __launch_bounds__(kBlock, kStage == 1 ? LUM_SHADE_STAGE1_WAVES : kStage == 2 ? LUM_SHADE_STAGE2_WAVES : (kSkyMode == kSkyConstantColor && !kWater) ? LUM_SHADE_WAVES_CONSTANT_SKY : LUM_SHADE_WAVES) void k_shade(DeviceScene sc, PathQueue in, PathQueue out, NeeQueue nee, ShadowQueue sq, float4* results,
                                                                    uint32_t* ctrl, uint32_t depth_const, uint64_t* counters, uint32_t ambient_reuse, const FusedResolve* __restrict__ fused_dev,
                                                                    uint32_t fused_flags) {
  const uint32_t n = ctrl[kCtlPaths];
  uint32_t* count_out = ctrl + kCtlStride + kCtlPaths;
  // fused resolve, fused_flags & 4: the previous depth's vertices that no entry continues (its k_shade listed them) are input too - entries n .. n_input - 1 have
  // a parent and nothing else - instead of a kernel of their own after that depth's visibility pass (k_resolve_ended), which only waited for memory
  const uint32_t n_input = n + ((LUM_SHADE_DYNAMIC && LUM_FAST && kStage == 0 && !kWater && (fused_flags & 5u) == 5u) ? (ctrl - kCtlStride)[kCtlSkyItems] : 0u);
  // (the fast flavour only: the exact flavour's sums land in the reference's order for every vertex, its kernels do not carry the code)
  // (its pointers come through memory, read where they are used: as kernel arguments they sat in scalar registers for the whole kernel, and the spills that
  // caused put 27 more lane reads into every iteration of the candidate loop)
  const bool resolve_parents = LUM_FAST && kStage == 0 && !kWater && (fused_flags & 1u) != 0u, announce_children = LUM_FAST && kStage == 0 && !kWater && (fused_flags & 2u) != 0u;
  // ambient_reuse (see AmbientReuse above): the ambient visibility of a surviving path comes from its next closest-hit ray; never with an ocean (the ambient
  // ray then ends at the water surface) or under the procedural sky (no ambient sample)
  const bool reuse_ambient = ambient_reuse != 0u && !kWater && kSkyMode != kSkyDefault;
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  const Col sky = (sc.sky_mode == kSkyConstantColor) ? col(sc.sky_constant_color[0], sc.sky_constant_color[1], sc.sky_constant_color[2]) : splat(0.0f);
  uint32_t vertices = 0;
#if LUM_SHADE_DYNAMIC
  // Input by cursor: a wave takes the depth's queue entries in chunks through one atomic per chunk, fetched a chunk ahead, until the queue is used up. With a
  // fixed share per wave (the grid-stride loop below) every wave of the grid ends on a partial batch of surface hits - one batch time per wave, 0.4 ms of a
  // 12.7 ms launch with 8 workgroups per resident place, measured as the intercept of time against pass size - and the more waves, the better the balance: with
  // the cursor the balance is the cursor's and the grid is the resident set (core.hip shade_grid). Chunks are guided: half of an equal share of what is left
  // (as far as the wave has seen the cursor), between 1 and kShadeChunkRounds rounds of 64 entries - long chunks while the queue is long (few atomics on one
  // word), single rounds at its end and in the short queues of deep depths, where a fixed 16 rounds left most of the chip without work (measured: +8 % on the
  // scan's and the Example-class scene's k_shade).
  // The cursor runs over a schedule of rounds of 64 entries: the queue's rounds with the rounds of the listed vertices (n .. n_input - 1: only a resolve, i.e.
  // only memory latency) spread evenly between them, one after every `every` of the queue's - at the queue's end they cost more than the kernel they replace
  // (all waves reach them together, with nothing to compute beside them: hall k_shade +9.5 ms per 3 steps for a kernel of 11.8).
  uint32_t* const cursor = ctrl + kCtlShadeCursor;
  const uint32_t n_listed = n_input - n, rounds_q = (n + 63u) / 64u, rounds_l = (n_listed + 63u) / 64u;
  const uint32_t every = rounds_l ? max(rounds_q / rounds_l, 1u) : 0u, groups = rounds_l ? min(rounds_l, rounds_q / every) : 0u;
  const uint32_t n_sched = (rounds_q + rounds_l) * 64u;
  auto entry_at = [&](uint32_t pos) -> uint32_t {  // the lane's entry at position `pos` of the schedule (wave-uniform, a multiple of 64): < n a queue entry, < n_input a listed vertex, else none
    const uint32_t t = pos / 64u, mixed = groups * (every + 1u);
    uint32_t base;
    bool listed;
    if (t < mixed) { const uint32_t g = t / (every + 1u), o = t - g * (every + 1u); listed = o == every; base = listed ? g : g * every + o; }
    else { const uint32_t r = t - mixed, left = rounds_q - groups * every; listed = r >= left; base = listed ? groups + (r - left) : groups * every + r; }
    const uint32_t e = base * 64u + lane;
    return listed ? (e < n_listed ? n + e : 0xFFFFFFFFu) : (e < n ? e : 0xFFFFFFFFu);
  };
  const uint32_t share_div = gridDim.x * (kBlock / 64u) * 2u * 64u;  // entries per round and wave, twice
  auto guided = [&](uint32_t seen) -> uint32_t { return 64u * min(max((seen < n_sched ? n_sched - seen : 0u) / share_div, 1u), kShadeChunkRounds); };
  uint32_t chunk_len = guided(0u), next_len = chunk_len;
  uint32_t grabbed = 0u;  // lane 0: the chunk after next (the atomic's result is only looked at when the current chunk ends)
  if (lane == 0u) grabbed = atomicAdd(cursor, chunk_len);
  uint32_t chunk = (uint32_t) __builtin_amdgcn_readfirstlane((int) grabbed), chunk_round = 0u;
  // a workgroup that only starts when the queue is used up (the grid is a little larger than the resident set, so that every place is taken from the start
  // wherever the dispatcher puts the workgroups) leaves before it stages anything
  if (!__syncthreads_or((int) (chunk < n_sched))) return;
  next_len = guided(chunk + chunk_len);
  if (lane == 0u) grabbed = atomicAdd(cursor, next_len);
#else
  const uint32_t rounds = (n_input + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
#endif
  // Paths that left the scene only add the sky term; the surface vertices are two orders of magnitude more work. A wave therefore
  // collects the indices of its surface hits in LDS and shades them 64 at a time, so that misses do not leave lanes idle during the
  // expensive part (the reference sorts tasks by hit type for the same reason, cuda/kernels.cuh:391-484).
  __shared__ uint32_t pending_hits[kBlock / 64][128];
  uint32_t* pending = pending_hits[threadIdx.x >> 6];
  uint32_t num_pending = 0;  // wave-uniform
  // the candidate loop's table lines and handles of the first lights, in LDS (dev_light.h StagedLights)
  StagedLights staged_lights{nullptr, nullptr, 0u};
#if LUM_LDS_LIGHTS
  if (kStage != 2) {
    __shared__ LdsF4 lds_light_table[4u * LUM_LDS_LIGHTS];
    __shared__ unsigned long long lds_light_handles[LUM_LDS_LIGHTS];
    const uint32_t staged = lights_present ? min(sc.num_lights, (uint32_t) LUM_LDS_LIGHTS) : 0u;
    for (uint32_t k = threadIdx.x; k < 4u * staged; k += kBlock) { const float4 v = sc.light_tri_table[k]; const LdsF4 t = {v.x, v.y, v.z, v.w}; lds_light_table[k] = t; }
    for (uint32_t k = threadIdx.x; k < staged; k += kBlock) { const uint2 h = sc.light_tri_handles[k]; lds_light_handles[k] = (unsigned long long) h.x | ((unsigned long long) h.y << 32); }
    __syncthreads();
    staged_lights.table = (const __attribute__((address_space(3))) LdsF4*) lds_light_table;
    staged_lights.handles = (const __attribute__((address_space(3))) unsigned long long*) lds_light_handles;
    staged_lights.count = staged;
  }
#endif
  ShadeClock clock;
  clock.start();
  // fused resolve: an entry's parent word and slot are fetched one round ahead (two registers across the batch in between: one dependent round trip less per round)
  uint32_t parent_ahead = 0u, slot_ahead = 0u;
  if (resolve_parents) {
#if LUM_SHADE_DYNAMIC
    const uint32_t i0 = entry_at(chunk);
#else
    const uint32_t i0 = blockIdx.x * kBlock + threadIdx.x;
#endif
    if (i0 < n) { parent_ahead = in.parent[i0]; slot_ahead = fbits(reinterpret_cast<const float*>(&in.dir_slot[i0])[3]); }
    else if (i0 < n_input) parent_ahead = fused_dev->ended_prev[i0 - n];
  }
  for (uint32_t round = 0;; round++) {
#if LUM_SHADE_DYNAMIC
    const bool input_done = chunk >= n_sched;
#else
    const bool input_done = round >= rounds;
#endif
    if (!input_done) {
#if LUM_SHADE_DYNAMIC
      const uint32_t i = entry_at(chunk + chunk_round * 64u);
      if (++chunk_round * 64u == chunk_len) {  // on to the chunk fetched ahead, and one more on its way
        chunk_round = 0u;
        chunk = (uint32_t) __builtin_amdgcn_readfirstlane((int) grabbed);
        chunk_len = next_len;
        next_len = guided(chunk + chunk_len);
        if (lane == 0u) grabbed = atomicAdd(cursor, next_len);
      }
      const uint32_t i_ahead = entry_at(chunk + chunk_round * 64u);  // (none once the schedule is used up)
      const bool have_ahead = true;
#else
      const uint32_t i = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
      const uint32_t i_ahead = ((round + 1u) * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
      const bool have_ahead = round + 1u < rounds;
#endif
      bool is_hit = false, is_sky = false;
      if (resolve_parents) {  // the vertex this entry continues: its sum, before the entry's own emission or sky term
        const FusedResolve& fused = *fused_dev;
        bool undecided = false;
        uint32_t ip = 0;
        uint4 amb = make_uint4(0u, 0u, 0u, 0u);
        const uint32_t p = parent_ahead;
        uint32_t slot = slot_ahead;
        {
          if (have_ahead && i_ahead < n) { parent_ahead = in.parent[i_ahead]; slot_ahead = fbits(reinterpret_cast<const float*>(&in.dir_slot[i_ahead])[3]); }
          else if (have_ahead && i_ahead < n_input) parent_ahead = fused.ended_prev[i_ahead - n];  // (its slot: with the vertex's records below)
        }
        if (i < n_input) {
          ip = p & kParentMask;
          // resolve_records<2> for a plain scene (no fog, no ocean: the reuse's condition), written out: the records, then - together - the hit word of a
          // deferred sample, the visibility words that matter and the result slot; the sums in the reference's order
          if (i >= n) slot = fbits(reinterpret_cast<const float*>(&fused.prev.dir_slot[ip])[3]);  // a listed vertex: the result slot from its own queue entry
          const uint4 paux = ld_stream(&fused.prev.aux[ip]);
          const float4 cl = ld_stream(&fused.nee_prev.geo_color_light[ip]), lc = ld_stream(&fused.nee_prev.bsdf_weight_sum[ip]);
          amb = ld_stream(&fused.nee_prev.ambient[ip]);
          uint4 sun = make_uint4(0u, 0u, 0u, 0u);
          if (kSkyMode == kSkyHdri) sun = fused.nee_prev.sun[ip];
          const bool deferred = (p & kParentDeferred) != 0u;
          const bool need_geo = fbits(cl.w) != kLightIdInvalid && lights_present && ((paux.w & kStVolumeScattered) == 0), need_bsdf = lc.w != 0.0f;
          const bool need_amb = !deferred && (amb.x != 0 || amb.y != 0), need_sun = kSkyMode == kSkyHdri && (sun.x != 0 || sun.y != 0);
          const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
          float4 vg = zero, vb = zero, va = zero, vs = zero;
          uint32_t word = 0u;
          const float4* __restrict__ vis = fused.fallback.vis;  // (the previous depth's visibility words: sq.vis)
          const uint32_t cap = fused.fallback.capacity;
          if (deferred) word = in.hit_scene_tri[i];
          if (need_geo) vg = ld_stream(&vis[ip]);
          if (need_bsdf) vb = ld_stream(&vis[cap + ip]);
          if (need_amb) va = ld_stream(&vis[2u * cap + ip]);
          if (need_sun) vs = vis[3u * cap + ip];
          const float4 before = results[slot];
          float ambient_vis = 0.0f;
          if (deferred) {  // nothing hit (and no cut-out skipped) = visible; an opaque nearest hit beyond eps = blocked; anything else is not decided here
            if (!(word & kHitTriHit)) { undecided = (word & kHitTriCutout) != 0u; ambient_vis = 1.0f; }
            else undecided = (word & (kHitTriBeyondEps | kHitTriOpaque)) != (kHitTriBeyondEps | kHitTriOpaque);
          }
          if (!undecided) {
            Col acc = splat(0.0f);
            acc = acc + col(cl.x, cl.y, cl.z) * col(vg.x, vg.y, vg.z);
            acc = acc + col(lc.x, lc.y, lc.z) * col(vb.x, vb.y, vb.z);
            if (kSkyMode == kSkyHdri) acc = acc + record_unpack(U2{sun.x, sun.y}) * col(vs.x, vs.y, vs.z);
            acc = acc + record_unpack(U2{amb.x, amb.y}) * (deferred ? splat(ambient_vis) : col(va.x, va.y, va.z));
            const Col v = acc * record_unpack(U2{paux.x, paux.y});
            if (any_positive(v)) results[slot] = make_float4(before.x + v.r, before.y + v.g, before.z + v.b, before.w);
          }
        }
        const unsigned long long bu = __ballot(undecided);
        if (bu) {  // rare: the vertex's own ambient ray, from its hit point along the record's packed direction (direct_lighting.cuh:388-405)
          uint32_t base = 0;
          if (lane == (uint32_t) __builtin_ctzll(bu)) {
            base = atomicAdd(ctrl + kCtlVolumeShadowItems, (uint32_t) __popcll(bu));
            atomicAdd((unsigned long long*) &counters[kCntAmbientFallback], (unsigned long long) __popcll(bu));
          }
          base = __shfl(base, __builtin_ctzll(bu));
          if (undecided) {
            const uint32_t k = base + (uint32_t) __popcll(bu & below);
            const float4 o4 = fused.prev.origin_t[ip], d4 = fused.prev.dir_slot[ip];  // the vertex's hit point as its own visibility rays had it (this path starts at the point projected onto the triangle)
            const V3 hit_origin = v3(o4.x, o4.y, o4.z) + v3(d4.x, d4.y, d4.z) * o4.w;
            const uint2 self = *reinterpret_cast<const uint2*>(&fused.prev.hit_id[ip]);
            const V3 ar = ray_unpack(U2{amb.z, amb.w});
            fused.fallback.origin_dist[k] = make_float4(hit_origin.x, hit_origin.y, hit_origin.z, kFltMax);
            fused.fallback.dir_out[k] = make_float4(ar.x, ar.y, ar.z, bitsf(2u * sq.capacity + ip));
            fused.fallback.ids[k] = make_uint4(0xFFFFFFFFu, 0u, self.x, self.y);
            fused.fallback.light_items[k] = ip;
          }
        }
      }
      if (i < n) {
        const uint32_t hit_type = in.hit_id[i].x;
        if (hit_type == kHitSky) {
          const uint4 aux = in.aux[i];
          if (kStage != 1 && (aux.w & kStAllowAmbient)) {
            if (kSkyMode == kSkyDefault) is_sky = true;  // the atmosphere is ray-marched by k_sky
            else if (kSkyMode == kSkyHdri) {
              const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
              add_to_result(results, fbits(d4.w), sky_hdri_color(sc, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), aux.w) * record_unpack(U2{aux.x, aux.y}));
            }
            else add_to_result(results, fbits(in.dir_slot[i].w), sky * record_unpack(U2{aux.x, aux.y}));
          }
        }
        else is_hit = hit_type <= kHitTriangleLimit;  // fog: scattering events are bounced by k_volume_bounce, kHitInvalid ended in k_volume_events
      }
      const unsigned long long bs = __ballot(is_sky);
      if (bs) {  // the list grows downwards from the end of the light-query list: a path is in at most one of the two
        uint32_t base = 0;
        if (lane == (uint32_t) __builtin_ctzll(bs)) base = atomicAdd(ctrl + kCtlSkyItems, (uint32_t) __popcll(bs));
        base = __shfl(base, __builtin_ctzll(bs));
        if (is_sky) sq.light_items[sq.capacity - 1u - (base + (uint32_t) __popcll(bs & below))] = i;
      }
      const unsigned long long bh = __ballot(is_hit);
      if (is_hit) pending[num_pending + (uint32_t) __popcll(bh & below)] = i;
      num_pending += (uint32_t) __popcll(bh);
    }
    if (num_pending < 64u && !(input_done && num_pending > 0u)) {
      if (input_done) break;
      continue;
    }
    __builtin_amdgcn_wave_barrier();
    LUM_LAP(clock, 8);
    const uint32_t take = min(num_pending, 64u);
    num_pending -= take;
    const bool valid = lane < take;
    const uint32_t i = valid ? pending[num_pending + lane] : 0u;
    __builtin_amdgcn_wave_barrier();
    bool survive = false, want_geo = false, want_amb = false, want_sun = false, want_lq = false;
    float4 n_o, n_d; uint4 n_aux, n_hid;
    float4 s_origin, s_geo_dir, s_amb_dir, s_sun_dir; uint4 s_geo_ids;
    bool want_amb2 = false, want_sun2 = false;  // kWater: second segments beyond the water surface
    float4 s_amb2_o, s_amb2_d, s_sun2_o, s_sun2_d;
    if (valid) {
      const float4 o4 = ld_stream(&in.origin_t[i]), d4 = ld_stream(&in.dir_slot[i]);
      const uint4 aux = ld_stream(&in.aux[i]), hid = ld_stream(&in.hit_id[i]);
      const uint32_t slot = fbits(d4.w), state = aux.w;
      const Col record_in = record_unpack(U2{aux.x, aux.y});
      {
        if (kStage != 1) vertices++;
        const V3 origin = v3(o4.x, o4.y, o4.z), ray = v3(d4.x, d4.y, d4.z);
        const V3 hit_origin = origin + ray * o4.w;
        SamplerT<kTable> smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
        smp.detect_uniform();
        smp.use_table(sc.sobol_table + (size_t) depth_const * kRndTargetCount * sc.sobol_stride, sc.sobol_stride, sc.sobol_first);  // (kTable: the pass has one)
        const GeoContext g = build_context(sc, hit_origin, ray, state, hid.x, hid.y, in.hit_scene_tri[i] & kHitTriMask, aux.z);
        if (LUM_DUP & 4) {  // (measurement: see LUM_DUP in dev_light.h)
          V3 ho = hit_origin; uint32_t tri = hid.y;
          asm volatile("" : "+v"(ho.x), "+v"(ho.y), "+v"(tri));
          const GeoContext again = build_context(sc, ho, ray, state, hid.x, tri, in.hit_scene_tri[i] & kHitTriMask, aux.z);
          dup_sink(again.position); dup_sink(again.normal); dup_sink(again.V); dup_sink(again.params.roughness()); dup_sink(again.params.emission()); dup_sink(again.face_normal_packed);
        }
        // the volume the vertex is in: without an ocean the stack holds the fog or nothing for the whole path
        const uint32_t top_volume = kWater ? volume_stack_peek(hid.w, false) : (sc.fog_active ? (uint32_t) kVolumeFog : (uint32_t) kVolumeNone);
        const uint32_t second_volume = kWater ? volume_stack_peek(hid.w, true) : (uint32_t) kVolumeNone;

        // NEE work (geometry.cuh:31-74; direct_lighting.cuh:352-443)
        const bool geo_allowed = lights_present && ((state & kStVolumeScattered) == 0);
        LUM_LAP(clock, 0);
        float4 geo_cl = make_float4(0.0f, 0.0f, 0.0f, bitsf(kLightIdInvalid));
        float4 bs_rp = make_float4(0.0f, 0.0f, 1.0f, 0.0f), bs_ws = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        s_origin = make_float4(hit_origin.x, hit_origin.y, hit_origin.z, 0.0f);
        s_geo_ids = make_uint4(0xFFFFFFFFu, 0u, hid.x, hid.y);
#ifndef LUM_ABLATE
#define LUM_ABLATE 0  // measurement only: 1 skips light sampling, 2 the BSDF light direction, 4 the bounce (results are wrong)
#endif
        float light_root_sum = 0.0f;
        if (kStage == 2) { if (geo_allowed) light_root_sum = reinterpret_cast<const float*>(&nee.bsdf_weight_sum[i])[3]; }
        else if (geo_allowed) {
          LightSample ls;
          if (LUM_ABLATE & 1) { ls.light_id = kLightIdInvalid; ls.root_sum = 1.0f; ls.color = splat(0.0f); ls.ray = v3(0.0f, 0.0f, 1.0f); ls.dist = 1.0f; }
          else ls = sample_light(sc, g, smp, clock, staged_lights);
          if (top_volume != kVolumeNone) ls.color = ls.color * volume_transmittance(sc, top_volume, g.position, ls.ray, ls.dist);  // direct_lighting.cuh:329-337
          geo_cl = make_float4(ls.color.r, ls.color.g, ls.color.b, bitsf(ls.light_id));
          light_root_sum = ls.root_sum;
          if (ls.light_id != kLightIdInvalid) {
            want_geo = true;
            const uint2 target = sc.light_tri_handles[ls.light_id];
            s_geo_dir = make_float4(ls.ray.x, ls.ray.y, ls.ray.z, ls.dist);
            s_geo_ids.x = target.x; s_geo_ids.y = target.y;
          }
        }
        if (kStage == 1) {
          st_stream(&nee.geo_color_light[i], geo_cl);
          if (geo_allowed) reinterpret_cast<float*>(&nee.bsdf_weight_sum[i])[3] = light_root_sum;
        }
        else {
        // the shading frame is formed after the light sampling: its thirteen registers need not live through the candidate loop
        const LocalFrame lf = local_frame(sc, g);
        if (LUM_DUP & 32) { const LocalFrame again = local_frame(sc, dup_launder(g)); dup_sink(again.V); dup_sink(again.face_normal); dup_sink(again.to_z.x); dup_sink(again.energy.conductor); dup_sink(again.energy.glossy); dup_sink(again.energy.dielectric); }
        if (geo_allowed) {
          LightDirSample lb;
          if (LUM_ABLATE & 2) { lb.ray = v3(0.0f, 0.0f, 1.0f); lb.weight = splat(0.0f); lb.probability = 0.0f; }
          else lb = sample_light_direction(lf, g, smp);
          if (LUM_DUP & 16) { const LightDirSample again = sample_light_direction(lf, dup_launder(g), smp); dup_sink(again.ray); dup_sink(again.weight); dup_sink(again.probability); }
          bs_rp = make_float4(lb.ray.x, lb.ray.y, lb.ray.z, lb.probability);
          if (lb.probability != 0.0f) {
            want_lq = true;
            bs_ws = make_float4(lb.weight.r, lb.weight.g, lb.weight.b, light_root_sum);
          }
          LUM_LAP(clock, 3);
        }
        BounceSample bounce;
        if (LUM_ABLATE & 4) { bounce.ray = g.normal; bounce.weight = splat(0.5f); bounce.transparent_pass = false; bounce.microfacet_based = false; }
        else bounce = sample_bounce(lf, g, smp, 0);
        if (LUM_DUP & 8) { const BounceSample again = sample_bounce(lf, dup_launder(g), smp, 0); dup_sink(again.ray); dup_sink(again.weight); dup_sink((uint32_t) again.transparent_pass); }
        uint4 amb = make_uint4(0u, 0u, 0u, 0u);
        if (kSkyMode != kSkyDefault) {  // ambient: the bounce direction doubles as the sample (direct_lighting.cuh:388-405)
          const Col ambient = (kSkyMode == kSkyHdri) ? sky_hdri_color(sc, g.position, bounce.ray, 0u) : sky;
          const U2 c = record_pack(ambient * bounce.weight), r = ray_pack(bounce.ray);
          amb = make_uint4(c.x, c.y, r.x, r.y);
          if (c.x != 0 || c.y != 0) {
            const V3 ar = ray_unpack(r);
            if (kWater) {
              const SkyRayPlan plan = plan_ambient_ray(sc, hit_origin, hid.x, top_volume, second_volume, true, ar);
              if (!plan.valid) amb = make_uint4(0u, 0u, r.x, r.y);
              else {
                want_amb = true;
                s_amb_dir = make_float4(ar.x, ar.y, ar.z, plan.limit);
                nee.amb_t1[i] = make_float4(plan.t1.r, plan.t1.g, plan.t1.b, plan.fresnel_factor);
                nee.amb_t2[i] = make_float4(plan.t2.r, plan.t2.g, plan.t2.b, bitsf(sky_ray_flags(plan)));
                if (plan.second && !plan.total_reflection) {
                  want_amb2 = true;
                  s_amb2_o = make_float4(plan.second_origin.x, plan.second_origin.y, plan.second_origin.z, kFltMax);
                  s_amb2_d = make_float4(plan.second_dir.x, plan.second_dir.y, plan.second_dir.z, 0.0f);
                }
              }
            }
            else {
              want_amb = true;
              s_amb_dir = make_float4(ar.x, ar.y, ar.z, kFltMax);
            }
          }
        }
        LUM_LAP(clock, 4);
        if (kSkyMode != kSkyConstantColor) {  // the sun: its own record and the fourth kind of visibility ray
          uint4 sun = make_uint4(0u, 0u, 0u, 0u);
          Col sun_light; V3 sun_dir;
          bool have_sun;
          if (kWater && top_volume == kVolumeOcean) have_sun = sun_caustic_sample(sc, sky_view(sc), g, smp, 0u, top_volume, second_volume, sun_light, sun_dir);  // direct_lighting.cuh:368-380
          else {
            have_sun = sample_sun(sc, sky_view(sc), lf, g, smp, sun_light, sun_dir);
            if (have_sun && top_volume != kVolumeNone) sun_light = sun_light * volume_transmittance(sc, top_volume, g.position, sun_dir, kFltMax);  // direct_lighting.cuh:104-108
          }
          if (have_sun) {
            const U2 c = record_pack(sun_light), r = ray_pack(sun_dir);
            sun = make_uint4(c.x, c.y, r.x, r.y);
            if (c.x != 0 || c.y != 0) {
              want_sun = true;
              const V3 ar = ray_unpack(r);
              s_sun_dir = make_float4(ar.x, ar.y, ar.z, kFltMax);
              if (kWater) {
                const SkyRayPlan plan = plan_sun_ray(sc, hit_origin, hid.x, top_volume, true, ar);
                s_sun_dir.w = plan.limit;
                nee.sun_water[i] = make_float4(plan.fresnel_factor, 0.0f, 0.0f, bitsf(sky_ray_flags(plan)));
                if (plan.second && !plan.total_reflection) {
                  want_sun2 = true;
                  s_sun2_o = make_float4(plan.second_origin.x, plan.second_origin.y, plan.second_origin.z, kFltMax);
                  s_sun2_d = make_float4(plan.second_dir.x, plan.second_dir.y, plan.second_dir.z, 0.0f);
                }
              }
            }
          }
          st_stream(&nee.sun[i], sun);
          LUM_LAP(clock, 5);
        }
        if (kStage == 0) st_stream(&nee.geo_color_light[i], geo_cl);
        st_stream(&nee.bsdf_ray_prob[i], bs_rp); st_stream(&nee.bsdf_weight_sum[i], bs_ws);
        st_stream(&nee.ambient[i], amb);

        // delta-path classification (geometry.cuh:80-101)
        const float roughness = g.params.roughness();
        bool is_delta;
        if (bounce.transparent_pass) {
          const float ior = g.params.ior();
          const float rs = (ior >= 1.0f) ? ior : 1.0f / ior;
          is_delta = roughness * fminf(rs - 1.0f, 1.0f) <= 0.05f;
        }
        else is_delta = bounce.microfacet_based && (roughness <= 0.05f);
        const bool pass_through = is_pass_through(g, bounce);

        // emission and throughput (geometry.cuh:103-119)
        Col record = record_in;
        const Col emission = g.params.emission();
        if (any_positive(emission)) add_to_result(results, slot, emission * record);
        record = record * bounce.weight;

        uint32_t new_state = state | kStUseIgnoreHandle;
        if (kSkyMode != kSkyDefault && !pass_through) new_state &= ~kStAllowAmbient; else new_state |= kStAllowAmbient;
        if (!is_delta) new_state &= ~kStDeltaPath;
        if (!pass_through) new_state &= ~(kStCameraDirection | kStAllowEmission);

        // russian roulette (directives.cuh:11-32)
        survive = true;
        if ((state & kStDeltaPath) == 0) {
          const float value = importance(record);
          if (value < sc.cam_rr_threshold) {
            const float p = (value > 0.0f) ? fmaxf(value / sc.cam_rr_threshold, 1.0f / 8.0f) : 0.0f;
            if (smp.next1(kRndRussianRoulette) > p) survive = false;
            else record = record * (1.0f / p);
          }
        }
        if (survive) {
          uint32_t medium = aux.z;
          if (bounce.transparent_pass) {
            const bool inside = (g.params.flags & kMatRefractionInside) != 0;
            float new_ior = 1.0f;
            if (!inside) new_ior = medium_ior_peek(medium, inside) / g.params.ior();
            medium = medium_ior_modify(medium, new_ior, !inside);
          }
          const U2 rp = record_pack(record);
          n_o = make_float4(g.position.x, g.position.y, g.position.z, kFltMax);
          n_d = make_float4(bounce.ray.x, bounce.ray.y, bounce.ray.z, d4.w);
          n_aux = make_uint4(rp.x, rp.y, medium, new_state);
          n_hid = make_uint4(g.instance_id, g.tri_id, hid.z, hid.w);
        }
        }  // kStage != 1
      }
    }
    if (kStage == 1) {  // the one list this stage appends to
      const unsigned long long bg = __ballot(want_geo);
      if (bg) {
        uint32_t base = 0;
        if (lane == (uint32_t) __builtin_ctzll(bg)) base = atomicAdd(ctrl + kCtlShadowItems, (uint32_t) __popcll(bg));
        base = __shfl(base, __builtin_ctzll(bg));
        if (want_geo) {
          const uint32_t j = base + (uint32_t) __popcll(bg & below);
          st_stream(&sq.origin_dist[j], make_float4(s_origin.x, s_origin.y, s_origin.z, s_geo_dir.w));
          st_stream(&sq.dir_out[j], make_float4(s_geo_dir.x, s_geo_dir.y, s_geo_dir.z, bitsf(i)));
          st_stream(&sq.ids[j], s_geo_ids);
        }
      }
      continue;
    }
    const bool amb_deferred = reuse_ambient && want_amb && survive;  // answered by the path's next closest-hit ray
    if (amb_deferred) { want_amb = false; vertices += 1u << 16; }   // counted in the upper half of the lane's vertex counter (a lane shades a few hundred vertices per launch)
    LUM_LAP(clock, 6);
    // Wave-aggregated appends. The three lists (survivors, visibility items, light queries) are reserved by ONE memory instruction: lanes 0-2
    // each bump one counter, so a batch waits for one atomic round trip instead of three in a row (the words sit on separate 128-byte lines).
    const unsigned long long ballot = __ballot(survive);
    const unsigned long long bg = __ballot(want_geo), ba = __ballot(want_amb), bn = __ballot(want_sun), bl = __ballot(want_lq);
    const unsigned long long b2a = kWater ? __ballot(want_amb2) : 0ull, b2s = kWater ? __ballot(want_sun2) : 0ull;
    const uint32_t ng = (uint32_t) __popcll(bg), na = (uint32_t) __popcll(ba), ns = (uint32_t) __popcll(bn), na2 = (uint32_t) __popcll(b2a);
    const unsigned long long be = announce_children ? __ballot(valid && !survive) : 0ull;  // fused resolve: vertices no entry will continue, listed for k_resolve_ended (a fourth counter, same instruction)
    const uint32_t want_count = (lane == 0) ? (uint32_t) __popcll(ballot) : (lane == 1) ? ng + na + ns + na2 + (uint32_t) __popcll(b2s) : (lane == 2) ? (uint32_t) __popcll(bl) :
                                (lane == 3) ? (uint32_t) __popcll(be) : 0u;
    uint32_t* const want_word = (lane == 0) ? count_out : (lane == 1) ? ctrl + kCtlShadowItems : (lane == 2) ? ctrl + kCtlLightItems : ctrl + kCtlSkyItems;
    uint32_t reserved = 0;
    if (want_count) reserved = atomicAdd(want_word, want_count);
    const uint32_t base_out = __builtin_amdgcn_readlane(reserved, 0), base_shadow = __builtin_amdgcn_readlane(reserved, 1), base_light = __builtin_amdgcn_readlane(reserved, 2);
    uint32_t amb_path = kNoAmbientPath;
    if (survive) {
      const uint32_t j = base_out + (uint32_t) __popcll(ballot & below);
      st_stream(&out.origin_t[j], n_o); st_stream(&out.dir_slot[j], n_d); st_stream(&out.aux[j], n_aux); st_stream(&out.hit_id[j], n_hid);
      amb_path = amb_deferred ? j : kNoAmbientPath;
      if (announce_children) out.parent[j] = i | (amb_deferred ? kParentDeferred : 0u);
    }
    if (reuse_ambient && valid && !announce_children) nee.amb_path[i] = amb_path;  // every vertex of the depth: k_resolve_reuse reads it for each of them (the fused resolve has the parent words and the list below instead)
    if (be != 0ull && valid && !survive) fused_dev->ended[__builtin_amdgcn_readlane(reserved, 3) + (uint32_t) __popcll(be & below)] = i;
    if (want_geo) {
      const uint32_t j = base_shadow + (uint32_t) __popcll(bg & below);
      st_stream(&sq.origin_dist[j], make_float4(s_origin.x, s_origin.y, s_origin.z, s_geo_dir.w));
      st_stream(&sq.dir_out[j], make_float4(s_geo_dir.x, s_geo_dir.y, s_geo_dir.z, bitsf(i)));
      st_stream(&sq.ids[j], s_geo_ids);
    }
    if (want_amb) {
      const uint32_t j = base_shadow + ng + (uint32_t) __popcll(ba & below);
      st_stream(&sq.origin_dist[j], make_float4(s_origin.x, s_origin.y, s_origin.z, s_amb_dir.w));
      st_stream(&sq.dir_out[j], make_float4(s_amb_dir.x, s_amb_dir.y, s_amb_dir.z, bitsf(2u * sq.capacity + i)));
      st_stream(&sq.ids[j], make_uint4(0xFFFFFFFFu, 0u, s_geo_ids.z, s_geo_ids.w));
    }
    if (want_sun) {
      const uint32_t j = base_shadow + ng + na + (uint32_t) __popcll(bn & below);
      st_stream(&sq.origin_dist[j], make_float4(s_origin.x, s_origin.y, s_origin.z, s_sun_dir.w));
      st_stream(&sq.dir_out[j], make_float4(s_sun_dir.x, s_sun_dir.y, s_sun_dir.z, bitsf(3u * sq.capacity + i)));
      st_stream(&sq.ids[j], make_uint4(0xFFFFFFFFu, 0u, s_geo_ids.z, s_geo_ids.w));
    }
    if (kWater) {  // second segments beyond the water surface
      if (want_amb2) {
        const uint32_t j = base_shadow + ng + na + ns + (uint32_t) __popcll(b2a & below);
        sq.origin_dist[j] = s_amb2_o;
        sq.dir_out[j] = make_float4(s_amb2_d.x, s_amb2_d.y, s_amb2_d.z, bitsf(kShadowKindAmbient2 * sq.capacity + i));
        sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
      }
      if (want_sun2) {
        const uint32_t j = base_shadow + ng + na + ns + na2 + (uint32_t) __popcll(b2s & below);
        sq.origin_dist[j] = s_sun2_o;
        sq.dir_out[j] = make_float4(s_sun2_d.x, s_sun2_d.y, s_sun2_d.z, bitsf(kShadowKindSun2 * sq.capacity + i));
        sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
      }
    }
    if (want_lq) sq.light_items[base_light + (uint32_t) __popcll(bl & below)] = i;
    LUM_LAP(clock, 7);
  }
  clock.flush();
  uint32_t deferred = vertices >> 16;
  vertices &= 0xFFFFu;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { vertices += __shfl_down(vertices, off); deferred += __shfl_down(deferred, off); }
  if ((threadIdx.x & 63) == 0 && vertices) atomicAdd((unsigned long long*) &counters[kCntVertices], (unsigned long long) vertices);
  if ((threadIdx.x & 63) == 0 && deferred) atomicAdd((unsigned long long*) &counters[kCntAmbientDeferred], (unsigned long long) deferred);
}

// ---- debug shading modes (settings.shading_mode != DEFAULT): one closest-hit pass, then a colour per hit (geometry_process_tasks_debug,
// cuda/geometry.cuh:182-246) or per miss (sky_process_tasks_debug, cuda/sky.cuh:635-665); queue: device/device_renderer.c:136-181 ----
__global__ __launch_bounds__(kBlock) void k_shade_debug(DeviceScene sc, PathQueue in, float4* results, const uint32_t* ctrl) {
  const uint32_t n = ctrl[kCtlPaths];
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
    const uint4 aux = in.aux[i], hid = in.hit_id[i];
    const V3 origin = v3(o4.x, o4.y, o4.z), ray = v3(d4.x, d4.y, d4.z);
    Col result = splat(0.0f);
    if (hid.x == kHitSky) {
      if (sc.shading_mode == 1u) {  // ALBEDO: sky_color_main(origin, ray, STATE_FLAG_CAMERA_DIRECTION)
        if (sc.sky_mode == kSkyDefault) {
          if (sc.sky_lut_transmittance && sc.sky_lut_multiscattering) {
            const SkyView sky = sky_view(sc);
            const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), 0u};
            result = sky_get_color(sc, sky, world_to_sky(sky, origin), ray, kFltMax, true, (int) sky.steps, smp.next1(kRndSkyStepOffset));
          }
        }
        else if (sc.sky_mode == kSkyHdri) result = sky_hdri_color(sc, origin, ray, kStCameraDirection);
        else result = col(sc.sky_constant_color[0], sc.sky_constant_color[1], sc.sky_constant_color[2]);
      }
      else if (sc.shading_mode == 4u) result = col(0.0f, 0.63f, 1.0f);  // IDENTIFICATION
    }
    else if (hid.x == kHitOcean) {  // ocean_process_tasks_debug, ocean.cuh:104-145
      if (sc.shading_mode == 2u) result = splat(saturate((1.0f / o4.w) * 2.0f));
      else if (sc.shading_mode == 3u) {
        const V3 nrm = ocean_get_normal(sc, origin + ray * o4.w);
        result = col(saturate(0.5f * nrm.x + 0.5f), saturate(0.5f * nrm.y + 0.5f), saturate(0.5f * nrm.z + 0.5f));
      }
      else if (sc.shading_mode == 4u) result = col(0.0f, 0.0f, 1.0f);
    }
    else if (particle_is_hit(hid.x)) {  // particle_process_tasks_debug, particle.cuh:110-163
      if (sc.shading_mode == 1u) result = particles_albedo(sc);
      else if (sc.shading_mode == 2u) result = splat(saturate((1.0f / o4.w) * 2.0f));
      else if (sc.shading_mode == 3u) {
        const V3 nrm = particle_context(sc, origin, ray, aux.w, hid.x).normal;
        result = col(saturate(nrm.x), saturate(nrm.y), saturate(nrm.z));
      }
      else if (sc.shading_mode == 4u) {
        const uint32_t v = squares32(0x55555555u, hid.x);
        result = col(((float) (v & 0x7FFu)) / 0x7FF, ((float) ((v >> 10) & 0x7FFu)) / 0x7FF, ((float) ((v >> 20) & 0x7FFu)) / 0x7FF);
      }
    }
    else if (hid.x <= kHitTriangleLimit) {  // with fog: scattering events and paths the sky fast path ended have no debug colour
      const V3 hit_origin = origin + ray * o4.w;
      if (sc.shading_mode == 2u) result = splat(saturate((1.0f / o4.w) * 2.0f));  // DEPTH
      else if (sc.shading_mode == 4u) {                                            // IDENTIFICATION
        const uint32_t v = squares32(0x55555555u, (hid.x << 16) | hid.y);
        result = col(((float) (v & 0x7FFu)) / 0x7FF, ((float) ((v >> 10) & 0x7FFu)) / 0x7FF, ((float) ((v >> 20) & 0x7FFu)) / 0x7FF);
      }
      else if (sc.shading_mode == 1u || sc.shading_mode == 3u || sc.shading_mode == 5u) {
        const GeoContext g = build_context(sc, hit_origin, ray, aux.w, hid.x, hid.y, in.hit_scene_tri[i] & kHitTriMask, aux.z);
        if (sc.shading_mode == 1u) result = g.params.albedo() + g.params.emission();                    // ALBEDO
        else if (sc.shading_mode == 3u) result = col(saturate(g.normal.x), saturate(g.normal.y), saturate(g.normal.z));  // NORMAL
        else result = g.params.albedo() * 0.025f + g.params.emission();                                   // LIGHTS
      }
    }
    add_to_result(results, fbits(d4.w), result);
  }
}

// ---- paths that left the scene, procedural sky: sky_process_tasks (cuda/sky.cuh:609-633) with sky_color_main's DEFAULT branch ----
__global__ __launch_bounds__(kBlock) void k_sky(DeviceScene sc, PathQueue in, ShadowQueue sq, float4* results, const uint32_t* ctrl, uint32_t depth_const) {
  const uint32_t n = ctrl[kCtlSkyItems];
  const SkyView sky = sky_view(sc);
  for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
    const uint32_t i = sq.light_items[sq.capacity - 1u - k];
    const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
    const uint4 aux = in.aux[i], hid = in.hit_id[i];
    const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
    const V3 sky_origin = world_to_sky(sky, v3(o4.x, o4.y, o4.z));
    const bool include_sun = (aux.w & (kStCameraDirection | kStAllowEmission)) != 0;
    const Col c = sky_get_color(sc, sky, sky_origin, v3(d4.x, d4.y, d4.z), kFltMax, include_sun, (int) sky.steps, smp.next1(kRndSkyStepOffset));
    add_to_result(results, fbits(d4.w), c * record_unpack(U2{aux.x, aux.y}));
  }
}

// ---- aerial perspective: sky_process_inscattering_events (cuda/kernels.cuh:357-388), between the closest-hit pass and shading ----
__global__ __launch_bounds__(kBlock, LUM_FEATURE_WAVES) void k_sky_inscattering(DeviceScene sc, PathQueue in, float4* results, const uint32_t* ctrl, uint32_t depth_const) {
  const uint32_t n = ctrl[kCtlPaths];
  const SkyView sky = sky_view(sc);
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const uint4 hid = in.hit_id[i];
    if (hid.x == kHitSky) continue;
    const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
    uint4 aux = in.aux[i];
    const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
    Col record = record_unpack(U2{aux.x, aux.y});
    const Col c = sky_trace_inscattering(sc, sky, world_to_sky(sky, v3(o4.x, o4.y, o4.z)), v3(d4.x, d4.y, d4.z), o4.w * 0.001f, record, depth_const == 0u,
                                         smp.next1(kRndSkyInscatteringStep), smp.next1(kRndSkyStepOffset));
    add_to_result(results, fbits(d4.w), c);
    const U2 rp = record_pack(record);
    aux.x = rp.x; aux.y = rp.y;
    in.aux[i] = aux;
  }
}

// ---- light queries: BSDF-sampled direction against the light-only BVH (cuda/direct_lighting.cuh:586-667) ----
__global__ __launch_bounds__(kBlock) void k_light_query(DeviceScene sc, PathQueue in, NeeQueue nee, ShadowQueue sq, uint32_t* ctrl, uint32_t depth_const,
                                                        uint64_t* counters) {
  const uint32_t n = ctrl[kCtlLightItems];
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  RayStats st{0, 0, 0};
  uint32_t light_queries = 0;
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  for (uint32_t round = 0; round < rounds; round++) {
    const uint32_t item = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
    bool want = false;
    float4 s_origin, s_dir; uint4 s_ids; uint32_t i = 0;
    if (item < n) {
      i = sq.light_items[item];
      const uint4 hid = in.hit_id[i];
      const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
      const V3 hit_origin = v3(o4.x, o4.y, o4.z) + v3(d4.x, d4.y, d4.z) * o4.w;
      const float4 rp = nee.bsdf_ray_prob[i], ws = nee.bsdf_weight_sum[i];
      const V3 ray = v3(rp.x, rp.y, rp.z);
      const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
      uint32_t num_hits = 0;
      const uint32_t light_id = light_query(sc, hit_origin, ray, hid.x, hid.y, smp.next1(kRndLightBsdfTrace), num_hits, st);
      light_queries++;
      bool valid = light_id != kLightIdInvalid;
      float dist = kFltMax;
      uint2 handle = make_uint2(0xFFFFFFFFu, 0u);
      Col lc = splat(0.0f);
      if (light_id != kLightIdInvalid) {
        handle = sc.light_tri_handles[light_id];
        const TableLight entry = load_tri_light_table(sc, light_id);  // the light's line of the table the candidate loop reads (dev_light.h): no chain handle -> mesh -> vertices / transform / material
        const TriLight& tl = entry.tri;
        F2 uv;
        dist = intersect_triangle(tl.vertex, tl.edge1, tl.edge2, hit_origin, ray, uv);
        if (dist != kFltMax) {
          lc = entry.textured ? tri_light_color(sc, tl, uv) : entry.color;
          const float mis = (ws.w == 0.0f) ? 1.0f : mis_base(rp.w, tri_light_solid_angle(tl, hit_origin), importance(lc) * entry.area, dist * dist, ws.w);  // mis_for_bsdf_ray with the table's area
          lc = lc * (mis * num_hits);
          lc = lc * col(ws.x, ws.y, ws.z);
        }
        else valid = false;
      }
      nee.bsdf_weight_sum[i] = make_float4(lc.r, lc.g, lc.b, valid ? 1.0f : 0.0f);
      if (valid && (sc.fog_active || sc.ocean_active)) {  // the volume's transmittance up to the light (direct_lighting.cuh:661-666) replaces the direction, which is used up
        const Col seen = volume_transmittance(sc, sc.ocean_active ? volume_stack_peek(hid.w, false) : (uint32_t) kVolumeFog, hit_origin, ray, dist);
        nee.bsdf_ray_prob[i] = make_float4(seen.r, seen.g, seen.b, 0.0f);
      }
      if (valid) {
        want = true;
        s_origin = make_float4(hit_origin.x, hit_origin.y, hit_origin.z, dist);
        s_dir = make_float4(ray.x, ray.y, ray.z, bitsf(sq.capacity + i));
        s_ids = make_uint4(handle.x, handle.y, hid.x, hid.y);
      }
    }
    const unsigned long long bw = __ballot(want);
    if (bw) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(ctrl + kCtlShadowItems, (uint32_t) __popcll(bw));
      base = __builtin_amdgcn_readfirstlane(base);
      if (want) {
        const uint32_t j = base + (uint32_t) __popcll(bw & below);
        sq.origin_dist[j] = s_origin; sq.dir_out[j] = s_dir; sq.ids[j] = s_ids;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) light_queries += __shfl_down(light_queries, off);
  if ((threadIdx.x & 63) == 0 && light_queries) atomicAdd((unsigned long long*) &counters[kCntLightBvh], (unsigned long long) light_queries);
  flush_stats(counters, st, 0, kCntLightBvh, kCntNodesLight, kCntTrisLight);
}

// ---- visibility rays: kernel_shadow.h (its own translation unit in the fast flavour, see there) ----
// ---- resolve: optix/optix_kernel_shadow.cu:15-100 sums sampled light, BSDF-sampled light, (sun,) ambient, then weights ----
// kAmbientKnown: the ambient sample's visibility is `ambient_vis` (from the path's next closest hit) instead of the visibility pass's word.
LUM_DEV bool resolves_here(uint32_t hit_type) {  // sky, and with volumes: scattering events and ended paths have nothing to resolve
  return !(hit_type > kHitTriangleLimit && !particle_is_hit(hit_type) && hit_type != kHitOcean);
}

__global__ __launch_bounds__(kBlock) void k_resolve(DeviceScene sc, PathQueue in, NeeQueue nee, ShadowQueue sq, float4* results, const uint32_t* ctrl) {
  const uint32_t n = ctrl[kCtlPaths];
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    if (!resolves_here(in.hit_id[i].x)) continue;
    resolve_vertex<false>(sc, in, nee, sq, results, i, lights_present, 0.0f);
  }
}

// Fused resolve (FusedResolve, above k_shade): the vertices of a depth that no entry of the next depth continues - the path ended there, its ambient ray was
// traced with the depth's visibility rays - are resolved here, after that pass; the others by the entries that continue them.
__global__ __launch_bounds__(kBlock) void k_resolve_ended(DeviceScene sc, PathQueue in, NeeQueue nee, ShadowQueue sq, float4* results, const uint32_t* ctrl, const uint32_t* list) {
  const uint32_t n = ctrl[kCtlSkyItems];
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock)
    resolve_vertex<false>(sc, in, nee, sq, results, list[k], lights_present, 0.0f);
}

// The resolve of a depth whose ambient samples were left to the next depth's closest-hit pass (ambient-visibility reuse, above TraceQuery): `next` is the
// queue that pass has just traced. A deferred sample reads its answer from the path's hit: nothing hit (and no cut-out skipped) = visible, an opaque
// nearest hit beyond eps = blocked - what a visibility ray over (eps, FLT_MAX) that ignores the vertex's own triangle reports. Undecided vertices are not
// resolved here: their ambient ray is queued (items from index 0: the depth's own items are spent) and their index listed for k_resolve_listed.
__global__ __launch_bounds__(kBlock) void k_resolve_reuse(DeviceScene sc, PathQueue in, PathQueue next, NeeQueue nee, ShadowQueue sq, float4* results, uint32_t* ctrl, uint64_t* counters) {
  const uint32_t n = ctrl[kCtlPaths];
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  for (uint32_t round = 0; round < rounds; round++) {
    const uint32_t i = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
    bool undecided = false;
    uint32_t j = kNoAmbientPath;
    uint2 self = make_uint2(0u, 0u);
    if (i < n) {
      const uint2 hid = *reinterpret_cast<const uint2*>(&in.hit_id[i]);
      j = nee.amb_path[i];  // (stale for an entry that was not shaded: only read below when the entry resolves)
      if (!resolves_here(hid.x)) j = kNoAmbientPath;
      else {
        const ResolveRecords rec = load_resolve_records(in, nee, i);
        if (!kReuseProves) {  // one call either way: the hit word of a deferred sample travels with the visibility words and the result slot
          if (!resolve_records<2>(sc, in, nee, sq, results, i, lights_present, 0.0f, rec, (j == kNoAmbientPath) ? nullptr : &next.hit_scene_tri[j])) { undecided = true; self = hid; }
        }
        else if (j == kNoAmbientPath) resolve_records<0>(sc, in, nee, sq, results, i, lights_present, 0.0f, rec);  // no ambient sample, or its visibility ray was traced (the path ended here)
        else {
          const uint32_t word = next.hit_scene_tri[j];
          int v;
          if (!(word & kHitTriHit)) v = (word & kHitTriCutout) ? -1 : 1;
          else v = ((word & (kHitTriBeyondEps | kHitTriOpaque)) == (kHitTriBeyondEps | kHitTriOpaque)) ? 0 : -1;
          if (kReuseProves) {
            // The exact flavour answers for the ambient ray ITSELF (from the hit point, along the record's packed direction - a last bit away from the ray the
            // path went on along): "blocked" only if that ray, mapped into the instance as a traversal maps it, passes the any-hit test of the very triangle
            // the closest-hit ray ended at (opaque on its own; the boxes are conservative, so the visibility traversal is certain to reach a triangle the
            // exact test accepts, and one opaque hit decides the ray whatever else it crosses). "Nothing in the way" cannot be proved from another ray:
            // those samples are traced. On a closed scene like the hall that still answers nine ambient rays in ten, bit for bit.
            if (v == 0) {
              const uint32_t inst = next.hit_id[j].x, stri = word & kHitTriMask;
              const float4 a = sc.vertices[3u * stri], b = sc.vertices[3u * stri + 1u], c = sc.vertices[3u * stri + 2u];
              const V3 p0 = v3(a.x, a.y, a.z), e1 = v3(b.x - a.x, b.y - a.y, b.z - a.z), e2 = v3(c.x - a.x, c.y - a.y, c.z - a.z);  // the traversal triangle's edges (core.hip)
              const float4 r0 = sc.instance_rows[3u * inst], r1 = sc.instance_rows[3u * inst + 1u], r2 = sc.instance_rows[3u * inst + 2u];
              const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
              const V3 wo = v3(o4.x, o4.y, o4.z) + v3(d4.x, d4.y, d4.z) * o4.w;
              const uint4 amb = nee.ambient[i];
              const V3 wd = ray_unpack(U2{amb.z, amb.w});
              const float px = wo.x - r0.w, py = wo.y - r1.w, pz = wo.z - r2.w;
              const V3 oo = v3(mat_row_apply(r0.x, r0.y, r0.z, px, py, pz), mat_row_apply(r1.x, r1.y, r1.z, px, py, pz), mat_row_apply(r2.x, r2.y, r2.z, px, py, pz));
              const V3 od = v3(mat_row_apply(r0.x, r0.y, r0.z, wd.x, wd.y, wd.z), mat_row_apply(r1.x, r1.y, r1.z, wd.x, wd.y, wd.z),
                               mat_row_apply(r2.x, r2.y, r2.z, wd.x, wd.y, wd.z));
              F2 uv;
              const float t = intersect_triangle(p0, e1, e2, oo, od, uv);
              if (!(t > kEps && t < kFltMax)) v = -1;
            }
            else if (v == 1) v = -1;
          }
          if (v >= 0) resolve_records<1>(sc, in, nee, sq, results, i, lights_present, (float) v, rec);
          else { undecided = true; self = hid; }  // (j is not needed any more: the ray below is the vertex's own ambient ray)
        }
      }
    }
    const unsigned long long bu = __ballot(undecided);
    if (bu) {
      uint32_t base = 0;
      if (lane == (uint32_t) __builtin_ctzll(bu)) {
        base = atomicAdd(ctrl + kCtlVolumeShadowItems, (uint32_t) __popcll(bu));
        atomicAdd((unsigned long long*) &counters[kCntAmbientFallback], (unsigned long long) __popcll(bu));
      }
      base = __shfl(base, __builtin_ctzll(bu));
      if (undecided) {
        const uint32_t k = base + (uint32_t) __popcll(bu & below);
        // exactly the item k_shade would have queued: from the hit point along the ambient record's packed direction (direct_lighting.cuh:388-405)
        const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
        const V3 hit_origin = v3(o4.x, o4.y, o4.z) + v3(d4.x, d4.y, d4.z) * o4.w;
        const uint4 amb = nee.ambient[i];
        const V3 ar = ray_unpack(U2{amb.z, amb.w});
        sq.origin_dist[k] = make_float4(hit_origin.x, hit_origin.y, hit_origin.z, kFltMax);
        sq.dir_out[k] = make_float4(ar.x, ar.y, ar.z, bitsf(2u * sq.capacity + i));
        sq.ids[k] = make_uint4(0xFFFFFFFFu, 0u, self.x, self.y);
        sq.light_items[k] = i;
      }
    }
  }
}

// ... and the vertices k_resolve_reuse listed, after the second visibility pass has traced their ambient rays.
__global__ __launch_bounds__(kBlock) void k_resolve_listed(DeviceScene sc, PathQueue in, NeeQueue nee, ShadowQueue sq, float4* results, const uint32_t* ctrl) {
  const uint32_t n = ctrl[kCtlVolumeShadowItems];
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock)
    resolve_vertex<false>(sc, in, nee, sq, results, sq.light_items[k], lights_present, 0.0f);
}


// ---- particles ----
// The particle pass of the closest-hit kernel (optix_kernel_raytrace.cu:97-131): delta paths are traced against the tiled particle cell, in a
// space scaled by 1 / particles_scale so that ray parameters stay world distances; a hit closer than the surface hit replaces it. Runs the
// same traversal on the particle tree (the scene argument carries that tree in place of the surfaces').
struct ParticleQuery {
  static constexpr bool kDual = false;
  static constexpr bool kSpeculate = false;
  static constexpr bool kOrdered = true;
  static constexpr int kFarFirst = 0;
  static constexpr bool kCull = true;
  PathQueue q;
  float best_t;
  uint32_t best_inst, best_tri;
  float particles_scale, particles_speed;
  V3 direction;
  const uint32_t* bluenoise;
  LUM_DEV bool load(const DeviceScene&, uint32_t i, V3& o, V3& d, float& tmax) {
    const uint4 aux = q.aux[i];
    if ((aux.w & kStDeltaPath) == 0) return false;  // particles are invisible to non-delta paths (negligible contribution)
    const float4 o4 = q.origin_t[i], d4 = q.dir_slot[i];
    const uint4 hid = q.hit_id[i];
    const Sampler smp{bluenoise, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), 0u};  // random_1D_consistent: depth 0
    const float time = smp.next1(kRndCameraTime);
    const V3 motion_offset = direction * (time * particles_speed);
    d = v3(d4.x, d4.y, d4.z) * (1.0f / particles_scale);
    V3 pos = (v3(o4.x, o4.y, o4.z) + motion_offset) * (1.0f / particles_scale);
    pos.x = pos.x - floorf(pos.x); pos.y = pos.y - floorf(pos.y); pos.z = pos.z - floorf(pos.z);
    o = pos;
    tmax = o4.w;
    best_t = o4.w; best_inst = 0xFFFFFFFFu; best_tri = 0xFFFFFFFFu;
    return true;
  }
  LUM_DEV bool on_tris(const DeviceScene& sc, uint32_t inst, uint32_t first, uint32_t count, V3 o, V3 d, float& tmax, RayStats& st) {
    LeafTris lt;
    lt.load<false>(sc.blas_tris, first, count);
#pragma unroll
    for (uint32_t j = 0; j < kBvhLeafMaxTri; j++) {
      if (j >= count) break;
      const float4 a = lt.a[j], b = lt.b[j], c = lt.c[j];
      const uint32_t id = fbits(a.w);
      st.tris++;
      F2 uv;
      const float t = intersect_triangle(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), o, d, uv);
      if (t < best_t || (t == best_t && t != kFltMax && best_tri != 0xFFFFFFFFu && (inst < best_inst || (inst == best_inst && id < best_tri)))) {
        const float dx = uv.x - 0.5f, dy = uv.y - 0.5f;  // particle_opacity_cutout, optix_common.cuh:67-74
        if (dx * dx + dy * dy > 0.25f) continue;
        best_t = t; best_inst = inst; best_tri = id; tmax = t;
      }
    }
    return false;
  }
  LUM_DEV void finish(const DeviceScene&, uint32_t i) {
    if (best_tri == 0xFFFFFFFFu) return;
    q.origin_t[i].w = best_t;
    uint4 hid = q.hit_id[i];
    hid.x = kHitParticleMin + (best_tri >> 1); hid.y = 0u;
    q.hit_id[i] = hid;
  }
};

__global__ LUM_TRACE_BOUNDS void k_trace_particles(DeviceScene particle_tree, PathQueue q, uint32_t* ctrl, uint32_t lds_nodes) {
  RayStats st{0, 0, 0};
  uint32_t rays = 0;
  ParticleQuery pq;
  pq.q = q;
  pq.particles_scale = particle_tree.particles_scale; pq.particles_speed = particle_tree.particles_speed;
  pq.direction = v3(particle_tree.particles_direction[0], particle_tree.particles_direction[1], particle_tree.particles_direction[2]);
  pq.bluenoise = particle_tree.bluenoise_2d;
  LUM_TRACE_ITEMS(particle_tree, ctrl[kCtlPaths], ctrl + kCtlParticleCursor, pq, st, rays, lds_nodes);
}

// particle_process_tasks (particle.cuh:7-108): light sample, sun, phase-function bounce whose direction doubles as the ambient sample. The
// records go where a surface vertex puts them, so k_resolve and the visibility pass treat both alike (optix_kernel_shadow.cu covers both).
__global__ __launch_bounds__(kBlock, LUM_FEATURE_WAVES) void k_particle_shade(DeviceScene sc, PathQueue in, PathQueue out, NeeQueue nee, ShadowQueue sq, uint32_t* ctrl, uint32_t depth_const) {
  const uint32_t n = ctrl[kCtlPaths];
  uint32_t* count_out = ctrl + kCtlStride + kCtlPaths;
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  const bool sun_allowed = sc.sky_mode != kSkyConstantColor && sc.sky_lut_transmittance != nullptr && sc.sky_lut_multiscattering != nullptr;
  const Col albedo = particles_albedo(sc);
  // Particle hits are a sparse subset of the queue: a wave collects their indices in LDS and shades them 64 at a time (as k_shade does with surface hits)
  __shared__ uint32_t pending_hits[kBlock / 64][128];
  uint32_t* pending = pending_hits[threadIdx.x >> 6];
  uint32_t num_pending = 0;  // wave-uniform
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  for (uint32_t round = 0;; round++) {
    const bool input_done = round >= rounds;
    if (!input_done) {
      const uint32_t idx = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
      const bool is_hit = idx < n && particle_is_hit(in.hit_id[idx].x);
      const unsigned long long bh = __ballot(is_hit);
      if (is_hit) pending[num_pending + (uint32_t) __popcll(bh & below)] = idx;
      num_pending += (uint32_t) __popcll(bh);
    }
    if (num_pending < (LUM_HIT_COMPACT ? 64u : 1u) && !(input_done && num_pending > 0u)) {
      if (input_done) break;
      continue;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t take = min(num_pending, 64u);
    num_pending -= take;
    const bool valid = lane < take;
    const uint32_t i = valid ? pending[num_pending + lane] : 0u;
    __builtin_amdgcn_wave_barrier();
    bool survive = false, want_geo = false, want_amb = false, want_sun = false, want_amb2 = false, want_sun2 = false;
    float4 n_o, n_d; uint4 n_aux, n_hid;
    float4 s_origin, s_geo_dir, s_amb_dir, s_sun_dir; uint4 s_geo_ids;
    float4 s_amb2_o, s_amb2_d, s_sun2_o, s_sun2_d;
    if (valid) {
      const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
      const uint4 aux = in.aux[i], hid = in.hit_id[i];
      const uint32_t state = aux.w;
      const V3 ray = v3(d4.x, d4.y, d4.z);
      const V3 position = v3(o4.x, o4.y, o4.z) + ray * o4.w;
      const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
      const ParticleContext pc = particle_context(sc, position, ray, state, hid.x);
      const uint32_t top_volume = volume_stack_peek(hid.w, false), second_volume = volume_stack_peek(hid.w, true);
      s_origin = make_float4(position.x, position.y, position.z, 0.0f);
      s_geo_ids = make_uint4(0xFFFFFFFFu, 0u, hid.x, 0u);
      float4 geo_cl = make_float4(0.0f, 0.0f, 0.0f, bitsf(kLightIdInvalid));
      if (lights_present && (state & kStVolumeScattered) == 0) {
        LightSample ls = particle_light_sample(sc, pc, smp);
        if (top_volume != kVolumeNone) ls.color = ls.color * volume_transmittance(sc, top_volume, position, ls.ray, ls.dist);
        geo_cl = make_float4(ls.color.r, ls.color.g, ls.color.b, bitsf(ls.light_id));
        if (ls.light_id != kLightIdInvalid) {
          want_geo = true;
          const uint2 target = sc.light_tri_handles[ls.light_id];
          s_geo_dir = make_float4(ls.ray.x, ls.ray.y, ls.ray.z, ls.dist);
          s_geo_ids.x = target.x; s_geo_ids.y = target.y;
        }
      }
      uint4 sun = make_uint4(0u, 0u, 0u, 0u);
      if (sun_allowed) {
        Col sun_light; V3 sun_dir;
        const bool have_sun = (top_volume == kVolumeOcean) ? sun_caustic_sample(sc, sky_view(sc), pc, smp, 0u, top_volume, second_volume, sun_light, sun_dir)
                                                            : particle_sun_sample(sc, sky_view(sc), pc, top_volume, smp, sun_light, sun_dir);
        if (have_sun) {
          const U2 c = record_pack(sun_light), r = ray_pack(sun_dir);
          sun = make_uint4(c.x, c.y, r.x, r.y);
          if (c.x != 0 || c.y != 0) {
            want_sun = true;
            const V3 ar = ray_unpack(r);
            s_sun_dir = make_float4(ar.x, ar.y, ar.z, kFltMax);
            if (sc.ocean_active) {
              const SkyRayPlan plan = plan_sun_ray(sc, position, hid.x, top_volume, true, ar);
              s_sun_dir.w = plan.limit;
              nee.sun_water[i] = make_float4(plan.fresnel_factor, 0.0f, 0.0f, bitsf(sky_ray_flags(plan)));
              if (plan.second && !plan.total_reflection) {
                want_sun2 = true;
                s_sun2_o = make_float4(plan.second_origin.x, plan.second_origin.y, plan.second_origin.z, kFltMax);
                s_sun2_d = make_float4(plan.second_dir.x, plan.second_dir.y, plan.second_dir.z, 0.0f);
              }
            }
          }
        }
      }
      // bsdf_sample<MATERIAL_PARTICLE> with RANDOM_GI (bsdf.cuh:320-331): weight = albedo
      const float random_choice = smp.next1(kRndBsdfGiResampling);
      const F2 random_dir = smp.next2(kRndBsdfGiDiffuse);
      const V3 bounce = je_phase_sample(sc.particles_phase, ray, random_dir, random_choice);
      uint4 amb = make_uint4(0u, 0u, 0u, 0u);
      if (sc.sky_mode != kSkyDefault) {
        const U2 c = record_pack(sky_color_no_compute(sc, position, bounce, 0u) * albedo), r = ray_pack(bounce);
        amb = make_uint4(c.x, c.y, r.x, r.y);
        if (c.x != 0 || c.y != 0) {
          const V3 ar = ray_unpack(r);
          if (sc.ocean_active) {
            const SkyRayPlan plan = plan_ambient_ray(sc, position, hid.x, top_volume, second_volume, true, ar);
            if (!plan.valid) amb = make_uint4(0u, 0u, r.x, r.y);
            else {
              want_amb = true;
              s_amb_dir = make_float4(ar.x, ar.y, ar.z, plan.limit);
              nee.amb_t1[i] = make_float4(plan.t1.r, plan.t1.g, plan.t1.b, plan.fresnel_factor);
              nee.amb_t2[i] = make_float4(plan.t2.r, plan.t2.g, plan.t2.b, bitsf(sky_ray_flags(plan)));
              if (plan.second && !plan.total_reflection) {
                want_amb2 = true;
                s_amb2_o = make_float4(plan.second_origin.x, plan.second_origin.y, plan.second_origin.z, kFltMax);
                s_amb2_d = make_float4(plan.second_dir.x, plan.second_dir.y, plan.second_dir.z, 0.0f);
              }
            }
          }
          else { want_amb = true; s_amb_dir = make_float4(ar.x, ar.y, ar.z, kFltMax); }
        }
      }
      nee.geo_color_light[i] = geo_cl;
      nee.bsdf_ray_prob[i] = make_float4(0.0f, 0.0f, 1.0f, 0.0f);
      nee.bsdf_weight_sum[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // no BSDF-sampled light for particles (direct_lighting.cuh:308-317)
      nee.ambient[i] = amb;
      if (sc.sky_mode != kSkyConstantColor) nee.sun[i] = sun;
      uint32_t new_state = state & ~(kStDeltaPath | kStCameraDirection | kStAllowEmission | kStUseIgnoreHandle);
      if (sc.sky_mode != kSkyDefault) new_state &= ~kStAllowAmbient; else new_state |= kStAllowAmbient;
      Col record = record_unpack(U2{aux.x, aux.y}) * albedo;
      survive = true;
      if ((state & kStDeltaPath) == 0) {  // directives.cuh:11-32
        const float value = importance(record);
        if (value < sc.cam_rr_threshold) {
          const float p = (value > 0.0f) ? fmaxf(value / sc.cam_rr_threshold, 1.0f / 8.0f) : 0.0f;
          if (smp.next1(kRndRussianRoulette) > p) survive = false;
          else record = record * (1.0f / p);
        }
      }
      if (survive) {
        const U2 rp = record_pack(record);
        n_o = make_float4(position.x, position.y, position.z, kFltMax);
        n_d = make_float4(bounce.x, bounce.y, bounce.z, d4.w);
        n_aux = make_uint4(rp.x, rp.y, aux.z, new_state);
        n_hid = make_uint4(0u, 0u, hid.z, hid.w);
      }
    }
    const unsigned long long ballot = __ballot(survive);
    if (ballot) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(count_out, (uint32_t) __popcll(ballot));
      base = __builtin_amdgcn_readfirstlane(base);
      if (survive) {
        const uint32_t j = base + (uint32_t) __popcll(ballot & below);
        out.origin_t[j] = n_o; out.dir_slot[j] = n_d; out.aux[j] = n_aux; out.hit_id[j] = n_hid;
      }
    }
    const unsigned long long bg = __ballot(want_geo), ba = __ballot(want_amb), bn = __ballot(want_sun);
    if (bg | ba | bn) {
      const uint32_t ng = (uint32_t) __popcll(bg), na = (uint32_t) __popcll(ba);
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(ctrl + kCtlShadowItems, ng + na + (uint32_t) __popcll(bn));
      base = __builtin_amdgcn_readfirstlane(base);
      if (want_geo) {
        const uint32_t j = base + (uint32_t) __popcll(bg & below);
        sq.origin_dist[j] = make_float4(s_origin.x, s_origin.y, s_origin.z, s_geo_dir.w);
        sq.dir_out[j] = make_float4(s_geo_dir.x, s_geo_dir.y, s_geo_dir.z, bitsf(i));
        sq.ids[j] = s_geo_ids;
      }
      if (want_amb) {
        const uint32_t j = base + ng + (uint32_t) __popcll(ba & below);
        sq.origin_dist[j] = make_float4(s_origin.x, s_origin.y, s_origin.z, s_amb_dir.w);
        sq.dir_out[j] = make_float4(s_amb_dir.x, s_amb_dir.y, s_amb_dir.z, bitsf(2u * sq.capacity + i));
        sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, s_geo_ids.z, s_geo_ids.w);
      }
      if (want_sun) {
        const uint32_t j = base + ng + na + (uint32_t) __popcll(bn & below);
        sq.origin_dist[j] = make_float4(s_origin.x, s_origin.y, s_origin.z, s_sun_dir.w);
        sq.dir_out[j] = make_float4(s_sun_dir.x, s_sun_dir.y, s_sun_dir.z, bitsf(3u * sq.capacity + i));
        sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, s_geo_ids.z, s_geo_ids.w);
      }
    }
    const unsigned long long b2a = __ballot(want_amb2), b2s = __ballot(want_sun2);
    if (b2a | b2s) {  // second segments beyond the water surface
      const uint32_t na2 = (uint32_t) __popcll(b2a);
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(ctrl + kCtlShadowItems, na2 + (uint32_t) __popcll(b2s));
      base = __builtin_amdgcn_readfirstlane(base);
      if (want_amb2) {
        const uint32_t j = base + (uint32_t) __popcll(b2a & below);
        sq.origin_dist[j] = s_amb2_o;
        sq.dir_out[j] = make_float4(s_amb2_d.x, s_amb2_d.y, s_amb2_d.z, bitsf(kShadowKindAmbient2 * sq.capacity + i));
        sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
      }
      if (want_sun2) {
        const uint32_t j = base + na2 + (uint32_t) __popcll(b2s & below);
        sq.origin_dist[j] = s_sun2_o;
        sq.dir_out[j] = make_float4(s_sun2_d.x, s_sun2_d.y, s_sun2_d.z, bitsf(kShadowKindSun2 * sq.capacity + i));
        sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
      }
    }
  }
}

// ---- ocean ----
// optix_raytrace_ocean (optix_kernel_raytrace.cu:134-144): after the surfaces and the particles, every path is marched against the height field up
// to its current hit; a closer water surface replaces the hit.
__global__ __launch_bounds__(kBlock) void k_trace_ocean(DeviceScene sc, PathQueue q, const uint32_t* ctrl) {
  const uint32_t n = ctrl[kCtlPaths];
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 o4 = q.origin_t[i], d4 = q.dir_slot[i];
    const float depth = ocean_intersection_distance(sc, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), o4.w);
    if (depth < o4.w) {
      q.origin_t[i].w = depth;
      uint4 hid = q.hit_id[i];
      hid.x = kHitOcean; hid.y = 0u;
      q.hit_id[i] = hid;
    }
  }
}

// ocean_process_tasks (ocean.cuh:12-102): the water surface is a smooth dielectric; BSDF-sampled light and the sun, no sampled light and no
// ambient sample. A refracted path enters or leaves the water: its medium and volume stacks change. Records go where a surface vertex puts them.
__global__ __launch_bounds__(kBlock, LUM_FEATURE_WAVES) void k_ocean_shade(DeviceScene sc, PathQueue in, PathQueue out, NeeQueue nee, ShadowQueue sq, uint32_t* ctrl, uint32_t depth_const) {
  const uint32_t n = ctrl[kCtlPaths];
  uint32_t* count_out = ctrl + kCtlStride + kCtlPaths;
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  const bool sun_allowed = sc.sky_mode != kSkyConstantColor && sc.sky_lut_transmittance != nullptr && sc.sky_lut_multiscattering != nullptr;
  // Water-surface hits are a sparse subset of the queue: a wave collects their indices in LDS and shades them 64 at a time (as k_shade does with surface hits)
  __shared__ uint32_t pending_hits[kBlock / 64][128];
  uint32_t* pending = pending_hits[threadIdx.x >> 6];
  uint32_t num_pending = 0;  // wave-uniform
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  for (uint32_t round = 0;; round++) {
    const bool input_done = round >= rounds;
    if (!input_done) {
      const uint32_t idx = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
      const bool is_hit = idx < n && in.hit_id[idx].x == kHitOcean;
      const unsigned long long bh = __ballot(is_hit);
      if (is_hit) pending[num_pending + (uint32_t) __popcll(bh & below)] = idx;
      num_pending += (uint32_t) __popcll(bh);
    }
    if (num_pending < (LUM_HIT_COMPACT ? 64u : 1u) && !(input_done && num_pending > 0u)) {
      if (input_done) break;
      continue;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t take = min(num_pending, 64u);
    num_pending -= take;
    const bool valid = lane < take;
    const uint32_t i = valid ? pending[num_pending + lane] : 0u;
    __builtin_amdgcn_wave_barrier();
    bool survive = false, want_sun = false, want_lq = false;
    float4 n_o, n_d; uint4 n_aux, n_hid;
    float4 s_origin, s_sun_dir;
    if (valid) {
      const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
      const uint4 aux = in.aux[i], hid = in.hit_id[i];
      const uint32_t state = aux.w;
      const V3 ray = v3(d4.x, d4.y, d4.z);
      const V3 position = v3(o4.x, o4.y, o4.z) + ray * o4.w;
      const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
      const GeoContext g = ocean_context(sc, position, ray, state, aux.z);
      const uint32_t top_volume = volume_stack_peek(hid.w, false);
      const LocalFrame lf = local_frame(sc, g);
      s_origin = make_float4(position.x, position.y, position.z, 0.0f);
      float4 bs_rp = make_float4(0.0f, 0.0f, 1.0f, 0.0f), bs_ws = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (lights_present && (state & kStVolumeScattered) == 0) {
        const LightDirSample lb = sample_light_direction(lf, g, smp);
        bs_rp = make_float4(lb.ray.x, lb.ray.y, lb.ray.z, lb.probability);
        if (lb.probability != 0.0f) { want_lq = true; bs_ws = make_float4(lb.weight.r, lb.weight.g, lb.weight.b, 0.0f); }
      }
      uint4 sun = make_uint4(0u, 0u, 0u, 0u);
      if (sun_allowed) {
        Col sun_light; V3 sun_dir;
        if (sample_sun(sc, sky_view(sc), lf, g, smp, sun_light, sun_dir)) {
          sun_light = sun_light * volume_transmittance(sc, top_volume, g.position, sun_dir, kFltMax);
          const U2 c = record_pack(sun_light), r = ray_pack(sun_dir);
          sun = make_uint4(c.x, c.y, r.x, r.y);
          if (c.x != 0 || c.y != 0) {
            want_sun = true;
            const V3 ar = ray_unpack(r);
            s_sun_dir = make_float4(ar.x, ar.y, ar.z, kFltMax);
            nee.sun_water[i] = make_float4(1.0f, 0.0f, 0.0f, bitsf(0u));  // the surface's own sun sample is never a caustics path
          }
        }
      }
      const BounceSample bounce = sample_bounce(lf, g, smp, 0);
      nee.geo_color_light[i] = make_float4(0.0f, 0.0f, 0.0f, bitsf(kLightIdInvalid));
      nee.bsdf_ray_prob[i] = bs_rp; nee.bsdf_weight_sum[i] = bs_ws;
      nee.ambient[i] = make_uint4(0u, 0u, 0u, 0u);
      if (sc.sky_mode != kSkyConstantColor) nee.sun[i] = sun;
      Col record = record_unpack(U2{aux.x, aux.y}) * bounce.weight;
      const float shift_length = 8.0f * kEps * (1.0f + sc.ocean_amplitude) * (1.0f + fabsf(sc.ocean_height));  // ocean_shift_vector, ocean_utils.cuh:519-523
      const V3 bounce_pos = g.position + g.normal * (bounce.transparent_pass ? -shift_length : shift_length);
      const uint32_t new_state = state & ~(kStCameraDirection | kStAllowEmission | kStUseIgnoreHandle);
      survive = true;
      if ((state & kStDeltaPath) == 0) {  // directives.cuh:11-32
        const float value = importance(record);
        if (value < sc.cam_rr_threshold) {
          const float p = (value > 0.0f) ? fmaxf(value / sc.cam_rr_threshold, 1.0f / 8.0f) : 0.0f;
          if (smp.next1(kRndRussianRoulette) > p) survive = false;
          else record = record * (1.0f / p);
        }
      }
      if (survive) {
        uint32_t medium = aux.z, volumes = hid.w;
        if (bounce.transparent_pass) {
          const bool inside = (g.params.flags & kMatRefractionInside) != 0;
          medium = medium_ior_modify(medium, sc.ocean_refractive_index, !inside);
          volumes = volume_stack_modify(volumes, kVolumeOcean, !inside);
        }
        const U2 rp = record_pack(record);
        n_o = make_float4(bounce_pos.x, bounce_pos.y, bounce_pos.z, kFltMax);
        n_d = make_float4(bounce.ray.x, bounce.ray.y, bounce.ray.z, d4.w);
        n_aux = make_uint4(rp.x, rp.y, medium, new_state);
        n_hid = make_uint4(0u, 0u, hid.z, volumes);
      }
    }
    const unsigned long long ballot = __ballot(survive);
    if (ballot) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(count_out, (uint32_t) __popcll(ballot));
      base = __builtin_amdgcn_readfirstlane(base);
      if (survive) {
        const uint32_t j = base + (uint32_t) __popcll(ballot & below);
        out.origin_t[j] = n_o; out.dir_slot[j] = n_d; out.aux[j] = n_aux; out.hit_id[j] = n_hid;
      }
    }
    const unsigned long long bn = __ballot(want_sun);
    if (bn) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(ctrl + kCtlShadowItems, (uint32_t) __popcll(bn));
      base = __builtin_amdgcn_readfirstlane(base);
      if (want_sun) {
        const uint32_t j = base + (uint32_t) __popcll(bn & below);
        sq.origin_dist[j] = make_float4(s_origin.x, s_origin.y, s_origin.z, s_sun_dir.w);
        sq.dir_out[j] = make_float4(s_sun_dir.x, s_sun_dir.y, s_sun_dir.z, bitsf(3u * sq.capacity + i));
        sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, kHitOcean, 0u);
      }
    }
    const unsigned long long bl = __ballot(want_lq);
    if (bl) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(ctrl + kCtlLightItems, (uint32_t) __popcll(bl));
      base = __builtin_amdgcn_readfirstlane(base);
      if (want_lq) sq.light_items[base + (uint32_t) __popcll(bl & below)] = i;
    }
  }
}

// ---- clouds ----
// cloud_process_tasks (cloud.cuh:340-384; device_renderer.c:78-82): after the volume events, every path is marched through the cloud layers up to its
// hit. The scattered light goes to the path's result, its throughput takes the layers' (and, with atmosphere_scattering, the air's) transmittance,
// and its origin moves up to the last layer entered so that the aerial-perspective pass and the sky see the rest of the ray.
// Three kernels. The marches differ wildly in length - a ray meets nothing, or up to ~100 in-cloud steps each with a sky march and two cloud-shadow
// marches of its own, and ends early below 10 % transmittance - so one lane per path leaves most lanes of a wave waiting for its longest ray (measured:
// VALU lane utilisation 0.27). k_clouds_list therefore lists the (path, layer) marches, k_clouds_march runs them with persistent lanes that take the
// next march as soon as theirs is over, and k_clouds composes the layers of a path in the order its ray enters them.
__global__ __launch_bounds__(kBlock) void k_clouds_list(DeviceScene sc, PathQueue in, CloudQueue cq, uint32_t* ctrl) {
  const uint32_t n = ctrl[kCtlPaths];
  const SkyView sky = sky_view(sc);
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  // one list reservation per 8 rounds of a wave (three per round, one per layer, kept the counter's atomic unit busier than the kernel's own work)
  constexpr uint32_t kListRounds = 8;
  uint32_t masks = 0;  // 3 bits per round: the layers this lane's path of that round reaches
  for (uint32_t round = 0; round < rounds; round++) {
    const uint32_t i = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
    uint32_t mask = 0;
    if (i < n && in.hit_id[i].x != kHitInvalid) {  // kHitInvalid: ended by the sky fast path of the volume events (no task exists for it in the reference)
      const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
      const V3 sky_origin = world_to_sky(sky, v3(o4.x, o4.y, o4.z)), ray = v3(d4.x, d4.y, d4.z);
      const float limit = o4.w * 0.001f;
#pragma unroll
      for (int l = 0; l < 3; l++) if (cloud_layer_intersection(sc, sky_origin, ray, limit, l).x != kFltMax) mask |= 1u << l;
    }
    const uint32_t slot = round % kListRounds;
    masks |= mask << (3u * slot);
    if (slot + 1u == kListRounds || round + 1u == rounds) {
      uint32_t total = 0;
#pragma unroll
      for (uint32_t b = 0; b < 3u * kListRounds; b++) total += (uint32_t) __popcll(__ballot((masks >> b) & 1u));
      if (total) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(ctrl + kCtlCloudItems, total);
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t group_start = round - slot;
#pragma unroll
        for (uint32_t b = 0; b < 3u * kListRounds; b++) {
          const bool want = (masks >> b) & 1u;
          const unsigned long long bal = __ballot(want);
          if (want) cq.items[base + (uint32_t) __popcll(bal & below)] = (((group_start + b / 3u) * gridDim.x + blockIdx.x) * kBlock + threadIdx.x) | ((b % 3u) << 30);
          base += (uint32_t) __popcll(bal);
        }
      }
      masks = 0;
    }
  }
}

__global__ __launch_bounds__(kBlock, LUM_CLOUD_WAVES) void k_clouds_march(DeviceScene sc, PathQueue in, CloudQueue cq, uint32_t* ctrl, uint32_t depth_const) {
  const uint32_t n = ctrl[kCtlCloudItems];
  uint32_t* cursor = ctrl + kCtlCloudCursor;
  const SkyView sky = sky_view(sc);
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  CloudMarch m;
  uint32_t slot = 0;      // where this lane's result goes
  bool active = false;
  uint32_t chunk_next = 0, chunk_end = 0;
  bool more = true;       // wave-uniform: the list may still hold marches
  for (;;) {
    // hand idle lanes the next marches: the wave reserves 64 list entries at a time
    unsigned long long idle = __ballot(!active);
    while (idle != 0ull && more) {
      if (chunk_next >= chunk_end) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cursor, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        chunk_next = base;
        chunk_end = min(base + 64u, n);
        more = base < n;
        if (!more) break;
      }
      const uint32_t avail = chunk_end - chunk_next, want = (uint32_t) __popcll(idle);
      const uint32_t rank = (uint32_t) __popcll(idle & below);
      if (!active && rank < avail) {
        const uint32_t item = cq.items[chunk_next + rank];
        const uint32_t i = item & 0x3FFFFFFFu;
        const int l = (int) (item >> 30);
        const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
        const uint4 hid = in.hit_id[i];
        const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
        const V3 sky_origin = world_to_sky(sky, v3(o4.x, o4.y, o4.z)), ray = v3(d4.x, d4.y, d4.z);
        const F2 isect = cloud_layer_intersection(sc, sky_origin, ray, o4.w * 0.001f, l);
        slot = (uint32_t) l * cq.capacity + i;
        active = m.begin(sc, sky, smp, sky_origin, ray, isect.x, isect.y, l);
        if (!active) { const CloudResult r = m.result(); cq.result[slot] = make_float4(r.scattered_light.r, r.scattered_light.g, r.scattered_light.b, r.transmittance); cq.hit_dist[slot] = r.hit_dist; }
      }
      chunk_next += min(want, avail);
      idle = __ballot(!active);
    }
    if (__ballot(active) == 0ull) break;
    V3 pos = m.origin;
    float density = 0.0f;
    bool over = false;
    if (active) over = !m.find(sc, pos, density);
    if (active && !over) over = !m.light(sc, sky, pos, density);
    if (active && over) {
      const CloudResult r = m.result();
      cq.result[slot] = make_float4(r.scattered_light.r, r.scattered_light.g, r.scattered_light.b, r.transmittance);
      cq.hit_dist[slot] = r.hit_dist;
      active = false;
    }
  }
}

__global__ __launch_bounds__(kBlock, LUM_CLOUD_WAVES) void k_clouds(DeviceScene sc, PathQueue in, CloudQueue cq, float4* results, const uint32_t* ctrl, uint32_t depth_const) {
  const uint32_t n = ctrl[kCtlPaths];
  const SkyView sky = sky_view(sc);
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  // Only the paths whose ray reached a layer have something to compose (with atmosphere_scattering: an aerial-perspective march per layer entered);
  // a wave collects them in LDS and handles them 64 at a time.
  __shared__ uint32_t pending_marches[kBlock / 64][128];
  uint32_t* pending = pending_marches[threadIdx.x >> 6];
  uint32_t num_pending = 0;  // wave-uniform
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  for (uint32_t round = 0;; round++) {
    const bool input_done = round >= rounds;
    if (!input_done) {
      const uint32_t i = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
      bool march = false;
      if (i < n && in.hit_id[i].x != kHitInvalid) {
        const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
        const V3 sky_origin = world_to_sky(sky, v3(o4.x, o4.y, o4.z)), ray = v3(d4.x, d4.y, d4.z);
        const float limit = o4.w * 0.001f;
#pragma unroll
        for (int l = 0; l < 3; l++) march |= cloud_layer_intersection(sc, sky_origin, ray, limit, l).x != kFltMax;
        // (a path that reaches no layer leaves clouds_render with nothing added, its throughput and origin unchanged: record_pack(record_unpack(x)) == x)
      }
      const unsigned long long bm = __ballot(march);
      if (march) pending[num_pending + (uint32_t) __popcll(bm & below)] = i;
      num_pending += (uint32_t) __popcll(bm);
    }
    if (num_pending < 64u && !(input_done && num_pending > 0u)) {
      if (input_done) break;
      continue;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t take = min(num_pending, 64u);
    num_pending -= take;
    const bool valid = lane < take;
    const uint32_t i = valid ? pending[num_pending + lane] : 0u;
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      const uint4 hid = in.hit_id[i];
      float4 o4 = in.origin_t[i];
      const float4 d4 = in.dir_slot[i];
      uint4 aux = in.aux[i];
      const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
      const V3 origin = v3(o4.x, o4.y, o4.z), ray = v3(d4.x, d4.y, d4.z);
      Col record = record_unpack(U2{aux.x, aux.y});
      Col color = splat(0.0f);
      float cloud_transmittance = 1.0f;
      const float cloud_offset = clouds_render_with(sc, sky, smp, world_to_sky(sky, origin), ray, o4.w * 0.001f, color, record, cloud_transmittance,
                                                    [&](int l, float start, float dist) {
#if LUM_CLOUD_PERSISTENT
                                                      (void) dist;
                                                      if (start == kFltMax) return CloudResult{splat(0.0f), 1.0f, start};  // not reached: clouds_compute's empty result
                                                      const float4 r = cq.result[(uint32_t) l * cq.capacity + i];
                                                      return CloudResult{col(r.x, r.y, r.z), r.w, cq.hit_dist[(uint32_t) l * cq.capacity + i]};
#else
                                                      return clouds_compute(sc, sky, smp, world_to_sky(sky, origin), ray, start, dist, l);
#endif
                                                    });
      if (sc.cloud_atmosphere_scattering && cloud_offset != kFltMax && cloud_offset > 0.0f) {
        const float cloud_world_offset = cloud_offset * 1000.0f;
        const V3 moved = origin + ray * cloud_world_offset;
        o4.x = moved.x; o4.y = moved.y; o4.z = moved.z;
        if (o4.w != kFltMax) o4.w -= cloud_world_offset;
        in.origin_t[i] = o4;
      }
      const U2 rp = record_pack(record);
      aux.x = rp.x; aux.y = rp.y;
      in.aux[i] = aux;
      add_to_result(results, fbits(d4.w), color);
    }
  }
}

// ---- HDRI bake (cuda/sky_hdri.cuh:13-160, device/device_sky.c:283-316): the sky without celestial bodies - and with the clouds, when active - seen from
// `origin`, as an equirectangular dim x dim image. 32 lanes per texel share its samples; their means go through the reference's trimmed mean. ----
#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// The material word of the traversal triangles (dev_scene.h): the albedo texture's id, or - untextured - whether a visibility ray cannot pass
// (kBvhTriOpaque: the decision of optix_anyhit.cuh:49-139 for alpha 1), taken once per triangle with the kernels' own material decoding; run at
// scene upload and again after a material edit.
__global__ __launch_bounds__(kBlock) void k_tri_opacity(DeviceScene sc, BvhTri* tris, uint32_t count) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  const uint32_t material = sc.tri_tex[tris[i].scene_index].w & 0xFFFFu;
  uint32_t word = kBvhTriNoTexture;
  if (material < sc.num_materials) {
    const Material m = load_material(sc, material);
    word = (m.albedo_tex != kTextureNone) ? m.albedo_tex : ((m.alpha == 1.0f) ? kBvhTriOpaque : kBvhTriNoTexture);  // textured: the texel decides
  }
  tris[i].albedo_tex = word;
}

// The emissive triangles in world space, one record per light id (load_tri_light_table, dev_light.h): light_triangle_init's result
// (light_triangle.cuh:37-72) evaluated once per light at scene upload instead of once per candidate and vertex.
__global__ __launch_bounds__(kBlock) void k_light_table(DeviceScene sc, float4* table) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= sc.num_lights) return;
  const uint2 handle = sc.light_tri_handles[i];
  const TriLight t = load_tri_light(sc, handle.x, handle.y);
  const Material m = load_material(sc, t.material_id);
  const bool textured = m.luminance_tex != kTextureNone || m.albedo_tex != kTextureNone;
  const Col color = textured ? splat(0.0f) : tri_light_color(sc, t, F2{0.0f, 0.0f});  // without textures the colour does not depend on the point
  table[4u * i] = make_float4(t.vertex.x, t.vertex.y, t.vertex.z, bitsf(t.material_id | (t.bidirectional ? 0x10000u : 0u)));
  table[4u * i + 1u] = make_float4(t.edge1.x, t.edge1.y, t.edge1.z, bitsf(t.scene_tri));
  table[4u * i + 2u] = make_float4(t.edge2.x, t.edge2.y, t.edge2.z, tri_light_area(t));
  table[4u * i + 3u] = make_float4(color.r, color.g, color.b, bitsf(textured ? 1u : 0u));
}
#endif

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
__global__ __launch_bounds__(256) void k_sky_hdri(DeviceScene sc, float ox, float oy, float oz, uint32_t dim, uint32_t sample_count, float4* __restrict__ dst) {
  __shared__ float values[256];
  const uint32_t pixel = (blockIdx.x * 256u + threadIdx.x) >> 5, lane = threadIdx.x & 31u;
  const bool in_range = pixel < dim * dim;
  const uint32_t y = in_range ? pixel / dim : 0u, x = in_range ? pixel - y * dim : 0u;
  const SkyView sky = sky_view(sc);
  const float step_size = 1.0f / (float) (dim - 1u);
  Col color = splat(0.0f);
  float alpha = 0.0f;
  uint32_t num_samples = 0;
  const bool clouds = sc.cloud_active && sc.cloud_noise_shape != nullptr;
  if (in_range) {
    for (uint32_t sample_id = lane; sample_id < sample_count; sample_id += 32u) {
      const Sampler smp{sc.bluenoise_2d, x, y, sample_id, 0};
      const F2 jitter = smp.next2(kRndCameraJitter);
      const float u = ((float) x + jitter.x) * step_size, v = 1.0f - ((float) y + jitter.y) * step_size;
      const float altitude = kPi * v - 0.5f * kPi, azimuth = 2.0f * kPi * u - kPi;
      const V3 ray = angles_to_direction(altitude, azimuth);
      Col sky_color = splat(0.0f), transmittance = splat(1.0f);
      float cloud_transmittance = 1.0f;
      V3 sky_origin = world_to_sky(sky, v3(ox, oy, oz));
      if (clouds) {  // sky_hdri.cuh:88-92: the clouds in front, the sky behind them dimmed by their transmittance
        const float offset = clouds_render(sc, sky, smp, sky_origin, ray, kFltMax, sky_color, transmittance, cloud_transmittance);
        sky_origin = sky_origin + ray * offset;
      }
      const Col behind = sky_get_color(sc, sky, sky_origin, ray, kFltMax, false, (int) sky.steps, smp.next1(kRndSkyStepOffset));
      sky_color = sky_color + behind * transmittance;
      color = color + sky_color;
      alpha += cloud_transmittance;
      num_samples++;
    }
  }
  const uint32_t buckets = min(32u, sample_count);
  float* group = values + (threadIdx.x & ~31u);
  float out[4];
  const float mean[4] = {num_samples ? color.r / (float) num_samples : 0.0f, num_samples ? color.g / (float) num_samples : 0.0f, num_samples ? color.b / (float) num_samples : 0.0f,
                         num_samples ? alpha / (float) num_samples : 0.0f};
#pragma unroll
  for (int ch = 0; ch < 4; ch++) {
    __syncthreads();
    values[threadIdx.x] = mean[ch];
    __syncthreads();
    out[ch] = (lane == 0u && in_range) ? sky_hdri_median_of_means(group, buckets) : 0.0f;
  }
  if (lane == 0u && in_range) dst[x + y * dim] = make_float4(out[0], out[1], out[2], out[3]);  // .w: the clouds' own transmittance (the reference's separate shadow texture), 1 without clouds
}
#endif

// ---- fog (cuda/volume.cuh; queue order device/device_renderer.c:64-76, :114-118) ----
// volume_process_inscattering (volume.cuh:31-98): what the fog scatters into the ray between its origin and its end point (the hit, or infinity
// for a ray that left the scene): a bridge to a sampled emissive triangle on delta paths, the sun and the ambient sample from a vertex on the
// ray. The visibility rays go through the ShadowQueue (17 kinds per path), k_volume_resolve sums up (optix_kernel_shadow_volume.cu:13-98).
__global__ __launch_bounds__(kBlock, LUM_VOLUME_WAVES) void k_volume_inscatter(DeviceScene sc, PathQueue in, VolumeQueue vq, ShadowQueue sq, uint32_t* ctrl, uint32_t depth_const) {
  const uint32_t n = ctrl[kCtlPaths];
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  const bool sun_allowed = sc.sky_mode != kSkyConstantColor && sc.sky_lut_transmittance != nullptr && sc.sky_lut_multiscattering != nullptr;
#ifndef LUM_INSCATTER_COMPACT
#define LUM_INSCATTER_COMPACT 1  // 0 (measurement only): every round works on what it finds, partial waves and all
#endif
  // With an ocean only the paths inside a volume have work here (a fog holds every path): their indices are collected per wave in LDS and handled 64 at a
  // time, the others get their empty records right away.
  __shared__ uint32_t pending_paths[kBlock / 64][128];
  uint32_t* pending = pending_paths[threadIdx.x >> 6];
  uint32_t num_pending = 0;  // wave-uniform
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  for (uint32_t round = 0;; round++) {
    const bool input_done = round >= rounds;
    if (!input_done) {
      const uint32_t idx = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
      bool in_volume = false;
      if (idx < n) {
        in_volume = volume_stack_peek(in.hit_id[idx].w, false) != kVolumeNone;
        if (!in_volume) {
          vq.bridge[idx] = make_float4(0.0f, 0.0f, 0.0f, bitsf(0u));
          vq.sky[idx] = make_uint4(0u, 0u, 0u, 0u);
          vq.weight[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
      }
      const unsigned long long bv = __ballot(in_volume);
      if (in_volume) pending[num_pending + (uint32_t) __popcll(bv & below)] = idx;
      num_pending += (uint32_t) __popcll(bv);
    }
    if (num_pending < (LUM_INSCATTER_COMPACT ? 64u : 1u) && !(input_done && num_pending > 0u)) {
      if (input_done) break;
      continue;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t take = min(num_pending, 64u);
    num_pending -= take;
    const bool valid = lane < take;
    const uint32_t i = valid ? pending[num_pending + lane] : 0u;
    __builtin_amdgcn_wave_barrier();
    uint32_t segments = 0;
    BridgeWalk walk;
    bool want_sun = false, want_amb = false, want_sun2 = false, want_amb2 = false;
    float4 sky_origin = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    V3 sun_dir = v3(0.0f, 0.0f, 1.0f), amb_dir = v3(0.0f, 0.0f, 1.0f);
    float sun_limit = kFltMax, amb_limit = kFltMax;
    V3 sun2_o = v3(0.0f, 0.0f, 0.0f), sun2_d = v3(0.0f, 0.0f, 1.0f), amb2_o = v3(0.0f, 0.0f, 0.0f), amb2_d = v3(0.0f, 0.0f, 1.0f);
    Sampler smp{sc.bluenoise_2d, 0u, 0u, 0u, depth_const};
    if (valid) {
      const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
      const uint4 aux = in.aux[i], hid = in.hit_id[i];
      const V3 origin = v3(o4.x, o4.y, o4.z), ray = v3(d4.x, d4.y, d4.z);
      const uint32_t state = aux.w;
      smp.px = hid.z & 0xFFFFu; smp.py = hid.z >> 16; smp.sample_id = path_sample_id(hid.w);
      const uint32_t top_volume = volume_stack_peek(hid.w, false), second_volume = volume_stack_peek(hid.w, true);
      VolContext ctx = volume_context(sc, top_volume, origin, ray, state, o4.w);
      float4 bridge = make_float4(0.0f, 0.0f, 0.0f, bitsf(0u));
      const bool bridges_allowed = top_volume != kVolumeNone && lights_present && (state & kStDeltaPath) != 0 && (state & kStVolumeScattered) == 0 &&
                                   (top_volume != kVolumeOcean || sc.ocean_triangle_light_contribution);  // direct_lighting.cuh:296-306
      if (bridges_allowed) {
        const BridgeSample bs = volume_light_sample(sc, ctx, smp);
        if (bs.light_id != kLightIdInvalid && bs.seed != 0xFFFFFFFFu) {
          walk = bridge_walk_begin(sc, ctx, bs, smp);
          segments = walk.vertex_count;
          bridge = make_float4(bs.color.r, bs.color.g, bs.color.b, bitsf(segments));
        }
      }
      Col w = splat(0.0f);
      uint4 sky_words = make_uint4(0u, 0u, 0u, 0u);
      if (top_volume != kVolumeNone) {
        w = volume_sky_initial_vertex(ctx, smp);  // the vertex the sun and the ambient sample start from
        sky_origin = make_float4(ctx.position.x, ctx.position.y, ctx.position.z, kFltMax);
        if (sun_allowed) {
          Col lc; V3 dir;
          const bool have = (top_volume == kVolumeOcean) ? sun_caustic_sample(sc, sky_view(sc), ctx, smp, 1u, top_volume, second_volume, lc, dir)  // direct_lighting.cuh:370-380
                                                         : volume_sun_sample(sc, sky_view(sc), ctx, smp, lc, dir);
          if (have) {
            const U2 c = record_pack(lc), r = ray_pack(dir);
            sky_words.x = c.x; sky_words.y = c.y;
            if (c.x != 0 || c.y != 0) {
              want_sun = true;
              sun_dir = ray_unpack(r);
              const SkyRayPlan plan = plan_sun_ray(sc, ctx.position, 0xFFFFFFFFu, top_volume, true, sun_dir);
              sun_limit = plan.limit;
              vq.sun_water[i] = make_float4(plan.fresnel_factor, 0.0f, 0.0f, bitsf(sky_ray_flags(plan)));
              if (plan.second && !plan.total_reflection) { want_sun2 = true; sun2_o = plan.second_origin; sun2_d = plan.second_dir; }
            }
          }
        }
        const V3 bounce = volume_bsdf_sample(sc, ctx, smp, kRndVolAmbientResampling, kRndVolAmbientDiffuse);
        if (sc.sky_mode != kSkyDefault) {  // direct_lighting.cuh:385-403, :521-584
          const U2 c = record_pack(sky_color_no_compute(sc, ctx.position, bounce, 0u) * splat(1.0f)), r = ray_pack(bounce);
          if (c.x != 0 || c.y != 0) {
            amb_dir = ray_unpack(r);
            const SkyRayPlan plan = plan_ambient_ray(sc, ctx.position, 0xFFFFFFFFu, top_volume, second_volume, true, amb_dir);
            if (plan.valid) {
              sky_words.z = c.x; sky_words.w = c.y;
              want_amb = true;
              amb_limit = plan.limit;
              vq.amb_t1[i] = make_float4(plan.t1.r, plan.t1.g, plan.t1.b, plan.fresnel_factor);
              vq.amb_t2[i] = make_float4(plan.t2.r, plan.t2.g, plan.t2.b, bitsf(sky_ray_flags(plan)));
              if (plan.second && !plan.total_reflection) { want_amb2 = true; amb2_o = plan.second_origin; amb2_d = plan.second_dir; }
            }
          }
        }
      }
      vq.bridge[i] = bridge;
      vq.sky[i] = sky_words;
      vq.weight[i] = make_float4(w.r, w.g, w.b, 0.0f);
    }
    // Visibility rays: ONE reservation per batch for everything its lanes ask for - a lane's bridge segments (1..15), sun, ambient and their second
    // segments - and every lane writes its own run. (One atomic per kind and bridge segment, as at first, made the camera rays' pass - every lane a
    // bridge - wait for up to 17 same-address atomics per batch: they, not the arithmetic, were most of its time; profiles/r02_ab_experiments.txt.)
    {
      const uint32_t mine = segments + (want_sun ? 1u : 0u) + (want_amb ? 1u : 0u) + (want_sun2 ? 1u : 0u) + (want_amb2 ? 1u : 0u);
      uint32_t prefix = mine;  // inclusive scan over the wave
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(prefix, off);
        if ((int) lane >= off) prefix += v;
      }
      const uint32_t total = __shfl(prefix, 63);
      if (total) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(ctrl + kCtlVolumeShadowItems, total);
        base = __builtin_amdgcn_readfirstlane(base);
        uint32_t j = base + prefix - mine;
        for (uint32_t k = 0; k < segments; k++) {
          sq.origin_dist[j] = make_float4(walk.vertex.x, walk.vertex.y, walk.vertex.z, walk.dist);
          sq.dir_out[j] = make_float4(walk.dir.x, walk.dir.y, walk.dir.z, bitsf(k * sq.capacity + i));
          sq.ids[j] = make_uint4(walk.light.x, walk.light.y, 0xFFFFFFFFu, 0u);  // the segment that reaches the light leaves that light out
          j++;
          if (k + 1 < segments) bridge_walk_next(walk, smp, k + 1);
        }
        if (want_sun) {
          sq.origin_dist[j] = make_float4(sky_origin.x, sky_origin.y, sky_origin.z, sun_limit);
          sq.dir_out[j] = make_float4(sun_dir.x, sun_dir.y, sun_dir.z, bitsf(kVolumeKindSun * sq.capacity + i));
          sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
          j++;
        }
        if (want_amb) {
          sq.origin_dist[j] = make_float4(sky_origin.x, sky_origin.y, sky_origin.z, amb_limit);
          sq.dir_out[j] = make_float4(amb_dir.x, amb_dir.y, amb_dir.z, bitsf(kVolumeKindAmbient * sq.capacity + i));
          sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
          j++;
        }
        if (want_sun2) {  // second segments beyond the water surface
          sq.origin_dist[j] = make_float4(sun2_o.x, sun2_o.y, sun2_o.z, kFltMax);
          sq.dir_out[j] = make_float4(sun2_d.x, sun2_d.y, sun2_d.z, bitsf(kVolumeKindSun2 * sq.capacity + i));
          sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
          j++;
        }
        if (want_amb2) {
          sq.origin_dist[j] = make_float4(amb2_o.x, amb2_o.y, amb2_o.z, kFltMax);
          sq.dir_out[j] = make_float4(amb2_d.x, amb2_d.y, amb2_d.z, bitsf(kVolumeKindAmbient2 * sq.capacity + i));
          sq.ids[j] = make_uint4(0xFFFFFFFFu, 0u, 0xFFFFFFFFu, 0u);
        }
      }
    }
  }
}

// optix_kernel_shadow_volume.cu:38-97: bridge colour x the visibilities of its segments, + (sun + ambient) x the weight of their vertex, x throughput
__global__ __launch_bounds__(kBlock) void k_volume_resolve(DeviceScene sc, PathQueue in, VolumeQueue vq, ShadowQueue sq, float4* results, const uint32_t* ctrl) {
  const uint32_t n = ctrl[kCtlPaths];
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 bridge = vq.bridge[i], wt = vq.weight[i];
    const uint4 sky = vq.sky[i];
    const uint32_t segments = fbits(bridge.w);
    Col acc = splat(0.0f);
    if (segments) {
      const float4 v0 = sq.vis[i];
      Col shadow = col(v0.x, v0.y, v0.z);
      for (uint32_t k = 1; k < segments; k++) { const float4 v = sq.vis[k * sq.capacity + i]; shadow = shadow * col(v.x, v.y, v.z); }
      acc = acc + col(bridge.x, bridge.y, bridge.z) * shadow;
    }
    const Col w = col(wt.x, wt.y, wt.z);
    if (sky.x != 0 || sky.y != 0) {
      const float4 v = sq.vis[kVolumeKindSun * sq.capacity + i], sw = vq.sun_water[i];
      Col vis2 = splat(1.0f);
      if ((fbits(sw.w) & kSkyRaySecond) && !(fbits(sw.w) & kSkyRayTotalReflection)) { const float4 v2 = sq.vis[kVolumeKindSun2 * sq.capacity + i]; vis2 = col(v2.x, v2.y, v2.z); }
      acc = acc + combine_sun_ray(record_unpack(U2{sky.x, sky.y}), col(v.x, v.y, v.z), sw.x, fbits(sw.w), vis2) * w;
    }
    if (sky.z != 0 || sky.w != 0) {
      const float4 v = sq.vis[kVolumeKindAmbient * sq.capacity + i], t1 = vq.amb_t1[i], t2 = vq.amb_t2[i];
      Col vis2 = splat(1.0f);
      if ((fbits(t2.w) & kSkyRaySecond) && !(fbits(t2.w) & kSkyRayTotalReflection)) { const float4 v2 = sq.vis[kVolumeKindAmbient2 * sq.capacity + i]; vis2 = col(v2.x, v2.y, v2.z); }
      acc = acc + combine_ambient_ray(record_unpack(U2{sky.z, sky.w}), col(v.x, v.y, v.z), col(t1.x, t1.y, t1.z), t1.w, col(t2.x, t2.y, t2.z), fbits(t2.w), vis2) * w;
    }
    const uint4 aux = in.aux[i];
    add_to_result(results, fbits(in.dir_slot[i].w), acc * record_unpack(U2{aux.x, aux.y}));
  }
}

// volume_process_events (volume.cuh:100-229): the distance to the next scattering event is sampled in closed form; a path that scatters before
// its hit becomes a volume hit (listed for k_volume_bounce), every path's throughput takes the transmittance over the sampling density. In the
// non-procedural sky modes a ray that left the scene adds the sky here and ends ("sky fast path").
__global__ __launch_bounds__(kBlock) void k_volume_events(DeviceScene sc, PathQueue in, VolumeQueue vq, float4* results, uint32_t* ctrl, uint32_t depth_const) {
  const uint32_t n = ctrl[kCtlPaths];
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  // The scattering events are listed for k_volume_bounce with one reservation per 8 rounds of a wave (one per round saturated the counter's atomic unit)
  constexpr uint32_t kListRounds = 8;
  uint32_t scattered_mask = 0;  // bit r: this lane's path of round (group start + r) scattered
  for (uint32_t round = 0; round < rounds; round++) {
    const uint32_t i = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
    bool scattered = false;
    if (i < n && volume_stack_peek(in.hit_id[i].w, false) != kVolumeNone) {
      float4 o4 = in.origin_t[i];
      const float4 d4 = in.dir_slot[i];
      uint4 aux = in.aux[i], hid = in.hit_id[i];
      const V3 origin = v3(o4.x, o4.y, o4.z), ray = v3(d4.x, d4.y, d4.z);
      const uint32_t state = aux.w;
      const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
      const uint32_t volume_type = volume_stack_peek(hid.w, false);
      const Volume vol = volume_descriptor(sc, volume_type);
      VolumePath path = volume_compute_path(sc, vol, origin, ray, o4.w, true);
      Col record = record_unpack(U2{aux.x, aux.y});
      uint32_t hit_inst = hid.x, hit_tri = hid.y;
      const bool sky_fast_path = hit_inst == kHitSky && sc.sky_mode != kSkyDefault && (state & kStAllowAmbient) != 0;
      if (sky_fast_path) {
        Col sky = sky_color_no_compute(sc, origin, ray, state) * record;
        sky = sky * volume_transmittance_length(vol, path.length);
        add_to_result(results, fbits(d4.w), sky);
        hit_inst = kHitInvalid;
      }
      float intersection_probability = ((state & kStDeltaPath) && !particle_is_hit(hit_inst)) ? 0.5f : 1.0f;  // bounds the variance of highlights seen through the volume
      if (volume_type == kVolumeOcean && !sc.ocean_multiscattering && (state & kStDeltaPath) == 0) intersection_probability = 0.0f;  // single scattering in the water
      const F2 randoms = smp.next2(kRndVolumeIntersection);
      float pdf = 1.0f;
      if (randoms.y < intersection_probability) {
        const float volume_dist = volume_sample_intersection(vol, path.start, path.length, randoms.x);
        if (volume_dist < o4.w) {
          const float sample_pdf = volume_sample_intersection_pdf(vol, path.start, volume_dist);
          o4.w = volume_dist; hit_inst = kHitVolumeBase | volume_type; hit_tri = 0u;
          record = record * vol.scat;
          pdf *= intersection_probability;
          pdf *= sample_pdf;
          scattered = true;
          path.length = o4.w - path.start;
        }
      }
      if (!scattered && !sky_fast_path) pdf *= (1.0f - intersection_probability) + intersection_probability * volume_miss_probability(vol, path.length);
      record = record * volume_transmittance_length(vol, path.length);
      record = record * (1.0f / pdf);
      const U2 rp = record_pack(record);
      aux.x = rp.x; aux.y = rp.y;
      hid.x = hit_inst; hid.y = hit_tri;
      in.origin_t[i] = o4; in.aux[i] = aux; in.hit_id[i] = hid;
    }
    const uint32_t slot = round % kListRounds;
    scattered_mask |= (scattered ? 1u : 0u) << slot;
    if (slot + 1u == kListRounds || round + 1u == rounds) {
      unsigned long long b[kListRounds];
      uint32_t total = 0;
#pragma unroll
      for (uint32_t r = 0; r < kListRounds; r++) { b[r] = __ballot((scattered_mask >> r) & 1u); total += (uint32_t) __popcll(b[r]); }
      if (total) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(ctrl + kCtlVolumeItems, total);
        base = __builtin_amdgcn_readfirstlane(base);
        const uint32_t group_start = round - slot;
#pragma unroll
        for (uint32_t r = 0; r < kListRounds; r++) {
          if ((scattered_mask >> r) & 1u) vq.items[base + (uint32_t) __popcll(b[r] & below)] = ((group_start + r) * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
          base += (uint32_t) __popcll(b[r]);
        }
      }
      scattered_mask = 0;
    }
  }
}

// volume_process_tasks (volume.cuh:231-288): the paths that scattered in the fog continue in a direction drawn from the phase function
__global__ __launch_bounds__(kBlock) void k_volume_bounce(DeviceScene sc, PathQueue in, PathQueue out, VolumeQueue vq, uint32_t* ctrl, uint32_t depth_const) {
  const uint32_t n = ctrl[kCtlVolumeItems];
  uint32_t* count_out = ctrl + kCtlStride + kCtlPaths;
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  // Every listed path continues, so a wave knows how many queue slots its next rounds need before it has done them: one reservation per 8 rounds
  // (one per round saturated the counter's atomic unit: this kernel and k_volume_events took 2 ms per launch for a few hundred MB of traffic).
  constexpr uint32_t kReserveRounds = 8;
  uint32_t reserved_base = 0;
  for (uint32_t round = 0; round < rounds; round++) {
    const uint32_t k = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
    const bool valid = k < n;
    if (round % kReserveRounds == 0) {
      uint32_t count = 0;  // this wave's paths in the rounds [round, round + kReserveRounds)
      const uint32_t first = (round * gridDim.x + blockIdx.x) * kBlock + (threadIdx.x & ~63u);
      for (uint32_t r = 0; r < kReserveRounds && round + r < rounds; r++) {
        const uint32_t lo = first + r * gridDim.x * kBlock;
        count += (lo < n) ? min(64u, n - lo) : 0u;
      }
      uint32_t base = 0;
      if (lane == 0 && count) base = atomicAdd(count_out, count);
      reserved_base = __builtin_amdgcn_readfirstlane(base);
    }
    float4 n_o, n_d; uint4 n_aux, n_hid;
    if (valid) {
      const uint32_t i = vq.items[k];
      const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
      const uint4 aux = in.aux[i], hid = in.hit_id[i];
      const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, path_sample_id(hid.w), depth_const};
      const V3 ray = v3(d4.x, d4.y, d4.z);
      const V3 origin = v3(o4.x, o4.y, o4.z) + ray * o4.w;
      const VolContext ctx = volume_context(sc, volume_stack_peek(hid.w, false), origin, ray, aux.w, 0.0f);
      const V3 bounce = volume_bsdf_sample(sc, ctx, smp, kRndVolGiResampling, kRndVolGiDiffuse);
      uint32_t state = aux.w & ~(kStDeltaPath | kStCameraDirection | kStAllowEmission | kStUseIgnoreHandle);
      if (sc.sky_mode != kSkyDefault) state &= ~kStAllowAmbient; else state |= kStAllowAmbient;
      state |= kStVolumeScattered;
      n_o = make_float4(origin.x, origin.y, origin.z, kFltMax);
      n_d = make_float4(bounce.x, bounce.y, bounce.z, d4.w);
      n_aux = make_uint4(aux.x, aux.y, aux.z, state);
      n_hid = make_uint4(0u, 0u, hid.z, hid.w);
    }
    const unsigned long long b = __ballot(valid);
    if (valid) {  // the wave's 64 list entries of a round are consecutive: the valid ones are its first lanes
      const uint32_t j = reserved_base + (uint32_t) __popcll(b & below);
      out.origin_t[j] = n_o; out.dir_slot[j] = n_d; out.aux[j] = n_aux; out.hit_id[j] = n_hid;
    }
    reserved_base += (uint32_t) __popcll(b);
  }
}

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// ---- accumulation (cuda/accumulation.cuh:63-84): samples of a pixel are added in sample order ----
__global__ __launch_bounds__(kBlock) void k_accumulate(const float4* results, uint32_t num_pixels, uint32_t batch, float* first_moment, float* second_moment) {
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < num_pixels; p += gridDim.x * kBlock) {
    float r = first_moment[p], g = first_moment[num_pixels + p], b = first_moment[2 * num_pixels + p];
    float s = second_moment ? second_moment[p] : 0.0f;
    for (uint32_t k = 0; k < batch; k++) {
      const float4 v = results[k * num_pixels + p];
      r += v.x; g += v.y; b += v.z;
      s += luminance(col(v.x * v.x, v.y * v.y, v.z * v.z));
    }
    first_moment[p] = r; first_moment[num_pixels + p] = g; first_moment[2 * num_pixels + p] = b;
    if (second_moment) second_moment[p] = s;
  }
}

// One sample of a subset of the frame's pixels (an iteration of the undersampling preview, kernels.cuh:47-95): result p belongs to frame
// pixel pixels[p]. accumulation_collect_results, accumulation.cuh:36-61, with one result per pixel.
__global__ __launch_bounds__(kBlock) void k_accumulate_scatter(const float4* results, const uint32_t* pixels, uint32_t count, uint32_t frame_pixels, float* first_moment,
                                                               float* second_moment) {
  for (uint32_t p = blockIdx.x * kBlock + threadIdx.x; p < count; p += gridDim.x * kBlock) {
    const uint32_t index = pixels[p];
    const float4 v = results[p];
    first_moment[index] += v.x; first_moment[frame_pixels + index] += v.y; first_moment[2 * frame_pixels + index] += v.z;
    if (second_moment) second_moment[index] += luminance(col(v.x * v.x, v.y * v.y, v.z * v.z));
  }
}

#endif

// ---- standalone closest-hit entry for traversal tests and the trace micro-benchmark ----
struct RaysQuery : ClosestState {
  const float* origins; const float* dirs; const uint32_t* ignore; uint32_t* out;
  LUM_DEV bool load(const DeviceScene&, uint32_t i, V3& o, V3& d, float& tmax) {
    const bool ign = ignore != nullptr && ignore[2 * i] != 0xFFFFFFFFu;
    begin(ign, ign ? ignore[2 * i] : 0u, ign ? ignore[2 * i + 1] : 0u);
    o = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]); d = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]); tmax = kFltMax;
    return true;
  }
  LUM_DEV void finish(const DeviceScene&, uint32_t i) {
    const Hit h = result();
    out[3 * i] = h.instance_id; out[3 * i + 1] = h.tri_id; out[3 * i + 2] = fbits(h.t);
  }
};

__global__ LUM_TRACE_BOUNDS void k_trace_rays(DeviceScene sc, uint32_t n, const float* origins, const float* dirs, const uint32_t* ignore,
                                                           uint32_t* out, uint32_t* cursor, uint64_t* counters, uint32_t lds_nodes) {
  RayStats st{0, 0, 0};
  uint32_t rays = 0;
  RaysQuery q;
  q.origins = origins; q.dirs = dirs; q.ignore = ignore; q.out = out;
  LUM_TRACE_ITEMS(sc, n, cursor, q, st, rays, lds_nodes);
  flush_stats(counters, st, rays, kCntTrace, kCntNodes, kCntTris, kCntNodesLds);
}

// The Sobol / Owen pairs of one pass (dev_sampler.h LUM_SOBOL_TABLE): one thread per (dimension, sample id).
__global__ __launch_bounds__(256) void k_sobol_table(uint2* __restrict__ table, uint32_t first_sample, uint32_t count, uint32_t stride, uint32_t dims) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t dim = t / stride, s = t - dim * stride;
  if (dim >= dims || s >= count) return;
  const U2 q = sobol_owen(first_sample + s, dim);
  table[t] = make_uint2(q.x, q.y);
}

#if !LUM_FAST  // flavour-neutral: compiled once, in the exact translation unit
// ---- camera ray of one pixel (first sample id), for pixel queries ----
__global__ void k_pixel_ray(DeviceScene sc, uint32_t x, uint32_t y, uint32_t sample_id, float* origin, float* dir) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const Sampler smp{sc.bluenoise_2d, x, y, sample_id, 0};
  V3 o, d;
  camera_ray(sc, smp, o, d);
  origin[0] = o.x; origin[1] = o.y; origin[2] = o.z;
  dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

// ---- BSDF energy LUTs (cuda/bsdf_lut.cuh:20-211): pixel (0,0), depth 0, sample id = iteration ----
LUM_DEV uint16_t quantise_energy(float sum) { return (uint16_t) (1 + (uint16_t) (ceilf(saturate(sum) * 0xFFFE))); }

__global__ void k_generate_lut(const uint32_t* bluenoise, int table, uint32_t count, const uint16_t* conductor, uint16_t* dst) {
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= count) return;
  uint32_t x, y, z = 0;
  if (table < 2) { y = id / 32; x = id - y * 32; }
  else { z = id / 1024; y = (id - z * 1024) / 32; x = id - y * 32 - z * 1024; }
  const float NdotV = fmaxf(32.0f * kEps, x * (1.0f / 31));
  const float roughness = y * (1.0f / 31);
  const V3 V = normalize(v3(0.0f, sqrtf(1.0f - NdotV * NdotV), NdotV));
  Sampler smp{bluenoise, 0, 0, 0, 0};
  float sum = 0.0f;
  if (table < 2) {
    const Col f0 = col(0.04f, 0.04f, 0.04f);
    for (uint32_t i = 0; i < 0x10000u; i++) {
      smp.sample_id = i;
      const V3 H = sample_vndf_bounded(V, roughness, smp.next2(kRndBsdfReflection));
      const V3 R = reflect(V, H);
      if (R.z > 0.0f) {
        float v = eval_microfacet_over_vndf(V, roughness, R.z, NdotV);
        if (table == 1) v = v * luminance(fresnel_schlick(f0, shadowed_f90(f0), fabsf(dot(H, V))));
        sum += v;
      }
    }
    sum /= 0x10000u;
    if (table == 1) sum /= conductor[id] * (1.0f / 0xFFFF);
  }
  else {
    const float ior_base = 1.0f + z * (1.0f / 31) * 2.0f;
    const float ior = (table == 2) ? 1.0f / ior_base : ior_base;
    for (uint32_t i = 0; i < 0x10000u; i++) {
      smp.sample_id = i;
      bool tot;
      V3 H = sample_vndf_bounded(V, roughness, smp.next2(kRndBsdfReflection));
      const V3 R = reflect(V, H);
      V3 T = refract(V, H, ior, tot);
      float fres = tot ? 1.0f : fresnel_dielectric(H, V, T, ior);
      if (R.z > 0.0f) sum += eval_microfacet_over_vndf(V, roughness, R.z, NdotV) * fres;
      H = sample_vndf_caps(V, roughness, smp.next2(kRndBsdfRefraction));
      T = refract(V, H, ior, tot);
      fres = tot ? ((table == 2) ? 1.0f : 0.0f) : fresnel_dielectric(H, V, T, ior);
      const float NdotR = -T.z;
      if (NdotR > 0.0f) sum += ggx_g2_over_g1(pow4(roughness), NdotR, NdotV) * (1.0f - fres);
    }
    sum /= 0x10000u;
  }
  dst[id] = quantise_energy(sum);
}

#endif

LUM_NS_END
