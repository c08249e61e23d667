// Wavefront path-tracing kernels for gfx950.
//
// One pass renders `batch` consecutive sample ids of `num_pixels` pixels: path slot = sample_in_batch * num_pixels + pixel.
// Live paths are kept compacted in a PathQueue (SoA of 16-byte words); every depth runs
//     trace (closest hit)  ->  shade (context, NEE sampling, bounce, emission, roulette; appends survivors to the other
//     queue with one wave-aggregated atomic)  ->  shadow (visibility of the NEE work, adds into the path's result slot)
// and the pass ends with an in-order accumulation into the frame moments, which reproduces the reference's
// one-sample-at-a-time sums bit for bit (cuda/accumulation.cuh:63-84).
// Reference schedule: device/device_renderer.c:53-134; per-kernel restatements cite their sources below.
#pragma once

#include "dev_trace.h"

namespace lum {

enum PathState : uint32_t {  // cuda/utils.cuh:114-121
  kStDeltaPath = 1, kStCameraDirection = 2, kStVolumeScattered = 4, kStAllowEmission = 8, kStAllowAmbient = 16, kStUseIgnoreHandle = 32
};
enum SkyMode : uint32_t { kSkyDefault = 0, kSkyHdri = 1, kSkyConstantColor = 2 };

constexpr int kBlock = 256;
#ifndef LUM_SHADE_WAVES
#define LUM_SHADE_WAVES 2  // minimum waves per SIMD the shade kernel is compiled for (register budget 512 / waves)
#endif

struct PassParams {
  const uint32_t* pixels;  // pixel index (x + y*width) per local pixel, or nullptr for identity
  uint32_t num_pixels, batch, first_sample;
};

LUM_DEV void flush_stats(uint64_t* counters, const RayStats& st, uint32_t rays, uint32_t ray_counter, uint32_t node_counter, uint32_t tri_counter) {
  // one atomic per wave and counter
  uint32_t n = st.nodes, t = st.tris, r = rays;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { n += __shfl_down(n, off); t += __shfl_down(t, off); r += __shfl_down(r, off); }
  if ((threadIdx.x & 63) == 0) {
    if (n) atomicAdd((unsigned long long*) &counters[node_counter], (unsigned long long) n);
    if (t) atomicAdd((unsigned long long*) &counters[tri_counter], (unsigned long long) t);
    if (r) atomicAdd((unsigned long long*) &counters[ray_counter], (unsigned long long) r);
  }
}

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
    q.aux[i]      = make_uint4(rec.x, rec.y, medium_ior_modify(0u, 1.0f, true), kStDeltaPath | kStCameraDirection | kStAllowEmission | kStAllowAmbient);
    q.hit_id[i]   = make_uint4(0u, 0u, x | (y << 16), pp.first_sample + b);
    results[i]    = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *count = total;
}

// ---- closest-hit pass (replaces optix/optix_kernel_raytrace.cu:147-183) ----
__global__ __launch_bounds__(kBlock) void k_trace(DeviceScene sc, PathQueue q, const uint32_t* count, uint64_t* counters) {
  const uint32_t n = *count;
  RayStats st{0, 0};
  uint32_t rays = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const float4 o4 = q.origin_t[i], d4 = q.dir_slot[i];
    const uint4 aux = q.aux[i], hid = q.hit_id[i];
    const Hit h = closest_hit(sc, v3(o4.x, o4.y, o4.z), v3(d4.x, d4.y, d4.z), (aux.w & kStUseIgnoreHandle) != 0, hid.x, hid.y, st);
    q.origin_t[i] = make_float4(o4.x, o4.y, o4.z, h.t);
    q.hit_id[i]   = make_uint4(h.instance_id, h.tri_id, hid.z, hid.w);
    rays++;
  }
  flush_stats(counters, st, rays, kCntTrace, kCntNodes, kCntTris);
}

// ---- surface context (cuda/geometry_utils.cuh:13-221, untextured) ----
LUM_DEV GeoContext build_context(const DeviceScene& sc, V3 hit_origin, V3 ray_world, uint32_t state, uint32_t inst, uint32_t tri, uint32_t medium) {
  const uint32_t mesh = sc.instance_mesh_ids[inst];
  const Transform tf = load_transform(sc, inst);
  const uint32_t tbase = sc.mesh_tri_offset[mesh] + tri;
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
  const V3 n0 = normal_unpack(fbits(a.w)), n1 = normal_unpack(fbits(b.w)), n2 = normal_unpack(fbits(c.w));
  const bool inside = dot(face, ray) > 0.0f;
  if (inside) face = face * -1.0f;
  V3 normal = lerp_normals(n0, n1 - n0, n2 - n0, co, face);
  normal = adapt_normal(ray * -1.0f, normal, face);
  Col albedo = mat.albedo;
  float alpha = mat.alpha;
  const bool emissive_side = !inside || (mat.flags & kDMatBidirectionalEmission);
  const bool emits = (mat.flags & kDMatEmission) && emissive_side && ((state & kStAllowEmission) != 0);
  const Col emission = emits ? mat.emission : col(0.0f, 0.0f, 0.0f);
  float roughness = mat.roughness;
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

// ---- shading pass: cuda/geometry.cuh:11-180 (+ miss handling of cuda/sky.cuh:609-633, roulette cuda/directives.cuh:11-32) ----
__global__ __launch_bounds__(kBlock, LUM_SHADE_WAVES) void k_shade(DeviceScene sc, PathQueue in, PathQueue out, NeeQueue nee, float4* results, const uint32_t* count_in,
                                                  uint32_t* count_out, uint32_t depth_const, uint64_t* counters) {
  const uint32_t n = *count_in;
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  const Col sky = (sc.sky_mode == kSkyConstantColor) ? col(sc.sky_constant_color[0], sc.sky_constant_color[1], sc.sky_constant_color[2]) : splat(0.0f);
  uint32_t vertices = 0;
  const uint32_t rounds = (n + gridDim.x * kBlock - 1) / (gridDim.x * kBlock);
  for (uint32_t round = 0; round < rounds; round++) {
    const uint32_t i = (round * gridDim.x + blockIdx.x) * kBlock + threadIdx.x;
    bool survive = false;
    float4 n_o, n_d; uint4 n_aux, n_hid;
    if (i < n) {
      const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
      const uint4 aux = in.aux[i], hid = in.hit_id[i];
      const uint32_t slot = fbits(d4.w), state = aux.w;
      const Col record_in = record_unpack(U2{aux.x, aux.y});
      if (hid.x == kHitSky) {
        if (state & kStAllowAmbient) add_to_result(results, slot, sky * record_in);
      }
      else {
        vertices++;
        const V3 origin = v3(o4.x, o4.y, o4.z), ray = v3(d4.x, d4.y, d4.z);
        const V3 hit_origin = origin + ray * o4.w;
        const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, hid.w, depth_const};
        const GeoContext g = build_context(sc, hit_origin, ray, state, hid.x, hid.y, aux.z);

        // NEE work (geometry.cuh:31-74; direct_lighting.cuh:352-443)
        const bool geo_allowed = lights_present && ((state & kStVolumeScattered) == 0);
        float root_sum = 0.0f;
        float4 geo_rd = make_float4(0.0f, 0.0f, 0.0f, 0.0f), geo_cl = make_float4(0.0f, 0.0f, 0.0f, bitsf(kLightIdInvalid));
        float4 bs_rp = make_float4(0.0f, 0.0f, 1.0f, 0.0f), bs_ws = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (geo_allowed) {
          const LightSample ls = sample_light(sc, g, smp);
          root_sum = ls.root_sum;
          geo_rd = make_float4(ls.ray.x, ls.ray.y, ls.ray.z, ls.dist);
          geo_cl = make_float4(ls.color.r, ls.color.g, ls.color.b, bitsf(ls.light_id));
          const LightDirSample lb = sample_light_direction(sc, g, smp);
          bs_rp = make_float4(lb.ray.x, lb.ray.y, lb.ray.z, lb.probability);
          bs_ws = make_float4(lb.weight.r, lb.weight.g, lb.weight.b, root_sum);
        }
        const BounceSample bounce = sample_bounce(sc, g, smp, 0);
        uint4 amb = make_uint4(0u, 0u, 0u, 0u);
        if (sc.sky_mode != kSkyDefault) {
          const U2 c = record_pack(sky * bounce.weight), r = ray_pack(bounce.ray);
          amb = make_uint4(c.x, c.y, r.x, r.y);
        }
        nee.geo_ray_dist[i] = geo_rd; nee.geo_color_light[i] = geo_cl;
        nee.bsdf_ray_prob[i] = bs_rp; nee.bsdf_weight_sum[i] = bs_ws;
        nee.ambient[i] = amb;

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
        if (sc.sky_mode != kSkyDefault && !pass_through) new_state &= ~kStAllowAmbient; else new_state |= kStAllowAmbient;
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
      }
    }
    // wave-aggregated append: one atomic per wave
    const unsigned long long ballot = __ballot(survive);
    if (ballot) {
      const uint32_t lane = threadIdx.x & 63;
      const uint32_t rank = __popcll(ballot & ((1ull << lane) - 1ull));
      uint32_t base = 0;
      const int leader = __ffsll((long long) ballot) - 1;
      if ((int) lane == leader) base = atomicAdd(count_out, (uint32_t) __popcll(ballot));
      base = __shfl(base, leader);
      if (survive) {
        const uint32_t j = base + rank;
        out.origin_t[j] = n_o; out.dir_slot[j] = n_d; out.aux[j] = n_aux; out.hit_id[j] = n_hid;
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vertices += __shfl_down(vertices, off);
  if ((threadIdx.x & 63) == 0 && vertices) atomicAdd((unsigned long long*) &counters[kCntVertices], (unsigned long long) vertices);
}

// ---- shadow pass: optix/optix_kernel_shadow.cu:15-100, cuda/direct_lighting.cuh:445-669 ----
__global__ __launch_bounds__(kBlock) void k_shadow(DeviceScene sc, PathQueue in, NeeQueue nee, float4* results, const uint32_t* count_in, uint32_t depth_const,
                                                   uint64_t* counters) {
  const uint32_t n = *count_in;
  const bool lights_present = sc.light_tree_root != nullptr && sc.num_lights > 0;
  RayStats st{0, 0};
  uint32_t rays = 0, light_queries = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const uint4 hid = in.hit_id[i];
    if (hid.x == kHitSky) continue;
    const float4 o4 = in.origin_t[i], d4 = in.dir_slot[i];
    const uint4 aux = in.aux[i];
    const V3 hit_origin = v3(o4.x, o4.y, o4.z) + v3(d4.x, d4.y, d4.z) * o4.w;
    const bool geo_allowed = lights_present && ((aux.w & kStVolumeScattered) == 0);
    Col acc = splat(0.0f);
    {  // sampled light (direct_lighting.cuh:445-464)
      const float4 rd = nee.geo_ray_dist[i], cl = nee.geo_color_light[i];
      const uint32_t light_id = fbits(cl.w);
      Col vis = splat(0.0f);
      if (light_id != kLightIdInvalid && geo_allowed) {
        const uint2 target = sc.light_tri_handles[light_id];
        vis = shadow_query(sc, hit_origin, v3(rd.x, rd.y, rd.z), rd.w, target.x, target.y, hid.x, hid.y, st);
        rays++;
      }
      acc = acc + col(cl.x, cl.y, cl.z) * vis;
    }
    {  // BSDF-sampled direction against the light-only BVH (direct_lighting.cuh:586-667)
      const float4 rp = nee.bsdf_ray_prob[i], ws = nee.bsdf_weight_sum[i];
      const V3 ray = v3(rp.x, rp.y, rp.z);
      bool valid = geo_allowed && rp.w != 0.0f;
      uint32_t light_id = kLightIdInvalid, num_hits = 0;
      if (valid) {
        const Sampler smp{sc.bluenoise_2d, hid.z & 0xFFFFu, hid.z >> 16, hid.w, depth_const};
        light_id = light_query(sc, hit_origin, ray, hid.x, hid.y, smp.next1(kRndLightBsdfTrace), num_hits, st);
        light_queries++;
      }
      valid = valid && light_id != kLightIdInvalid;
      float dist = kFltMax;
      uint2 handle = make_uint2(0xFFFFFFFFu, 0u);
      Col lc = splat(0.0f);
      if (light_id != kLightIdInvalid) {
        handle = sc.light_tri_handles[light_id];
        const TriLight tl = load_tri_light(sc, handle.x, handle.y);
        F2 uv;
        dist = intersect_triangle(tl.vertex, tl.edge1, tl.edge2, hit_origin, ray, uv);
        if (dist != kFltMax) {
          lc = tri_light_color(sc, tl);
          const float mis = mis_for_bsdf_ray(hit_origin, tl, lc, dist, rp.w, ws.w);
          lc = lc * (mis * num_hits);
          lc = lc * col(ws.x, ws.y, ws.z);
        }
        else valid = false;
      }
      Col vis = splat(0.0f);
      if (valid) { vis = shadow_query(sc, hit_origin, ray, dist, handle.x, handle.y, hid.x, hid.y, st); rays++; }
      acc = acc + lc * vis;
    }
    // sun: disabled in constant-colour mode (direct_lighting.cuh:262); procedural sky is out of scope
    {  // ambient (direct_lighting.cuh:521-584)
      const uint4 amb = nee.ambient[i];
      const bool allowed = sc.sky_mode != kSkyDefault;
      if (allowed) {
        Col vis = splat(0.0f);
        if (amb.x != 0 || amb.y != 0) { vis = shadow_query(sc, hit_origin, ray_unpack(U2{amb.z, amb.w}), kFltMax, 0xFFFFFFFFu, 0u, hid.x, hid.y, st); rays++; }
        acc = acc + record_unpack(U2{amb.x, amb.y}) * vis;
      }
    }
    add_to_result(results, fbits(d4.w), acc * record_unpack(U2{aux.x, aux.y}));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) light_queries += __shfl_down(light_queries, off);
  if ((threadIdx.x & 63) == 0 && light_queries) atomicAdd((unsigned long long*) &counters[kCntLightBvh], (unsigned long long) light_queries);
  flush_stats(counters, st, rays, kCntShadow, kCntNodesShadow, kCntTrisShadow);
}

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

// ---- standalone closest-hit entry for traversal tests and the trace micro-benchmark ----
__global__ __launch_bounds__(kBlock) void k_trace_rays(DeviceScene sc, uint32_t n, const float* origins, const float* dirs, const uint32_t* ignore,
                                                       uint32_t* out, uint64_t* counters) {
  RayStats st{0, 0};
  uint32_t rays = 0;
  for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
    const bool ign = ignore != nullptr && ignore[2 * i] != 0xFFFFFFFFu;
    const Hit h = closest_hit(sc, v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), ign,
                              ign ? ignore[2 * i] : 0u, ign ? ignore[2 * i + 1] : 0u, st);
    out[3 * i] = h.instance_id; out[3 * i + 1] = h.tri_id; out[3 * i + 2] = fbits(h.t);
    rays++;
  }
  flush_stats(counters, st, rays, kCntTrace, kCntNodes, kCntTris);
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

}  // namespace lum
