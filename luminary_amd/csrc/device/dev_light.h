// Next-event estimation: light-tree traversal with resampling, solid-angle triangle sampling, BSDF-driven light
// directions and the MIS weights between them.
// Reference: cuda/ris.cuh:22-158, cuda/light_tree.cuh:71-320, cuda/light_triangle.cuh:10-280, cuda/light.cuh:49-159,
// cuda/light_bsdf.cuh:24-146, cuda/mis.cuh:19-57. Tree node formats: device_utils.h:283-327.
#pragma once

#include "dev_bsdf.h"

LUM_NS_BEGIN

constexpr uint32_t kLightTreeOutputs = 8;
#ifndef LUM_ABLATE_LIGHT
#define LUM_ABLATE_LIGHT 0  // measurement only: 1 no BSDF evaluation, 2 no MIS weight, 4 one resampling lane in the root pass (results are wrong)
#endif
constexpr uint32_t kLightIdInvalid   = 0xFFFFFFFFu;

LUM_DEV Transform load_transform(const DeviceScene& sc, uint32_t inst) {
  const float4 a = sc.instance_transforms[2 * inst], b = sc.instance_transforms[2 * inst + 1];
  Transform t;
  t.translation = v3(a.x, a.y, a.z);
  t.scale = v3(a.w, b.x, b.y);
  t.rot_xy = fbits(b.z); t.rot_zw = fbits(b.w);
  return t;
}

// ---- resampling (ris.cuh) ----
struct Reservoir {
  float sum_weight, selected_target, random;
  LUM_DEV void reset() { sum_weight = 0.0f; selected_target = 0.0f; }
  LUM_DEV bool add(float target, float sampling_weight) {
    const float w = target * sampling_weight;
    sum_weight += w;
    if (w == 0.0f) return false;
    const float prob = w / sum_weight;
    const bool accept = random < prob;
    selected_target = accept ? target : selected_target;
    const float shift = accept ? 0.0f : prob, scale = accept ? prob : 1.0f - prob;
    random = clamp_random((random - shift) / scale);
    return accept;
  }
  LUM_DEV float sampling_weight() const { return (selected_target > 0.0f) ? sum_weight / selected_target : 0.0f; }
};

// ---- light tree ----
LUM_DEV float tree_importance(const GeoContext& g, float power, V3 mean, float std_dev) {  // light_tree.cuh:71-89
  const V3 po = mean - g.position;
  const float dist_sq = dot(po, po);
  const float variance = std_dev * std_dev;
  const float inv = 1.0f / (dist_sq + variance);
  const float r = power * inv;
  if ((g.params.flags & kMatSubstrateMask) == kMatTranslucent) return r;
  const float t = variance * inv;
  const float NdotL = saturate(dot(po, g.normal) * sqrtf(inv));
  return r * (NdotL * (1.0f - t) + t);
}
LUM_DEV uint32_t byte_of(uint32_t lo, uint32_t hi, uint32_t i) { return ((i < 4 ? lo : hi) >> ((i & 3) * 8)) & 0xFFu; }

struct ChildBlock { uint32_t mx0, mx1, my0, my1, mz0, mz1, sd0, sd1; };  // 4 x 8 bytes: rel mean x,y,z, rel std dev
// The tree is walked for a surface vertex or for a ray segment through the fog (VolContext, dev_volume.h): the contexts differ in
// tree_importance() and in the random set they draw from (material.cuh:60-63, :78-81).
struct GeoTreeTargets { static constexpr uint32_t kPrepass = kRndLightTreePrepass, kPostpass = kRndLightTreePostpass; };
template <class Ctx> struct TreeTargets : GeoTreeTargets {};
template <class Ctx>
LUM_DEV float child_importance(const Ctx& g, const ChildBlock& b, uint32_t power_q, V3 base, V3 ex, float exp_v, uint32_t i) {
  if (power_q == 0) return 0.0f;
  const float power = (float) power_q;
  const float std_dev = byte_of(b.sd0, b.sd1, i) * exp_v;
  const V3 mean = v3((float) byte_of(b.mx0, b.mx1, i), (float) byte_of(b.my0, b.my1, i), (float) byte_of(b.mz0, b.mz1, i)) * ex + base;
  return fmaxf(tree_importance(g, power, mean, std_dev), 0.0f);
}

struct TreeWork { uint32_t cont[kLightTreeOutputs]; float root_sum; };  // cont: is_light | index << 1 | probability(20 bit) << 9

// Root pass (light_tree.cuh:191-255): one scan over <= 128 children feeds 8 independent resampling lanes.
template <class Ctx>
LUM_DEV TreeWork tree_prepass(const DeviceScene& sc, const Ctx& g, const Sampler& smp) {
  const uint4 h = sc.light_tree_root[0];
  const uint32_t num_root_lights = h.y >> 16, num_sections = (h.z >> 16) & 0xFFu;
  const V3 base = v3(bfloat_unpack(h.x), bfloat_unpack(h.x >> 16), bfloat_unpack(h.y));
  const V3 ex = v3(exp2i((int8_t) (h.w & 0xFF)), exp2i((int8_t) ((h.w >> 8) & 0xFF)), exp2i((int8_t) ((h.w >> 16) & 0xFF)));
  const float exp_v = exp2i((int8_t) (h.w >> 24));
  float lane_random[kLightTreeOutputs], lane_target[kLightTreeOutputs];
  uint32_t lane_pick[kLightTreeOutputs];
#pragma unroll
  for (uint32_t l = 0; l < kLightTreeOutputs; l++) {
    lane_random[l] = smp.next1(TreeTargets<Ctx>::kPrepass + l);
    lane_target[l] = 0.0f;
    lane_pick[l] = 0;
  }
  float total = 0.0f, sum = 0.0f;
  for (uint32_t s = 0; s < num_sections; s++) {
    const uint4 q0 = sc.light_tree_root[1 + 3 * s], q1 = sc.light_tree_root[2 + 3 * s], q2 = sc.light_tree_root[3 + 3 * s];
    const ChildBlock blk{q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    const uint32_t pw[4] = {q2.x, q2.y, q2.z, q2.w};
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
      const uint32_t power_q = (pw[c >> 1] >> ((c & 1) * 16)) & 0xFFFFu;
      const float target = child_importance(g, blk, power_q, base, ex, exp_v, c);
      total += target;
      const float prob = (target > 0.0f) ? target / total : 0.0f;
      if (prob == 0.0f) continue;
      sum += target;
      // The eight lanes divide by one of two numbers per child; the reciprocals are formed once (the reference, built with
      // --use_fast_math, multiplies by a reciprocal here as well: cuda/ris.cuh:138-148).
      float inv_accept = 1.0f / prob, inv_reject = 1.0f / (1.0f - prob);
      // keep them two divisions: without the barrier the compiler rewrites `accept ? 1/p : 1/(1-p)` as `1 / (accept ? p : 1-p)` in
      // each of the eight lanes (same bits, eight correctly rounded divisions of 11 instructions instead of two)
      asm volatile("" : "+v"(inv_accept), "+v"(inv_reject));
#pragma unroll
      for (uint32_t l = 0; l < ((LUM_ABLATE_LIGHT & 4) ? 1u : kLightTreeOutputs); l++) {
        const bool accept = lane_random[l] < prob;
        lane_target[l] = accept ? target : lane_target[l];
        const float shifted = accept ? lane_random[l] : lane_random[l] - prob;
        lane_random[l] = clamp_random(shifted * (accept ? inv_accept : inv_reject));
        if (accept) lane_pick[l] = s * 8 + c;
      }
    }
  }
  TreeWork w;
  w.root_sum = sum * (bfloat_unpack(h.z) / 65535.0f);
#pragma unroll
  for (uint32_t l = 0; l < kLightTreeOutputs; l++) {
    const bool is_light = lane_pick[l] < num_root_lights;
    const uint32_t index = (is_light ? lane_pick[l] : lane_pick[l] - num_root_lights) & 0xFFu;
    const float p = (total > 0.0f) ? lane_target[l] / total : 0.0f;
    uint32_t q = 0;
    if (p > 0.0f) q = max((uint32_t) ((1048575.0f * p) + 0.5f), 1u) & 0xFFFFFu;
    w.cont[l] = (is_light ? 1u : 0u) | (index << 1) | (q << 9);
  }
  return w;
}

struct TreePick { uint32_t light_id; float weight; };

// Descent of one lane through the 8-wide nodes (light_tree.cuh:257-320).
template <class Ctx>
LUM_DEV TreePick tree_postpass(const DeviceScene& sc, const Ctx& g, const Sampler& smp, uint32_t lane, const TreeWork& w) {
  const uint32_t cont = w.cont[lane];
  const float cp = (cont >> 9) * (1.0f / 1048575.0f) * kLightTreeOutputs;
  TreePick r;
  r.light_id = kLightIdInvalid;
  r.weight = (cp > 0.0f) ? 1.0f / cp : 0.0f;
  if (cp == 0.0f) return r;
  const uint32_t index = (cont >> 1) & 0xFFu;
  if (cont & 1u) { r.light_id = index; return r; }
  uint32_t node_id = index;
  Reservoir rv;
  rv.random = smp.next1(TreeTargets<Ctx>::kPostpass + lane);
  rv.reset();
  while (r.light_id == kLightIdInvalid) {
    LUM_STAT(12, 13);
    const uint4 n0 = sc.light_tree_nodes[4 * node_id], n1 = sc.light_tree_nodes[4 * node_id + 1], n2 = sc.light_tree_nodes[4 * node_id + 2],
                n3 = sc.light_tree_nodes[4 * node_id + 3];
    const V3 base = v3(bfloat_unpack(n0.x), bfloat_unpack(n0.x >> 16), bfloat_unpack(n0.y));
    const V3 ex = v3(exp2i((int8_t) (n0.z & 0xFF)), exp2i((int8_t) ((n0.z >> 8) & 0xFF)), exp2i((int8_t) ((n0.z >> 16) & 0xFF)));
    const float exp_v = exp2i((int8_t) (n0.z >> 24));
    const uint32_t num_lights = n0.w & 0xFFu, child_ptr = n1.x, light_ptr = n1.y;
    const ChildBlock blk{n1.z, n1.w, n2.x, n2.y, n2.z, n2.w, n3.x, n3.y};
    uint32_t pick = 0xFF;
#pragma unroll
    for (uint32_t c = 0; c < 8; c++) {
      const float target = child_importance(g, blk, byte_of(n3.z, n3.w, c), base, ex, exp_v, c);
      if (rv.add(target, 1.0f)) pick = c;
    }
    if (pick == 0xFF) break;
    r.weight *= rv.sampling_weight();
    if (pick < num_lights) { r.light_id = light_ptr + pick; break; }
    node_id = child_ptr + (pick - num_lights);
    rv.reset();
  }
  return r;
}

// ---- triangle lights ----
struct TriLight { V3 vertex, edge1, edge2; uint32_t material_id, scene_tri; bool bidirectional; };

LUM_DEV TriLight load_tri_light(const DeviceScene& sc, uint32_t inst, uint32_t tri) {  // light_triangle.cuh:37-72
  const uint32_t mesh = sc.instance_mesh_ids[inst];
  const Transform tf = load_transform(sc, inst);
  const uint32_t base = (sc.mesh_tri_offset[mesh] + tri) * 3;
  const float4 a = sc.vertices[base], b = sc.vertices[base + 1], c = sc.vertices[base + 2];
  const V3 p0 = v3(a.x, a.y, a.z);
  TriLight t;
  t.vertex = xf_point(tf, p0);
  t.edge1 = xf_rel(tf, v3(b.x, b.y, b.z) - p0);
  t.edge2 = xf_rel(tf, v3(c.x, c.y, c.z) - p0);
  t.scene_tri = sc.mesh_tri_offset[mesh] + tri;
  t.material_id = sc.tri_tex[t.scene_tri].w & 0xFFFFu;
  t.bidirectional = (sc.materials[2 * t.material_id].x & kDMatBidirectionalEmission) != 0;
  return t;
}
LUM_DEV float tri_light_solid_angle(const TriLight& t, V3 origin) {  // light_triangle.cuh:94-108
  const V3 a = normalize(t.vertex - origin), b = normalize((t.vertex + t.edge1) - origin), c = normalize((t.vertex + t.edge2) - origin);
  const float G0 = fabsf(dot(cross(a, b), c)), G1 = dot(a, c) + dot(b, c), G2 = 1.0f + dot(a, b);
  return 2.0f * atan2_det(G0, G1 + G2);
}
LUM_DEV float tri_light_area(const TriLight& t) { return length(cross(t.edge1, t.edge2)) * 0.5f; }
LUM_DEV bool not_finite(float a) { return isnan(a) || isinf(a); }
// Solid-angle sampling, Peters 2021 (light_triangle.cuh:114-157)
LUM_DEV bool sample_tri_solid_angle(V3 origin, const TriLight& t, F2 rnd, V3& ray, float& solid_angle) {
  const V3 a = normalize(t.vertex - origin), b = normalize((t.vertex + t.edge1) - origin), c = normalize((t.vertex + t.edge2) - origin);
  const float G0s = dot(cross(a, b), c);
  if (!t.bidirectional && (G0s >= 0.0f)) return false;
  const float G0 = fabsf(G0s), G1 = dot(a, c) + dot(b, c), G2 = 1.0f + dot(a, b);
  solid_angle = 2.0f * atan2_det(G0, G1 + G2);
  if (not_finite(solid_angle) || solid_angle < 1e-7f) return false;
  const float ssa = rnd.x * solid_angle;
  float sh, ch; sincos_det(0.5f * ssa, sh, ch);
  const V3 r = a * (G0 * ch - G1 * sh) + c * (G2 * sh);
  const V3 ct = r * (2.0f * dot(a, r) / dot(r, r)) - a;
  const float s2 = dot(b, ct);
  const float sv = (1.0f - rnd.y) + rnd.y * s2;
  const float tt = sqrtf(fmaxf((1.0f - sv * sv) / (1.0f - s2 * s2), 0.0f));
  ray = normalize(b * (sv - tt * s2) + ct * tt);
  return !(not_finite(ray.x) || not_finite(ray.y) || not_finite(ray.z));
}
// light_get_color, light_triangle.cuh:245-280. `coords`: barycentrics of the point on the light (light_triangle_sample_finalize_dist_and_uvs,
// :74-92); the texture coordinates are only formed for materials that have an emission or albedo texture.
LUM_DEV Col tri_light_color(const DeviceScene& sc, const TriLight& t, F2 coords) {
  const Material m = load_material(sc, t.material_id);
  Col c = m.emission;
  float alpha = m.alpha;
  if (m.luminance_tex != kTextureNone || m.albedo_tex != kTextureNone) {
    const F2 tex = triangle_uv(sc.tri_tex[t.scene_tri], coords);
    if (m.luminance_tex != kTextureNone) {
      const float4 e = texture_load(sc, m.luminance_tex, tex, true, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
      c = col(e.x, e.y, e.z) * m.emission_scale;
    }
    if (any_positive(c) && m.albedo_tex != kTextureNone) alpha = texture_load(sc, m.albedo_tex, tex, true, make_float4(0.0f, 0.0f, 0.0f, 1.0f)).w;
  }
  if (any_positive(c)) c = c * alpha;
  return c;
}

// ---- BSDF-driven light direction (light_bsdf.cuh) ----
struct LightDirSample { V3 ray; Col weight; float probability; };
LUM_DEV float light_dir_roughness(float r) { return lerpf(r, 1.0f, 0.04f); }
LUM_DEV float light_dir_rr(float r) { return remap01(r, 0.5f, 0.1f); }

LUM_DEV LightDirSample sample_light_direction(const LocalFrame& lf, const GeoContext& g, const Sampler& smp) {
  LightDirSample out;
  out.ray = v3(0.0f, 0.0f, 1.0f); out.weight = splat(0.0f); out.probability = 0.0f;
  const MatParams& p = g.params;
  const Quat to_z = lf.to_z;
  const V3 Vl = lf.V, fnl = lf.face_normal;
  const V3 up = v3(0.0f, 0.0f, 1.0f);
  const bool with_refraction = (p.flags & kMatSubstrateMask) == kMatTranslucent;
  const uint32_t num_tech = with_refraction ? 2 : 1;
  const float refr_prob = with_refraction ? 1.0f / num_tech : 0.0f;
  const uint32_t tech = (uint32_t) (smp.next1(kRndLightBsdfChoice) * num_tech);
  const bool refraction = (tech == 1) && with_refraction;
  const float roughness = p.roughness();
  const float rr = light_dir_rr(roughness);
  if (smp.next1(kRndLightBsdfRR) >= rr) return out;
  const float sr = light_dir_roughness(roughness);
  if (!refraction) {
    const V3 m = sample_vndf_bounded(Vl, sr, smp.next2(kRndLightBsdfDirection));
    const V3 ray = reflect(Vl, m);
    const RayTerms c = sampled_direction_terms(p, up, Vl, m, ray, false);
    const float pdf = pdf_vndf_bounded(Vl, sr, c.NdotH, c.NdotV);
    out.weight = eval_with_face_normal(lf.energy, p, c, kHintGeneral, ray, fnl, 1.0f / pdf);
    out.ray = ray;
    out.probability = (1.0f - refr_prob) * pdf;
  }
  else {
    const float ior = p.ior();
    bool total_reflection;
    const V3 m = sample_vndf_caps(Vl, sr, smp.next2(kRndLightBsdfDirection));
    const V3 ray = refract(Vl, m, ior, total_reflection);
    const RayTerms c = sampled_direction_terms(p, up, Vl, m, ray, !total_reflection);
    const float pdf = pdf_refraction(sr, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior);
    out.weight = eval_with_face_normal(lf.energy, p, c, kHintGeneral, ray, fnl, 1.0f / pdf);
    out.ray = ray;
    out.probability = refr_prob * pdf;
  }
  out.weight = out.weight * (1.0f / rr);
  out.probability *= rr;
  out.ray = normalize(qapply(qinv(to_z), out.ray));
  return out;
}
LUM_DEV float light_direction_probability(const GeoContext& g, V3 L) {  // light_bsdf.cuh:104-146
  const MatParams& p = g.params;
  const Quat to_z = rotation_to_z(g.normal);
  const V3 Vl = normalize(qapply(to_z, g.V)), Ll = normalize(qapply(to_z, L));
  const bool with_refraction = (p.flags & kMatSubstrateMask) == kMatTranslucent;
  const uint32_t num_tech = with_refraction ? 2 : 1;
  const float refr_prob = with_refraction ? 1.0f / num_tech : 0.0f;
  const RayTerms c = analyze_direction(p, v3(0.0f, 0.0f, 1.0f), Vl, Ll);
  const float roughness = p.roughness();
  const float sr = light_dir_roughness(roughness);
  float prob;
  if (c.is_refraction) prob = refr_prob * pdf_refraction(sr, c.NdotH, c.NdotV, c.HdotV, c.HdotL, p.ior());
  else prob = (1.0f - refr_prob) * pdf_vndf_bounded(Vl, sr, c.NdotH, c.NdotV);
  return prob * light_dir_rr(roughness);
}

// ---- MIS (mis.cuh) ----
LUM_DEV float mis_base(float gi_pdf, float solid_angle, float power, float dist_sq, float root_sum) {
  const float dl_pdf = 8 * (1.0f / solid_angle) * (power / dist_sq) * (1.0f / root_sum);
  return (dl_pdf > 0.0f) ? gi_pdf / (gi_pdf + dl_pdf) : 1.0f;
}
LUM_DEV float mis_for_bsdf_ray(V3 origin, const TriLight& t, Col color, float dist, float gi_pdf, float root_sum) {
  if (root_sum == 0.0f) return 1.0f;
  const float area = tri_light_area(t), sa = tri_light_solid_angle(t, origin);
  return mis_base(gi_pdf, sa, importance(color) * area, dist * dist, root_sum);
}
LUM_DEV float mis_for_light_sample(const GeoContext& g, V3 L, const TriLight& t, Col color, float dist, float solid_angle, float root_sum) {
  const float power = importance(color) * tri_light_area(t);
  return 1.0f - mis_base(light_direction_probability(g, L), solid_angle, power, dist * dist, root_sum);
}

// ---- light sampling (light.cuh:84-159) ----
struct LightSample { uint32_t light_id; V3 ray; Col color; float dist, root_sum; };

LUM_DEV LightSample sample_light(const DeviceScene& sc, const GeoContext& g, const Sampler& smp) {
  LUM_STAT(14, 15);
  const TreeWork work = tree_prepass(sc, g, smp);
  const Energy energy = energy_terms(sc, g.params, world_ndotv(g));
  LightSample out;
  out.light_id = kLightIdInvalid; out.ray = v3(0.0f, 0.0f, 0.0f); out.color = splat(0.0f); out.dist = 0.0f;
  Reservoir rv;
  rv.random = smp.next1(kRndLightGeoResampling);
  rv.reset();
#ifndef LUM_ABLATE_LANES
#define LUM_ABLATE_LANES kLightTreeOutputs  // measurement only: fewer resampling lanes evaluated (results are wrong)
#endif
#pragma nounroll
  for (uint32_t lane = 0; lane < LUM_ABLATE_LANES; lane++) {
    LUM_STAT(8, 9);
    const TreePick pick = tree_postpass(sc, g, smp, lane, work);
    if (pick.light_id == kLightIdInvalid) continue;
    const uint2 handle = sc.light_tri_handles[pick.light_id];
    if (handle.x == g.instance_id && handle.y == g.tri_id) continue;
    const TriLight tl = load_tri_light(sc, handle.x, handle.y);
    V3 ray; float sa;
    if (!sample_tri_solid_angle(g.position, tl, smp.next2(kRndLightGeoRay + lane), ray, sa)) continue;
    F2 uv;
    const float dist = intersect_triangle(tl.vertex, tl.edge1, tl.edge2, g.position, ray, uv);
    if (dist == kFltMax) continue;
    LUM_STAT(10, 11);
    Col lc = tri_light_color(sc, tl, uv);
    bool is_refraction;
#ifndef LUM_ABLATE_LIGHT_DEFINED_BELOW
#define LUM_ABLATE_LIGHT_DEFINED_BELOW
#endif
#if 0
#define LUM_ABLATE_LIGHT 0  // measurement only: 1 no BSDF evaluation, 2 no MIS weight, 4 one resampling lane in the root pass (results are wrong)
#endif
    const Col bw = (LUM_ABLATE_LIGHT & 1) ? splat(0.5f) : eval_bsdf(energy, g, ray, kHintGeneral, is_refraction, 1.0f);
    const float mis = (LUM_ABLATE_LIGHT & 2) ? 0.5f : mis_for_light_sample(g, ray, tl, lc, dist, sa, work.root_sum);
    lc = (lc * bw) * mis;
    if (rv.add(importance(lc), pick.weight * sa)) { out.light_id = pick.light_id; out.ray = ray; out.color = lc; out.dist = dist; }
  }
  out.color = out.color * rv.sampling_weight();
  out.root_sum = work.root_sum;
  return out;
}

LUM_NS_END
