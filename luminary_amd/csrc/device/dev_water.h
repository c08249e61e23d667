// What the water surface does to the sun and ambient samples of a vertex below it, and the surface's own shading context.
// Reference: cuda/direct_lighting.cuh:123-243 (the sun through the surface: direct_lighting_sun_caustic), :466-584 (evaluation of sun and ambient
// samples: one visibility ray, or two when the vertex is under water), cuda/caustics.cuh (connection point on the surface: the fast path and the
// resampled patch), cuda/ris.cuh:176-259 (stratified reservoir), cuda/ocean_utils.cuh:477-523 (surface context, origin shift).
#pragma once

#include "dev_particle.h"

LUM_NS_BEGIN

// ---- visibility of a sun / ambient sample (direct_lighting.cuh:466-584), split into what is known when the sample is created (the segment(s) to
// trace and the factors between them) and the combination with the traced visibilities (k_resolve / k_volume_resolve) ----
struct SkyRayPlan {
  bool valid, second, total_reflection;
  float limit, fresnel_factor;  // first segment's length; 1 - Fresnel reflection at the surface
  V3 second_origin, second_dir;
  Col t1, t2;                    // ambient only: transmittance of the vertex's volume up to the limit, of the volume beyond the surface
};
LUM_DEV SkyRayPlan plan_sun_ray(const DeviceScene& sc, V3 origin, uint32_t self_inst, uint32_t volume_id, bool nonzero, V3 ray) {
  SkyRayPlan p{nonzero, false, false, kFltMax, 1.0f, v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 1.0f), splat(1.0f), splat(1.0f)};
  const bool is_caustics_path = volume_id == kVolumeOcean && self_inst != kHitOcean;
  if (is_caustics_path && ray.y > 0.0f) {
    const float dist = (ocean_max_height(sc) - origin.y) / ray.y;
    p.limit = (dist > 0.0f) ? dist : kFltMax;
  }
  if (p.valid && is_caustics_path && p.limit != kFltMax) {
    p.second = true;
    p.second_origin = origin + ray * p.limit;
    const bool fast_path = sc.ocean_amplitude == 0.0f || !sc.ocean_caustics_active;  // caustics_is_fast_path<GEOMETRY>, caustics.cuh:50-58
    const V3 ocean_normal = fast_path ? v3(0.0f, -1.0f, 0.0f) : ocean_get_normal(sc, p.second_origin) * -1.0f;
    p.second_dir = refract(ray * -1.0f, ocean_normal, sc.ocean_refractive_index, p.total_reflection);
    p.fresnel_factor = 1.0f - ocean_reflection_coefficient(ocean_normal, ray, p.second_dir, 1.0f / sc.ocean_refractive_index);
  }
  return p;
}
LUM_DEV SkyRayPlan plan_ambient_ray(const DeviceScene& sc, V3 origin, uint32_t self_inst, uint32_t volume_id, uint32_t second_volume, bool nonzero, V3 ray) {
  SkyRayPlan p{nonzero, false, false, kFltMax, 1.0f, v3(0.0f, 0.0f, 0.0f), v3(0.0f, 0.0f, 1.0f), splat(1.0f), splat(1.0f)};
  const bool is_caustics_path = volume_id == kVolumeOcean && self_inst != kHitOcean;
  if (is_caustics_path && ray.y > 0.0f) {
    const float dist = (ocean_max_height(sc) - origin.y) / ray.y;
    p.limit = (dist > 0.0f) ? dist : kFltMax;
  }
  else if (ray.y < 0.0f && sc.ocean_active && origin.y > ocean_min_height(sc)) p.valid = false;  // the sample would have to cross the water from above
  p.t1 = volume_transmittance(sc, volume_id, origin, ray, p.limit);
  if (p.valid && is_caustics_path && p.limit != kFltMax) {
    p.second = true;
    p.second_origin = origin + ray * p.limit;
    const V3 ocean_normal = ocean_get_normal(sc, p.second_origin) * -1.0f;
    const V3 ocean_V = ray * -1.0f;
    p.second_dir = refract(ocean_V, ocean_normal, sc.ocean_refractive_index, p.total_reflection);
    p.fresnel_factor = 1.0f - fresnel_dielectric(ocean_normal, ocean_V, p.second_dir, sc.ocean_refractive_index);
    p.t2 = volume_transmittance(sc, second_volume, p.second_origin, p.second_dir, kFltMax);
  }
  return p;
}
constexpr uint32_t kSkyRaySecond = 1u, kSkyRayTotalReflection = 2u;
LUM_DEV uint32_t sky_ray_flags(const SkyRayPlan& p) { return (p.second ? kSkyRaySecond : 0u) | (p.total_reflection ? kSkyRayTotalReflection : 0u); }
// light = colour x vis1 [x (1 - fresnel)] x vis2
LUM_DEV Col combine_sun_ray(Col color, Col vis1, float fresnel_factor, uint32_t flags, Col vis2) {
  Col light = color * vis1;
  if (flags & kSkyRaySecond) { light = light * fresnel_factor; light = light * ((flags & kSkyRayTotalReflection) ? splat(0.0f) : vis2); }
  return light;
}
// light = colour x vis1 x t1 [x (1 - fresnel) x t2] x vis2
LUM_DEV Col combine_ambient_ray(Col color, Col vis1, Col t1, float fresnel_factor, Col t2, uint32_t flags, Col vis2) {
  Col light = color * vis1;
  light = light * t1;
  if (flags & kSkyRaySecond) { light = light * fresnel_factor; light = light * t2; light = light * ((flags & kSkyRayTotalReflection) ? splat(0.0f) : vis2); }
  return light;
}

// ---- the sun seen from under water ----
LUM_DEV Col sun_evaluate_ctx(const DeviceScene& sc, const GeoContext& g, V3 dir, float one_over_pdf) {
  const Energy energy = energy_terms(sc, g.params, world_ndotv(g));
  bool is_refraction;
  return eval_bsdf(energy, g, dir, kHintGeneral, is_refraction, one_over_pdf);
}
LUM_DEV Col sun_evaluate_ctx(const DeviceScene& sc, const VolContext& c, V3 dir, float one_over_pdf) { return splat(volume_phase_evaluate(sc, c, dir) * one_over_pdf); }
LUM_DEV Col sun_evaluate_ctx(const DeviceScene& sc, const ParticleContext& c, V3 dir, float one_over_pdf) { return particles_albedo(sc) * (particle_phase(sc, c, dir) * one_over_pdf); }
template <class Ctx> struct CausticsTraits { static constexpr bool kSurface = false; };
template <> struct CausticsTraits<GeoContext> { static constexpr bool kSurface = true; };

struct CausticsDomain { bool valid; V3 base, edge1, edge2; float area; bool fast_path; };
template <class Ctx>
LUM_DEV CausticsDomain caustics_get_domain(const DeviceScene& sc, const SkyView& sky, const Ctx& c, V3 L) {  // caustics.cuh:21-35, :60-123, under water
  bool total_reflection;
  const V3 ray = refract(L, v3(0.0f, 1.0f, 0.0f), 1.0f / sc.ocean_refractive_index, total_reflection) * -1.0f;
  const float dist = ocean_intersection_distance(sc, c.position, ray, kFltMax);
  const V3 center = c.position + ray * dist;
  CausticsDomain d;
  d.valid = dist != kFltMax;
  d.fast_path = !CausticsTraits<Ctx>::kSurface || sc.ocean_amplitude == 0.0f || !sc.ocean_caustics_active;
  if (d.fast_path) {
    d.base = center; d.edge1 = v3(0.0f, 0.0f, 0.0f); d.edge2 = v3(0.0f, 0.0f, 0.0f);
    d.area = sphere_solid_angle(sky.sun_pos, kSkySunRadius, world_to_sky(sky, c.position));
    return d;
  }
  const V3 center_dir = normalize(center - c.position);
  const float altitude = asin_det(center_dir.y);  // direction_to_angles, math.cuh:790-797
  float azimuth = atan2_det(center_dir.z, center_dir.x);
  if (azimuth < 0.0f) azimuth += 2.0f * kPi;
  const float angle = 0.3f * sc.ocean_caustics_domain_scale, plane_height = center.y;
  V3 vd[3];
  const float alts[3] = {altitude - angle, altitude - angle, altitude + angle}, azis[3] = {azimuth - angle, azimuth + angle, azimuth - angle};
#pragma unroll
  for (int k = 0; k < 3; k++) {  // angles_to_direction, math.cuh:781-788
    float sa, ca, sz, cz;
    sincos_det(alts[k], sa, ca); sincos_det(azis[k], sz, cz);
    const V3 dir = v3(cz * ca, sa, sz * ca);
    const float dd = fabsf(c.position.y - plane_height) / fmaxf(0.01f, fabsf(dir.y));
    vd[k] = c.position + dir * dd;
  }
  d.base = vd[0]; d.edge1 = vd[1] - vd[0]; d.edge2 = vd[2] - vd[0];
  d.area = length(cross(d.edge1, d.edge2));
  return d;
}
template <class Ctx, class Smp>
LUM_DEV bool caustics_find_connection_point(const DeviceScene& sc, const SkyView& sky, const Ctx& c, const Smp& smp, uint32_t rnd_initial, const CausticsDomain& d,
                                            uint32_t iteration, uint32_t num_iterations, V3& point, float& sample_weight) {  // caustics.cuh:125-163, refraction
  if (d.fast_path) { point = d.base; sample_weight = d.area; return true; }
  const F2 r = smp.next2(rnd_initial + iteration);
  const float sx = (iteration + r.x) * (1.0f / num_iterations), sy = r.y;  // ris_transform_stratum_2D, ris.cuh:166-174
  point = d.base + (d.edge1 * sx + d.edge2 * sy);
  V3 V = c.position - point;
  const float dist_sq = dot(V, V);
  V = V * rsqrt_ieee(dist_sq);
  const V3 normal = ocean_get_normal_fast(sc, point) * -1.0f;
  if (dot(V, normal) < 0.0f) return false;
  bool total_reflection;
  const V3 L = refract(V, normal, sc.ocean_refractive_index, total_reflection);
  if (!sphere_hit(L, world_to_sky(sky, point), sky.sun_pos, kSkySunRadius)) return false;
  sample_weight = fabsf(V.y) * d.area / dist_sq;
  return true;
}
// direct_lighting_sun_caustic; `set`: the context's LIGHT_SUN random set (0 surface / particle, 1 volume)
template <class Ctx, class Smp>
LUM_DEV bool sun_caustic_sample(const DeviceScene& sc, const SkyView& sky, const Ctx& c, const Smp& smp, uint32_t set, uint32_t volume_type, uint32_t second_volume,
                                Col& light_out, V3& dir_out) {
  const uint32_t rnd_initial = 81u + 128u * set, rnd_resampling = 338u + set, rnd_sun_ray = 341u + set;  // CAUSTIC_INITIAL / _RESAMPLING / _SUN_RAY
  const V3 sky_pos = world_to_sky(sky, c.position);
  float solid_angle;
  const V3 sun_dir = sample_sphere(sky.sun_pos, kSkySunRadius, sky_pos, smp.next2(rnd_sun_ray), solid_angle);
  const CausticsDomain domain = caustics_get_domain(sc, sky, c, sun_dir);
  if (!domain.valid) return false;
  V3 connection_point = v3(0.0f, 0.0f, 0.0f);
  float connection_weight;
  if (domain.fast_path) caustics_find_connection_point(sc, sky, c, smp, rnd_initial, domain, 0u, 1u, connection_point, connection_weight);
  else {
    const uint32_t num_samples = sc.ocean_caustics_ris_sample_count + 1u;
    // ris_stratified_reservoir (ris.cuh:176-259): strata are taken from both ends, the side whose weight sum lags behind the random split is extended
    uint32_t iteration = 0, index_front = 0xFFFFFFFFu, index_back = num_samples;
    float sum_front = 0.0f, sum_back = 0.0f, selected_target = 0.0f;
    const float random = smp.next1(rnd_resampling);
    const float mis_weight = 1.0f / num_samples;
#pragma nounroll
    for (;;) {
      if (iteration > num_samples) break;
      const bool compute_front = sum_front <= random * (sum_front + sum_back);
      if (!compute_front && iteration == num_samples) break;
      iteration++;
      const uint32_t index = compute_front ? ++index_front : --index_back;
      if (index == num_samples) break;
      V3 sample_point; float sample_weight = 0.0f;
      const bool valid_hit = caustics_find_connection_point(sc, sky, c, smp, rnd_initial, domain, index, num_samples, sample_point, sample_weight);
      const float target = valid_hit ? 1.0f : 0.0f;
      sample_weight = valid_hit ? mis_weight * sample_weight : 0.0f;
      const float weight = target * sample_weight;
      if (weight == 0.0f) continue;
      const bool front = sum_front <= random * (sum_front + sum_back);
      selected_target = front ? target : selected_target;
      if (iteration <= num_samples) { if (front) sum_front += weight; else sum_back += weight; }
      if (front) connection_point = sample_point;
    }
    connection_weight = (selected_target > 0.0f) ? (sum_front + sum_back) / selected_target : 0.0f;
    connection_weight *= sc.ocean_refractive_index * sc.ocean_refractive_index;
    connection_weight *= sc.ocean_refractive_index * sc.ocean_refractive_index * 2.0f;
  }
  if (connection_weight == 0.0f) return false;
  const V3 pos_to_ocean = connection_point - c.position;
  const float dist = length(pos_to_ocean);
  const V3 dir = normalize(pos_to_ocean);
  Col light = sky_sun_color(sky, world_to_sky(sky, connection_point), sun_dir);
  light = light * sun_evaluate_ctx(sc, c, dir, connection_weight);
  if (importance(light) == 0.0f) return false;
  light = light * volume_transmittance(sc, volume_type, c.position, dir, dist);
  light = light * volume_transmittance(sc, second_volume, connection_point, sun_dir, kFltMax);
  light_out = light; dir_out = dir;
  return true;
}

// ocean_get_context (ocean_utils.cuh:477-517): the water surface as a smooth translucent material
LUM_DEV GeoContext ocean_context(const DeviceScene& sc, V3 position, V3 ray, uint32_t state, uint32_t medium) {
  V3 normal = ocean_get_normal(sc, position);
  const bool inside_water = dot(ray, normal) > 0.0f;
  if (inside_water) normal = normal * -1.0f;
  uint32_t flags = kMatTranslucent;
  if (inside_water) flags |= kMatRefractionInside;
  const float other_ior = medium_ior_peek(medium, inside_water);
  const float ior_ratio = inside_water ? sc.ocean_refractive_index / other_ior : other_ior / sc.ocean_refractive_index;
  const float roughness = (state & kStDeltaPath) ? 0.02f * 2.0f : 0.25f;  // BSDF_ROUGHNESS_CLAMP * 2
  GeoContext g;
  g.instance_id = kHitOcean; g.tri_id = 0u;
  g.normal = normal;
  g.face_normal_packed = normal_pack(normal);
  g.position = position;
  g.V = ray * -1.0f;
  g.state = state;
  g.params.flags = flags;
  g.params.set(splat(1.0f), 1.0f, roughness, splat(0.0f), ior_ratio);
  return g;
}

LUM_NS_END
