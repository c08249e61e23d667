// Clouds, second part: the ray march through the three layers (cuda/cloud.cuh). Per step inside a cloud: the sun through the cloud (a short march of
// `shadow_steps` quadratically spaced density samples) and one ambient direction (the sky marched with half the sky's step count, through the cloud
// likewise), scattered by the Jendersie-Eon phase function of the droplet diameter and summed over `octaves` octaves that halve scattering,
// extinction and the phase function's asymmetry (an approximation of multiple scattering). dev_cloud.h holds the density function.
#pragma once

#include "dev_volume.h"

LUM_NS_BEGIN

LUM_DEV float cloud_extinction(const DeviceScene& sc, V3 origin, V3 ray, int layer) {  // cloud.cuh:49-81
  const float iter_step = 1.0f / (float) (int) sc.cloud_shadow_steps;
  float optical_depth = 0.0f;
#pragma nounroll
  for (float i = 0.0f; i < 1.0f; i += iter_step) {
    float t0 = i, t1 = i + iter_step;
    t0 = t0 * t0;
    t1 = t1 * t1;
    const float step_size = t1 - t0;
    const float reach = t0 + step_size * 0.5f;
    const V3 pos = origin + ray * reach;
    const float height = cloud_height(sc, pos, layer);
    if (height > 1.0f || height < 0.0f) break;
    const CloudWeather w = cloud_weather(sc, pos, height, layer);
    if (cloud_significant_point(height, w, layer)) optical_depth -= cloud_density(sc, pos, height, w, layer) * step_size;
  }
  optical_depth *= kCloudExtinctionDensity;
  return exp_det(optical_depth);
}
struct CloudResult { Col scattered_light; float transmittance, hit_dist; };
LUM_DEV float je_phase_function_ms(const float* p, float c, float ms_factor) {  // math.cuh:1234-1239 with the octave's factor on both asymmetries
  return (1.0f - p[3]) * hg_phase(c, p[0] * ms_factor) + p[3] * draine_phase(c, p[1] * ms_factor, p[2]);
}
// The march of one ray through one layer (clouds_compute, cloud.cuh:86-262) as a state the caller steps: `find` advances to the next step that lies in a
// cloud (cheap, trip count differs from ray to ray), `light` lights that step (a sky march and two cloud-shadow marches). The reference interleaves both
// in one loop; cut in two, a wave lights the steps of all its rays together, and k_clouds_march can hand a finished lane the next ray. Per ray the
// operations and their order are the reference's.
struct CloudMarch {
  V3 origin, ray, ambient_ray;
  Col scattered_light;
  float step_size, reach, sun_solid_angle, ambient_cos_angle, transmittance, hit_dist, sky_step_random;
  int layer, i, step_count;
  bool hit;
  // false: nothing to march, result() holds the empty result
  LUM_DEV bool begin(const DeviceScene& sc, const SkyView& sky, const Sampler& smp, V3 origin_, V3 ray_, float start, float dist, int layer_) {
    origin = origin_; ray = ray_; layer = layer_;
    scattered_light = splat(0.0f); transmittance = 1.0f; hit_dist = start; hit = false; i = 0; step_count = 0;
    if (dist < 0.0f || start == kFltMax) return false;
    const float* L = sc.cloud_layers[layer];
    const float span = L[kClHeightMax] - L[kClHeightMin];
    dist = fminf(6.0f * span, dist);
    const int base_steps = (layer == kCloudLow) ? (int) sc.cloud_steps : (layer == kCloudMid) ? (int) sc.cloud_steps / 4 : (int) sc.cloud_steps / 8;
    step_count = (int) ((float) base_steps * saturate(dist / (6.0f * span)));
    step_count = (int) ((float) step_count + 8.0f * smp.next1(kRndCloudStepCount + (uint32_t) layer));
    start = fmaxf(0.0f, start);
    step_size = dist / (float) step_count;
    const float random_offset = smp.next1(kRndCloudStepOffset + (uint32_t) layer);
    reach = start + (0.1f + random_offset * 0.9f) * step_size;
    sun_solid_angle = sphere_solid_angle(sky.sun_pos, kSkySunRadius, origin + ray * reach);
    hit_dist = start;
    const F2 ambient_r = smp.next2(kRndCloudDir);
    ambient_ray = sample_ray_sphere(2.0f * ambient_r.x - 1.0f, ambient_r.y);
    ambient_cos_angle = dot(ray, ambient_ray);
    sky_step_random = smp.next1(kRndSkyStepOffset);
    return true;
  }
  // (A) false: the march is over (the ray left the layer or used up its steps)
  LUM_DEV bool find(const DeviceScene& sc, V3& pos, float& density) {
#pragma nounroll
    for (; i < step_count; i++) {
      pos = origin + ray * reach;
      if (!hit) hit_dist = reach;
      const float height = cloud_height(sc, pos, layer);
      if (height < 0.0f || height > 1.0f) return false;
      const CloudWeather w = cloud_weather(sc, pos, height, layer);
      if (!cloud_significant_point(height, w, layer)) { reach += step_size; continue; }
      density = cloud_density(sc, pos, height, w, layer);
      if (density > 0.0f) return true;
      reach += step_size;
    }
    return false;
  }
  // (B) false: the march is over (less than 10 % of the light behind gets through)
  LUM_DEV bool light(const DeviceScene& sc, const SkyView& sky, V3 pos, float density) {
    hit = true;
    const Col ambient_color = sky_get_color(sc, sky, pos, ambient_ray, kFltMax, false, (int) (sky.steps / 2u), sky_step_random);
    float ambient_extinction = cloud_extinction(sc, pos, ambient_ray, layer);
    Col sun_color;
    float sun_extinction, sun_cos_angle;
    const V3 sun_ray = normalize(sky.sun_pos - pos);
    if (!sph_hit_p0(sun_ray, pos, kSkyEarthRadius)) {
      sun_color = sky_sun_color(sky, pos, sun_ray, false);
      sun_cos_angle = dot(ray, sun_ray);
      sun_extinction = cloud_extinction(sc, pos, sun_ray, layer);
    }
    else { sun_color = splat(0.0f); sun_extinction = 1.0f; sun_cos_angle = 0.0f; }
    float scattering = density * kCloudScatteringDensity;
    float extinction = fmaxf(density * kCloudExtinctionDensity, 0.0001f);
    float phase_factor = 1.0f;
#pragma nounroll
    for (uint32_t o = 0; o < sc.cloud_octaves; o++) {
      scattering *= 0.5f;
      extinction *= 0.5f;
      const float sun_phase = je_phase_function_ms(sc.cloud_phase, sun_cos_angle, phase_factor);
      const float ambient_phase = je_phase_function_ms(sc.cloud_phase, ambient_cos_angle, phase_factor);
      phase_factor *= 0.5f;
      const Col sun_color_i = sun_color * (sun_extinction * sun_phase * sun_solid_angle);
      const Col ambient_color_i = ambient_color * (ambient_extinction * ambient_phase * 4.0f * kPi);
      sun_extinction = sqrtf(sun_extinction);
      ambient_extinction = sqrtf(ambient_extinction);
      Col S = sun_color_i + ambient_color_i;
      S = S * scattering;
      const float step_trans = exp_det(-extinction * step_size);
      S = (S - S * step_trans) * (1.0f / extinction);
      scattered_light = scattered_light + S * transmittance;
    }
    transmittance *= exp_det(-density * kCloudExtinctionDensity * step_size);
    if (transmittance < 0.1f) { transmittance = 0.0f; return false; }
    reach += step_size;
    i++;
    return true;
  }
  LUM_DEV CloudResult result() const { return CloudResult{scattered_light, transmittance, hit_dist}; }
};
LUM_DEV CloudResult clouds_compute(const DeviceScene& sc, const SkyView& sky, const Sampler& smp, V3 origin, V3 ray, float start, float dist, int layer) {
  CloudMarch m;
  if (m.begin(sc, sky, smp, origin, ray, start, dist, layer)) {
    V3 pos = origin;
    float density = 0.0f;
#pragma nounroll
    while (m.find(sc, pos, density)) {
      if (!m.light(sc, sky, pos, density)) break;
    }
  }
  return m.result();
}
// clouds_render (cloud.cuh:268-334): the layers in the order a ray enters them; with atmosphere_scattering the air between them is marched as well.
// `layer_result(l, start, distance)`: the march of layer l - computed on the spot (the panorama bake), or read back (k_clouds after k_clouds_march).
template <class LayerResult>
LUM_DEV float clouds_render_with(const DeviceScene& sc, const SkyView& sky, const Sampler& smp, V3 origin, V3 ray, float limit, Col& color, Col& transmittance,
                                 float& transmittance_cloud_only, LayerResult&& layer_result) {
  float starts[3];
  CloudResult results[3];
#pragma nounroll
  for (int l = 0; l < 3; l++) {
    const F2 isect = cloud_layer_intersection(sc, origin, ray, limit, l);
    starts[l] = isect.x;
    results[l] = layer_result(l, isect.x, isect.y);
  }
  const bool less01 = starts[0] <= starts[1], less02 = starts[0] <= starts[2], less12 = starts[1] <= starts[2];
  int order[3];
  if (less01) {
    if (less02) { order[0] = 0; order[1] = less12 ? 1 : 2; order[2] = less12 ? 2 : 1; }
    else { order[0] = 2; order[1] = less01 ? 0 : 1; order[2] = less01 ? 1 : 0; }
  }
  else if (less12) { order[0] = 1; order[1] = less02 ? 0 : 2; order[2] = less02 ? 2 : 0; }
  else { order[0] = 2; order[1] = less01 ? 0 : 1; order[2] = less01 ? 1 : 0; }
  float prev_start = 0.0f;
#pragma nounroll
  for (int i = 0; i < 3; i++) {
    const CloudResult r = results[order[i]];
    if (r.hit_dist == kFltMax) break;
    if (sc.cloud_atmosphere_scattering) {
      color = color + sky_trace_inscattering(sc, sky, origin, ray, r.hit_dist - prev_start, transmittance, smp.depth == 0u, smp.next1(kRndSkyInscatteringStep),
                                             smp.next1(kRndSkyStepOffset));
      origin = origin + ray * (r.hit_dist - prev_start);
    }
    color = color + r.scattered_light * transmittance;
    transmittance = transmittance * r.transmittance;
    transmittance_cloud_only *= r.transmittance;
    prev_start = r.hit_dist;
  }
  return prev_start;
}
LUM_DEV float clouds_render(const DeviceScene& sc, const SkyView& sky, const Sampler& smp, V3 origin, V3 ray, float limit, Col& color, Col& transmittance,
                            float& transmittance_cloud_only) {
  return clouds_render_with(sc, sky, smp, origin, ray, limit, color, transmittance, transmittance_cloud_only,
                            [&](int l, float start, float dist) { return clouds_compute(sc, sky, smp, origin, ray, start, dist, l); });
}

LUM_NS_END
