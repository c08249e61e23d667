/*
 * ORACLE (test infrastructure, not product): the per-sample path loop.
 * Follows the reference's kernel schedule, device/device_renderer.c:53-134 (one pass per depth 0..max_ray_depth):
 *   tasks_create                 cuda/kernels.cuh:45-193, cuda/camera.cuh:11-38, cuda/camera_thin_lens.cuh:8-86
 *   trace                        optix/optix_kernel_raytrace.cu:147-183          (o_trace.h)
 *   geometry_process_tasks       cuda/geometry.cuh:11-180, cuda/geometry_utils.cuh:54-221, cuda/direct_lighting.cuh:352-443
 *   shadow                       optix/optix_kernel_shadow.cu:15-100, cuda/direct_lighting.cuh:445-669
 *   sky_process_tasks            cuda/sky.cuh:567-633                               (o_sky.h)
 *   volume_process_* (fog, water) cuda/volume.cuh, cuda/light_bridges.cuh             (o_volume.h)
 *   particle_process_tasks       cuda/particle.cuh:7-108, optix_kernel_raytrace.cu:97-131
 *   ocean_process_tasks          cuda/ocean.cuh:12-102, cuda/caustics.cuh, optix_kernel_raytrace.cu:134-144 (o_ocean.h)
 *   accumulate                   cuda/memory.cuh:359-368, cuda/accumulation.cuh:63-84
 * Paths are independent (all randomness is a function of pixel, sample id, depth and target), so the oracle walks
 * them one at a time instead of in wavefronts; per path the order of every floating-point operation is the reference's.
 * Quirk kept: the depth constant the sampler sees is not advanced before the last pass (device_renderer.c:126-130).
 * Out of scope: clouds, physical camera.
 */
#define _GNU_SOURCE /* qsort_r (o_trace.h) */
#include <stdio.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "o_trace.h"
#include "o_output.h"
#include "o_adaptive.h"
enum { ST_DELTA_PATH = 1, ST_CAMERA_DIRECTION = 2, ST_VOLUME_SCATTERED = 4, ST_ALLOW_EMISSION = 8, ST_ALLOW_AMBIENT = 16, ST_USE_IGNORE_HANDLE = 32 };
#include "o_sky.h"
#include "o_volume.h"
#include "o_cloud.h"

enum { SKY_MODE_DEFAULT = 0, SKY_MODE_HDRI = 1, SKY_MODE_CONSTANT_COLOR = 2 };
#define GEOMETRY_DELTA_PATH_CUTOFF 0.05f
#define BSDF_ROUGHNESS_CLAMP 2e-2f
#define RUSSIAN_ROULETTE_CLAMP (1.0f / 8.0f)
#define CAMERA_COMMON_SCALE 0.001f
#define CAMERA_COMMON_INV_SCALE (1.0f / CAMERA_COMMON_SCALE)

/* ---- camera (camera_thin_lens.cuh:8-86, camera.cuh:29-35) ---- */
static void camera_sample(const OracleScene* s, const Sampler* smp, vec3* origin, vec3* ray) {
  const uint2_t jq = rng_2d_u32(s->bluenoise_2d, RT_CAMERA_JITTER, 0, 0, smp->sample_id, 0); /* camera_utils.cuh:23-27 */
  const float jx = u32_to_unit(jq.x), jy = u32_to_unit(jq.y);
  const float step = 2.0f * (s->cam_fov / s->width);
  const float vfov = step * s->height * 0.5f;
  vec3 sp;
  sp.x = s->cam_fov - step * (smp->px + jx);
  sp.y = -vfov + step * (smp->py + jy);
  sp.z = 1.0f;
  const vec3 s2f = v_norm(v_sub(v3(0.0f, 0.0f, 0.0f), sp));
  const float focal = fmaxf(s->cam_object_distance * CAMERA_COMMON_INV_SCALE, 0.01f);
  const vec3 fp = v_scale(s2f, -focal / s2f.z);
  vec3 ap = v3(0.0f, 0.0f, 0.0f);
  if (s->cam_aperture_size != 0.0f) {
    const float2_t r = rnd2(smp, RT_LENS);
    const float asz = s->cam_aperture_size * CAMERA_COMMON_INV_SCALE;
    float sx, sy;
    if (s->cam_aperture_shape == 1) {
      const int blade = (int) (rnd1(smp, RT_LENS_BLADE) * s->cam_aperture_blade_count);
      const float alpha = sqrtf(r.x), beta = r.y;
      const float u = 1.0f - alpha, v = alpha * beta;
      const float astep = (2.0f * O_PI) / s->cam_aperture_blade_count;
      float s1, c1, s2, c2;
      o_sincos(astep * blade, &s1, &c1); o_sincos(astep * (blade + 1), &s2, &c2);
      sx = (s1 * u + s2 * v) * asz; sy = (c1 * u + c2 * v) * asz;
    }
    else {
      const float alpha = r.x * 2.0f * O_PI, beta = sqrtf(r.y) * asz;
      float sa, ca; o_sincos(alpha, &sa, &ca);
      sx = ca * beta; sy = sa * beta;
    }
    ap = v3(sx, sy, 0.0f);
  }
  const Quat q = {s->cam_rotation[0], s->cam_rotation[1], s->cam_rotation[2], s->cam_rotation[3]};
  vec3 o = q_apply(q, ap);
  o = v_scale(o, s->cam_scale * CAMERA_COMMON_SCALE);
  o = v_add(o, v3(s->cam_pos[0], s->cam_pos[1], s->cam_pos[2]));
  *origin = o;
  *ray = q_apply(q, v_norm(v_sub(fp, ap)));
}

/* ---- geometry context (geometry_utils.cuh:13-221, untextured) ---- */
static GeoCtx geometry_get_context(const OracleScene* s, vec3 hit_origin, vec3 task_ray, uint16_t state, uint32_t inst, uint32_t tri, uint32_t medium_ior) {
  const uint32_t mesh = s->instance_mesh_ids[inst];
  const OTransform tf = scene_transform(s, inst);
  const OVertex a = scene_vertex(s, mesh, tri, 0), b = scene_vertex(s, mesh, tri, 1), c = scene_vertex(s, mesh, tri, 2);
  const uint32_t* tt = scene_tritex(s, mesh, tri);
  vec3 position = t_apply_inv(tf, hit_origin);
  const vec3 ray = t_rot_inv(tf, task_ray);
  const vec3 e1 = v_sub(b.pos, a.pos), e2 = v_sub(c.pos, a.pos);
  vec3 face_normal = v_norm(v_cross(e1, e2));
  const float2_t co = tri_coords(a.pos, e1, e2, position);
  position = v_add(a.pos, v_add(v_scale(e1, co.x), v_scale(e2, co.y)));
  position = t_apply(tf, position);
  const OMaterial mat = scene_material(s, tt[3] & 0xFFFF);
  const vec3 n0 = normal_unpack(a.normal), n1 = normal_unpack(b.normal), n2 = normal_unpack(c.normal);
  const vec3 e1n = v_sub(n1, n0), e2n = v_sub(n2, n0);
  /* geometry_compute_normal, geometry_utils.cuh:13-52 */
  const bool is_inside = v_dot(face_normal, ray) > 0.0f;
  if (is_inside) face_normal = v_scale(face_normal, -1.0f);
  vec3 normal = lerp_normals(n0, e1n, e2n, co, face_normal);
  const UV tex_coords = triangle_uv(tt, co);
  if (mat.normal_tex != TEXTURE_NONE) { /* geometry_utils.cuh:25-50 */
    const bool valid = mat.normal_tex < s->num_textures;
    const float4_t nf = texture_load(s, mat.normal_tex, tex_coords, false, f4(0.0f, 0.0f, 1.0f, 0.0f));
    vec3 mn = v3(nf.x, nf.y, nf.z);
    if ((mat.flags & DMAT_NORMAL_MAP_COMPRESSED) && valid) mn = v_sub(v_scale(mn, 2.0f), v3(1.0f, 1.0f, 1.0f));
    mn = v_norm(mn);
    normal = q_apply(q_inverse(q_rotation_to_z(normal)), mn);
  }
  normal = normal_adaptation_apply(v_scale(ray, -1.0f), normal, face_normal);

  RGBAF albedo = mat.albedo;
  if (mat.albedo_tex != TEXTURE_NONE) { /* geometry_utils.cuh:109-121 */
    const float4_t af = texture_load(s, mat.albedo_tex, tex_coords, true, f4(0.9f, 0.9f, 0.9f, 1.0f));
    albedo.r = af.x; albedo.g = af.y; albedo.b = af.z; albedo.a = af.w;
  }
  const bool emissive_side = (!is_inside) || (mat.flags & DMAT_BIDIRECTIONAL_EMISSION);
  const bool has_emission = (mat.flags & DMAT_EMISSION) && emissive_side;
  const bool include_emission = has_emission && ((state & ST_ALLOW_EMISSION) != 0);
  RGBF emission = c3(0.0f, 0.0f, 0.0f);
  if (include_emission) emission = mat.emission;
  if (include_emission && mat.luminance_tex != TEXTURE_NONE) { /* geometry_utils.cuh:130-137 */
    const float4_t lf = texture_load(s, mat.luminance_tex, tex_coords, true, f4(0.0f, 0.0f, 0.0f, 0.0f));
    emission = c_scale(c3(lf.x, lf.y, lf.z), albedo.a * mat.emission_scale);
  }
  float roughness = mat.roughness;
  if (mat.roughness_tex != TEXTURE_NONE) roughness = texture_load(s, mat.roughness_tex, tex_coords, true, f4(0.5f, 0.0f, 0.0f, 0.0f)).x; /* :140-150 */
  if (mat.flags & DMAT_ROUGHNESS_AS_SMOOTHNESS) roughness = 1.0f - roughness;
  roughness = fmaxf(roughness, BSDF_ROUGHNESS_CLAMP);
  if ((state & ST_DELTA_PATH) == 0) roughness = fmaxf(roughness, mat.roughness_clamp);
  uint32_t flags = mat.flags & DMAT_SUBSTRATE_MASK;
  if (mat.metallic_tex != TEXTURE_NONE) { /* stochastic metallic textures: not implemented in the reference either */ }
  else if (mat.flags & DMAT_METALLIC) flags |= MAT_METALLIC;
  if (mat.flags & DMAT_COLORED_TRANSPARENCY) flags |= MAT_COLORED_TRANSPARENCY;
  if (is_inside) flags |= MAT_REFRACTION_IS_INSIDE;
  const bool refr_inside = (flags & MAT_REFRACTION_IS_INSIDE) != 0;
  const float other_ior = medium_ior_peek(medium_ior, refr_inside);
  const float ior_in = refr_inside ? mat.refraction_index : other_ior;
  const float ior_out = refr_inside ? other_ior : mat.refraction_index;
  if (((flags & MAT_SUBSTRATE_MASK) == MAT_TRANSLUCENT) && (fabsf(1.0f - ior_in / ior_out) < 1e-4f)) {
    if ((flags & MAT_COLORED_TRANSPARENCY) == 0) {
      albedo.r = o_lerp(1.0f, albedo.r, albedo.a);
      albedo.g = o_lerp(1.0f, albedo.g, albedo.a);
      albedo.b = o_lerp(1.0f, albedo.b, albedo.a);
    }
    albedo.a = 0.0f;
    flags |= MAT_COLORED_TRANSPARENCY;
  }
  GeoCtx g;
  g.instance_id = inst; g.tri_id = tri;
  g.normal = t_rot(tf, normal);
  g.face_normal = normal_pack(face_normal);
  g.position = position;
  g.V = v_scale(task_ray, -1.0f);
  g.state = state;
  g.params.data[0] = g.params.data[1] = g.params.data[2] = 0;
  g.params.flags = flags;
  mp_set_albedo(&g.params, c3(albedo.r, albedo.g, albedo.b));
  mp_set_opacity(&g.params, albedo.a);
  mp_set_roughness(&g.params, roughness);
  mp_set_emission(&g.params, emission);
  mp_set_ior(&g.params, ior_in / ior_out);
  return g;
}

static inline void beauty_add(RGBF* result, RGBF v) { if (c_any(v)) *result = c_add(*result, v); } /* memory.cuh:359-368 */

/* directives.cuh:11-32 */
static bool russian_roulette(const OracleScene* s, const Sampler* smp, uint16_t old_state, RGBF* record) {
  if (old_state & ST_DELTA_PATH) return true;
  const float value = c_importance(*record);
  if (value < s->cam_rr_threshold) {
    const float p = (value > 0.0f) ? fmaxf(value / s->cam_rr_threshold, RUSSIAN_ROULETTE_CLAMP) : 0.0f;
    if (rnd1(smp, RT_RUSSIAN_ROULETTE) > p) return false;
    *record = c_scale(*record, 1.0f / p);
  }
  return true;
}

/* Debug shading modes: one closest-hit pass, then a colour per hit (geometry_process_tasks_debug, cuda/geometry.cuh:182-246) or per miss
 * (sky_process_tasks_debug, cuda/sky.cuh:635-665); queue: device/device_renderer.c:136-181. */
static void volume_events(const OracleScene* s, const Sampler* smp_p, vec3 origin, vec3 ray, uint16_t state, uint32_t volume_type, OHit* hit_p, uint2_t* record_pp, RGBF* result_p);
static void particles_trace(const OracleScene* s, const OTracer* tr, const Sampler* smp, vec3 origin, vec3 ray, uint16_t state, OHit* hit);
static inline bool particle_is_hit(uint32_t instance_id);
#define HIT_TYPE_PARTICLE_MASK 0x7FFFFFFFu
static RGBF render_path_debug(const OracleScene* s, const OTracer* tr, uint32_t px, uint32_t py, uint32_t sample_id, uint64_t* cnt) {
  Sampler smp = {s->bluenoise_2d, px, py, sample_id, 0};
  RGBF result = c_splat(0.0f);
  vec3 origin, ray;
  camera_sample(s, &smp, &origin, &ray);
  const uint16_t state = ST_DELTA_PATH | ST_CAMERA_DIRECTION | ST_ALLOW_EMISSION | ST_ALLOW_AMBIENT;
  const uint32_t medium = medium_ior_modify(0, (s->ocean_active && origin.y < s->ocean_height) ? s->ocean_refractive_index : 1.0f, true);
  OHit hit = trace_closest(tr, origin, ray, false, 0, 0);
  cnt[ORACLE_CNT_TRACE]++;
  particles_trace(s, tr, &smp, origin, ray, state, &hit);
  if (s->ocean_active) {
    const float ocean_depth = ocean_intersection_distance(s, origin, ray, hit.t);
    if (ocean_depth < hit.t) { hit.t = ocean_depth; hit.instance_id = HIT_TYPE_OCEAN; hit.tri_id = 0; }
  }
  uint32_t volumes = 0;
  if (s->fog_active) volumes = volume_stack_modify(volumes, VOLUME_TYPE_FOG, true);
  if (s->ocean_active && ocean_is_underwater(s, origin)) volumes = volume_stack_modify(volumes, VOLUME_TYPE_OCEAN, true);
  uint2_t record_p = record_pack(c_splat(1.0f));
  if (volume_stack_peek(volumes, false) != VOLUME_TYPE_NONE) { /* the debug queue keeps volume_process_events (device_renderer.c:145-147): the sky fast path shows through, a scattering event stays black */
    volume_events(s, &smp, origin, ray, state, volume_stack_peek(volumes, false), &hit, &record_p, &result);
  }
  if (s->sky_aerial_perspective && s->sky_mode != SKY_MODE_CONSTANT_COLOR && s->sky_lut_transmittance && s->sky_lut_multiscattering && hit.instance_id != HIT_TYPE_SKY) {
    /* the debug queue keeps sky_process_inscattering_events (device_renderer.c:150-154): the air's light shows in every mode, on every task that is not a sky hit */
    const OSky view = osky_view(s);
    RGBF record = record_unpack(record_p);
    beauty_add(&result, sky_trace_inscattering(&view, world_to_sky(&view, origin), ray, hit.t * 0.001f, &record, true, rnd1(&smp, RANDOM_TARGET_SKY_INSCATTERING_STEP),
                                               rnd1(&smp, RANDOM_TARGET_SKY_STEP_OFFSET)));
  }
  if (hit.instance_id == HIT_TYPE_INVALID || (hit.instance_id >= HIT_TYPE_VOLUME_BASE && hit.instance_id <= HIT_TYPE_VOLUME_MAX)) return result;
  if (hit.instance_id == HIT_TYPE_OCEAN) { /* ocean_process_tasks_debug, ocean.cuh:104-145 */
    if (s->shading_mode == 2) beauty_add(&result, c_splat(o_saturate((1.0f / hit.t) * 2.0f)));
    else if (s->shading_mode == 3) {
      const vec3 n = ocean_get_normal(s, v_add(origin, v_scale(ray, hit.t)));
      beauty_add(&result, c3(o_saturate(0.5f * n.x + 0.5f), o_saturate(0.5f * n.y + 0.5f), o_saturate(0.5f * n.z + 0.5f)));
    }
    else if (s->shading_mode == 4) beauty_add(&result, c3(0.0f, 0.0f, 1.0f));
    return result;
  }
  if (hit.instance_id == HIT_TYPE_SKY) {
    if (s->shading_mode == 1) { /* ALBEDO: sky_color_main(origin, ray, STATE_FLAG_CAMERA_DIRECTION) */
      RGBF sky = c3(s->sky_constant_color[0], s->sky_constant_color[1], s->sky_constant_color[2]);
      if (s->sky_mode == SKY_MODE_DEFAULT) {
        sky = c_splat(0.0f);
        if (s->sky_lut_transmittance && s->sky_lut_multiscattering) {
          const OSky view = osky_view(s);
          sky = sky_get_color(&view, world_to_sky(&view, origin), ray, FLT_MAX, true, (int) view.steps, rnd1(&smp, RANDOM_TARGET_SKY_STEP_OFFSET));
        }
      }
      else if (s->sky_mode == SKY_MODE_HDRI) sky = sky_hdri_color(s, origin, ray, ST_CAMERA_DIRECTION);
      beauty_add(&result, sky);
    }
    else if (s->shading_mode == 4) beauty_add(&result, c3(0.0f, 0.63f, 1.0f)); /* IDENTIFICATION */
    return result;
  }
  if (particle_is_hit(hit.instance_id)) { /* particle_process_tasks_debug, particle.cuh:110-163 */
    switch (s->shading_mode) {
      case 1: beauty_add(&result, c3(s->particles_albedo[0], s->particles_albedo[1], s->particles_albedo[2])); break;
      case 2: beauty_add(&result, c_splat(o_saturate((1.0f / hit.t) * 2.0f))); break;
      case 3: {
        const float* qn = s->particle_normals + 4 * (size_t) (hit.instance_id & HIT_TYPE_PARTICLE_MASK);
        const vec3 quad_normal = v3(qn[0], qn[1], qn[2]);
        const vec3 n = (v_dot(ray, quad_normal) < 0.0f) ? quad_normal : v_scale(quad_normal, -1.0f);
        beauty_add(&result, c3(o_saturate(n.x), o_saturate(n.y), o_saturate(n.z)));
      } break;
      case 4: {
        const uint32_t v = squares32(0x55555555u, hit.instance_id);
        beauty_add(&result, c3(((float) (v & 0x7ffu)) / 0x7ff, ((float) ((v >> 10) & 0x7ffu)) / 0x7ff, ((float) ((v >> 20) & 0x7ffu)) / 0x7ff));
      } break;
      default: break;
    }
    return result;
  }
  cnt[ORACLE_CNT_VERTICES]++;
  const vec3 hit_origin = v_add(origin, v_scale(ray, hit.t));
  switch (s->shading_mode) {
    case 1: { /* ALBEDO */
      const GeoCtx g = geometry_get_context(s, hit_origin, ray, state, hit.instance_id, hit.tri_id, medium);
      beauty_add(&result, c_add(mp_albedo(&g.params), mp_emission(&g.params)));
    } break;
    case 2: beauty_add(&result, c_splat(o_saturate((1.0f / hit.t) * 2.0f))); break; /* DEPTH */
    case 3: { /* NORMAL */
      const GeoCtx g = geometry_get_context(s, hit_origin, ray, state, hit.instance_id, hit.tri_id, medium);
      beauty_add(&result, c3(o_saturate(g.normal.x), o_saturate(g.normal.y), o_saturate(g.normal.z)));
    } break;
    case 4: { /* IDENTIFICATION */
      const uint32_t v = squares32(0x55555555u, (hit.instance_id << 16) | hit.tri_id);
      beauty_add(&result, c3(((float) (v & 0x7ffu)) / 0x7ff, ((float) ((v >> 10) & 0x7ffu)) / 0x7ff, ((float) ((v >> 20) & 0x7ffu)) / 0x7ff));
    } break;
    case 5: { /* LIGHTS */
      const GeoCtx g = geometry_get_context(s, hit_origin, ray, state, hit.instance_id, hit.tri_id, medium);
      beauty_add(&result, c_add(c_scale(mp_albedo(&g.params), 0.025f), mp_emission(&g.params)));
    } break;
    default: break;
  }
  return result;
}


/* ---- evaluation of the sun and ambient samples of a vertex (direct_lighting.cuh:466-584): one visibility ray, or two when the vertex is under water -
 * up to the water surface, then along the direction refracted there, weighted by the Fresnel transmission and the second volume's transmittance ---- */
static RGBF sun_evaluate(const OracleScene* s, const OTracer* tr, uint64_t* cnt, vec3 origin, uint32_t self_inst, uint32_t self_tri, uint32_t volume_id, uint2_t color_p,
                         uint2_t ray_p, bool allowed) {
  const vec3 ray = ray_unpack(ray_p);
  float limit = FLT_MAX;
  const bool is_caustics_path = volume_id == VOLUME_TYPE_OCEAN && self_inst != HIT_TYPE_OCEAN;
  if (is_caustics_path && ray.y > 0.0f) {
    const float dist = (ocean_max_height(s) - origin.y) / ray.y;
    limit = (dist > 0.0f) ? dist : FLT_MAX;
  }
  const bool valid = (color_p.x != 0 || color_p.y != 0) && allowed;
  RGBF vis = c_splat(0.0f);
  if (valid) { cnt[ORACLE_CNT_SHADOW]++; vis = trace_shadow(tr, origin, ray, limit, 0xFFFFFFFFu, 0, self_inst, self_tri); }
  RGBF light = c_mul(record_unpack(color_p), vis);
  RGBF vis2 = c_splat(1.0f); /* OPTIX_TRACE_STATUS_OPTIONAL_UNUSED */
  if (valid && is_caustics_path && limit != FLT_MAX) {
    const vec3 ocean_pos = v_add(origin, v_scale(ray, limit));
    /* caustics_is_fast_path<GEOMETRY> (caustics.cuh:50-58; its fourth term is constant false by operator precedence) */
    const bool fast_path = s->ocean_amplitude == 0.0f || !s->ocean_caustics_active;
    const vec3 ocean_normal = fast_path ? v3(0.0f, -1.0f, 0.0f) : v_scale(ocean_get_normal(s, ocean_pos), -1.0f);
    bool total_reflection;
    const vec3 refraction = refract_vector(v_scale(ray, -1.0f), ocean_normal, s->ocean_refractive_index, &total_reflection);
    const float fresnel = ocean_reflection_coefficient(ocean_normal, ray, refraction, 1.0f / s->ocean_refractive_index);
    light = c_scale(light, 1.0f - fresnel);
    if (total_reflection) vis2 = c_splat(0.0f);
    else { cnt[ORACLE_CNT_SHADOW]++; vis2 = trace_shadow(tr, ocean_pos, refraction, FLT_MAX, 0xFFFFFFFFu, 0, 0xFFFFFFFFu, 0); }
  }
  if (!valid) vis2 = c_splat(1.0f);
  return c_mul(light, vis2);
}
static RGBF ambient_evaluate(const OracleScene* s, const OTracer* tr, uint64_t* cnt, vec3 origin, uint32_t self_inst, uint32_t self_tri, uint32_t volume_id, uint32_t second_volume,
                             uint2_t color_p, uint2_t ray_p, bool allowed) {
  const vec3 ray = ray_unpack(ray_p);
  float limit = FLT_MAX;
  const bool is_caustics_path = volume_id == VOLUME_TYPE_OCEAN && self_inst != HIT_TYPE_OCEAN;
  bool valid = (color_p.x != 0 || color_p.y != 0) && allowed;
  if (is_caustics_path && ray.y > 0.0f) {
    const float dist = (ocean_max_height(s) - origin.y) / ray.y;
    limit = (dist > 0.0f) ? dist : FLT_MAX;
  }
  else if (ray.y < 0.0f && s->ocean_active && origin.y > ocean_min_height(s)) valid = false; /* the sample would have to cross the water from above */
  RGBF vis = c_splat(0.0f);
  if (valid) { cnt[ORACLE_CNT_SHADOW]++; vis = trace_shadow(tr, origin, ray, limit, 0xFFFFFFFFu, 0, self_inst, self_tri); }
  RGBF light = c_mul(record_unpack(color_p), vis);
  light = c_mul(light, volume_transmittance(s, volume_id, origin, ray, limit));
  RGBF vis2 = c_splat(1.0f);
  if (valid && is_caustics_path && limit != FLT_MAX) {
    const vec3 ocean_pos = v_add(origin, v_scale(ray, limit));
    const vec3 ocean_normal = v_scale(ocean_get_normal(s, ocean_pos), -1.0f);
    const vec3 ocean_V = v_scale(ray, -1.0f);
    bool total_reflection;
    const vec3 refraction = refract_vector(ocean_V, ocean_normal, s->ocean_refractive_index, &total_reflection);
    const float fresnel_term = bsdf_fresnel(ocean_normal, ocean_V, refraction, s->ocean_refractive_index);
    light = c_scale(light, 1.0f - fresnel_term);
    light = c_mul(light, volume_transmittance(s, second_volume, ocean_pos, refraction, FLT_MAX));
    if (total_reflection) vis2 = c_splat(0.0f);
    else { cnt[ORACLE_CNT_SHADOW]++; vis2 = trace_shadow(tr, ocean_pos, refraction, FLT_MAX, 0xFFFFFFFFu, 0, 0xFFFFFFFFu, 0); }
  }
  if (!valid) vis2 = c_splat(1.0f);
  return c_mul(light, vis2);
}

/* ---- the sun seen from under water: direct_lighting_sun_caustic (direct_lighting.cuh:123-243) with caustics.cuh. A point on the water surface connects the
 * vertex to the sun: the point straight "below" the refracted sun direction (fast path), or one resampled from a patch of the surface around it ---- */
typedef struct { int kind; /* 0 surface, 1 volume, 2 particle */ vec3 position; uint16_t state; const void* ctx; const OLuts* luts; } SunCtx;
typedef struct { vec3 position, normal, V; uint16_t state; } ParticleCtx;
static float particle_phase(const OracleScene* s, const ParticleCtx* c, vec3 L);
static RGBF sunctx_evaluate(const OracleScene* s, const SunCtx* c, vec3 dir, float one_over_pdf) {
  if (c->kind == 0) { bool r; return bsdf_evaluate(c->luts, (const GeoCtx*) c->ctx, dir, HINT_GENERAL, &r, one_over_pdf); }
  if (c->kind == 1) return c_splat(volume_phase_evaluate(s, (const VolCtx*) c->ctx, dir) * one_over_pdf);
  return c_scale(c3(s->particles_albedo[0], s->particles_albedo[1], s->particles_albedo[2]), particle_phase(s, (const ParticleCtx*) c->ctx, dir) * one_over_pdf);
}
typedef struct { bool valid; vec3 base, edge1, edge2; float area, ior; bool fast_path; } CausticsDomain;
static CausticsDomain caustics_get_domain(const OracleScene* s, const OSky* sky, const SunCtx* c, vec3 L) { /* caustics.cuh:21-35, :60-123, under water */
  bool total_reflection;
  vec3 ray = refract_vector(L, v3(0.0f, 1.0f, 0.0f), 1.0f / s->ocean_refractive_index, &total_reflection);
  ray = v_scale(ray, -1.0f);
  const float dist = ocean_intersection_distance(s, c->position, ray, FLT_MAX);
  const vec3 center = v_add(c->position, v_scale(ray, dist));
  CausticsDomain d;
  d.valid = dist != FLT_MAX;
  d.ior = s->ocean_refractive_index;
  d.fast_path = (c->kind != 0) || s->ocean_amplitude == 0.0f || !s->ocean_caustics_active;
  if (d.fast_path) {
    d.base = center; d.edge1 = v3(0.0f, 0.0f, 0.0f); d.edge2 = v3(0.0f, 0.0f, 0.0f);
    d.area = sphere_solid_angle(sky->sun_pos, SKY_SUN_RADIUS, world_to_sky(sky, c->position));
    return d;
  }
  const vec3 center_dir = v_norm(v_sub(center, c->position));
  float altitude = o_asin(center_dir.y), azimuth = o_atan2(center_dir.z, center_dir.x); /* direction_to_angles, math.cuh:790-797 */
  if (azimuth < 0.0f) azimuth += 2.0f * O_PI;
  const float angle = 0.3f * s->ocean_caustics_domain_scale, plane_height = center.y;
  vec3 vd[3];
  const float alts[3] = {altitude - angle, altitude - angle, altitude + angle}, azis[3] = {azimuth - angle, azimuth + angle, azimuth - angle};
  for (int k = 0; k < 3; k++) { /* angles_to_direction, math.cuh:781-788 */
    float sa, ca, sz, cz; o_sincos(alts[k], &sa, &ca); o_sincos(azis[k], &sz, &cz);
    const vec3 dir = v3(cz * ca, sa, sz * ca);
    const float dd = fabsf(c->position.y - plane_height) / fmaxf(0.01f, fabsf(dir.y));
    vd[k] = v_add(c->position, v_scale(dir, dd));
  }
  d.base = vd[0]; d.edge1 = v_sub(vd[1], vd[0]); d.edge2 = v_sub(vd[2], vd[0]);
  d.area = v_len(v_cross(d.edge1, d.edge2));
  return d;
}
static bool caustics_find_connection_point(const OracleScene* s, const OSky* sky, const SunCtx* c, const Sampler* smp, uint32_t rt_initial, const CausticsDomain* d, uint32_t iteration,
                                           uint32_t num_iterations, vec3* point, float* sample_weight) { /* caustics.cuh:125-163, refraction */
  if (d->fast_path) { *point = d->base; *sample_weight = d->area; return true; }
  const float2_t r = rnd2(smp, rt_initial + iteration);
  const float sx = (iteration + r.x) * (1.0f / num_iterations), sy = r.y; /* ris_transform_stratum_2D, ris.cuh:166-174 */
  *point = v_add(d->base, v_add(v_scale(d->edge1, sx), v_scale(d->edge2, sy)));
  vec3 V = v_sub(c->position, *point);
  const float dist_sq = v_dot(V, V);
  V = v_scale(V, o_rsqrt(dist_sq));
  const vec3 normal = v_scale(ocean_get_normal_fast(s, *point), -1.0f);
  if (v_dot(V, normal) < 0.0f) return false;
  bool total_reflection;
  const vec3 L = refract_vector(V, normal, s->ocean_refractive_index, &total_reflection);
  if (!sphere_hit(L, world_to_sky(sky, *point), sky->sun_pos, SKY_SUN_RADIUS)) return false;
  *sample_weight = fabsf(V.y) * d->area / dist_sq;
  return true;
}
static bool sun_caustic_sample(const OracleScene* s, const OSky* sky, const SunCtx* c, const Sampler* smp, uint32_t set, uint32_t volume_type, uint32_t second_volume,
                               RGBF* light_out, vec3* dir_out) {
  const uint32_t rt_initial = 81u + 128u * set, rt_resampling = 338u + set, rt_sun_ray = 341u + set; /* CAUSTIC_INITIAL / _RESAMPLING / _SUN_RAY of LIGHT_SUN<set> */
  const vec3 sky_pos = world_to_sky(sky, c->position);
  float solid_angle;
  const vec3 sun_dir = sample_sphere(sky->sun_pos, SKY_SUN_RADIUS, sky_pos, rnd2(smp, rt_sun_ray), &solid_angle);
  const CausticsDomain domain = caustics_get_domain(s, sky, c, sun_dir);
  if (!domain.valid) return false;
  vec3 connection_point = v3(0.0f, 0.0f, 0.0f);
  float connection_weight;
  if (domain.fast_path) caustics_find_connection_point(s, sky, c, smp, rt_initial, &domain, 0, 1, &connection_point, &connection_weight);
  else {
    const uint32_t num_samples = s->ocean_caustics_ris_sample_count + 1;
    /* ris_stratified_reservoir (ris.cuh:176-259): strata are taken from both ends, the side whose weight sum lags behind the random split is extended */
    uint32_t iteration = 0, index_front = 0xFFFFFFFFu, index_back = num_samples;
    float sum_front = 0.0f, sum_back = 0.0f, selected_target = 0.0f;
    const float random = rnd1(smp, rt_resampling);
    const float mis_weight = 1.0f / num_samples;
    for (;;) {
      if (iteration > num_samples) break;
      const bool compute_front = sum_front <= random * (sum_front + sum_back);
      if (!compute_front && iteration == num_samples) break;
      iteration++;
      const uint32_t index = compute_front ? ++index_front : --index_back;
      if (index == num_samples) break;
      vec3 sample_point; float sample_weight = 0.0f;
      const bool valid_hit = caustics_find_connection_point(s, sky, c, smp, rt_initial, &domain, index, num_samples, &sample_point, &sample_weight);
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
    connection_weight *= s->ocean_refractive_index * s->ocean_refractive_index;
    connection_weight *= s->ocean_refractive_index * s->ocean_refractive_index * 2.0f;
  }
  if (connection_weight == 0.0f) return false;
  const vec3 pos_to_ocean = v_sub(connection_point, c->position);
  const float dist = v_len(pos_to_ocean);
  const vec3 dir = v_norm(pos_to_ocean);
  RGBF light = sky_sun_color(sky, world_to_sky(sky, connection_point), sun_dir);
  light = c_mul(light, sunctx_evaluate(s, c, dir, connection_weight));
  if (c_importance(light) == 0.0f) return false;
  light = c_mul(light, volume_transmittance(s, volume_type, c->position, dir, dist));
  light = c_mul(light, volume_transmittance(s, second_volume, connection_point, sun_dir, FLT_MAX));
  *light_out = light; *dir_out = dir;
  return true;
}

/* ---- fog: what the volume kernels do to one path at one depth (cuda/volume.cuh, optix/optix_kernel_shadow_volume.cu) ---- */
/* light_sample<MATERIAL_VOLUME> (light.cuh:84-159): the eight tree outputs are bridge candidates (light_evaluate_candidate<VOLUME>, :84-98) */
static BridgeSample volume_light_sample(const OracleScene* s, const VolCtx* c, const Sampler* smp) {
  LTQuery query = {NULL, c, RT_VOL_TREE_PREPASS, RT_VOL_TREE_POSTPASS, NULL};
  const LTWork work = light_tree_prepass(s, &query, smp);
  BridgeSample res;
  res.light_id = LIGHT_ID_INVALID; res.light_color = c_splat(0.0f); res.seed = 0; res.rotation.x = res.rotation.y = res.rotation.z = 0.0f; res.rotation.w = 1.0f; res.scale = 0.0f;
  RISReservoir rv = ris_init(rnd1(smp, RT_VOL_GEO_RESAMPLING));
  for (uint32_t out = 0; out < LIGHT_TREE_NUM_OUTPUTS; out++) {
    const LTResult o = light_tree_postpass(s, &query, smp, out, &work);
    if (o.light_id == LIGHT_ID_INVALID) continue;
    const uint32_t inst = s->light_tri_handles[2 * o.light_id], tri = s->light_tri_handles[2 * o.light_id + 1];
    uint32_t uvp[3];
    TriLight tl = light_triangle_init(s, inst, tri, uvp);
    float target, weight;
    const BridgeSample bs = bridges_sample(s, c, &tl, o.light_id, uvp, smp, out, &target, &weight);
    if (ris_add(&rv, target, weight * o.weight)) res = bs;
  }
  res.light_color = c_scale(res.light_color, ris_sampling_weight(&rv));
  return res;
}
/* bridges_sample_apply_shadowing (light_bridges.cuh:356-446): the path is rebuilt from its seed, rotated by the 16-bit quaternion of the task
 * and scaled; one visibility ray per segment (the segment that reaches the light leaves that light out) */
static RGBF bridges_apply_shadowing(const OracleScene* s, const OTracer* tr, const VolCtx* c, const BridgeSample* task, const Sampler* smp, uint64_t* cnt) {
  const uint32_t seed = task->seed;
  const uint32_t l_inst = s->light_tri_handles[2 * task->light_id], l_tri = s->light_tri_handles[2 * task->light_id + 1];
  uint32_t uvp[3];
  TriLight light = light_triangle_init(s, l_inst, l_tri, uvp);
  const vec3 point_on_light = light_triangle_sample_bridges(&light, rnd2(smp, RT_BRIDGE_LIGHT_POINT + seed));
  RGBF att; float ipdf;
  const vec3 initial_vertex = bridges_sample_initial_vertex(c, point_on_light, smp, seed, &att, &ipdf);
  vec3 light_dir; float area, light_dist;
  light_triangle_finalize_bridges(&light, uvp, initial_vertex, point_on_light, &light_dir, &light_dist, &area);
  const vec3 light_vector = v_scale(light_dir, light_dist);
  float vc_pdf;
  const uint32_t vertex_count = bridges_sample_vertex_count(s, &c->vol, v_len(light_vector), seed, smp, &vc_pdf);
  const Quat16 rotation = quaternion_pack16(task->rotation);
  const float scale = task->scale;
  vec3 current_vertex = initial_vertex;
  vec3 dir_sampled = v_norm(light_vector);
  vec3 dir = q16_apply(rotation, dir_sampled);
  float dist = -o_log(rnd1(smp, RT_BRIDGE_DISTANCE + seed * LIGHT_GEO_MAX_BRIDGE_LENGTH + 0)) * scale;
  cnt[ORACLE_CNT_SHADOW]++;
  RGBF shadow = trace_shadow(tr, current_vertex, dir, dist, l_inst, l_tri, 0xFFFFFFFFu, 0);
  for (uint32_t v = 1; v < BRIDGES_MAX_VERTEX_COUNT; v++) {
    if (v >= vertex_count) break; /* OPTIX_TRACE_STATUS_OPTIONAL_UNUSED: visibility 1 */
    current_vertex = v_add(current_vertex, v_scale(dir, dist));
    dir_sampled = bridges_phase_sample(dir_sampled, rnd2(smp, RT_BRIDGE_PHASE + seed * LIGHT_GEO_MAX_BRIDGE_LENGTH + v));
    dir = q16_apply(rotation, dir_sampled);
    dist = -o_log(rnd1(smp, RT_BRIDGE_DISTANCE + seed * LIGHT_GEO_MAX_BRIDGE_LENGTH + v)) * scale;
    cnt[ORACLE_CNT_SHADOW]++;
    shadow = c_mul(shadow, trace_shadow(tr, current_vertex, dir, dist, l_inst, l_tri, 0xFFFFFFFFu, 0));
  }
  return c_mul(task->light_color, shadow);
}
/* sky_color_no_compute (sky.cuh:534-565) */
static RGBF sky_color_no_compute(const OracleScene* s, vec3 origin, vec3 ray, uint16_t state) {
  if (s->sky_mode == SKY_MODE_HDRI) return sky_hdri_color(s, origin, ray, state);
  if (s->sky_mode == SKY_MODE_CONSTANT_COLOR) return c3(s->sky_constant_color[0], s->sky_constant_color[1], s->sky_constant_color[2]);
  return c_splat(0.0f);
}
/* volume_process_inscattering (volume.cuh:31-98) + the shadow pass over its three tasks (optix_kernel_shadow_volume.cu:13-98): light that the
 * fog scatters into the ray between its origin and its end point (`depth_t`: the hit distance, FLT_MAX for a ray that left the scene) */
static RGBF volume_inscattering(const OracleScene* s, const OTracer* tr, const Sampler* smp, vec3 origin, vec3 ray, uint16_t state, float depth_t, bool lights_present,
                                uint32_t volume_type, uint32_t second_volume, uint64_t* cnt) {
  VolCtx ctx = volume_context(s, volume_type, origin, ray, state, depth_t);
  RGBF acc = c_splat(0.0f);
  const bool bridges_allowed = lights_present && (state & ST_DELTA_PATH) != 0 && (state & ST_VOLUME_SCATTERED) == 0 &&
                               (volume_type != VOLUME_TYPE_OCEAN || s->ocean_triangle_light_contribution); /* direct_lighting.cuh:296-306 */
  if (bridges_allowed) {
    const BridgeSample bs = volume_light_sample(s, &ctx, smp);
    if (bs.light_id != LIGHT_ID_INVALID && bs.seed != 0xFFFFFFFFu) acc = c_add(acc, bridges_apply_shadowing(s, tr, &ctx, &bs, smp, cnt));
  }
  const RGBF w = volume_sky_initial_vertex(&ctx, smp); /* the vertex the sun and the ambient sample start from */
  const bool sun_allowed = s->sky_mode != SKY_MODE_CONSTANT_COLOR && s->sky_lut_transmittance && s->sky_lut_multiscattering;
  if (sun_allowed) {
    const OSky sky_v = osky_view(s);
    RGBF lc; vec3 dir;
    uint2_t sun_color = {0, 0}, sun_ray = {0, 0};
    bool have;
    if (volume_type == VOLUME_TYPE_OCEAN) { /* direct_lighting.cuh:370-380: under water the sun arrives through the surface */
      const SunCtx sc = {1, ctx.position, state, &ctx, NULL};
      have = sun_caustic_sample(s, &sky_v, &sc, smp, 1, volume_type, second_volume, &lc, &dir);
    }
    else have = volume_sun_sample(s, &sky_v, &ctx, smp, &lc, &dir);
    if (have) { sun_color = record_pack(lc); sun_ray = ray_pack(dir); }
    acc = c_add(acc, c_mul(sun_evaluate(s, tr, cnt, ctx.position, 0xFFFFFFFFu, 0, volume_type, sun_color, sun_ray, true), w));
  }
  const vec3 bounce = volume_bsdf_sample(s, &ctx, smp, RT_VOL_AMBIENT_RESAMPLING, RT_VOL_AMBIENT_DIFFUSE);
  if (s->sky_mode != SKY_MODE_DEFAULT) { /* direct_lighting.cuh:385-403, :521-584 */
    const uint2_t amb_color = record_pack(c_mul(sky_color_no_compute(s, ctx.position, bounce, 0), c_splat(1.0f)));
    const uint2_t amb_ray = ray_pack(bounce);
    acc = c_add(acc, c_mul(ambient_evaluate(s, tr, cnt, ctx.position, 0xFFFFFFFFu, 0, volume_type, second_volume, amb_color, amb_ray, true), w));
  }
  return acc;
}


/* ---- particles (optix_kernel_raytrace.cu:97-131, cuda/particle.cuh:7-108, particle_utils.cuh) ---- */
#define HIT_TYPE_PARTICLE_MIN 0x80000000u
#define HIT_TYPE_PARTICLE_MAX 0xEFFFFFFFu
#define RT_CAMERA_TIME_TARGET 65u
static inline bool particle_is_hit(uint32_t instance_id) { return instance_id <= HIT_TYPE_PARTICLE_MAX && instance_id >= HIT_TYPE_PARTICLE_MIN; }
/* the particle pass of the closest-hit kernel: only delta paths see particles; a hit closer than the surface replaces it */
static void particles_trace(const OracleScene* s, const OTracer* tr, const Sampler* smp, vec3 origin, vec3 ray, uint16_t state, OHit* hit) {
  if (!s->particles_active || (state & ST_DELTA_PATH) == 0) return;
  Sampler first = *smp;
  first.depth = 0; /* random_1D_consistent, random.cuh:356-361 */
  const float time = rnd1(&first, RT_CAMERA_TIME_TARGET);
  const vec3 motion_offset = v_scale(v3(s->particles_direction[0], s->particles_direction[1], s->particles_direction[2]), time * s->particles_speed);
  const vec3 scaled_ray = v_scale(ray, 1.0f / s->particles_scale);
  vec3 pos = v_scale(v_add(origin, motion_offset), 1.0f / s->particles_scale);
  pos.x = pos.x - floorf(pos.x); pos.y = pos.y - floorf(pos.y); pos.z = pos.z - floorf(pos.z);
  float t;
  const uint32_t tri = trace_particles(tr, pos, scaled_ray, hit->t, &t);
  if (tri != 0xFFFFFFFFu) { hit->instance_id = HIT_TYPE_PARTICLE_MIN + (tri >> 1); hit->tri_id = 0; hit->t = t; }
}
static float particle_phase(const OracleScene* s, const ParticleCtx* c, vec3 L) { return je_phase_function(s->particles_phase, -v_dot(c->V, L)); }
/* light_sample<MATERIAL_PARTICLE> (light.cuh:49-82, :100-159): BSDF value = albedo x phase function, MIS weight 1 (mis.cuh:41-47) */
static LightSample particle_light_sample(const OracleScene* s, const ParticleCtx* c, const Sampler* smp) {
  LTQuery query = {NULL, NULL, RT_LIGHT_GEO_TREE_PREPASS, RT_LIGHT_GEO_TREE_POSTPASS, &c->position};
  const LTWork work = light_tree_prepass(s, &query, smp);
  const RGBF albedo = c3(s->particles_albedo[0], s->particles_albedo[1], s->particles_albedo[2]);
  LightSample res;
  res.light_id = LIGHT_ID_INVALID; res.ray = v3(0.0f, 0.0f, 0.0f); res.light_color = c_splat(0.0f); res.dist = 0.0f; res.root_sum = 0.0f;
  RISReservoir rv = ris_init(rnd1(smp, RT_LIGHT_GEO_RESAMPLING));
  for (uint32_t out = 0; out < LIGHT_TREE_NUM_OUTPUTS; out++) {
    const LTResult o = light_tree_postpass(s, &query, smp, out, &work);
    if (o.light_id == LIGHT_ID_INVALID) continue;
    const uint32_t inst = s->light_tri_handles[2 * o.light_id], tri = s->light_tri_handles[2 * o.light_id + 1];
    uint32_t uvp[3];
    TriLight tl = light_triangle_init(s, inst, tri, uvp);
    vec3 ray; float dist, sa;
    if (!light_triangle_finalize(&tl, uvp, c->position, rnd2(smp, RT_LIGHT_GEO_RAY + out), &ray, &dist, &sa)) continue;
    RGBF lc = light_get_color(s, &tl);
    const RGBF bw = c_scale(albedo, particle_phase(s, c, ray) * 1.0f);
    lc = c_scale(c_mul(lc, bw), 1.0f);
    if (ris_add(&rv, c_importance(lc), o.weight * sa)) { res.light_id = o.light_id; res.ray = ray; res.light_color = lc; res.dist = dist; }
  }
  res.light_color = c_scale(res.light_color, ris_sampling_weight(&rv));
  return res;
}
/* direct_lighting_sun_create_task / _direct for a particle (direct_lighting.cuh:20-121, :352-383; random set LIGHT_SUN<0>) */
static bool particle_sun_sample(const OracleScene* s, const OSky* sky, const ParticleCtx* c, uint32_t volume_type, const Sampler* smp, RGBF* light_out, vec3* dir_out) {
  const vec3 sky_pos = world_to_sky(sky, c->position);
  const bool sun_below_horizon = sph_hit_p0(v_norm(v_sub(sky->sun_pos, sky_pos)), sky_pos, SKY_EARTH_RADIUS);
  const bool inside_earth = v_len(sky_pos) < SKY_EARTH_RADIUS;
  if (sun_below_horizon || inside_earth) return false;
  const RGBF albedo = c3(s->particles_albedo[0], s->particles_albedo[1], s->particles_albedo[2]);
  const float2_t random_dir = rnd2(smp, RT_SUN_BSDF);
  const float random_method = rnd1(smp, RT_SUN_BSDF_METHOD);
  const vec3 dir_bsdf = je_phase_sample(s->particles_phase, v_scale(c->V, -1.0f), random_dir, random_method);
  RGBF light_bsdf = c_splat(0.0f);
  if (sphere_hit(dir_bsdf, sky_pos, sky->sun_pos, SKY_SUN_RADIUS)) light_bsdf = c_mul(sky_sun_color(sky, sky_pos, dir_bsdf), c_scale(albedo, particle_phase(s, c, dir_bsdf) * 1.0f));
  float solid_angle;
  const vec3 dir_sa = sample_sphere(sky->sun_pos, SKY_SUN_RADIUS, sky_pos, rnd2(smp, RT_SUN_RAY), &solid_angle);
  const RGBF light_sa = c_mul(sky_sun_color(sky, sky_pos, dir_sa), c_scale(albedo, particle_phase(s, c, dir_sa) * 1.0f));
  const float target_bsdf = c_importance(light_bsdf), target_sa = c_importance(light_sa);
  const float mis_bsdf = solid_angle / (particle_phase(s, c, dir_bsdf) * solid_angle + 1.0f);
  const float mis_sa = solid_angle / (particle_phase(s, c, dir_sa) * solid_angle + 1.0f);
  const float weight_bsdf = target_bsdf * mis_bsdf, weight_sa = target_sa * mis_sa;
  const float sum_weights = weight_bsdf + weight_sa;
  if (sum_weights == 0.0f) return false;
  float target;
  RGBF light;
  if (rnd1(smp, RT_SUN_RESAMPLING) * sum_weights < weight_bsdf) { *dir_out = dir_bsdf; target = target_bsdf; light = light_bsdf; }
  else { *dir_out = dir_sa; target = target_sa; light = light_sa; }
  light = c_scale(light, sum_weights / target);
  if (target == 0.0f) return false;
  if (c_importance(light) == 0.0f) return false;
  *light_out = c_mul(light, volume_transmittance(s, volume_type, c->position, *dir_out, FLT_MAX));
  return true;
}


/* direct_lighting_bsdf_evaluate_task (direct_lighting.cuh:586-667): the BSDF-sampled direction against the light-only BVH, then a visibility ray */
static RGBF bsdf_light_evaluate(const OracleScene* s, const OTracer* tr, uint64_t* cnt, const Sampler* smp, vec3 hit_origin, uint32_t self_inst, uint32_t self_tri,
                                const LightBSDFSample* lb, float root_sum, uint32_t volume_id, bool allowed) {
  bool valid = allowed && lb->sampling_probability != 0.0f;
  uint32_t light_id = LIGHT_ID_INVALID, num_hits = 0;
  if (valid) {
    cnt[ORACLE_CNT_LIGHT_BVH]++;
    light_id = trace_light_bvh(tr, hit_origin, lb->ray, self_inst, self_tri, rnd1(smp, RT_LIGHT_BSDF_TRACE), &num_hits);
  }
  valid = valid && light_id != LIGHT_ID_INVALID;
  float dist = FLT_MAX;
  uint32_t lh_inst = 0xFFFFFFFFu, lh_tri = 0;
  RGBF lc = c_splat(0.0f);
  if (light_id != LIGHT_ID_INVALID) {
    lh_inst = s->light_tri_handles[2 * light_id]; lh_tri = s->light_tri_handles[2 * light_id + 1];
    uint32_t uvp[3];
    TriLight tl = light_triangle_init(s, lh_inst, lh_tri, uvp);
    if (light_triangle_finalize_dist(&tl, uvp, hit_origin, lb->ray, &dist)) {
      lc = light_get_color(s, &tl);
      const float mis = mis_weight_gi(hit_origin, &tl, lc, dist, lb->sampling_probability, root_sum);
      lc = c_scale(lc, mis * num_hits);
      lc = c_mul(lc, lb->weight);
    }
    else valid = false;
  }
  RGBF vis = c_splat(0.0f);
  if (valid) { cnt[ORACLE_CNT_SHADOW]++; vis = trace_shadow(tr, hit_origin, lb->ray, dist, lh_inst, lh_tri, self_inst, self_tri); }
  lc = c_mul(lc, vis);
  return c_mul(lc, volume_transmittance(s, volume_id, hit_origin, lb->ray, dist)); /* direct_lighting.cuh:661-666 */
}
/* ocean_get_context (ocean_utils.cuh:477-517): the water surface as a smooth translucent material */
static GeoCtx ocean_get_context(const OracleScene* s, vec3 position, vec3 ray, uint16_t state, uint32_t medium) {
  vec3 normal = ocean_get_normal(s, position);
  const bool inside_water = v_dot(ray, normal) > 0.0f;
  if (inside_water) normal = v_scale(normal, -1.0f);
  uint32_t flags = MAT_TRANSLUCENT;
  if (inside_water) flags |= MAT_REFRACTION_IS_INSIDE;
  const float other_ior = medium_ior_peek(medium, inside_water);
  const float ior_ratio = inside_water ? s->ocean_refractive_index / other_ior : other_ior / s->ocean_refractive_index;
  const float roughness = (state & ST_DELTA_PATH) ? 0.02f * 2.0f : 0.25f; /* BSDF_ROUGHNESS_CLAMP * 2 */
  GeoCtx g;
  g.instance_id = HIT_TYPE_OCEAN; g.tri_id = 0;
  g.normal = normal;
  g.face_normal = normal_pack(normal);
  g.position = position;
  g.V = v_scale(ray, -1.0f);
  g.state = state;
  g.params.data[0] = g.params.data[1] = g.params.data[2] = 0;
  g.params.flags = flags;
  mp_set_albedo(&g.params, c_splat(1.0f));
  mp_set_opacity(&g.params, 1.0f);
  mp_set_roughness(&g.params, roughness);
  mp_set_emission(&g.params, c_splat(0.0f));
  mp_set_ior(&g.params, ior_ratio);
  return g;
}

/* volume_process_events (volume.cuh:100-229): closed-form distance sampling. A path that scatters before its hit becomes a volume hit, the throughput
 * takes transmittance over sampling density; in the non-procedural sky modes a ray that left the scene adds the sky here and ends (sky fast path). */
static void volume_events(const OracleScene* s, const Sampler* smp_p, vec3 origin, vec3 ray, uint16_t state, uint32_t volume_type, OHit* hit_p, uint2_t* record_pp, RGBF* result_p) {
  const Sampler smp = *smp_p;
  OHit hit = *hit_p;
  uint2_t record_p = *record_pp;
  RGBF result = *result_p;
  {
      const OVolume vol = volume_descriptor(s, volume_type);
      OVolumePath path = volume_compute_path(s, &vol, origin, ray, hit.t, true);
      RGBF record = record_unpack(record_p);
      const bool sky_fast_path = hit.instance_id == HIT_TYPE_SKY && s->sky_mode != SKY_MODE_DEFAULT && (state & ST_ALLOW_AMBIENT) != 0;
      if (sky_fast_path) {
        RGBF sky = c_mul(sky_color_no_compute(s, origin, ray, state), record);
        sky = c_mul(sky, volume_transmittance_length(&vol, path.length));
        beauty_add(&result, sky);
        hit.instance_id = HIT_TYPE_INVALID;
      }
      float intersection_probability = ((state & ST_DELTA_PATH) && !particle_is_hit(hit.instance_id)) ? 0.5f : 1.0f; /* bounds the variance of highlights seen through the volume */
      if (volume_type == VOLUME_TYPE_OCEAN && !s->ocean_multiscattering && (state & ST_DELTA_PATH) == 0) intersection_probability = 0.0f; /* single scattering in the water */
      const float2_t randoms = rnd2(&smp, RT_VOLUME_INTERSECTION);
      bool sampled = false;
      float pdf = 1.0f;
      if (randoms.y < intersection_probability) {
        const float volume_dist = volume_sample_intersection(&vol, path.start, path.length, randoms.x);
        if (volume_dist < hit.t) {
          const float sample_pdf = volume_sample_intersection_pdf(&vol, path.start, volume_dist);
          hit.t = volume_dist; hit.instance_id = HIT_TYPE_VOLUME_BASE | volume_type; hit.tri_id = 0;
          record = c_mul(record, vol.scat);
          pdf *= intersection_probability;
          pdf *= sample_pdf;
          sampled = true;
          path.length = hit.t - path.start;
        }
      }
      if (!sampled && !sky_fast_path) pdf *= (1.0f - intersection_probability) + intersection_probability * volume_miss_probability(&vol, path.length);
      record = c_mul(record, volume_transmittance_length(&vol, path.length));
      record = c_scale(record, 1.0f / pdf);
      record_p = record_pack(record);
  }
  *hit_p = hit; *record_pp = record_p; *result_p = result;
}

static RGBF render_path(const OracleScene* s, const OTracer* tr, uint32_t px, uint32_t py, uint32_t sample_id, uint64_t* cnt) {
  if (s->shading_mode != 0) return render_path_debug(s, tr, px, py, sample_id, cnt);
  const OLuts luts = scene_luts(s);
  const bool lights_present = s->light_tree_root != NULL && s->num_lights > 0;
  Sampler smp = {s->bluenoise_2d, px, py, sample_id, 0};
  RGBF result = c_splat(0.0f);
  vec3 origin, ray;
  camera_sample(s, &smp, &origin, &ray);
  uint16_t state = ST_DELTA_PATH | ST_CAMERA_DIRECTION | ST_ALLOW_EMISSION | ST_ALLOW_AMBIENT;
  uint2_t record_p = record_pack(c_splat(1.0f));
  /* kernels.cuh:146, :172-186: the medium the camera is in (bsdf_refraction_index_ambient, bsdf_utils.cuh:128-133) and the volumes around it */
  const float ambient_ior = (s->ocean_active && origin.y < s->ocean_height) ? s->ocean_refractive_index : 1.0f;
  uint32_t medium = medium_ior_modify(0, ambient_ior, true);
  uint32_t volumes = 0;
  if (s->fog_active) volumes = volume_stack_modify(volumes, VOLUME_TYPE_FOG, true);
  if (s->ocean_active && ocean_is_underwater(s, origin)) volumes = volume_stack_modify(volumes, VOLUME_TYPE_OCEAN, true);
  const bool render_volumes = s->fog_active || s->ocean_active; /* device_manager.c:478 */
  uint32_t ign_inst = 0, ign_tri = 0;
  const RGBF sky_color = (s->sky_mode == SKY_MODE_CONSTANT_COLOR) ? c3(s->sky_constant_color[0], s->sky_constant_color[1], s->sky_constant_color[2]) : c_splat(0.0f);

  for (uint32_t depth = 0; depth <= s->max_ray_depth; depth++) {
    smp.depth = (depth == s->max_ray_depth && depth > 0) ? depth - 1 : depth;
    OHit hit = trace_closest(tr, origin, ray, (state & ST_USE_IGNORE_HANDLE) != 0, ign_inst, ign_tri);
    cnt[ORACLE_CNT_TRACE]++;
    particles_trace(s, tr, &smp, origin, ray, state, &hit);
    if (s->ocean_active) { /* optix_raytrace_ocean, optix_kernel_raytrace.cu:134-144 */
      const float ocean_depth = ocean_intersection_distance(s, origin, ray, hit.t);
      if (ocean_depth < hit.t) { hit.t = ocean_depth; hit.instance_id = HIT_TYPE_OCEAN; hit.tri_id = 0; }
    }
    const uint32_t top_volume = volume_stack_peek(volumes, false), second_volume = volume_stack_peek(volumes, true);
    if (render_volumes && top_volume != VOLUME_TYPE_NONE) {
      /* device_renderer.c:64-76: in-scattering and its shadow pass, then the distance sampling (volume_process_events, volume.cuh:100-229) */
      const RGBF in = volume_inscattering(s, tr, &smp, origin, ray, state, hit.t, lights_present, top_volume, second_volume, cnt);
      beauty_add(&result, c_mul(in, record_unpack(record_p)));
      volume_events(s, &smp, origin, ray, state, top_volume, &hit, &record_p, &result);
    }
    if (s->cloud_active && s->sky_mode == SKY_MODE_DEFAULT && s->sky_lut_transmittance && s->sky_lut_multiscattering && s->cloud_noise_shape && s->cloud_noise_detail &&
        s->cloud_noise_weather) {
      /* cloud_process_tasks (cloud.cuh:340-384; device_renderer.c:78-82, device_manager.c:474): what the cloud layers scatter into the ray and take from
       * it; with atmosphere_scattering the ray's origin moves up to the last layer it entered (the air up to there was marched here) */
      const OSky view = osky_view(s);
      RGBF record = record_unpack(record_p);
      RGBF color = c_splat(0.0f);
      float cloud_transmittance = 1.0f;
      const float cloud_offset = clouds_render(s, &view, &smp, world_to_sky(&view, origin), ray, hit.t * 0.001f, &color, &record, &cloud_transmittance);
      if (s->cloud_atmosphere_scattering && cloud_offset != FLT_MAX && cloud_offset > 0.0f) {
        const float cloud_world_offset = cloud_offset * 1000.0f;
        origin = v_add(origin, v_scale(ray, cloud_world_offset));
        if (hit.t != FLT_MAX) hit.t -= cloud_world_offset;
      }
      record_p = record_pack(record);
      beauty_add(&result, color);
    }
    if (hit.instance_id == HIT_TYPE_SKY) {
      if (state & ST_ALLOW_AMBIENT) {
        RGBF sky = sky_color;
        if (s->sky_mode == SKY_MODE_DEFAULT && s->sky_lut_transmittance && s->sky_lut_multiscattering) { /* sky_color_main, sky.cuh:567-577 */
          const OSky view = osky_view(s);
          const bool include_sun = (state & (ST_CAMERA_DIRECTION | ST_ALLOW_EMISSION)) != 0;
          sky = sky_get_color(&view, world_to_sky(&view, origin), ray, FLT_MAX, include_sun, (int) view.steps, rnd1(&smp, RANDOM_TARGET_SKY_STEP_OFFSET));
        }
        else if (s->sky_mode == SKY_MODE_HDRI) sky = sky_hdri_color(s, origin, ray, state); /* sky.cuh:579-595 */
        beauty_add(&result, c_mul(sky, record_unpack(record_p)));
      }
      break;
    }
    if (s->sky_aerial_perspective && s->sky_mode != SKY_MODE_CONSTANT_COLOR && s->sky_lut_transmittance && s->sky_lut_multiscattering) {
      /* sky_process_inscattering_events, cuda/kernels.cuh:357-388 (device_manager.c:475): before the hit is shaded */
      const OSky view = osky_view(s);
      RGBF record = record_unpack(record_p);
      const RGBF c = sky_trace_inscattering(&view, world_to_sky(&view, origin), ray, hit.t * 0.001f, &record, smp.depth == 0, rnd1(&smp, RANDOM_TARGET_SKY_INSCATTERING_STEP),
                                            rnd1(&smp, RANDOM_TARGET_SKY_STEP_OFFSET));
      beauty_add(&result, c);
      record_p = record_pack(record);
    }
    if (hit.instance_id == HIT_TYPE_INVALID) break; /* the sky fast path of the volume events ended the path (no task type counts it) */
    if (hit.instance_id >= HIT_TYPE_VOLUME_BASE && hit.instance_id <= HIT_TYPE_VOLUME_MAX) { /* volume_process_tasks (volume.cuh:231-288); not queued at the last depth (device_renderer.c:114) */
      if (depth == s->max_ray_depth) break;
      origin = v_add(origin, v_scale(ray, hit.t));
      const VolCtx vctx = volume_context(s, top_volume, origin, ray, state, 0.0f);
      ray = volume_bsdf_sample(s, &vctx, &smp, RT_VOL_GI_RESAMPLING, RT_VOL_GI_DIFFUSE);
      state &= ~(ST_DELTA_PATH | ST_CAMERA_DIRECTION | ST_ALLOW_EMISSION | ST_USE_IGNORE_HANDLE);
      if (s->sky_mode != SKY_MODE_DEFAULT) state &= ~ST_ALLOW_AMBIENT; else state |= ST_ALLOW_AMBIENT;
      state |= ST_VOLUME_SCATTERED;
      continue;
    }
    if (particle_is_hit(hit.instance_id)) { /* particle_process_tasks (particle.cuh:7-108) and its share of the shadow pass (optix_kernel_shadow.cu) */
      ParticleCtx pc;
      pc.position = v_add(origin, v_scale(ray, hit.t));
      const float* qn = s->particle_normals + 4 * (size_t) (hit.instance_id & HIT_TYPE_PARTICLE_MASK);
      const vec3 quad_normal = v3(qn[0], qn[1], qn[2]);
      pc.normal = (v_dot(ray, quad_normal) < 0.0f) ? quad_normal : v_scale(quad_normal, -1.0f);
      pc.V = v_scale(ray, -1.0f);
      pc.state = state;
      const RGBF albedo = c3(s->particles_albedo[0], s->particles_albedo[1], s->particles_albedo[2]);
      const bool p_geo_allowed = lights_present && ((state & ST_VOLUME_SCATTERED) == 0);
      LightSample ls; ls.light_id = LIGHT_ID_INVALID; ls.light_color = c_splat(0.0f); ls.ray = v3(0, 0, 0); ls.dist = 0.0f;
      if (p_geo_allowed) {
        ls = particle_light_sample(s, &pc, &smp);
        ls.light_color = c_mul(ls.light_color, volume_transmittance(s, top_volume, pc.position, ls.ray, ls.dist));
      }
      const bool p_sun_allowed = s->sky_mode != SKY_MODE_CONSTANT_COLOR && s->sky_lut_transmittance && s->sky_lut_multiscattering;
      uint2_t sun_color = {0, 0}, sun_ray = {0, 0};
      if (p_sun_allowed) {
        const OSky sky_v = osky_view(s);
        RGBF lc; vec3 dir;
        bool have;
        if (top_volume == VOLUME_TYPE_OCEAN) { const SunCtx sc = {2, pc.position, state, &pc, NULL}; have = sun_caustic_sample(s, &sky_v, &sc, &smp, 0, top_volume, second_volume, &lc, &dir); }
        else have = particle_sun_sample(s, &sky_v, &pc, top_volume, &smp, &lc, &dir);
        if (have) { sun_color = record_pack(lc); sun_ray = ray_pack(dir); }
      }
      /* bsdf_sample<MATERIAL_PARTICLE> with RANDOM_GI (bsdf.cuh:320-331): weight = albedo */
      const float random_choice = rnd1(&smp, RT_BSDF_RESAMPLING);
      const float2_t random_dir = rnd2(&smp, RT_BSDF_DIFFUSE);
      const vec3 bounce = je_phase_sample(s->particles_phase, ray, random_dir, random_choice);
      const bool p_ambient_allowed = s->sky_mode != SKY_MODE_DEFAULT;
      uint2_t amb_color = {0, 0}, amb_ray = {0, 0};
      if (p_ambient_allowed) { amb_color = record_pack(c_mul(sky_color_no_compute(s, pc.position, bounce, 0), albedo)); amb_ray = ray_pack(bounce); }
      const RGBF record_in = record_unpack(record_p);
      {
        RGBF acc = c_splat(0.0f);
        if (ls.light_id != LIGHT_ID_INVALID && p_geo_allowed) {
          cnt[ORACLE_CNT_SHADOW]++;
          const RGBF vis = trace_shadow(tr, pc.position, ls.ray, ls.dist, s->light_tri_handles[2 * ls.light_id], s->light_tri_handles[2 * ls.light_id + 1], hit.instance_id, 0);
          acc = c_add(acc, c_mul(ls.light_color, vis));
        }
        if (p_sun_allowed) acc = c_add(acc, sun_evaluate(s, tr, cnt, pc.position, hit.instance_id, 0, top_volume, sun_color, sun_ray, true));
        if (p_ambient_allowed) acc = c_add(acc, ambient_evaluate(s, tr, cnt, pc.position, hit.instance_id, 0, top_volume, second_volume, amb_color, amb_ray, true));
        beauty_add(&result, c_mul(acc, record_in));
      }
      uint16_t new_state = state & ~(ST_DELTA_PATH | ST_CAMERA_DIRECTION | ST_ALLOW_EMISSION | ST_USE_IGNORE_HANDLE);
      if (s->sky_mode != SKY_MODE_DEFAULT) new_state &= ~ST_ALLOW_AMBIENT; else new_state |= ST_ALLOW_AMBIENT;
      RGBF record = c_mul(record_in, albedo);
      if (!russian_roulette(s, &smp, state, &record)) break;
      record_p = record_pack(record);
      state = new_state;
      origin = pc.position;
      ray = bounce;
      continue;
    }
    if (hit.instance_id == HIT_TYPE_OCEAN) { /* ocean_process_tasks (ocean.cuh:12-102) and its share of the shadow pass */
      const vec3 position = v_add(origin, v_scale(ray, hit.t));
      const GeoCtx g = ocean_get_context(s, position, ray, state, medium);
      const bool o_bsdf_allowed = lights_present && ((state & ST_VOLUME_SCATTERED) == 0);
      LightBSDFSample lb; lb.sampling_probability = 0.0f; lb.weight = c_splat(0.0f); lb.ray = v3(0, 0, 1);
      if (o_bsdf_allowed) lb = light_bsdf_get_sample(&luts, &g, &smp);
      const bool o_sun_allowed = s->sky_mode != SKY_MODE_CONSTANT_COLOR && s->sky_lut_transmittance && s->sky_lut_multiscattering;
      uint2_t sun_color = {0, 0}, sun_ray = {0, 0};
      if (o_sun_allowed) {
        const OSky sky_v = osky_view(s);
        RGBF lc; vec3 dir;
        if (sun_sample(&sky_v, &luts, &g, &smp, &lc, &dir)) {
          lc = c_mul(lc, volume_transmittance(s, top_volume, g.position, dir, FLT_MAX));
          sun_color = record_pack(lc); sun_ray = ray_pack(dir);
        }
      }
      const BSDFSample bounce = bsdf_sample(&luts, &g, &smp, 0);
      const RGBF record_in = record_unpack(record_p);
      RGBF record = c_mul(record_in, bounce.weight);
      const float shift_length = 8.0f * O_EPS * (1.0f + s->ocean_amplitude) * (1.0f + fabsf(s->ocean_height)); /* ocean_shift_vector, ocean_utils.cuh:519-523 */
      const vec3 bounce_pos = v_add(g.position, v_scale(g.normal, bounce.is_transparent_pass ? -shift_length : shift_length));
      {
        RGBF acc = c_splat(0.0f);
        acc = c_add(acc, bsdf_light_evaluate(s, tr, cnt, &smp, position, HIT_TYPE_OCEAN, 0, &lb, 0.0f, top_volume, o_bsdf_allowed));
        if (o_sun_allowed) acc = c_add(acc, sun_evaluate(s, tr, cnt, position, HIT_TYPE_OCEAN, 0, top_volume, sun_color, sun_ray, true));
        beauty_add(&result, c_mul(acc, record_in));
      }
      const uint16_t new_state = state & ~(ST_CAMERA_DIRECTION | ST_ALLOW_EMISSION | ST_USE_IGNORE_HANDLE);
      if (!russian_roulette(s, &smp, state, &record)) break;
      record_p = record_pack(record);
      if (bounce.is_transparent_pass) {
        const bool refr_inside = (g.params.flags & MAT_REFRACTION_IS_INSIDE) != 0;
        medium = medium_ior_modify(medium, s->ocean_refractive_index, !refr_inside);
        volumes = volume_stack_modify(volumes, VOLUME_TYPE_OCEAN, !refr_inside);
      }
      state = new_state;
      origin = bounce_pos;
      ray = bounce.ray;
      continue;
    }
    cnt[ORACLE_CNT_VERTICES]++;
    const vec3 hit_origin = v_add(origin, v_scale(ray, hit.t));
    const GeoCtx g = geometry_get_context(s, hit_origin, ray, state, hit.instance_id, hit.tri_id, medium);

    /* ---- NEE task creation (geometry.cuh:31-74) ---- */
    float root_sum = 0.0f;
    LightSample ls; ls.light_id = LIGHT_ID_INVALID; ls.light_color = c_splat(0.0f); ls.ray = v3(0, 0, 0); ls.dist = 0.0f;
    const bool geo_allowed = lights_present && ((state & ST_VOLUME_SCATTERED) == 0);
    if (geo_allowed) {
      ls = light_sample(s, &g, &smp); root_sum = ls.root_sum;
      ls.light_color = c_mul(ls.light_color, volume_transmittance(s, top_volume, g.position, ls.ray, ls.dist)); /* direct_lighting.cuh:329-337 */
    }
    LightBSDFSample lb; lb.sampling_probability = 0.0f; lb.weight = c_splat(0.0f); lb.ray = v3(0, 0, 1);
    const bool bsdf_allowed = geo_allowed;
    if (bsdf_allowed) lb = light_bsdf_get_sample(&luts, &g, &smp);
    const BSDFSample bounce = bsdf_sample(&luts, &g, &smp, 0);
    const bool ambient_allowed = s->sky_mode != SKY_MODE_DEFAULT;
    /* sun (direct_lighting.cuh:352-383): allowed outside constant-colour mode (:257-263); needs the procedural sky's tables */
    const bool sun_allowed = s->sky_mode != SKY_MODE_CONSTANT_COLOR && s->sky_lut_transmittance && s->sky_lut_multiscattering;
    uint2_t sun_color = {0, 0}, sun_ray = {0, 0};
    if (sun_allowed) {
      const OSky sky_v = osky_view(s);
      RGBF lc; vec3 dir;
      if (top_volume == VOLUME_TYPE_OCEAN) { /* direct_lighting.cuh:368-380: a surface under water receives the sun through the water surface */
        const SunCtx sc = {0, g.position, state, &g, &luts};
        if (sun_caustic_sample(s, &sky_v, &sc, &smp, 0, top_volume, second_volume, &lc, &dir)) { sun_color = record_pack(lc); sun_ray = ray_pack(dir); }
      }
      else if (sun_sample(&sky_v, &luts, &g, &smp, &lc, &dir)) {
        lc = c_mul(lc, volume_transmittance(s, top_volume, g.position, dir, FLT_MAX)); /* direct_lighting.cuh:104-108 */
        sun_color = record_pack(lc); sun_ray = ray_pack(dir);
      }
    }
    uint2_t amb_color = {0, 0}, amb_ray = {0, 0};
    if (ambient_allowed) { /* direct_lighting.cuh:385-403 */
      const RGBF ambient = (s->sky_mode == SKY_MODE_HDRI) ? sky_hdri_color(s, g.position, bounce.ray, 0) : sky_color; /* sky_color_no_compute(.., 0) */
      amb_color = record_pack(c_mul(ambient, bounce.weight));
      amb_ray = ray_pack(bounce.ray);
    }

    /* ---- delta-path classification (geometry.cuh:80-101) ---- */
    const float roughness = mp_roughness(&g.params);
    bool is_delta;
    if (bounce.is_transparent_pass) {
      const float ior = mp_ior(&g.params);
      const float rs = (ior >= 1.0f) ? ior : 1.0f / ior;
      is_delta = roughness * fminf(rs - 1.0f, 1.0f) <= GEOMETRY_DELTA_PATH_CUTOFF;
    }
    else is_delta = bounce.is_microfacet_based && (roughness <= GEOMETRY_DELTA_PATH_CUTOFF);
    const bool pass_through = bsdf_is_pass_through_ray(&g, &bounce);

    /* ---- emission and throughput (geometry.cuh:103-119) ---- */
    const RGBF record_in = record_unpack(record_p);
    RGBF record = record_in;
    const RGBF emission = mp_emission(&g.params);
    if (c_any(emission)) beauty_add(&result, c_mul(emission, record));
    record = c_mul(record, bounce.weight);

    /* ---- shadow pass for this vertex (optix_kernel_shadow.cu:15-100) ---- */
    {
      RGBF acc = c_splat(0.0f);
      { /* direct_lighting.cuh:445-464 */
        const bool valid = (ls.light_id != LIGHT_ID_INVALID) && geo_allowed;
        RGBF vis = c_splat(0.0f);
        if (valid) {
          cnt[ORACLE_CNT_SHADOW]++;
          vis = trace_shadow(tr, hit_origin, ls.ray, ls.dist, s->light_tri_handles[2 * ls.light_id], s->light_tri_handles[2 * ls.light_id + 1], hit.instance_id, hit.tri_id);
        }
        acc = c_add(acc, c_mul(ls.light_color, vis));
      }
      acc = c_add(acc, bsdf_light_evaluate(s, tr, cnt, &smp, hit_origin, hit.instance_id, hit.tri_id, &lb, root_sum, top_volume, bsdf_allowed));
      { /* direct_lighting.cuh:466-519 */
        RGBF lc = sun_evaluate(s, tr, cnt, hit_origin, hit.instance_id, hit.tri_id, top_volume, sun_color, sun_ray, sun_allowed);
        if (!sun_allowed) lc = c_splat(0.0f);
        acc = c_add(acc, lc);
      }
      { /* direct_lighting.cuh:521-584 */
        RGBF lc = ambient_evaluate(s, tr, cnt, hit_origin, hit.instance_id, hit.tri_id, top_volume, second_volume, amb_color, amb_ray, ambient_allowed);
        if (!ambient_allowed) lc = c_splat(0.0f);
        acc = c_add(acc, lc);
      }
      beauty_add(&result, c_mul(acc, record_in));
    }

    /* ---- bounce (geometry.cuh:121-176) ---- */
    uint16_t new_state = state | ST_USE_IGNORE_HANDLE;
    if (s->sky_mode != SKY_MODE_DEFAULT && !pass_through) new_state &= ~ST_ALLOW_AMBIENT;
    else new_state |= ST_ALLOW_AMBIENT;
    if (!is_delta) new_state &= ~ST_DELTA_PATH;
    if (!pass_through) { new_state &= ~ST_CAMERA_DIRECTION; new_state &= ~ST_ALLOW_EMISSION; }
    if (!russian_roulette(s, &smp, state, &record)) break;
    record_p = record_pack(record);
    if (bounce.is_transparent_pass) {
      const bool refr_inside = (g.params.flags & MAT_REFRACTION_IS_INSIDE) != 0;
      float new_ior = 1.0f;
      if (!refr_inside) new_ior = medium_ior_peek(medium, refr_inside) / mp_ior(&g.params);
      medium = medium_ior_modify(medium, new_ior, !refr_inside);
    }
    state = new_state;
    origin = g.position;
    ray = bounce.ray;
    ign_inst = g.instance_id; ign_tri = g.tri_id;
  }
  return result;
}

int oracle_render(
  const OracleScene* s, const uint32_t* pixels, uint32_t num_pixels, uint32_t first_sample, uint32_t num_samples, int use_bvh, int threads,
  float* first_moment, float* second_moment, uint64_t* counters) {
  if (!s || !first_moment) return 1;
  OTracer tr;
  tracer_init(&tr, s, use_bvh);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  uint64_t total[ORACLE_CNT_COUNT] = {0, 0, 0, 0};
#pragma omp parallel
  {
    uint64_t cnt[ORACLE_CNT_COUNT] = {0, 0, 0, 0};
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t) num_pixels; i++) {
      const uint32_t index = pixels ? pixels[i] : (uint32_t) i;
      const uint32_t y = index / s->width, x = index - y * s->width;
      for (uint32_t k = 0; k < num_samples; k++) {
        const uint32_t sample_id = first_sample + k;
        if (sample_id >= MAX_GLOBAL_SAMPLES) continue;
        const RGBF r = render_path(s, &tr, x, y, sample_id, cnt);
        first_moment[i] += r.r;
        first_moment[(size_t) num_pixels + i] += r.g;
        first_moment[2 * (size_t) num_pixels + i] += r.b;
        if (second_moment) second_moment[i] += c_luminance(c_mul(r, r));
      }
    }
#pragma omp critical
    for (int k = 0; k < ORACLE_CNT_COUNT; k++) total[k] += cnt[k];
  }
  if (counters) for (int k = 0; k < ORACLE_CNT_COUNT; k++) counters[k] += total[k];
  tracer_free(&tr);
  return 0;
}

int oracle_trace_closest(
  const OracleScene* s, uint32_t num_rays, const float* origins, const float* dirs, const uint32_t* ignore_handles, int use_bvh, uint32_t* out_hits) {
  OTracer tr;
  tracer_init(&tr, s, use_bvh);
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < (int64_t) num_rays; i++) {
    const vec3 o = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), d = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    const bool ign = ignore_handles != NULL && ignore_handles[2 * i] != 0xFFFFFFFFu;
    const OHit h = trace_closest(&tr, o, d, ign, ign ? ignore_handles[2 * i] : 0, ign ? ignore_handles[2 * i + 1] : 0);
    out_hits[3 * i] = h.instance_id; out_hits[3 * i + 1] = h.tri_id; out_hits[3 * i + 2] = f2u(h.t);
  }
  tracer_free(&tr);
  return 0;
}

/* ---- BSDF energy LUTs (bsdf_lut.cuh:20-211); pixel (0,0), depth 0, sample id = iteration ---- */
#define LUT_ITER 0x10000u
static uint16_t lut_quant(float sum) { return (uint16_t) (1 + (uint16_t) (ceilf(o_saturate(sum) * 0xFFFE))); }

int oracle_generate_lut(const uint32_t* bn, int table, uint32_t first, uint32_t count, const uint16_t* conductor, uint16_t* dst) {
#pragma omp parallel for schedule(dynamic, 1)
  for (int64_t k = 0; k < (int64_t) count; k++) {
    const uint32_t id = first + (uint32_t) k;
    uint32_t x, y, z = 0;
    if (table < 2) { y = id / 32; x = id - y * 32; }
    else { z = id / 1024; y = (id - z * 1024) / 32; x = id - y * 32 - z * 1024; }
    const float NdotV = fmaxf(32.0f * O_EPS, x * (1.0f / 31));
    const float roughness_in = y * (1.0f / 31);
    /* the kernels pass the raw float roughness to the sampling functions, not the 10-bit quantised one */
    const float roughness = roughness_in;
    const vec3 V = v_norm(v3(0.0f, sqrtf(1.0f - NdotV * NdotV), NdotV));
    Sampler smp = {bn, 0, 0, 0, 0};
    float sum = 0.0f;
    if (table == 0 || table == 1) {
      const RGBF f0 = c3(0.04f, 0.04f, 0.04f);
      for (uint32_t i = 0; i < LUT_ITER; i++) {
        smp.sample_id = i;
        const vec3 H = microfacet_sample_normal(V, roughness, rnd2(&smp, RT_BSDF_REFLECTION));
        const vec3 R = v_reflect(V, H);
        const float NdotL = R.z;
        if (NdotL > 0.0f) {
          float v = microfacet_eval_sampled_microfacet(V, roughness, NdotL, NdotV);
          if (table == 1) v = v * c_luminance(bsdf_fresnel_schlick(f0, bsdf_shadowed_F90(f0), fabsf(v_dot(H, V))));
          sum += v;
        }
      }
      sum /= LUT_ITER;
      if (table == 1) sum /= conductor[id] * (1.0f / 0xFFFF);
    }
    else {
      const float ior_base = 1.0f + z * (1.0f / 31) * 2.0f;
      const float ior = (table == 2) ? 1.0f / ior_base : ior_base;
      for (uint32_t i = 0; i < LUT_ITER; i++) {
        smp.sample_id = i;
        bool tot;
        vec3 H = microfacet_sample_normal(V, roughness, rnd2(&smp, RT_BSDF_REFLECTION));
        vec3 R = v_reflect(V, H);
        vec3 T = refract_vector(V, H, ior, &tot);
        float fres = tot ? 1.0f : bsdf_fresnel(H, V, T, ior);
        if (R.z > 0.0f) sum += microfacet_eval_sampled_microfacet(V, roughness, R.z, NdotV) * fres;
        H = microfacet_refraction_sample_normal(V, roughness, rnd2(&smp, RT_BSDF_REFRACTION));
        T = refract_vector(V, H, ior, &tot);
        fres = tot ? ((table == 2) ? 1.0f : 0.0f) : bsdf_fresnel(H, V, T, ior);
        const float HdotV = fabsf(v_dot(H, V));
        const float NdotR = -T.z;
        if (NdotR > 0.0f) {
          const float r2 = roughness * roughness;
          const float val = ggx_G2_over_G1(r2 * r2, NdotR, NdotV);
          (void) HdotV;
          sum += val * (1.0f - fres);
        }
      }
      sum /= LUT_ITER;
    }
    dst[k] = lut_quant(sum);
  }
  return 0;
}

/* ---- unit-level entry points ---- */
uint32_t oracle_squares32(uint32_t key, uint32_t counter) { return squares32(key, counter); }
void oracle_sobol(uint32_t offset, uint32_t dimension, uint32_t out[2]) { const uint2_t r = rng_sobol(offset, dimension); out[0] = r.x; out[1] = r.y; }
void oracle_random_2d(const uint32_t* bn, uint32_t target, uint32_t px, uint32_t py, uint32_t sample, uint32_t depth, uint32_t out[2]) {
  const uint2_t r = rng_2d_u32(bn, target, px, py, sample, depth); out[0] = r.x; out[1] = r.y;
}
void oracle_record_roundtrip(const float in[3], uint32_t packed[2], float out[3]) {
  const uint2_t p = record_pack(c3(in[0], in[1], in[2])); packed[0] = p.x; packed[1] = p.y;
  const RGBF r = record_unpack(p); out[0] = r.r; out[1] = r.g; out[2] = r.b;
}
void oracle_ray_roundtrip(const float in[3], uint32_t packed[2], float out[3]) {
  const uint2_t p = ray_pack(v3(in[0], in[1], in[2])); packed[0] = p.x; packed[1] = p.y;
  const vec3 r = ray_unpack(p); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
uint32_t oracle_normal_pack(const float in[3]) { return normal_pack(v3(in[0], in[1], in[2])); }
void oracle_normal_unpack(uint32_t packed, float out[3]) { const vec3 n = normal_unpack(packed); out[0] = n.x; out[1] = n.y; out[2] = n.z; }
void oracle_sincos(float x, float out[2]) { o_sincos(x, &out[0], &out[1]); }
float oracle_atan2(float y, float x) { return o_atan2(y, x); }
void oracle_camera_ray(const OracleScene* s, uint32_t x, uint32_t y, uint32_t sample_id, float out[6]) {
  Sampler smp = {s->bluenoise_2d, x, y, sample_id, 0};
  vec3 o, d; camera_sample(s, &smp, &o, &d);
  out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = d.x; out[4] = d.y; out[5] = d.z;
}
_Static_assert(sizeof(OracleOutputParams) == sizeof(OracleOutputParamsAbi), "output parameter structs must match");
uint32_t oracle_scene_sizeof(void) { return (uint32_t) sizeof(OracleScene); }

void oracle_generate_output(const OracleOutputParamsAbi* params, const float* first_moment, const uint16_t* bluenoise_1d, float* frame_output, uint32_t* argb8) {
  OracleOutputParams p;
  memcpy(&p, params, sizeof(p));
  output_generate(&p, first_moment, bluenoise_1d, frame_output, argb8);
}
void oracle_post_bloom(float* image, uint32_t full_width, uint32_t full_height, uint32_t stage, float blend) { output_bloom(image, full_width, full_height, stage, blend); }
void oracle_result_undersampled(const float* first_moment, uint32_t width, uint32_t height, uint32_t stage, uint32_t iteration, float* result) {
  output_result_undersampled(first_moment, width, height, stage, iteration, result);
}
float oracle_log2(float x) { return o_log2(x); }
float oracle_exp2(float x) { return o_exp2(x); }
float oracle_pow(float x, float y) { return o_pow(x, y); }

/* ---- adaptive sampling (o_adaptive.h) ---- */
int oracle_render_counts(
  const OracleScene* s, const uint32_t* first_sample, const uint32_t* num_samples, int use_bvh, int threads, float* first_moment, float* second_moment,
  uint64_t* counters) {
  if (!s || !first_moment || !first_sample || !num_samples) return 1;
  OTracer tr;
  tracer_init(&tr, s, use_bvh);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  const uint32_t num_pixels = s->width * s->height;
  uint64_t total[ORACLE_CNT_COUNT] = {0, 0, 0, 0};
#pragma omp parallel
  {
    uint64_t cnt[ORACLE_CNT_COUNT] = {0, 0, 0, 0};
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t) num_pixels; i++) {
      const uint32_t y = (uint32_t) i / s->width, x = (uint32_t) i - y * s->width;
      for (uint32_t k = 0; k < num_samples[i]; k++) {
        const uint32_t sample_id = first_sample[i] + k;
        if (sample_id >= MAX_GLOBAL_SAMPLES) break;
        const RGBF r = render_path(s, &tr, x, y, sample_id, cnt);
        first_moment[i] += r.r;
        first_moment[(size_t) num_pixels + i] += r.g;
        first_moment[2 * (size_t) num_pixels + i] += r.b;
        if (second_moment) second_moment[i] += c_luminance(c_mul(r, r));
      }
    }
#pragma omp critical
    for (int k = 0; k < ORACLE_CNT_COUNT; k++) total[k] += cnt[k];
  }
  if (counters) for (int k = 0; k < ORACLE_CNT_COUNT; k++) counters[k] += total[k];
  tracer_free(&tr);
  return 0;
}

static OAdaptive adaptive_view(uint32_t width, uint32_t height, const uint32_t executions[5], uint32_t stage_id, const uint32_t* stage_counts) {
  OAdaptive a;
  a.stage_counts = stage_counts;
  a.blocks_x = (width + 3u) >> 2; a.blocks_y = (height + 3u) >> 2;
  for (int k = 0; k < 5; k++) a.executions[k] = executions ? executions[k] : 0u;
  a.stage_id = stage_id;
  return a;
}

void oracle_adaptive_build_stage(
  uint32_t width, uint32_t height, const uint32_t executions[5], uint32_t current_stage, uint32_t max_rate, uint32_t avg_rate, float exposure,
  const OracleOutputParamsAbi* op, const float* first_moment, const float* second_moment, uint32_t* stage_counts, float* block_variance, float* total) {
  const OAdaptive a = adaptive_view(width, height, executions, current_stage, stage_counts);
  const uint32_t nb = a.blocks_x * a.blocks_y;
  float* bv = block_variance ? block_variance : (float*) malloc(sizeof(float) * nb);
  oa_block_variance(&a, (const OracleOutputParams*) op, width, height, exposure, first_moment, second_moment, bv);
  const float t = oa_variance_total(bv, nb);
  if (total) *total = t;
  oa_stage_counts(bv, t, nb, current_stage, max_rate, avg_rate, stage_counts);
  if (!block_variance) free(bv);
}

/* The two halves of the stage build, for a render partitioned over ranks: block variances of this rank's accumulators, then the rates
 * from the complete variance array. */
void oracle_adaptive_block_variance(
  uint32_t width, uint32_t height, const uint32_t executions[5], uint32_t current_stage, float exposure, const OracleOutputParamsAbi* op,
  const float* first_moment, const float* second_moment, const uint32_t* stage_counts, float* block_variance) {
  const OAdaptive a = adaptive_view(width, height, executions, current_stage, stage_counts);
  oa_block_variance(&a, (const OracleOutputParams*) op, width, height, exposure, first_moment, second_moment, block_variance);
}
float oracle_adaptive_counts_from(
  uint32_t width, uint32_t height, uint32_t current_stage, uint32_t max_rate, uint32_t avg_rate, const float* block_variance, uint32_t* stage_counts) {
  const uint32_t nb = ((width + 3u) >> 2) * ((height + 3u) >> 2);
  const float t = oa_variance_total(block_variance, nb);
  oa_stage_counts(block_variance, t, nb, current_stage, max_rate, avg_rate, stage_counts);
  return t;
}

void oracle_pixel_samples(uint32_t width, uint32_t height, const uint32_t executions[5], const uint32_t* stage_counts, uint32_t* out) {
  const OAdaptive a = adaptive_view(width, height, executions, 0, stage_counts);
  for (uint32_t y = 0; y < height; y++)
    for (uint32_t x = 0; x < width; x++) out[x + y * width] = oa_pixel_samples(&a, stage_counts[oa_block_of(&a, x, y)]);
}

void oracle_generate_result(
  uint32_t width, uint32_t height, uint32_t mode, uint32_t local_error_minimization, uint32_t uniform_samples, float exposure, const uint32_t executions[5],
  uint32_t stage_id, const uint32_t* stage_counts, const OracleOutputParamsAbi* op, const float* first_moment, const float* second_moment, float* frame_result) {
  const OAdaptive a = adaptive_view(width, height, executions, stage_id, stage_counts);
  const OResultParams rp = {width, height, mode, local_error_minimization, uniform_samples, exposure};
  oa_generate_result(&a, &rp, (const OracleOutputParams*) op, first_moment, second_moment, frame_result);
}

/* ---- procedural sky (o_sky.h) ---- */
void oracle_sky_generate_luts(const OracleScene* scene, float* transmittance, float* multiscattering) {
  OSky s = osky_view(scene);
  sky_transmittance_lut(&s, transmittance);
  s.tm = transmittance;
  sky_multiscattering_lut(&s, multiscattering);
}
void oracle_sky_color(const OracleScene* scene, const float origin_world[3], const float ray[3], int include_sun, float random_offset, float out[3]) {
  const OSky s = osky_view(scene);
  const RGBF c = sky_get_color(&s, world_to_sky(&s, v3(origin_world[0], origin_world[1], origin_world[2])), v3(ray[0], ray[1], ray[2]), FLT_MAX, include_sun != 0,
                               (int) s.steps, random_offset);
  out[0] = c.r; out[1] = c.g; out[2] = c.b;
}
void oracle_sky_hdri_color(const OracleScene* scene, const float origin_world[3], const float ray[3], uint32_t state, float out[3]) {
  const RGBF c = sky_hdri_color(scene, v3(origin_world[0], origin_world[1], origin_world[2]), v3(ray[0], ray[1], ray[2]), state);
  out[0] = c.r; out[1] = c.g; out[2] = c.b;
}
void oracle_sky_hdri(const OracleScene* scene, const float origin_world[3], uint32_t dim, uint32_t samples, float* rgba) {
  sky_hdri_bake(scene, v3(origin_world[0], origin_world[1], origin_world[2]), dim, samples, rgba);
}

/* ---- probes for the analytic checks of the restatement (tests/test_oracle_analytic.py): single functions of the path on caller-made
 * shading contexts. Nothing here is on a rendering path. ---- */
static GeoCtx probe_context(const OracleProbeMaterial* m, const float position[3], const float normal[3], const float V[3]) {
  GeoCtx g;
  g.instance_id = 0xFFFFFFF0u; g.tri_id = 0;
  g.position = v3(position[0], position[1], position[2]);
  g.normal = v_norm(v3(normal[0], normal[1], normal[2]));
  g.V = v_norm(v3(V[0], V[1], V[2]));
  g.face_normal = normal_pack(g.normal);
  g.state = 0;
  g.params.data[0] = g.params.data[1] = g.params.data[2] = 0;
  g.params.flags = m->flags;
  mp_set_albedo(&g.params, c3(m->albedo[0], m->albedo[1], m->albedo[2]));
  mp_set_opacity(&g.params, m->opacity);
  mp_set_roughness(&g.params, m->roughness);
  mp_set_emission(&g.params, c_splat(0.0f));
  mp_set_ior(&g.params, m->ior_ratio);
  return g;
}
/* bsdf_sample<GEOMETRY> for sample ids first..first+count-1 of pixel (px, py), depth constant 0: direction, weight, flags (1 transparent pass, 2 microfacet based) */
void oracle_probe_bsdf_sample(const OracleScene* s, const OracleProbeMaterial* m, const float normal[3], const float V[3], uint32_t px, uint32_t py, uint32_t first,
                              uint32_t count, float* rays, float* weights, uint32_t* flags) {
  const OLuts luts = scene_luts(s);
  const float origin[3] = {0.0f, 0.0f, 0.0f};
  const GeoCtx g = probe_context(m, origin, normal, V);
  for (uint32_t i = 0; i < count; i++) {
    const Sampler smp = {s->bluenoise_2d, px, py, first + i, 0};
    const BSDFSample b = bsdf_sample(&luts, &g, &smp, 0);
    rays[3 * i] = b.ray.x; rays[3 * i + 1] = b.ray.y; rays[3 * i + 2] = b.ray.z;
    weights[3 * i] = b.weight.r; weights[3 * i + 1] = b.weight.g; weights[3 * i + 2] = b.weight.b;
    if (flags) flags[i] = (b.is_transparent_pass ? 1u : 0u) | (b.is_microfacet_based ? 2u : 0u);
  }
}
/* bsdf_evaluate (general hint) for `count` directions L: value = f * |cos| * inv_pdf as the path tracer uses it */
void oracle_probe_bsdf_eval(const OracleScene* s, const OracleProbeMaterial* m, const float normal[3], const float V[3], uint32_t count, const float* L, float inv_pdf,
                            float* values) {
  const OLuts luts = scene_luts(s);
  const float origin[3] = {0.0f, 0.0f, 0.0f};
  const GeoCtx g = probe_context(m, origin, normal, V);
  for (uint32_t i = 0; i < count; i++) {
    bool is_refraction;
    const RGBF v = bsdf_evaluate(&luts, &g, v3(L[3 * i], L[3 * i + 1], L[3 * i + 2]), 0, &is_refraction, inv_pdf);
    values[3 * i] = v.r; values[3 * i + 1] = v.g; values[3 * i + 2] = v.b;
  }
}
/* bsdf_microfacet_pdf of the bounded VNDF for reflected directions L (local frame: normal = +z) */
void oracle_probe_microfacet_pdf(const float V[3], float roughness, uint32_t count, const float* L, float* pdf) {
  const vec3 v = v_norm(v3(V[0], V[1], V[2]));
  for (uint32_t i = 0; i < count; i++) {
    const vec3 l = v3(L[3 * i], L[3 * i + 1], L[3 * i + 2]);
    const vec3 h = v_norm(v_add(v, l));
    pdf[i] = (l.z > 0.0f) ? microfacet_pdf(v, roughness, h.z, v.z) : 0.0f;
  }
}
/* light_triangle_sample_solid_angle for caller-supplied random pairs: ok flag, direction, reported solid angle */
void oracle_probe_triangle_sample(const float origin[3], const float tri[9], int bidirectional, uint32_t count, const float* rnd, float* rays, float* solid_angles,
                                  uint32_t* ok) {
  const vec3 o = v3(origin[0], origin[1], origin[2]), p0 = v3(tri[0], tri[1], tri[2]);
  const vec3 e1 = v_sub(v3(tri[3], tri[4], tri[5]), p0), e2 = v_sub(v3(tri[6], tri[7], tri[8]), p0);
  for (uint32_t i = 0; i < count; i++) {
    vec3 ray = v3(0.0f, 0.0f, 0.0f);
    float sa = 0.0f;
    const float2_t r = {rnd[2 * i], rnd[2 * i + 1]};
    ok[i] = light_triangle_sample_solid_angle(o, p0, e1, e2, r, bidirectional != 0, &ray, &sa) ? 1u : 0u;
    rays[3 * i] = ray.x; rays[3 * i + 1] = ray.y; rays[3 * i + 2] = ray.z;
    solid_angles[i] = sa;
  }
}
/* the light tree at a shading point: for sample ids first..first+count-1 the eight resampling lanes' picks and their weights (1 / (8 p)) */
void oracle_probe_light_tree(const OracleScene* s, const OracleProbeMaterial* m, const float position[3], const float normal[3], const float V[3], uint32_t px, uint32_t py,
                             uint32_t first, uint32_t count, uint32_t* light_ids, float* weights, float* root_sums) {
  const GeoCtx g = probe_context(m, position, normal, V);
  for (uint32_t i = 0; i < count; i++) {
    const Sampler smp = {s->bluenoise_2d, px, py, first + i, 0};
    const LTQuery query = lt_query_geometry(&g);
    const LTWork work = light_tree_prepass(s, &query, &smp);
    if (root_sums) root_sums[i] = work.root_sum;
    for (uint32_t lane = 0; lane < LIGHT_TREE_NUM_OUTPUTS; lane++) {
      const LTResult r = light_tree_postpass(s, &query, &smp, lane, &work);
      light_ids[i * LIGHT_TREE_NUM_OUTPUTS + lane] = r.light_id;
      weights[i * LIGHT_TREE_NUM_OUTPUTS + lane] = r.weight;
    }
  }
}
/* light_sample: the resampled light sample of a vertex (direction, colour = radiance x BSDF x MIS x resampling weight, distance) */
void oracle_probe_light_sample(const OracleScene* s, const OracleProbeMaterial* m, const float position[3], const float normal[3], const float V[3], uint32_t px, uint32_t py,
                               uint32_t first, uint32_t count, uint32_t* light_ids, float* rays, float* colors, float* dists) {
  const GeoCtx g = probe_context(m, position, normal, V);
  for (uint32_t i = 0; i < count; i++) {
    const Sampler smp = {s->bluenoise_2d, px, py, first + i, 0};
    const LightSample ls = light_sample(s, &g, &smp);
    light_ids[i] = ls.light_id;
    rays[3 * i] = ls.ray.x; rays[3 * i + 1] = ls.ray.y; rays[3 * i + 2] = ls.ray.z;
    colors[3 * i] = ls.light_color.r; colors[3 * i + 1] = ls.light_color.g; colors[3 * i + 2] = ls.light_color.b;
    dists[i] = ls.dist;
  }
}

/* the light-only BVH query of a BSDF-sampled direction (optix_anyhit.cuh:145-205 restated order-independently, o_trace.h trace_light_bvh): for every
 * random number the light picked among those the ray crosses and their count */
void oracle_probe_light_bvh(const OracleScene* s, const float origin[3], const float dir[3], uint32_t count, const float* randoms, uint32_t* light_ids, uint32_t* num_hits) {
  OTracer tr;
  tracer_init(&tr, s, 1);
  for (uint32_t i = 0; i < count; i++)
    light_ids[i] = trace_light_bvh(&tr, v3(origin[0], origin[1], origin[2]), v3(dir[0], dir[1], dir[2]), 0xFFFFFFFFu, 0u, randoms[i], &num_hits[i]);
  tracer_free(&tr);
}

/* ---- probes of the fog's building blocks (tests/test_fog.py) ---- */
void oracle_probe_volume_path(const float cam_pos[3], float dist, float height, uint32_t count, const float* origins, const float* dirs, const float* limits, float* out) {
  OracleScene sc;
  memset(&sc, 0, sizeof(sc));
  sc.cam_pos[0] = cam_pos[0]; sc.cam_pos[1] = cam_pos[1]; sc.cam_pos[2] = cam_pos[2];
  sc.fog_active = 1; sc.fog_density = 1.0f; sc.fog_dist = dist; sc.fog_height = height;
  const OVolume vol = fog_volume(&sc);
  for (uint32_t i = 0; i < count; i++) {
    const OVolumePath p = volume_compute_path(&sc, &vol, v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), limits[i], false);
    out[2 * i] = p.start; out[2 * i + 1] = p.length;
  }
}
void oracle_probe_fog_phase(const OracleScene* s, uint32_t count, const float* cos_angle, float* out) {
  for (uint32_t i = 0; i < count; i++) out[i] = fog_phase_function(s, cos_angle[i]);
}
/* rnd: 3 numbers per sample (direction x, direction y, lobe choice); out: the cosine between the incoming and the sampled direction */
void oracle_probe_fog_phase_sample(const OracleScene* s, uint32_t count, const float* rnd, float* out) {
  const vec3 in = v_norm(v3(0.3f, -0.5f, 0.8f));
  for (uint32_t i = 0; i < count; i++) {
    const float2_t r = {rnd[3 * i], rnd[3 * i + 1]};
    out[i] = v_dot(fog_phase_sample(s, in, r, rnd[3 * i + 2]), in);
  }
}
void oracle_probe_volume_sampling(float scattering, float max_length, uint32_t count, const float* rnd, float* t, float* pdf) {
  OVolume v;
  memset(&v, 0, sizeof(v));
  v.scattering = scattering; v.dist = 1.0f; v.max_height = 1.0f; v.type = VOLUME_TYPE_FOG;
  for (uint32_t i = 0; i < count; i++) { t[i] = volume_sample_bounded(&v, max_length, rnd[i]); pdf[i] = volume_sample_bounded_pdf(&v, max_length, t[i]); }
}

/* clouds (o_cloud.h): the three noise textures, RGBA8 (shape 128^3, detail 32^3, weather 1024^2 from the seed) */
void oracle_cloud_noise(uint32_t seed, uint32_t* shape, uint32_t* detail, uint32_t* weather) {
  if (shape) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t) CLOUD_SHAPE_RES * CLOUD_SHAPE_RES * CLOUD_SHAPE_RES; i++) {
      const uint32_t z = (uint32_t) (i / (CLOUD_SHAPE_RES * CLOUD_SHAPE_RES)), y = (uint32_t) ((i / CLOUD_SHAPE_RES) % CLOUD_SHAPE_RES), x = (uint32_t) (i % CLOUD_SHAPE_RES);
      shape[i] = cloud_shape_texel(x, y, z, CLOUD_SHAPE_RES);
    }
  }
  if (detail) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t) CLOUD_DETAIL_RES * CLOUD_DETAIL_RES * CLOUD_DETAIL_RES; i++) {
      const uint32_t z = (uint32_t) (i / (CLOUD_DETAIL_RES * CLOUD_DETAIL_RES)), y = (uint32_t) ((i / CLOUD_DETAIL_RES) % CLOUD_DETAIL_RES), x = (uint32_t) (i % CLOUD_DETAIL_RES);
      detail[i] = cloud_detail_texel(x, y, z, CLOUD_DETAIL_RES);
    }
  }
  if (weather) {
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < (int64_t) CLOUD_WEATHER_RES * CLOUD_WEATHER_RES; i++)
      weather[i] = cloud_weather_texel((uint32_t) (i % CLOUD_WEATHER_RES), (uint32_t) (i / CLOUD_WEATHER_RES), CLOUD_WEATHER_RES, (float) seed);
  }
}
/* probes for tests/test_clouds.py: one value of the tiling Perlin / Worley octaves; the density and the shadow at points in sky space */
void oracle_probe_cloud_noise(uint32_t count, const float* p, float scale, int octaves, float seed, float persistence, float* out_perlin, float* out_worley) {
  for (uint32_t i = 0; i < count; i++) {
    const vec3 q = v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    out_perlin[i] = perlin_octaves(q, scale, octaves, true);
    out_worley[i] = worley_octaves(q, scale, octaves, seed, persistence);
  }
}
void oracle_probe_cloud_density(const OracleScene* s, int layer, uint32_t count, const float* sky_pos, float* out_height, float* out_density) {
  for (uint32_t i = 0; i < count; i++) {
    const vec3 pos = v3(sky_pos[3 * i], sky_pos[3 * i + 1], sky_pos[3 * i + 2]);
    const float height = cloud_height(s, pos, layer);
    out_height[i] = height;
    out_density[i] = 0.0f;
    if (height < 0.0f || height > 1.0f) continue;
    const CloudWeather w = cloud_weather(s, pos, height, layer);
    if (cloud_significant_point(height, &w, layer)) out_density[i] = cloud_density(s, pos, height, &w, layer);
  }
}

void oracle_probe_ocean_height(const OracleScene* s, uint32_t count, const float* xz, float* out) {
  for (uint32_t i = 0; i < count; i++) out[i] = ocean_get_height(s, v3(xz[2 * i], 0.0f, xz[2 * i + 1]), OCEAN_ITERATIONS);
}
void oracle_probe_ocean_trace(const OracleScene* s, uint32_t count, const float* origins, const float* dirs, const float* limits, float* out_t, float* out_residual,
                              float* out_normal) {
  for (uint32_t i = 0; i < count; i++) {
    const vec3 o = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]), d = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    const float t = ocean_intersection_distance(s, o, d, limits[i]);
    out_t[i] = t;
    out_residual[i] = 0.0f; out_normal[3 * i] = out_normal[3 * i + 1] = out_normal[3 * i + 2] = 0.0f;
    if (t < limits[i]) {
      const vec3 p = v_add(o, v_scale(d, t));
      const vec3 n = ocean_get_normal(s, p);
      out_residual[i] = ocean_relative_height(s, p, OCEAN_ITERATIONS);
      out_normal[3 * i] = n.x; out_normal[3 * i + 1] = n.y; out_normal[3 * i + 2] = n.z;
    }
  }
}
void oracle_probe_ocean_fresnel(float ior, uint32_t count, const float* dirs, float* out_reflection, float* out_refracted) {
  for (uint32_t i = 0; i < count; i++) {
    const vec3 ray = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), normal = v3(0.0f, 1.0f, 0.0f);
    bool total_reflection;
    const vec3 refraction = refract_vector(v_scale(ray, -1.0f), normal, 1.0f / ior, &total_reflection);
    out_reflection[i] = ocean_reflection_coefficient(normal, ray, refraction, 1.0f / ior);
    out_refracted[3 * i] = refraction.x; out_refracted[3 * i + 1] = refraction.y; out_refracted[3 * i + 2] = refraction.z;
  }
}

/* lattice tracer of the particles on explicit rays (tests/test_particles.py): pos in [0,1)^3, dir = direction / particles_scale */
void oracle_probe_particle_trace(const OracleScene* s, uint32_t count, const float* pos, const float* dir, const float* tmax, float* out_t, uint32_t* out_tri) {
  OTracer tr;
  tracer_init(&tr, s, 1);
  for (uint32_t i = 0; i < count; i++) {
    float t = tmax[i];
    out_tri[i] = trace_particles(&tr, v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), v3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), tmax[i], &t);
    out_t[i] = t;
  }
  tracer_free(&tr);
}
