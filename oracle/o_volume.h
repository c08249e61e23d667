/*
 * ORACLE (test infrastructure, not product): the fog volume.
 * Follows /root/reference/src/luminary/device/cuda/volume_utils.cuh (descriptor, path through the disk-box, closed-form distance sampling,
 * transmittance), cuda/volume.cuh (in-scattering, events, bounce), cuda/light_bridges.cuh + light_common.cuh:17-32 (bridges to emissive
 * triangles), cuda/math.cuh:1169-1322 (phase functions), cuda/bsdf.cuh:302-318,:404-421,:458-474 (phase sampling as the volume's "BSDF"),
 * cuda/direct_lighting.cuh:20-121 (sun), :385-403,:521-584 (ambient), optix/optix_kernel_shadow_volume.cu.
 * Two volume types: the fog (scalar scattering, no absorption, a disk-box around the camera) and the ocean's water (RGB Jerlov coefficients, everything
 * below the surface). A path carries a stack of the volumes it is in (medium_stack.cuh:29-45): the fog from tasks_create on, the water when the camera
 * starts below the surface or a path refracts through it. Numerics contract as everywhere: expf := o_exp, logf := o_log2 * ln 2, cbrtf := o_exp2(o_log2 / 3).
 */
#ifndef ORACLE_O_VOLUME_H
#define ORACLE_O_VOLUME_H

#include "o_ocean.h"

#define HIT_TYPE_VOLUME_BASE 0xFFFE0000u /* VOLUME_ID_TO_HIT_ID(type) = HIT_TYPE_VOLUME_BASE | type (cuda/utils.cuh:61-63, :84-85; utils.h:39) */
#define HIT_TYPE_VOLUME_MAX 0xFFFEFFFFu
#define HIT_TYPE_VOLUME_FOG 0xFFFE0001u
#define BRIDGES_HG_G 0.85f
#define BRIDGES_FORWARD_PROB 0.95f
#define BRIDGES_MAX_VERTEX_COUNT 15u
#define LIGHT_GEO_MAX_BRIDGE_LENGTH 8u

static inline float o_log(float x) { return o_log2(x) * 0.693147181f; }
static inline float o_cbrt(float x) { return (x == 0.0f) ? 0.0f : copysignf(o_exp2(o_log2(fabsf(x)) * 0.333333333f), x); }
static inline float o_clamp(float x, float a, float b) { return fminf(b, fmaxf(a, x)); }

/* ---- descriptors (volume_utils.cuh:8-57). `scattering` is the reference's max_scattering: what distances are sampled with ---- */
enum { VOLUME_TYPE_NONE = 0, VOLUME_TYPE_FOG = 1, VOLUME_TYPE_OCEAN = 2 }; /* utils.h:39 */
typedef struct { float scattering, dist, max_height, min_height; RGBF scat, absorb; float max_absorption; uint32_t type; } OVolume;
static inline OVolume volume_descriptor(const OracleScene* s, uint32_t type) {
  OVolume v;
  memset(&v, 0, sizeof(v));
  v.type = type;
  if (type == VOLUME_TYPE_FOG) {
    v.scattering = 0.001f * s->fog_density;
    v.scat = c_splat(v.scattering); v.absorb = c_splat(0.0f); v.max_absorption = 0.0f;
    v.dist = s->fog_dist;
    v.max_height = s->fog_height;
    v.min_height = s->ocean_active ? ocean_max_height(s) : -65535.0f;
  }
  else if (type == VOLUME_TYPE_OCEAN) {
    v.absorb = c3(s->ocean_absorption[0], s->ocean_absorption[1], s->ocean_absorption[2]);
    v.scat = c3(s->ocean_scattering[0], s->ocean_scattering[1], s->ocean_scattering[2]);
    v.dist = 10000.0f; v.max_height = 65535.0f; v.min_height = -65535.0f;
    v.max_absorption = c_importance(v.absorb);
    v.scattering = c_importance(v.scat);
  }
  return v;
}
static inline OVolume fog_volume(const OracleScene* s) { return volume_descriptor(s, VOLUME_TYPE_FOG); }
/* the path's volume stack: four 8-bit ids, newest in the low byte (medium_stack_volume_peek / _modify, medium_stack.cuh:29-45, with 16-bit ids there) */
static inline uint32_t volume_stack_peek(uint32_t stack, bool previous) { return previous ? (stack >> 8) & 0xFFu : stack & 0xFFu; }
static inline uint32_t volume_stack_modify(uint32_t stack, uint32_t id, bool push) { return push ? (stack << 8) | id : stack >> 8; }

/* volume_utils.cuh:88-170: start >= 0 iff the ray passes through the volume within `limit` */
typedef struct { float start, length; } OVolumePath;
static inline OVolumePath volume_compute_path(const OracleScene* s, const OVolume* vol, vec3 origin, vec3 ray, float limit, bool ocean_fast_path) {
  const OVolumePath none = {-FLT_MAX, 0.0f};
  if (limit <= 0.0f) return none;
  if (vol->max_height <= vol->min_height) return none;
  if (vol->type == VOLUME_TYPE_NONE) return none;
  float start_y, end_y;
  if (vol->type == VOLUME_TYPE_OCEAN) {
    start_y = 0.0f;
    end_y = ocean_fast_path ? limit : ocean_intersection_distance(s, origin, ray, limit);
  }
  else if (fabsf(ray.y) < 0.005f) {
    if (origin.y >= vol->min_height && origin.y <= vol->max_height) { start_y = 0.0f; end_y = vol->dist; }
    else return none;
  }
  else {
    const float sy1 = (vol->min_height - origin.y) / ray.y, sy2 = (vol->max_height - origin.y) / ray.y;
    start_y = fmaxf(fminf(sy1, sy2), 0.0f);
    end_y = fmaxf(sy1, sy2);
  }
  const float rn = o_rsqrt(ray.x * ray.x + ray.z * ray.z);
  const float rx = ray.x * rn, rz = ray.z * rn;
  const float dx = origin.x - s->cam_pos[0], dz = origin.z - s->cam_pos[2];
  const float dot = dx * rx + dz * rz;
  const float r2 = vol->dist * vol->dist;
  const float c = (dx * dx + dz * dz) - r2;
  const float kx = dx - rx * dot, kz = dz - rz * dot;
  const float d = r2 - (kx * kx + kz * kz);
  if (d < 0.0f) return none;
  const float sd = sqrtf(d);
  const float q = -dot - copysignf(sd, dot);
  const float t0 = fmaxf(0.0f, c / q), t1 = fmaxf(0.0f, q);
  const float start_xz = fminf(t0, t1), end_xz = fmaxf(t0, t1);
  if (end_xz < start_xz || limit < start_xz) return none;
  const float start = fmaxf(start_xz, start_y);
  const float dist = fminf(fminf(end_xz, end_y) - start, limit - start);
  if (dist < 0.0f) return none;
  const OVolumePath p = {start, dist};
  return p;
}
/* volume_utils.cuh:182-214 */
static inline float volume_sample_intersection(const OVolume* v, float start, float max_length, float random) {
  const float t = (-o_log(random)) / v->scattering;
  return (t > max_length) ? FLT_MAX : start + t;
}
static inline float volume_sample_intersection_pdf(const OVolume* v, float start, float t) { return v->scattering * o_exp(-v->scattering * (t - start)); }
static inline float volume_miss_probability(const OVolume* v, float depth) { return o_exp(-v->scattering * depth); }
static inline float volume_sample_bounded(const OVolume* v, float max_length, float random) {
  const float prob_hit_at_max = 1.0f - o_exp(-v->scattering * max_length);
  return -o_log(1.0f - random * prob_hit_at_max) / v->scattering;
}
static inline float volume_sample_bounded_pdf(const OVolume* v, float max_length, float t) {
  const float prob_hit_at_max = 1.0f - o_exp(-v->scattering * max_length);
  return v->scattering * o_exp(-v->scattering * t) / prob_hit_at_max;
}
/* volume_integrate_transmittance_precomputed (volume_utils.cuh:245-254) */
static inline RGBF volume_transmittance_length(const OVolume* v, float length) {
  return c3(o_exp(-length * (v->absorb.r + v->scat.r)), o_exp(-length * (v->absorb.g + v->scat.g)), o_exp(-length * (v->absorb.b + v->scat.b)));
}
/* volume_integrate_transmittance (volume_utils.cuh:292-308) of the volume a vertex is in: 1 without one */
static inline RGBF volume_transmittance(const OracleScene* s, uint32_t type, vec3 origin, vec3 ray, float depth) {
  const OVolume v = volume_descriptor(s, type);
  const OVolumePath p = volume_compute_path(s, &v, origin, ray, depth, false);
  return (p.start >= 0.0f) ? volume_transmittance_length(&v, p.length) : c_splat(1.0f);
}

/* ---- phase functions (math.cuh:1169-1322); the Jendersie-Eon parameters of the droplet diameter come with the scene ---- */
static inline float draine_phase(float c, float g, float alpha) { return hg_phase(c, g) * ((1.0f + alpha * c * c) / (1.0f + (alpha / 3.0f) * (1.0f + 2.0f * g * g))); }
static inline float je_phase_function(const float p[4], float c) { /* jendersie_eon_phase_function, math.cuh:1234-1239; p = g_hg, g_d, alpha, w_d */
  return (1.0f - p[3]) * hg_phase(c, p[0]) + p[3] * draine_phase(c, p[1], p[2]);
}
static inline float fog_phase_function(const OracleScene* s, float c) { return je_phase_function(s->fog_phase, c); }
static inline vec3 phase_sample_basis(float alpha, float beta, vec3 basis) { /* math.cuh:1249-1272 */
  vec3 u1, u2;
  if (basis.z < -0.9999805689f) { u1 = v3(0.0f, -1.0f, 0.0f); u2 = v3(-1.0f, 0.0f, 0.0f); }
  else {
    const float a = 1.0f / (1.0f + basis.z);
    const float b = -basis.x * basis.y * a;
    u1 = v3(1.0f - basis.x * basis.x * a, b, -basis.x);
    u2 = v3(b, 1.0f - basis.y * basis.y * a, -basis.y);
  }
  const vec3 sp = sample_ray_sphere(alpha, beta);
  return v_norm(v3(sp.x * u1.x + sp.y * u2.x + sp.z * basis.x, sp.x * u1.y + sp.y * u2.y + sp.z * basis.y, sp.x * u1.z + sp.y * u2.z + sp.z * basis.z));
}
static inline float hg_phase_sample(float g, float r) {
  const float g2 = g * g;
  const float t = (1.0f - g2) / (1.0f - g + 2.0f * g * r);
  return (1.0f + g2 - t * t) / (2.0f * g);
}
static inline float draine_phase_sample(float g, float alpha, float r) { /* math.cuh:1283-1300 */
  const float g2 = g * g, g4 = g2 * g2;
  const float t0 = alpha - alpha * g2;
  const float t1 = alpha * g4 - alpha;
  const float t2 = -3.0f * (4.0f * (g4 - g2) + t1 * (1.0f + g2));
  const float t3 = g * (2.0f * r - 1.0f);
  const float t4 = 3.0f * g2 * (1.0f + t3) + alpha * (2.0f + g2 * (1.0f + (1.0f + 2.0f * g2) * t3));
  const float t5 = t0 * (t1 * t2 + t4 * t4) + t1 * t1 * t1;
  const float t6 = t0 * 4.0f * (g4 - g2);
  const float t7 = o_cbrt(t5 + sqrtf(t5 * t5 - t6 * t6 * t6));
  const float t8 = 2.0f * ((t1 + (t6 / t7) + t7) / t0);
  const float t9 = sqrtf(6.0f * (1.0f + g2) + t8);
  const float h = sqrtf(6.0f * (1.0f + g2) - t8 + 8.0f * t4 / (t0 * t9)) - t9;
  return 0.5f * g + ((1.0f / (2.0f * g)) - (1.0f / (8.0f * g)) * (h * h));
}
static inline vec3 je_phase_sample(const float p[4], vec3 ray, float2_t r_dir, float r_choice) { /* jendersie_eon_phase_sample, math.cuh:1311-1323 */
  const float cos_angle = (r_choice < p[3]) ? draine_phase_sample(p[1], p[2], r_dir.x) : hg_phase_sample(p[0], r_dir.x);
  return phase_sample_basis(cos_angle, r_dir.y, ray);
}
static inline vec3 fog_phase_sample(const OracleScene* s, vec3 ray, float2_t r_dir, float r_choice) { return je_phase_sample(s->fog_phase, ray, r_dir, r_choice); }

/* the phase function is drawn from by volume type (bsdf.cuh:310-314; ocean_phase_sampling, ocean_utils.cuh:412-425) */
static inline vec3 volume_phase_sample(const OracleScene* s, uint32_t type, vec3 ray, float2_t r_dir, float r_choice) {
  if (type == VOLUME_TYPE_OCEAN) return phase_sample_basis(ocean_phase_sample_cos(s, r_dir.x, r_choice), r_dir.y, ray);
  return fog_phase_sample(s, ray, r_dir, r_choice);
}

/* ---- the volume's shading context (material.cuh:76-89, volume_utils.cuh:310-321) ---- */
typedef struct VolCtx { OVolume vol; vec3 position, V; uint16_t state; float max_dist; } VolCtx;
static inline VolCtx volume_context(const OracleScene* s, uint32_t type, vec3 origin, vec3 ray, uint16_t state, float max_dist) {
  VolCtx c;
  c.vol = volume_descriptor(s, type); c.position = origin; c.V = v_scale(ray, -1.0f); c.state = state; c.max_dist = max_dist;
  return c;
}
/* bsdf_sample<MATERIAL_VOLUME> (bsdf.cuh:302-318): the weight is 1 */
static inline vec3 volume_bsdf_sample(const OracleScene* s, const VolCtx* c, const Sampler* smp, uint32_t rt_resampling, uint32_t rt_diffuse) {
  const float random_choice = rnd1(smp, rt_resampling);
  const float2_t random_dir = rnd2(smp, rt_diffuse);
  return volume_phase_sample(s, c->vol.type, v_scale(c->V, -1.0f), random_dir, random_choice);
}
/* volume_phase_evaluate (volume_utils.cuh:216-243) = bsdf_evaluate<VOLUME> = bsdf_sample_for_sun_pdf<VOLUME> */
static inline float volume_phase_evaluate(const OracleScene* s, const VolCtx* c, vec3 L) {
  const float cos_angle = -v_dot(c->V, L);
  return (c->vol.type == VOLUME_TYPE_OCEAN) ? ocean_phase(s, cos_angle) : fog_phase_function(s, cos_angle);
}

/* volume_sample_sky_dl_initial_vertex (volume_utils.cuh:323-352): moves the context to a vertex on the ray; returns its weight */
static inline RGBF volume_sky_initial_vertex(VolCtx* c, const Sampler* smp) {
  const float dist = volume_sample_bounded(&c->vol, c->max_dist, rnd1(smp, RT_LIGHT_SUN_INITIAL_VERTEX));
  c->position = v_add(c->position, v_scale(c->V, -dist));
  const RGBF w = c3(o_exp(-dist * (c->vol.absorb.r + c->vol.scat.r)) * c->vol.scat.r, o_exp(-dist * (c->vol.absorb.g + c->vol.scat.g)) * c->vol.scat.g,
                    o_exp(-dist * (c->vol.absorb.b + c->vol.scat.b)) * c->vol.scat.b);
  return c_scale(w, 1.0f / volume_sample_bounded_pdf(&c->vol, c->max_dist, dist));
}

/* direct_lighting_sun_create_task + direct_lighting_sun_direct for a volume vertex (direct_lighting.cuh:20-121, :352-383; random set LIGHT_SUN<1>) */
static inline bool volume_sun_sample(const OracleScene* s, const OSky* sky, const VolCtx* c, const Sampler* smp, RGBF* light_out, vec3* dir_out) {
  const vec3 sky_pos = world_to_sky(sky, c->position);
  const bool sun_below_horizon = sph_hit_p0(v_norm(v_sub(sky->sun_pos, sky_pos)), sky_pos, SKY_EARTH_RADIUS);
  const bool inside_earth = v_len(sky_pos) < SKY_EARTH_RADIUS;
  if (sun_below_horizon || inside_earth) return false;
  /* bsdf_sample_for_sun<VOLUME>, bsdf.cuh:404-421 */
  const float2_t random_dir = rnd2(smp, RT_VOL_SUN_BSDF);
  const float random_method = rnd1(smp, RT_VOL_SUN_BSDF_METHOD);
  const vec3 dir_bsdf = volume_phase_sample(s, c->vol.type, v_scale(c->V, -1.0f), random_dir, random_method);
  RGBF light_bsdf = c_splat(0.0f);
  if (sphere_hit(dir_bsdf, sky_pos, sky->sun_pos, SKY_SUN_RADIUS)) light_bsdf = c_mul(sky_sun_color(sky, sky_pos, dir_bsdf), c_splat(volume_phase_evaluate(s, c, dir_bsdf) * 1.0f));
  float solid_angle;
  const vec3 dir_sa = sample_sphere(sky->sun_pos, SKY_SUN_RADIUS, sky_pos, rnd2(smp, RT_VOL_SUN_RAY), &solid_angle);
  const RGBF light_sa = c_mul(sky_sun_color(sky, sky_pos, dir_sa), c_splat(volume_phase_evaluate(s, c, dir_sa) * 1.0f));
  const float target_bsdf = c_importance(light_bsdf), target_sa = c_importance(light_sa);
  const float mis_bsdf = solid_angle / (volume_phase_evaluate(s, c, dir_bsdf) * solid_angle + 1.0f);
  const float mis_sa = solid_angle / (volume_phase_evaluate(s, c, dir_sa) * solid_angle + 1.0f);
  const float weight_bsdf = target_bsdf * mis_bsdf, weight_sa = target_sa * mis_sa;
  const float sum_weights = weight_bsdf + weight_sa;
  if (sum_weights == 0.0f) return false;
  float target;
  RGBF light;
  if (rnd1(smp, RT_VOL_SUN_RESAMPLING) * sum_weights < weight_bsdf) { *dir_out = dir_bsdf; target = target_bsdf; light = light_bsdf; }
  else { *dir_out = dir_sa; target = target_sa; light = light_sa; }
  light = c_scale(light, sum_weights / target);
  if (target == 0.0f) return false;
  if (c_importance(light) == 0.0f) return false;
  /* volume transmittance towards the sun (direct_lighting.cuh:104-108) */
  *light_out = c_mul(light, volume_transmittance(s, c->vol.type, c->position, *dir_out, FLT_MAX));
  return true;
}

/* ---- bridges (light_bridges.cuh) ---- */
static inline vec3 bridges_phase_sample(vec3 ray, float2_t r_dir) { return phase_sample_basis(hg_phase_sample(BRIDGES_HG_G, r_dir.x), r_dir.y, ray); }
static inline Quat bridges_compute_rotation(vec3 initial_vertex, vec3 light_point, vec3 end_vertex) { /* :16-52 */
  const vec3 target_dir = v_norm(v_sub(light_point, initial_vertex)), actual_dir = v_norm(v_sub(end_vertex, initial_vertex));
  const float dot = v_dot(actual_dir, target_dir);
  Quat r;
  if (dot > 0.999f) { r.x = 0.0f; r.y = 0.0f; r.z = 0.0f; r.w = 1.0f; return r; }
  if (dot < -0.999f) { r.x = 1.0f; r.y = 0.0f; r.z = 0.0f; r.w = 0.0f; return r; }
  const vec3 cr = v_cross(actual_dir, target_dir);
  r.x = cr.x; r.y = cr.y; r.z = cr.z; r.w = 1.0f + dot;
  const float scale = o_rsqrt(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w); /* normalize_quaternion, math.cuh:353-364 */
  r.x *= scale; r.y *= scale; r.z *= scale; r.w *= scale;
  return r;
}
static inline Quat16 quaternion_pack16(Quat q) { /* math.cuh:1687-1696: the inverse is stored */
  Quat16 d;
  d.x = (uint16_t) (((1.0f - q.x) * 0x7FFF) + 0.5f); d.y = (uint16_t) (((1.0f - q.y) * 0x7FFF) + 0.5f);
  d.z = (uint16_t) (((1.0f - q.z) * 0x7FFF) + 0.5f); d.w = (uint16_t) (((1.0f + q.w) * 0x7FFF) + 0.5f);
  return d;
}
static inline float bridges_log_factorial(uint32_t vertex_count) { /* :54-65, Ramanujan */
  if (vertex_count == 1) return 0.0f;
  const float n = (float) (vertex_count - 1);
  const float t0 = n * o_log(n);
  const float t1 = (1.0f / 6.0f) * o_log(n * (1.0f + 4.0f * n * (1.0f + 2.0f * n)));
  const float t2 = 0.5f * o_log(O_PI);
  return t0 + t1 + t2 - n;
}
static inline float bridges_vertex_count_importance(const float* lut_all, uint32_t vertex_count, float effective_dist) { /* :67-108 */
  const float* lut = lut_all + (vertex_count - 1) * 21;
  const float min_dist = lut[0], center_dist = lut[1], max_dist = lut[2];
  if (effective_dist > max_dist) return 0.0f;
  if (effective_dist < min_dist) return lut[3] * effective_dist / min_dist;
  const bool low = effective_dist < center_dist;
  const float low_dist = low ? min_dist : center_dist, high_dist = low ? center_dist : max_dist;
  const float step = (high_dist - low_dist) * 0.25f;
  const uint32_t step_id = (uint32_t) ((effective_dist - low_dist) / step);
  const float floor_dist = low_dist + step_id * step;
  const uint32_t index = low ? (3 + 2 * step_id) : (3 + 2 * (step_id + 4));
  const float y0 = lut[index], dy0 = lut[index + 1], y1 = lut[index + 2], dy1 = lut[index + 3];
  const float t = o_saturate((effective_dist - floor_dist) / step);
  const float t2 = t * t, t3 = t2 * t;
  const float h00 = 2.0f * t3 - 3.0f * t2 + 1.0f, h10 = t3 - 2.0f * t2 + t, h01 = -2.0f * t3 + 3.0f * t2, h11 = t3 - t2;
  return h00 * y0 + h10 * step * dy0 + h01 * y1 + h11 * step * dy1;
}
static inline uint32_t bridges_sample_vertex_count(const OracleScene* s, const OVolume* vol, float light_dist, uint32_t seed, const Sampler* smp, float* pdf) { /* :110-140 */
  const float effective_dist = light_dist * vol->scattering;
  RISReservoir rv = ris_init(rnd1(smp, RT_BRIDGE_VERTEX_COUNT + seed));
  const uint32_t max_num_vertices = (s->bridge_max_num_vertices < BRIDGES_MAX_VERTEX_COUNT) ? s->bridge_max_num_vertices : BRIDGES_MAX_VERTEX_COUNT;
  uint32_t selected = max_num_vertices - 1;
  for (uint32_t vc = 0; vc < max_num_vertices; vc++) {
    const float importance = bridges_vertex_count_importance(s->bridge_lut, vc + 1, effective_dist);
    if (ris_add(&rv, importance, 1.0f)) selected = vc;
  }
  *pdf = (rv.sum_weight > 0.0f) ? rv.selected_target / rv.sum_weight : 1.0f; /* ris_reservoir_get_sampling_prob, ris.cuh */
  return 1 + selected;
}
/* :142-222 */
static inline RGBF bridges_sample_bridge(const OracleScene* s, const VolCtx* c, vec3 light_point, vec3 initial_vertex, uint32_t seed, const Sampler* smp, float* path_pdf,
                                         vec3* end_vertex, float* scale) {
  const vec3 light_vector = v_sub(light_point, initial_vertex);
  const float target_scale = v_len(light_vector);
  float vertex_count_pdf;
  const uint32_t vertex_count = bridges_sample_vertex_count(s, &c->vol, target_scale, seed, smp, &vertex_count_pdf);
  vec3 current_vertex = initial_vertex, current_direction = v_norm(light_vector);
  float sum_dist = 0.0f;
  {
    const float dist = -o_log(rnd1(smp, RT_BRIDGE_DISTANCE + seed * LIGHT_GEO_MAX_BRIDGE_LENGTH + 0));
    current_vertex = v_add(current_vertex, v_scale(current_direction, dist));
    sum_dist += dist;
  }
  for (uint32_t i = 1; i < vertex_count; i++) {
    current_direction = bridges_phase_sample(current_direction, rnd2(smp, RT_BRIDGE_PHASE + seed * LIGHT_GEO_MAX_BRIDGE_LENGTH + i));
    const float dist = -o_log(rnd1(smp, RT_BRIDGE_DISTANCE + seed * LIGHT_GEO_MAX_BRIDGE_LENGTH + i));
    current_vertex = v_add(current_vertex, v_scale(current_direction, dist));
    sum_dist += dist;
  }
  const float actual_scale = v_len(v_sub(current_vertex, initial_vertex));
  if (actual_scale == 0.0f) { *path_pdf = 0.0f; return c_splat(0.0f); }
  *scale = target_scale / actual_scale;
  sum_dist *= *scale;
  *end_vertex = current_vertex;
  const RGBF sc = c->vol.scat, ab = c->vol.absorb;
  const RGBF w = c3(o_exp(vertex_count * o_log(sc.r) - sum_dist * (sc.r + ab.r)), o_exp(vertex_count * o_log(sc.g) - sum_dist * (sc.g + ab.g)),
                    o_exp(vertex_count * o_log(sc.b) - sum_dist * (sc.b + ab.b)));
  const float log_path_pdf = bridges_log_factorial(vertex_count) - vertex_count * o_log(sum_dist);
  *path_pdf = vertex_count_pdf * o_exp(log_path_pdf) * target_scale * target_scale * target_scale;
  return w;
}
/* :224-266 */
static inline vec3 bridges_sample_initial_vertex(const VolCtx* c, vec3 point_on_light, const Sampler* smp, uint32_t output_id, RGBF* attenuation, float* pdf) {
  float random_intersection = rnd1(smp, RT_LIGHT_GEO_INITIAL_VERTEX + output_id);
  const vec3 PO = v_sub(point_on_light, c->position);
  const float dist_to_light = fmaxf(-v_dot(PO, c->V), 0.0f);
  const float forward_prob = (dist_to_light < c->max_dist) ? BRIDGES_FORWARD_PROB : 1.0f;
  float max_dist, t_offset;
  if (random_intersection < forward_prob) {
    random_intersection = random_intersection / forward_prob;
    max_dist = o_clamp(dist_to_light, 0.0f, c->max_dist);
    t_offset = 0.0f;
    *pdf = forward_prob;
  }
  else {
    random_intersection = (random_intersection - forward_prob) / (1.0f - forward_prob);
    max_dist = c->max_dist - dist_to_light;
    t_offset = dist_to_light;
    *pdf = 1.0f - forward_prob;
  }
  const float t = t_offset + volume_sample_bounded(&c->vol, max_dist, random_intersection);
  *attenuation = c3(o_exp(-t * (c->vol.absorb.r + c->vol.scat.r)) * c->vol.scat.r, o_exp(-t * (c->vol.absorb.g + c->vol.scat.g)) * c->vol.scat.g,
                    o_exp(-t * (c->vol.absorb.b + c->vol.scat.b)) * c->vol.scat.b);
  *pdf *= volume_sample_bounded_pdf(&c->vol, max_dist, t - t_offset);
  return v_add(c->position, v_scale(c->V, -t));
}
/* light_triangle.cuh:209-243 */
static inline vec3 light_triangle_sample_bridges(const TriLight* t, float2_t random) {
  const float r1 = sqrtf(random.x), r2 = random.y;
  const float u = 1.0f - r1, v = r1 * r2;
  return v_add(t->vertex, v_add(v_scale(t->edge1, u), v_scale(t->edge2, v)));
}
static inline bool light_triangle_finalize_bridges(TriLight* t, const uint32_t uvp[3], vec3 origin, vec3 point_on_light, vec3* ray, float* dist, float* area) {
  const vec3 cr = v_cross(t->edge1, t->edge2);
  *area = v_len(cr) * 0.5f;
  *ray = v_sub(point_on_light, origin);
  if (!t->bidirectional && v_dot(*ray, cr) >= 0.0f) { *dist = FLT_MAX; return false; }
  *ray = v_norm(*ray);
  return light_triangle_finalize_dist(t, uvp, origin, *ray, dist);
}

typedef struct { uint32_t light_id; RGBF light_color; uint32_t seed; Quat rotation; float scale; } BridgeSample; /* LightSampleResult<VOLUME>, light_common.cuh:51-58 */

/* bridges_sample, light_bridges.cuh:268-350: one candidate of the volume's light resampling. `target`/`weight` as the reference leaves them
 * on every early return: the reservoir sees (0, 1) once the light point was finalised, and garbage-free zeros before (declared, never read). */
static inline BridgeSample bridges_sample(const OracleScene* s, const VolCtx* c, TriLight* light, uint32_t light_id, const uint32_t uvp[3], const Sampler* smp, uint32_t output_id,
                                          float* target, float* weight) {
  BridgeSample res;
  res.light_id = LIGHT_ID_INVALID; res.light_color = c_splat(0.0f); res.seed = 0; res.rotation.x = res.rotation.y = res.rotation.z = 0.0f; res.rotation.w = 1.0f; res.scale = 0.0f;
  *target = 0.0f; *weight = 1.0f;
  const vec3 point_on_light = light_triangle_sample_bridges(light, rnd2(smp, RT_BRIDGE_LIGHT_POINT + output_id));
  RGBF initial_attenuation; float initial_pdf;
  const vec3 initial_vertex = bridges_sample_initial_vertex(c, point_on_light, smp, output_id, &initial_attenuation, &initial_pdf);
  if (initial_pdf == 0.0f || c_importance(initial_attenuation) == 0.0f) return res;
  vec3 light_dir; float area, light_dist;
  light_triangle_finalize_bridges(light, uvp, initial_vertex, point_on_light, &light_dir, &light_dist, &area);
  if (light_dist == FLT_MAX || area < O_EPS) return res;
  RGBF light_color = c_mul(light_get_color(s, light), initial_attenuation);
  if (c_importance(light_color) == 0.0f) return res;
  const vec3 light_point = v_add(initial_vertex, v_scale(light_dir, light_dist));
  if (light_point.y < c->vol.min_height || light_point.y > c->vol.max_height) return res;
  float sample_weight = area / initial_pdf;
  float path_pdf, path_scale; vec3 path_end;
  RGBF path_weight = bridges_sample_bridge(s, c, light_point, initial_vertex, output_id, smp, &path_pdf, &path_end, &path_scale);
  if (path_pdf == 0.0f) return res;
  sample_weight *= 1.0f / path_pdf;
  const Quat rot = bridges_compute_rotation(initial_vertex, light_point, path_end);
  const vec3 rotated = q_apply(rot, light_dir);
  const float cos_angle = -v_dot(rotated, c->V);
  light_color = c_scale(light_color, hg_phase(cos_angle, BRIDGES_HG_G));
  path_weight = c_mul(path_weight, light_color);
  *target = c_importance(path_weight); *weight = sample_weight;
  res.light_id = light_id; res.light_color = path_weight; res.rotation = rot; res.scale = path_scale; res.seed = output_id;
  return res;
}

/* light_tree_importance<VOLUME>, light_tree.cuh:91-122 (absorption 0: the transmittance factor is exp(-0 * depth) = 1, kept as a multiplication) */
static inline float light_tree_importance_volume(const VolCtx* c, float power, vec3 mean, float std_dev) {
  const vec3 PO = v_sub(mean, c->position);
  const float dist_along_ray = -v_dot(PO, c->V);
  const float clamped = o_clamp(dist_along_ray, 0.0f, c->max_dist);
  const vec3 perp = v_sub(PO, v_scale(c->V, clamped));
  const float perp_sq = v_dot(perp, perp);
  const float falloff = 1.0f / (perp_sq + std_dev);
  const float variance = std_dev * std_dev;
  const float transmittance_depth = fmaxf(perp_sq + clamped * clamped - variance, 0.0f);
  const float transmittance = o_exp(-c->vol.max_absorption * transmittance_depth);
  const float scattering = 1.0f - o_exp(-c->vol.scattering * (variance + clamped));
  return power * falloff * transmittance * scattering;
}

#endif
