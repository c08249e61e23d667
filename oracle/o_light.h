/*
 * ORACLE (test infrastructure, not product): next-event estimation.
 * Follows /root/reference/src/luminary/device/cuda/{ris,light_tree,light_triangle,light,light_bsdf,mis}.cuh.
 */
#ifndef ORACLE_O_LIGHT_H
#define ORACLE_O_LIGHT_H

#include "o_bsdf.h"
#include "oracle.h"

#define LIGHT_TREE_NUM_OUTPUTS 8
#define LIGHT_GEO_MAX_SAMPLES 8
#define LIGHT_ID_INVALID 0xFFFFFFFFu

/* ---- scene accessors (memory.cuh:373-411, 506-524) ---- */
typedef struct { vec3 pos; uint32_t normal; } OVertex;
static inline OVertex scene_vertex(const OracleScene* s, uint32_t mesh, uint32_t tri, uint32_t k) {
  const float* p = s->vertices + ((size_t) (s->mesh_tri_offset[mesh] + tri) * 3 + k) * 4;
  OVertex v;
  v.pos = v3(p[0], p[1], p[2]);
  v.normal = f2u(p[3]);
  return v;
}
static inline const uint32_t* scene_tritex(const OracleScene* s, uint32_t mesh, uint32_t tri) {
  return s->tri_tex + (size_t) (s->mesh_tri_offset[mesh] + tri) * 4;
}
static inline OTransform scene_transform(const OracleScene* s, uint32_t inst) {
  const float* p = s->instance_transforms + (size_t) inst * 8;
  OTransform t;
  t.translation = v3(p[0], p[1], p[2]);
  t.scale = v3(p[3], p[4], p[5]);
  const uint32_t a = f2u(p[6]), b = f2u(p[7]);
  t.rotation.x = (uint16_t) (a & 0xFFFF); t.rotation.y = (uint16_t) (a >> 16);
  t.rotation.z = (uint16_t) (b & 0xFFFF); t.rotation.w = (uint16_t) (b >> 16);
  return t;
}
static inline OMaterial scene_material(const OracleScene* s, uint32_t id) { return material_load((const OMaterialC*) (s->materials + (size_t) id * 16)); }
/* ---- textures (cuda/texture_utils.cuh:20-45): normalised coordinates, wrap addressing, linear filter, mip level 0
 * (texture.c:77-90, device_texture.c:247-268). Exact float lerps instead of the texture unit's 8-bit weights, o_pow for the gamma.
 * `def` is returned for an invalid handle. ---- */
typedef struct { float x, y, z, w; } float4_t;
static inline float4_t f4(float x, float y, float z, float w) { float4_t r = {x, y, z, w}; return r; }
static inline float4_t texel_unpack(uint32_t t) {
  return f4((t & 0xFFu) * (1.0f / 255.0f), ((t >> 8) & 0xFFu) * (1.0f / 255.0f), ((t >> 16) & 0xFFu) * (1.0f / 255.0f), (t >> 24) * (1.0f / 255.0f));
}
static inline float4_t texture_load(const OracleScene* s, uint32_t tex, UV uv, bool apply_gamma, float4_t def) {
  if (tex >= s->num_textures) return def;
  const uint32_t* t = s->texture_table + 4 * (size_t) tex;
  const int w = (int) t[1], h = (int) t[2];
  const float u = uv.u, v = 1.0f - uv.v; /* flip_v */
  const float xb = (u - floorf(u)) * (float) w - 0.5f, yb = (v - floorf(v)) * (float) h - 0.5f;
  const float xf = floorf(xb), yf = floorf(yb);
  const float ax = xb - xf, ay = yb - yf;
  int x0 = (int) xf, y0 = (int) yf, x1 = x0 + 1, y1 = y0 + 1;
  if (x0 < 0) x0 += w;
  if (y0 < 0) y0 += h;
  if (x1 >= w) x1 -= w;
  if (y1 >= h) y1 -= h;
  const uint32_t* base = s->texels + t[0];
  const float4_t c00 = texel_unpack(base[x0 + y0 * w]), c10 = texel_unpack(base[x1 + y0 * w]);
  const float4_t c01 = texel_unpack(base[x0 + y1 * w]), c11 = texel_unpack(base[x1 + y1 * w]);
  float4_t r;
  { const float top = c00.x + ax * (c10.x - c00.x), bot = c01.x + ax * (c11.x - c01.x); r.x = top + ay * (bot - top); }
  { const float top = c00.y + ax * (c10.y - c00.y), bot = c01.y + ax * (c11.y - c01.y); r.y = top + ay * (bot - top); }
  { const float top = c00.z + ax * (c10.z - c00.z), bot = c01.z + ax * (c11.z - c01.z); r.z = top + ay * (bot - top); }
  { const float top = c00.w + ax * (c10.w - c00.w), bot = c01.w + ax * (c11.w - c01.w); r.w = top + ay * (bot - top); }
  const float gamma = u2f(t[3]);
  if (apply_gamma && gamma != 1.0f) { r.x = o_pow(r.x, gamma); r.y = o_pow(r.y, gamma); r.z = o_pow(r.z, gamma); } /* never the alpha */
  return r;
}
/* cuda/math.cuh:246-253, cuda/memory.cuh:414-425 */
static inline UV triangle_uv(const uint32_t* tri_tex, float2_t coords) {
  const UV a = uv_unpack(tri_tex[0]), b = uv_unpack(tri_tex[1]), c = uv_unpack(tri_tex[2]);
  UV r = {a.u + coords.x * (b.u - a.u) + coords.y * (c.u - a.u), a.v + coords.x * (b.v - a.v) + coords.y * (c.v - a.v)};
  return r;
}
/* optix_get_albedo_for_shadowing, optix_common.cuh:48-65 */
static inline RGBAF albedo_for_shadowing(const OracleScene* s, const OMaterial* m, const uint32_t* tri_tex, float2_t coords) {
  RGBAF albedo = m->albedo;
  if (m->albedo_tex != TEXTURE_NONE) {
    if (m->albedo_tex >= s->num_textures) { RGBAF d = {0.9f, 0.9f, 0.9f, 1.0f}; return d; }
    const float4_t t = texture_load(s, m->albedo_tex, triangle_uv(tri_tex, coords), true, f4(0.0f, 0.0f, 0.0f, 0.0f));
    albedo.r = t.x; albedo.g = t.y; albedo.b = t.z; albedo.a = t.w;
  }
  return albedo;
}

static inline OLuts scene_luts(const OracleScene* s) {
  OLuts l = {s->lut_conductor, s->lut_glossy, s->lut_dielectric, s->lut_dielectric_inv};
  return l;
}

/* ---- RIS (ris.cuh:22-158) ---- */
typedef struct { float sum_weight, selected_target, random; } RISReservoir;
static inline RISReservoir ris_init(float random) { RISReservoir r = {0.0f, 0.0f, random}; return r; }
static inline void ris_reset(RISReservoir* r) { r->sum_weight = 0.0f; r->selected_target = 0.0f; }
static inline bool ris_add(RISReservoir* r, float target, float sampling_weight) {
  const float w = target * sampling_weight;
  r->sum_weight += w;
  if (w == 0.0f) return false;
  const float prob = w / r->sum_weight;
  const bool acc = r->random < prob;
  r->selected_target = acc ? target : r->selected_target;
  const float shift = acc ? 0.0f : prob, scale = acc ? prob : 1.0f - prob;
  r->random = rng_saturate((r->random - shift) / scale);
  return acc;
}
static inline float ris_sampling_weight(const RISReservoir* r) { return (r->selected_target > 0.0f) ? r->sum_weight / r->selected_target : 0.0f; }

/* ris.cuh:138-148. The lanes of one stream divide by one of two numbers per sample; the reciprocals are formed once per sample
 * and multiplied in (the reference is built with --use_fast_math, where the division is a reciprocal multiply as well). */
typedef struct { float selected_target, random; } RISLane;
static inline bool ris_lane_add(RISLane* l, float prob, float target, float inv_accept, float inv_reject) {
  const bool acc = l->random < prob;
  l->selected_target = acc ? target : l->selected_target;
  const float shifted = acc ? l->random : l->random - prob;
  l->random = rng_saturate(shifted * (acc ? inv_accept : inv_reject));
  return acc;
}

/* ---- light tree (light_tree.cuh:71-320) ---- */
typedef struct { uint32_t is_light, child_index, probability; } LTCont; /* bitfield 1/8/20 in the reference */
typedef struct { LTCont data[LIGHT_TREE_NUM_OUTPUTS]; float root_sum; } LTWork;
typedef struct { uint32_t light_id; float weight; } LTResult;

/* light_tree.cuh:71-89 (geometry) */
static inline float light_tree_importance(const GeoCtx* g, float power, vec3 mean, float std_dev) {
  const vec3 PO = v_sub(mean, g->position);
  const float dist_sq = v_dot(PO, PO);
  const float variance = std_dev * std_dev;
  const float inv = 1.0f / (dist_sq + variance);
  float r = power * inv;
  if ((g->params.flags & MAT_SUBSTRATE_MASK) == MAT_TRANSLUCENT) return r;
  const float t = variance * inv;
  const float NdotL = o_saturate(v_dot(PO, g->normal) * sqrtf(inv));
  return r * (NdotL * (1.0f - t) + t);
}
/* The tree is walked for a surface vertex (light_tree_importance<GEOMETRY>) or for a ray segment through a volume (<VOLUME>, light_tree.cuh:91-122,
 * defined in o_volume.h); the two contexts draw from different random sets (material.cuh:60-63, :78-81). */
struct VolCtx;
static inline float light_tree_importance_volume(const struct VolCtx* c, float power, vec3 mean, float std_dev);
typedef struct { const GeoCtx* geo; const struct VolCtx* vol; uint32_t rt_prepass, rt_postpass; const vec3* particle_position; } LTQuery;
static inline LTQuery lt_query_geometry(const GeoCtx* g) { LTQuery q = {g, NULL, RT_LIGHT_GEO_TREE_PREPASS, RT_LIGHT_GEO_TREE_POSTPASS, NULL}; return q; }
static inline float lt_importance(const LTQuery* q, float power, vec3 mean, float std_dev) {
  if (q->particle_position) { /* light_tree_importance<PARTICLE>, light_tree.cuh:124-131 */
    const vec3 PO = v_sub(mean, *q->particle_position);
    const float dist_sq = v_dot(PO, PO) + std_dev * std_dev;
    return power / dist_sq;
  }
  return q->vol ? light_tree_importance_volume(q->vol, power, mean, std_dev) : light_tree_importance(q->geo, power, mean, std_dev);
}
/* light_tree.cuh:133-161: rel_* arrays are 8 entries each; power is u16 in root sections, u8 in nodes */
static inline float lt_child_importance(const LTQuery* g, const uint8_t* mx, const uint8_t* my, const uint8_t* mz, const uint8_t* sd, uint32_t power_q,
                                        vec3 base, vec3 exp, float exp_v, uint32_t i) {
  if (power_q == 0) return 0.0f;
  const float power = (float) power_q;
  const float std_dev = sd[i] * exp_v;
  const vec3 mean = v_add(v_mul(v3(mx[i], my[i], mz[i]), exp), base);
  return fmaxf(lt_importance(g, power, mean, std_dev), 0.0f);
}
/* light_tree.cuh:176-189 */
static inline LTCont lt_cont_pack(uint32_t child_index, float probability, bool is_light) {
  LTCont c;
  c.is_light = is_light ? 1 : 0;
  c.child_index = child_index & 0xFF;
  uint32_t q = 0;
  if (probability > 0.0f) { q = (uint32_t) ((0xFFFFF * probability) + 0.5f); if (q < 1) q = 1; }
  c.probability = q & 0xFFFFF;
  return c;
}
static inline float lt_cont_prob(LTCont c) { return c.probability * (1.0f / 0xFFFFF) * LIGHT_TREE_NUM_OUTPUTS; }

/* light_tree.cuh:191-255. Root header (device_utils.h:304-317): u16 x,y,z,num_root_lights,power_normalization; u8 num_sections,pad; s8 exp x,y,z,std */
static inline LTWork light_tree_prepass(const OracleScene* s, const LTQuery* g, const Sampler* smp) {
  const uint8_t* root = s->light_tree_root;
  uint16_t h16[5]; memcpy(h16, root, 10);
  const uint32_t num_sections = root[10];
  const int8_t ex = (int8_t) root[12], ey = (int8_t) root[13], ez = (int8_t) root[14], es = (int8_t) root[15];
  const uint32_t num_root_lights = h16[3];

  float agg_sum = 0.0f;
  RISLane lane[LIGHT_TREE_NUM_OUTPUTS];
  uint32_t selected[LIGHT_TREE_NUM_OUTPUTS];
  for (uint32_t i = 0; i < LIGHT_TREE_NUM_OUTPUTS; i++) {
    lane[i].random = rnd1(smp, g->rt_prepass + i);
    lane[i].selected_target = 0.0f;
    selected[i] = 0;
  }
  const vec3 base = v3(bfloat_unpack(h16[0]), bfloat_unpack(h16[1]), bfloat_unpack(h16[2]));
  const vec3 exp = v3(o_exp2i(ex), o_exp2i(ey), o_exp2i(ez));
  const float exp_v = o_exp2i(es);
  float sum = 0.0f;
  for (uint32_t sec = 0; sec < num_sections; sec++) {
    const uint8_t* sp = root + 16 + 48 * sec;
    uint16_t pw[8]; memcpy(pw, sp + 32, 16);
    for (uint32_t c = 0; c < 8; c++) {
      const float target = lt_child_importance(g, sp, sp + 8, sp + 16, sp + 24, pw[c], base, exp, exp_v, c);
      agg_sum += target; /* ris_aggregator_add_sample with sampling weight 1 */
      const float prob = (target > 0.0f) ? target / agg_sum : 0.0f;
      if (prob == 0.0f) continue;
      sum += target;
      const float inv_accept = 1.0f / prob, inv_reject = 1.0f / (1.0f - prob);
      for (uint32_t l = 0; l < LIGHT_TREE_NUM_OUTPUTS; l++)
        if (ris_lane_add(&lane[l], prob, target, inv_accept, inv_reject)) selected[l] = sec * 8 + c;
    }
  }
  LTWork w;
  w.root_sum = sum * (bfloat_unpack(h16[4]) / 0xFFFF);
  for (uint32_t l = 0; l < LIGHT_TREE_NUM_OUTPUTS; l++) {
    const bool is_light = selected[l] < num_root_lights;
    const uint32_t index = is_light ? selected[l] : selected[l] - num_root_lights;
    const float p = (agg_sum > 0.0f) ? (lane[l].selected_target / agg_sum) : 0.0f;
    w.data[l] = lt_cont_pack(index, p, is_light);
  }
  return w;
}

/* light_tree.cuh:257-320. Node (device_utils.h:283-302): u16 x,y,z,pad; s8 exp x,y,z,std; u8 num_lights,pad; u16 pad; u32 child_ptr, light_ptr; 5 x u8[8] */
static inline LTResult light_tree_postpass(const OracleScene* s, const LTQuery* g, const Sampler* smp, uint32_t lane_id, const LTWork* work) {
  const LTCont cont = work->data[lane_id];
  const float cp = lt_cont_prob(cont);
  LTResult res;
  res.light_id = LIGHT_ID_INVALID;
  res.weight = (cp > 0.0f) ? 1.0f / cp : 0.0f;
  if (cp == 0.0f) return res;
  if (cont.is_light) { res.light_id = cont.child_index; return res; }
  const uint8_t* node = s->light_tree_nodes + 64 * (size_t) cont.child_index;
  RISReservoir rv = ris_init(rnd1(smp, g->rt_postpass + lane_id));
  while (res.light_id == LIGHT_ID_INVALID) {
    uint16_t b16[3]; memcpy(b16, node, 6);
    const int8_t ex = (int8_t) node[8], ey = (int8_t) node[9], ez = (int8_t) node[10], es = (int8_t) node[11];
    const uint32_t num_lights = node[12];
    uint32_t child_ptr, light_ptr; memcpy(&child_ptr, node + 16, 4); memcpy(&light_ptr, node + 20, 4);
    const vec3 base = v3(bfloat_unpack(b16[0]), bfloat_unpack(b16[1]), bfloat_unpack(b16[2]));
    const vec3 exp = v3(o_exp2i(ex), o_exp2i(ey), o_exp2i(ez));
    const float exp_v = o_exp2i(es);
    uint32_t sel = 0xFF;
    for (uint32_t c = 0; c < 8; c++) {
      const float target = lt_child_importance(g, node + 24, node + 32, node + 40, node + 48, node[56 + c], base, exp, exp_v, c);
      if (ris_add(&rv, target, 1.0f)) sel = c;
    }
    if (sel == 0xFF) break;
    res.weight *= ris_sampling_weight(&rv);
    if (sel < num_lights) { res.light_id = light_ptr + sel; break; }
    node = s->light_tree_nodes + 64 * (size_t) (child_ptr + (sel - num_lights));
    ris_reset(&rv);
  }
  return res;
}

/* ---- triangle lights (light_common.cuh:69-79, light_triangle.cuh) ---- */
typedef struct { vec3 vertex, edge1, edge2; UV tex; uint16_t material_id; bool bidirectional; } TriLight;

/* light_triangle.cuh:37-72 */
static inline TriLight light_triangle_init(const OracleScene* s, uint32_t instance_id, uint32_t tri_id, uint32_t uv_packed[3]) {
  const uint32_t mesh = s->instance_mesh_ids[instance_id];
  const OTransform tr = scene_transform(s, instance_id);
  const vec3 p0 = scene_vertex(s, mesh, tri_id, 0).pos, p1 = scene_vertex(s, mesh, tri_id, 1).pos, p2 = scene_vertex(s, mesh, tri_id, 2).pos;
  const uint32_t* tt = scene_tritex(s, mesh, tri_id);
  TriLight t;
  t.vertex = t_apply(tr, p0);
  t.edge1 = t_rel(tr, v_sub(p1, p0));
  t.edge2 = t_rel(tr, v_sub(p2, p0));
  uv_packed[0] = tt[0]; uv_packed[1] = tt[1]; uv_packed[2] = tt[2];
  t.material_id = (uint16_t) (tt[3] & 0xFFFF);
  const uint8_t flags = (uint8_t) (s->materials[(size_t) t.material_id * 16] & 0xFF);
  t.bidirectional = (flags & DMAT_BIDIRECTIONAL_EMISSION) != 0;
  t.tex.u = 0.0f; t.tex.v = 0.0f;
  return t;
}
/* light_triangle.cuh:74-92 */
static inline bool light_triangle_finalize_dist(TriLight* t, const uint32_t uv_packed[3], vec3 origin, vec3 ray, float* dist) {
  float2_t c;
  *dist = tri_intersect(t->vertex, t->edge1, t->edge2, origin, ray, &c);
  if (*dist == FLT_MAX) return false;
  const UV a = uv_unpack(uv_packed[0]), b = uv_unpack(uv_packed[1]), d = uv_unpack(uv_packed[2]);
  t->tex.u = a.u + c.x * (b.u - a.u) + c.y * (d.u - a.u);
  t->tex.v = a.v + c.x * (b.v - a.v) + c.y * (d.v - a.v);
  return true;
}
/* light_triangle.cuh:94-108 */
static inline float light_triangle_solid_angle(const TriLight* t, vec3 origin) {
  const vec3 a = v_norm(v_sub(t->vertex, origin)), b = v_norm(v_sub(v_add(t->vertex, t->edge1), origin)),
             c = v_norm(v_sub(v_add(t->vertex, t->edge2), origin));
  const float G0 = fabsf(v_dot(v_cross(a, b), c)), G1 = v_dot(a, c) + v_dot(b, c), G2 = 1.0f + v_dot(a, b);
  return 2.0f * o_atan2(G0, G1 + G2);
}
/* light_triangle.cuh:110-112 */
static inline float light_triangle_area(const TriLight* t) { return v_len(v_cross(t->edge1, t->edge2)) * 0.5f; }
static inline bool nonfinite(float a) { return isnan(a) || isinf(a); }
/* light_triangle.cuh:114-157 (Peters 2021) */
static inline bool light_triangle_sample_solid_angle(vec3 origin, vec3 vertex, vec3 e1, vec3 e2, float2_t rnd, bool bidirectional, vec3* ray, float* solid_angle) {
  const vec3 a = v_norm(v_sub(vertex, origin)), b = v_norm(v_sub(v_add(vertex, e1), origin)), c = v_norm(v_sub(v_add(vertex, e2), origin));
  const float G0s = v_dot(v_cross(a, b), c);
  if (!bidirectional && (G0s >= 0.0f)) return false;
  const float G0 = fabsf(G0s), G1 = v_dot(a, c) + v_dot(b, c), G2 = 1.0f + v_dot(a, b);
  *solid_angle = 2.0f * o_atan2(G0, G1 + G2);
  if (nonfinite(*solid_angle) || *solid_angle < 1e-7f) return false;
  const float ssa = rnd.x * *solid_angle;
  float sh, ch; o_sincos(0.5f * ssa, &sh, &ch);
  const vec3 r = v_add(v_scale(a, G0 * ch - G1 * sh), v_scale(c, G2 * sh));
  const vec3 c_t = v_sub(v_scale(r, 2.0f * v_dot(a, r) / v_dot(r, r)), a);
  const float s2 = v_dot(b, c_t);
  const float sv = (1.0f - rnd.y) + rnd.y * s2;
  const float t = sqrtf(fmaxf((1.0f - sv * sv) / (1.0f - s2 * s2), 0.0f));
  *ray = v_norm(v_add(v_scale(b, sv - t * s2), v_scale(c_t, t)));
  if (nonfinite(ray->x) || nonfinite(ray->y) || nonfinite(ray->z)) return false;
  return true;
}
/* light_triangle.cuh:163-174: both halves are always evaluated (success &= ...) */
static inline bool light_triangle_finalize(TriLight* t, const uint32_t uv_packed[3], vec3 origin, float2_t rnd, vec3* ray, float* dist, float* solid_angle) {
  bool ok = true;
  *ray = v3(0.0f, 0.0f, 0.0f); *solid_angle = 0.0f;
  ok &= light_triangle_sample_solid_angle(origin, t->vertex, t->edge1, t->edge2, rnd, t->bidirectional, ray, solid_angle);
  if (!ok) { *dist = FLT_MAX; return false; } /* the second half cannot turn a failure into a success; its outputs are unused then */
  ok &= light_triangle_finalize_dist(t, uv_packed, origin, *ray, dist);
  return ok;
}
/* light_triangle.cuh:245-280 */
static inline RGBF light_get_color(const OracleScene* s, const TriLight* t) {
  const OMaterial m = scene_material(s, t->material_id);
  RGBF c = m.emission;
  if (m.luminance_tex != TEXTURE_NONE) {
    const float4_t e = texture_load(s, m.luminance_tex, t->tex, true, f4(0.0f, 0.0f, 0.0f, 0.0f));
    c = c_scale(c3(e.x, e.y, e.z), m.emission_scale);
  }
  if (c_any(c)) {
    const float alpha = (m.albedo_tex != TEXTURE_NONE) ? texture_load(s, m.albedo_tex, t->tex, true, f4(0.0f, 0.0f, 0.0f, 1.0f)).w : m.albedo.a;
    c = c_scale(c, alpha);
  }
  return c;
}

/* ---- BSDF-sampled light direction (light_bsdf.cuh) ---- */
typedef struct { vec3 ray; RGBF weight; float sampling_probability; bool is_refraction; } LightBSDFSample;
static inline float light_bsdf_sampling_roughness(float r) { return o_lerp(r, 1.0f, 0.04f); }
static inline float light_bsdf_rr_prob(float r) { return o_remap01(r, 0.5f, 0.1f); }

/* light_bsdf.cuh:24-102 */
static inline LightBSDFSample light_bsdf_get_sample(const OLuts* l, const GeoCtx* g, const Sampler* smp) {
  LightBSDFSample res;
  res.ray = v3(0.0f, 0.0f, 1.0f); res.weight = c_splat(0.0f); res.sampling_probability = 0.0f; res.is_refraction = false;
  const MatParams* p = &g->params;
  const Quat rot = q_rotation_to_z(g->normal);
  const vec3 Vl = q_apply(rot, g->V);
  const vec3 fnl = q_apply(rot, normal_unpack(g->face_normal));
  const vec3 up = v3(0.0f, 0.0f, 1.0f);
  const bool include_refraction = (p->flags & MAT_SUBSTRATE_MASK) == MAT_TRANSLUCENT;
  const uint32_t num_tech = 1 + (include_refraction ? 1 : 0);
  const float refr_prob = include_refraction ? 1.0f / num_tech : 0.0f;
  const float choice = rnd1(smp, RT_LIGHT_BSDF_CHOICE);
  const uint32_t tech_id = (uint32_t) (choice * num_tech);
  const bool refraction = (tech_id == 1 && include_refraction);
  const float roughness = mp_roughness(p);
  const float rrr = rnd1(smp, RT_LIGHT_BSDF_RR);
  const float rrp = light_bsdf_rr_prob(roughness);
  if (rrr >= rrp) return res;
  const float sr = light_bsdf_sampling_roughness(roughness);
  if (!refraction) {
    const vec3 m = microfacet_sample_normal(Vl, sr, rnd2(smp, RT_LIGHT_BSDF_DIRECTION));
    const vec3 ray = v_reflect(Vl, m);
    const BSDFRayCtx c = bsdf_sample_context(p, up, Vl, m, ray, false);
    const float pdf = microfacet_pdf(Vl, sr, c.NdotH, c.NdotV);
    res.weight = bsdf_evaluate_core(l, p, &c, HINT_GENERAL, ray, fnl, 1.0f / pdf);
    res.ray = ray; res.is_refraction = false;
    res.sampling_probability = (1.0f - refr_prob) * pdf;
  }
  else {
    const float ior = mp_ior(p);
    bool tot;
    const vec3 m = microfacet_refraction_sample_normal(Vl, sr, rnd2(smp, RT_LIGHT_BSDF_DIRECTION));
    const vec3 ray = refract_vector(Vl, m, ior, &tot);
    const BSDFRayCtx c = bsdf_sample_context(p, up, Vl, m, ray, !tot);
    const float pdf = microfacet_refraction_pdf(sr, c.NdotH, c.NdotV, c.HdotV, c.HdotL, ior);
    res.weight = bsdf_evaluate_core(l, p, &c, HINT_GENERAL, ray, fnl, 1.0f / pdf);
    res.ray = ray; res.is_refraction = !tot;
    res.sampling_probability = refr_prob * pdf;
  }
  res.weight = c_scale(res.weight, 1.0f / rrp);
  res.sampling_probability *= rrp;
  res.ray = v_norm(q_apply(q_inverse(rot), res.ray));
  return res;
}
/* light_bsdf.cuh:104-146 */
static inline float light_bsdf_get_probability(const GeoCtx* g, vec3 L) {
  const MatParams* p = &g->params;
  const Quat rot = q_rotation_to_z(g->normal);
  const vec3 Vl = v_norm(q_apply(rot, g->V)), Ll = v_norm(q_apply(rot, L));
  const bool include_refraction = (p->flags & MAT_SUBSTRATE_MASK) == MAT_TRANSLUCENT;
  const uint32_t num_tech = 1 + (include_refraction ? 1 : 0);
  const float refr_prob = include_refraction ? 1.0f / num_tech : 0.0f;
  const BSDFRayCtx c = bsdf_evaluate_analyze(p, v3(0.0f, 0.0f, 1.0f), Vl, Ll);
  const float roughness = mp_roughness(p);
  const float sr = light_bsdf_sampling_roughness(roughness);
  float prob;
  if (c.is_refraction) prob = refr_prob * microfacet_refraction_pdf(sr, c.NdotH, c.NdotV, c.HdotV, c.HdotL, mp_ior(p));
  else prob = (1.0f - refr_prob) * microfacet_pdf(Vl, sr, c.NdotH, c.NdotV);
  return prob * light_bsdf_rr_prob(roughness);
}

/* ---- MIS (mis.cuh:19-57) ---- */
static inline float mis_weight_base(float gi_pdf, float solid_angle, float power, float dist_sq, float root_sum) {
  const float dl_pdf = LIGHT_GEO_MAX_SAMPLES * (1.0f / solid_angle) * (power / dist_sq) * (1.0f / root_sum);
  return (dl_pdf > 0.0f) ? gi_pdf / (gi_pdf + dl_pdf) : 1.0f;
}
static inline float mis_weight_gi(vec3 origin, const TriLight* t, RGBF color, float dist, float gi_pdf, float root_sum) {
  if (root_sum == 0.0f) return 1.0f;
  const float area = light_triangle_area(t), sa = light_triangle_solid_angle(t, origin);
  return mis_weight_base(gi_pdf, sa, c_importance(color) * area, dist * dist, root_sum);
}
static inline float mis_weight_dl(const GeoCtx* g, vec3 L, const TriLight* t, RGBF color, float dist, float solid_angle, float root_sum) {
  const float power = c_importance(color) * light_triangle_area(t);
  return 1.0f - mis_weight_base(light_bsdf_get_probability(g, L), solid_angle, power, dist * dist, root_sum);
}

/* ---- light sampling (light.cuh:49-159) ---- */
typedef struct { uint32_t light_id; vec3 ray; RGBF light_color; float dist; float root_sum; } LightSample;

static inline LightSample light_sample(const OracleScene* s, const GeoCtx* g, const Sampler* smp) {
  const OLuts luts = scene_luts(s);
  const LTQuery query = lt_query_geometry(g);
  const LTWork work = light_tree_prepass(s, &query, smp);
  LightSample res;
  res.light_id = LIGHT_ID_INVALID;
  res.ray = v3(0.0f, 0.0f, 0.0f); res.light_color = c_splat(0.0f); res.dist = 0.0f;
  RISReservoir rv = ris_init(rnd1(smp, RT_LIGHT_GEO_RESAMPLING));
  for (uint32_t out = 0; out < LIGHT_TREE_NUM_OUTPUTS; out++) {
    const LTResult o = light_tree_postpass(s, &query, smp, out, &work);
    if (o.light_id == LIGHT_ID_INVALID) continue;
    const uint32_t inst = s->light_tri_handles[2 * o.light_id], tri = s->light_tri_handles[2 * o.light_id + 1];
    if (inst == g->instance_id && tri == g->tri_id) continue;
    uint32_t uvp[3];
    TriLight tl = light_triangle_init(s, inst, tri, uvp);
    /* light_evaluate_candidate, light.cuh:49-82 */
    const float2_t rr = rnd2(smp, RT_LIGHT_GEO_RAY + out);
    vec3 ray; float dist, sa;
    if (!light_triangle_finalize(&tl, uvp, g->position, rr, &ray, &dist, &sa)) continue;
    RGBF lc = light_get_color(s, &tl);
    bool is_refr;
    const RGBF bw = bsdf_evaluate(&luts, g, ray, HINT_GENERAL, &is_refr, 1.0f);
    const float mis = mis_weight_dl(g, ray, &tl, lc, dist, sa, work.root_sum);
    lc = c_scale(c_mul(lc, bw), mis);
    const float target = c_importance(lc);
    if (ris_add(&rv, target, o.weight * sa)) { res.light_id = o.light_id; res.ray = ray; res.light_color = lc; res.dist = dist; }
  }
  res.light_color = c_scale(res.light_color, ris_sampling_weight(&rv));
  res.root_sum = work.root_sum;
  return res;
}

#endif
