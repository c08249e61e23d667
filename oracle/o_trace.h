/*
 * ORACLE (test infrastructure, not product): ray queries.
 *
 * The reference delegates traversal to OptiX (closed source, "parity unpinned" by definition). What is
 * restated here are the observable semantics of its programs:
 *   closest hit   optix/optix_kernel_raytrace.cu:82-95, cuda/optix_anyhit.cuh:15-31, cuda/optix_closesthit.cuh:15-26
 *   shadow        cuda/optix_common.cuh:76-106, cuda/optix_anyhit.cuh:49-139, cuda/optix_closesthit.cuh:44-58
 *   light-BVH     cuda/optix_anyhit.cuh:145-205, cuda/direct_lighting.cuh:596-611
 * with the triangle test of cuda/math.cuh:1337-1358 on object-space rays (instance transform = S*R*v + T, math.cuh:459-489).
 * The world->object map is applied as a 3x4 matrix per instance, like the instance matrices the reference hands to OptiX
 * (device/optix_bvh.c:16-66): column j = transform_apply_relative_inv(e_j) (math.cuh:476-482), rows evaluated as
 * (a*x + b*y) + c*z on (origin - T) and on the direction. Distances along the ray are preserved by the affine map.
 *
 * Two intersectors give the same answers: brute force over every triangle, and a median-split BVH per mesh with a
 * conservative slab test. Order-independent tie-breaks (lowest t, then instance id, then triangle id) make the result a
 * function of the scene only, so any correct traversal (the HIP one included) must reproduce it exactly.
 */
#ifndef ORACLE_O_TRACE_H
#define ORACLE_O_TRACE_H

#include <stdlib.h>

#include "o_light.h"

#define HIT_TYPE_SKY 0xFFFFFFFEu
#define HIT_TYPE_INVALID 0xFFFFFFFFu
#define HIT_TYPE_TRIANGLE_ID_LIMIT 0x7FFFFFFFu

typedef struct { float lo[3], hi[3]; uint32_t left, right; uint32_t first, count; } OBvhNode; /* count > 0 => leaf */
typedef struct { OBvhNode* nodes; uint32_t num_nodes; uint32_t* tri_ids; } OBvh;
typedef struct {
  const OracleScene* scene;
  OBvh* mesh_bvh;  /* per mesh, NULL when brute force */
  OBvh light_bvh;  /* over world-space light triangles */
  OBvh particle_bvh; /* over the 2 x particles_count triangles of the particle unit cell */
  float* inst_inv; /* 12 per instance: rows (m_i0, m_i1, m_i2, T_i) of the world->object matrix */
  int use_bvh;
} OTracer;

static inline float mat_row_apply(const float* r, float x, float y, float z) { return (r[0] * x + r[1] * y) + r[2] * z; }
static inline void inst_ray(const OTracer* tr, uint32_t inst, vec3 origin, vec3 dir, vec3* o, vec3* d) {
  const float* m = tr->inst_inv + (size_t) inst * 12;
  const float px = origin.x - m[3], py = origin.y - m[7], pz = origin.z - m[11];
  *o = v3(mat_row_apply(m, px, py, pz), mat_row_apply(m + 4, px, py, pz), mat_row_apply(m + 8, px, py, pz));
  *d = v3(mat_row_apply(m, dir.x, dir.y, dir.z), mat_row_apply(m + 4, dir.x, dir.y, dir.z), mat_row_apply(m + 8, dir.x, dir.y, dir.z));
}

typedef struct { vec3 p0, e1, e2; } OTri;
static inline OTri mesh_tri(const OracleScene* s, uint32_t mesh, uint32_t tri) {
  OTri t;
  t.p0 = scene_vertex(s, mesh, tri, 0).pos;
  t.e1 = v_sub(scene_vertex(s, mesh, tri, 1).pos, t.p0);
  t.e2 = v_sub(scene_vertex(s, mesh, tri, 2).pos, t.p0);
  return t;
}
static inline OTri light_tri(const OracleScene* s, uint32_t light) {
  const float* p = s->light_bvh_tris + (size_t) light * 12;
  OTri t;
  t.p0 = v3(p[0], p[1], p[2]);
  t.e1 = v_sub(v3(p[4], p[5], p[6]), t.p0);
  t.e2 = v_sub(v3(p[8], p[9], p[10]), t.p0);
  return t;
}

/* ---- BVH build (oracle's own: median split over the longest centroid axis, leaves <= 4) ---- */
typedef struct { float c[3]; float lo[3], hi[3]; uint32_t id; } OBuildPrim;
static int cmp_prim(const void* a, const void* b, void* axis_) {
  const int axis = *(const int*) axis_;
  const float x = ((const OBuildPrim*) a)->c[axis], y = ((const OBuildPrim*) b)->c[axis];
  if (x < y) return -1;
  if (x > y) return 1;
  const uint32_t ia = ((const OBuildPrim*) a)->id, ib = ((const OBuildPrim*) b)->id;
  return (ia < ib) ? -1 : (ia > ib);
}
/* nodes of the subtree over `count` primitives: the split is always count / 2, so the layout (node, left subtree, right subtree) is known
   before anything is sorted and the two halves can be built by different threads */
static uint32_t bvh_subtree_nodes(uint32_t count) {
  if (count <= 4) return 1;
  const uint32_t half = count / 2;
  if (half == count - half) return 1 + 2 * bvh_subtree_nodes(half);
  return 1 + bvh_subtree_nodes(half) + bvh_subtree_nodes(count - half);
}
static void bvh_build_rec(OBvh* b, OBuildPrim* prims, uint32_t first, uint32_t count, uint32_t idx) {
  OBvhNode n;
  float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int k = 0; k < 3; k++) { n.lo[k] = FLT_MAX; n.hi[k] = -FLT_MAX; }
  for (uint32_t i = first; i < first + count; i++)
    for (int k = 0; k < 3; k++) {
      n.lo[k] = fminf(n.lo[k], prims[i].lo[k]); n.hi[k] = fmaxf(n.hi[k], prims[i].hi[k]);
      clo[k] = fminf(clo[k], prims[i].c[k]); chi[k] = fmaxf(chi[k], prims[i].c[k]);
    }
  /* conservative padding so that a triangle hit computed by the Moeller-Trumbore test is never culled by rounding */
  for (int k = 0; k < 3; k++) {
    const float pad = 1e-5f * fmaxf(fmaxf(fabsf(n.lo[k]), fabsf(n.hi[k])), 1e-20f) + 1e-30f;
    n.lo[k] -= pad; n.hi[k] += pad;
  }
  n.left = n.right = 0; n.first = first; n.count = 0;
  if (count <= 4) { n.count = count; b->nodes[idx] = n; return; }
  int axis = 0;
  if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
  if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
  qsort_r(prims + first, count, sizeof(OBuildPrim), cmp_prim, &axis);
  const uint32_t half = count / 2;
  n.left = idx + 1; n.right = idx + 1 + bvh_subtree_nodes(half);
  b->nodes[idx] = n;
#pragma omp task default(shared) if (count > 65536)
  bvh_build_rec(b, prims, first, half, n.left);
#pragma omp task default(shared) if (count > 65536)
  bvh_build_rec(b, prims, first + half, count - half, n.right);
#pragma omp taskwait
}
static void bvh_build(OBvh* b, uint32_t count, OTri (*get)(const OracleScene*, uint32_t, uint32_t), const OracleScene* s, uint32_t mesh) {
  b->nodes = NULL; b->tri_ids = NULL; b->num_nodes = 0;
  if (count == 0) return;
  OBuildPrim* prims = (OBuildPrim*) malloc(sizeof(OBuildPrim) * count);
  for (uint32_t i = 0; i < count; i++) {
    const OTri t = get(s, mesh, i);
    const vec3 a = t.p0, c1 = v_add(t.p0, t.e1), c2 = v_add(t.p0, t.e2);
    const float ax[3] = {a.x, a.y, a.z}, bx[3] = {c1.x, c1.y, c1.z}, cx[3] = {c2.x, c2.y, c2.z};
    for (int k = 0; k < 3; k++) {
      prims[i].lo[k] = fminf(ax[k], fminf(bx[k], cx[k]));
      prims[i].hi[k] = fmaxf(ax[k], fmaxf(bx[k], cx[k]));
      prims[i].c[k] = (prims[i].lo[k] + prims[i].hi[k]) * 0.5f;
    }
    prims[i].id = i;
  }
  b->nodes = (OBvhNode*) malloc(sizeof(OBvhNode) * (2 * (size_t) count));
  b->num_nodes = bvh_subtree_nodes(count);
#pragma omp parallel
#pragma omp single
  bvh_build_rec(b, prims, 0, count, 0);
  b->tri_ids = (uint32_t*) malloc(sizeof(uint32_t) * count);
  for (uint32_t i = 0; i < count; i++) b->tri_ids[i] = prims[i].id;
  free(prims);
}
static OTri light_tri_adapter(const OracleScene* s, uint32_t mesh, uint32_t i) { (void) mesh; return light_tri(s, i); }
static OTri particle_tri(const OracleScene* s, uint32_t mesh, uint32_t i) { /* triangle i of the particle vertex buffer (particle.cuh:202-207) */
  (void) mesh;
  const float* p = s->particle_vertices + (size_t) i * 12;
  OTri t;
  t.p0 = v3(p[0], p[1], p[2]);
  t.e1 = v_sub(v3(p[4], p[5], p[6]), t.p0);
  t.e2 = v_sub(v3(p[8], p[9], p[10]), t.p0);
  return t;
}

static void tracer_init(OTracer* t, const OracleScene* s, int use_bvh) {
  t->scene = s; t->use_bvh = use_bvh; t->mesh_bvh = NULL;
  t->light_bvh.nodes = NULL; t->light_bvh.tri_ids = NULL; t->light_bvh.num_nodes = 0;
  t->particle_bvh.nodes = NULL; t->particle_bvh.tri_ids = NULL; t->particle_bvh.num_nodes = 0;
  if (s->particles_active && s->particle_vertices) bvh_build(&t->particle_bvh, 2 * s->particles_count, particle_tri, s, 0); /* also for brute-force scenes */
  t->inst_inv = (float*) malloc(sizeof(float) * 12 * ((size_t) s->num_instances + 1));
  for (uint32_t i = 0; i < s->num_instances; i++) {
    const OTransform tf = scene_transform(s, i);
    const vec3 c0 = t_rel_inv(tf, v3(1.0f, 0.0f, 0.0f)), c1 = t_rel_inv(tf, v3(0.0f, 1.0f, 0.0f)), c2 = t_rel_inv(tf, v3(0.0f, 0.0f, 1.0f));
    float* m = t->inst_inv + (size_t) i * 12;
    m[0] = c0.x; m[1] = c1.x; m[2] = c2.x; m[3] = tf.translation.x;
    m[4] = c0.y; m[5] = c1.y; m[6] = c2.y; m[7] = tf.translation.y;
    m[8] = c0.z; m[9] = c1.z; m[10] = c2.z; m[11] = tf.translation.z;
  }
  if (!use_bvh) return;
  t->mesh_bvh = (OBvh*) calloc(s->num_meshes, sizeof(OBvh));
  for (uint32_t m = 0; m < s->num_meshes; m++) bvh_build(&t->mesh_bvh[m], s->mesh_tri_offset[m + 1] - s->mesh_tri_offset[m], mesh_tri, s, m);
  bvh_build(&t->light_bvh, s->num_lights, light_tri_adapter, s, 0);
}
static void tracer_free(OTracer* t) {
  free(t->inst_inv);
  if (t->mesh_bvh) {
    for (uint32_t m = 0; m < t->scene->num_meshes; m++) { free(t->mesh_bvh[m].nodes); free(t->mesh_bvh[m].tri_ids); }
    free(t->mesh_bvh);
  }
  free(t->light_bvh.nodes); free(t->light_bvh.tri_ids);
  free(t->particle_bvh.nodes); free(t->particle_bvh.tri_ids);
}

static inline bool slab_hit(const OBvhNode* n, vec3 o, vec3 inv, float tmax) {
  float t0 = 0.0f, t1 = tmax;
  const float oo[3] = {o.x, o.y, o.z}, ii[3] = {inv.x, inv.y, inv.z};
  for (int k = 0; k < 3; k++) {
    float a = (n->lo[k] - oo[k]) * ii[k], b = (n->hi[k] - oo[k]) * ii[k];
    if (a != a || b != b) continue; /* 0 * inf: origin on the slab plane of an axis-parallel ray -> no constraint */
    if (a > b) { const float tmp = a; a = b; b = tmp; }
    t0 = fmaxf(t0, a); t1 = fminf(t1, b);
  }
  return t0 <= t1 * 1.0000005f + 1e-30f;
}

/* Visits every triangle of one triangle set whose padded boxes the ray segment [0, tmax_ref] may touch. */
#define OBVH_FOREACH_TRI(bvh, count_all, o, d, tmax_expr, TRI_ID, ...)                                        \
  do {                                                                                                        \
    if ((bvh) == NULL || (bvh)->nodes == NULL) {                                                              \
      for (uint32_t TRI_ID = 0; TRI_ID < (count_all); TRI_ID++) { __VA_ARGS__ }                                      \
    }                                                                                                         \
    else {                                                                                                    \
      const vec3 inv__ = v3(1.0f / (d).x, 1.0f / (d).y, 1.0f / (d).z);                                        \
      uint32_t stack__[128]; int sp__ = 0; stack__[sp__++] = 0;                                               \
      while (sp__ > 0) {                                                                                      \
        const OBvhNode* n__ = &(bvh)->nodes[stack__[--sp__]];                                                 \
        if (!slab_hit(n__, (o), inv__, (tmax_expr))) continue;                                                \
        if (n__->count > 0) {                                                                                 \
          for (uint32_t k__ = 0; k__ < n__->count; k__++) { const uint32_t TRI_ID = (bvh)->tri_ids[n__->first + k__]; __VA_ARGS__ } \
        }                                                                                                     \
        else { stack__[sp__++] = n__->left; stack__[sp__++] = n__->right; }                                   \
      }                                                                                                       \
    }                                                                                                         \
  } while (0)

typedef struct { uint32_t instance_id, tri_id; float t; } OHit;

/* Closest hit in [0, FLT_MAX); `use_ignore` skips the triangle the path is leaving (STATE_FLAG_USE_IGNORE_HANDLE). */
static inline OHit trace_closest(const OTracer* tr, vec3 origin, vec3 dir, bool use_ignore, uint32_t ign_inst, uint32_t ign_tri) {
  const OracleScene* s = tr->scene;
  OHit best = {HIT_TYPE_SKY, 0, FLT_MAX};
  for (uint32_t inst = 0; inst < s->num_instances; inst++) {
    const uint32_t mesh = s->instance_mesh_ids[inst];
    if (mesh >= s->num_meshes) continue;
    vec3 o, d;
    inst_ray(tr, inst, origin, dir, &o, &d);
    const uint32_t ntri = s->mesh_tri_offset[mesh + 1] - s->mesh_tri_offset[mesh];
    const OBvh* bvh = tr->use_bvh ? &tr->mesh_bvh[mesh] : NULL;
    OBVH_FOREACH_TRI(bvh, ntri, o, d, best.t, tri, {
      if (!(use_ignore && inst == ign_inst && tri == ign_tri)) {
        const OTri t = mesh_tri(s, mesh, tri);
        float2_t c;
        const float th = tri_intersect(t.p0, t.e1, t.e2, o, d, &c);
        if (th < best.t || (th == best.t && th != FLT_MAX && (inst < best.instance_id || (inst == best.instance_id && tri < best.tri_id)))) {
          /* alpha cut-outs: a texel with alpha 0 does not exist for the ray (optix_common.cuh:20-46, optix_anyhit.cuh:26-30) */
          const uint32_t* tt = scene_tritex(s, mesh, tri);
          const uint16_t albedo_tex = scene_material(s, tt[3] & 0xFFFF).albedo_tex;
          const bool cut = albedo_tex != TEXTURE_NONE && albedo_tex < s->num_textures &&
                           texture_load(s, albedo_tex, triangle_uv(tt, c), true, f4(0.0f, 0.0f, 0.0f, 1.0f)).w == 0.0f;
          if (!cut) { best.instance_id = inst; best.tri_id = tri; best.t = th; }
        }
      }
    });
  }
  if (best.t == FLT_MAX) { best.instance_id = HIT_TYPE_SKY; best.tri_id = 0; }
  return best;
}

/*
 * Particles (optix_kernel_raytrace.cu:97-131, device_particle.c:23-82): the unit cell of quads is instanced on the 25 x 25 x 25 integer lattice around
 * the origin; `pos` lies in [0, 1)^3 and `dir` is the path's direction divided by the particle scale, so ray parameters are world distances. Nearest
 * hit closer than `tmax` whose barycentrics lie inside the disc of optix_common.cuh:67-74; returns the triangle index or 0xFFFFFFFF. Every lattice cell
 * whose (padded) box the segment touches is searched - no acceleration structure over the cells, unlike the product's top-level tree.
 */
#define PARTICLES_BLOCK_DIM 25
static inline uint32_t trace_particles(const OTracer* tr, vec3 pos, vec3 dir, float tmax, float* t_out) {
  const OBvh* bvh = &tr->particle_bvh;
  if (!bvh->nodes) return 0xFFFFFFFFu;
  const OracleScene* s = tr->scene;
  const OBvhNode* root = &bvh->nodes[0];
  const vec3 inv = v3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  float best_t = tmax;
  uint32_t best_tri = 0xFFFFFFFFu, best_cell = 0xFFFFFFFFu;
  const float id_x[3] = {1.0f, 0.0f, 0.0f}, id_y[3] = {0.0f, 1.0f, 0.0f}, id_z[3] = {0.0f, 0.0f, 1.0f};
  uint32_t cell = 0;
  for (int xi = 0; xi < PARTICLES_BLOCK_DIM; xi++)
    for (int yi = 0; yi < PARTICLES_BLOCK_DIM; yi++)
      for (int zi = 0; zi < PARTICLES_BLOCK_DIM; zi++, cell++) {
        const float cx = (float) (xi - (PARTICLES_BLOCK_DIM >> 1)), cy = (float) (yi - (PARTICLES_BLOCK_DIM >> 1)), cz = (float) (zi - (PARTICLES_BLOCK_DIM >> 1));
        OBvhNode box = *root;
        box.lo[0] += cx; box.hi[0] += cx; box.lo[1] += cy; box.hi[1] += cy; box.lo[2] += cz; box.hi[2] += cz;
        for (int k = 0; k < 3; k++) { box.lo[k] -= 1e-4f; box.hi[k] += 1e-4f; }
        if (!slab_hit(&box, pos, inv, best_t)) continue;
        /* the instance's world->object map, rows (1,0,0 | cx) ... as the product's top-level leaf holds them */
        const float px = pos.x - cx, py = pos.y - cy, pz = pos.z - cz;
        const vec3 o = v3(mat_row_apply(id_x, px, py, pz), mat_row_apply(id_y, px, py, pz), mat_row_apply(id_z, px, py, pz));
        const vec3 d = v3(mat_row_apply(id_x, dir.x, dir.y, dir.z), mat_row_apply(id_y, dir.x, dir.y, dir.z), mat_row_apply(id_z, dir.x, dir.y, dir.z));
        OBVH_FOREACH_TRI(bvh, 2 * s->particles_count, o, d, best_t, tri, {
          const OTri t = particle_tri(s, 0, tri);
          float2_t c;
          const float th = tri_intersect(t.p0, t.e1, t.e2, o, d, &c);
          if (th < best_t || (th == best_t && th != FLT_MAX && best_tri != 0xFFFFFFFFu && (cell < best_cell || (cell == best_cell && tri < best_tri)))) {
            const float dx = c.x - 0.5f, dy = c.y - 0.5f;
            if (!(dx * dx + dy * dy > 0.25f)) { best_t = th; best_tri = tri; best_cell = cell; }
          }
        });
      }
  *t_out = best_t;
  return best_tri;
}

/*
 * Shadow query on (tmin, tmax) = (eps, dist): product of the transparencies of every surface crossed, 0 as soon as one is
 * opaque. `target` (the sampled light) and `self` (the surface being shaded) are skipped.
 */
static inline RGBF trace_shadow(const OTracer* tr, vec3 origin, vec3 dir, float dist, uint32_t tgt_inst, uint32_t tgt_tri, uint32_t self_inst, uint32_t self_tri) {
  const OracleScene* s = tr->scene;
  /* The crossing order of a traversal is arbitrary and a float product of three or more factors depends on it: the product is
   * carried in binary64 (exact for two factors) and rounded to binary32 once; the HIP path does the same. */
  double thr[3] = {1.0, 1.0, 1.0};
  bool blocked = false;
  for (uint32_t inst = 0; inst < s->num_instances && !blocked; inst++) {
    const uint32_t mesh = s->instance_mesh_ids[inst];
    if (mesh >= s->num_meshes) continue;
    vec3 o, d;
    inst_ray(tr, inst, origin, dir, &o, &d);
    const uint32_t ntri = s->mesh_tri_offset[mesh + 1] - s->mesh_tri_offset[mesh];
    const OBvh* bvh = tr->use_bvh ? &tr->mesh_bvh[mesh] : NULL;
    OBVH_FOREACH_TRI(bvh, ntri, o, d, dist, tri, {
      if (!blocked && !(inst == tgt_inst && tri == tgt_tri) && !(inst == self_inst && tri == self_tri)) {
        const OTri t = mesh_tri(s, mesh, tri);
        float2_t c;
        const float th = tri_intersect(t.p0, t.e1, t.e2, o, d, &c);
        if (th > O_EPS && th < dist) {
          const uint32_t* tt = scene_tritex(s, mesh, tri);
          const OMaterial m = scene_material(s, tt[3] & 0xFFFF);
          const RGBAF albedo = albedo_for_shadowing(s, &m, tt, c);
          const bool colored = (m.flags & DMAT_COLORED_TRANSPARENCY) != 0;
          if (albedo.a == 1.0f) blocked = true;
          else if (!(albedo.a == 0.0f && !colored)) {
            const float tp = 1.0f - albedo.a;
            const RGBF f = colored ? c_scale(c3(albedo.r, albedo.g, albedo.b), tp) : c_splat(tp);
            thr[0] *= (double) f.r; thr[1] *= (double) f.g; thr[2] *= (double) f.b;
          }
        }
      }
    });
  }
  return blocked ? c_splat(0.0f) : c3((float) thr[0], (float) thr[1], (float) thr[2]);
}

/*
 * Light-BVH query on (eps, FLT_MAX). The reference keeps a reservoir over the any-hit invocations, whose order OptiX leaves
 * unspecified; opaque lights shorten the ray so farther lights may or may not be seen. Restated order-independently:
 *   1. t* = distance of the nearest opaque light (FLT_MAX if none);
 *   2. candidates = lights hit with eps < t <= t*, except the surface being shaded and fully transparent uncoloured ones;
 *   3. num_hits = |candidates|; the selected one minimises squares32(0x9E3779B9*light_id + random_bits) (ties: lower id),
 *      i.e. a uniform choice driven by the same random number.
 */
static inline uint32_t trace_light_bvh(const OTracer* tr, vec3 origin, vec3 dir, uint32_t self_inst, uint32_t self_tri, float random, uint32_t* num_hits) {
  const OracleScene* s = tr->scene;
  const OBvh* bvh = tr->use_bvh ? &tr->light_bvh : NULL;
  float tstar = FLT_MAX;
  for (int pass = 0; pass < 2; pass++) {
    uint32_t n = 0, best_id = LIGHT_ID_INVALID, best_key = 0xFFFFFFFFu;
    OBVH_FOREACH_TRI(bvh, s->num_lights, origin, dir, tstar, light, {
      const uint32_t inst = s->light_tri_handles[2 * light], tri = s->light_tri_handles[2 * light + 1];
      if (!(inst == self_inst && tri == self_tri)) {
        const OTri t = light_tri(s, light);
        float2_t c;
        const float th = tri_intersect(t.p0, t.e1, t.e2, origin, dir, &c);
        if (th > O_EPS && th != FLT_MAX && th <= tstar) {
          const uint32_t mesh = s->instance_mesh_ids[inst];
          const uint32_t* tt = scene_tritex(s, mesh, tri);
          const OMaterial m = scene_material(s, tt[3] & 0xFFFF);
          const float alpha = albedo_for_shadowing(s, &m, tt, c).a; /* optix_anyhit.cuh:158 */
          const bool colored = (m.flags & DMAT_COLORED_TRANSPARENCY) != 0;
          if (!(alpha == 0.0f && !colored)) {
            if (pass == 0) { if (alpha == 1.0f && th < tstar) tstar = th; }
            else {
              n++;
              const uint32_t key = squares32(0xfcbd6e15u, 0x9E3779B9u * light + f2u(random));
              if (key < best_key || (key == best_key && light < best_id)) { best_key = key; best_id = light; }
            }
          }
        }
      }
    });
    if (pass == 1) { *num_hits = n; return best_id; }
  }
  *num_hits = 0;
  return LIGHT_ID_INVALID;
}

#endif
