/*
 * ORACLE (test infrastructure, not product): independent restatement of the reference's scene encoders and light-tree build. See o_scene.h.
 * Written from the reference text; every function cites the lines it follows (paths under /root/reference/src/luminary).
 *
 * Numerics contract: IEEE single/double operations in the reference's order, no contraction (Makefile: -ffp-contract=off) except where the
 * reference itself asks for a fused multiply-add (vec128_fmadd -> _mm_fmadd_ps, host_intrinsics.h:94-100: fmaf here). The reference's own
 * build (-O3 -march=native, CMakeLists.txt:62-68) leaves contraction of its scalar expressions to the compiler, so no restatement can be
 * pinned tighter than this. Vec128 lane semantics are kept, including the w lanes the reference drags along (see rotate_quaternion below).
 */
#include "o_scene.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "o_light.h" /* the oracle's own texture fetch (texture_load) and uv_unpack: what the renderer side of the oracle uses */
/* o_light.h declares the volume variant of the importance (defined in o_volume.h for the renderer); nothing here calls it */
static inline float light_tree_importance_volume(const struct VolCtx* c, float power, vec3 mean, float std_dev) { (void) c; (void) power; (void) mean; (void) std_dev; return 0.0f; }

/* ------------------------------------------------------------------------------------------------------------------------------------
 * device_packing.c
 * ---------------------------------------------------------------------------------------------------------------------------------- */

/* device_packing.c:6-35: octahedral normal, computed in double, two rounded u16 */
static uint32_t sc_pack_normal(const float n[3]) {
  double x = n[0], y = n[1], z = n[2];
  const double recip_norm = 1.0 / (fabs(x) + fabs(y) + fabs(z));
  x *= recip_norm;
  y *= recip_norm;
  z *= recip_norm;
  const double t = fmax(fmin(-z, 1.0), 0.0);
  x += (x >= 0.0) ? t : -t;
  y += (y >= 0.0) ? t : -t;
  x = fmax(fmin(x, 1.0), -1.0);
  y = fmax(fmin(y, 1.0), -1.0);
  x = (x + 1.0) * 0.5;
  y = (y + 1.0) * 0.5;
  const uint32_t xu = (uint32_t) (x * 0xFFFF + 0.5);
  const uint32_t yu = (uint32_t) (y * 0xFFFF + 0.5);
  return (yu << 16) | xu;
}

/* device_packing.c:37-47: two truncated bfloat16 */
static uint32_t sc_pack_uv(float u, float v) {
  uint32_t ub, vb;
  memcpy(&ub, &u, 4);
  memcpy(&vb, &v, 4);
  return (ub & 0xFFFF0000u) | (vb >> 16);
}

enum { SC_ROUND = 0, SC_CEIL = 1, SC_FLOOR = 2 };
/* device_packing.c:49-72 */
static uint16_t sc_pack_float(float val, int mode) {
  uint32_t b;
  memcpy(&b, &val, 4);
  switch (mode) {
    case SC_ROUND: b += (1u << 15); break;
    case SC_CEIL: if (val >= 0.0f) b += (1u << 16) - 1u; break;
    case SC_FLOOR: if (val < 0.0f) b += (1u << 16) - 1u; break;
    default: break;
  }
  return (uint16_t) (b >> 16);
}
/* device_packing.c:74-85 */
static float sc_unpack_float(uint16_t v) {
  const uint32_t b = ((uint32_t) v) << 16;
  float f;
  memcpy(&f, &b, 4);
  return f;
}

/* ------------------------------------------------------------------------------------------------------------------------------------
 * device_structs.c
 * ---------------------------------------------------------------------------------------------------------------------------------- */

/* device_structs.c:251-253 */
static uint16_t sc_float01_to_u16(float f) { return (uint16_t) (f * 0xFFFFu + 0.5f); }
/* device_structs.c:255-260: 8 exponent bits and 8 mantissa bits of a non-negative float */
static uint16_t sc_float_to_u16(float f) {
  uint32_t b;
  memcpy(&b, &f, 4);
  return (uint16_t) ((b >> 15) & 0xFFFFu);
}

/* device_structs.c:262-311 -> DeviceMaterialCompressed (device_structs.h:216-238), 16 halfwords */
static void sc_encode_material(const OSceneMaterial* m, uint16_t out[16]) {
  uint8_t flags = 0;
  flags |= m->emission_active ? 0x02 : 0;
  flags |= m->thin_walled ? 0x04 : 0;
  flags |= m->metallic ? 0x08 : 0;
  flags |= m->colored_transparency ? 0x10 : 0;
  flags |= m->roughness_as_smoothness ? 0x20 : 0;
  flags |= m->normal_map_is_compressed ? 0x40 : 0;
  flags |= m->bidirectional_emission ? 0x80 : 0;
  flags |= (m->base_substrate == 1) ? 0x01 : 0x00;
  const uint8_t roughness_clamp = (uint8_t) (sc_float01_to_u16(m->roughness_clamp) >> 8);
  float er = m->emission[0], eg = m->emission[1], eb = m->emission[2];
  const float emission_normalization = 1.0f / fminf(fmaxf(fmaxf(er, eg), eb) + 1.0f, (float) 0xFFFFu);
  er *= emission_normalization;
  eg *= emission_normalization;
  eb *= emission_normalization;
  out[0] = (uint16_t) (flags | ((uint16_t) roughness_clamp << 8)); /* u8 flags, u8 roughness_clamp (little endian) */
  out[1] = m->metallic_tex;
  out[2] = sc_float01_to_u16(m->roughness);
  out[3] = sc_float01_to_u16(0.5f * (m->refraction_index - 1.0f));
  out[4] = sc_float01_to_u16(m->albedo[0]);
  out[5] = sc_float01_to_u16(m->albedo[1]);
  out[6] = sc_float01_to_u16(m->albedo[2]);
  out[7] = sc_float01_to_u16(m->albedo[3]);
  out[8] = sc_float01_to_u16(er);
  out[9] = sc_float01_to_u16(eg);
  out[10] = sc_float01_to_u16(eb);
  out[11] = sc_float_to_u16(m->emission_scale / emission_normalization);
  out[12] = m->albedo_tex;
  out[13] = m->luminance_tex;
  out[14] = m->roughness_tex;
  out[15] = m->normal_tex;
}

typedef struct { float x, y, z, w; } SQuat;
/* host_math.c:6-21 */
static SQuat sc_euler_to_quaternion(const float r[3]) {
  const float cr = cosf(r[0] * 0.5f), sr = sinf(r[0] * 0.5f);
  const float cp = cosf(r[1] * 0.5f), sp = sinf(r[1] * 0.5f);
  const float cy = cosf(r[2] * 0.5f), sy = sinf(r[2] * 0.5f);
  SQuat q;
  q.w = cr * cp * cy + sr * sp * sy;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  return q;
}

/* device_structs.c:386-412 -> DeviceTransform (device_structs.h:295-300): translation, scale, Quaternion16 of the INVERSE rotation */
static void sc_encode_transform(const OSceneInstance* inst, float out[8]) {
  const SQuat q = sc_euler_to_quaternion(inst->rotation);
  memcpy(out, inst->translation, 12);
  memcpy(out + 3, inst->scale, 12);
  const uint16_t q16[4] = {(uint16_t) (((1.0f - q.x) * 0x7FFF) + 0.5f), (uint16_t) (((1.0f - q.y) * 0x7FFF) + 0.5f), (uint16_t) (((1.0f - q.z) * 0x7FFF) + 0.5f),
                           (uint16_t) (((1.0f + q.w) * 0x7FFF) + 0.5f)};
  memcpy(out + 6, q16, 8);
}

/* ------------------------------------------------------------------------------------------------------------------------------------
 * host_intrinsics.h: the four-lane arithmetic the light tree is written in. Lane order x, y, z, w.
 * ---------------------------------------------------------------------------------------------------------------------------------- */
typedef struct { float d[4]; } V4;
static V4 v4_set(float x, float y, float z, float w) { V4 r = {{x, y, z, w}}; return r; }
static V4 v4_set1(float a) { return v4_set(a, a, a, a); }
static V4 v4_add(V4 a, V4 b) { V4 r; for (int k = 0; k < 4; k++) r.d[k] = a.d[k] + b.d[k]; return r; }
static V4 v4_sub(V4 a, V4 b) { V4 r; for (int k = 0; k < 4; k++) r.d[k] = a.d[k] - b.d[k]; return r; }
static V4 v4_mul(V4 a, V4 b) { V4 r; for (int k = 0; k < 4; k++) r.d[k] = a.d[k] * b.d[k]; return r; }
static V4 v4_scale(V4 a, float b) { return v4_mul(a, v4_set1(b)); }
static V4 v4_fmadd(V4 a, V4 b, V4 c) { V4 r; for (int k = 0; k < 4; k++) r.d[k] = fmaf(a.d[k], b.d[k], c.d[k]); return r; } /* _mm_fmadd_ps */
static V4 v4_min(V4 a, V4 b) { V4 r; for (int k = 0; k < 4; k++) r.d[k] = (a.d[k] < b.d[k]) ? a.d[k] : b.d[k]; return r; }    /* _mm_min_ps */
static V4 v4_max(V4 a, V4 b) { V4 r; for (int k = 0; k < 4; k++) r.d[k] = (a.d[k] > b.d[k]) ? a.d[k] : b.d[k]; return r; }    /* _mm_max_ps */
static V4 v4_w0(V4 a) { a.d[3] = 0.0f; return a; }
/* host_intrinsics.h:102-105 */
static V4 v4_cross(V4 a, V4 b) { return v4_set(a.d[1] * b.d[2] - a.d[2] * b.d[1], a.d[2] * b.d[0] - a.d[0] * b.d[2], a.d[0] * b.d[1] - a.d[1] * b.d[0], 0.0f); }
/* host_intrinsics.h:107-117: (x + z) + (y + w) */
static float v4_hsum(V4 a) { return (a.d[0] + a.d[2]) + (a.d[1] + a.d[3]); }
/* host_intrinsics.h:185-195: max(max(x, z), max(y, w)) with _mm_max_ps's operand order */
static float v4_hmax(V4 a) {
  const float m0 = (a.d[0] > a.d[2]) ? a.d[0] : a.d[2], m1 = (a.d[1] > a.d[3]) ? a.d[1] : a.d[3];
  return (m0 > m1) ? m0 : m1;
}
static float v4_norm2(V4 a) { return sqrtf(v4_hsum(v4_mul(a, a))); }
static float v4_dot(V4 a, V4 b) { return v4_hsum(v4_mul(a, b)); }
/* host_intrinsics.h:208-216: lanes (x y, x z, y z, w w) summed as (xy + yz) + (xz + ww); w is 0 at every call */
static float v4_box_area(V4 a) { return v4_hsum(v4_set(a.d[0] * a.d[1], a.d[0] * a.d[2], a.d[1] * a.d[2], a.d[3] * a.d[3])); }
/* host_intrinsics.h:221-233. NOTE the w lane: scale(q, 2 dot_qa) multiplies q.w too, so a rotated vertex leaves here with w = 2 q.w dot(q.xyz, a.xyz) -
 * not 0. The reference never clears it: it travels through the fragment's v0/v1/v2/middle, enters the node variance through the four-lane dot product
 * (device_light.c:538-547) and ends in the light BVH's vertex buffer. Kept. */
static V4 v4_rotate_quaternion(V4 a, V4 q) {
  const float dot_qa = a.d[0] * q.d[0] + a.d[1] * q.d[1] + a.d[2] * q.d[2];
  const float dot_qq = q.d[0] * q.d[0] + q.d[1] * q.d[1] + q.d[2] * q.d[2];
  const V4 cross = v4_cross(q, a);
  V4 result = v4_scale(q, 2.0f * dot_qa);
  result = v4_add(result, v4_scale(a, q.d[3] * q.d[3] - dot_qq));
  result = v4_add(result, v4_scale(cross, 2.0f * q.d[3]));
  if (getenv("O_SCENE_CLEAR_W")) result.d[3] = 0.0f; /* diagnosis only: what an implementation without the w lane computes */
  return result;
}

/* ------------------------------------------------------------------------------------------------------------------------------------
 * device_light.c
 * ---------------------------------------------------------------------------------------------------------------------------------- */
#define SC_MAX_VALUE 1e10f                       /* device_light.c:106 */
#define SC_FRAGMENT_ERROR_COMP (FLT_EPSILON * 16.0f) /* :108 */
#define SC_BIN_COUNT 32                          /* :102-103 */
#define SC_NULL 0xFFFFFFFFu
#define SC_ROOT_MAX_CHILDREN 128 /* device_utils.h:45 */
#define SC_NODE_CHILDREN 8       /* device_utils.h:46, :48 */

/* device_light.h LightTreeFragment: what the build reads of it */
typedef struct {
  V4 low, high, middle, v0, v1, v2;
  float power, intensity;
  uint32_t instance_id, tri_id /* mesh triangle id: what the handle map wants */;
} SFragment;

typedef struct { /* device_light.c:40-58, the fields the later stages read */
  uint32_t triangle_count, triangles_address, child_address;
  int internal;
  float left_power, right_power;
} SBinaryNode;

typedef struct { /* :60-76 */
  V4 left_mean, right_mean; /* xyz used */
  float left_variance, right_variance, left_power, right_power;
  uint32_t child_ptr, light_ptr, light_count;
} SNode;

typedef struct { float mean[3]; float variance, power; int is_leaf; } SChild; /* :78-83 */

typedef struct { V4 high, low; int32_t entry, exit; float power; } SBin; /* :95-102 */

/* :118-138 */
static void sc_fit_bounds(const SFragment* f, uint32_t n, V4* high_out, V4* low_out) {
  V4 high = v4_set1(-SC_MAX_VALUE), low = v4_set1(SC_MAX_VALUE);
  for (uint32_t i = 0; i < n; i++) { high = v4_max(high, f[i].high); low = v4_min(low, f[i].low); }
  *high_out = high;
  *low_out = low;
}
/* :140-160 */
static void sc_fit_bounds_of_bins(const SBin* b, uint32_t n, V4* high_out, V4* low_out) {
  V4 high = v4_set1(-SC_MAX_VALUE), low = v4_set1(SC_MAX_VALUE);
  for (uint32_t i = 0; i < n; i++) { high = v4_max(high, b[i].high); low = v4_min(low, b[i].low); }
  *high_out = high;
  *low_out = low;
}

/* :172-243: 32 bins over the fragments' BOUNDS along `axis`, a fragment goes where its centroid falls; returns the bin width (0 = no split on this axis) */
static double sc_construct_bins(SBin* bins, const SFragment* f, uint32_t n, int axis, double* offset) {
  V4 high, low;
  sc_fit_bounds(f, n, &high, &low);
  const double high_axis = high.d[axis], low_axis = low.d[axis];
  const double span = high_axis - low_axis;
  const double interval = span / SC_BIN_COUNT;
  if (interval <= SC_FRAGMENT_ERROR_COMP * fabs(low_axis)) return 0.0;
  *offset = low_axis;
  for (int k = 0; k < SC_BIN_COUNT; k++) {
    bins[k].high = v4_set(-SC_MAX_VALUE, -SC_MAX_VALUE, -SC_MAX_VALUE, 0.0f); /* the designated initialiser leaves w at 0 */
    bins[k].low = v4_set(SC_MAX_VALUE, SC_MAX_VALUE, SC_MAX_VALUE, 0.0f);
    bins[k].entry = 0; bins[k].exit = 0; bins[k].power = 0.0f;
  }
  const double inv_interval = 1.0 / interval;
  for (uint32_t i = 0; i < n; i++) {
    const double value = f[i].middle.d[axis];
    int32_t pos = ((int32_t) ceil((value - low_axis) * inv_interval)) - 1;
    if (pos < 0) pos = 0;
    if (pos >= SC_BIN_COUNT) pos = SC_BIN_COUNT - 1;
    bins[pos].entry++;
    bins[pos].exit++;
    bins[pos].power += f[i].power;
    bins[pos].high = v4_max(bins[pos].high, f[i].high);
    bins[pos].low = v4_min(bins[pos].low, f[i].low);
  }
  return interval;
}

/* :245-268: centroids above the plane go to the back, in the reference's swap order */
static void sc_divide_along_axis(double split, int axis, SFragment* f, uint32_t n) {
  uint32_t left = 0, right = 0;
  while (left + right < n) {
    const SFragment frag = f[left];
    const double middle = frag.middle.d[axis];
    if (middle > split) {
      const uint32_t swap_index = n - 1 - right;
      const SFragment temp = f[swap_index];
      f[swap_index] = frag;
      f[left] = temp;
      right++;
    }
    else left++;
  }
}

/* :270-486: binary tree by binned SAH weighted by power, breadth first, leaves hold one light */
static SBinaryNode* sc_build_binary(SFragment* fragments, uint32_t fragments_count, uint32_t* nodes_count_out) {
  SBinaryNode* nodes = (SBinaryNode*) calloc((size_t) 2 * fragments_count + 2, sizeof(SBinaryNode));
  uint32_t num_nodes = 0;
  *nodes_count_out = 0;
  if (fragments_count == 0) return nodes;
  nodes[num_nodes].triangles_address = 0;
  nodes[num_nodes].triangle_count = fragments_count;
  nodes[num_nodes].internal = 0;
  num_nodes++;
  SBin bins[SC_BIN_COUNT];
  uint32_t begin = 0, end = 1;
  while (begin != end) {
    for (uint32_t node_ptr = begin; node_ptr < end; node_ptr++) {
      SBinaryNode node = nodes[node_ptr];
      const uint32_t fptr = node.triangles_address, fcount = node.triangle_count;
      if (fcount == 1) continue;
      V4 high_parent, low_parent;
      sc_fit_bounds(fragments + fptr, fcount, &high_parent, &low_parent);
      const V4 diff = v4_w0(v4_sub(high_parent, low_parent));
      const float max_axis_interval = v4_hmax(diff);
      double optimal_cost = DBL_MAX;
      int axis = 0;
      double optimal_splitting_plane = 0.0;
      int found_split = 0;
      uint32_t optimal_split = 0;
      float optimal_left_power = 0.0f, optimal_right_power = 0.0f;
      for (int a = 0; a < 3; a++) {
        double low_split = 0.0;
        const double interval = sc_construct_bins(bins, fragments + fptr, fcount, a, &low_split);
        if (interval == 0.0) continue;
        const double interval_cost = max_axis_interval / interval;
        uint32_t left = 0;
        float left_power = 0.0f, right_power = 0.0f;
        for (int k = 0; k < SC_BIN_COUNT; k++) right_power += bins[k].power;
        V4 high_left = v4_set1(-SC_MAX_VALUE), high_right = v4_set1(-SC_MAX_VALUE), low_left = v4_set1(SC_MAX_VALUE), low_right = v4_set1(SC_MAX_VALUE);
        for (int k = 1; k < SC_BIN_COUNT; k++) {
          high_left = v4_max(high_left, bins[k - 1].high);
          low_left = v4_min(low_left, bins[k - 1].low);
          sc_fit_bounds_of_bins(bins + k, SC_BIN_COUNT - k, &high_right, &low_right);
          left_power += bins[k - 1].power;
          right_power -= bins[k - 1].power;
          const V4 diff_left = v4_sub(high_left, low_left), diff_right = v4_sub(high_right, low_right);
          const float left_area = v4_box_area(v4_w0(diff_left)), right_area = v4_box_area(v4_w0(diff_right));
          const double total_cost = interval_cost * (left_power * left_area + right_power * right_area);
          left += (uint32_t) bins[k - 1].entry;
          if (left == 0 || left == fcount) continue;
          if (total_cost < optimal_cost) {
            optimal_cost = total_cost;
            optimal_split = left;
            optimal_splitting_plane = low_split + k * interval;
            found_split = 1;
            axis = a;
            optimal_left_power = left_power;
            optimal_right_power = right_power;
          }
        }
      }
      if (found_split) sc_divide_along_axis(optimal_splitting_plane, axis, fragments + fptr, fcount);
      else { /* no plane separates the centroids: halve the list */
        optimal_split = fcount / 2;
        optimal_left_power = 0.0f;
        optimal_right_power = 0.0f;
        uint32_t id = 0;
        for (; id < optimal_split; id++) optimal_left_power += fragments[fptr + id].power;
        for (; id < fcount; id++) optimal_right_power += fragments[fptr + id].power;
      }
      node.left_power = optimal_left_power;
      node.right_power = optimal_right_power;
      node.child_address = num_nodes;
      SBinaryNode left_node; memset(&left_node, 0, sizeof(left_node));
      left_node.triangle_count = optimal_split;
      left_node.triangles_address = fptr;
      nodes[num_nodes++] = left_node;
      SBinaryNode right_node; memset(&right_node, 0, sizeof(right_node));
      right_node.triangle_count = node.triangle_count - optimal_split;
      right_node.triangles_address = fptr + optimal_split;
      nodes[num_nodes++] = right_node;
      node.internal = 1;
      nodes[node_ptr] = node;
    }
    begin = end;
    end = num_nodes;
  }
  *nodes_count_out = num_nodes;
  return nodes;
}

/* :488-584 (the build's default: power weighting, variance as the weighted mean square distance of the three vertices) */
static void sc_mean_and_variance(const SFragment* fragments, SBinaryNode node, float parent_power, float* power, V4* mean, float* variance) {
  if (*power < parent_power * 1e-5f) { /* numerically unstable: add it up again */
    float new_power = 0.0f;
    for (uint32_t i = 0; i < node.triangle_count; i++) new_power += fragments[node.triangles_address + i].power;
    *power = new_power;
  }
  const float inverse_total_power = 1.0f / *power;
  V4 p = v4_set1(0.0f);
  for (uint32_t i = 0; i < node.triangle_count; i++) {
    const SFragment* f = fragments + node.triangles_address + i;
    const float weight = f->power * inverse_total_power;
    p = v4_fmadd(f->middle, v4_set1(weight), p);
  }
  float spatial_variance = 0.0f;
  for (uint32_t i = 0; i < node.triangle_count; i++) {
    const SFragment* f = fragments + node.triangles_address + i;
    const float weight = (1.0f / 3.0f) * f->power * inverse_total_power;
    const V4 diff0 = v4_sub(f->v0, p);
    spatial_variance += weight * v4_dot(diff0, diff0);
    const V4 diff1 = v4_sub(f->v1, p);
    spatial_variance += weight * v4_dot(diff1, diff1);
    const V4 diff2 = v4_sub(f->v2, p);
    spatial_variance += weight * v4_dot(diff2, diff2);
  }
  *mean = p;
  *variance = spatial_variance;
}

/* :589-649 */
static SNode* sc_build_traversal_structure(const SFragment* fragments, const SBinaryNode* bnodes, uint32_t count) {
  SNode* nodes = (SNode*) calloc((size_t) count + 1, sizeof(SNode));
  for (uint32_t i = 0; i < count; i++) {
    const SBinaryNode b = bnodes[i];
    SNode n; memset(&n, 0, sizeof(n));
    n.left_power = b.left_power;
    n.right_power = b.right_power;
    n.light_count = b.triangle_count;
    n.light_ptr = b.triangles_address;
    if (b.internal) {
      const float parent_power = b.left_power + b.right_power;
      n.child_ptr = b.child_address;
      sc_mean_and_variance(fragments, bnodes[b.child_address], parent_power, &n.left_power, &n.left_mean, &n.left_variance);
      sc_mean_and_variance(fragments, bnodes[b.child_address + 1], parent_power, &n.right_power, &n.right_mean, &n.right_variance);
    }
    else n.child_ptr = SC_NULL;
    nodes[i] = n;
  }
  return nodes;
}

typedef struct {
  uint32_t* binary_node_indices; /* the job queue */
  uint32_t num_node_jobs;
  uint32_t* new_fragments;
  uint32_t triangles_ptr;
  uint8_t* root; uint32_t root_bytes;
  uint8_t* nodes; uint32_t num_nodes;
} SCollapse;

/* :663-832: one wide node out of a binary subtree: split the child with the largest power x variance until the node is full */
static void sc_collapse_binary_node(SCollapse* cw, SNode base, const SNode* bnodes, SChild* children, uint32_t* child_binary_index, uint32_t max_child_count,
                                    uint32_t* light_ptr, uint32_t* child_count_out, uint32_t* leaf_count_out) {
  uint32_t child_count = 0;
  int children_require_work = 0;
  if (base.light_count > 1) {
    SChild l; memset(&l, 0, sizeof(l));
    l.mean[0] = base.left_mean.d[0]; l.mean[1] = base.left_mean.d[1]; l.mean[2] = base.left_mean.d[2];
    l.variance = base.left_variance; l.power = base.left_power;
    child_binary_index[child_count] = base.child_ptr;
    children[child_count++] = l;
    SChild r; memset(&r, 0, sizeof(r));
    r.mean[0] = base.right_mean.d[0]; r.mean[1] = base.right_mean.d[1]; r.mean[2] = base.right_mean.d[2];
    r.variance = base.right_variance; r.power = base.right_power;
    child_binary_index[child_count] = base.child_ptr + 1;
    children[child_count++] = r;
    children_require_work = (child_count < max_child_count);
  }
  else { /* one light in the whole scene */
    SChild c; memset(&c, 0, sizeof(c));
    c.is_leaf = 1;
    c.power = 1.0f;
    child_binary_index[child_count] = 0;
    children[child_count++] = c;
  }
  while (children_require_work) {
    children_require_work = 0;
    float optimal_cost = 0.0f;
    uint32_t selected = 0;
    for (uint32_t c = 0; c < max_child_count; c++) {
      const uint32_t bi = child_binary_index[c];
      if (bi == SC_NULL) continue;
      const SNode bn = bnodes[bi];
      if (bn.light_count == 1) continue;
      const float cost = (bn.left_power + bn.right_power) * (bn.left_variance + bn.right_variance);
      if (cost > optimal_cost) { optimal_cost = cost; selected = c; children_require_work = 1; }
    }
    if (!children_require_work) break;
    const SNode bn = bnodes[child_binary_index[selected]];
    SChild l; memset(&l, 0, sizeof(l));
    l.mean[0] = bn.left_mean.d[0]; l.mean[1] = bn.left_mean.d[1]; l.mean[2] = bn.left_mean.d[2];
    l.variance = bn.left_variance; l.power = bn.left_power;
    child_binary_index[selected] = bn.child_ptr;
    children[selected] = l;
    SChild r; memset(&r, 0, sizeof(r));
    r.mean[0] = bn.right_mean.d[0]; r.mean[1] = bn.right_mean.d[1]; r.mean[2] = bn.right_mean.d[2];
    r.variance = bn.right_variance; r.power = bn.right_power;
    uint32_t slot = 0;
    for (; slot < max_child_count; slot++) if (child_binary_index[slot] == SC_NULL) break;
    child_binary_index[slot] = bn.child_ptr + 1;
    children[slot] = r;
    child_count++;
    if (child_count == max_child_count) break;
  }
  if (child_count < max_child_count) { /* non-null children first */
    for (uint32_t c = 0; c < child_count; c++) {
      if (child_binary_index[c] == SC_NULL) {
        uint32_t s = child_count;
        for (; s < max_child_count; s++) if (child_binary_index[s] != SC_NULL) break;
        const SChild tc = children[c]; children[c] = children[s]; children[s] = tc;
        const uint32_t ti = child_binary_index[c]; child_binary_index[c] = child_binary_index[s]; child_binary_index[s] = ti;
      }
    }
  }
  uint32_t num_leaf_nodes = 0;
  for (uint32_t c = 0; c < child_count; c++) { /* a child that holds one light is a leaf: the light takes the next place in the new order */
    const SNode bn = bnodes[child_binary_index[c]];
    if (bn.light_count == 1) {
      if (*light_ptr == SC_NULL) *light_ptr = cw->triangles_ptr; /* :655-658 */
      cw->new_fragments[cw->triangles_ptr++] = bn.light_ptr;
      children[c].is_leaf = 1;
      child_binary_index[c] = SC_NULL;
      num_leaf_nodes++;
    }
  }
  for (uint32_t c = 0; c < num_leaf_nodes; c++) { /* leaves first, their order untouched */
    if (!children[c].is_leaf) {
      uint32_t s = c + 1;
      for (; s < child_count; s++) if (children[s].is_leaf) break;
      const SChild tc = children[c]; children[c] = children[s]; children[s] = tc;
      const uint32_t ti = child_binary_index[c]; child_binary_index[c] = child_binary_index[s]; child_binary_index[s] = ti;
    }
  }
  *child_count_out = child_count;
  *leaf_count_out = num_leaf_nodes;
}

/* the conversion `int8_t field = <float expression>` does on x86-64 (cvttss2si to 32 bits, then the low byte): -inf -> 0x80000000 -> 0 */
static int8_t sc_float_to_i8(float f) {
  int32_t i;
  if (!(f >= -2147483648.0f && f < 2147483648.0f)) i = (int32_t) 0x80000000u; /* the "integer indefinite" value */
  else i = (int32_t) f;
  return (int8_t) (uint8_t) (i & 0xFF);
}

typedef struct { float min_mean[3]; float cx, cy, cz, cv; float max_power; uint16_t bx, by, bz; int8_t ex, ey, ez, es; } SQuantiser;
/* :852-896 and :1068-1105 (identical in both) */
static SQuantiser sc_quantiser(const SChild* children, uint32_t child_count) {
  float mn[3] = {SC_MAX_VALUE, SC_MAX_VALUE, SC_MAX_VALUE}, mx[3] = {-SC_MAX_VALUE, -SC_MAX_VALUE, -SC_MAX_VALUE};
  float max_variance = 0.0f, max_power = 0.0f;
  for (uint32_t c = 0; c < child_count; c++) {
    for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], children[c].mean[k]); mx[k] = fmaxf(mx[k], children[c].mean[k]); }
    max_variance = fmaxf(max_variance, children[c].variance);
    max_power = fmaxf(max_power, children[c].power);
  }
  const float max_std_dev = sqrtf(max_variance);
  SQuantiser q;
  q.bx = sc_pack_float(mn[0], SC_FLOOR); q.by = sc_pack_float(mn[1], SC_FLOOR); q.bz = sc_pack_float(mn[2], SC_FLOOR);
  q.min_mean[0] = sc_unpack_float(q.bx); q.min_mean[1] = sc_unpack_float(q.by); q.min_mean[2] = sc_unpack_float(q.bz);
  q.ex = sc_float_to_i8((mx[0] != q.min_mean[0]) ? ceilf(log2f((mx[0] - q.min_mean[0]) * 1.0f / 255.0f)) : 0);
  q.ey = sc_float_to_i8((mx[1] != q.min_mean[1]) ? ceilf(log2f((mx[1] - q.min_mean[1]) * 1.0f / 255.0f)) : 0);
  q.ez = sc_float_to_i8((mx[2] != q.min_mean[2]) ? ceilf(log2f((mx[2] - q.min_mean[2]) * 1.0f / 255.0f)) : 0);
  q.es = sc_float_to_i8(ceilf(log2f(max_std_dev * 1.0f / 255.0f)));
  q.cx = 1.0f / exp2f(q.ex); q.cy = 1.0f / exp2f(q.ey); q.cz = 1.0f / exp2f(q.ez); q.cv = 1.0f / exp2f(q.es);
  q.max_power = max_power;
  return q;
}
static uint64_t sc_max_u64(uint64_t a, uint64_t b) { return a > b ? a : b; }

/* :834-1022 -> DeviceLightTreeRootHeader (16 B) + DeviceLightTreeRootSection (48 B each), device_utils.h:304-327 */
static void sc_collapse_root(SCollapse* cw, const SNode* bnodes) {
  SChild children[SC_ROOT_MAX_CHILDREN];
  uint32_t child_binary_index[SC_ROOT_MAX_CHILDREN];
  for (int i = 0; i < SC_ROOT_MAX_CHILDREN; i++) child_binary_index[i] = SC_NULL;
  uint32_t light_ptr = SC_NULL, child_count = 0, num_lights = 0;
  sc_collapse_binary_node(cw, bnodes[0], bnodes, children, child_binary_index, SC_ROOT_MAX_CHILDREN, &light_ptr, &child_count, &num_lights);
  const SQuantiser q = sc_quantiser(children, child_count);
  const uint32_t num_sections = (child_count + SC_NODE_CHILDREN - 1) / SC_NODE_CHILDREN;
  cw->root_bytes = 16 + 48 * num_sections;
  cw->root = (uint8_t*) calloc(cw->root_bytes, 1);
  uint8_t* h = cw->root;
  memcpy(h + 0, &q.bx, 2); memcpy(h + 2, &q.by, 2); memcpy(h + 4, &q.bz, 2);
  const uint16_t nrl = (uint16_t) num_lights;
  memcpy(h + 6, &nrl, 2);
  const uint16_t pn = sc_pack_float(q.max_power, SC_CEIL);
  memcpy(h + 8, &pn, 2);
  h[10] = (uint8_t) num_sections;
  h[11] = 0;
  h[12] = (uint8_t) q.ex; h[13] = (uint8_t) q.ey; h[14] = (uint8_t) q.ez; h[15] = (uint8_t) q.es;
  for (uint32_t s = 0; s < num_sections; s++) {
    uint8_t* sec = cw->root + 16 + 48 * s; /* rel_mean_x[8] rel_mean_y[8] rel_mean_z[8] rel_std_dev[8] u16 rel_power[8] */
    const uint32_t c0 = s * SC_NODE_CHILDREN, c1 = (child_count < (s + 1) * SC_NODE_CHILDREN) ? child_count : (s + 1) * SC_NODE_CHILDREN;
    for (uint32_t c = c0; c < c1; c++) {
      const SChild ch = children[c];
      uint64_t rx = (uint64_t) floorf((ch.mean[0] - q.min_mean[0]) * q.cx + 0.5f);
      uint64_t ry = (uint64_t) floorf((ch.mean[1] - q.min_mean[1]) * q.cy + 0.5f);
      uint64_t rz = (uint64_t) floorf((ch.mean[2] - q.min_mean[2]) * q.cz + 0.5f);
      uint64_t rs = (uint64_t) (sqrtf(ch.variance) * q.cv + 0.5f);
      uint64_t rp = (uint64_t) floorf(0xFFFF * ch.power / q.max_power + 0.5f);
      rs = sc_max_u64(rs, 1);
      rp = sc_max_u64(rp, 1);
      const uint32_t k = c - c0;
      sec[k] = (uint8_t) rx; sec[8 + k] = (uint8_t) ry; sec[16 + k] = (uint8_t) rz; sec[24 + k] = (uint8_t) rs;
      const uint16_t rp16 = (uint16_t) rp;
      memcpy(sec + 32 + 2 * k, &rp16, 2);
    }
  }
  for (uint32_t i = 0; i < child_count; i++) {
    if (child_binary_index[i] == SC_NULL) continue;
    cw->binary_node_indices[cw->num_node_jobs++] = child_binary_index[i];
  }
}

/* :1024-1136 -> DeviceLightTreeNode (64 B), device_utils.h:283-302 */
static void sc_collapse_nodes(SCollapse* cw, const SNode* bnodes) {
  for (uint32_t job = 0; job < cw->num_node_jobs; job++) {
    const uint32_t binary_index = cw->binary_node_indices[job];
    SChild children[SC_NODE_CHILDREN];
    uint32_t child_binary_index[SC_NODE_CHILDREN];
    for (int i = 0; i < SC_NODE_CHILDREN; i++) child_binary_index[i] = SC_NULL;
    uint32_t light_ptr = SC_NULL, child_count = 0, num_lights = 0;
    sc_collapse_binary_node(cw, bnodes[binary_index], bnodes, children, child_binary_index, SC_NODE_CHILDREN, &light_ptr, &child_count, &num_lights);
    const SQuantiser q = sc_quantiser(children, child_count);
    uint8_t* n = cw->nodes + (size_t) 64 * cw->num_nodes++;
    memset(n, 0, 64);
    memcpy(n + 0, &q.bx, 2); memcpy(n + 2, &q.by, 2); memcpy(n + 4, &q.bz, 2);
    n[8] = (uint8_t) q.ex; n[9] = (uint8_t) q.ey; n[10] = (uint8_t) q.ez; n[11] = (uint8_t) q.es;
    n[12] = (uint8_t) num_lights;
    const uint32_t child_ptr = cw->num_node_jobs; /* where this node's inner children will be queued */
    memcpy(n + 16, &child_ptr, 4);
    memcpy(n + 20, &light_ptr, 4);
    for (uint32_t c = 0; c < child_count; c++) {
      const SChild ch = children[c];
      uint64_t rx = (uint64_t) floorf((ch.mean[0] - q.min_mean[0]) * q.cx + 0.5f);
      uint64_t ry = (uint64_t) floorf((ch.mean[1] - q.min_mean[1]) * q.cy + 0.5f);
      uint64_t rz = (uint64_t) floorf((ch.mean[2] - q.min_mean[2]) * q.cz + 0.5f);
      uint64_t rs = (uint64_t) (sqrtf(ch.variance) * q.cv + 0.5f);
      uint64_t rp = (uint64_t) floorf(0xFF * ch.power / q.max_power + 0.5f);
      rs = sc_max_u64(rs, 1);
      rp = sc_max_u64(rp, 1);
      n[24 + c] = (uint8_t) rx; n[32 + c] = (uint8_t) ry; n[40 + c] = (uint8_t) rz; n[48 + c] = (uint8_t) rs; n[56 + c] = (uint8_t) rp;
    }
    for (uint32_t i = 0; i < child_count; i++) {
      if (child_binary_index[i] == SC_NULL) continue;
      cw->binary_node_indices[cw->num_node_jobs++] = child_binary_index[i];
    }
  }
}

/* ---- textured emitters: light_compute_intensity, cuda/light.cuh:191-270 with light_microtriangle.cuh:8-61 ---- */
static void sc_microtriangle_bary(uint32_t id, float b0[2], float b1[2], float b2[2]) {
  static const uint32_t row_len[8] = {15, 13, 11, 9, 7, 5, 3, 1};
  uint32_t row = 7, col = 0, upper = 0, start = 0;
  for (uint32_t r = 0; r < 7; r++) { /* the reference's chain: `id <= 15`, `id <= 15 + 13`, ...: inclusive bounds, the column counted from the sum of the earlier rows */
    upper += row_len[r];
    if (id <= upper) { row = r; col = (id - start) >> 1; break; }
    start += row_len[r];
  }
  const int is_top = (id & 1u) == (row & 1u);
  b0[0] = (float) row; b0[1] = (float) (col + 1);
  b1[0] = (float) (row + 1); b1[1] = (float) col;
  if (is_top) { b2[0] = (float) row; b2[1] = (float) col; } else { b2[0] = (float) (row + 1); b2[1] = (float) (col + 1); }
  for (int k = 0; k < 2; k++) { b0[k] *= 1.0f / 8.0f; b1[k] *= 1.0f / 8.0f; b2[k] *= 1.0f / 8.0f; }
}

static float sc_max_emission(const OracleScene* ts, uint32_t tex, UV vertex, UV edge1, UV edge2, uint32_t microtriangle_id) {
  if (tex >= ts->num_textures) return 0.0f; /* texture_is_valid */
  float b0[2], b1[2], b2[2];
  sc_microtriangle_bary(microtriangle_id, b0, b1, b2);
  const UV m0 = {vertex.u + b0[0] * edge1.u + b0[1] * edge2.u, vertex.v + b0[0] * edge1.v + b0[1] * edge2.v};
  const UV m1 = {vertex.u + b1[0] * edge1.u + b1[1] * edge2.u, vertex.v + b1[0] * edge1.v + b1[1] * edge2.v};
  const UV m2 = {vertex.u + b2[0] * edge1.u + b2[1] * edge2.u, vertex.v + b2[0] * edge1.v + b2[1] * edge2.v};
  const UV e1 = {m1.u - m0.u, m1.v - m0.v}, e2 = {m2.u - m0.u, m2.v - m0.v};
  const uint32_t* t = ts->texture_table + 4 * (size_t) tex;
  const float steps_u = fmaxf(fabsf(e1.u), fabsf(e2.u)) * (float) (uint16_t) t[1]; /* DeviceTextureObject.width / .height are u16 */
  const float steps_v = fmaxf(fabsf(e1.v), fabsf(e2.v)) * (float) (uint16_t) t[2];
  const float steps = ceilf(fmaxf(steps_u, steps_v));
  const float step_size = 1.0f / steps;
  float mr = 0.0f, mg = 0.0f, mb = 0.0f;
  for (float a = 0.0f; a < 1.0f; a += step_size) {
    for (float b = 0.0f; a + b < 1.0f; b += step_size) {
      const UV uv = {m0.u + a * e1.u + b * e2.u, m0.v + a * e1.v + b * e2.v};
      const float4_t texel = texture_load(ts, tex, uv, true, f4(0.0f, 0.0f, 0.0f, 0.0f));
      mr = fmaxf(mr, texel.x); mg = fmaxf(mg, texel.y); mb = fmaxf(mb, texel.z);
    }
  }
  return fmaxf(mr, fmaxf(mg, mb)); /* color_importance, math.cuh:1066-1068 */
}

/* the kernel reads the triangle's texture coordinates from the DEVICE triangle (truncated bfloat16 pairs), one warp per light, two micro-triangles per lane */
static float sc_triangle_intensity(const OracleScene* ts, uint32_t tex, const uint32_t tri_tex[4]) {
  const UV v0 = uv_unpack(tri_tex[0]), v1 = uv_unpack(tri_tex[1]), v2 = uv_unpack(tri_tex[2]);
  const UV e1 = {v1.u - v0.u, v1.v - v0.v}, e2 = {v2.u - v0.u, v2.v - v0.v};
  float best = 0.0f; /* warp_reduce_max over fmaxf pairs: a maximum is a maximum in any order */
  for (uint32_t id = 0; id < 64; id++) best = fmaxf(best, sc_max_emission(ts, tex, v0, e1, e2, id));
  return best;
}

/* ------------------------------------------------------------------------------------------------------------------------------------
 * The scalar conversions and the derived parameters
 * ---------------------------------------------------------------------------------------------------------------------------------- */

/* cuda/math.cuh:1189-1232 (the reference evaluates this on the device per ray; the product once per scene) */
static void sc_jendersie_eon(float d, float out[4]) {
  float g_hg = 0.0f, g_d = 0.0f, alpha = 0.0f, w_d = 0.0f; /* beyond 50 micrometres the reference leaves its struct unset */
  if (d >= 5.0f && d <= 50.0f) {
    g_hg = expf(-0.0990567f / (d - 1.67154f));
    g_d = expf(-(2.20679f / (d + 3.91029f)) - 0.428934f);
    alpha = expf(3.62489f - (8.29288f / (d + 5.52825f)));
    w_d = expf(-(0.599085f / (d - 0.641583f)) - 0.665888f);
  }
  else if (d >= 1.5f && d < 5.0f) {
    g_hg = 0.0604931f * logf(logf(d)) + 0.940256f;
    g_d = 0.500411f - (0.081287f / (-2.0f * logf(d) + tanf(logf(d)) + 1.27551f));
    alpha = 7.30354f * logf(d) + 6.31675f;
    w_d = 0.026914f * (logf(d) - cosf(5.68947f * (logf(logf(d)) - 0.0292149f))) + 0.376475f;
  }
  else if (d >= 0.1f && d < 1.5f) {
    g_hg = 0.862f - 0.143f * logf(d) * logf(d);
    g_d = 0.379685f * cosf(1.19692f * cosf(((logf(d) - 0.238604f) * (logf(d) + 1.00667f)) / (0.507522f - 0.15677f * logf(d))) + 1.37932f * logf(d) + 0.0625835f) + 0.344213f;
    alpha = 250.0f;
    w_d = 0.146209f * cosf(3.38707f * logf(d) + 2.11193f) + 0.316072f + 0.0778917f * logf(d);
  }
  else if (d < 0.1f) {
    g_hg = 13.8f * d * d;
    g_d = 1.1456f * d * sinf(9.29044f * d);
    alpha = 250.0f;
    w_d = 0.252977f - 312.983f * powf(d, 4.3f);
  }
  out[0] = g_hg; out[1] = g_d; out[2] = alpha; out[3] = w_d;
}

/* device_structs.c:131-171: direction from (azimuth, altitude) in double, scaled to the body's distance, seen from the point sky.geometry_offset above the
 * earth's centre (sky_defines.h:4-8) */
static void sc_body_position(float azimuth, float altitude, double distance, const float offset[3], float out[3]) {
  double x = cos(azimuth) * cos(altitude);
  double y = sin(altitude);
  double z = sin(azimuth) * cos(altitude);
  const double scale = 1.0 / (sqrt(x * x + y * y + z * z));
  x *= scale * distance;
  y *= scale * distance;
  z *= scale * distance;
  y -= 6371.0f;
  x -= offset[0];
  y -= offset[1];
  z -= offset[2];
  out[0] = (float) x; out[1] = (float) y; out[2] = (float) z;
}

void oracle_scene_constants(const OSceneEntities* in, OSceneConstants* out) {
  memset(out, 0, sizeof(*out));
  out->width = in->width << in->supersampling;
  out->height = in->height << in->supersampling;
  const SQuat q = sc_euler_to_quaternion(in->cam_rotation);
  out->cam_rotation[0] = q.x; out->cam_rotation[1] = q.y; out->cam_rotation[2] = q.z; out->cam_rotation[3] = q.w;
  sc_body_position(in->sky_azimuth, in->sky_altitude, 149597870.0f, in->sky_geometry_offset, out->sky_sun_pos);
  sc_body_position(in->sky_moon_azimuth, in->sky_moon_altitude, 384399.0f, in->sky_geometry_offset, out->sky_moon_pos);
  sc_jendersie_eon(in->sky_mie_diameter, out->sky_mie_phase);
  sc_jendersie_eon(in->fog_droplet_diameter, out->fog_phase);
  sc_jendersie_eon(in->particles_phase_diameter, out->particles_phase);
  sc_jendersie_eon(in->cloud_droplet_diameter, out->cloud_phase);
  out->particles_direction[0] = cosf(in->particles_direction_azimuth) * cosf(in->particles_direction_altitude); /* math.cuh:781-788 */
  out->particles_direction[1] = sinf(in->particles_direction_altitude);
  out->particles_direction[2] = sinf(in->particles_direction_azimuth) * cosf(in->particles_direction_altitude);
  /* cuda/ocean_utils.cuh:300-385: Jerlov water types I, IA, IB, II, III, 1C, 3C, 5C, 7C, 9C (structs.h:212-221) */
  static const float scat[10][3] = {{0.001f, 0.002f, 0.004f}, {0.002f, 0.004f, 0.007f}, {0.045f, 0.054f, 0.07f}, {0.27f, 0.365f, 0.516f}, {0.737f, 0.998f, 1.413f},
                                    {0.274f, 0.372f, 0.526f}, {0.904f, 1.071f, 1.532f}, {3.589f, 1.382f, 1.857f}, {1.772f, 2.394f, 3.376f}, {2.347f, 3.18f, 4.496f}};
  static const float absb[10][3] = {{0.309f, 0.053f, 0.009f}, {0.309f, 0.054f, 0.014f}, {0.309f, 0.054f, 0.015f}, {0.31f, 0.054f, 0.016f}, {0.31f, 0.056f, 0.031f},
                                    {0.316f, 0.067f, 0.105f}, {0.508f, 0.052f, 0.161f}, {4.638f, 0.222f, 0.216f}, {0.351f, 0.188f, 0.574f}, {0.398f, 0.349f, 0.995f}};
  static const float molw[10] = {0.93f, 0.44f, 0.06f, 0.007f, 0.003f, 0.005f, 0.003f, 0.001f, 0.0f, 0.0f};
  if (in->ocean_water_type < 10u) {
    memcpy(out->ocean_scattering, scat[in->ocean_water_type], 12);
    memcpy(out->ocean_absorption, absb[in->ocean_water_type], 12);
    out->ocean_molecular_weight = molw[in->ocean_water_type];
  }
  out->ocean_caustics_ris_sample_count = (in->ocean_caustics_ris_sample_count > 1u ? in->ocean_caustics_ris_sample_count : 1u) - 1u;
  for (int l = 0; l < 3; l++) { out->cloud_wind[l][0] = cosf(in->cloud_wind_angle[l]); out->cloud_wind[l][1] = sinf(in->cloud_wind_angle[l]); }
}

/* ------------------------------------------------------------------------------------------------------------------------------------ */
int oracle_scene_encode(const OSceneInput* in, OSceneOutput* out) {
  memset(out, 0, sizeof(*out));
  /* ---- geometry containers: meshes back to back (device_mesh.c:19-51 uploads one buffer pair per mesh; the product concatenates them) ---- */
  out->mesh_tri_offset = (uint32_t*) calloc((size_t) in->num_meshes + 1, 4);
  uint32_t total = 0;
  for (uint32_t m = 0; m < in->num_meshes; m++) { out->mesh_tri_offset[m] = total; total += in->meshes[m].triangle_count; }
  out->mesh_tri_offset[in->num_meshes] = total;
  out->total_triangles = total;
  out->vertices = (float*) calloc((size_t) total * 12 + 4, 4);
  out->tri_tex = (uint32_t*) calloc((size_t) total * 4 + 4, 4);
  for (uint32_t m = 0; m < in->num_meshes; m++) {
    const OSceneMesh* mesh = in->meshes + m;
    for (uint32_t t = 0; t < mesh->triangle_count; t++) {
      const size_t g = (size_t) out->mesh_tri_offset[m] + t;
      for (int k = 0; k < 3; k++) { /* device_struct_vertex_convert, device_structs.c:349-361 */
        float* v = out->vertices + (g * 3 + k) * 4;
        v[0] = mesh->positions[t * 9 + 3 * k + 0];
        v[1] = mesh->positions[t * 9 + 3 * k + 1];
        v[2] = mesh->positions[t * 9 + 3 * k + 2];
        const uint32_t pn = sc_pack_normal(mesh->normals + t * 9 + 3 * k);
        memcpy(v + 3, &pn, 4);
      }
      uint32_t* tt = out->tri_tex + g * 4; /* device_struct_triangle_texture_convert, :363-374 */
      tt[0] = sc_pack_uv(mesh->uvs[t * 6 + 0], mesh->uvs[t * 6 + 1]);
      tt[1] = sc_pack_uv(mesh->uvs[t * 6 + 2], mesh->uvs[t * 6 + 3]);
      tt[2] = sc_pack_uv(mesh->uvs[t * 6 + 4], mesh->uvs[t * 6 + 5]);
      tt[3] = mesh->material_ids[t]; /* u16 material id, u16 padding */
    }
  }
  out->materials = (uint16_t*) calloc((size_t) in->num_materials * 16 + 16, 2);
  for (uint32_t i = 0; i < in->num_materials; i++) sc_encode_material(in->materials + i, out->materials + (size_t) i * 16);
  out->instance_mesh_ids = (uint32_t*) calloc((size_t) in->num_instances + 1, 4);
  out->instance_transforms = (float*) calloc((size_t) in->num_instances * 8 + 8, 4);
  for (uint32_t i = 0; i < in->num_instances; i++) {
    const OSceneInstance* inst = in->instances + i;
    out->instance_mesh_ids[i] = (inst->active && inst->mesh_id < in->num_meshes) ? inst->mesh_id : 0xFFFFFFFFu;
    sc_encode_transform(inst, out->instance_transforms + (size_t) i * 8);
  }

  /* ---- light tree: caches (device_light.c:1615-1863), fragments (:2020-2113), collect (:2155-2197) ---- */
  OracleScene ts; memset(&ts, 0, sizeof(ts));
  ts.num_textures = in->num_textures; ts.texture_table = in->texture_table; ts.texels = in->texels;
  /* material cache: has_emission <=> intensity > 0 (:1823-1863) */
  float* mat_intensity = (float*) calloc((size_t) in->num_materials + 1, 4);
  uint8_t* mat_textured = (uint8_t*) calloc((size_t) in->num_materials + 1, 1);
  for (uint32_t i = 0; i < in->num_materials; i++) {
    const OSceneMaterial* m = in->materials + i;
    float intensity = 0.0f;
    if (m->emission_active) {
      mat_textured[i] = m->luminance_tex != 0xFFFF;
      intensity = mat_textured[i] ? m->emission_scale : fmaxf(m->emission[0], fmaxf(m->emission[1], m->emission[2]));
    }
    mat_intensity[i] = intensity;
  }
  /* count an upper bound of fragments */
  size_t cap = 0;
  for (uint32_t i = 0; i < in->num_instances; i++) {
    const OSceneInstance* inst = in->instances + i;
    if (inst->active && inst->mesh_id < in->num_meshes) cap += in->meshes[inst->mesh_id].triangle_count;
  }
  SFragment* fragments = (SFragment*) calloc(cap + 1, sizeof(SFragment));
  float* frag_avg = (float*) calloc(cap + 1, 4);
  uint32_t fragments_count = 0;
  /* per-mesh average intensities of textured emitters (one integration per mesh triangle, :1904-1950), computed on first use */
  float** mesh_avg = (float**) calloc((size_t) in->num_meshes + 1, sizeof(float*));
  for (uint32_t i = 0; i < in->num_instances; i++) {
    const OSceneInstance* inst = in->instances + i;
    if (!inst->active || inst->mesh_id >= in->num_meshes) continue; /* :2170, :2049 */
    const OSceneMesh* mesh = in->meshes + inst->mesh_id;
    /* the mesh cache's material slots in order of first appearance (:1647-1668) */
    uint16_t* slots = (uint16_t*) calloc((size_t) mesh->triangle_count + 1, 2);
    uint32_t num_slots = 0;
    int mesh_has_emission = 0;
    for (uint32_t t = 0; t < mesh->triangle_count; t++) {
      const uint16_t id = mesh->material_ids[t];
      uint32_t s = 0;
      for (; s < num_slots; s++) if (slots[s] == id) break;
      if (s == num_slots) { slots[num_slots++] = id; if (id < in->num_materials && mat_intensity[id] > 0.0f) mesh_has_emission = 1; }
    }
    if (!mesh_has_emission) { free(slots); continue; } /* :2052 */
    const SQuat rq = sc_euler_to_quaternion(inst->rotation);
    const V4 offset = v4_set(inst->translation[0], inst->translation[1], inst->translation[2], 0.0f);
    const V4 scale = v4_set(inst->scale[0], inst->scale[1], inst->scale[2], 1.0f);
    const V4 rotation = v4_set(-rq.x, -rq.y, -rq.z, rq.w);
    for (uint32_t s = 0; s < num_slots; s++) {
      const uint16_t material_id = slots[s];
      if (material_id >= in->num_materials) continue;
      if (!(mat_intensity[material_id] > 0.0f)) continue;
      for (uint32_t t = 0; t < mesh->triangle_count; t++) {
        if (mesh->material_ids[t] != material_id) continue;
        float average_intensity = 1.0f; /* :1690 */
        if (mat_textured[material_id]) {
          if (!mesh_avg[inst->mesh_id]) {
            mesh_avg[inst->mesh_id] = (float*) malloc(sizeof(float) * ((size_t) mesh->triangle_count + 1));
            for (uint32_t k = 0; k < mesh->triangle_count; k++) mesh_avg[inst->mesh_id][k] = -1.0f;
          }
          if (mesh_avg[inst->mesh_id][t] < 0.0f)
            mesh_avg[inst->mesh_id][t] = sc_triangle_intensity(&ts, in->materials[material_id].luminance_tex, out->tri_tex + ((size_t) out->mesh_tri_offset[inst->mesh_id] + t) * 4);
          average_intensity = mesh_avg[inst->mesh_id][t];
        }
        const float* p = mesh->positions + (size_t) t * 9;
        const V4 cv0 = v4_set(p[0], p[1], p[2], 0.0f), cv1 = v4_set(p[3], p[4], p[5], 0.0f), cv2 = v4_set(p[6], p[7], p[8], 0.0f);
        const V4 vertex = v4_add(v4_mul(v4_rotate_quaternion(cv0, rotation), scale), offset);
        const V4 vertex1 = v4_add(v4_mul(v4_rotate_quaternion(cv1, rotation), scale), offset);
        const V4 vertex2 = v4_add(v4_mul(v4_rotate_quaternion(cv2, rotation), scale), offset);
        const V4 cross = v4_cross(v4_sub(vertex1, vertex), v4_sub(vertex2, vertex));
        const float area = 0.5f * v4_norm2(cross);
        if (area == 0.0f || average_intensity == 0.0f) continue;
        SFragment f;
        f.low = v4_min(vertex, v4_min(vertex1, vertex2));
        f.high = v4_max(vertex, v4_max(vertex1, vertex2));
        f.middle = v4_scale(v4_add(vertex, v4_add(vertex1, vertex2)), 1.0f / 3.0f);
        f.v0 = vertex; f.v1 = vertex1; f.v2 = vertex2;
        f.power = mat_intensity[material_id] * area * average_intensity;
        f.intensity = mat_intensity[material_id] * average_intensity;
        f.instance_id = i;
        f.tri_id = t;
        frag_avg[fragments_count] = average_intensity;
        fragments[fragments_count++] = f;
      }
    }
    free(slots);
  }
  /* remember the average intensity by (instance, triangle) through the permutations: keep it inside the fragment's unused `intensity` twin */
  for (uint32_t i = 0; i < fragments_count; i++) fragments[i].intensity = frag_avg[i];

  /* ---- build (:2236-2265) ---- */
  uint32_t bcount = 0;
  SBinaryNode* bnodes = sc_build_binary(fragments, fragments_count, &bcount);
  out->num_lights = fragments_count;
  if (bcount > 0) {
    SNode* nodes = sc_build_traversal_structure(fragments, bnodes, bcount);
    SCollapse cw; memset(&cw, 0, sizeof(cw));
    cw.binary_node_indices = (uint32_t*) calloc((size_t) bcount + 1, 4);
    cw.new_fragments = (uint32_t*) malloc(sizeof(uint32_t) * ((size_t) fragments_count + 1));
    memset(cw.new_fragments, 0xFF, sizeof(uint32_t) * ((size_t) fragments_count + 1));
    cw.nodes = (uint8_t*) calloc((size_t) bcount + 1, 64);
    sc_collapse_root(&cw, nodes);
    sc_collapse_nodes(&cw, nodes);
    if (cw.triangles_ptr != fragments_count) { free(cw.binary_node_indices); free(cw.new_fragments); free(cw.nodes); free(cw.root); free(nodes); free(bnodes); return 2; } /* a light was lost (:1180) */
    SFragment* swap = (SFragment*) malloc(sizeof(SFragment) * ((size_t) fragments_count + 1));
    memcpy(swap, fragments, sizeof(SFragment) * fragments_count);
    for (uint32_t i = 0; i < fragments_count; i++) fragments[i] = swap[cw.new_fragments[i]]; /* :1199-1201 */
    free(swap);
    out->light_tree_root = cw.root; out->light_tree_root_bytes = cw.root_bytes;
    out->light_tree_nodes = cw.nodes; out->num_light_tree_nodes = cw.num_nodes;
    free(cw.binary_node_indices); free(cw.new_fragments); free(nodes);
  }
  free(bnodes);
  /* ---- finalize (:1226-1288): handle map and the light-only BVH's vertices in the new light order ---- */
  out->light_tri_handles = (uint32_t*) calloc((size_t) fragments_count * 2 + 2, 4);
  out->light_bvh_tris = (float*) calloc((size_t) fragments_count * 12 + 12, 4);
  out->light_intensities = (float*) calloc((size_t) fragments_count + 1, 4);
  for (uint32_t i = 0; i < fragments_count; i++) {
    out->light_tri_handles[2 * i] = fragments[i].instance_id;
    out->light_tri_handles[2 * i + 1] = fragments[i].tri_id;
    memcpy(out->light_bvh_tris + (size_t) i * 12, fragments[i].v0.d, 16);
    memcpy(out->light_bvh_tris + (size_t) i * 12 + 4, fragments[i].v1.d, 16);
    memcpy(out->light_bvh_tris + (size_t) i * 12 + 8, fragments[i].v2.d, 16);
    out->light_intensities[i] = fragments[i].intensity;
  }
  for (uint32_t m = 0; m < in->num_meshes; m++) free(mesh_avg[m]);
  free(mesh_avg); free(fragments); free(frag_avg); free(mat_intensity); free(mat_textured);
  return 0;
}

void oracle_scene_free(OSceneOutput* out) {
  free(out->mesh_tri_offset); free(out->vertices); free(out->tri_tex); free(out->instance_mesh_ids); free(out->instance_transforms); free(out->materials);
  free(out->light_tree_root); free(out->light_tree_nodes); free(out->light_tri_handles); free(out->light_bvh_tris); free(out->light_intensities);
  memset(out, 0, sizeof(*out));
}
