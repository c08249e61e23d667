// Software ray queries on the two-level BVH4 that replaces the reference's OptiX acceleration structures.
// Semantics restated from the reference's OptiX programs:
//   closest hit  optix/optix_kernel_raytrace.cu:82-95, cuda/optix_anyhit.cuh:15-31, cuda/optix_closesthit.cuh:15-26
//   shadow       cuda/optix_common.cuh:76-106, cuda/optix_anyhit.cuh:49-139, cuda/optix_closesthit.cuh:44-58
//   light query  cuda/optix_anyhit.cuh:145-205 (reservoir over all lights hit), cuda/direct_lighting.cuh:596-611
// Triangle test: cuda/math.cuh:1337-1358 on the object-space ray (instance transform world = S*R*v + T, math.cuh:459-489),
// so hit distances are the same numbers in world and object space.
// Results do not depend on traversal order: ties are broken by (t, instance, triangle) and the light pick is a function of
// the candidate set only (see light_query).
#pragma once

#include "dev_light.h"

namespace lum {

constexpr uint32_t kHitSky       = 0xFFFFFFFEu;  // cuda/utils.cuh:50-64
constexpr uint32_t kLeaveInstance = 0xFFFFFFFDu;  // stack marker: back from a bottom-level BVH to the top level
constexpr uint32_t kNoInstance    = 0xFFFFFFFFu;
constexpr int kStackSize          = 96;  // top level + marker + bottom level; the host builder caps each BVH4 at 20 levels

struct RayStats { uint32_t nodes, tris; };

LUM_DEV float safe_inv(float d) { return (fabsf(d) < 1e-30f) ? copysignf(1e30f, d) : 1.0f / d; }

// Slab test of the four children of a node: t = lo * inv - o * inv as one fused multiply-add per plane (the box test only
// decides what gets visited, never a result, so it does not have to follow the IEEE-only contract). Boxes are padded by the
// builder and the comparison is relaxed so that a triangle accepted by the exact test is never culled.
LUM_DEV uint32_t test_children(const Bvh4Node& n, V3 inv, V3 oi, float tmax, float tnear[4]) {
  uint32_t mask = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float ax = __builtin_fmaf(n.lo_x[k], inv.x, -oi.x), bx = __builtin_fmaf(n.hi_x[k], inv.x, -oi.x);
    const float ay = __builtin_fmaf(n.lo_y[k], inv.y, -oi.y), by = __builtin_fmaf(n.hi_y[k], inv.y, -oi.y);
    const float az = __builtin_fmaf(n.lo_z[k], inv.z, -oi.z), bz = __builtin_fmaf(n.hi_z[k], inv.z, -oi.z);
    const float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fminf(fmaxf(az, bz), tmax));
    tnear[k] = t0;
    if (n.child[k] != kBvhEmpty && t0 <= t1 * 1.000004f + 1e-30f) mask |= 1u << k;
  }
  return mask;
}

struct TraversalStack {
  uint32_t node[kStackSize];
  float tnear[kStackSize];
  int sp = 0;
  LUM_DEV void push(uint32_t n, float t) { if (sp < kStackSize) { node[sp] = n; tnear[sp] = t; sp++; } }
};

// Visits the children of `n` that the ray may touch: the nearest becomes `cur`, the others are pushed far-to-near.
LUM_DEV bool descend(const Bvh4Node& n, V3 inv, V3 oi, float tmax, TraversalStack& stk, uint32_t& cur) {
  float tn[4];
  uint32_t mask = test_children(n, inv, oi, tmax, tn);
  if (mask == 0) return false;
  int near = -1;
  float near_t = kFltMax;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (((mask >> k) & 1u) && tn[k] < near_t) { near_t = tn[k]; near = k; }
  if (near < 0) near = __ffs((int) mask) - 1;  // all entry distances are FLT_MAX/NaN: any order
  mask &= ~(1u << near);
  while (mask) {
    int far = -1;
    float ft = -1.0f;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (((mask >> k) & 1u) && tn[k] >= ft) { ft = tn[k]; far = k; }
    if (far < 0) far = __ffs((int) mask) - 1;
    mask &= ~(1u << far);
    stk.push(n.child[far], tn[far]);
  }
  cur = n.child[near];
  return true;
}

// Single-level traversal (light BVH). `on_leaf(first, count, tmax)` may shrink tmax and returns true to stop the query.
template <typename LeafFn>
LUM_DEV bool traverse_bvh4(const Bvh4Node* __restrict__ nodes, V3 o, V3 d, float& tmax, RayStats& st, LeafFn&& on_leaf) {
  const V3 inv = v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  const V3 oi = o * inv;
  TraversalStack stk;
  uint32_t cur = 0;
  while (true) {
    bool pop = true;
    if (cur & kBvhLeafBit) {
      if (on_leaf(cur & 0x0FFFFFFFu, ((cur >> 28) & 0x7u) + 1u, tmax)) return true;
    }
    else {
      st.nodes++;
      pop = !descend(nodes[cur], inv, oi, tmax, stk, cur);
    }
    if (pop) {
      do {
        if (stk.sp == 0) return false;
        stk.sp--;
        cur = stk.node[stk.sp];
      } while (!(stk.tnear[stk.sp] <= tmax * 1.000004f + 1e-30f));
    }
  }
}

LUM_DEV V3 tri_p0(const BvhTri& t) { return v3(t.p0[0], t.p0[1], t.p0[2]); }
LUM_DEV V3 tri_e1(const BvhTri& t) { return v3(t.e1[0], t.e1[1], t.e1[2]); }
LUM_DEV V3 tri_e2(const BvhTri& t) { return v3(t.e2[0], t.e2[1], t.e2[2]); }

// Two-level traversal, "while-while" form: every lane first walks inner nodes until it holds a leaf (or is done), then all lanes
// handle their leaves together; lanes never wait inside a per-instance sub-loop. Top-level leaves hold exactly one instance;
// entering it pushes a marker, switches the node base and maps the ray with the instance's world->object matrix (an affine map
// preserves distances along the ray, so `tmax` and the stacked entry distances stay valid across levels).
// `on_tris(inst, mesh, tris, first, count, o, d, tmax)` returns true to stop.
constexpr uint32_t kTraversalDone = 0xFFFFFFFCu;

template <typename TriFn>
LUM_DEV bool traverse_scene(const DeviceScene& sc, V3 wo, V3 wd, float& tmax, RayStats& st, TriFn&& on_tris) {
  TraversalStack stk;
  V3 o = wo, d = wd;
  V3 inv = v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
  V3 oi = o * inv;
  const Bvh4Node* __restrict__ nodes = sc.tlas_nodes;
  const BvhTri* __restrict__ tris = nullptr;
  uint32_t inst = kNoInstance, mesh = 0;
  uint32_t cur = 0;

  auto pop = [&]() {
    while (true) {
      if (stk.sp == 0) { cur = kTraversalDone; return; }
      stk.sp--;
      cur = stk.node[stk.sp];
      if (cur == kLeaveInstance) {
        inst = kNoInstance;
        nodes = sc.tlas_nodes;
        o = wo; d = wd;
        inv = v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
        oi = o * inv;
        continue;
      }
      if (stk.tnear[stk.sp] <= tmax * 1.000004f + 1e-30f) return;
    }
  };

  while (true) {
    // inner nodes (top or bottom level) until this lane holds a leaf
    while (!(cur & kBvhLeafBit)) {
      st.nodes++;
      if (!descend(nodes[cur], inv, oi, tmax, stk, cur)) pop();
    }
    if (cur == kTraversalDone) return false;
    if (inst == kNoInstance) {
      inst = sc.tlas_prims[cur & 0x0FFFFFFFu];
      mesh = sc.instance_mesh_ids[inst];
      const float4 r0 = sc.instance_inv[3 * inst], r1 = sc.instance_inv[3 * inst + 1], r2 = sc.instance_inv[3 * inst + 2];
      const float px = wo.x - r0.w, py = wo.y - r1.w, pz = wo.z - r2.w;
      o = v3(mat_row_apply(r0.x, r0.y, r0.z, px, py, pz), mat_row_apply(r1.x, r1.y, r1.z, px, py, pz), mat_row_apply(r2.x, r2.y, r2.z, px, py, pz));
      d = v3(mat_row_apply(r0.x, r0.y, r0.z, wd.x, wd.y, wd.z), mat_row_apply(r1.x, r1.y, r1.z, wd.x, wd.y, wd.z),
             mat_row_apply(r2.x, r2.y, r2.z, wd.x, wd.y, wd.z));
      inv = v3(safe_inv(d.x), safe_inv(d.y), safe_inv(d.z));
      oi = o * inv;
      nodes = sc.blas_nodes + sc.mesh_node_offset[mesh];
      tris = sc.blas_tris + sc.mesh_bvhtri_offset[mesh];
      stk.push(kLeaveInstance, 0.0f);
      cur = 0;
    }
    else {
      if (on_tris(inst, mesh, tris, cur & 0x0FFFFFFFu, ((cur >> 28) & 0x7u) + 1u, o, d, tmax)) return true;
      pop();
    }
  }
}

struct Hit { uint32_t instance_id, tri_id; float t; };

// Nearest hit in [0, FLT_MAX); optionally ignoring the triangle the path is leaving (STATE_FLAG_USE_IGNORE_HANDLE).
LUM_DEV Hit closest_hit(const DeviceScene& sc, V3 origin, V3 dir, bool use_ignore, uint32_t ign_inst, uint32_t ign_tri, RayStats& st) {
  Hit best{kHitSky, 0u, kFltMax};
  float tmax = kFltMax;
  traverse_scene(sc, origin, dir, tmax, st, [&](uint32_t inst, uint32_t, const BvhTri* __restrict__ tris, uint32_t first, uint32_t count, V3 o, V3 d, float& tm) {
    for (uint32_t j = 0; j < count; j++) {
      const BvhTri tr = tris[first + j];
      st.tris++;
      if (use_ignore && inst == ign_inst && tr.id == ign_tri) continue;
      F2 uv;
      const float t = intersect_triangle(tri_p0(tr), tri_e1(tr), tri_e2(tr), o, d, uv);
      if (t < best.t || (t == best.t && t != kFltMax && (inst < best.instance_id || (inst == best.instance_id && tr.id < best.tri_id)))) {
        best.instance_id = inst; best.tri_id = tr.id; best.t = t; tm = t;
      }
    }
    return false;
  });
  if (best.t == kFltMax) { best.instance_id = kHitSky; best.tri_id = 0; }
  return best;
}

// Transparency along (eps, dist): product over crossed surfaces, zero as soon as one is opaque. Skips the sampled light
// (`target`) and the surface being shaded (`self`).
LUM_DEV Col shadow_query(const DeviceScene& sc, V3 origin, V3 dir, float dist, uint32_t tgt_inst, uint32_t tgt_tri, uint32_t self_inst,
                         uint32_t self_tri, RayStats& st) {
  Col through = splat(1.0f);
  float tmax = dist;
  const bool blocked = traverse_scene(sc, origin, dir, tmax, st, [&](uint32_t inst, uint32_t mesh, const BvhTri* __restrict__ tris, uint32_t first, uint32_t count, V3 o, V3 d, float&) {
    for (uint32_t j = 0; j < count; j++) {
      const BvhTri tr = tris[first + j];
      st.tris++;
      if ((inst == tgt_inst && tr.id == tgt_tri) || (inst == self_inst && tr.id == self_tri)) continue;
      F2 uv;
      const float t = intersect_triangle(tri_p0(tr), tri_e1(tr), tri_e2(tr), o, d, uv);
      if (!(t > kEps && t < dist)) continue;
      const Material m = load_material(sc, sc.tri_tex[sc.mesh_tri_offset[mesh] + tr.id].w & 0xFFFFu);
      const bool colored = (m.flags & kDMatColoredTransparency) != 0;
      if (m.alpha == 1.0f) return true;
      if (m.alpha == 0.0f && !colored) continue;
      const float tp = 1.0f - m.alpha;
      through = through * (colored ? m.albedo * tp : splat(tp));
    }
    return false;
  });
  return blocked ? splat(0.0f) : through;
}

// Light-BVH query on (eps, FLT_MAX). OptiX leaves the any-hit order unspecified, so the reference's reservoir is restated
// order-independently: pass 0 finds t* = nearest opaque light; pass 1 counts the lights with eps < t <= t* (except `self`
// and fully transparent uncoloured ones) and picks the one minimising squares32(0x9E3779B9*id + random bits), a uniform
// choice driven by the same random number.
LUM_DEV uint32_t light_query(const DeviceScene& sc, V3 origin, V3 dir, uint32_t self_inst, uint32_t self_tri, float random, uint32_t& num_hits,
                             RayStats& st) {
  float tstar = kFltMax;
  uint32_t best_id = kLightIdInvalid, best_key = 0xFFFFFFFFu, n = 0;
  for (int pass = 0; pass < 2; pass++) {
    float tmax = tstar;
    traverse_bvh4(sc.light_nodes, origin, dir, tmax, st, [&](uint32_t first, uint32_t count, float& tm) {
      for (uint32_t j = 0; j < count; j++) {
        const BvhTri tr = sc.light_tris[first + j];
        st.tris++;
        const uint2 handle = sc.light_tri_handles[tr.id];
        if (handle.x == self_inst && handle.y == self_tri) continue;
        F2 uv;
        const float t = intersect_triangle(tri_p0(tr), tri_e1(tr), tri_e2(tr), origin, dir, uv);
        if (!(t > kEps && t != kFltMax && t <= tstar)) continue;
        const uint32_t mesh = sc.instance_mesh_ids[handle.x];
        const Material m = load_material(sc, sc.tri_tex[sc.mesh_tri_offset[mesh] + handle.y].w & 0xFFFFu);
        if (m.alpha == 0.0f && (m.flags & kDMatColoredTransparency) == 0) continue;
        if (pass == 0) { if (m.alpha == 1.0f && t < tstar) { tstar = t; tm = t; } }
        else {
          n++;
          const uint32_t key = squares32(0xfcbd6e15u, 0x9E3779B9u * tr.id + fbits(random));
          if (key < best_key || (key == best_key && tr.id < best_id)) { best_key = key; best_id = tr.id; }
        }
      }
      return false;
    });
  }
  num_hits = n;
  return best_id;
}

}  // namespace lum
