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

// The root's children as the kernels read them: dequantised once at scene upload (core.hip: mean = byte * 2^e + base, sigma = byte * 2^e_sigma, power
// = the 16-bit integer, all exact in binary32, so these are the numbers the device used to derive per vertex), eight floats per child
// {mean.xyz, sigma, power, 0, 0, 0}. The table is the same for every lane of every wave: it is read through the constant address space, i.e. with
// scalar loads into SGPRs, and costs the vector unit nothing.
typedef const float __attribute__((address_space(4)))* RootTable;
typedef const uint32_t __attribute__((address_space(4)))* RootHeader;
typedef float f2 __attribute__((ext_vector_type(2)));  // two binary32 numbers in a register pair: v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 work on both at once

// tree_importance for two children at a time. The generic form evaluates them one after the other; the surface vertex's form (the hot one: k_shade)
// runs the same operations in the same order on register pairs, so both give the bits of tree_importance().
template <class Ctx>
LUM_DEV f2 tree_importance_pair(const Ctx& g, f2 power, f2 mx, f2 my, f2 mz, f2 sd) {
  return f2{tree_importance(g, power.x, v3(mx.x, my.x, mz.x), sd.x), tree_importance(g, power.y, v3(mx.y, my.y, mz.y), sd.y)};
}
LUM_DEV f2 tree_importance_pair(const GeoContext& g, f2 power, f2 mx, f2 my, f2 mz, f2 sd) {
  const f2 px = mx - g.position.x, py = my - g.position.y, pz = mz - g.position.z;
  const f2 dist_sq = px * px + py * py + pz * pz;
  const f2 variance = sd * sd;
  const f2 denom = dist_sq + variance;
#if LUM_FAST
  // one v_rsq_f32 per child instead of v_rcp_f32 and v_sqrt_f32 (quarter-rate instructions, 64 children per vertex): 1 / x = rsq(x)^2
  const f2 rs = f2{__builtin_amdgcn_rsqf(denom.x), __builtin_amdgcn_rsqf(denom.y)};
  const f2 inv = rs * rs;
#else
  const f2 inv = f2{1.0f / denom.x, 1.0f / denom.y};
#endif
  const f2 r = power * inv;
  if ((g.params.flags & kMatSubstrateMask) == kMatTranslucent) return r;
  const f2 t = variance * inv;
  const f2 d = px * g.normal.x + py * g.normal.y + pz * g.normal.z;
#if LUM_FAST
  const f2 ndl = d * rs;
#else
  const f2 ndl = d * f2{sqrtf(inv.x), sqrtf(inv.y)};
#endif
  const f2 NdotL = f2{saturate(ndl.x), saturate(ndl.y)};
  return r * (NdotL * (1.0f - t) + t);
}

// Root pass (light_tree.cuh:191-255): one scan over <= 128 children feeds 8 independent resampling lanes.
// Two thirds of the pass are the eight reservoirs' updates (ris.cuh:138-148, per child and lane: accept = r < p; r = clamp(accept ? r / p : (r - p) / (1 - p))).
// Both quotients are formed for two lanes at a time with packed multiplies - they do not depend on the comparison - and the comparison only selects:
// the same operations on the same operands as the one-lane-at-a-time form, in 5.5 instead of 8 instructions per child and lane.
// Round 6, measured by ablation (one lane instead of eight: k_shade -30 % on the hall, whose 64 lights are all root children - profiles/r06_ab_experiments.txt): the
// seven other lanes' updates are the largest single item of k_shade, more than the eight candidates' sampling and BSDF work together (four candidates instead of
// eight: -13 %). Two changes since:
//  * both flavours: the clamp after the update is an upper bound only. The selected value is never negative - an accepted r is multiplied by a positive
//    reciprocal, a rejected one has r >= p, and r - p is exact or rounds to a non-negative number - so max(x, 0) never acts and v_min replaces v_med3. Same bits.
//  * fast flavour: the scan in threshold form (LUM_ROOT_THRESHOLD). With W_c the running sum after child c, and for a lane whose last accepted child left it the
//    random r0 at running sum W0: the rejections of children a .. b telescope to r_b = (r0 W_b - (W_b - W0)) / W0, so the lane accepts child c exactly when
//    t W_c > 1 with t = (1 - r0) / W0 - a constant of the lane until it accepts again - and an accepted child turns t into (t W_c - 1) * W_{c-1} / (w_c W_c),
//    whose second factor is the same for all eight lanes. Per child and lane: one packed fma and one packed multiply for two lanes, a comparison and three
//    selects, nothing for the rejected random number - 4 + 2 x 0.5 instead of 5 + 3 x 0.5 instructions, none of them a v_med3. In real arithmetic the picks
//    are the reference's; in binary32 the two forms agree with real arithmetic equally often (the scan is an expanding map - every accepted child multiplies the
//    lane's rounding error by 1 / p - so ~5 % of all 64-child scans end on another child than real arithmetic's in EITHER form, and in any build that rounds a
//    reciprocal differently: tests/test_root_pass_forms.py). The exact flavour keeps the reference's operation order and stays bit-identical to the oracle.
#ifndef LUM_ROOT_THRESHOLD
#define LUM_ROOT_THRESHOLD LUM_FAST
#endif
// ... and with it (LUM_ROOT_KEY) the picked child and its importance travel in ONE word per lane - the importance's bits with the child's index (< 128) in place of the
// seven lowest mantissa bits - so that an accepted child costs two selects (threshold constant, key) instead of three. The importance only serves the lane's
// selection probability, which is quantised to 20 bits right after the scan (light_tree.cuh:176-183); 16 mantissa bits of it move that probability by at most 2^-16 of itself.
// Measured (hall, same box, profiles/r06_ab_experiments.txt): threshold form k_shade -4.9 %, with the key another -1.9 % (+2.4 % and +0.8 % samples/s). LUM_ROOT_KEY=2 - the
// selects as integer instructions on vector registers, below - is level with 1 (the compiler then issues the packed products one lane at a time): kept as a record.
#ifndef LUM_ROOT_KEY
#define LUM_ROOT_KEY LUM_ROOT_THRESHOLD
#endif
LUM_DEV float clamp_random_top(float r) { return fminf(r, bitsf(0x3F7FFFFFu)); }  // clamp_random for r >= 0
template <class Ctx, class Smp>
LUM_DEV TreeWork tree_prepass(const DeviceScene& sc, const Ctx& g, const Smp& smp) {
  const RootHeader hp = (RootHeader) sc.light_tree_root;
  const uint4 h = make_uint4(hp[0], hp[1], hp[2], hp[3]);
  const uint32_t num_root_lights = h.y >> 16, num_children = ((h.z >> 16) & 0xFFu) * 8u;
  const RootTable table = (RootTable) sc.light_root_children;
  f2 lane_random[kLightTreeOutputs / 2];  // LUM_ROOT_THRESHOLD: the lane's random number until its first child, its threshold constant t from then on
  float lane_target[kLightTreeOutputs];
  uint32_t lane_pick[kLightTreeOutputs];
#pragma unroll
  for (uint32_t l = 0; l < kLightTreeOutputs; l++) {
    lane_random[l >> 1][l & 1] = smp.next1(TreeTargets<Ctx>::kPrepass + l);
    lane_target[l] = 0.0f;
    lane_pick[l] = 0;
  }
  float total = 0.0f, sum = 0.0f;
  for (uint32_t c = 0; c < num_children; c += 2) {
    const RootTable e = table + 8u * c;
    const f2 power = f2{e[4], e[12]};
    const f2 imp = tree_importance_pair(g, power, f2{e[0], e[8]}, f2{e[1], e[9]}, f2{e[2], e[10]}, f2{e[3], e[11]});
#pragma unroll
    for (uint32_t k = 0; k < 2; k++) {
#if LUM_ROOT_THRESHOLD
      // (a child without power has importance power * ... = 0, or NaN where the vertex sits on its mean with no variance, and v_max_f32 drops a NaN operand:
      //  the reference's separate test for power 0 selects what the maximum already gives; one raw v_max instead of a canonicalising pair, a comparison and a select)
      float target;
      asm("v_max_f32 %0, 0, %1" : "=v"(target) : "v"(imp[k]));
#else
      const float target = (power[k] == 0.0f) ? 0.0f : fmaxf(imp[k], 0.0f);
#endif
#if LUM_ROOT_THRESHOLD
      if (!(target > 0.0f)) continue;
      const float before = total;
      total += target;
      sum += target;
      if (before == 0.0f) {  // the vertex's first child with any importance: probability 1, every lane takes it and keeps its random number
        const float inv = 1.0f / target;
#pragma unroll
        for (uint32_t l = 0; l < kLightTreeOutputs; l++) {
          lane_random[l >> 1][l & 1] = (1.0f - lane_random[l >> 1][l & 1]) * inv;
#if LUM_ROOT_KEY
          lane_pick[l] = (fbits(target) & 0xFFFFFF80u) | (c + k);
#else
          lane_target[l] = target;
          lane_pick[l] = c + k;
#endif
        }
        continue;
      }
#if LUM_ROOT_KEY
      const uint32_t key = (fbits(target) & 0xFFFFFF80u) | (c + k);
#endif
      const float step = before / (target * total);  // W_{c-1} / (w_c W_c)
      const f2 w2 = f2{total, total}, s2 = f2{step, step};
#pragma unroll
      for (uint32_t q = 0; q < ((LUM_ABLATE_LIGHT & 4) ? 1u : kLightTreeOutputs / 2); q++) {
        const f2 excess = lane_random[q] * w2 - 1.0f;  // t W_c - 1 (one packed fma under the fast flavour's contraction): positive = accepted
        const f2 next = excess * s2;
#pragma unroll
        for (uint32_t hlf = 0; hlf < 2; hlf++) {
          const uint32_t l = 2 * q + hlf;
#if LUM_ROOT_KEY == 2
          // No comparison and no select: a rejected child's `next` is negative (or -0), and as an unsigned word a negative float is larger than every positive one - the
          // unsigned minimum keeps the lane's constant unless the child was accepted (an accepted child's constant is the smaller one: t' <= 1 / W_c < t) - and the
          // same sign, spread over a word, is the bit mask of the key's select (v_ashrrev_i32, v_bfi_b32, v_min_u32: three vector instructions on vector registers,
          // where a comparison writes a scalar mask that three selects then wait two cycles for). t W_c = 1 exactly counts as accepted here.
          // (as instructions: written as C the compiler turns the mask back into a comparison and a select, and splits the packed products)
          uint32_t rejected;
          asm("v_ashrrev_i32 %0, 31, %1" : "=v"(rejected) : "v"(next[hlf]));
          asm("v_bfi_b32 %0, %1, %0, %2" : "+v"(lane_pick[l]) : "v"(rejected), "v"(key));
          float kept = lane_random[q][hlf];
          asm("v_min_u32 %0, %0, %1" : "+v"(kept) : "v"(next[hlf]));
          lane_random[q][hlf] = kept;
          continue;
#endif
          const bool accept = excess[hlf] > 0.0f;
#if LUM_ROOT_KEY
          lane_pick[l] = accept ? key : lane_pick[l];
#else
          lane_target[l] = accept ? target : lane_target[l];
          lane_pick[l] = accept ? c + k : lane_pick[l];
#endif
          lane_random[q][hlf] = accept ? next[hlf] : lane_random[q][hlf];
        }
      }
#else
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
      const f2 ia = f2{inv_accept, inv_accept}, ir = f2{inv_reject, inv_reject}, p2 = f2{prob, prob};
#pragma unroll
      for (uint32_t q = 0; q < ((LUM_ABLATE_LIGHT & 4) ? 1u : kLightTreeOutputs / 2); q++) {
        const f2 if_accepted = lane_random[q] * ia, if_rejected = (lane_random[q] - p2) * ir;
#pragma unroll
        for (uint32_t hlf = 0; hlf < 2; hlf++) {
          const uint32_t l = 2 * q + hlf;
          const bool accept = lane_random[q][hlf] < prob;
          lane_target[l] = accept ? target : lane_target[l];  // (evaluating the picked child's importance again after the scan instead costs more in spills than this select)
          lane_pick[l] = accept ? c + k : lane_pick[l];
          lane_random[q][hlf] = clamp_random_top(accept ? if_accepted[hlf] : if_rejected[hlf]);
        }
      }
#endif
    }
  }
  TreeWork w;
  w.root_sum = sum * (bfloat_unpack(h.z) / 65535.0f);
#if LUM_ROOT_THRESHOLD && LUM_ROOT_KEY
#pragma unroll
  for (uint32_t l = 0; l < kLightTreeOutputs; l++) {  // (a lane that met no child keeps key 0: child 0, importance 0 - as the separate words would say)
    lane_target[l] = bitsf(lane_pick[l] & 0xFFFFFF80u);
    lane_pick[l] &= 0x7Fu;
  }
#endif
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
template <class Ctx, class Smp>
LUM_DEV TreePick tree_postpass(const DeviceScene& sc, const Ctx& g, const Smp& smp, uint32_t lane, const TreeWork& w) {
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
// The same triangle from the per-light table k_light_table writes at scene upload (four 16-byte words per light, one 64-byte line: vertex | material
// id and the bidirectional flag, edge1 | scene triangle, edge2 | area, emitted colour | whether it needs the textures): one round trip of four parallel
// loads instead of the chain handle -> mesh -> triangle offset -> vertices / transform / material word -> material, and neither the instance
// transform's arithmetic nor the area (a cross product and a square root) nor the material's decoding per candidate. The table holds what
// load_tri_light, tri_light_area and - for an emitter without emission or albedo texture - tri_light_color return in the exact flavour, bit for bit.
struct TableLight { TriLight tri; float area; Col color; bool textured; };
LUM_DEV TableLight load_tri_light_table(const DeviceScene& sc, uint32_t light_id) {
  const float4 a = sc.light_tri_table[4u * light_id], b = sc.light_tri_table[4u * light_id + 1u], c = sc.light_tri_table[4u * light_id + 2u],
               d = sc.light_tri_table[4u * light_id + 3u];
  TableLight e;
  TriLight& t = e.tri;
  t.vertex = v3(a.x, a.y, a.z); t.edge1 = v3(b.x, b.y, b.z); t.edge2 = v3(c.x, c.y, c.z);
  t.material_id = fbits(a.w) & 0xFFFFu;
  t.bidirectional = (fbits(a.w) >> 16) != 0u;
  t.scene_tri = fbits(b.w);
  e.area = c.w;
  e.color = col(d.x, d.y, d.z);
  e.textured = fbits(d.w) != 0u;
  return e;
}
// The first LUM_LDS_LIGHTS lights' table lines and handles, staged in LDS by every workgroup of k_shade (round 5): the candidate loop reads a line of 64 bytes
// and a handle per candidate through an index it has just computed - five dependent 16-byte gathers, eight times per vertex, from a table of a few KB that every
// lane of every wave reads. From LDS they cost the address unit nothing and return in a quarter of an L1 hit's time. Lights beyond the staged ones are read from
// memory as before (one branch per candidate); the values are the same words either way.
#ifndef LUM_LDS_LIGHTS
#define LUM_LDS_LIGHTS 256
#endif
typedef float LdsF4 __attribute__((ext_vector_type(4)));
struct StagedLights {
  const __attribute__((address_space(3))) LdsF4* table;
  const __attribute__((address_space(3))) unsigned long long* handles;
  uint32_t count;
};
LUM_DEV TableLight load_tri_light_table(const DeviceScene& sc, const StagedLights& sl, uint32_t light_id, uint2& handle) {
  float4 a, b, c, d;
  if (light_id < sl.count) {
    const LdsF4 va = sl.table[4u * light_id], vb = sl.table[4u * light_id + 1u], vc = sl.table[4u * light_id + 2u], vd = sl.table[4u * light_id + 3u];
    const unsigned long long h = sl.handles[light_id];
    a = make_float4(va.x, va.y, va.z, va.w); b = make_float4(vb.x, vb.y, vb.z, vb.w); c = make_float4(vc.x, vc.y, vc.z, vc.w); d = make_float4(vd.x, vd.y, vd.z, vd.w);
    handle = make_uint2((uint32_t) h, (uint32_t) (h >> 32));
  }
  else {
    a = sc.light_tri_table[4u * light_id]; b = sc.light_tri_table[4u * light_id + 1u]; c = sc.light_tri_table[4u * light_id + 2u]; d = sc.light_tri_table[4u * light_id + 3u];
    handle = sc.light_tri_handles[light_id];
  }
  TableLight e;
  TriLight& t = e.tri;
  t.vertex = v3(a.x, a.y, a.z); t.edge1 = v3(b.x, b.y, b.z); t.edge2 = v3(c.x, c.y, c.z);
  t.material_id = fbits(a.w) & 0xFFFFu;
  t.bidirectional = (fbits(a.w) >> 16) != 0u;
  t.scene_tri = fbits(b.w);
  e.area = c.w;
  e.color = col(d.x, d.y, d.z);
  e.textured = fbits(d.w) != 0u;
  return e;
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

template <class Smp>
LUM_DEV LightDirSample sample_light_direction(const LocalFrame& lf, const GeoContext& g, const Smp& smp) {
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
LUM_DEV float mis_for_light_sample(const GeoContext& g, V3 L, float area, Col color, float dist, float solid_angle, float root_sum) {
  const float power = importance(color) * area;
  return 1.0f - mis_base(light_direction_probability(g, L), solid_angle, power, dist * dist, root_sum);
}
#if LUM_FAST
// The fast flavour's form inside the candidate loop: the probability from the direction terms the BSDF evaluation has just formed in world space (the
// products N.V, N.H, H.V, H.L do not depend on the frame they are taken in) instead of a second analysis of the same pair of directions rotated
// into the shading frame - a rotation, a normalisation and a half vector less per candidate. `Vl` = the view direction in that frame (per vertex).
LUM_DEV float light_direction_probability_terms(const MatParams& p, V3 Vl, const RayTerms& c) {
  const bool with_refraction = (p.flags & kMatSubstrateMask) == kMatTranslucent;
  const float refr_prob = with_refraction ? 0.5f : 0.0f;
  const float roughness = p.roughness();
  const float sr = light_dir_roughness(roughness);
  float prob;
  if (c.is_refraction) prob = refr_prob * pdf_refraction(sr, c.NdotH, c.NdotV, c.HdotV, c.HdotL, p.ior());
  else prob = (1.0f - refr_prob) * pdf_vndf_bounded(Vl, sr, c.NdotH, c.NdotV);
  return prob * light_dir_rr(roughness);
}
#endif

// ---- light sampling (light.cuh:84-159) ----
struct LightSample { uint32_t light_id; V3 ray; Col color; float dist, root_sum; };

// Measurement switch LUM_DUP (bit mask; profiles/r06_ab_experiments.txt "what the parts of k_shade cost"): the named part of a vertex's work is done TWICE - the second time
// on inputs the compiler cannot see through, its results handed to an empty asm statement - so that the part's cost shows as the kernel's extra time while every result,
// every path and every other kernel stay what they are. (Leaving a part OUT - the LUM_ABLATE switches - also removes whatever only it kept alive, and changes the paths.)
//   1 the light tree's root pass   2 the candidate loop   4 the surface context   8 the bounce sample   16 the BSDF-driven light direction   32 the local frame
#ifndef LUM_DUP
#define LUM_DUP 0
#endif
LUM_DEV void dup_sink(float x) { asm volatile("" :: "v"(x)); }
LUM_DEV void dup_sink(uint32_t x) { asm volatile("" :: "v"(x)); }
LUM_DEV void dup_sink(V3 a) { asm volatile("" :: "v"(a.x), "v"(a.y), "v"(a.z)); }
LUM_DEV void dup_sink(Col a) { asm volatile("" :: "v"(a.r), "v"(a.g), "v"(a.b)); }
LUM_DEV GeoContext dup_launder(GeoContext g) { asm volatile("" : "+v"(g.position.x), "+v"(g.position.y), "+v"(g.position.z), "+v"(g.normal.x), "+v"(g.V.x)); return g; }

template <class Smp>
LUM_DEV LightSample light_candidates(const DeviceScene& sc, const GeoContext& g, const Smp& smp, ShadeClock& clock, const StagedLights& staged, const TreeWork& work,
                                     const Energy& energy);
template <class Smp>
LUM_DEV LightSample sample_light(const DeviceScene& sc, const GeoContext& g, const Smp& smp, ShadeClock& clock, const StagedLights& staged) {
  LUM_STAT(14, 15);
  const TreeWork work = tree_prepass(sc, g, smp);
  if (LUM_DUP & 1) {
    const TreeWork again = tree_prepass(sc, dup_launder(g), smp);
#pragma unroll
    for (uint32_t l = 0; l < kLightTreeOutputs; l++) dup_sink(again.cont[l]);
    dup_sink(again.root_sum);
  }
  const Energy energy = energy_terms(sc, g.params, world_ndotv(g));
  LUM_LAP(clock, 1);
  if (LUM_DUP & 2) {
    TreeWork w2 = work;
    asm volatile("" : "+v"(w2.cont[0]), "+v"(w2.root_sum));
    const LightSample again = light_candidates(sc, dup_launder(g), smp, clock, staged, w2, energy);
    dup_sink(again.light_id); dup_sink(again.ray); dup_sink(again.color); dup_sink(again.dist);
  }
  return light_candidates(sc, g, smp, clock, staged, work, energy);
}
template <class Smp>
LUM_DEV LightSample light_candidates(const DeviceScene& sc, const GeoContext& g, const Smp& smp, ShadeClock& clock, const StagedLights& staged, const TreeWork& work,
                                     const Energy& energy) {
  LightSample out;
  out.light_id = kLightIdInvalid; out.ray = v3(0.0f, 0.0f, 0.0f); out.color = splat(0.0f); out.dist = 0.0f;
  Reservoir rv;
  rv.random = smp.next1(kRndLightGeoResampling);
  rv.reset();
#if LUM_FAST
  const V3 view_local = normalize(qapply(rotation_to_z(g.normal), g.V));  // light_direction_probability's Vl
#endif
#ifndef LUM_ABLATE_LANES
#define LUM_ABLATE_LANES kLightTreeOutputs  // measurement only: fewer resampling lanes evaluated (results are wrong)
#endif
  // Experiment (LUM_PREFETCH_RANDOM): a candidate's random pair is a table word and a blue-noise texel - two gathers the iteration waits for right at its top. With the
  // switch the pair of candidate l + 1 is requested while candidate l is worked on (the raw integers: two registers live across the loop body).
#ifndef LUM_PREFETCH_RANDOM
#define LUM_PREFETCH_RANDOM 0
#endif
#if LUM_PREFETCH_RANDOM
  U2 next_pair = smp.raw2(kRndLightGeoRay);
#endif
#pragma nounroll
  for (uint32_t lane = 0; lane < LUM_ABLATE_LANES; lane++) {
    LUM_STAT(8, 9);
#if LUM_PREFETCH_RANDOM
    const U2 this_pair = next_pair;
    next_pair = smp.raw2(kRndLightGeoRay + min(lane + 1u, (uint32_t) kLightTreeOutputs - 1u));
#endif
    const TreePick pick = tree_postpass(sc, g, smp, lane, work);
    if (pick.light_id == kLightIdInvalid) continue;
#if LUM_LDS_LIGHTS
    uint2 handle;
    const TableLight entry = load_tri_light_table(sc, staged, pick.light_id, handle);
#else
    const uint2 handle = sc.light_tri_handles[pick.light_id];
    const TableLight entry = load_tri_light_table(sc, pick.light_id);
#endif
    const TriLight& tl = entry.tri;
    if (handle.x == g.instance_id && handle.y == g.tri_id) continue;
    V3 ray; float sa;
#if LUM_PREFETCH_RANDOM
    const F2 pair = F2{unit_float(this_pair.x), unit_float(this_pair.y)};
#else
    const F2 pair = smp.next2(kRndLightGeoRay + lane);
#endif
    if (!sample_tri_solid_angle(g.position, tl, pair, ray, sa)) continue;
    F2 uv;
    const float dist = intersect_triangle(tl.vertex, tl.edge1, tl.edge2, g.position, ray, uv);
    if (dist == kFltMax) continue;
    LUM_STAT(10, 11);
    LUM_LAP(clock, 10);
    Col lc = entry.textured ? tri_light_color(sc, tl, uv) : entry.color;
    bool is_refraction;
#ifndef LUM_ABLATE_LIGHT_DEFINED_BELOW
#define LUM_ABLATE_LIGHT_DEFINED_BELOW
#endif
#if 0
#define LUM_ABLATE_LIGHT 0  // measurement only: 1 no BSDF evaluation, 2 no MIS weight, 4 one resampling lane in the root pass (results are wrong)
#endif
#if LUM_FAST
    const RayTerms terms = analyze_direction(g.params, g.normal, g.V, ray);
    is_refraction = terms.is_refraction;
    const Col bw = (LUM_ABLATE_LIGHT & 1) ? splat(0.5f) : eval_with_face_normal(energy, g.params, terms, kHintGeneral, ray, normal_unpack(g.face_normal_packed), 1.0f);
    LUM_LAP(clock, 11);
    const float mis = (LUM_ABLATE_LIGHT & 2) ? 0.5f : 1.0f - mis_base(light_direction_probability_terms(g.params, view_local, terms), sa, importance(lc) * entry.area, dist * dist, work.root_sum);
#else
    const Col bw = (LUM_ABLATE_LIGHT & 1) ? splat(0.5f) : eval_bsdf(energy, g, ray, kHintGeneral, is_refraction, 1.0f);
    LUM_LAP(clock, 11);
    const float mis = (LUM_ABLATE_LIGHT & 2) ? 0.5f : mis_for_light_sample(g, ray, entry.area, lc, dist, sa, work.root_sum);
#endif
    lc = (lc * bw) * mis;
    if (rv.add(importance(lc), pick.weight * sa)) { out.light_id = pick.light_id; out.ray = ray; out.color = lc; out.dist = dist; }
    LUM_LAP(clock, 12);
  }
  LUM_LAP(clock, 2);
  out.color = out.color * rv.sampling_weight();
  out.root_sum = work.root_sum;
  return out;
}

LUM_NS_END
