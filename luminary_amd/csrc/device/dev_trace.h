// Software ray queries on the two-level BVH4 that replaces the reference's OptiX acceleration structures.
// Semantics restated from the reference's OptiX programs:
//   closest hit  optix/optix_kernel_raytrace.cu:82-95, cuda/optix_anyhit.cuh:15-31, cuda/optix_closesthit.cuh:15-26
//   shadow       cuda/optix_common.cuh:76-106, cuda/optix_anyhit.cuh:49-139, cuda/optix_closesthit.cuh:44-58
//   light query  cuda/optix_anyhit.cuh:145-205 (reservoir over all lights hit), cuda/direct_lighting.cuh:596-611
// Triangle test: cuda/math.cuh:1337-1358 on the object-space ray (instance transform world = S*R*v + T, math.cuh:459-489),
// so hit distances are the same numbers in world and object space.
// Results do not depend on traversal order: ties are broken by (t, instance, triangle) and the light pick is a function of
// the candidate set only (see light_query).
//
// Execution model (gfx950; the counters quoted here are rounds 4-5's, profiles/r05_ab_experiments.txt; the current ones: profiles/pmc_counters.json and DESIGN.md section 0): one ray per lane,
// persistent waves, one workgroup of 1024 threads = 16 waves per CU = 4 waves per SIMD in BOTH flavours since round 4 - 114 / 110 VGPRs and nothing spilled since round 6's
// two-triangle leaves (LUM_LEAF_MAX, dev_scene.h: the leaf registers are 3 x float4 per slot; with four slots k_trace sat at 128 + 12 spilled). On the hall a closest-hit
// launch runs at a VALU lane utilisation of 0.45 (visibility: 0.53), its waves wait 0.56 (0.61) of their cycles, and the vector-memory address unit is busy 0.71
// of the launch: a divergent 16-byte lane load costs it one cycle per LANE (7 per node visit that misses the staged top, 3 per triangle), whatever the number of
// wave instructions that carry them. The kernels sit between the two: every variant that removed lane loads added instructions and became issue-bound (64-byte
// and 8-wide quantised nodes), every variant that removed instructions per ray or filled the waves left the lane loads in place and gained nothing - round 5's
// LDS phase queues (dev_trace_pool.h, LUM_PHASE_QUEUES) raise the lane utilisation to 0.68 / 0.76 with a third fewer vector-memory instructions at the SAME number
// of vector instructions and run 25-80 % slower, because twice the rays per CU halve each ray's share of LDS (stack bottoms, staged nodes). Memory bandwidth is
// not the limit (0.52 / 0.38 of 8 TB/s memory-side), latency as such neither (an L1->L2 read returns after ~400 cycles, a tenth of a wave iteration).
// What an iteration costs:
//   * (a leaf = at most two triangles, 3 x 16-byte loads each; trees: the builders' binary SAH tree cut into 4-wide nodes by a dynamic programme, bvh_build.cpp CollapsePlan)
//   * one node visit = 7 x 16-byte loads (near/far planes picked by the ray's direction signs, so no per-axis min/max),
//     24 fma, v_max3/v_min3, a 5-comparator sorting network on (entry distance, child) pairs and conditional pushes: ~110 vector instructions;
//   * every wave iteration runs ONE phase - node visit, instance entry or triangle tests - chosen by a vote over its lanes, so a
//     lane that holds a leaf does not wait for the slowest lane of the wave to find one (plain while-while: 0.33-0.41 lane occupancy);
//   * persistent waves fetch rays from a global cursor and refill idle lanes when too few are still traversing;
//   * the first nodes of the array (breadth-first across both levels; 448 of them beside 96 KB of stack bottoms since round 6, 704 beside 64 KB before) are staged in LDS by
//     every workgroup, the oldest 12 (visibility rays: 24) stack entries of every lane live there too: LDS hit rate of the node visits 0.62 / 0.72 with the 704.
// Structural alternatives that were built and measured, and lost (profiles/r0*_ab_experiments.txt): 8-wide quantised nodes with a sorting network (twice),
// 64-byte quantised 4-wide nodes, dual-node visits, speculative traversal past a leaf (LUM_SPECULATE, round 4: +7 % node visits, +16-19 % time),
// physical ray reordering between bounces, per-XCD work ranges, LDS-DMA prefetch, rays regrouped by phase through LDS (round 5, above), an MFMA slab test for packets
// (tools/microbench/mfma_slab.hip: 0.62 x the vector path). The 8-wide octant-order node (no sort, one group entry per visit) was
// prototyped as a visit routine and a CPU walk before a rewrite (tools/microbench/node_visit.hip, tools/bvh_quality.cpp BQ_WIDE): see DESIGN.md section 4.
#pragma once

#include "dev_light.h"

LUM_NS_BEGIN

#ifndef LUM_TRACE_BLOCK
#define LUM_TRACE_BLOCK 1024  // threads per workgroup of the persistent ray kernels = one workgroup per CU at 4 waves per SIMD (128 VGPRs): one LDS copy of the tree
                              // top per CU and room for the lanes' traversal stacks (256 x 3 copies measured 1-2 % slower, 512 13 % slower). Fast flavour against 768
                              // threads / 3 waves: visibility kernel -16 %, closest-hit kernel -9 % (round 2); the exact flavour, which kept 768 until round 4:
                              // -12 % / -14 % (140 -> 128 VGPRs, 13 spilled), +6.5 % samples/s
#endif
constexpr int kTraceBlock = LUM_TRACE_BLOCK;

constexpr uint32_t kHitSky        = 0xFFFFFFFEu;  // cuda/utils.cuh:50-64
constexpr uint32_t kLeaveInstance = 0xFFFFFFFDu;  // stack marker: back from a bottom-level BVH to the top level
constexpr uint32_t kTraversalDone = 0xFFFFFFFCu;
constexpr uint32_t kNoInstance    = 0xFFFFFFFFu;
// Stack bound: the host builder caps the top level at 16 and every bottom level at 26 BVH4 levels (core.hip); a level pushes at
// most 3 entries and entering an instance pushes one marker: 3*16 + 1 + 3*26 = 127.
constexpr int kStackSize = 128;
#ifndef LUM_CHUNK_MAX
#define LUM_CHUNK_MAX 256u  // most items a wave reserves per atomic
#endif
#ifndef LUM_VOTE_TRIS
#define LUM_VOTE_TRIS 1u   // the triangle phase runs when lanes-with-triangles * LUM_VOTE_TRIS >= the larger other group * LUM_VOTE_NODES
#define LUM_VOTE_NODES 2u  // (1:1 .. 1:4 and refill thresholds 32 .. 56 measured within 2 % of each other)
#endif
#ifndef LUM_REFILL
#define LUM_REFILL 40  // persistent waves refill their idle lanes when fewer than this many lanes are still traversing
#endif

#ifndef LUM_LDS_INSTANCES
#define LUM_LDS_INSTANCES 64  // top-level leaf records (64 bytes each) a ray workgroup keeps in LDS next to the staged tree top
#endif
#ifndef LUM_DEFER_FINISH
#define LUM_DEFER_FINISH 0  // 1: a finished ray's result is written when its lane takes the next ray (or at the end), not inside the phase loop (see trace_items; measured neutral)
#endif
#ifndef LUM_LDS_TURN
#define LUM_LDS_TURN 0   // lanes on staged nodes get wave iterations of their own while at least this many of them exist (0: off); see the phase vote
#endif
#ifndef LUM_LDS_FIRST
#define LUM_LDS_FIRST 1  // such iterations come before triangle and instance-entry phases (0: only where the node phase would have run anyway)
#endif

// Speculative traversal (after Aila, Laine: "Understanding the Efficiency of Ray Traversal on GPUs", HPG 2009): a lane that reaches a triangle leaf while
// the wave goes on visiting nodes does not sit the node phases out. It sets the leaf aside (`postponed`, one register) and takes on the newest stack
// entry when that is an inner node it may still have to visit; the leaf is tested when the wave's triangle phase comes. Results do not depend on the
// order in which leaves and nodes are met (closest hits are a minimum, visibility a product / any blocker); what it can cost is visits that the
// postponed leaf's hit would have culled. 1: closest-hit rays, 2: visibility rays, 3: both.
#ifndef LUM_SPECULATE
#define LUM_SPECULATE 0
#endif
#ifndef LUM_DUAL_VISIT
#define LUM_DUAL_VISIT 0  // experiment, measured negative (visibility kernel +18 % on the hall): see visit_two_nodes
#endif
struct RayStats { uint32_t nodes, tris, lds_nodes; };
constexpr uint32_t kPrefetchSinkWords = 64u * 16u;  // one dword per lane for up to 16 waves of a ray workgroup (LUM_PREFETCH)

// Diagnostic build (-DLUM_PHASE_STATS): wave-level iteration counts of the traversal phases, to see where lanes idle.
//   0 node-phase iterations   1 instance-entry iterations   2 lanes entering   3 triangle-phase iterations   4 lanes in them
//   5 outer iterations (refill checks)   6 pop iterations   7 lanes popping
#ifdef LUM_PHASE_STATS
#define LUM_PHASE(k) do { const unsigned long long act_ = __ballot(true); if ((threadIdx.x & 63u) == (uint32_t) __builtin_ctzll(act_)) phase_[k]++; } while (0)
#define LUM_PHASE_LANES(k) do { phase_[k]++; } while (0)
#else
#define LUM_PHASE(k) do {} while (0)
#define LUM_PHASE_LANES(k) do {} while (0)
#endif
#ifdef LUM_PHASE_STATS
#define LUM_TIME_BEGIN() const unsigned long long t_begin_ = __builtin_readcyclecounter()
#define LUM_TIME_END(k) do { ptime_[k] += __builtin_readcyclecounter() - t_begin_; ptime_[(k) + 1]++; } while (0)
#else
#define LUM_TIME_BEGIN() do {} while (0)
#define LUM_TIME_END(k) do {} while (0)
#endif  // lds_nodes: node visits served from the LDS-staged top of the tree

// Reciprocal for the box test only (v_rcp_f32, 1 ulp): like the min/max below it decides what gets visited, never a result; the
// padded boxes and the relaxed comparison absorb its error. A correctly rounded division costs a dozen instructions, three times
// per ray and per instance entered.
LUM_DEV float safe_inv(float d) { return (fabsf(d) < 1e-30f) ? copysignf(1e30f, d) : __builtin_amdgcn_rcpf(d); }

// Raw three-operand min/max: the box test only decides what gets visited, never a result, so it is outside the IEEE-only contract
// and does not need the NaN canonicalisation the compiler adds around fminf/fmaxf.
LUM_DEV float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
LUM_DEV float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
LUM_DEV float vmax0(float a) { float r; asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(a)); return r; }
LUM_DEV float vmin2(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// Ray in the space of the BVH being walked, with the byte offsets of its near/far planes inside a node.
struct TRay {
  V3 o, d, inv, noi;  // noi = -(o * inv)
  uint32_t nx, ny, nz, fx, fy, fz;
  uint32_t oct;       // direction signs, bit a set: inv_a < 0 (the octant-slot nodes of LUM_BVH8O are walked in the order slot ^ oct)
  LUM_DEV void set(V3 origin, V3 dir) {
    o = origin; d = dir;
    inv = v3(safe_inv(dir.x), safe_inv(dir.y), safe_inv(dir.z));
    noi = v3(-(origin.x * inv.x), -(origin.y * inv.y), -(origin.z * inv.z));
    oct = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
    nx = (inv.x < 0.0f) ? 48u : 0u;  fx = 48u - nx;    // lo_x at 0, hi_x at 48
    ny = (inv.y < 0.0f) ? 64u : 16u; fy = 80u - ny;    // lo_y at 16, hi_y at 64
    nz = (inv.z < 0.0f) ? 80u : 32u; fz = 112u - nz;   // lo_z at 32, hi_z at 80
  }
  // the same for an 8-wide quantised node (Bvh8Node, dev_scene.h): lo_x at 48, lo_y 56, lo_z 64, hi_x 72, hi_y 80, hi_z 88
  LUM_DEV uint32_t nx8() const { return (inv.x < 0.0f) ? 72u : 48u; }
  LUM_DEV uint32_t ny8() const { return (inv.y < 0.0f) ? 80u : 56u; }
  LUM_DEV uint32_t nz8() const { return (inv.z < 0.0f) ? 88u : 64u; }
};

LUM_DEV float4 node_f4(const Bvh4Node* nodes, uint32_t byte_offset) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(nodes) + byte_offset);
}
LUM_DEV uint4 node_u4(const Bvh4Node* nodes, uint32_t byte_offset) {
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(nodes) + byte_offset);
}

// Entry distance of one child box, +inf when the segment [0, tmax] misses it. Empty children carry inverted boxes and fail the
// test on their own. Boxes are padded by the builder and the comparison is relaxed, so a triangle accepted by the exact test is
// never culled by rounding here.
// kFarFirst: the key is the negated entry distance, so that the same sorting network orders the children farthest first (misses stay +inf, last).
template <int kFarFirst = 0>
LUM_DEV float child_entry(float nx, float ny, float nz, float fx, float fy, float fz, const TRay& r, float tmax) {
  const float ax = __builtin_fmaf(nx, r.inv.x, r.noi.x), ay = __builtin_fmaf(ny, r.inv.y, r.noi.y), az = __builtin_fmaf(nz, r.inv.z, r.noi.z);
  const float bx = __builtin_fmaf(fx, r.inv.x, r.noi.x), by = __builtin_fmaf(fy, r.inv.y, r.noi.y), bz = __builtin_fmaf(fz, r.inv.z, r.noi.z);
  const float tn = vmax3(ax, ay, vmax0(az));
  const float tf = vmin3(bx, by, vmin2(bz, tmax));
  const bool far_first = kFarFirst == 1 || (kFarFirst == 2 && tmax >= 3.0e38f) || (kFarFirst == 3 && tmax < 3.0e38f);  // 2 / 3: measurement only
  return (tn <= __builtin_fmaf(tf, 1.000004f, 1e-30f)) ? (far_first ? -tn : tn) : __builtin_inff();
}

LUM_DEV void cswap(float& ka, uint32_t& ca, float& kb, uint32_t& cb) {
  const bool s = kb < ka;
  const float k0 = s ? kb : ka, k1 = s ? ka : kb;
  const uint32_t c0 = s ? cb : ca, c1 = s ? ca : cb;
  ka = k0; kb = k1; ca = c0; cb = c1;
}

LUM_DEV bool within(float tnear, float tmax) { return tnear <= __builtin_fmaf(tmax, 1.000004f, 1e-30f); }

// Visits inner node `cur`: returns the nearest child the ray may touch after pushing the others far-to-near, or kBvhEmpty when
// the ray misses all four (the caller pops).
// The first `lds_count` nodes of the array (the top of the tree in breadth-first order, core.hip) are staged in LDS by every
// workgroup of the persistent ray kernels: a divergent 16-byte LDS read costs a fraction of a divergent L1 access.
struct NodeSource { const Bvh4Node* global; const char* lds; uint32_t lds_count; };

// Where the sixteen-byte word `i` of the node array sits in the staged copy. A lane reads the same word (say, the near x planes) of whatever node it
// stands on; unswizzled, that word of every even node lies on the same four LDS banks and of every odd node on four others, so a wave's read of
// 64 different nodes queued up on eight banks (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.5-0.6). XOR-ing the word's position inside its node with
// bits 1-3 of the node index spreads the same word of sixteen consecutive nodes over all 64 banks. Float-box nodes only (LUM_LDS_SWIZZLE).
// Measured (profiles/r03_ab_experiments.txt, same box): hall 221.1 / 373.6 ms per 3 steps of the closest-hit / visibility kernel without, 222.3 / 374.0 with
// it; scan 65.1 / 70.7 against 65.9 / 70.5 - the LDS reads are not on the kernels' critical path. Off: it costs nine instructions per staged visit.
#ifndef LUM_LDS_SWIZZLE
#define LUM_LDS_SWIZZLE 0
#endif
LUM_DEV uint32_t lds_slot_swizzle(uint32_t word_index) {
#if LUM_LDS_SWIZZLE
  return word_index ^ ((word_index >> 4) & 7u);  // word_index = node * 8 + slot: bits 1-3 of the node index are bits 4-6 here
#else
  return word_index;
#endif
}
LUM_DEV uint32_t lds_node_swizzle_bytes(uint32_t node) {
#if LUM_LDS_SWIZZLE
  return ((node >> 1) & 7u) << 4;
#else
  (void) node;
  return 0u;
#endif
}

// Traversal stack entry. Closest-hit rays keep the child's entry distance next to its index so that a pop can drop what lies beyond the hit
// found meanwhile (8 bytes). A visibility ray's segment never shrinks, every stacked child stays within reach, so its entries are the
// index alone (4 bytes): half the scratch traffic of the kernel that writes most of it (rocprofv3 WRITE_SIZE, C3: 4.9 GB per launch of which
// 0.33 GB are results; the stacks of the 196 k resident lanes do not fit the 4 MB L2 of an XCD next to the nodes).
template <bool kCull> struct StackEntry;
template <> struct StackEntry<true> {
  using E = uint2;
  static LUM_DEV E make(uint32_t node, float tnear) { return make_uint2(node, fbits(tnear)); }
  static LUM_DEV uint32_t node(E e) { return e.x; }
  static LUM_DEV bool reachable(E e, float tmax) { return bitsf(e.y) <= __builtin_fmaf(tmax, 1.000004f, 1e-30f); }
};
template <> struct StackEntry<false> {
  using E = uint32_t;
  static LUM_DEV E make(uint32_t node, float) { return node; }
  static LUM_DEV uint32_t node(E e) { return e; }
  static LUM_DEV bool reachable(E, float) { return true; }
};

// A lane's traversal stack. Its oldest entries live in the workgroup's LDS behind the staged tree top (LUM_LDS_STACK_BYTES of it, entry i of
// thread t at word i * blockDim + t: a lane only ever touches its own bank), the rest in scratch: the stacks of the resident lanes do not fit
// L2 next to the nodes, and with everything in scratch they are a tenth to a fifth of a ray kernel's memory-side traffic.
// how an entry is kept in memory: a plain machine word (the HIP vector classes have no assignment through an address-space-qualified pointer)
template <typename E> struct StackWord;
template <> struct StackWord<uint32_t> {
  typedef uint32_t W;
  static LUM_DEV W pack(uint32_t e) { return e; }
  static LUM_DEV uint32_t unpack(W w) { return w; }
};
template <> struct StackWord<uint2> {
  typedef unsigned long long W;
  static LUM_DEV W pack(uint2 e) { return (W) e.x | ((W) e.y << 32); }
  static LUM_DEV uint2 unpack(W w) { return make_uint2((uint32_t) w, (uint32_t) (w >> 32)); }
};
template <typename E> struct TraversalStack {
  // Typed address spaces on purpose: with generic pointers the compiler merges the two branches of load() into one flat_load through a selected
  // pointer - every pop then went through the vector-memory address unit, whether its entry sat in LDS or not, and (flat loads count on both
  // counters) the wave waited for it at once. The LDS entries are kTraceBlock entries apart (a compile-time shift or multiply, not blockDim.x).
  typedef typename StackWord<E>::W W;
  typedef __attribute__((address_space(5))) W* ScratchPtr;
  typedef __attribute__((address_space(3))) W* LdsPtr;
  ScratchPtr scratch; LdsPtr lds; int lds_entries;
  LUM_DEV void store(int i, E e) { if (i < lds_entries) lds[(uint32_t) i * (uint32_t) kTraceBlock] = StackWord<E>::pack(e); else scratch[i] = StackWord<E>::pack(e); }
  LUM_DEV E load(int i) const { if (i < lds_entries) return StackWord<E>::unpack(lds[(uint32_t) i * (uint32_t) kTraceBlock]); return StackWord<E>::unpack(scratch[i]); }
};
template <bool kOrdered, bool kCull, int kFarFirst = 0, typename S>
LUM_DEV uint32_t visit_node(const NodeSource& src, uint32_t cur, const TRay& r, float tmax, S& stk, int& sp,
                            typename StackEntry<kCull>::E& top, RayStats& st) {
  static_assert(!(kFarFirst && kCull), "the negated keys are not distances: only for queries that do not cull by them");
  using SE = StackEntry<kCull>;
  const uint32_t b = cur << 7;
  float4 nx, ny, nz, fx, fy, fz;
  uint4 ch;
  if (cur < src.lds_count) {
    const char* p = src.lds + b;
    const uint32_t z = lds_node_swizzle_bytes(cur);
    nx = *reinterpret_cast<const float4*>(p + (r.nx ^ z)); ny = *reinterpret_cast<const float4*>(p + (r.ny ^ z)); nz = *reinterpret_cast<const float4*>(p + (r.nz ^ z));
    fx = *reinterpret_cast<const float4*>(p + (r.fx ^ z)); fy = *reinterpret_cast<const float4*>(p + (r.fy ^ z)); fz = *reinterpret_cast<const float4*>(p + (r.fz ^ z));
    ch = *reinterpret_cast<const uint4*>(p + (96u ^ z));
    st.lds_nodes++;
  }
  else {
    const Bvh4Node* __restrict__ nodes = src.global;
    nx = node_f4(nodes, b + r.nx); ny = node_f4(nodes, b + r.ny); nz = node_f4(nodes, b + r.nz);
    fx = node_f4(nodes, b + r.fx); fy = node_f4(nodes, b + r.fy); fz = node_f4(nodes, b + r.fz);
    ch = node_u4(nodes, b + 96u);
  }
  float k0 = child_entry<kFarFirst>(nx.x, ny.x, nz.x, fx.x, fy.x, fz.x, r, tmax);
  float k1 = child_entry<kFarFirst>(nx.y, ny.y, nz.y, fx.y, fy.y, fz.y, r, tmax);
  float k2 = child_entry<kFarFirst>(nx.z, ny.z, nz.z, fx.z, fy.z, fz.z, r, tmax);
  float k3 = child_entry<kFarFirst>(nx.w, ny.w, nz.w, fx.w, fy.w, fz.w, r, tmax);
  uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
  if (kOrdered) {  // nearest first; visibility rays visit everything on the segment anyway, any order does
    cswap(k0, c0, k1, c1); cswap(k2, c2, k3, c3); cswap(k0, c0, k2, c2); cswap(k1, c1, k3, c3); cswap(k1, c1, k2, c2);
  }
  const float inf = __builtin_inff();
#ifndef LUM_PUSH_COND
#define LUM_PUSH_COND 1  // measured (hall / scan / example, fast flavour): closest-hit kernel -15 / -13 / -7 %, visibility kernel -4 / -4 / -3 %
#endif
#if LUM_PUSH_COND
  // Pushes only what is real. After the sort the children the ray may touch come first, so "child j is real" implies the same of every
  // child before it: one branch skips all three pushes in the common case of at most one child hit, and nothing is written for children the
  // ray misses (branch-free pushes store 3 entries per visit, used or not: most of the kernel's L2 requests were those stores).
  if (k1 < inf) {
    if (k2 < inf) {
      if (k3 < inf) { stk.store(sp, top); sp++; top = SE::make(c3, k3); }
      stk.store(sp, top); sp++; top = SE::make(c2, k2);
    }
    stk.store(sp, top); sp++; top = SE::make(c1, k1);
  }
#else
  // Branch-free pushes. The newest entry lives in registers (`top`), older ones in scratch: a push spills the old top to a slot
  // that is only kept if the push is real, so a pop never waits for a scratch load before it can fetch the next node.
  {
    const bool v = k3 < inf;
    stk.store(sp, top); sp += v ? 1 : 0;
    top = v ? SE::make(c3, k3) : top;
  }
  {
    const bool v = k2 < inf;
    stk.store(sp, top); sp += v ? 1 : 0;
    top = v ? SE::make(c2, k2) : top;
  }
  {
    const bool v = k1 < inf;
    stk.store(sp, top); sp += v ? 1 : 0;
    top = v ? SE::make(c1, k1) : top;
  }
#endif
  return (k0 < inf) ? c0 : kBvhEmpty;
}

// ---- 8-wide nodes with quantised child boxes (Bvh8Node): the scene's and the particles' trees ----
// Eight children per 128-byte line instead of four: 22 % fewer node visits on the hall, the boxes 8-bit offsets from the node's corner in units of a
// power of two per axis, rounded outwards by the builder (after Ylitie, Karras, Laine 2017, without their octant ordering: the eight entry
// distances are sorted by a 19-comparator network). Measured twice (rounds 1 and 2): a visit costs 215 instead of 110 vector instructions - 48
// byte->float conversions and the network's 95 - and the kernels are issue-bound (section 4): closest-hit +9 %, visibility +10 % time. Off.
LUM_DEV float byte_f(uint32_t w, uint32_t k) { return (float) ((w >> (8u * k)) & 0xFFu); }  // v_cvt_f32_ubyte{k}
template <bool kOrdered, bool kCull, typename S>
LUM_DEV uint32_t visit_node8(const NodeSource& src, uint32_t cur, const TRay& r, float tmax, S& stk, int& sp,
                             typename StackEntry<kCull>::E& top, RayStats& st) {
  using SE = StackEntry<kCull>;
  const uint32_t b = cur << 7;
  const uint32_t onx = r.nx8(), ony = r.ny8(), onz = r.nz8();
  float4 head;
  uint4 ca, cb;
  uint2 qnx, qny, qnz, qfx, qfy, qfz;
  if (cur < src.lds_count) {
    const char* p = src.lds + b;
    head = *reinterpret_cast<const float4*>(p); ca = *reinterpret_cast<const uint4*>(p + 16u); cb = *reinterpret_cast<const uint4*>(p + 32u);
    qnx = *reinterpret_cast<const uint2*>(p + onx); qny = *reinterpret_cast<const uint2*>(p + ony); qnz = *reinterpret_cast<const uint2*>(p + onz);
    qfx = *reinterpret_cast<const uint2*>(p + (120u - onx)); qfy = *reinterpret_cast<const uint2*>(p + (136u - ony)); qfz = *reinterpret_cast<const uint2*>(p + (152u - onz));
    st.lds_nodes++;
  }
  else {
    const char* __restrict__ p = reinterpret_cast<const char*>(src.global) + b;
    head = *reinterpret_cast<const float4*>(p); ca = *reinterpret_cast<const uint4*>(p + 16u); cb = *reinterpret_cast<const uint4*>(p + 32u);
    qnx = *reinterpret_cast<const uint2*>(p + onx); qny = *reinterpret_cast<const uint2*>(p + ony); qnz = *reinterpret_cast<const uint2*>(p + onz);
    qfx = *reinterpret_cast<const uint2*>(p + (120u - onx)); qfy = *reinterpret_cast<const uint2*>(p + (136u - ony)); qfz = *reinterpret_cast<const uint2*>(p + (152u - onz));
  }
  // plane distance = (origin + q * scale - o) * inv = q * (scale * inv) + (origin * inv + noi)
  const uint32_t ew = fbits(head.w);
  const float sx = bitsf((ew & 0xFFu) << 23) * r.inv.x, sy = bitsf(((ew >> 8) & 0xFFu) << 23) * r.inv.y, sz = bitsf(((ew >> 16) & 0xFFu) << 23) * r.inv.z;
  const float bx = __builtin_fmaf(head.x, r.inv.x, r.noi.x), by = __builtin_fmaf(head.y, r.inv.y, r.noi.y), bz = __builtin_fmaf(head.z, r.inv.z, r.noi.z);
  const float inf = __builtin_inff();
  const float lim = vmin2(tmax, inf);
  float k[8];
  uint32_t c[8] = {ca.x, ca.y, ca.z, ca.w, cb.x, cb.y, cb.z, cb.w};
#pragma unroll
  for (uint32_t j = 0; j < 8; j++) {
    const uint32_t wnx = j < 4 ? qnx.x : qnx.y, wny = j < 4 ? qny.x : qny.y, wnz = j < 4 ? qnz.x : qnz.y;
    const uint32_t wfx = j < 4 ? qfx.x : qfx.y, wfy = j < 4 ? qfy.x : qfy.y, wfz = j < 4 ? qfz.x : qfz.y;
    const float ax = __builtin_fmaf(byte_f(wnx, j & 3u), sx, bx), ay = __builtin_fmaf(byte_f(wny, j & 3u), sy, by), az = __builtin_fmaf(byte_f(wnz, j & 3u), sz, bz);
    const float fx = __builtin_fmaf(byte_f(wfx, j & 3u), sx, bx), fy = __builtin_fmaf(byte_f(wfy, j & 3u), sy, by), fz = __builtin_fmaf(byte_f(wfz, j & 3u), sz, bz);
    const float tn = vmax3(ax, ay, vmax0(az));
    const float tf = vmin3(fx, fy, vmin2(fz, lim));
    k[j] = (c[j] != kBvhEmpty && tn <= __builtin_fmaf(tf, 1.000004f, 1e-30f)) ? tn : inf;
  }
  if (kOrdered) {  // Batcher's odd-even merge sort, 19 comparators
    cswap(k[0], c[0], k[1], c[1]); cswap(k[2], c[2], k[3], c[3]); cswap(k[4], c[4], k[5], c[5]); cswap(k[6], c[6], k[7], c[7]);
    cswap(k[0], c[0], k[2], c[2]); cswap(k[1], c[1], k[3], c[3]); cswap(k[4], c[4], k[6], c[6]); cswap(k[5], c[5], k[7], c[7]);
    cswap(k[1], c[1], k[2], c[2]); cswap(k[5], c[5], k[6], c[6]);
    cswap(k[0], c[0], k[4], c[4]); cswap(k[1], c[1], k[5], c[5]); cswap(k[2], c[2], k[6], c[6]); cswap(k[3], c[3], k[7], c[7]);
    cswap(k[2], c[2], k[4], c[4]); cswap(k[3], c[3], k[5], c[5]);
    cswap(k[1], c[1], k[2], c[2]); cswap(k[3], c[3], k[4], c[4]); cswap(k[5], c[5], k[6], c[6]);
    // pushes only what is real, far to near; sorted, so "child j is real" implies the same of every child before it
    if (k[1] < inf) {
      if (k[2] < inf) {
        if (k[3] < inf) {
          if (k[4] < inf) {
            if (k[5] < inf) {
              if (k[6] < inf) {
                if (k[7] < inf) { stk.store(sp, top); sp++; top = SE::make(c[7], k[7]); }
                stk.store(sp, top); sp++; top = SE::make(c[6], k[6]);
              }
              stk.store(sp, top); sp++; top = SE::make(c[5], k[5]);
            }
            stk.store(sp, top); sp++; top = SE::make(c[4], k[4]);
          }
          stk.store(sp, top); sp++; top = SE::make(c[3], k[3]);
        }
        stk.store(sp, top); sp++; top = SE::make(c[2], k[2]);
      }
      stk.store(sp, top); sp++; top = SE::make(c[1], k[1]);
    }
    return (k[0] < inf) ? c[0] : kBvhEmpty;
  }
  // unordered: continue with the first real child, push the others
  uint32_t next = kBvhEmpty;
#pragma unroll
  for (uint32_t j = 0; j < 8; j++) {
    if (k[j] < inf) {
      if (next == kBvhEmpty) next = c[j];
      else { stk.store(sp, top); sp++; top = SE::make(c[j], k[j]); }
    }
  }
  return next;
}

// ---- 4-wide nodes with quantised child boxes in 64 bytes (Bvh4QNode) ----
template <bool kOrdered, bool kCull, typename S>
LUM_DEV uint32_t visit_node_q(const NodeSource& src, uint32_t cur, const TRay& r, float tmax, S& stk, int& sp,
                              typename StackEntry<kCull>::E& top, RayStats& st) {
  using SE = StackEntry<kCull>;
  const uint32_t b = cur << 6;
  float4 head;
  uint4 ch, q0;
  uint2 q1;
  if (cur < src.lds_count) {
    const char* p = src.lds + b;
    head = *reinterpret_cast<const float4*>(p); ch = *reinterpret_cast<const uint4*>(p + 16u); q0 = *reinterpret_cast<const uint4*>(p + 32u); q1 = *reinterpret_cast<const uint2*>(p + 48u);
    st.lds_nodes++;
  }
  else {
    const char* __restrict__ p = reinterpret_cast<const char*>(src.global) + b;
    head = *reinterpret_cast<const float4*>(p); ch = *reinterpret_cast<const uint4*>(p + 16u); q0 = *reinterpret_cast<const uint4*>(p + 32u); q1 = *reinterpret_cast<const uint2*>(p + 48u);
  }
  // q0 = lo_x, lo_y, lo_z, hi_x; q1 = hi_y, hi_z; near plane of an axis = the lower one unless the ray runs against it
  const bool bx_neg = r.inv.x < 0.0f, by_neg = r.inv.y < 0.0f, bz_neg = r.inv.z < 0.0f;
  const uint32_t wnx = bx_neg ? q0.w : q0.x, wfx = bx_neg ? q0.x : q0.w;
  const uint32_t wny = by_neg ? q1.x : q0.y, wfy = by_neg ? q0.y : q1.x;
  const uint32_t wnz = bz_neg ? q1.y : q0.z, wfz = bz_neg ? q0.z : q1.y;
  // plane distance = (origin + q * scale - o) * inv = q * (scale * inv) + (origin * inv + noi)
  const uint32_t ew = fbits(head.w);
  const float sx = bitsf((ew & 0xFFu) << 23) * r.inv.x, sy = bitsf(((ew >> 8) & 0xFFu) << 23) * r.inv.y, sz = bitsf(((ew >> 16) & 0xFFu) << 23) * r.inv.z;
  const float ox = __builtin_fmaf(head.x, r.inv.x, r.noi.x), oy = __builtin_fmaf(head.y, r.inv.y, r.noi.y), oz = __builtin_fmaf(head.z, r.inv.z, r.noi.z);
  const float inf = __builtin_inff();
  float k[4];
  uint32_t c[4] = {ch.x, ch.y, ch.z, ch.w};
#pragma unroll
  for (uint32_t j = 0; j < 4; j++) {
    const float ax = __builtin_fmaf(byte_f(wnx, j), sx, ox), ay = __builtin_fmaf(byte_f(wny, j), sy, oy), az = __builtin_fmaf(byte_f(wnz, j), sz, oz);
    const float fx = __builtin_fmaf(byte_f(wfx, j), sx, ox), fy = __builtin_fmaf(byte_f(wfy, j), sy, oy), fz = __builtin_fmaf(byte_f(wfz, j), sz, oz);
    const float tn = vmax3(ax, ay, vmax0(az));
    const float tf = vmin3(fx, fy, vmin2(fz, tmax));
    k[j] = (c[j] != kBvhEmpty && tn <= __builtin_fmaf(tf, 1.000004f, 1e-30f)) ? tn : inf;
  }
  float k0 = k[0], k1 = k[1], k2 = k[2], k3 = k[3];
  uint32_t c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
  if (kOrdered) { cswap(k0, c0, k1, c1); cswap(k2, c2, k3, c3); cswap(k0, c0, k2, c2); cswap(k1, c1, k3, c3); cswap(k1, c1, k2, c2); }
  else {  // unordered: real children first (stable), so that the conditional pushes below see them in front
    cswap(k0, c0, k1, c1); cswap(k2, c2, k3, c3); cswap(k0, c0, k2, c2); cswap(k1, c1, k3, c3); cswap(k1, c1, k2, c2);
  }
  if (k1 < inf) {
    if (k2 < inf) {
      if (k3 < inf) { stk.store(sp, top); sp++; top = SE::make(c3, k3); }
      stk.store(sp, top); sp++; top = SE::make(c2, k2);
    }
    stk.store(sp, top); sp++; top = SE::make(c1, k1);
  }
  return (k0 < inf) ? c0 : kBvhEmpty;
}

struct NodeData { float4 nx, ny, nz, fx, fy, fz; uint4 ch; };
LUM_DEV NodeData load_node(const NodeSource& src, uint32_t id, const TRay& r, RayStats& st) {
  NodeData n;
  const uint32_t b = id << 7;
  if (id < src.lds_count) {
    const char* p = src.lds + b;
    const uint32_t z = lds_node_swizzle_bytes(id);
    n.nx = *reinterpret_cast<const float4*>(p + (r.nx ^ z)); n.ny = *reinterpret_cast<const float4*>(p + (r.ny ^ z)); n.nz = *reinterpret_cast<const float4*>(p + (r.nz ^ z));
    n.fx = *reinterpret_cast<const float4*>(p + (r.fx ^ z)); n.fy = *reinterpret_cast<const float4*>(p + (r.fy ^ z)); n.fz = *reinterpret_cast<const float4*>(p + (r.fz ^ z));
    n.ch = *reinterpret_cast<const uint4*>(p + (96u ^ z));
    st.lds_nodes++;
  }
  else {
    const Bvh4Node* __restrict__ nodes = src.global;
    n.nx = node_f4(nodes, b + r.nx); n.ny = node_f4(nodes, b + r.ny); n.nz = node_f4(nodes, b + r.nz);
    n.fx = node_f4(nodes, b + r.fx); n.fy = node_f4(nodes, b + r.fy); n.fz = node_f4(nodes, b + r.fz);
    n.ch = node_u4(nodes, b + 96u);
  }
  return n;
}

// Visibility rays visit every stacked node anyway (their segment never shrinks), so a lane whose newest stack entry is an inner node of the same
// level takes it along: both nodes' lines are requested before either is tested, which halves the dependent round trips of a ray (the idea of
// round 2, when the kernels were thought to be bound by those; the counters of round 3 say issue rate and address unit). Measured (LUM_DUAL_VISIT=1): NOT faster - visibility kernel 121.9 -> 144.4 ms per 3 steps
// on the hall, 32.6 -> 36.1 on the scan: rays that find an occluder have fetched a node they would never have visited (nodes per ray 15.2 -> 16.2),
// lanes with and without a second node diverge, and the iteration carries twice the registers. Off. `second` = kBvhEmpty for lanes without such an entry. Every child of the second node
// that the ray may touch is pushed; of the first node's children the nearest is continued with, as in visit_node.
template <bool kCull, typename S>
LUM_DEV uint32_t visit_two_nodes(const NodeSource& src, uint32_t cur, uint32_t second, const TRay& r, float tmax, S& stk,
                                 int& sp, typename StackEntry<kCull>::E& top, RayStats& st) {
  using SE = StackEntry<kCull>;
  const float inf = __builtin_inff();
  const NodeData a = load_node(src, cur, r, st);
  NodeData b;
  const bool two = second != kBvhEmpty;
  if (two) b = load_node(src, second, r, st);
  {
    float e0 = inf, e1 = inf, e2 = inf, e3 = inf;
    uint32_t d0 = kBvhEmpty, d1 = kBvhEmpty, d2 = kBvhEmpty, d3 = kBvhEmpty;
    if (two) {
      e0 = child_entry(b.nx.x, b.ny.x, b.nz.x, b.fx.x, b.fy.x, b.fz.x, r, tmax);
      e1 = child_entry(b.nx.y, b.ny.y, b.nz.y, b.fx.y, b.fy.y, b.fz.y, r, tmax);
      e2 = child_entry(b.nx.z, b.ny.z, b.nz.z, b.fx.z, b.fy.z, b.fz.z, r, tmax);
      e3 = child_entry(b.nx.w, b.ny.w, b.nz.w, b.fx.w, b.fy.w, b.fz.w, r, tmax);
      d0 = b.ch.x; d1 = b.ch.y; d2 = b.ch.z; d3 = b.ch.w;
      if (e3 < inf) { stk.store(sp, top); sp++; top = SE::make(d3, e3); }
      if (e2 < inf) { stk.store(sp, top); sp++; top = SE::make(d2, e2); }
      if (e1 < inf) { stk.store(sp, top); sp++; top = SE::make(d1, e1); }
      if (e0 < inf) { stk.store(sp, top); sp++; top = SE::make(d0, e0); }
    }
  }
  float k0 = child_entry(a.nx.x, a.ny.x, a.nz.x, a.fx.x, a.fy.x, a.fz.x, r, tmax);
  float k1 = child_entry(a.nx.y, a.ny.y, a.nz.y, a.fx.y, a.fy.y, a.fz.y, r, tmax);
  float k2 = child_entry(a.nx.z, a.ny.z, a.nz.z, a.fx.z, a.fy.z, a.fz.z, r, tmax);
  float k3 = child_entry(a.nx.w, a.ny.w, a.nz.w, a.fx.w, a.fy.w, a.fz.w, r, tmax);
  uint32_t c0 = a.ch.x, c1 = a.ch.y, c2 = a.ch.z, c3 = a.ch.w;
  cswap(k0, c0, k1, c1); cswap(k2, c2, k3, c3); cswap(k0, c0, k2, c2); cswap(k1, c1, k3, c3); cswap(k1, c1, k2, c2);
  if (k1 < inf) {
    if (k2 < inf) {
      if (k3 < inf) { stk.store(sp, top); sp++; top = SE::make(c3, k3); }
      stk.store(sp, top); sp++; top = SE::make(c2, k2);
    }
    stk.store(sp, top); sp++; top = SE::make(c1, k1);
  }
  return (k0 < inf) ? c0 : kBvhEmpty;
}

// Pops the newest entry into (node, tnear) and refills the register top from scratch. The bottom of the stack is a sentinel
// (kTraversalDone) that is never removed.
template <typename S>
LUM_DEV uint2 stack_pop(const S& stk, int& sp, uint2& top) {
  const uint2 e = top;
  if (sp > 0) { sp--; top = stk.load(sp); }
  else top = make_uint2(kTraversalDone, 0u);
  return e;
}
template <typename S, typename E>
LUM_DEV void stack_push(S& stk, int& sp, E& top, E e) { stk.store(sp, top); sp++; top = e; }

LUM_DEV float4 tri_f4(const BvhTri* tris, uint32_t index, uint32_t word) { return reinterpret_cast<const float4*>(tris + index)[word]; }

// All triangles of a leaf are fetched before the first test, so a leaf costs one memory round trip instead of one per triangle.
// The loads (3 per slot; twelve with the four slots of rounds 1-5, six since round 6) are unconditional - slots beyond the leaf's count re-read its last triangle (same cache lines, nothing new is fetched): with a
// branch per slot the compiler parked the slots in other registers behind `s_waitcnt`s of their own, and a four-triangle leaf waited for memory
// up to four times in a row (measured: 9.5 us per triangle phase of a wave against 2 us per node phase).
struct LeafTris {
  float4 a[kBvhLeafMaxTri], b[kBvhLeafMaxTri], c[kBvhLeafMaxTri];
  // kAllSlots: slots beyond the leaf's count re-read its last triangle (see above); otherwise a branch per slot. Measured on the hall, same box: the
  // visibility kernel 386 -> 374 ms per 3 steps with all slots, the closest-hit kernel 215 -> 221 - each query type takes what suits it (round 6, two slots: closest-hit +2.2 % with all slots).
  template <bool kAllSlots>
  LUM_DEV void load(const BvhTri* __restrict__ tris, uint32_t first, uint32_t count) {
#pragma unroll
    for (uint32_t j = 0; j < kBvhLeafMaxTri; j++) {
      if (kAllSlots) {
        const uint32_t t = first + min(j, count - 1u);
        a[j] = tri_f4(tris, t, 0); b[j] = tri_f4(tris, t, 1); c[j] = tri_f4(tris, t, 2);
      }
      else if (j < count) { a[j] = tri_f4(tris, first + j, 0); b[j] = tri_f4(tris, first + j, 1); c[j] = tri_f4(tris, first + j, 2); }
    }
  }
};

// Experiment (LUM_PREFETCH, off): after a node visit the entry that will be popped next is known (`top`); its 128-byte line is requested with
// an LDS-DMA load into a per-wave junk area (no destination register, nothing waits for the data), so that the later visit finds it in L1/L2.
#ifndef LUM_PREFETCH
#define LUM_PREFETCH 0
#endif
LUM_DEV void prefetch_line(const void* p, uint32_t* wave_sink) {
#if LUM_PREFETCH
  __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*) p, (void __attribute__((address_space(3)))*) wave_sink, 4, 0, 0);
#else
  (void) p; (void) wave_sink;
#endif
}

// ---- the persistent two-level traversal ----
// A query type Q provides (all per lane):
//   bool load(sc, idx, origin, dir, tmax)   read item idx; false = nothing to trace
//   bool on_tris(sc, inst, first, count, o, d, tmax)   test `count` triangles starting at blas_tris[first] against the
//                                                      object-space ray; may shrink tmax; true = stop this ray
//   void finish(sc, idx)                    write the result of the finished ray
// Top-level leaves hold exactly one instance; entering it pushes a marker and maps the ray with the instance's world->object
// matrix (an affine map preserves distances along the ray, so tmax and the stacked entry distances stay valid across levels).
template <class Q>
LUM_DEV void trace_items(const DeviceScene& sc, uint32_t n, uint32_t* __restrict__ cursor, Q& q, RayStats& st, uint32_t& rays, uint32_t lds_count) {
  if (n == 0u) return;  // wave-uniform (a kernel argument read from the control words): an empty launch - the ambient reuse's fallback list on an opaque scene - stages nothing
  using SE = StackEntry<Q::kCull>;
  using E = typename SE::E;
  // Dual visits (visit_two_nodes) push all four children of the second node, which is not depth-first any more: the guard below allows them only
  // while fewer than kDualLimit entries are stacked, after which single visits need at most 3 more per remaining level (<= 126): 256 entries.
  constexpr int kDualLimit = 120;
  E stack_in_scratch[(LUM_DUAL_VISIT && Q::kDual) ? 2 * kStackSize : kStackSize];
  int sp = 0;
  TRay r;
  V3 wo = v3(0.0f, 0.0f, 0.0f), wd = v3(0.0f, 0.0f, 1.0f);
  float tmax = 0.0f;
  uint32_t cur = kTraversalDone, inst = kNoInstance, idx = 0;
  uint32_t postponed = kBvhEmpty;  // Q::kSpeculate: a leaf of the current instance set aside for the wave's next triangle phase
  r.set(wo, wd);
  bool more = true;
  // Experiment (LUM_DEFER_FINISH): results are stores, and on this part a store counts on the same in-order counter as the loads (vmcnt): written where
  // the ray ends, inside the phase loop, it sits in front of the next iteration's node or triangle loads, whose wait then also waits for the store to
  // be acknowledged - and about one ray of a wave ends per iteration. With the flag a lane keeps its result until it takes its next ray (the stores go
  // out in the refill, before the new rays' own loads). Measured, same box: hall visibility kernel 373.5 -> 380.5 ms per 3 steps, closest-hit
  // unchanged; scan 70.9 -> 69.5 and 64.1 -> 63.7: the acknowledgements are not what the iterations wait for. Off.
  bool unwritten = false;
  const uint32_t lane = threadIdx.x & 63u;
  const unsigned long long below = (1ull << lane) - 1ull;
  // stage the top of the tree
  extern __shared__ float4 lds_top[];
  {
    const float4* __restrict__ g = reinterpret_cast<const float4*>(sc.bvh_nodes);
    for (uint32_t i = threadIdx.x; i < lds_count * (kNodeBytes / 16u); i += blockDim.x) lds_top[lds_slot_swizzle(i)] = g[i];
    __syncthreads();
  }
  // ... and the records of the first top-level leaves (the rows of an instance's world->object matrix): entering one of those instances costs no
  // memory round trip. A scene of one mesh, like the hall, has one; 72 instances are 4.5 KB.
  __shared__ float4 lds_leaves[4u * LUM_LDS_INSTANCES];
  const uint32_t staged_leaves = min(sc.tlas_num_leaves, (uint32_t) LUM_LDS_INSTANCES);
  for (uint32_t i = threadIdx.x; i < 4u * staged_leaves; i += blockDim.x) lds_leaves[i] = sc.tlas_leaves[i];
  __syncthreads();
  const NodeSource nodes{sc.bvh_nodes, reinterpret_cast<const char*>(lds_top), lds_count};
  typedef typename TraversalStack<E>::W StackW;
  static_assert(sizeof(StackW) == sizeof(E), "a stack entry is one machine word");
  TraversalStack<E> stk{(typename TraversalStack<E>::ScratchPtr) reinterpret_cast<StackW*>(stack_in_scratch),
                        (typename TraversalStack<E>::LdsPtr) (reinterpret_cast<StackW*>(reinterpret_cast<char*>(lds_top) + lds_count * kNodeBytes) + threadIdx.x),
                        (int) (LUM_LDS_STACK_BYTES / (kRayBlockMax * (uint32_t) sizeof(E)))};
#if LUM_PREFETCH
  __shared__ uint32_t prefetch_sink[kPrefetchSinkWords];
  uint32_t* wave_sink = prefetch_sink + (threadIdx.x >> 6) * 64u;
#endif

  E top = SE::make(kTraversalDone, 0.0f);
#ifdef LUM_PHASE_STATS
  uint32_t phase_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long ptime_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_kernel_ = __builtin_readcyclecounter();
#endif
  auto pop = [&]() {
    LUM_PHASE(6); LUM_PHASE_LANES(7);
    // One exit condition and a predicated load per iteration; the loop carries only (sp, top): restoring the world-space ray inside it
    // made every ray register loop-carried (two dozen v_mov per iteration). The bottom of the stack is the kTraversalDone sentinel, so
    // any other entry has something below it.
    bool left_instance = false, again;
    E e;
    do {
      e = top;
      const bool done = SE::node(e) == kTraversalDone, leave = SE::node(e) == kLeaveInstance;
      // (an unconditional load, with the sentinel kept in memory or selected afterwards, was measured 2-16 % slower: the loaded entry
      // must flow into `top` untouched so that nothing waits for it before the next node's loads are in flight)
      if (!done) { sp--; top = stk.load(sp); }
      left_instance |= leave;
      again = !done && (leave || !SE::reachable(e, tmax));
    } while (again);
    cur = SE::node(e);
    if (left_instance) { inst = kNoInstance; r.set(wo, wd); }
  };

  // Work distribution: a wave reserves a chunk of consecutive items with one atomic and hands them to its idle lanes; a single
  // global cursor bumped once per refill would serialise every wave of the GPU on one L2 atomic.
  const uint32_t waves = gridDim.x * (blockDim.x / 64u);
  uint32_t chunk = n / (waves * 2u);
  chunk = (min(max(chunk, 64u), LUM_CHUNK_MAX) + 63u) & ~63u;
  uint32_t chunk_next = 0, chunk_end = 0;
#ifndef LUM_XCD_RANGES
#define LUM_XCD_RANGES 0
#endif
#if LUM_XCD_RANGES
  // Experiment: every XCD has its own L2. The queue is cut into 8 contiguous ranges (the queue order is pixel order at depth 0 and stays
  // roughly that through the compactions), XCD x works on range x first and helps with the others when its own is used up, so that the eight
  // L2s hold different parts of the scene instead of eight copies of the same hot set. `cursor` then points at 8 words.
  const uint32_t xcd = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;  // HW_REG_XCC_ID
  const uint32_t range_len = (n + 7u) / 8u;
  uint32_t ranges_left = 8u, range = xcd;
#endif

  while (true) {
    const unsigned long long idle = __ballot(cur == kTraversalDone);
#ifdef LUM_PHASE_STATS
    const unsigned long long t_refill_ = __builtin_readcyclecounter();
    const bool refilling_ = idle != 0ull && more;
#endif
    if (idle != 0ull && more) {  // wave-uniform
      if (chunk_next >= chunk_end) {
#if LUM_XCD_RANGES
        while (ranges_left > 0u) {
          const uint32_t lo = min(range * range_len, n), hi = min(lo + range_len, n);
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(cursor + range, chunk);
          base = __builtin_amdgcn_readfirstlane(base) + lo;
          if (base < hi) { chunk_next = base; chunk_end = min(base + chunk, hi); break; }
          range = (range + 1u) & 7u;  // this range is handed out completely: help with the next one
          ranges_left--;
        }
        more = ranges_left > 0u;
#else
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(cursor, chunk);
        base = __builtin_amdgcn_readfirstlane(base);
        chunk_next = base;
        chunk_end = min(base + chunk, n);
        more = base < n;
#endif
      }
      if (more) {
        const uint32_t avail = chunk_end - chunk_next, want = (uint32_t) __popcll(idle);
        const uint32_t rank = (uint32_t) __popcll(idle & below);
        if (cur == kTraversalDone && rank < avail) {
          if (unwritten) { q.finish(sc, idx); unwritten = false; }
          idx = chunk_next + rank;
          if (q.load(sc, idx, wo, wd, tmax)) {
            rays++;
            // A ray with a non-finite origin or direction hits nothing (every comparison of the CPU oracle's slab test is false for it). Here
            // the hardware's min/max drop NaN operands, which would make every child slot - the empty ones too - look entered.
            const uint32_t e = 0x7F800000u;
            const bool finite = (fbits(wo.x) & e) != e && (fbits(wo.y) & e) != e && (fbits(wo.z) & e) != e && (fbits(wd.x) & e) != e && (fbits(wd.y) & e) != e &&
                                (fbits(wd.z) & e) != e;
            if (finite) { r.set(wo, wd); cur = 0; sp = 0; top = SE::make(kTraversalDone, 0.0f); inst = kNoInstance; }
            else { if (LUM_DEFER_FINISH) unwritten = true; else q.finish(sc, idx); }
          }
        }
        chunk_next += min(want, avail);
      }
    }
#ifdef LUM_PHASE_STATS
    if (refilling_) { ptime_[8] += __builtin_readcyclecounter() - t_refill_; ptime_[9]++; }
#endif
    if (__ballot(cur != kTraversalDone) == 0ull) break;
    LUM_PHASE(5);

    // Phase vote: every iteration the wave runs ONE phase (node visit, instance entry or triangle tests), the one most of its lanes
    // are waiting for. Lanes holding a triangle leaf idle while the others keep walking nodes only until they are the larger group
    // (and the other way round), instead of waiting until every lane has found a leaf: measured lane occupancy of the node phase
    // was 0.33-0.41 with the plain while-while loop.
    while (true) {
      if (Q::kSpeculate) {
        // a lane on a leaf of an instance, nothing set aside yet, whose newest stack entry is an inner node within reach (markers, leaves and the
        // sentinel carry the leaf bit): the leaf waits, the lane walks on. Such an entry always has something below it (the sentinel at least).
        const uint32_t t = SE::node(top);
        if (cur != kTraversalDone && (cur & kBvhLeafBit) && inst != kNoInstance && postponed == kBvhEmpty && !(t & kBvhLeafBit) && SE::reachable(top, tmax)) {
          postponed = cur; cur = t; sp--; top = stk.load(sp);
        }
      }
      const bool live = cur != kTraversalDone;
      const bool at_leaf = live && (cur & kBvhLeafBit);
      // (with speculation: a lane counts for the triangle vote when it cannot go on without the triangle phase; lanes that hold a postponed leaf and a
      // node take part in both phases)
      const bool want_tris = at_leaf && inst != kNoInstance, want_enter = at_leaf && inst == kNoInstance;
      const uint32_t n_live = (uint32_t) __popcll(__ballot(live)), n_tris = (uint32_t) __popcll(__ballot(want_tris)), n_enter = (uint32_t) __popcll(__ballot(want_enter));
      if (n_live == 0u) break;
      const uint32_t n_nodes = n_live - n_tris - n_enter;
#if LUM_LDS_TURN
      // A wave iteration ends when its slowest lane has its data: one lane whose node comes from memory makes all lanes wait a memory round trip
      // (about 3 us under this kernel's own load), lanes whose node sits in the staged tree top included - and those are 60 % of the node visits.
      // So while enough lanes stand on staged nodes they get iterations of their own, which issue no vector-memory load at all and take a
      // quarter of the time; the lanes on other nodes wait those out and are then served together.
      const bool on_staged = live && ((!at_leaf && cur < lds_count) || (want_enter && (cur & 0x0FFFFFFFu) < staged_leaves));
      const uint32_t n_staged = (uint32_t) __popcll(__ballot(on_staged));
      const bool staged_turn = n_staged >= LUM_LDS_TURN && (LUM_LDS_FIRST || n_tris * LUM_VOTE_TRIS < max(n_nodes, n_enter) * LUM_VOTE_NODES) && (LUM_LDS_FIRST || n_enter < n_nodes);
      const bool run_tris = !staged_turn && n_tris * LUM_VOTE_TRIS >= max(n_nodes, n_enter) * LUM_VOTE_NODES;
      const bool run_enter = !staged_turn && !run_tris && n_enter >= n_nodes;
#else
      const bool run_tris = n_tris * LUM_VOTE_TRIS >= max(n_nodes, n_enter) * LUM_VOTE_NODES;
      const bool run_enter = !run_tris && n_enter >= n_nodes;
#endif
      // Three per-lane conditions of which the vote leaves at most one non-empty. Written as independent divergent ifs on purpose: with
      // wave-uniform if/else-if branches the compiler routed every ray register through a temporary and back at the merge point
      // (about forty v_mov per iteration).
#if LUM_LDS_TURN
      const bool do_tris = run_tris && want_tris, do_enter = (run_enter && want_enter) || (staged_turn && want_enter && on_staged),
                 do_node = !run_tris && !run_enter && live && !at_leaf && (!staged_turn || on_staged);
#else
      const bool do_tris = run_tris && (want_tris || (Q::kSpeculate && postponed != kBvhEmpty)), do_enter = run_enter && want_enter, do_node = !run_tris && !run_enter && live && !at_leaf;
#endif
      LUM_TIME_BEGIN();
#ifdef LUM_PHASE_STATS
      const bool memory_node_ = __ballot(do_node && cur >= lds_count) != 0ull;
#endif
      {
        if (do_tris) {
          LUM_PHASE(3); LUM_PHASE_LANES(4);
          if (Q::kSpeculate && postponed != kBvhEmpty) {  // the leaf set aside first; the lane's place in the traversal (cur, a node or another leaf) stays
            const uint32_t leaf = postponed;
            postponed = kBvhEmpty;
            if (q.on_tris(sc, inst, leaf & 0x0FFFFFFFu, ((leaf >> 28) & 0x7u) + 1u, r.o, r.d, tmax, st)) cur = kTraversalDone;
          }
          else if (q.on_tris(sc, inst, cur & 0x0FFFFFFFu, ((cur >> 28) & 0x7u) + 1u, r.o, r.d, tmax, st)) cur = kTraversalDone;
          else pop();
          if (cur == kTraversalDone) { if (LUM_DEFER_FINISH) unwritten = true; else q.finish(sc, idx); }
        }
      }
      {
        if (do_enter) {
          LUM_PHASE(1); LUM_PHASE_LANES(2);
          const uint32_t leaf_index = cur & 0x0FFFFFFFu;
          float4 r0, r1, r2, meta;
          if (leaf_index < staged_leaves) { const float4* leaf = lds_leaves + 4u * leaf_index; r0 = leaf[0]; r1 = leaf[1]; r2 = leaf[2]; meta = leaf[3]; }
          else { const float4* __restrict__ leaf = sc.tlas_leaves + 4u * leaf_index; r0 = leaf[0]; r1 = leaf[1]; r2 = leaf[2]; meta = leaf[3]; }
          inst = fbits(meta.x);
          const float px = wo.x - r0.w, py = wo.y - r1.w, pz = wo.z - r2.w;
          const V3 oo = v3(mat_row_apply(r0.x, r0.y, r0.z, px, py, pz), mat_row_apply(r1.x, r1.y, r1.z, px, py, pz), mat_row_apply(r2.x, r2.y, r2.z, px, py, pz));
          const V3 od = v3(mat_row_apply(r0.x, r0.y, r0.z, wd.x, wd.y, wd.z), mat_row_apply(r1.x, r1.y, r1.z, wd.x, wd.y, wd.z),
                           mat_row_apply(r2.x, r2.y, r2.z, wd.x, wd.y, wd.z));
          r.set(oo, od);
          stack_push(stk, sp, top, SE::make(kLeaveInstance, 0.0f));
          cur = fbits(meta.y);
        }
      }
      {
        if (do_node) {
          LUM_PHASE(0);
          st.nodes++;
#if LUM_DUAL_VISIT
          if (Q::kDual) {
            uint32_t second = kBvhEmpty;
            const uint32_t t = SE::node(top);
            if (!(t & kBvhLeafBit) && sp < kDualLimit) {  // the newest entry is an inner node (markers and leaves carry the leaf bit): same level as `cur`
              second = t;
              st.nodes++;
              if (sp > 0) { sp--; top = stk.load(sp); }
              else top = SE::make(kTraversalDone, 0.0f);
            }
            cur = visit_two_nodes<Q::kCull>(nodes, cur, second, r, tmax, stk, sp, top, st);
          }
          else
#endif
#if LUM_BVH8
          cur = visit_node8<Q::kOrdered, Q::kCull>(nodes, cur, r, tmax, stk, sp, top, st);
#elif LUM_BVH4Q
          cur = visit_node_q<Q::kOrdered, Q::kCull>(nodes, cur, r, tmax, stk, sp, top, st);
#else
          cur = visit_node<Q::kOrdered, Q::kCull, Q::kFarFirst>(nodes, cur, r, tmax, stk, sp, top, st);
#endif
#if LUM_PREFETCH
          {
            const uint32_t t = SE::node(top);
            if (!(t & kBvhLeafBit)) { if (t >= lds_count) prefetch_line(sc.bvh_nodes + t, wave_sink); }
#if LUM_PREFETCH >= 2
            else if (t < kTraversalDone && inst != kNoInstance) prefetch_line(sc.blas_tris + (t & 0x0FFFFFFFu), wave_sink);
#endif
          }
#endif
          if (cur == kBvhEmpty) {
            // (with a leaf set aside nothing is popped: the next entry might be the way out of the instance the leaf belongs to)
            if (Q::kSpeculate && postponed != kBvhEmpty) { cur = postponed; postponed = kBvhEmpty; }
            else {
              pop();
              if (cur == kTraversalDone) { if (LUM_DEFER_FINISH) unwritten = true; else q.finish(sc, idx); }
            }
          }
        }
      }
#ifdef LUM_PHASE_STATS
      {  // a wave iteration ends when every lane's registers are written: make the clock wait for what the phase loaded
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const int kind = run_tris ? 4 : (run_enter ? 6 : (memory_node_ ? 0 : 2));
        LUM_TIME_END(kind);
      }
#endif
      if (more && n_live < LUM_REFILL) break;
    }
  }
  if (unwritten) q.finish(sc, idx);  // the lanes' last rays
#ifdef LUM_PHASE_STATS
  ptime_[10] = __builtin_readcyclecounter() - t_kernel_; ptime_[11] = 1;
  for (int k = 0; k < 8; k++) if (phase_[k]) atomicAdd(&g_phase[k], (unsigned long long) phase_[k]);
  if ((threadIdx.x & 63u) == 0u) for (int k = 0; k < 12; k++) if (ptime_[k]) atomicAdd(&g_phase_time[k], ptime_[k]);
#endif
}

LUM_NS_END
#if LUM_BVH8O
#include "dev_trace8.h"
#define LUM_TRACE_ITEMS trace_items8
#else
#define LUM_TRACE_ITEMS trace_items
#endif
LUM_NS_BEGIN

struct Hit { uint32_t instance_id, tri_id; float t; uint32_t scene_tri; };

// Nearest hit in [0, FLT_MAX); optionally ignoring the triangle the path is leaving (STATE_FLAG_USE_IGNORE_HANDLE).
struct ClosestState {
  static constexpr bool kDual = false;
  static constexpr bool kSpeculate = (LUM_SPECULATE & 1) != 0;
  static constexpr bool kOrdered = true;
  static constexpr int kFarFirst = 0;
  static constexpr bool kCull = true;  // stack entries carry the entry distance: a pop drops children beyond the nearest hit so far
  // Two facts ride along for the ambient-visibility reuse (TraceQuery / k_resolve_reuse, kernels.h) without a vector register of their own: bit 31 of
  // best.scene_tri says that the nearest hit so far is an untextured alpha-1 triangle (kBvhTriOpaque: it would stop a visibility ray on its own), `cutout`
  // (a lane mask in scalar registers, like use_ignore) that an alpha cut-out was skipped on the way (a visibility ray may have to multiply such a
  // texel's colour in). A visibility ray along the same ray over (eps, FLT_MAX) that ignores the same triangle reports 1 where nothing was hit and no
  // cut-out skipped, 0 where the nearest hit lies beyond eps and is opaque on its own; everything else has to be traced. result() hands out the clean
  // triangle index.
  static constexpr uint32_t kOpaqueBit = 0x80000000u;
  bool use_ignore, cutout;
  uint32_t ign_inst, ign_tri;
  Hit best;
  LUM_DEV void begin(bool ignore, uint32_t inst, uint32_t tri) { use_ignore = ignore; cutout = false; ign_inst = inst; ign_tri = tri; best = Hit{kHitSky, 0u, kFltMax, 0u}; }
#ifndef LUM_CLOSEST_ALL_SLOTS
#define LUM_CLOSEST_ALL_SLOTS 0  // 1: the closest-hit rays fetch every leaf slot unconditionally, as the visibility rays do (LeafTris::load)
#endif
  LUM_DEV bool on_tris(const DeviceScene& sc, uint32_t inst, uint32_t first, uint32_t count, V3 o, V3 d, float& tmax, RayStats& st) {
    LeafTris lt;
    lt.load<LUM_CLOSEST_ALL_SLOTS != 0>(sc.blas_tris, first, count);
#pragma unroll
    for (uint32_t j = 0; j < kBvhLeafMaxTri; j++) {
      if (j >= count) break;
      const float4 a = lt.a[j], b = lt.b[j], c = lt.c[j];
      const uint32_t id = fbits(a.w);
      st.tris++;
      if (use_ignore && inst == ign_inst && id == ign_tri) continue;
      F2 uv;
      const float t = intersect_triangle(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), o, d, uv);
      if (t < best.t || (t == best.t && t != kFltMax && (inst < best.instance_id || (inst == best.instance_id && id < best.tri_id)))) {
        // alpha cut-outs: texels with alpha 0 do not exist for the ray (optix_common.cuh:20-46, optix_anyhit.cuh:26-30)
        if (fbits(c.w) < sc.num_textures &&  // kBvhTriNoTexture and kBvhTriOpaque are no texture ids
            texture_load(sc, fbits(c.w), triangle_uv(sc.tri_tex[fbits(b.w)], uv), true, make_float4(0.0f, 0.0f, 0.0f, 1.0f)).w == 0.0f) {
          cutout = true;
          continue;
        }
        best.instance_id = inst; best.tri_id = id; best.t = t; best.scene_tri = fbits(b.w) | (fbits(c.w) == kBvhTriOpaque ? kOpaqueBit : 0u); tmax = t;
      }
    }
    return false;
  }
  LUM_DEV Hit result() const { return (best.t == kFltMax) ? Hit{kHitSky, 0u, kFltMax, 0u} : Hit{best.instance_id, best.tri_id, best.t, best.scene_tri & ~kOpaqueBit}; }
};

// Transparency along (eps, dist): product over crossed surfaces, zero as soon as one is opaque. Skips the sampled light
// (`target`) and the surface being shaded (`self`). The order in which a traversal meets the surfaces is arbitrary and a float
// product of three or more factors depends on it, so the product is carried in binary64 (exact for two factors, 29 guard bits
// beyond that) and rounded to binary32 once at the end; the oracle does the same.
struct ShadowState {
  static constexpr bool kOrdered = true;  // an unordered visit was measured: fewer instructions, same time, so the common path is kept
  // Child order of a visibility ray: FARTHEST first. Any occluder ends the ray, so the order should lead to one quickly, and for a ray that leaves a
  // surface the near boxes are the worst place to look: they hold the surface the ray has just left and its neighbours, which the ray moves away
  // from, while the far end of a ray into a closed scene ends in a wall. Measured (hall / scan / Example-class scene): node visits per visibility ray
  // 15.2 -> 11.4 / 13.0 -> 11.1 / 11.4 -> 9.6, the kernel -29 % / -11 % / -14 %; all of it from the rays without an end point (ambient, sun) -
  // for the segments towards sampled lights neither order is better (modes 2 / 3 below apply the order to one kind only; tools/bvh_quality.cpp
  // models it on the CPU: 17.3 -> 8.3 visits for cosine-distributed rays from the hall's surfaces). Results do not depend on the order (see below).
#ifndef LUM_SHADOW_ORDER
#define LUM_SHADOW_ORDER 1  // 0 nearest child first, 1 farthest first, 2 farthest first for rays without an end point only, 3 for segments only
#endif
  static constexpr int kFarFirst = LUM_SHADOW_ORDER;
#ifndef LUM_SHADOW_CULL
#define LUM_SHADOW_CULL 0
#endif
  static constexpr bool kCull = LUM_SHADOW_CULL != 0;  // the segment never shrinks: 4-byte stack entries (StackEntry<false>)
  static constexpr bool kDual = true;                  // two nodes per visit where the stack offers a second one (visit_two_nodes)
  static constexpr bool kSpeculate = (LUM_SPECULATE & 2) != 0;
  uint32_t tgt_inst, tgt_tri, self_inst, self_tri;
  float dist;
#if LUM_FAST
  using Acc = float;   // the fast flavour does not promise the last bit: the product in traversal order, three registers fewer
#else
  using Acc = double;
#endif
  Acc tr, tg, tb;
  bool blocked;
  LUM_DEV void begin(uint4 ids, float d) { tgt_inst = ids.x; tgt_tri = ids.y; self_inst = ids.z; self_tri = ids.w; dist = d; tr = tg = tb = (Acc) 1.0; blocked = false; }
  LUM_DEV bool on_tris(const DeviceScene& sc, uint32_t inst, uint32_t first, uint32_t count, V3 o, V3 d, float&, RayStats& st) {
    LeafTris lt;
    lt.load<true>(sc.blas_tris, first, count);
#pragma unroll
    for (uint32_t j = 0; j < kBvhLeafMaxTri; j++) {
      if (j >= count) break;
      const float4 a = lt.a[j], b = lt.b[j], c = lt.c[j];
      const uint32_t id = fbits(a.w);
      st.tris++;
      if ((inst == tgt_inst && id == tgt_tri) || (inst == self_inst && id == self_tri)) continue;
      F2 uv;
      const float t = intersect_triangle(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), o, d, uv);
      if (!(t > kEps && t < dist)) continue;
      if (fbits(c.w) == kBvhTriOpaque) { blocked = true; return true; }  // untextured, alpha 1 (k_tri_opacity): what the material fetch below would find
      const uint4 tt = sc.tri_tex[fbits(b.w)];  // b.w = triangle index in the scene arrays
      const Material m = load_material(sc, tt.w & 0xFFFFu);
      Col albedo = m.albedo;
      float alpha = m.alpha;
      if (m.albedo_tex != kTextureNone) {  // optix_get_albedo_for_shadowing, optix_common.cuh:48-65
        const float4 af = (m.albedo_tex < sc.num_textures) ? texture_load(sc, m.albedo_tex, triangle_uv(tt, uv), true, make_float4(0.0f, 0.0f, 0.0f, 0.0f))
                                                           : make_float4(0.9f, 0.9f, 0.9f, 1.0f);
        albedo = col(af.x, af.y, af.z);
        alpha = af.w;
      }
      const bool colored = (m.flags & kDMatColoredTransparency) != 0;
      if (alpha == 1.0f) { blocked = true; return true; }
      if (alpha == 0.0f && !colored) continue;
      const float tp = 1.0f - alpha;
      const Col f = colored ? albedo * tp : splat(tp);
      tr *= (Acc) f.r; tg *= (Acc) f.g; tb *= (Acc) f.b;
    }
    return false;
  }
  LUM_DEV Col result() const { return blocked ? splat(0.0f) : col((float) tr, (float) tg, (float) tb); }
};

// ---- single-level traversal of the light-only BVH (rare: BSDF-sampled light directions) ----
template <typename LeafFn>
LUM_DEV void traverse_lights(const DeviceScene& sc, V3 o, V3 d, float& tmax, RayStats& st, LeafFn&& on_leaf) {
  uint2 stack_in_scratch[kStackSize];
  TraversalStack<uint2> stk{(TraversalStack<uint2>::ScratchPtr) reinterpret_cast<unsigned long long*>(stack_in_scratch), nullptr, 0};  // the light tree is shallow: scratch only
  int sp = 0;
  uint2 top = make_uint2(kTraversalDone, 0u);
  TRay r;
  r.set(o, d);
  uint32_t cur = 0;
  while (true) {
    if (cur & kBvhLeafBit) {
      on_leaf(cur & 0x0FFFFFFFu, ((cur >> 28) & 0x7u) + 1u, tmax);
      cur = kBvhEmpty;
    }
    else {
      st.nodes++;
      cur = visit_node<true, true>(NodeSource{sc.light_nodes, nullptr, 0u}, cur, r, tmax, stk, sp, top, st);
    }
    if (cur == kBvhEmpty) {
      while (true) {
        const uint2 e = stack_pop(stk, sp, top);
        cur = e.x;
        if (cur == kTraversalDone) return;
        if (within(bitsf(e.y), tmax)) break;
      }
    }
  }
}

// Light-BVH query on (eps, FLT_MAX). OptiX leaves the any-hit order unspecified, so the reference's reservoir is restated
// order-independently: pass 0 finds t* = nearest opaque light; pass 1 counts the lights with eps < t <= t* (except `self`
// and fully transparent uncoloured ones) and picks the one minimising squares32(0x9E3779B9*id + random bits), a uniform
// choice driven by the same random number.
LUM_DEV uint32_t light_query(const DeviceScene& sc, V3 origin, V3 dir, uint32_t self_inst, uint32_t self_tri, float random, uint32_t& num_hits,
                             RayStats& st) {
  float tstar = kFltMax;
  uint32_t best_id = kLightIdInvalid, best_key = 0xFFFFFFFFu, n = 0;
  {  // a non-finite ray hits no light (see trace_items): a BSDF-sampled direction can be NaN, e.g. at grazing refraction through the water surface
    const uint32_t e = 0x7F800000u;
    const bool finite = (fbits(origin.x) & e) != e && (fbits(origin.y) & e) != e && (fbits(origin.z) & e) != e && (fbits(dir.x) & e) != e && (fbits(dir.y) & e) != e &&
                        (fbits(dir.z) & e) != e;
    if (!finite) { num_hits = 0; return kLightIdInvalid; }
  }
  for (int pass = 0; pass < 2; pass++) {
    float tmax = tstar;
    traverse_lights(sc, origin, dir, tmax, st, [&](uint32_t first, uint32_t count, float& tm) {
      for (uint32_t j = 0; j < count; j++) {
        const float4 a = tri_f4(sc.light_tris, first + j, 0), b = tri_f4(sc.light_tris, first + j, 1), c = tri_f4(sc.light_tris, first + j, 2);
        const uint32_t light = fbits(a.w);
        st.tris++;
        const uint2 handle = sc.light_tri_handles[light];
        if (handle.x == self_inst && handle.y == self_tri) continue;
        F2 uv;
        const float t = intersect_triangle(v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), origin, dir, uv);
        if (!(t > kEps && t != kFltMax && t <= tstar)) continue;
        const uint32_t mesh = sc.instance_mesh_ids[handle.x];
        const uint4 tt = sc.tri_tex[sc.mesh_tri_offset[mesh] + handle.y];
        const Material m = load_material(sc, tt.w & 0xFFFFu);
        float alpha = m.alpha;
        if (m.albedo_tex != kTextureNone)  // optix_anyhit.cuh:158 -> optix_get_albedo_for_shadowing
          alpha = (m.albedo_tex < sc.num_textures) ? texture_load(sc, m.albedo_tex, triangle_uv(tt, uv), true, make_float4(0.0f, 0.0f, 0.0f, 0.0f)).w : 1.0f;
        if (alpha == 0.0f && (m.flags & kDMatColoredTransparency) == 0) continue;
        if (pass == 0) { if (alpha == 1.0f && t < tstar) { tstar = t; tm = t; } }
        else {
          n++;
          const uint32_t key = squares32(0xfcbd6e15u, 0x9E3779B9u * light + fbits(random));
          if (key < best_key || (key == best_key && light < best_id)) { best_key = key; best_id = light; }
        }
      }
    });
  }
  num_hits = n;
  return best_id;
}

// ---- what every translation unit with a persistent ray kernel needs (kernels.h, kernel_shadow.h) ----
// kTraceBlock (threads per workgroup of the persistent ray kernels) is defined in dev_trace.h, which lays the lanes' traversal stacks out by it
#ifndef LUM_TRACE_MIN_WAVES
#define LUM_TRACE_MIN_WAVES 0  // experiment: register budget of the ray kernels as waves per SIMD (0: whatever one workgroup of kTraceBlock threads per CU allows)
#endif
#if LUM_TRACE_MIN_WAVES
#define LUM_TRACE_BOUNDS __launch_bounds__(kTraceBlock, LUM_TRACE_MIN_WAVES)
#else
#define LUM_TRACE_BOUNDS __launch_bounds__(kTraceBlock)
#endif

LUM_DEV void flush_stats(uint64_t* counters, const RayStats& st, uint32_t rays, uint32_t ray_counter, uint32_t node_counter, uint32_t tri_counter,
                         uint32_t lds_counter = kCntCount) {
  // one atomic per wave and counter
  uint32_t n = st.nodes, t = st.tris, r = rays, l = st.lds_nodes;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { n += __shfl_down(n, off); t += __shfl_down(t, off); r += __shfl_down(r, off); l += __shfl_down(l, off); }
  if ((threadIdx.x & 63) == 0) {
    if (l && lds_counter != kCntCount) atomicAdd((unsigned long long*) &counters[lds_counter], (unsigned long long) l);
    if (n) atomicAdd((unsigned long long*) &counters[node_counter], (unsigned long long) n);
    if (t) atomicAdd((unsigned long long*) &counters[tri_counter], (unsigned long long) t);
    if (r) atomicAdd((unsigned long long*) &counters[ray_counter], (unsigned long long) r);
  }
}


LUM_NS_END
