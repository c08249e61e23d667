// What does one node visit cost, in the ray kernels' own setting? A prototype of the 8-wide octant-order node (Ylitie, Karras, Laine 2017: children
// stored in octant slots, quantised boxes, one "group" stack entry per visit, no distance sort; the reference's own dead software path walks such a
// tree: cuda/bvh.cuh:82-106, :146; node utils.h:123-138; slot assignment bvh.c:1093-1145) against the kernels' real 4-wide visit (dev_trace.h
// visit_node, included from the product's headers), both as endless random walks over a table of synthetic nodes: dependent gathers of 128-byte slots,
// the stack in LDS as in trace_items, 16 waves per CU, 64 or 32 lanes of a wave holding a ray. Output: visits per second and the static VALU
// instruction count of each visit routine (from the ISA: tools/isa_stats.py on this binary). Combined with the visits per ray of the CPU model
// (tools/bvh_quality.cpp BQ_WIDE=2: static slots, no distance cull at a pop) this says what an 8-wide closest-hit kernel could gain before
// trace_items is rewritten for it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLUM_FAST=1 -ffp-contract=fast -fno-slp-vectorize -I luminary_amd/csrc/device -o /tmp/node_visit tools/microbench/node_visit.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "dev_trace.h"

using namespace lum;

// ---- the 8-wide octant-order node: 80 bytes in a 128-byte slot ----
//   16 B  origin xyz | exponent x, y, z (biased) | imask (bit s: slot s holds an inner node)
//   16 B  child_base (the inner children are consecutive nodes, in slot order) | leaf_base | meta[8] (leaf slots: offset 5 bits | count - 1 in 2 bits)
//   48 B  lo_x[8] lo_y[8] | lo_z[8] hi_x[8] | hi_y[8] hi_z[8]   one byte per slot
// Slot s holds the child whose centre lies towards the octant direction s (bit a set: the +a side); a ray with direction signs `oct` (bit a set: d_a < 0)
// meets the slots roughly front to back in the order s ^ oct = 0, 1, 2 ... 7.
struct alignas(128) Node8 {
  float origin[3];
  uint8_t exp[3], imask;
  uint32_t child_base, leaf_base;
  uint8_t meta[8];
  uint8_t lo_x[8], lo_y[8], lo_z[8], hi_x[8], hi_y[8], hi_z[8];
  uint32_t pad[12];
};
static_assert(sizeof(Node8) == 128, "one slot");

struct Group { uint32_t base; uint32_t bits; };  // bits: hit mask in priority order (bit r: slot r ^ oct) | imask << 8

// Visit of node `cur`: the slots the ray may touch, as a mask in priority order. 5 x 16-byte loads.
__device__ __forceinline__ uint32_t visit8(const char* __restrict__ nodes, uint32_t cur, const TRay& r, uint32_t oct, float tmax, uint32_t& child_base, uint32_t& imask) {
  const char* p = nodes + ((size_t) cur << 7);
  const float4 head = *reinterpret_cast<const float4*>(p);
  const uint4 link = *reinterpret_cast<const uint4*>(p + 16);
  const uint4 q0 = *reinterpret_cast<const uint4*>(p + 32), q1 = *reinterpret_cast<const uint4*>(p + 48), q2 = *reinterpret_cast<const uint4*>(p + 64);
  const uint32_t ew = fbits(head.w);
  imask = ew >> 24;
  child_base = link.x;
  const float sx = bitsf((ew & 0xFFu) << 23) * r.inv.x, sy = bitsf(((ew >> 8) & 0xFFu) << 23) * r.inv.y, sz = bitsf(((ew >> 16) & 0xFFu) << 23) * r.inv.z;
  const float bx = __builtin_fmaf(head.x, r.inv.x, r.noi.x), by = __builtin_fmaf(head.y, r.inv.y, r.noi.y), bz = __builtin_fmaf(head.z, r.inv.z, r.noi.z);
  // near / far planes per axis by the direction's sign: lo_x = q0.xy, lo_y = q0.zw, lo_z = q1.xy, hi_x = q1.zw, hi_y = q2.xy, hi_z = q2.zw
  const bool nx = (oct & 1u) != 0, ny = (oct & 2u) != 0, nz = (oct & 4u) != 0;
  const uint32_t ax0 = nx ? q1.z : q0.x, ax1 = nx ? q1.w : q0.y, fx0 = nx ? q0.x : q1.z, fx1 = nx ? q0.y : q1.w;
  const uint32_t ay0 = ny ? q2.x : q0.z, ay1 = ny ? q2.y : q0.w, fy0 = ny ? q0.z : q2.x, fy1 = ny ? q0.w : q2.y;
  const uint32_t az0 = nz ? q2.z : q1.x, az1 = nz ? q2.w : q1.y, fz0 = nz ? q1.x : q2.z, fz1 = nz ? q1.y : q2.w;
  uint32_t hits = 0;
#pragma unroll
  for (uint32_t j = 0; j < 8; j++) {
    const uint32_t wnx = j < 4 ? ax0 : ax1, wny = j < 4 ? ay0 : ay1, wnz = j < 4 ? az0 : az1, wfx = j < 4 ? fx0 : fx1, wfy = j < 4 ? fy0 : fy1, wfz = j < 4 ? fz0 : fz1;
    const float tnx = __builtin_fmaf(byte_f(wnx, j & 3u), sx, bx), tny = __builtin_fmaf(byte_f(wny, j & 3u), sy, by), tnz = __builtin_fmaf(byte_f(wnz, j & 3u), sz, bz);
    const float tfx = __builtin_fmaf(byte_f(wfx, j & 3u), sx, bx), tfy = __builtin_fmaf(byte_f(wfy, j & 3u), sy, by), tfz = __builtin_fmaf(byte_f(wfz, j & 3u), sz, bz);
    const float tn = vmax3(tnx, tny, vmax0(tnz)), tf = vmin3(tfx, tfy, vmin2(tfz, tmax));
    hits |= (tn <= __builtin_fmaf(tf, 1.000004f, 1e-30f)) ? (1u << j) : 0u;  // empty slots carry inverted boxes
  }
  // slot order -> priority order: bit s moves to bit s ^ oct (three conditional swaps of neighbouring bit groups)
  hits = (oct & 1u) ? (((hits & 0x55u) << 1) | ((hits >> 1) & 0x55u)) : hits;
  hits = (oct & 2u) ? (((hits & 0x33u) << 2) | ((hits >> 2) & 0x33u)) : hits;
  hits = (oct & 4u) ? (((hits & 0x0Fu) << 4) | ((hits >> 4) & 0x0Fu)) : hits;
  return hits;
}

template <int KIND>
__global__ __launch_bounds__(1024) void k_walk(const char* __restrict__ table, uint32_t num_nodes, uint32_t steps, uint32_t active, uint32_t lds_nodes, unsigned long long* __restrict__ out) {
  extern __shared__ float4 lds_top[];
  for (uint32_t i = threadIdx.x; i < lds_nodes * 8u; i += blockDim.x) lds_top[i] = reinterpret_cast<const float4*>(table)[i];
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u;
  if (lane >= active) return;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t rng = gid * 2654435761u + 12345u;
  auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; };
  auto unit = [&]() { return (next() >> 8) * (1.0f / 16777216.0f); };
  TRay r;
  V3 o = v3(unit(), unit(), unit()), d = v3(unit() - 0.5f, unit() - 0.5f, unit() - 0.5f);
  d = d * (1.0f / sqrtf(d.x * d.x + d.y * d.y + d.z * d.z + 1e-12f));
  r.set(o, d);
  const uint32_t oct = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
  float tmax = 3.0f;
  unsigned long long visits = 0, acc = 0, hit_children = 0;
  RayStats st{0, 0, 0};
  if (KIND == 4) {
    using SE = StackEntry<true>;
    using E = SE::E;
    E scratch_stack[kStackSize];
    typedef TraversalStack<E>::W StackW;
    TraversalStack<E> stk{(TraversalStack<E>::ScratchPtr) reinterpret_cast<StackW*>(scratch_stack),
                          (TraversalStack<E>::LdsPtr) (reinterpret_cast<StackW*>(reinterpret_cast<char*>(lds_top) + lds_nodes * 128u) + threadIdx.x), 8};
    const NodeSource src{reinterpret_cast<const Bvh4Node*>(table), reinterpret_cast<const char*>(lds_top), lds_nodes};
    int sp = 0;
    E top = SE::make(kTraversalDone, 0.0f);
    uint32_t cur = next() % num_nodes;
    for (uint32_t s = 0; s < steps; s++) {
      visits++;
      const int sp_before = sp;
      cur = visit_node<true, true, 0>(src, cur, r, tmax, stk, sp, top, st);
      hit_children += (uint32_t) (sp - sp_before) + (cur != kBvhEmpty ? 1u : 0u);
      if (cur == kBvhEmpty) {  // pop; an empty stack or a deep one starts over somewhere else (the walk never ends)
        if (sp > 0 && sp < 24) { cur = SE::node(top); sp--; top = stk.load(sp); }
        else { sp = 0; top = SE::make(kTraversalDone, 0.0f); cur = next() % num_nodes; }
      }
      acc += cur;
    }
  }
  else {
    unsigned long long scratch_stack[kStackSize];
    typedef __attribute__((address_space(3))) unsigned long long* LdsPtr;
    LdsPtr lds_stack = (LdsPtr) (reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(lds_top) + lds_nodes * 128u) + threadIdx.x);
    auto store = [&](int i, unsigned long long e) { if (i < 8) lds_stack[(uint32_t) i * 1024u] = e; else scratch_stack[i] = e; };
    auto load = [&](int i) { return i < 8 ? lds_stack[(uint32_t) i * 1024u] : scratch_stack[i]; };
    int sp = 0;
    Group g{0u, 0u};
    uint32_t cur = next() % num_nodes;
    for (uint32_t s = 0; s < steps; s++) {
      visits++;
      uint32_t child_base, imask;
      uint32_t hits;
      if (cur < lds_nodes) { hits = visit8(reinterpret_cast<const char*>(lds_top), cur, r, oct, tmax, child_base, imask); st.lds_nodes++; }
      else hits = visit8(table, cur, r, oct, tmax, child_base, imask);
      hit_children += (uint32_t) __builtin_popcount(hits);
      // (a real traversal splits the mask by imask into inner children and leaves; the walk treats every slot as an inner node)
      if (g.bits & 0xFFu) { store(sp, (unsigned long long) g.base | ((unsigned long long) g.bits << 32)); sp++; }  // what is left of the previous group waits
      g = Group{child_base, hits | (imask << 8)};
      if ((g.bits & 0xFFu) == 0u) {  // nothing hit: the newest group with something left
        if (sp > 0 && sp < 24) { sp--; const unsigned long long e = load(sp); g = Group{(uint32_t) e, (uint32_t) (e >> 32)}; }
        else { sp = 0; g = Group{next() % num_nodes, 0x01u | (0xFFu << 8)}; }
      }
      // nearest slot of the group: lowest priority bit -> slot -> the child's index = base + rank of the slot among the inner slots
      const uint32_t rbit = (uint32_t) __builtin_ctz(g.bits & 0xFFu);
      g.bits &= ~(1u << rbit);
      const uint32_t slot = rbit ^ oct;
      const uint32_t im = (g.bits >> 8) & 0xFFu;
      cur = (g.base + (uint32_t) __builtin_popcount(im & ((1u << slot) - 1u))) % num_nodes;
      acc += cur;
    }
  }
  if (out) { atomicAdd(out, visits); atomicAdd(out + 1, hit_children); if (acc == 0x123456789ull) out[2] = acc + st.lds_nodes; }
}

int main(int argc, char** argv) {
  uint32_t steps = 3000;
  float ext4 = 1.0f, ext8 = 1.0f;  // scale of the synthetic child boxes: sets how many children a visit finds (printed as hits_per_visit)
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = (uint32_t) atoi(argv[++i]);
    else if (!strcmp(argv[i], "--ext4") && i + 1 < argc) ext4 = (float) atof(argv[++i]);
    else if (!strcmp(argv[i], "--ext8") && i + 1 < argc) ext8 = (float) atof(argv[++i]);
  }
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const uint32_t max_nodes = 1u << 20;  // 128 MiB: the hall's working set (Infinity Cache)
  // synthetic nodes: child boxes are random sub-boxes of the unit cube, sized so that a random ray through the cube meets ~1.4 of 4 / ~2 of 8 of them
  std::vector<Bvh4Node> n4(max_nodes);
  std::vector<Node8> n8(max_nodes);
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  for (uint32_t i = 0; i < max_nodes; i++) {
    Bvh4Node& a = n4[i];
    for (int k = 0; k < 4; k++) {
      const float ext = ext4 * (0.18f + 0.25f * U(rng));
      const float c[3] = {U(rng), U(rng), U(rng)};
      a.lo_x[k] = c[0] - ext; a.hi_x[k] = c[0] + ext; a.lo_y[k] = c[1] - ext; a.hi_y[k] = c[1] + ext; a.lo_z[k] = c[2] - ext; a.hi_z[k] = c[2] + ext;
      a.child[k] = rng() % max_nodes;
      a.pad[k] = 0;
    }
    Node8& b = n8[i];
    std::memset(&b, 0, sizeof(b));
    b.origin[0] = b.origin[1] = b.origin[2] = -0.5f;
    b.exp[0] = b.exp[1] = b.exp[2] = 127 - 7;  // scale 2^-7: the byte range covers [-0.5, 1.5)
    b.imask = 0xFF;
    b.child_base = rng() % max_nodes;
    for (int k = 0; k < 8; k++) {
      const float ext = ext8 * (0.14f + 0.2f * U(rng));
      float c[3];
      for (int a3 = 0; a3 < 3; a3++) c[a3] = ((k >> a3) & 1) ? 0.5f + 0.5f * U(rng) : 0.5f * U(rng);  // slot k: towards octant k
      uint8_t* lo[3] = {b.lo_x, b.lo_y, b.lo_z};
      uint8_t* hi[3] = {b.hi_x, b.hi_y, b.hi_z};
      for (int a3 = 0; a3 < 3; a3++) {
        lo[a3][k] = (uint8_t) std::min(255.0f, std::max(0.0f, std::floor((c[a3] - ext + 0.5f) * 128.0f)));
        hi[a3][k] = (uint8_t) std::min(255.0f, std::max(0.0f, std::ceil((c[a3] + ext + 0.5f) * 128.0f)));
      }
    }
  }
  char *d4, *d8;
  unsigned long long* out;
  hipMalloc(&d4, (size_t) max_nodes * 128); hipMalloc(&d8, (size_t) max_nodes * 128); hipMalloc(&out, 32);
  hipMemcpy(d4, n4.data(), (size_t) max_nodes * 128, hipMemcpyHostToDevice);
  hipMemcpy(d8, n8.data(), (size_t) max_nodes * 128, hipMemcpyHostToDevice);
  const uint32_t lds_nodes = 512;  // 64 KB of staged nodes + 64 KB of stack, as the ray kernels
  const size_t lds = (size_t) lds_nodes * 128 + 65536;
  hipFuncSetAttribute((const void*) k_walk<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
  hipFuncSetAttribute((const void*) k_walk<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
  printf("# %s, %d CUs, 1 workgroup of 1024 per CU, %u steps per ray\n", prop.name, cus, steps);
  for (uint32_t nodes : {512u /* all staged in LDS */, 1u << 14 /* 2 MiB: L2 */, 1u << 20 /* 128 MiB: Infinity Cache */}) {
    for (uint32_t active : {64u, 32u}) {
      for (int kind : {4, 8}) {
        for (int rep = 0; rep < 2; rep++) {
          hipMemset(out, 0, 32);
          hipEvent_t a, b;
          hipEventCreate(&a); hipEventCreate(&b);
          hipEventRecord(a);
          if (kind == 4) hipLaunchKernelGGL(k_walk<4>, dim3(cus), dim3(1024), lds, 0, d4, nodes, steps, active, lds_nodes, out);
          else hipLaunchKernelGGL(k_walk<8>, dim3(cus), dim3(1024), lds, 0, d8, nodes, steps, active, lds_nodes, out);
          hipEventRecord(b);
          hipEventSynchronize(b);
          float ms = 0;
          hipEventElapsedTime(&ms, a, b);
          unsigned long long v[2] = {0, 0};
          hipMemcpy(v, out, 16, hipMemcpyDeviceToHost);
          if (rep == 1) printf("{\"node\": \"%s\", \"table_kib\": %.0f, \"staged_in_lds\": %s, \"active_per_wave\": %u, \"ms\": %.3f, \"gvisits_per_s\": %.2f, \"hits_per_visit\": %.2f}\n",
                               kind == 4 ? "4-wide float (visit_node)" : "8-wide octant (prototype)", nodes * 128.0 / 1024.0, nodes <= lds_nodes ? "true" : "false", active, ms,
                               v[0] / (ms * 1e-3) / 1e9, (double) v[1] / (double) v[0]);
          hipEventDestroy(a); hipEventDestroy(b);
        }
      }
    }
  }
  return 0;
}
