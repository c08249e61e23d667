// GPU LBVH builder: Morton-ordered binary radix tree (Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d
// Trees", HPG 2012) collapsed into the 128-byte 4-wide nodes the ray kernels walk.
// It replaces what the reference gets from optixAccelBuild (src/luminary/device/optix_bvh.c:150-684) when build time matters more than
// tree quality (scene edits, very large meshes): the binned-SAH host builder (bvh_build.cpp) stays the default because its trees trace
// faster. Any valid tree gives the same image: hits are resolved by (t, instance, triangle), never by traversal order.
//
//   1. k_lbvh_codes       63-bit Morton code of every primitive's box centre inside the mesh bounds
//   2. hipcub radix sort  (code, primitive) pairs
//   3. k_lbvh_hierarchy   one thread per internal node: its key range and split from longest common prefixes (ties: index bits)
//   4. k_lbvh_fit         leaves walk up, the second arrival at a node merges the child boxes
//   5. k_lbvh_collapse    breadth-first, one thread per 4-wide node: expands the binary children by surface area until four, subtrees
//                         of at most `max_leaf` primitives become leaves (their primitives are consecutive in Morton order)
// The result comes back to the host in the same form as the SAH builder's (nodes in breadth-first order, primitive order), so the
// scene assembly in core.hip does not care which builder ran.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cfloat>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#include "bvh_build.h"

namespace lum {
namespace {

struct BinBox { float lo[3], hi[3]; };

__device__ __forceinline__ uint64_t expand21(uint32_t v) {  // 21 bits -> every third bit of 63
  uint64_t x = v & 0x1FFFFFull;
  x = (x | (x << 32)) & 0x1F00000000FFFFull;
  x = (x | (x << 16)) & 0x1F0000FF0000FFull;
  x = (x | (x << 8)) & 0x100F00F00F00F00Full;
  x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

__global__ void k_lbvh_codes(const BinBox* __restrict__ boxes, uint32_t n, BinBox bounds, uint64_t* __restrict__ codes, uint32_t* __restrict__ ids) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t q[3];
  for (int a = 0; a < 3; a++) {
    const float ext = bounds.hi[a] - bounds.lo[a];
    const float c = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
    float u = (ext > 0.0f) ? (c - bounds.lo[a]) / ext : 0.0f;
    u = fminf(fmaxf(u, 0.0f), 1.0f);
    q[a] = min((uint32_t) (u * 2097152.0f), 2097151u);
  }
  codes[i] = (expand21(q[0]) << 2) | (expand21(q[1]) << 1) | expand21(q[2]);
  ids[i] = i;
}

// Length of the common prefix of keys i and j; equal codes fall back to the index bits, out-of-range j gives -1.
__device__ __forceinline__ int lcp(const uint64_t* __restrict__ codes, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  const uint64_t a = codes[i], b = codes[j];
  if (a != b) return __clzll((long long) (a ^ b));
  return 64 + __clz(i ^ j);
}

// Binary nodes: internal node i in [0, n-1), leaf k is node (n - 1) + k. Node 0 is the root.
__global__ void k_lbvh_hierarchy(const uint64_t* __restrict__ codes, int n, int2* __restrict__ children, int2* __restrict__ ranges, int* __restrict__ parent) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const int d = (lcp(codes, n, i, i + 1) - lcp(codes, n, i, i - 1)) >= 0 ? 1 : -1;
  const int delta_min = lcp(codes, n, i, i - d);
  int lmax = 2;
  while (lcp(codes, n, i, i + lmax * d) > delta_min) lmax <<= 1;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (lcp(codes, n, i, i + (l + t) * d) > delta_min) l += t;
  const int j = i + l * d;
  const int delta_node = lcp(codes, n, i, j);
  int s = 0, t = l;
  do {
    t = (t + 1) >> 1;
    if (lcp(codes, n, i, i + (s + t) * d) > delta_node) s += t;
  } while (t > 1);
  const int gamma = i + s * d + min(d, 0);
  const int first = min(i, j), last = max(i, j);
  const int left = (first == gamma) ? (n - 1) + gamma : gamma;
  const int right = (last == gamma + 1) ? (n - 1) + gamma + 1 : gamma + 1;
  children[i] = make_int2(left, right);
  ranges[i] = make_int2(first, last);
  parent[left] = i;
  parent[right] = i;
  if (i == 0) parent[0] = -1;
}

__global__ void k_lbvh_fit(const BinBox* __restrict__ boxes, const uint32_t* __restrict__ ids, int n, const int2* __restrict__ children, const int* __restrict__ parent,
                           BinBox* __restrict__ node_box, uint32_t* __restrict__ arrivals, int2* __restrict__ ranges) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  node_box[(n - 1) + k] = boxes[ids[k]];
  ranges[(n - 1) + k] = make_int2(k, k);
  __threadfence();
  int node = parent[(n - 1) + k];
  while (node >= 0) {
    if (atomicAdd(&arrivals[node], 1u) == 0u) return;  // the sibling subtree is not finished yet: its last thread continues
    __threadfence();
    const int2 c = children[node];
    // the other child's box was written by another thread: read it past the caches of this CU
    BinBox a, b;
    for (int x = 0; x < 3; x++) {
      a.lo[x] = __builtin_nontemporal_load(&node_box[c.x].lo[x]); a.hi[x] = __builtin_nontemporal_load(&node_box[c.x].hi[x]);
      b.lo[x] = __builtin_nontemporal_load(&node_box[c.y].lo[x]); b.hi[x] = __builtin_nontemporal_load(&node_box[c.y].hi[x]);
    }
    BinBox m;
    for (int x = 0; x < 3; x++) { m.lo[x] = fminf(a.lo[x], b.lo[x]); m.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
    node_box[node] = m;
    __threadfence();
    node = parent[node];
  }
}

__device__ __forceinline__ float half_area(const BinBox& b) {
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  return dx * dy + dy * dz + dz * dx;
}

struct CollapseItem { int bin; uint32_t node4; };

// ---- which binary nodes become 4-wide nodes: the optimal cut (bvh_build.cpp CollapsePlan, the same expressions in the same order: the host builder's trees) ----
// F[b] = (the cheapest way to hand b's subtree to a parent that has 1, 2, 3, 4 slots for it); leaves (at most max_leaf primitives): 0. A node's values need its
// children's, so the plan is made in passes over all nodes: pass p finishes the nodes whose children were finished before p (as many passes as the binary tree is
// high, each a few microseconds); `done_pass` says in which pass a node was finished, so that a value written in this pass is never read in it.
constexpr int kPlanPending = 0x7FFFFFFF;
__global__ void k_plan_init(int total, const int2* __restrict__ ranges, uint32_t max_leaf, int* __restrict__ done_pass, float4* __restrict__ F) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= total) return;
  const bool leaf = (uint32_t) (ranges[b].y - ranges[b].x + 1) <= max_leaf;
  done_pass[b] = leaf ? 0 : kPlanPending;
  F[b] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
__global__ void k_plan_pass(int total, const int2* __restrict__ children, const BinBox* __restrict__ node_box, int pass, int* __restrict__ done_pass, float4* __restrict__ F) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= total || done_pass[b] != kPlanPending) return;
  const int2 c = children[b];
  if (done_pass[c.x] >= pass || done_pass[c.y] >= pass) return;
  const float4 l4 = F[c.x], r4 = F[c.y];
  const float L[4] = {l4.x, l4.y, l4.z, l4.w}, R[4] = {r4.x, r4.y, r4.z, r4.w};
  float G[5] = {0.0f, 0.0f, FLT_MAX, FLT_MAX, FLT_MAX};
  for (int k = 2; k <= 4; k++)
    for (int a = 1; a < k; a++) G[k] = fminf(G[k], L[a - 1] + R[k - a - 1]);
  const float f0 = half_area(node_box[b]) + G[4];
  F[b] = make_float4(f0, fminf(f0, G[2]), fminf(f0, G[3]), fminf(f0, G[4]));
  done_pass[b] = pass;
}
__device__ __forceinline__ float plan_f(const float4* __restrict__ F, int b, int k) { const float4 f = F[b]; return k == 1 ? f.x : k == 2 ? f.y : k == 3 ? f.z : f.w; }

// One 4-wide node per queue entry. Children that stay inner nodes get consecutive new indices and go to the next level's queue.
// F == nullptr: the greedy rule of rounds 1-5 (the child with the largest box is opened until four are reached; LUM_BVH_COLLAPSE=0).
__global__ void k_lbvh_collapse(int n, const int2* __restrict__ children, const int2* __restrict__ ranges, const BinBox* __restrict__ node_box, uint32_t max_leaf,
                                const CollapseItem* __restrict__ in, uint32_t in_count, CollapseItem* __restrict__ out, uint32_t* __restrict__ out_count,
                                uint32_t* __restrict__ node_count, Bvh4Node* __restrict__ nodes, const float4* __restrict__ F) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= in_count) return;
  const CollapseItem item = in[q];
  auto prim_count = [&](int b) -> uint32_t { return (uint32_t) (ranges[b].y - ranges[b].x + 1); };  // ranges cover the leaves too (2n - 1 entries)
  auto is_leaf = [&](int b) -> bool { return prim_count(b) <= max_leaf; };
  int kids[4];
  int nk = 0;
  if (is_leaf(item.bin)) kids[nk++] = item.bin;  // whole mesh fits one leaf: root with a single leaf child
  else if (F) {
    // CollapsePlan::split / cut without recursion: (node, slots, must be opened), left pieces before right ones
    int st_node[4], st_k[4]; bool st_open[4];
    int sp = 0;
    st_node[sp] = item.bin; st_k[sp] = 4; st_open[sp] = true; sp++;
    while (sp > 0) {
      sp--;
      const int b = st_node[sp], k = st_k[sp];
      const bool open = st_open[sp];
      if (!open && (is_leaf(b) || k == 1 || plan_f(F, b, k) == plan_f(F, b, 1))) { kids[nk++] = b; continue; }
      const int2 c = children[b];
      int best_a = 1; float best = FLT_MAX;
      for (int a = 1; a < k; a++) { const float cost = plan_f(F, c.x, a) + plan_f(F, c.y, k - a); if (cost < best) { best = cost; best_a = a; } }
      st_node[sp] = c.y; st_k[sp] = k - best_a; st_open[sp] = false; sp++;
      st_node[sp] = c.x; st_k[sp] = best_a; st_open[sp] = false; sp++;
    }
  }
  else {
    const int2 c = children[item.bin]; kids[nk++] = c.x; kids[nk++] = c.y;
    while (nk < 4) {
      int pick = -1;
      float best = -1.0f;
      for (int k = 0; k < nk; k++) {
        if (is_leaf(kids[k])) continue;
        const float a = half_area(node_box[kids[k]]);
        if (a > best) { best = a; pick = k; }
      }
      if (pick < 0) break;
      const int2 c2 = children[kids[pick]];
      kids[pick] = c2.x;
      kids[nk++] = c2.y;
    }
  }
  uint32_t inner = 0;
  for (int k = 0; k < nk; k++) inner += is_leaf(kids[k]) ? 0u : 1u;
  uint32_t first_new = 0, first_slot = 0;
  if (inner) { first_new = atomicAdd(node_count, inner); first_slot = atomicAdd(out_count, inner); }
  Bvh4Node node;
  for (int k = 0; k < 4; k++) {
    node.child[k] = kBvhEmpty; node.pad[k] = 0;
    node.lo_x[k] = node.lo_y[k] = node.lo_z[k] = FLT_MAX;
    node.hi_x[k] = node.hi_y[k] = node.hi_z[k] = -FLT_MAX;
  }
  uint32_t used = 0;
  for (int k = 0; k < nk; k++) {
    const BinBox b = node_box[kids[k]];
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) {  // same conservative padding as the host builder (bvh_build.cpp set_child_box)
      const float pad = 1e-5f * fmaxf(fmaxf(fabsf(b.lo[a]), fabsf(b.hi[a])), 1e-20f) + 1e-30f;
      lo[a] = b.lo[a] - pad; hi[a] = b.hi[a] + pad;
    }
    node.lo_x[k] = lo[0]; node.lo_y[k] = lo[1]; node.lo_z[k] = lo[2];
    node.hi_x[k] = hi[0]; node.hi_y[k] = hi[1]; node.hi_z[k] = hi[2];
    if (is_leaf(kids[k])) {
      const uint32_t first = (uint32_t) ranges[kids[k]].x;
      node.child[k] = kBvhLeafBit | ((prim_count(kids[k]) - 1u) << 28) | first;
    }
    else {
      node.child[k] = first_new + used;
      out[first_slot + used] = CollapseItem{kids[k], first_new + used};
      used++;
    }
  }
  nodes[item.node4] = node;
}


// ---- parallel locally-ordered clustering (Meister, Bittner: "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", TVCG
// 2018) over the same Morton order: bottom-up, every cluster looks `radius` places to either side for the neighbour whose union with it has the
// smallest box, mutual choices merge, the array is compacted, until one cluster is left. The tree is no longer tied to the Morton code's bit
// boundaries: on architectural meshes the closest-hit rays visit 13 % fewer nodes than in the radix tree (tools/bvh_quality.cpp models both), at
// two to three times its build time. Node numbering as above (inner nodes [0, n-1), leaf k = n-1+k, root 0): merge number q creates node n-2-q.
constexpr int kPlocRadius = 8;
constexpr int kPlocBlock = 256;

__global__ void k_ploc_init(const BinBox* __restrict__ boxes, const uint32_t* __restrict__ ids, int n, BinBox* __restrict__ node_box, uint32_t* __restrict__ count,
                            int* __restrict__ clusters, int* __restrict__ parent) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  node_box[(n - 1) + k] = boxes[ids[k]];
  count[(n - 1) + k] = 1u;
  clusters[k] = (n - 1) + k;
  if (k == 0) parent[0] = -1;
}

__global__ __launch_bounds__(kPlocBlock) void k_ploc_nearest(const int* __restrict__ clusters, int m, const BinBox* __restrict__ node_box, int* __restrict__ nearest) {
  __shared__ BinBox tile[kPlocBlock + 2 * kPlocRadius];
  const int base = blockIdx.x * kPlocBlock - kPlocRadius;
  for (int t = threadIdx.x; t < kPlocBlock + 2 * kPlocRadius; t += kPlocBlock) {
    const int j = base + t;
    if (j >= 0 && j < m) tile[t] = node_box[clusters[j]];
  }
  __syncthreads();
  const int i = blockIdx.x * kPlocBlock + threadIdx.x;
  if (i >= m) return;
  const BinBox a = tile[threadIdx.x + kPlocRadius];
  float best = FLT_MAX;
  int pick = -1;
  for (int d = -kPlocRadius; d <= kPlocRadius; d++) {  // ascending j: of equal areas the lower index wins, on both sides of a pair
    const int j = i + d;
    if (d == 0 || j < 0 || j >= m) continue;
    const BinBox b = tile[threadIdx.x + kPlocRadius + d];
    BinBox u;
    for (int x = 0; x < 3; x++) { u.lo[x] = fminf(a.lo[x], b.lo[x]); u.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
    const float area = half_area(u);
    if (area < best) { best = area; pick = j; }
  }
  nearest[i] = pick;
}

// x = this place opens a merged cluster (the lower index of a mutual pair), y = this place stays in the array (everything but the pair's upper index)
__global__ void k_ploc_flags(const int* __restrict__ nearest, int m, uint2* __restrict__ flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int j = nearest[i];
  const bool mutual = j >= 0 && nearest[j] == i;
  flags[i] = make_uint2((mutual && i < j) ? 1u : 0u, (mutual && i > j) ? 0u : 1u);
}
struct AddPair { __host__ __device__ uint2 operator()(const uint2& a, const uint2& b) const { return make_uint2(a.x + b.x, a.y + b.y); } };

__global__ void k_ploc_merge(const int* __restrict__ clusters, const int* __restrict__ nearest, const uint2* __restrict__ flags, const uint2* __restrict__ offsets, int m, int n,
                             uint32_t merges_before, int2* __restrict__ children, int* __restrict__ parent, BinBox* __restrict__ node_box, uint32_t* __restrict__ count,
                             int* __restrict__ clusters_out, uint2* __restrict__ totals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const uint2 f = flags[i], o = offsets[i];
  if (i == m - 1) *totals = make_uint2(o.x + f.x, o.y + f.y);
  if (!f.y) return;
  int id = clusters[i];
  if (f.x) {
    const int left = id, right = clusters[nearest[i]];
    id = (n - 2) - (int) (merges_before + o.x);
    const BinBox a = node_box[left], b = node_box[right];
    BinBox u;
    for (int x = 0; x < 3; x++) { u.lo[x] = fminf(a.lo[x], b.lo[x]); u.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
    node_box[id] = u;
    count[id] = count[left] + count[right];
    children[id] = make_int2(left, right);
    parent[left] = id; parent[right] = id;
  }
  clusters_out[o.y] = id;
}

// Every subtree's primitives as one range of the output order: a node starts where the subtrees to the left of its path from the root end.
__global__ void k_ploc_ranges(int n, const int2* __restrict__ children, const int* __restrict__ parent, const uint32_t* __restrict__ count, const uint32_t* __restrict__ ids,
                              int2* __restrict__ ranges, uint32_t* __restrict__ prims) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= 2 * n - 1) return;
  uint32_t first = 0;
  for (int x = b, p = parent[b]; p >= 0; x = p, p = parent[p])
    if (children[p].y == x) first += count[children[p].x];
  ranges[b] = make_int2((int) first, (int) (first + count[b] - 1u));
  if (b >= n - 1) prims[first] = ids[b - (n - 1)];
}

#define LBVH_TRY(expr) do { if ((expr) != hipSuccess) { ok = false; goto done; } } while (0)

bool collapse_rule_optimal() { const char* e = std::getenv("LUM_BVH_COLLAPSE"); return !e || std::atoi(e) != 0; }  // as the host builder's (bvh_build.cpp collapse)

// Fills F (see k_plan_pass) for the binary tree of `total` nodes whose root is node 0. False: a HIP call failed, or the tree is higher than any tree the builders make.
bool make_collapse_plan(int total, const int2* d_children, const int2* d_ranges, const BinBox* d_node_box, uint32_t max_leaf, int* d_done, float4* d_F) {
  const int threads = 256, blocks = (total + threads - 1) / threads;
  hipLaunchKernelGGL(k_plan_init, dim3(blocks), dim3(threads), 0, 0, total, d_ranges, max_leaf, d_done, d_F);
  for (int pass = 1; pass <= 4096; pass++) {
    hipLaunchKernelGGL(k_plan_pass, dim3(blocks), dim3(threads), 0, 0, total, d_children, d_node_box, pass, d_done, d_F);
    if ((pass & 3) == 0 || pass < 4) {  // (a look every fourth pass: the download costs as much as a pass)
      int root = kPlanPending;
      if (hipMemcpy(&root, d_done, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return false;
      if (root != kPlanPending) return true;
    }
  }
  return false;
}

}  // namespace

namespace {
Bvh4 build_on_device(const Aabb* boxes, uint32_t count, uint32_t max_leaf, uint32_t max_depth, bool ploc) {
  static_assert(sizeof(Aabb) == sizeof(BinBox), "box layouts must match");
  Bvh4 result;
  if (count < 2) return build_bvh4(boxes, count, max_leaf, max_depth);  // nothing to sort
  max_leaf = max_leaf < 1 ? 1 : (max_leaf > kBvhLeafMaxTri ? kBvhLeafMaxTri : max_leaf);
  const int n = (int) count;
  BinBox bounds{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
  for (uint32_t i = 0; i < count; i++)
    for (int a = 0; a < 3; a++) {
      const float c = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
      bounds.lo[a] = std::min(bounds.lo[a], c); bounds.hi[a] = std::max(bounds.hi[a], c);
    }
  BinBox* d_boxes = nullptr; BinBox* d_node_box = nullptr;
  uint64_t* d_codes = nullptr; uint64_t* d_codes_sorted = nullptr;
  uint32_t* d_ids = nullptr; uint32_t* d_ids_sorted = nullptr; uint32_t* d_arrivals = nullptr; uint32_t* d_counters = nullptr;
  int2* d_children = nullptr; int2* d_ranges = nullptr; int* d_parent = nullptr;
  uint32_t* d_count = nullptr; uint32_t* d_prims = nullptr; int* d_clusters[2] = {nullptr, nullptr}; int* d_nearest = nullptr;
  uint2* d_flags = nullptr; uint2* d_offsets = nullptr; uint2* d_totals = nullptr; void* d_scan_temp = nullptr;
  size_t scan_bytes = 0;
  CollapseItem* d_queue[2] = {nullptr, nullptr};
  Bvh4Node* d_nodes = nullptr;
  int* d_plan_done = nullptr; float4* d_plan = nullptr;
  const bool optimal = collapse_rule_optimal();
  void* d_temp = nullptr;
  size_t temp_bytes = 0;
  bool ok = true;
  uint32_t node_count = 1, level_count = 1, depth = 0;
  const uint32_t max_nodes = count;  // a 4-wide node has at least two children, so fewer nodes than primitives
  const int threads = 256;
  LBVH_TRY(hipMalloc((void**) &d_boxes, sizeof(BinBox) * count));
  LBVH_TRY(hipMalloc((void**) &d_node_box, sizeof(BinBox) * (2 * (size_t) count - 1)));
  LBVH_TRY(hipMalloc((void**) &d_codes, sizeof(uint64_t) * count));
  LBVH_TRY(hipMalloc((void**) &d_codes_sorted, sizeof(uint64_t) * count));
  LBVH_TRY(hipMalloc((void**) &d_ids, sizeof(uint32_t) * count));
  LBVH_TRY(hipMalloc((void**) &d_ids_sorted, sizeof(uint32_t) * count));
  LBVH_TRY(hipMalloc((void**) &d_arrivals, sizeof(uint32_t) * count));
  LBVH_TRY(hipMalloc((void**) &d_counters, sizeof(uint32_t) * 4));
  LBVH_TRY(hipMalloc((void**) &d_children, sizeof(int2) * count));
  LBVH_TRY(hipMalloc((void**) &d_ranges, sizeof(int2) * (2 * (size_t) count - 1)));
  LBVH_TRY(hipMalloc((void**) &d_parent, sizeof(int) * (2 * (size_t) count - 1)));
  LBVH_TRY(hipMalloc((void**) &d_queue[0], sizeof(CollapseItem) * count));
  LBVH_TRY(hipMalloc((void**) &d_queue[1], sizeof(CollapseItem) * count));
  LBVH_TRY(hipMalloc((void**) &d_nodes, sizeof(Bvh4Node) * max_nodes));
  LBVH_TRY(hipMemcpy(d_boxes, boxes, sizeof(BinBox) * count, hipMemcpyHostToDevice));
  LBVH_TRY(hipMemset(d_arrivals, 0, sizeof(uint32_t) * count));
  hipLaunchKernelGGL(k_lbvh_codes, dim3((count + threads - 1) / threads), dim3(threads), 0, 0, (const BinBox*) d_boxes, count, bounds, d_codes, d_ids);
  LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, temp_bytes, d_codes, d_codes_sorted, d_ids, d_ids_sorted, n, 0, 63));
  LBVH_TRY(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 16));
  LBVH_TRY(hipcub::DeviceRadixSort::SortPairs(d_temp, temp_bytes, d_codes, d_codes_sorted, d_ids, d_ids_sorted, n, 0, 63));
  if (!ploc) {
    hipLaunchKernelGGL(k_lbvh_hierarchy, dim3((count + threads - 1) / threads), dim3(threads), 0, 0, (const uint64_t*) d_codes_sorted, n, d_children, d_ranges, d_parent);
    hipLaunchKernelGGL(k_lbvh_fit, dim3((count + threads - 1) / threads), dim3(threads), 0, 0, (const BinBox*) d_boxes, (const uint32_t*) d_ids_sorted, n,
                       (const int2*) d_children, (const int*) d_parent, d_node_box, d_arrivals, d_ranges);
    LBVH_TRY(hipGetLastError());
  }
  else {
    LBVH_TRY(hipMalloc((void**) &d_count, sizeof(uint32_t) * (2 * (size_t) count - 1)));
    LBVH_TRY(hipMalloc((void**) &d_prims, sizeof(uint32_t) * count));
    LBVH_TRY(hipMalloc((void**) &d_clusters[0], sizeof(int) * count));
    LBVH_TRY(hipMalloc((void**) &d_clusters[1], sizeof(int) * count));
    LBVH_TRY(hipMalloc((void**) &d_nearest, sizeof(int) * count));
    LBVH_TRY(hipMalloc((void**) &d_flags, sizeof(uint2) * count));
    LBVH_TRY(hipMalloc((void**) &d_offsets, sizeof(uint2) * count));
    LBVH_TRY(hipMalloc((void**) &d_totals, sizeof(uint2)));
    LBVH_TRY(hipcub::DeviceScan::ExclusiveScan(nullptr, scan_bytes, d_flags, d_offsets, AddPair(), make_uint2(0u, 0u), n));
    LBVH_TRY(hipMalloc(&d_scan_temp, scan_bytes ? scan_bytes : 16));
    hipLaunchKernelGGL(k_ploc_init, dim3((count + threads - 1) / threads), dim3(threads), 0, 0, (const BinBox*) d_boxes, (const uint32_t*) d_ids_sorted, n, d_node_box, d_count,
                       d_clusters[0], d_parent);
    int m = n, cur = 0;
    uint32_t merges = 0;
    while (m > 1) {
      const uint32_t blocks = (uint32_t) (m + threads - 1) / threads;
      hipLaunchKernelGGL(k_ploc_nearest, dim3((m + kPlocBlock - 1) / kPlocBlock), dim3(kPlocBlock), 0, 0, (const int*) d_clusters[cur], m, (const BinBox*) d_node_box, d_nearest);
      hipLaunchKernelGGL(k_ploc_flags, dim3(blocks), dim3(threads), 0, 0, (const int*) d_nearest, m, d_flags);
      size_t bytes = scan_bytes;
      LBVH_TRY(hipcub::DeviceScan::ExclusiveScan(d_scan_temp, bytes, d_flags, d_offsets, AddPair(), make_uint2(0u, 0u), m));
      hipLaunchKernelGGL(k_ploc_merge, dim3(blocks), dim3(threads), 0, 0, (const int*) d_clusters[cur], (const int*) d_nearest, (const uint2*) d_flags, (const uint2*) d_offsets, m, n,
                         merges, d_children, d_parent, d_node_box, d_count, d_clusters[cur ^ 1], d_totals);
      uint2 totals;
      LBVH_TRY(hipMemcpy(&totals, d_totals, sizeof(totals), hipMemcpyDeviceToHost));
      if (totals.x == 0u || (int) totals.y >= m) { ok = false; goto done; }  // cannot happen: the pair with the smallest union is always mutual
      merges += totals.x;
      m = (int) totals.y;
      cur ^= 1;
    }
    if (merges != count - 1u) { ok = false; goto done; }
    hipLaunchKernelGGL(k_ploc_ranges, dim3((2 * count - 1 + threads - 1) / threads), dim3(threads), 0, 0, n, (const int2*) d_children, (const int*) d_parent, (const uint32_t*) d_count,
                       (const uint32_t*) d_ids_sorted, d_ranges, d_prims);
    LBVH_TRY(hipGetLastError());
  }
  {
    const CollapseItem root{0, 0u};
    LBVH_TRY(hipMemcpy(d_queue[0], &root, sizeof(root), hipMemcpyHostToDevice));
    const uint32_t init[4] = {0u, 1u, 0u, 0u};  // [0] next level's queue length, [1] nodes allocated
    LBVH_TRY(hipMemcpy(d_counters, init, sizeof(init), hipMemcpyHostToDevice));
  }
  if (optimal) {
    const int total = 2 * n - 1;
    LBVH_TRY(hipMalloc((void**) &d_plan_done, sizeof(int) * (size_t) total));
    LBVH_TRY(hipMalloc((void**) &d_plan, sizeof(float4) * (size_t) total));
    if (!make_collapse_plan(total, d_children, d_ranges, d_node_box, max_leaf, d_plan_done, d_plan)) { ok = false; goto done; }
  }
  for (int cur = 0; level_count > 0; cur ^= 1) {
    depth++;
    if (depth > max_depth) { ok = false; goto done; }  // deeper than the traversal stack allows: the caller falls back to the host builder
    hipLaunchKernelGGL(k_lbvh_collapse, dim3((level_count + threads - 1) / threads), dim3(threads), 0, 0, n, (const int2*) d_children, (const int2*) d_ranges,
                       (const BinBox*) d_node_box, max_leaf, (const CollapseItem*) d_queue[cur], level_count, d_queue[cur ^ 1], d_counters, d_counters + 1, d_nodes,
                       (const float4*) d_plan);
    uint32_t host_counters[2];
    LBVH_TRY(hipMemcpy(host_counters, d_counters, sizeof(host_counters), hipMemcpyDeviceToHost));
    level_count = host_counters[0];
    node_count = host_counters[1];
    if (node_count > max_nodes) { ok = false; goto done; }
    LBVH_TRY(hipMemset(d_counters, 0, sizeof(uint32_t)));
  }
  result.nodes.resize(node_count);
  result.prims.resize(count);
  LBVH_TRY(hipMemcpy(result.nodes.data(), d_nodes, sizeof(Bvh4Node) * node_count, hipMemcpyDeviceToHost));
  LBVH_TRY(hipMemcpy(result.prims.data(), ploc ? d_prims : d_ids_sorted, sizeof(uint32_t) * count, hipMemcpyDeviceToHost));
  result.max_depth = depth;
done:
  {
    void* bufs[] = {d_boxes, d_node_box, d_codes, d_codes_sorted, d_ids, d_ids_sorted, d_arrivals, d_counters, d_children, d_ranges, d_parent, d_queue[0], d_queue[1], d_nodes, d_temp,
                    d_count, d_prims, d_clusters[0], d_clusters[1], d_nearest, d_flags, d_offsets, d_totals, d_scan_temp, d_plan_done, d_plan};
    for (void* b : bufs) if (b) (void) hipFree(b);
  }
  if (!ok) return Bvh4();
  return result;
}
}  // namespace

// ---- binned SAH on the device: the host builder's trees (bvh_build.cpp Builder), built breadth first ----
// The host builder decides every split from 16 bins per axis over the centroid bounds of the node's primitives; nothing in that needs a serial order.
// Here every level of the binary tree is one round of four kernels over ALL primitives: (1) bin - every primitive of a node that is still to be split
// grows the boxes and counts of its three bins (atomics on order-preserving integer images of the floats: min / max are exact, so the bins hold the
// bits the host's loop produces; a block whose primitives all belong to one node - the top levels - gathers in LDS first and flushes 336 words); (2)
// split - one thread per node sweeps the bins with the host's own expressions (same operand order, no contraction: this file is compiled with the exact
// flavour's flags) and allocates the two children; (3) a prefix sum of "goes left" over the whole array and a scatter make every node's partition
// stable, as the host's; (4) bounds - the children's boxes and centroid bounds, again by atomics. One small download per level (how many nodes are
// left to split). The binary tree then goes through the LBVH's collapse kernel. For meshes without degenerate sets (all centroids of a set equal: the
// host falls back to a median split by nth_element, this builder halves the set by position) the tree, and the leaf order, are the host builder's.
namespace {

constexpr int kSahBins = 16, kSahBinWords = 7;          // per bin: box (6 ordered words) | count
constexpr int kSahNodeBinWords = 3 * kSahBins * kSahBinWords;

__host__ __device__ __forceinline__ uint32_t ord_enc(float f) { uint32_t b; std::memcpy(&b, &f, 4); return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u); }
__host__ __device__ __forceinline__ float ord_dec(uint32_t k) { const uint32_t b = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu); float f; std::memcpy(&f, &b, 4); return f; }

struct SahNode {
  uint32_t first, count;
  uint32_t box[6];    // lo xyz, hi xyz as ordered words (atomics)
  uint32_t cbox[6];   // centroid bounds
  int active_rank;    // index into this level's bins, or -1
  int split_axis, split_bin;  // chosen this level (split_axis -1: halve by position)
  uint32_t left_count;
  int left, right;
};

__device__ __forceinline__ float sah_centre(const BinBox& b, int a) { return 0.5f * (b.lo[a] + b.hi[a]); }
__device__ __forceinline__ int sah_bin_of(const BinBox& b, int a, float clo, float chi) {
  const float ext = chi - clo;
  const float scale = ext > 0.0f ? (float) kSahBins / ext : 0.0f;
  int bin = (int) ((sah_centre(b, a) - clo) * scale);
  return min(max(bin, 0), kSahBins - 1);
}

// box and centroid bounds of the nodes created this level (`fresh` marks them), from the primitives in their new order
__global__ __launch_bounds__(256) void k_sah_bounds(const BinBox* __restrict__ boxes, const int* __restrict__ owner, uint32_t n, SahNode* __restrict__ nodes, const uint8_t* __restrict__ fresh) {
  __shared__ uint32_t acc[12];
  __shared__ int uniform_node;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lo_i = blockIdx.x * 256u, hi_i = min(lo_i + 255u, n - 1u);
  if (threadIdx.x == 0) uniform_node = (owner[lo_i] == owner[hi_i]) ? owner[lo_i] : -1;  // positions of a node are contiguous: equal ends = one node
  if (threadIdx.x < 12) acc[threadIdx.x] = threadIdx.x % 6 < 3 ? ord_enc(FLT_MAX) : ord_enc(-FLT_MAX);
  __syncthreads();
  const int un = uniform_node;
  if (i < n) {
    const int node = owner[i];
    if (fresh[node]) {
      const BinBox b = boxes[i];
      uint32_t* box = un >= 0 ? acc : nodes[node].box;
      uint32_t* cbox = un >= 0 ? acc + 6 : nodes[node].cbox;
      for (int a = 0; a < 3; a++) {
        const uint32_t c = ord_enc(sah_centre(b, a));
        atomicMin(&box[a], ord_enc(b.lo[a])); atomicMax(&box[3 + a], ord_enc(b.hi[a]));
        atomicMin(&cbox[a], c); atomicMax(&cbox[3 + a], c);
      }
    }
  }
  __syncthreads();
  if (un >= 0 && fresh[un] && threadIdx.x < 12) {
    uint32_t* dst = threadIdx.x < 6 ? &nodes[un].box[threadIdx.x] : &nodes[un].cbox[threadIdx.x - 6];
    if (threadIdx.x % 6 < 3) atomicMin(dst, acc[threadIdx.x]); else atomicMax(dst, acc[threadIdx.x]);
  }
}

__global__ void k_sah_set_counters(uint32_t* counters, uint32_t a, uint32_t b) { counters[0] = a; counters[1] = b; }
__global__ void k_sah_clear_bins(uint32_t* __restrict__ bins, size_t words) {
  for (size_t w = blockIdx.x * (size_t) blockDim.x + threadIdx.x; w < words; w += (size_t) gridDim.x * blockDim.x) {
    const int f = (int) (w % kSahBinWords);
    bins[w] = f < 3 ? ord_enc(FLT_MAX) : f < 6 ? ord_enc(-FLT_MAX) : 0u;
  }
}

__global__ __launch_bounds__(256) void k_sah_bin(const BinBox* __restrict__ boxes, const int* __restrict__ owner, uint32_t n, const SahNode* __restrict__ nodes, uint32_t* __restrict__ bins) {
  __shared__ uint32_t acc[kSahNodeBinWords];
  __shared__ int uniform_node;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lo_i = blockIdx.x * 256u, hi_i = min(lo_i + 255u, n - 1u);
  if (threadIdx.x == 0) uniform_node = (owner[lo_i] == owner[hi_i]) ? owner[lo_i] : -1;
  __syncthreads();
  const int un = uniform_node;
  const bool local = un >= 0 && nodes[un].active_rank >= 0;
  if (local) for (int w = threadIdx.x; w < kSahNodeBinWords; w += 256) { const int f = w % kSahBinWords; acc[w] = f < 3 ? ord_enc(FLT_MAX) : f < 6 ? ord_enc(-FLT_MAX) : 0u; }
  __syncthreads();
  if (i < n) {
    const int node = owner[i];
    const int rank = nodes[node].active_rank;
    if (rank >= 0) {
      const BinBox b = boxes[i];
      uint32_t* base = local ? acc : bins + (size_t) rank * kSahNodeBinWords;
      for (int a = 0; a < 3; a++) {
        const float clo = ord_dec(nodes[node].cbox[a]), chi = ord_dec(nodes[node].cbox[3 + a]);
        if (!(chi - clo > 0.0f)) continue;  // the host skips an axis without extent (scale 0)
        uint32_t* w = base + (a * kSahBins + sah_bin_of(b, a, clo, chi)) * kSahBinWords;
        for (int k = 0; k < 3; k++) { atomicMin(&w[k], ord_enc(b.lo[k])); atomicMax(&w[3 + k], ord_enc(b.hi[k])); }
        atomicAdd(&w[6], 1u);
      }
    }
  }
  __syncthreads();
  if (local) {
    uint32_t* dst = bins + (size_t) nodes[un].active_rank * kSahNodeBinWords;
    for (int w = threadIdx.x; w < kSahNodeBinWords; w += 256) {
      const int f = w % kSahBinWords;
      if (f < 3) atomicMin(&dst[w], acc[w]); else if (f < 6) atomicMax(&dst[w], acc[w]); else if (acc[w]) atomicAdd(&dst[w], acc[w]);
    }
  }
}

__device__ __forceinline__ void bin_grow(BinBox& a, const uint32_t* w) {
  for (int k = 0; k < 3; k++) { a.lo[k] = fminf(a.lo[k], ord_dec(w[k])); a.hi[k] = fmaxf(a.hi[k], ord_dec(w[3 + k])); }
}
__device__ __forceinline__ float sah_half_area(const BinBox& b) {
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  if (dx < 0.0f) return 0.0f;
  return dx * dy + dy * dz + dz * dx;
}

// One thread per node to be split: the host's sweep over its bins (bvh_build.cpp Builder::split), the children, who of them is split next level.
__global__ void k_sah_split(const int* __restrict__ active, uint32_t num_active, const uint32_t* __restrict__ bins, SahNode* __restrict__ nodes, uint32_t max_leaf,
                            uint32_t* __restrict__ counters /* [0] nodes allocated, [1] next level's active count */, int* __restrict__ next_active, uint8_t* __restrict__ fresh,
                            int2* __restrict__ children, int2* __restrict__ ranges) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= num_active) return;
  const int id = active[q];
  SahNode nd = nodes[id];
  const uint32_t* my = bins + (size_t) q * kSahNodeBinWords;
  float best_cost = FLT_MAX;
  int best_axis = -1, best_bin = -1;
  uint32_t best_left = 0;
  for (int a = 0; a < 3; a++) {
    const float clo = ord_dec(nd.cbox[a]), chi = ord_dec(nd.cbox[3 + a]);
    if (!(chi - clo > 0.0f)) continue;
    float right_area[kSahBins];
    uint32_t right_cnt[kSahBins];
    BinBox acc{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    uint32_t c = 0;
    for (int b = kSahBins - 1; b > 0; b--) { const uint32_t* w = my + (a * kSahBins + b) * kSahBinWords; bin_grow(acc, w); c += w[6]; right_area[b] = sah_half_area(acc); right_cnt[b] = c; }
    acc = BinBox{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}}; c = 0;
    for (int b = 0; b < kSahBins - 1; b++) {
      const uint32_t* w = my + (a * kSahBins + b) * kSahBinWords;
      bin_grow(acc, w); c += w[6];
      if (c == 0 || right_cnt[b + 1] == 0) continue;
      const float cost = sah_half_area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
      if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; best_left = c; }
    }
  }
  if (best_axis < 0) { best_bin = -1; best_left = nd.count / 2; }  // no axis separates the centroids: halve the set where it stands
  const int l = (int) atomicAdd(&counters[0], 2u), r = l + 1;
  nd.split_axis = best_axis; nd.split_bin = best_bin; nd.left_count = best_left; nd.left = l; nd.right = r;
  nodes[id] = nd;
  children[id] = make_int2(l, r);
  for (int side = 0; side < 2; side++) {
    SahNode ch;
    ch.first = side == 0 ? nd.first : nd.first + best_left;
    ch.count = side == 0 ? best_left : nd.count - best_left;
    for (int k = 0; k < 3; k++) { ch.box[k] = ch.cbox[k] = ord_enc(FLT_MAX); ch.box[3 + k] = ch.cbox[3 + k] = ord_enc(-FLT_MAX); }
    ch.split_axis = -1; ch.split_bin = -1; ch.left_count = 0; ch.left = ch.right = -1;
    ch.active_rank = -1;
    const int cid = side == 0 ? l : r;
    if (ch.count > max_leaf) { ch.active_rank = (int) atomicAdd(&counters[1], 1u); next_active[ch.active_rank] = cid; }
    nodes[cid] = ch;
    fresh[cid] = 1;
    ranges[cid] = make_int2((int) ch.first, (int) (ch.first + ch.count - 1u));
  }
}

__global__ void k_sah_flags(const BinBox* __restrict__ boxes, const int* __restrict__ owner, uint32_t n, const SahNode* __restrict__ nodes, uint32_t* __restrict__ flags) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const SahNode& nd = nodes[owner[i]];
  uint32_t f = 0;
  if (nd.left >= 0) {  // split this level
    if (nd.split_axis < 0) f = (i - nd.first) < nd.left_count ? 1u : 0u;
    else f = sah_bin_of(boxes[i], nd.split_axis, ord_dec(nd.cbox[nd.split_axis]), ord_dec(nd.cbox[3 + nd.split_axis])) <= nd.split_bin ? 1u : 0u;
  }
  flags[i] = f;
}
__global__ void k_sah_scatter(const BinBox* __restrict__ boxes, const uint32_t* __restrict__ ids, const int* __restrict__ owner, uint32_t n, const SahNode* __restrict__ nodes,
                              const uint32_t* __restrict__ flags, const uint32_t* __restrict__ offsets, BinBox* __restrict__ boxes_out, uint32_t* __restrict__ ids_out,
                              int* __restrict__ owner_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int node = owner[i];
  const SahNode& nd = nodes[node];
  uint32_t pos = i;
  int own = node;
  if (nd.left >= 0) {
    const uint32_t lefts_before = offsets[i] - offsets[nd.first];  // exclusive prefix of "goes left" inside the node
    if (flags[i]) { pos = nd.first + lefts_before; own = nd.left; }
    else { pos = nd.first + nd.left_count + ((i - nd.first) - lefts_before); own = nd.right; }
  }
  boxes_out[pos] = boxes[i]; ids_out[pos] = ids[i]; owner_out[pos] = own;
}
// what was split this level is done with; the collapse kernel wants the boxes as floats
__global__ void k_sah_finish_level(const int* __restrict__ active, uint32_t num_active, SahNode* __restrict__ nodes) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < num_active) { SahNode& nd = nodes[active[q]]; nd.active_rank = -1; nd.left = -1; }
}
__global__ void k_sah_export(const SahNode* __restrict__ nodes, uint32_t count, BinBox* __restrict__ node_box) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  BinBox b;
  for (int k = 0; k < 3; k++) { b.lo[k] = ord_dec(nodes[i].box[k]); b.hi[k] = ord_dec(nodes[i].box[3 + k]); }
  node_box[i] = b;
}

}  // namespace

Bvh4 build_bvh4_sah_gpu(const Aabb* boxes, uint32_t count, uint32_t max_leaf, uint32_t max_depth) {
  Bvh4 result;
  if (count < 2) return build_bvh4(boxes, count, max_leaf, max_depth);
  max_leaf = max_leaf < 1 ? 1 : (max_leaf > kBvhLeafMaxTri ? kBvhLeafMaxTri : max_leaf);
  const uint32_t n = count;
  const size_t max_nodes2 = 2 * (size_t) count + 2;  // binary nodes
  const size_t max_active = (size_t) count / (max_leaf + 1u) + 2;
  BinBox* d_boxes[2] = {nullptr, nullptr}; uint32_t* d_ids[2] = {nullptr, nullptr}; int* d_owner[2] = {nullptr, nullptr};
  SahNode* d_nodes2 = nullptr; uint32_t* d_bins = nullptr; int* d_active[2] = {nullptr, nullptr}; uint8_t* d_fresh = nullptr;
  uint32_t* d_flags = nullptr; uint32_t* d_offsets = nullptr; uint32_t* d_counters = nullptr; void* d_scan_temp = nullptr;
  int2* d_children = nullptr; int2* d_ranges = nullptr; BinBox* d_node_box = nullptr;
  CollapseItem* d_queue[2] = {nullptr, nullptr}; Bvh4Node* d_nodes = nullptr;
  int* d_plan_done = nullptr; float4* d_plan = nullptr;
  const bool optimal = collapse_rule_optimal();
  char* d_pool = nullptr;
  size_t scan_bytes = 0;
  bool ok = true;
  const int threads = 256;
  const uint32_t blocks_n = (n + threads - 1) / threads;
  uint32_t num_active = 1, nodes_allocated = 1, levels = 0;
  int cur = 0, acur = 0;
  uint32_t node_count = 1, level_count = 1, depth = 0;
  const uint32_t max_nodes4 = count;
  {  // one allocation for everything (a dozen hipMallocs of hundreds of megabytes cost more than a level of the build)
    LBVH_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, d_flags, d_offsets, (int) n));
    size_t total = 0;
    auto reserve = [&](size_t bytes) { const size_t at = total; total += (bytes + 255) & ~(size_t) 255; return at; };
    const size_t o_boxes[2] = {reserve(sizeof(BinBox) * n), reserve(sizeof(BinBox) * n)}, o_ids[2] = {reserve(sizeof(uint32_t) * n), reserve(sizeof(uint32_t) * n)};
    const size_t o_owner[2] = {reserve(sizeof(int) * n), reserve(sizeof(int) * n)}, o_active[2] = {reserve(sizeof(int) * max_active), reserve(sizeof(int) * max_active)};
    const size_t o_queue[2] = {reserve(sizeof(CollapseItem) * count), reserve(sizeof(CollapseItem) * count)};
    const size_t o_nodes2 = reserve(sizeof(SahNode) * max_nodes2), o_bins = reserve(sizeof(uint32_t) * kSahNodeBinWords * max_active), o_fresh = reserve(max_nodes2);
    const size_t o_flags = reserve(sizeof(uint32_t) * n), o_offsets = reserve(sizeof(uint32_t) * n), o_counters = reserve(sizeof(uint32_t) * 4);
    const size_t o_children = reserve(sizeof(int2) * max_nodes2), o_ranges = reserve(sizeof(int2) * max_nodes2), o_node_box = reserve(sizeof(BinBox) * max_nodes2);
    const size_t o_nodes = reserve(sizeof(Bvh4Node) * max_nodes4), o_scan = reserve(scan_bytes ? scan_bytes : 16);
    const size_t o_plan_done = reserve(optimal ? sizeof(int) * max_nodes2 : 0), o_plan = reserve(optimal ? sizeof(float4) * max_nodes2 : 0);
    LBVH_TRY(hipMalloc((void**) &d_pool, total));
    if (optimal) { d_plan_done = (int*) (d_pool + o_plan_done); d_plan = (float4*) (d_pool + o_plan); }
    for (int k = 0; k < 2; k++) {
      d_boxes[k] = (BinBox*) (d_pool + o_boxes[k]); d_ids[k] = (uint32_t*) (d_pool + o_ids[k]); d_owner[k] = (int*) (d_pool + o_owner[k]);
      d_active[k] = (int*) (d_pool + o_active[k]); d_queue[k] = (CollapseItem*) (d_pool + o_queue[k]);
    }
    d_nodes2 = (SahNode*) (d_pool + o_nodes2); d_bins = (uint32_t*) (d_pool + o_bins); d_fresh = (uint8_t*) (d_pool + o_fresh);
    d_flags = (uint32_t*) (d_pool + o_flags); d_offsets = (uint32_t*) (d_pool + o_offsets); d_counters = (uint32_t*) (d_pool + o_counters);
    d_children = (int2*) (d_pool + o_children); d_ranges = (int2*) (d_pool + o_ranges); d_node_box = (BinBox*) (d_pool + o_node_box);
    d_nodes = (Bvh4Node*) (d_pool + o_nodes); d_scan_temp = (void*) (d_pool + o_scan);
  }
  LBVH_TRY(hipMemcpy(d_boxes[0], boxes, sizeof(BinBox) * n, hipMemcpyHostToDevice));
  {
    std::vector<uint32_t> ids(n);
    for (uint32_t i = 0; i < n; i++) ids[i] = i;
    LBVH_TRY(hipMemcpy(d_ids[0], ids.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    LBVH_TRY(hipMemset(d_owner[0], 0, sizeof(int) * n));
    SahNode root;
    root.first = 0; root.count = n;
    for (int k = 0; k < 3; k++) { root.box[k] = root.cbox[k] = ord_enc(FLT_MAX); root.box[3 + k] = root.cbox[3 + k] = ord_enc(-FLT_MAX); }
    root.active_rank = n > max_leaf ? 0 : -1; root.split_axis = -1; root.split_bin = -1; root.left_count = 0; root.left = root.right = -1;
    LBVH_TRY(hipMemcpy(d_nodes2, &root, sizeof(root), hipMemcpyHostToDevice));
    const int2 r0 = make_int2(0, (int) n - 1);
    LBVH_TRY(hipMemcpy(d_ranges, &r0, sizeof(r0), hipMemcpyHostToDevice));
    const int zero = 0;
    LBVH_TRY(hipMemcpy(d_active[0], &zero, sizeof(int), hipMemcpyHostToDevice));
    LBVH_TRY(hipMemset(d_fresh, 0, max_nodes2));
    const uint8_t one = 1;
    LBVH_TRY(hipMemcpy(d_fresh, &one, 1, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_sah_bounds, dim3(blocks_n), dim3(threads), 0, 0, (const BinBox*) d_boxes[0], (const int*) d_owner[0], n, d_nodes2, (const uint8_t*) d_fresh);
    LBVH_TRY(hipMemset(d_fresh, 0, 1));
    num_active = n > max_leaf ? 1u : 0u;
  }
  while (num_active > 0) {
    if (++levels > 96u) { ok = false; goto done; }
    {  // empty bins for this level's nodes: lo = +max, hi = -max, count 0
      const size_t words = (size_t) num_active * kSahNodeBinWords;
      hipLaunchKernelGGL(k_sah_clear_bins, dim3((uint32_t) std::min<size_t>((words + 255) / 256, 65535)), dim3(256), 0, 0, d_bins, words);
    }
    hipLaunchKernelGGL(k_sah_bin, dim3(blocks_n), dim3(threads), 0, 0, (const BinBox*) d_boxes[cur], (const int*) d_owner[cur], n, (const SahNode*) d_nodes2, d_bins);
    hipLaunchKernelGGL(k_sah_set_counters, dim3(1), dim3(1), 0, 0, d_counters, nodes_allocated, 0u);
    hipLaunchKernelGGL(k_sah_split, dim3((num_active + threads - 1) / threads), dim3(threads), 0, 0, (const int*) d_active[acur], num_active, (const uint32_t*) d_bins, d_nodes2, max_leaf,
                       d_counters, d_active[acur ^ 1], d_fresh, d_children, d_ranges);
    hipLaunchKernelGGL(k_sah_flags, dim3(blocks_n), dim3(threads), 0, 0, (const BinBox*) d_boxes[cur], (const int*) d_owner[cur], n, (const SahNode*) d_nodes2, d_flags);
    {
      size_t bytes = scan_bytes;
      LBVH_TRY(hipcub::DeviceScan::ExclusiveSum(d_scan_temp, bytes, d_flags, d_offsets, (int) n));
    }
    hipLaunchKernelGGL(k_sah_scatter, dim3(blocks_n), dim3(threads), 0, 0, (const BinBox*) d_boxes[cur], (const uint32_t*) d_ids[cur], (const int*) d_owner[cur], n,
                       (const SahNode*) d_nodes2, (const uint32_t*) d_flags, (const uint32_t*) d_offsets, d_boxes[cur ^ 1], d_ids[cur ^ 1], d_owner[cur ^ 1]);
    hipLaunchKernelGGL(k_sah_finish_level, dim3((num_active + threads - 1) / threads), dim3(threads), 0, 0, (const int*) d_active[acur], num_active, d_nodes2);
    cur ^= 1;
    hipLaunchKernelGGL(k_sah_bounds, dim3(blocks_n), dim3(threads), 0, 0, (const BinBox*) d_boxes[cur], (const int*) d_owner[cur], n, d_nodes2, (const uint8_t*) d_fresh);
    uint32_t host_counters[2];
    LBVH_TRY(hipMemcpy(host_counters, d_counters, sizeof(host_counters), hipMemcpyDeviceToHost));
    LBVH_TRY(hipMemset(d_fresh + nodes_allocated, 0, host_counters[0] - nodes_allocated));
    nodes_allocated = host_counters[0];
    num_active = host_counters[1];
    acur ^= 1;
    if (nodes_allocated + 2u > max_nodes2 || num_active > max_active) { ok = false; goto done; }
  }
  LBVH_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_sah_export, dim3((nodes_allocated + threads - 1) / threads), dim3(threads), 0, 0, (const SahNode*) d_nodes2, nodes_allocated, d_node_box);
  {
    const CollapseItem root{0, 0u};
    LBVH_TRY(hipMemcpy(d_queue[0], &root, sizeof(root), hipMemcpyHostToDevice));
    const uint32_t init[4] = {0u, 1u, 0u, 0u};
    LBVH_TRY(hipMemcpy(d_counters, init, sizeof(init), hipMemcpyHostToDevice));
  }
  if (optimal && !make_collapse_plan((int) nodes_allocated, d_children, d_ranges, d_node_box, max_leaf, d_plan_done, d_plan)) { ok = false; goto done; }
  for (int q = 0; level_count > 0; q ^= 1) {
    depth++;
    if (depth > max_depth) { ok = false; goto done; }
    hipLaunchKernelGGL(k_lbvh_collapse, dim3((level_count + threads - 1) / threads), dim3(threads), 0, 0, (int) n, (const int2*) d_children, (const int2*) d_ranges,
                       (const BinBox*) d_node_box, max_leaf, (const CollapseItem*) d_queue[q], level_count, d_queue[q ^ 1], d_counters, d_counters + 1, d_nodes,
                       (const float4*) d_plan);
    uint32_t host_counters[2];
    LBVH_TRY(hipMemcpy(host_counters, d_counters, sizeof(host_counters), hipMemcpyDeviceToHost));
    level_count = host_counters[0];
    node_count = host_counters[1];
    if (node_count > max_nodes4) { ok = false; goto done; }
    LBVH_TRY(hipMemset(d_counters, 0, sizeof(uint32_t)));
  }
  result.nodes.resize(node_count);
  result.prims.resize(count);
  LBVH_TRY(hipMemcpy(result.nodes.data(), d_nodes, sizeof(Bvh4Node) * node_count, hipMemcpyDeviceToHost));
  LBVH_TRY(hipMemcpy(result.prims.data(), d_ids[cur], sizeof(uint32_t) * count, hipMemcpyDeviceToHost));
  result.max_depth = depth;
done:
  {
    if (d_pool) (void) hipFree(d_pool);
  }
  if (!ok) return Bvh4();
  return result;
}

Bvh4 build_bvh4_lbvh(const Aabb* boxes, uint32_t count, uint32_t max_leaf, uint32_t max_depth) { return build_on_device(boxes, count, max_leaf, max_depth, false); }
Bvh4 build_bvh4_ploc(const Aabb* boxes, uint32_t count, uint32_t max_leaf, uint32_t max_depth) { return build_on_device(boxes, count, max_leaf, max_depth, true); }

}  // namespace lum
