// Offline measure of the host builder's trees: node visits, leaf visits and triangle tests per ray of a CPU walk that follows the device traversal
// (dev_trace.h: nearest child first, far children stacked, closest-hit rays cull by the hit distance, visibility rays stop at the first hit).
// The ray kernels' time follows the node visits (profiles/r03_ab_experiments.txt: +10 % visits = +28 % time on the hall), so builder changes can be judged
// here before they go to the GPU.
//   g++ -O2 -std=c++17 -fopenmp -I luminary_amd/csrc/host -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -o /tmp/bvh_quality tools/bvh_quality.cpp luminary_amd/csrc/host/bvh_build.cpp -lpthread
//   /tmp/bvh_quality vertices.f32 [rays] [sah|lbvh|ploc] [radius]     vertices.f32 = the mesh as the device scene holds it: 3 x float4 per triangle
//   lbvh / ploc: CPU models of the GPU builders (lbvh.hip) - the Morton-ordered radix tree, and parallel locally-ordered clustering (Meister, Bittner 2018)
//   over the same order - collapsed to 4-wide nodes by the rule k_lbvh_collapse uses; to judge a builder's trees before it is written for the GPU.
//   hybrid: the SAH builder's tree over the boxes of the last K clusters of the clustering, the clusters' subtrees below it (argv: radius K).
// Rays: origins on random triangles (pushed off the surface), cosine-distributed directions about the normal - what a path tracer's bounces look like.
// Environment knobs (each a question that was asked of the model before anything was written for the GPU; answers in profiles/r03_ab_experiments.txt):
//   BQ_ANYHIT_ORDER=1..8   child order of any-hit rays: 1 farthest entry first (what the visibility kernel does since), 2 largest box, 3 farthest exit, 4 longest overlap,
//                          5 overlap x area, 6 / 8 farthest first in the first 4096 / 65536 nodes only, 7 everywhere but there;  BQ_ANYHIT_UNSORTED=1 stored order
//   BQ_AXIS_ORDER=1|2      closest-hit rays: children ordered without distances, along the node's longest axis / the ray's dominant axis
//   BQ_TIE=1|2             closest-hit rays: equal entry distances broken by the longer / shorter stay in the box
//   BQ_MAX_LEAF=n          leaf size of the lbvh / ploc models;  BQ_PLOC_CRIT=1|2 merge criterion;  BQ_EMC=n extended Morton codes with a size bit every n position bits
#include <chrono>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "bvh_build.h"

using namespace lum;

struct V { float x, y, z; };
static V sub(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V add(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V mul(V a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V norm(V a) { const float l = std::sqrt(dot(a, a)); return l > 0 ? mul(a, 1.0f / l) : V{0, 0, 1}; }

static float hit_tri(const float* p, V o, V d) {  // Moeller-Trumbore on p0, p1, p2 (float4 stride)
  const V p0{p[0], p[1], p[2]}, e1 = sub(V{p[4], p[5], p[6]}, p0), e2 = sub(V{p[8], p[9], p[10]}, p0);
  const V h = cross(d, e2);
  const float a = dot(e1, h);
  if (std::fabs(a) < 1e-20f) return INFINITY;
  const float f = 1.0f / a;
  const V s = sub(o, p0);
  const float u = f * dot(s, h);
  if (u < 0 || u > 1) return INFINITY;
  const V q = cross(s, e1);
  const float v = f * dot(d, q);
  if (v < 0 || u + v > 1) return INFINITY;
  const float t = f * dot(e2, q);
  return t > 1e-4f ? t : INFINITY;
}

struct Stats { double nodes = 0, leaves = 0, tris = 0, rays = 0, hits = 0; };

static void walk(const Bvh4& bvh, const std::vector<float>& verts, V o, V d, float tmax, bool any_hit, Stats& st) {
  const float inv[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
  const float oo[3] = {o.x, o.y, o.z};
  struct E { uint32_t node; float t; };
  E stack[256];
  int sp = 0;
  uint32_t cur = 0;
  float best = tmax;
  st.rays++;
  while (true) {
    if (cur & kBvhLeafBit) {
      const uint32_t first = cur & 0x0FFFFFFFu, count = ((cur >> 28) & 7u) + 1u;
      st.leaves++;
      bool stop = false;
      for (uint32_t j = 0; j < count; j++) {
        st.tris++;
        const float t = hit_tri(&verts[(size_t) bvh.prims[first + j] * 12], o, d);
        if (t < best) { best = t; if (any_hit) { stop = true; break; } }
      }
      if (stop) { st.hits++; return; }
    }
    else {
      st.nodes++;
      const Bvh4Node& n = bvh.nodes[cur];
      float k[4], kf[4]; uint32_t c[4];
      for (int j = 0; j < 4; j++) {
        c[j] = n.child[j];
        k[j] = INFINITY; kf[j] = INFINITY;
        if (c[j] == kBvhEmpty) continue;
        const float lo[3] = {n.lo_x[j], n.lo_y[j], n.lo_z[j]}, hi[3] = {n.hi_x[j], n.hi_y[j], n.hi_z[j]};
        float tn = 0.0f, tf = best;
        for (int a = 0; a < 3; a++) {
          float t0 = (lo[a] - oo[a]) * inv[a], t1 = (hi[a] - oo[a]) * inv[a];
          if (t0 > t1) std::swap(t0, t1);
          tn = std::max(tn, t0); tf = std::min(tf, t1);
        }
        if (tn <= tf) { k[j] = tn; kf[j] = tf; }
      }
      static const int axis_order = std::getenv("BQ_AXIS_ORDER") ? std::atoi(std::getenv("BQ_AXIS_ORDER")) : 0;
      if (axis_order && !any_hit) {  // closest-hit rays: children ordered by their centres along the node's longest axis (1) / the ray's dominant axis (2), by the direction's sign - no distances
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int j = 0; j < 4; j++) {
          if (n.child[j] == kBvhEmpty) continue;
          lo[0] = std::min(lo[0], n.lo_x[j]); lo[1] = std::min(lo[1], n.lo_y[j]); lo[2] = std::min(lo[2], n.lo_z[j]);
          hi[0] = std::max(hi[0], n.hi_x[j]); hi[1] = std::max(hi[1], n.hi_y[j]); hi[2] = std::max(hi[2], n.hi_z[j]);
        }
        int ax = 0;
        if (axis_order == 1) { for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[ax] - lo[ax]) ax = a; }
        else { const float dd[3] = {std::fabs(d.x), std::fabs(d.y), std::fabs(d.z)}; for (int a = 1; a < 3; a++) if (dd[a] > dd[ax]) ax = a; }
        const float sgn = (ax == 0 ? d.x : ax == 1 ? d.y : d.z) < 0 ? -1.0f : 1.0f;
        float key[4];
        for (int j = 0; j < 4; j++) {
          const float c0 = ax == 0 ? n.lo_x[j] + n.hi_x[j] : ax == 1 ? n.lo_y[j] + n.hi_y[j] : n.lo_z[j] + n.hi_z[j];
          key[j] = k[j] < INFINITY ? sgn * c0 : INFINITY;
        }
        for (int a = 0; a < 4; a++)
          for (int b = a + 1; b < 4; b++) if (key[b] < key[a]) { std::swap(key[a], key[b]); std::swap(k[a], k[b]); std::swap(kf[a], kf[b]); std::swap(c[a], c[b]); }
      }
      else
      if (!(any_hit && (std::getenv("BQ_ANYHIT_UNSORTED") || std::getenv("BQ_ANYHIT_ORDER")))) {
      static const int tie = std::getenv("BQ_TIE") ? std::atoi(std::getenv("BQ_TIE")) : 0;  // ties of the entry distance (origin inside both boxes): 1 longer stay first, 2 shorter stay first
      for (int a = 0; a < 4; a++)  // sort by entry distance (4 entries)
        for (int b = a + 1; b < 4; b++) {
          bool sw = k[b] < k[a];
          if (tie && k[b] == k[a] && k[a] < INFINITY) sw = tie == 1 ? kf[b] > kf[a] : kf[b] < kf[a];
          if (sw) { std::swap(k[a], k[b]); std::swap(kf[a], kf[b]); std::swap(c[a], c[b]); }
        }
      }
      if (any_hit && std::getenv("BQ_ANYHIT_ORDER")) {  // 1 farthest first, 2 largest box first
        const int mode = std::atoi(std::getenv("BQ_ANYHIT_ORDER"));
        float key[4];
        for (int a = 0; a < 4; a++) {
          key[a] = INFINITY;
          if (k[a] == INFINITY) continue;
          int j = 0; for (; j < 4; j++) if (n.child[j] == c[a]) break;
          const float dx = n.hi_x[j] - n.lo_x[j], dy = n.hi_y[j] - n.lo_y[j], dz = n.hi_z[j] - n.lo_z[j];
          key[a] = mode == 1 ? -k[a] : mode == 6 ? (cur < 4096u ? -k[a] : k[a]) : mode == 7 ? (cur < 4096u ? k[a] : -k[a]) : mode == 8 ? (cur < 65536u ? -k[a] : k[a]) : mode == 2 ? -(dx * dy + dy * dz + dz * dx) : mode == 3 ? -kf[j] : mode == 4 ? -(kf[j] - k[a]) : -(kf[j] - k[a]) * (dx * dy + dy * dz + dz * dx);
        }
        for (int a = 0; a < 4; a++)
          for (int b = a + 1; b < 4; b++) if (key[b] < key[a]) { std::swap(key[a], key[b]); std::swap(k[a], k[b]); std::swap(c[a], c[b]); }
      }
      if (any_hit && std::getenv("BQ_ANYHIT_UNSORTED")) {  // stored order, misses moved to the end
        int w = 0; float k2[4]; uint32_t c2[4];
        for (int a = 0; a < 4; a++) if (k[a] < INFINITY) { k2[w] = k[a]; c2[w] = c[a]; w++; }
        for (int a = w; a < 4; a++) { k2[a] = INFINITY; c2[a] = kBvhEmpty; }
        for (int a = 0; a < 4; a++) { k[a] = k2[a]; c[a] = c2[a]; }
      }
      for (int j = 3; j >= 1; j--) if (k[j] < INFINITY) stack[sp++] = {c[j], k[j]};
      if (k[0] < INFINITY) { cur = c[0]; continue; }
    }
    bool found = false;
    while (sp > 0) { const E e = stack[--sp]; if (any_hit || e.t <= best) { cur = e.node; found = true; break; } }
    if (!found) break;
  }
  if (best < tmax) st.hits++;
}

// ---- CPU models of the GPU builders ----
#include <algorithm>
#include <functional>
#include <numeric>
struct BinTree { std::vector<int> left, right; std::vector<Aabb> box; std::vector<uint32_t> count; int root = -1; uint32_t leaves = 0; };  // node k < leaves: primitive order[k]
static Aabb merge(const Aabb& a, const Aabb& b) { Aabb r; for (int k = 0; k < 3; k++) { r.lo[k] = std::min(a.lo[k], b.lo[k]); r.hi[k] = std::max(a.hi[k], b.hi[k]); } return r; }
static float half_area(const Aabb& b) { const float x = b.hi[0] - b.lo[0], y = b.hi[1] - b.lo[1], z = b.hi[2] - b.lo[2]; return x * y + y * z + z * x; }
static uint64_t spread21(uint64_t v) { v &= 0x1FFFFF; v = (v | v << 32) & 0x1F00000000FFFFull; v = (v | v << 16) & 0x1F0000FF0000FFull; v = (v | v << 8) & 0x100F00F00F00F00Full; v = (v | v << 4) & 0x10C30C30C30C30C3ull; v = (v | v << 2) & 0x1249249249249249ull; return v; }
static std::vector<uint32_t> morton_order(const std::vector<Aabb>& boxes, std::vector<uint64_t>& codes) {
  Aabb all = boxes[0];
  for (const Aabb& b : boxes) all = merge(all, b);
  const size_t n = boxes.size();
  codes.resize(n);
  std::vector<uint32_t> order(n);
  for (size_t i = 0; i < n; i++) {
    uint64_t q[3];
    for (int k = 0; k < 3; k++) { const float c = 0.5f * (boxes[i].lo[k] + boxes[i].hi[k]), e = std::max(all.hi[k] - all.lo[k], 1e-30f); q[k] = (uint64_t) std::min(std::max((c - all.lo[k]) / e * 2097152.0f, 0.0f), 2097151.0f); }
    codes[i] = spread21(q[0]) << 2 | spread21(q[1]) << 1 | spread21(q[2]);
    if (std::getenv("BQ_EMC")) {  // extended Morton code (Vinkler et al. 2017): a size bit after every BQ_EMC position bits, large primitives separate early
      const int every = std::atoi(std::getenv("BQ_EMC"));
      float diag = 0, ext = 0;
      for (int k = 0; k < 3; k++) { const float e = boxes[i].hi[k] - boxes[i].lo[k]; diag += e * e; const float a = all.hi[k] - all.lo[k]; ext += a * a; }
      const uint64_t sz = (uint64_t) std::min(std::max(std::sqrt(diag / std::max(ext, 1e-30f)) * 1048576.0f, 0.0f), 1048575.0f);  // 20 bits
      const uint64_t pos = codes[i];
      uint64_t out = 0; int ob = 63, pb = 62, sb = 19, run = 0;
      while (ob >= 0 && pb >= 0) {
        if (run == every && sb >= 0) { out |= ((sz >> sb) & 1ull) << ob; sb--; ob--; run = 0; continue; }
        out |= ((pos >> pb) & 1ull) << ob; pb--; ob--; run++;
      }
      codes[i] = out;
    }
    order[i] = (uint32_t) i;
  }
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return codes[a] != codes[b] ? codes[a] < codes[b] : a < b; });
  return order;
}
static BinTree build_lbvh_model(const std::vector<Aabb>& boxes, std::vector<uint32_t>& order) {
  std::vector<uint64_t> codes;
  order = morton_order(boxes, codes);
  const uint32_t n = (uint32_t) boxes.size();
  BinTree t; t.leaves = n;
  t.left.assign(n, -1); t.right.assign(n, -1); t.box.resize(n); t.count.assign(n, 1);
  for (uint32_t k = 0; k < n; k++) t.box[k] = boxes[order[k]];
  std::function<int(uint32_t, uint32_t)> rec = [&](uint32_t a, uint32_t b) -> int {  // [a, b]
    if (a == b) return (int) a;
    const uint64_t ca = codes[order[a]], cb = codes[order[b]];
    uint32_t split;
    if (ca == cb) split = (a + b) / 2;
    else {
      const int bit = 63 - __builtin_clzll(ca ^ cb);
      uint32_t lo = a, hi = b;  // last index whose bit is 0
      while (lo < hi) { const uint32_t m = (lo + hi + 1) / 2; if ((codes[order[m]] >> bit) & 1) hi = m - 1; else lo = m; }
      split = lo;
    }
    const int l = rec(a, split), r = rec(split + 1, b);
    t.left.push_back(l); t.right.push_back(r); t.box.push_back(merge(t.box[l], t.box[r])); t.count.push_back(t.count[l] + t.count[r]);
    return (int) t.left.size() - 1;
  };
  t.root = rec(0, n - 1);
  return t;
}
static BinTree build_ploc_model(const std::vector<Aabb>& boxes, std::vector<uint32_t>& order, int radius, size_t stop_at = 1, std::vector<int>* roots = nullptr) {
  std::vector<uint64_t> codes;
  order = morton_order(boxes, codes);
  const uint32_t n = (uint32_t) boxes.size();
  BinTree t; t.leaves = n;
  t.left.assign(n, -1); t.right.assign(n, -1); t.box.resize(n); t.count.assign(n, 1);
  for (uint32_t k = 0; k < n; k++) t.box[k] = boxes[order[k]];
  std::vector<int> cur(n), next;
  std::iota(cur.begin(), cur.end(), 0);
  std::vector<int> nn;
  int iterations = 0;
  while (cur.size() > stop_at) {
    const int m = (int) cur.size();
    nn.assign(m, -1);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < m; i++) {
      float best = INFINITY; int bj = -1;
      for (int j = std::max(0, i - radius); j <= std::min(m - 1, i + radius); j++) {
        if (j == i) continue;
        float a = half_area(merge(t.box[cur[i]], t.box[cur[j]]));
        static const int crit = std::getenv("BQ_PLOC_CRIT") ? std::atoi(std::getenv("BQ_PLOC_CRIT")) : 0;
        if (crit == 1) a *= (float) (t.count[cur[i]] + t.count[cur[j]]);
        else if (crit == 2) a -= half_area(t.box[cur[i]]) + half_area(t.box[cur[j]]);
        if (a < best) { best = a; bj = j; }  // ties: the lower index, on both sides of the pair
      }
      nn[i] = bj;
    }
    next.clear();
    for (int i = 0; i < m; i++) {
      const int j = nn[i];
      if (nn[j] == i) {
        if (i < j) {
          const int l = cur[i], r = cur[j];
          t.left.push_back(l); t.right.push_back(r); t.box.push_back(merge(t.box[l], t.box[r])); t.count.push_back(t.count[l] + t.count[r]);
          next.push_back((int) t.left.size() - 1);
        }
      }
      else next.push_back(cur[i]);
    }
    cur.swap(next);
    iterations++;
  }
  t.root = cur[0];
  if (roots) *roots = cur;
  std::printf("ploc: radius %d, %d iterations, %zu clusters left\n", radius, iterations, cur.size());
  return t;
}
// Insertion-based optimisation (Bittner et al. 2013; bvh_build.cpp Reinserter) on the model trees: BQ_REINSERT=<passes>. What would a clustered tree gain from it?
static void reinsert_model(BinTree& t, int passes) {
  const size_t n = t.left.size();
  std::vector<int> parent(n, -1);
  for (size_t i = 0; i < n; i++) if (t.left[i] >= 0) { parent[t.left[i]] = (int) i; parent[t.right[i]] = (int) i; }
  auto refit_up = [&](int i) { while (i >= 0) { t.box[i] = merge(t.box[t.left[i]], t.box[t.right[i]]); i = parent[i]; } };
  std::vector<std::pair<float, int>> heap;
  auto cmp = [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first > b.first; };
  for (int pass = 0; pass < passes; pass++) {
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::vector<float> key(n);
    for (size_t i = 0; i < n; i++) key[i] = half_area(t.box[i]);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return key[a] > key[b] || (key[a] == key[b] && a < b); });
    double before = 0; for (size_t i = 0; i < n; i++) if (t.left[i] >= 0) before += half_area(t.box[i]);
    size_t moved = 0;
    for (int nd : order) {
      const int p = parent[nd];
      if (p < 0 || p == t.root) continue;
      const int g = parent[p], s = t.left[p] == nd ? t.right[p] : t.left[p];
      (t.left[g] == p ? t.left[g] : t.right[g]) = s; parent[s] = g; refit_up(g);
      const Aabb nb = t.box[nd]; const float na = half_area(nb);
      float best_cost = INFINITY; int best = s;
      heap.clear(); heap.emplace_back(0.0f, t.root);
      while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        const float induced = heap.back().first; const int x = heap.back().second; heap.pop_back();
        if (induced + na >= best_cost) break;
        const float total = induced + half_area(merge(t.box[x], nb));
        if (total < best_cost) { best_cost = total; best = x; }
        if (t.left[x] >= 0) {
          const float ci = total - half_area(t.box[x]);
          if (ci + na < best_cost) { heap.emplace_back(ci, t.left[x]); std::push_heap(heap.begin(), heap.end(), cmp); heap.emplace_back(ci, t.right[x]); std::push_heap(heap.begin(), heap.end(), cmp); }
        }
      }
      if (parent[best] < 0) best = s;
      const int bp = parent[best];
      (t.left[bp] == best ? t.left[bp] : t.right[bp]) = p; parent[p] = bp;
      t.left[p] = best; t.right[p] = nd; parent[best] = p; parent[nd] = p;
      refit_up(p);
      moved += best != s;
    }
    double after = 0; for (size_t i = 0; i < n; i++) if (t.left[i] >= 0) after += half_area(t.box[i]);
    std::printf("reinsertion pass %d: %zu subtrees moved, inner area %.6g -> %.6g\n", pass, moved, before, after);
  }
  // subtree sizes
  std::function<uint32_t(int)> cnt = [&](int b) -> uint32_t { if (t.left[b] < 0) return t.count[b] = 1; return t.count[b] = cnt(t.left[b]) + cnt(t.right[b]); };
  std::vector<int> stack{t.root}, post;
  while (!stack.empty()) { const int b = stack.back(); stack.pop_back(); post.push_back(b); if (t.left[b] >= 0) { stack.push_back(t.left[b]); stack.push_back(t.right[b]); } }
  for (size_t k = post.size(); k-- > 0;) { const int b = post[k]; t.count[b] = t.left[b] < 0 ? 1u : t.count[t.left[b]] + t.count[t.right[b]]; }
}

// k_lbvh_collapse's rule: open the child with the largest box until there are four; a subtree of at most max_leaf primitives is a leaf.
static Bvh4 collapse_model(const BinTree& t, const std::vector<uint32_t>& order, uint32_t max_leaf) {
  Bvh4 out;
  // primitive order: depth-first over the binary tree, so that every subtree is a contiguous range
  std::vector<uint32_t> first(t.left.size(), 0);
  {
    std::vector<std::pair<int, uint32_t>> stack{{t.root, 0u}};
    out.prims.resize(t.leaves);
    while (!stack.empty()) {
      auto [b, off] = stack.back(); stack.pop_back();
      first[b] = off;
      if (t.left[b] < 0) { out.prims[off] = order[b]; continue; }
      stack.push_back({t.left[b], off});
      stack.push_back({t.right[b], off + t.count[t.left[b]]});
    }
  }
  struct Item { int bin; uint32_t node4; uint32_t depth; };
  std::vector<Item> queue{{t.root, 0u, 1u}};
  out.nodes.resize(1);
  for (size_t q = 0; q < queue.size(); q++) {
    const Item item = queue[q];
    out.max_depth = std::max(out.max_depth, item.depth);
    auto is_leaf = [&](int b) { return t.count[b] <= max_leaf; };
    int kids[4]; int nk = 0;
    if (is_leaf(item.bin)) kids[nk++] = item.bin;
    else { kids[nk++] = t.left[item.bin]; kids[nk++] = t.right[item.bin]; }
    while (nk < 4) {
      int pick = -1; float best = -1.0f;
      for (int k = 0; k < nk; k++) { if (is_leaf(kids[k])) continue; const float a = half_area(t.box[kids[k]]); if (a > best) { best = a; pick = k; } }
      if (pick < 0) break;
      const int b = kids[pick];
      kids[pick] = t.left[b]; kids[nk++] = t.right[b];
    }
    Bvh4Node node;
    for (int k = 0; k < 4; k++) { node.child[k] = kBvhEmpty; node.pad[k] = 0; node.lo_x[k] = node.lo_y[k] = node.lo_z[k] = 3.4e38f; node.hi_x[k] = node.hi_y[k] = node.hi_z[k] = -3.4e38f; }
    for (int k = 0; k < nk; k++) {
      const Aabb& b = t.box[kids[k]];
      node.lo_x[k] = b.lo[0]; node.lo_y[k] = b.lo[1]; node.lo_z[k] = b.lo[2]; node.hi_x[k] = b.hi[0]; node.hi_y[k] = b.hi[1]; node.hi_z[k] = b.hi[2];
      if (is_leaf(kids[k])) node.child[k] = kBvhLeafBit | ((t.count[kids[k]] - 1u) << 28) | first[kids[k]];
      else { node.child[k] = (uint32_t) out.nodes.size(); queue.push_back({kids[k], (uint32_t) out.nodes.size(), item.depth + 1}); out.nodes.emplace_back(); }
    }
    out.nodes[item.node4] = node;
  }
  return out;
}

// Top of the tree by the host's SAH builder over the boxes of the clusters PLOC stopped at, the clusters' subtrees collapsed below it.
static Bvh4 hybrid_model(const BinTree& t, const std::vector<uint32_t>& order, const std::vector<int>& roots, uint32_t max_leaf) {
  std::vector<Aabb> cb(roots.size());
  for (size_t k = 0; k < roots.size(); k++) cb[k] = t.box[roots[k]];
  const Bvh4 top = build_bvh4(cb.data(), (uint32_t) cb.size(), 1, 26);
  Bvh4 out;
  out.nodes = top.nodes;
  out.max_depth = top.max_depth;
  // primitive order: cluster after cluster (in the top tree's leaf order), depth-first inside
  std::vector<uint32_t> first(t.left.size(), 0);
  out.prims.resize(t.leaves);
  uint32_t base = 0;
  std::vector<uint32_t> cluster_first(roots.size());
  for (uint32_t c : top.prims) {
    cluster_first[c] = base;
    std::vector<std::pair<int, uint32_t>> stack{{roots[c], base}};
    while (!stack.empty()) {
      auto [b, off] = stack.back(); stack.pop_back();
      first[b] = off;
      if (t.left[b] < 0) { out.prims[off] = order[b]; continue; }
      stack.push_back({t.left[b], off});
      stack.push_back({t.right[b], off + t.count[t.left[b]]});
    }
    base += t.count[roots[c]];
  }
  struct Item { int bin; uint32_t node4; uint32_t depth; };
  std::vector<Item> queue;
  auto is_leaf = [&](int b) { return t.count[b] <= max_leaf; };
  for (Bvh4Node& n : out.nodes)
    for (int k = 0; k < 4; k++) {
      if (n.child[k] == kBvhEmpty || !(n.child[k] & kBvhLeafBit)) continue;
      const uint32_t c = top.prims[n.child[k] & 0x0FFFFFFFu];  // one cluster per top-level leaf
      const int b = roots[c];
      if (is_leaf(b)) n.child[k] = kBvhLeafBit | ((t.count[b] - 1u) << 28) | first[b];
      else { n.child[k] = (uint32_t) (out.nodes.size() + queue.size()); queue.push_back({b, n.child[k], top.max_depth + 1}); }
    }
  out.nodes.resize(out.nodes.size() + queue.size());
  for (size_t q = 0; q < queue.size(); q++) {
    const Item item = queue[q];
    out.max_depth = std::max(out.max_depth, item.depth);
    int kids[4]; int nk = 0;
    kids[nk++] = t.left[item.bin]; kids[nk++] = t.right[item.bin];
    while (nk < 4) {
      int pick = -1; float best = -1.0f;
      for (int k = 0; k < nk; k++) { if (is_leaf(kids[k])) continue; const float a = half_area(t.box[kids[k]]); if (a > best) { best = a; pick = k; } }
      if (pick < 0) break;
      const int b = kids[pick];
      kids[pick] = t.left[b]; kids[nk++] = t.right[b];
    }
    Bvh4Node node;
    for (int k = 0; k < 4; k++) { node.child[k] = kBvhEmpty; node.pad[k] = 0; node.lo_x[k] = node.lo_y[k] = node.lo_z[k] = 3.4e38f; node.hi_x[k] = node.hi_y[k] = node.hi_z[k] = -3.4e38f; }
    for (int k = 0; k < nk; k++) {
      const Aabb& b = t.box[kids[k]];
      node.lo_x[k] = b.lo[0]; node.lo_y[k] = b.lo[1]; node.lo_z[k] = b.lo[2]; node.hi_x[k] = b.hi[0]; node.hi_y[k] = b.hi[1]; node.hi_z[k] = b.hi[2];
      if (is_leaf(kids[k])) node.child[k] = kBvhLeafBit | ((t.count[kids[k]] - 1u) << 28) | first[kids[k]];
      else { node.child[k] = (uint32_t) out.nodes.size(); queue.push_back({kids[k], (uint32_t) out.nodes.size(), item.depth + 1}); out.nodes.emplace_back(); }
    }
    out.nodes[item.node4] = node;
  }
  return out;
}

// ---- 8-wide model: the 4-wide tree's nodes widened by opening their largest inner children until there are eight; walked nearest first (exact
// entry distances) or in the order of the children's centres along the ray's octant diagonal (what a node with children stored in octant order gives
// without computing or sorting any distance) ----
struct Wide { float lo[8][3], hi[8][3]; uint32_t child[8]; int n; };
static std::vector<Wide> widen(const Bvh4& b) {
  std::vector<Wide> out;
  struct Item { uint32_t node4; uint32_t wide; };
  std::vector<Item> queue{{0u, 0u}};
  out.emplace_back();
  for (size_t q = 0; q < queue.size(); q++) {
    struct Kid { Aabb box; uint32_t ref; };
    std::vector<Kid> kids;
    auto add_children = [&](uint32_t n4) {
      const Bvh4Node& n = b.nodes[n4];
      for (int k = 0; k < 4; k++) if (n.child[k] != kBvhEmpty) kids.push_back({Aabb{{n.lo_x[k], n.lo_y[k], n.lo_z[k]}, {n.hi_x[k], n.hi_y[k], n.hi_z[k]}}, n.child[k]});
    };
    add_children(queue[q].node4);
    while (true) {
      int pick = -1; float best = -1.0f;
      for (size_t k = 0; k < kids.size(); k++) {
        if (kids[k].ref & kBvhLeafBit) continue;
        const Bvh4Node& c = b.nodes[kids[k].ref];
        int cn = 0; for (int j = 0; j < 4; j++) cn += c.child[j] != kBvhEmpty;
        if (kids.size() - 1 + cn > 8) continue;
        const float a = half_area(kids[k].box);
        if (a > best) { best = a; pick = (int) k; }
      }
      if (pick < 0) break;
      const uint32_t open = kids[pick].ref;
      kids.erase(kids.begin() + pick);
      add_children(open);
    }
    Wide w; w.n = (int) kids.size();
    for (int k = 0; k < w.n; k++) {
      for (int a = 0; a < 3; a++) { w.lo[k][a] = kids[k].box.lo[a]; w.hi[k][a] = kids[k].box.hi[a]; }
      if (kids[k].ref & kBvhLeafBit) w.child[k] = kids[k].ref;
      else { w.child[k] = (uint32_t) out.size(); queue.push_back({kids[k].ref, (uint32_t) out.size()}); out.emplace_back(); }
    }
    out[queue[q].wide] = w;
  }
  return out;
}
static void walk_wide(const std::vector<Wide>& nodes, const Bvh4& bvh, const std::vector<float>& verts, V o, V d, bool octant_order, Stats& st) {
  const float inv[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z}, oo[3] = {o.x, o.y, o.z}, sg[3] = {d.x < 0 ? -1.0f : 1.0f, d.y < 0 ? -1.0f : 1.0f, d.z < 0 ? -1.0f : 1.0f};
  struct E { uint32_t node; float t; };
  E stack[512]; int sp = 0; uint32_t cur = 0; float best = INFINITY;
  st.rays++;
  while (true) {
    if (cur & kBvhLeafBit) {
      const uint32_t first = cur & 0x0FFFFFFFu, count = ((cur >> 28) & 7u) + 1u;
      st.leaves++;
      for (uint32_t j = 0; j < count; j++) { st.tris++; const float t = hit_tri(&verts[(size_t) bvh.prims[first + j] * 12], o, d); if (t < best) best = t; }
    }
    else {
      st.nodes++;
      const Wide& n = nodes[cur];
      float k[8], key[8]; uint32_t c[8]; int m = 0;
      for (int j = 0; j < n.n; j++) {
        float tn = 0.0f, tf = best;
        for (int a = 0; a < 3; a++) { float t0 = (n.lo[j][a] - oo[a]) * inv[a], t1 = (n.hi[j][a] - oo[a]) * inv[a]; if (t0 > t1) std::swap(t0, t1); tn = std::max(tn, t0); tf = std::min(tf, t1); }
        if (tn > tf) continue;
        k[m] = tn; c[m] = n.child[j];
        key[m] = octant_order ? sg[0] * (n.lo[j][0] + n.hi[j][0]) + sg[1] * (n.lo[j][1] + n.hi[j][1]) + sg[2] * (n.lo[j][2] + n.hi[j][2]) : tn;
        m++;
      }
      for (int a = 0; a < m; a++) for (int b2 = a + 1; b2 < m; b2++) if (key[b2] < key[a]) { std::swap(key[a], key[b2]); std::swap(k[a], k[b2]); std::swap(c[a], c[b2]); }
      for (int j = m - 1; j >= 1; j--) stack[sp++] = {c[j], k[j]};
      if (m > 0) { cur = c[0]; continue; }
    }
    bool found = false;
    while (sp > 0) { const E e = stack[--sp]; if (e.t <= best) { cur = e.node; found = true; break; } }
    if (!found) break;
  }
  if (best < INFINITY) st.hits++;
}

// Mode 2 of the 8-wide model - what a compressed wide BVH really does (Ylitie, Karras, Laine 2017; the reference's dead software path: slot assignment
// bvh.c:1093-1145, traversal order cuda/bvh.cuh:82-106): every child sits in one of eight octant slots chosen at build time (greedy: the (child, slot) pair
// with the largest projection of the child's offset from the node's centre onto the slot's diagonal first), a ray meets the slots in the order
// slot ^ octant, and the stack holds GROUPS (node, slots left) without distances: a group popped later is walked even when the hit found meanwhile lies
// in front of it (every child is still tested against the current hit distance when it is visited). Quantised boxes are not modelled.
static void assign_slots(std::vector<Wide>& nodes) {
  for (Wide& w : nodes) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < w.n; k++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], w.lo[k][a]); hi[a] = std::max(hi[a], w.hi[k][a]); }
    float cost[8][8];
    for (int k = 0; k < w.n; k++)
      for (int sl = 0; sl < 8; sl++) {
        float c = 0;
        for (int a = 0; a < 3; a++) c += (((sl >> a) & 1) ? 1.0f : -1.0f) * (0.5f * (w.lo[k][a] + w.hi[k][a]) - 0.5f * (lo[a] + hi[a]));
        cost[k][sl] = c;
      }
    int slot_of[8]; bool used_k[8] = {false}, used_s[8] = {false};
    for (int it = 0; it < w.n; it++) {
      int bk = -1, bs = -1; float best = -INFINITY;
      for (int k = 0; k < w.n; k++) if (!used_k[k]) for (int sl = 0; sl < 8; sl++) if (!used_s[sl] && cost[k][sl] > best) { best = cost[k][sl]; bk = k; bs = sl; }
      used_k[bk] = used_s[bs] = true; slot_of[bk] = bs;
    }
    Wide out = w;
    for (int sl = 0; sl < 8; sl++) out.child[sl] = kBvhEmpty;
    for (int k = 0; k < w.n; k++) { const int sl = slot_of[k]; out.child[sl] = w.child[k]; for (int a = 0; a < 3; a++) { out.lo[sl][a] = w.lo[k][a]; out.hi[sl][a] = w.hi[k][a]; } }
    out.n = 8;
    w = out;
  }
}
static void walk_groups(const std::vector<Wide>& nodes, const Bvh4& bvh, const std::vector<float>& verts, V o, V d, Stats& st) {
  const float inv[3] = {1.0f / d.x, 1.0f / d.y, 1.0f / d.z}, oo[3] = {o.x, o.y, o.z};
  const int oct = (d.x < 0 ? 1 : 0) | (d.y < 0 ? 2 : 0) | (d.z < 0 ? 4 : 0);
  struct G { uint32_t node; uint32_t mask; };  // mask in priority order: bit r = slot r ^ oct
  G stack[512]; int sp = 0; float best = INFINITY;
  st.rays++;
  auto visit = [&](uint32_t node) -> uint32_t {
    st.nodes++;
    const Wide& n = nodes[node];
    uint32_t mask = 0;
    for (int sl = 0; sl < 8; sl++) {
      if (n.child[sl] == kBvhEmpty) continue;
      float tn = 0.0f, tf = best;
      for (int a = 0; a < 3; a++) { float t0 = (n.lo[sl][a] - oo[a]) * inv[a], t1 = (n.hi[sl][a] - oo[a]) * inv[a]; if (t0 > t1) std::swap(t0, t1); tn = std::max(tn, t0); tf = std::min(tf, t1); }
      if (tn <= tf) mask |= 1u << (sl ^ oct);
    }
    return mask;
  };
  G g{0u, visit(0u)};
  while (true) {
    if (g.mask == 0) { if (sp == 0) break; g = stack[--sp]; continue; }
    const int r = __builtin_ctz(g.mask);
    g.mask &= g.mask - 1;
    const uint32_t child = nodes[g.node].child[r ^ oct];
    if (child & kBvhLeafBit) {
      const uint32_t first = child & 0x0FFFFFFFu, count = ((child >> 28) & 7u) + 1u;
      st.leaves++;
      for (uint32_t j = 0; j < count; j++) { st.tris++; const float t = hit_tri(&verts[(size_t) bvh.prims[first + j] * 12], o, d); if (t < best) best = t; }
    }
    else {
      if (g.mask) stack[sp++] = g;
      g = G{child, visit(child)};
    }
  }
  if (best < INFINITY) st.hits++;
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: bvh_quality vertices.f32 [rays]\n"); return 1; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror("open"); return 1; }
  std::fseek(f, 0, SEEK_END);
  const size_t bytes = (size_t) std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<float> verts(bytes / 4);
  if (std::fread(verts.data(), 4, verts.size(), f) != verts.size()) return 1;
  std::fclose(f);
  const uint32_t nt = (uint32_t) (verts.size() / 12);
  const uint32_t nrays = argc > 2 ? (uint32_t) std::atoi(argv[2]) : 400000;
  std::vector<Aabb> boxes(nt);
  for (uint32_t t = 0; t < nt; t++) {
    Aabb b{{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
    for (int v = 0; v < 3; v++) for (int a = 0; a < 3; a++) { const float x = verts[(size_t) t * 12 + 4 * v + a]; b.lo[a] = std::min(b.lo[a], x); b.hi[a] = std::max(b.hi[a], x); }
    boxes[t] = b;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const std::string builder = argc > 3 ? argv[3] : "sah";
  Bvh4 bvh;
  if (builder == "sah") bvh = build_bvh4(boxes.data(), nt, kBvhLeafMaxTri, 26);
  else {
    std::vector<uint32_t> order;
    if (builder == "hybrid") {  // bvh_quality verts rays hybrid radius clusters
      std::vector<int> roots;
      const BinTree t = build_ploc_model(boxes, order, argc > 4 ? std::atoi(argv[4]) : 8, argc > 5 ? (size_t) std::atoi(argv[5]) : 4096, &roots);
      bvh = hybrid_model(t, order, roots, kBvhLeafMaxTri);
    }
    else {
    BinTree t = builder == "lbvh" ? build_lbvh_model(boxes, order) : build_ploc_model(boxes, order, argc > 4 ? std::atoi(argv[4]) : 16);
    if (std::getenv("BQ_REINSERT")) reinsert_model(t, std::atoi(std::getenv("BQ_REINSERT")));
    bvh = collapse_model(t, order, std::getenv("BQ_MAX_LEAF") ? (uint32_t) std::atoi(std::getenv("BQ_MAX_LEAF")) : kBvhLeafMaxTri);
    }
  }
  const double build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (bvh.nodes.empty()) { std::printf("build failed (too deep)\n"); return 1; }
  size_t kids = 0, leaf_refs = 0, leaf_count = 0;
  for (const Bvh4Node& n : bvh.nodes)
    for (int j = 0; j < 4; j++) if (n.child[j] != kBvhEmpty) { kids++; if (n.child[j] & kBvhLeafBit) { leaf_count++; leaf_refs += ((n.child[j] >> 28) & 7u) + 1u; } }
  std::printf("triangles %u  nodes %zu  children per node %.2f  leaves %zu  references %zu (%.3f per triangle)  depth %u  build %.2f s\n", nt, bvh.nodes.size(),
              (double) kids / bvh.nodes.size(), leaf_count, leaf_refs, (double) leaf_refs / nt, bvh.max_depth, build_s);
  Stats closest, shadow;
#pragma omp parallel
  {
    Stats c, s;
    std::mt19937 rng(1234u + 977u * (unsigned) (
#ifdef _OPENMP
        omp_get_thread_num()
#else
        0
#endif
        ));
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < (int64_t) nrays; i++) {
      const uint32_t t = (uint32_t) (U(rng) * nt) % nt;
      const float* p = &verts[(size_t) t * 12];
      const V p0{p[0], p[1], p[2]}, e1 = sub(V{p[4], p[5], p[6]}, p0), e2 = sub(V{p[8], p[9], p[10]}, p0);
      float u = U(rng), v = U(rng);
      if (u + v > 1) { u = 1 - u; v = 1 - v; }
      V n = norm(cross(e1, e2));
      if (U(rng) < 0.5f) n = mul(n, -1.0f);
      const V o = add(add(p0, add(mul(e1, u), mul(e2, v))), mul(n, 1e-3f));
      const float r1 = U(rng), r2 = U(rng), phi = 6.2831853f * r1, sr = std::sqrt(r2);
      const V tan = norm(std::fabs(n.x) < 0.9f ? cross(n, V{1, 0, 0}) : cross(n, V{0, 1, 0})), bit = cross(n, tan);
      const V d = norm(add(add(mul(tan, sr * std::cos(phi)), mul(bit, sr * std::sin(phi))), mul(n, std::sqrt(1 - r2))));
      walk(bvh, verts, o, d, INFINITY, false, c);
      walk(bvh, verts, o, d, INFINITY, true, s);
    }
#pragma omp critical
    {
      closest.nodes += c.nodes; closest.leaves += c.leaves; closest.tris += c.tris; closest.rays += c.rays; closest.hits += c.hits;
      shadow.nodes += s.nodes; shadow.leaves += s.leaves; shadow.tris += s.tris; shadow.rays += s.rays; shadow.hits += s.hits;
    }
  }
  if (std::getenv("BQ_WIDE")) {
    const std::vector<Wide> wide = widen(bvh);
    double kids8 = 0; for (const Wide& w : wide) kids8 += w.n;
    std::vector<Wide> slotted = wide;
    assign_slots(slotted);
    for (int mode = 0; mode < 3; mode++) {
      Stats ws;
#pragma omp parallel
      {
        Stats c;
        std::mt19937 rng(1234u + 977u * (unsigned) omp_get_thread_num());
        std::uniform_real_distribution<float> U(0.0f, 1.0f);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < (int64_t) nrays; i++) {
          const uint32_t t = (uint32_t) (U(rng) * nt) % nt;
          const float* p = &verts[(size_t) t * 12];
          const V p0{p[0], p[1], p[2]}, e1 = sub(V{p[4], p[5], p[6]}, p0), e2 = sub(V{p[8], p[9], p[10]}, p0);
          float u = U(rng), v = U(rng);
          if (u + v > 1) { u = 1 - u; v = 1 - v; }
          V n = norm(cross(e1, e2));
          if (U(rng) < 0.5f) n = mul(n, -1.0f);
          const V o = add(add(p0, add(mul(e1, u), mul(e2, v))), mul(n, 1e-3f));
          const float r1 = U(rng), r2 = U(rng), phi = 6.2831853f * r1, sr = std::sqrt(r2);
          const V tan = norm(std::fabs(n.x) < 0.9f ? cross(n, V{1, 0, 0}) : cross(n, V{0, 1, 0})), bit = cross(n, tan);
          const V d = norm(add(add(mul(tan, sr * std::cos(phi)), mul(bit, sr * std::sin(phi))), mul(n, std::sqrt(1 - r2))));
          if (mode == 2) walk_groups(slotted, bvh, verts, o, d, c);
          else walk_wide(wide, bvh, verts, o, d, mode == 1, c);
        }
#pragma omp critical
        { ws.nodes += c.nodes; ws.leaves += c.leaves; ws.tris += c.tris; ws.rays += c.rays; ws.hits += c.hits; }
      }
      std::printf("8-wide (%zu nodes, %.2f children per node), %s: nodes %.2f  leaves %.2f  triangles %.2f per closest-hit ray\n", wide.size(), kids8 / wide.size(),
                  mode == 2 ? "static octant slots, group stack, no distance cull at a pop" : mode ? "children in octant-diagonal order (no distances)" : "nearest first", ws.nodes / ws.rays, ws.leaves / ws.rays, ws.tris / ws.rays);
    }
  }
  std::printf("closest: nodes %.2f  leaves %.2f  triangles %.2f per ray (hit %.2f)\n", closest.nodes / closest.rays, closest.leaves / closest.rays, closest.tris / closest.rays, closest.hits / closest.rays);
  std::printf("any-hit: nodes %.2f  leaves %.2f  triangles %.2f per ray (hit %.2f)\n", shadow.nodes / shadow.rays, shadow.leaves / shadow.rays, shadow.tris / shadow.rays, shadow.hits / shadow.rays);
  return 0;
}
