#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <memory>
#include <numeric>
#include <thread>

namespace lum {
namespace {

struct BinNode {
  Aabb box;
  uint32_t left = 0, right = 0;  // children when count == 0
  uint32_t first = 0, count = 0;
};

inline void grow(Aabb& a, const Aabb& b) {
  for (int k = 0; k < 3; k++) { a.lo[k] = std::min(a.lo[k], b.lo[k]); a.hi[k] = std::max(a.hi[k], b.hi[k]); }
}
inline Aabb empty_box() { return Aabb{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}}; }
inline float half_area(const Aabb& b) {
  const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
  if (dx < 0.0f) return 0.0f;
  return dx * dy + dy * dz + dz * dx;
}

constexpr int kMaxBins = 64;
// Builder knobs for experiments (environment; defaults = what the measured trees were built with): LUM_BVH_BINS, LUM_BVH_MAX_LEAF,
// LUM_BVH_SAH_LEAF=<traversal cost>: a set that fits a leaf is still split when the SAH says the split is cheaper than testing all its triangles.
int env_int(const char* name, int def) { const char* e = std::getenv(name); return e ? std::atoi(e) : def; }
float env_float(const char* name, float def) { const char* e = std::getenv(name); return e ? (float) std::atof(e) : def; }

// Work over [0, n) cut into one contiguous chunk per thread; fn(begin, end, thread). Threads are started per call: the top of a 10 M-triangle
// tree has a few hundred calls, each worth milliseconds.
inline unsigned build_threads() {
  static const unsigned n = [] {
    const int e = env_int("LUM_BVH_THREADS", 0);
    if (e > 0) return (unsigned) std::min(e, 256);
    const unsigned hc = std::thread::hardware_concurrency();
    return std::min(std::max(hc, 1u), 64u);
  }();
  return n;
}
template <class F>
void parallel_chunks(size_t n, unsigned threads, F&& fn) {
  threads = (unsigned) std::min<size_t>(std::max<size_t>(threads, 1), std::max<size_t>(n / 65536, 1));
  if (threads <= 1) { fn((size_t) 0, n, 0u); return; }
  std::vector<std::thread> pool;
  pool.reserve(threads - 1);
  const size_t chunk = (n + threads - 1) / threads;
  for (unsigned t = 1; t < threads; t++) pool.emplace_back([&, t] { const size_t b = std::min(n, t * chunk), e = std::min(n, b + chunk); if (b < e) fn(b, e, t); });
  fn((size_t) 0, std::min(n, chunk), 0u);
  for (auto& th : pool) th.join();
}

// Binned-SAH builder. The primitives travel with their boxes (one 28-byte item each, partitioned in place), so every pass over a node's set is a
// sequential read - with an index array into the caller's boxes the passes over the top of a 10 M-triangle tree were cache misses throughout.
// Large nodes are split with all threads (binning and a stable partition, both deterministic: boxes by min / max, counts as integers), the
// subtrees below them are built by one thread each; the tree does not depend on the number of threads.
struct Item { float lo[3], hi[3]; uint32_t prim; };

struct Builder {
  std::vector<Item> items, scratch;
  std::vector<uint32_t> order;   // filled at the end: items[i].prim
  std::vector<BinNode> nodes;
  bool balanced = false;
  uint32_t max_leaf = 4;
  int kBins = 16;
  float sah_leaf_traversal_cost = -1.0f;  // < 0: every set that fits a leaf becomes one

  static float centre(const Item& it, int a) { return 0.5f * (it.lo[a] + it.hi[a]); }

  struct SetBounds { Aabb box; float clo[3], chi[3]; };
  static void merge(SetBounds& a, const SetBounds& b) {
    grow(a.box, b.box);
    for (int k = 0; k < 3; k++) { a.clo[k] = std::min(a.clo[k], b.clo[k]); a.chi[k] = std::max(a.chi[k], b.chi[k]); }
  }
  SetBounds bounds_of(uint32_t first, uint32_t count, unsigned threads) const {
    std::vector<SetBounds> part(std::max(threads, 1u));
    for (auto& p : part) { p.box = empty_box(); for (int k = 0; k < 3; k++) { p.clo[k] = FLT_MAX; p.chi[k] = -FLT_MAX; } }
    parallel_chunks(count, threads, [&](size_t b, size_t e, unsigned t) {
      SetBounds s = part[t];
      for (size_t i = first + b; i < first + e; i++) {
        const Item& it = items[i];
        for (int k = 0; k < 3; k++) {
          s.box.lo[k] = std::min(s.box.lo[k], it.lo[k]); s.box.hi[k] = std::max(s.box.hi[k], it.hi[k]);
          const float c = centre(it, k);
          s.clo[k] = std::min(s.clo[k], c); s.chi[k] = std::max(s.chi[k], c);
        }
      }
      part[t] = s;
    });
    SetBounds out = part[0];
    for (size_t t = 1; t < part.size(); t++) merge(out, part[t]);
    return out;
  }

  struct Bins { Aabb box[3][kMaxBins]; uint32_t cnt[3][kMaxBins]; };
  void bin_all(uint32_t first, uint32_t count, const SetBounds& sb, unsigned threads, Bins& out) const {
    std::vector<Bins> part(std::max(threads, 1u));
    float scale[3];
    for (int a = 0; a < 3; a++) { const float ext = sb.chi[a] - sb.clo[a]; scale[a] = ext > 0.0f ? kBins / ext : 0.0f; }
    for (auto& p : part) for (int a = 0; a < 3; a++) for (int b = 0; b < kBins; b++) { p.box[a][b] = empty_box(); p.cnt[a][b] = 0; }
    parallel_chunks(count, threads, [&](size_t b0, size_t e0, unsigned t) {
      Bins& my = part[t];
      for (size_t i = first + b0; i < first + e0; i++) {
        const Item& it = items[i];
        Aabb ib; std::memcpy(ib.lo, it.lo, 12); std::memcpy(ib.hi, it.hi, 12);
        for (int a = 0; a < 3; a++) {
          if (!(scale[a] > 0.0f)) continue;
          int b = (int) ((centre(it, a) - sb.clo[a]) * scale[a]);
          b = std::min(std::max(b, 0), kBins - 1);
          grow(my.box[a][b], ib); my.cnt[a][b]++;
        }
      }
    });
    out = part[0];
    for (size_t t = 1; t < part.size(); t++)
      for (int a = 0; a < 3; a++) for (int b = 0; b < kBins; b++) { grow(out.box[a][b], part[t].box[a][b]); out.cnt[a][b] += part[t].cnt[a][b]; }
  }

  // items of [first, first + count) with pred first, both sides in their old order; returns the number of them
  template <class P>
  uint32_t stable_partition(uint32_t first, uint32_t count, unsigned threads, P&& pred) {
    if (scratch.size() < items.size()) scratch.resize(items.size());
    threads = std::max(threads, 1u);
    std::vector<uint32_t> yes(threads + 1, 0);
    parallel_chunks(count, threads, [&](size_t b, size_t e, unsigned t) {
      uint32_t c = 0;
      for (size_t i = first + b; i < first + e; i++) c += pred(items[i]) ? 1u : 0u;
      yes[t + 1] = c;
    });
    // chunk boundaries as parallel_chunks cuts them (it may use fewer threads than asked for on small sets: unused slots stay 0)
    for (unsigned t = 0; t < threads; t++) yes[t + 1] += yes[t];
    const uint32_t total_yes = yes[threads];
    std::vector<size_t> begin_of(threads + 1, 0);
    parallel_chunks(count, threads, [&](size_t b, size_t e, unsigned t) { begin_of[t] = b; (void) e; });
    parallel_chunks(count, threads, [&](size_t b, size_t e, unsigned t) {
      uint32_t y = yes[t], n = total_yes + (uint32_t) b - yes[t];
      for (size_t i = first + b; i < first + e; i++) { if (pred(items[i])) scratch[first + y++] = items[i]; else scratch[first + n++] = items[i]; }
    });
    parallel_chunks(count, threads, [&](size_t b, size_t e, unsigned) { std::memcpy(&items[first + b], &scratch[first + b], (e - b) * sizeof(Item)); });
    return total_yes;
  }

  // Decides what happens to the set: leaf (returns false) or the split position `mid` (items partitioned in place).
  bool split(uint32_t first, uint32_t count, int depth, unsigned threads, Aabb& box, uint32_t& mid) {
    const SetBounds sb = bounds_of(first, count, threads);
    box = sb.box;
    const bool fits_leaf = count <= max_leaf;
    if (fits_leaf && (count <= 1 || sah_leaf_traversal_cost < 0.0f || balanced)) return false;
    int axis = 0;
    if (sb.chi[1] - sb.clo[1] > sb.chi[axis] - sb.clo[axis]) axis = 1;
    if (sb.chi[2] - sb.clo[2] > sb.chi[axis] - sb.clo[axis]) axis = 2;
    mid = first + count / 2;
    bool use_median = balanced || depth > 40 || !(sb.chi[axis] > sb.clo[axis]);
    if (!use_median) {
      Bins bins;
      bin_all(first, count, sb, threads, bins);
      float best_cost = FLT_MAX;
      int best_axis = -1, best_bin = -1;
      for (int a = 0; a < 3; a++) {
        if (!(sb.chi[a] - sb.clo[a] > 0.0f)) continue;
        float right_area[kMaxBins];
        uint32_t right_cnt[kMaxBins];
        Aabb acc = empty_box();
        uint32_t c = 0;
        for (int b = kBins - 1; b > 0; b--) { grow(acc, bins.box[a][b]); c += bins.cnt[a][b]; right_area[b] = half_area(acc); right_cnt[b] = c; }
        acc = empty_box(); c = 0;
        for (int b = 0; b < kBins - 1; b++) {
          grow(acc, bins.box[a][b]); c += bins.cnt[a][b];
          if (c == 0 || right_cnt[b + 1] == 0) continue;
          const float cost = half_area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
        }
      }
      if (fits_leaf) {  // SAH leaf termination: split only if cheaper than testing the whole set
        const float area = half_area(box);
        if (best_axis < 0 || !(area > 0.0f) || sah_leaf_traversal_cost + best_cost / area >= (float) count) return false;
      }
      if (best_axis < 0) use_median = true;
      else {
        const float ext = sb.chi[best_axis] - sb.clo[best_axis], scale = kBins / ext, lo = sb.clo[best_axis];
        const uint32_t left = stable_partition(first, count, threads, [&](const Item& it) {
          int b = (int) ((centre(it, best_axis) - lo) * scale);
          b = std::min(std::max(b, 0), kBins - 1);
          return b <= best_bin;
        });
        mid = first + left;
        if (mid == first || mid == first + count) use_median = true;
      }
    }
    if (use_median) {
      mid = first + count / 2;
      std::nth_element(items.begin() + first, items.begin() + mid, items.begin() + first + count, [&](const Item& x, const Item& y) {
        const float cx = centre(x, axis), cy = centre(y, axis);
        return cx < cy || (cx == cy && x.prim < y.prim);
      });
    }
    return true;
  }

  // one thread, depth first, into `out` (indices relative to `out`)
  uint32_t build_serial(uint32_t first, uint32_t count, int depth, std::vector<BinNode>& out) {
    const uint32_t idx = (uint32_t) out.size();
    out.emplace_back();
    Aabb box; uint32_t mid = 0;
    const bool inner = split(first, count, depth, 1u, box, mid);
    out[idx].box = box;
    out[idx].first = first;
    if (!inner) { out[idx].count = count; return idx; }
    const uint32_t l = build_serial(first, mid - first, depth + 1, out);
    const uint32_t r = build_serial(mid, first + count - mid, depth + 1, out);
    out[idx].left = l; out[idx].right = r;
    return idx;
  }

  void build(uint32_t count) {
    const unsigned threads = build_threads();
    nodes.clear();
    constexpr uint32_t kTaskSize = 1u << 17;  // sets of at most this many primitives become one thread's subtree
    if (threads <= 1 || count <= kTaskSize) { build_serial(0, count, 0, nodes); }
    else {
      struct Task { uint32_t first, count; int depth; uint32_t parent; bool is_left; std::vector<BinNode> sub; };
      std::vector<Task> tasks;
      struct Open { uint32_t node, first, count; int depth; };
      std::vector<Open> open;
      nodes.emplace_back();
      open.push_back({0u, 0u, count, 0});
      for (size_t h = 0; h < open.size(); h++) {  // the top of the tree, every split with all threads
        const Open o = open[h];
        Aabb box; uint32_t mid = 0;
        const bool inner = split(o.first, o.count, o.depth, threads, box, mid);
        nodes[o.node].box = box; nodes[o.node].first = o.first;
        if (!inner) { nodes[o.node].count = o.count; continue; }
        const uint32_t halves[2][2] = {{o.first, mid - o.first}, {mid, o.first + o.count - mid}};
        for (int side = 0; side < 2; side++) {
          if (halves[side][1] > kTaskSize) {
            const uint32_t id = (uint32_t) nodes.size();
            nodes.emplace_back();
            (side == 0 ? nodes[o.node].left : nodes[o.node].right) = id;
            open.push_back({id, halves[side][0], halves[side][1], o.depth + 1});
          }
          else tasks.push_back(Task{halves[side][0], halves[side][1], o.depth + 1, o.node, side == 0, {}});
        }
      }
      std::atomic<size_t> next{0};
      auto worker = [&] { for (size_t t = next.fetch_add(1); t < tasks.size(); t = next.fetch_add(1)) build_serial(tasks[t].first, tasks[t].count, tasks[t].depth, tasks[t].sub); };
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < threads; t++) pool.emplace_back(worker);
      worker();
      for (auto& th : pool) th.join();
      for (Task& t : tasks) {  // subtrees behind the top nodes, in task order
        const uint32_t base = (uint32_t) nodes.size();
        for (BinNode n : t.sub) { if (n.count == 0) { n.left += base; n.right += base; } nodes.push_back(n); }
        (t.is_left ? nodes[t.parent].left : nodes[t.parent].right) = base;
      }
    }
    order.resize(count);
    for (uint32_t i = 0; i < count; i++) order[i] = items[i].prim;
    std::vector<Item>().swap(scratch);
  }
};

// ---- insertion-based optimisation of the finished binary tree (Bittner, Hapala, Havran: "Fast Insertion-Based Optimization of Bounding Volume
// Hierarchies", CGF 2013) ----
// A top-down builder decides every split with what it sees at that node; the result is locally good and globally improvable. One step takes a subtree
// out (its parent node is removed, the sibling moves up, the ancestors' boxes shrink), looks - branch and bound over the whole tree, cheapest
// enlargement first - for the place where putting it back enlarges the tree's boxes least, and inserts it there with the freed parent node. The cost
// that falls is the sum of the inner nodes' surface areas, which is what the expected number of node visits of a random ray is proportional to.
// Leaves (sets of at most max_leaf primitives) are moved whole: the primitive order, and with it the leaf ranges, stay as built.
struct Reinserter {
  std::vector<BinNode>& nodes;
  std::vector<uint32_t> parent;
  explicit Reinserter(std::vector<BinNode>& n) : nodes(n), parent(n.size(), 0xFFFFFFFFu) {
    for (uint32_t i = 0; i < nodes.size(); i++) if (nodes[i].count == 0) { parent[nodes[i].left] = i; parent[nodes[i].right] = i; }
  }
  static Aabb join(const Aabb& a, const Aabb& b) { Aabb r = a; grow(r, b); return r; }
  bool leaf(uint32_t i) const { return nodes[i].count > 0; }
  void refit_up(uint32_t i) {
    while (i != 0xFFFFFFFFu) {
      const Aabb b = join(nodes[nodes[i].left].box, nodes[nodes[i].right].box);
      if (std::memcmp(&b, &nodes[i].box, sizeof(Aabb)) == 0) break;
      nodes[i].box = b;
      i = parent[i];
    }
  }
  double inner_area() const { double a = 0; for (const BinNode& n : nodes) if (n.count == 0) a += half_area(n.box); return a; }

  // Takes `n` out and puts it back where the tree grows least. Returns true when the tree changed.
  bool reinsert(uint32_t n, std::vector<std::pair<float, uint32_t>>& heap) {
    const uint32_t p = parent[n];
    if (p == 0xFFFFFFFFu || p == 0u) return false;  // the root and its children stay (the root keeps index 0)
    const uint32_t g = parent[p];
    const uint32_t s = nodes[p].left == n ? nodes[p].right : nodes[p].left;
    // remove: s takes p's place
    (nodes[g].left == p ? nodes[g].left : nodes[g].right) = s;
    parent[s] = g;
    refit_up(g);
    // search
    const Aabb nb = nodes[n].box;
    const float na = half_area(nb);
    float best_cost = FLT_MAX;
    uint32_t best = s;
    heap.clear();
    heap.emplace_back(0.0f, 0u);
    auto cmp = [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first > b.first; };  // min-heap on the induced cost
    while (!heap.empty()) {
      std::pop_heap(heap.begin(), heap.end(), cmp);
      const float induced = heap.back().first;
      const uint32_t x = heap.back().second;
      heap.pop_back();
      if (induced + na >= best_cost) break;
      const float direct = half_area(join(nodes[x].box, nb));
      const float total = induced + direct;
      if (total < best_cost) { best_cost = total; best = x; }
      if (!leaf(x)) {
        const float child_induced = total - half_area(nodes[x].box);
        if (child_induced + na < best_cost) {
          heap.emplace_back(child_induced, nodes[x].left); std::push_heap(heap.begin(), heap.end(), cmp);
          heap.emplace_back(child_induced, nodes[x].right); std::push_heap(heap.begin(), heap.end(), cmp);
        }
      }
    }
    // insert: p becomes the parent of (best, n) where best was
    const uint32_t bp = parent[best];
    if (bp == 0xFFFFFFFFu) {  // best is the root: keep index 0 as the root by moving the root's content into p... simpler: insert below the root's nearer child instead
      // (cannot happen for a subtree that was not a child of the root unless it is huge; fall back to the old place)
      best = s;
    }
    const uint32_t bp2 = parent[best];
    (nodes[bp2].left == best ? nodes[bp2].left : nodes[bp2].right) = p;
    parent[p] = bp2;
    nodes[p].left = best; nodes[p].right = n; nodes[p].count = 0;
    parent[best] = p; parent[n] = p;
    nodes[p].box = join(nodes[best].box, nb);
    refit_up(bp2);
    return best != s;
  }

  void run(int passes, float fraction) {
    std::vector<std::pair<float, uint32_t>> heap;
    std::vector<uint32_t> order(nodes.size());
    for (int pass = 0; pass < passes; pass++) {
      std::iota(order.begin(), order.end(), 0u);
      std::vector<float> key(nodes.size());
      for (uint32_t i = 0; i < nodes.size(); i++) key[i] = half_area(nodes[i].box);
      std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return key[a] > key[b] || (key[a] == key[b] && a < b); });
      const size_t take = (size_t) (fraction * order.size());
      size_t moved = 0;
      const double before = inner_area();
      for (size_t k = 0; k < take; k++) moved += reinsert(order[k], heap) ? 1 : 0;
      if (env_int("LUM_BVH_TIMING", 0)) std::fprintf(stderr, "[bvh] reinsertion pass %d: %zu of %zu subtrees moved, inner area %.6g -> %.6g\n", pass, moved, take, before, inner_area());
    }
  }
};

// ---- builder with spatial splits (bvh_build.h build_bvh4_triangles) ----
struct Ref { uint32_t prim; Aabb box; };

inline Aabb intersect_boxes(const Aabb& a, const Aabb& b) {
  Aabb r;
  for (int k = 0; k < 3; k++) { r.lo[k] = std::max(a.lo[k], b.lo[k]); r.hi[k] = std::min(a.hi[k], b.hi[k]); }
  return r;
}
inline bool valid_box(const Aabb& b) { return b.lo[0] <= b.hi[0] && b.lo[1] <= b.hi[1] && b.lo[2] <= b.hi[2]; }

// Bounds of the part of triangle `p` (3 x float4) between the planes x_axis = lo and x_axis = hi, intersected with `within`.
inline Aabb clip_triangle(const float* p, int axis, float lo, float hi, const Aabb& within) {
  Aabb out = empty_box();
  for (int e = 0; e < 3; e++) {
    const float* a = p + 4 * e;
    const float* b = p + 4 * ((e + 1) % 3);
    const float ta = a[axis], tb = b[axis];
    if (ta >= lo && ta <= hi) for (int k = 0; k < 3; k++) { out.lo[k] = std::min(out.lo[k], a[k]); out.hi[k] = std::max(out.hi[k], a[k]); }
    const float planes[2] = {lo, hi};
    for (int s = 0; s < 2; s++) {
      const float pl = planes[s];
      if ((ta < pl && tb > pl) || (ta > pl && tb < pl)) {
        const float t = (pl - ta) / (tb - ta);
        for (int k = 0; k < 3; k++) {
          const float x = (k == axis) ? pl : a[k] + t * (b[k] - a[k]);
          out.lo[k] = std::min(out.lo[k], x); out.hi[k] = std::max(out.hi[k], x);
        }
      }
    }
  }
  // rounding of the interpolated points must not push the piece outside the reference it came from or outside the slab
  out.lo[axis] = std::max(out.lo[axis], lo); out.hi[axis] = std::min(out.hi[axis], hi);
  return intersect_boxes(out, within);
}

struct SplitBuilder {
  const float* vertices;
  const uint8_t* splittable;
  std::vector<uint32_t> order;      // primitive ids in leaf order, one per reference
  std::vector<BinNode> nodes;
  uint32_t max_leaf = 4;
  int kBins = 16, kSpatialBins = 32;
  float root_area = 1.0f;
  float alpha = 1e-5f;              // spatial splits are tried when overlap area / root area exceeds this (Stich et al., section 4.5)
  size_t budget = 0;                // references the build may still add
  float leaf_visit_cost = 0.0f;

  uint32_t make_leaf(uint32_t idx, const std::vector<Ref>& refs) {
    nodes[idx].first = (uint32_t) order.size();
    nodes[idx].count = (uint32_t) refs.size();
    for (const Ref& r : refs) order.push_back(r.prim);
    return idx;
  }

  uint32_t build(std::vector<Ref>& refs, int depth) {
    const uint32_t idx = (uint32_t) nodes.size();
    nodes.emplace_back();
    Aabb box = empty_box(), cbox = empty_box();
    for (const Ref& r : refs) {
      grow(box, r.box);
      for (int k = 0; k < 3; k++) { const float c = 0.5f * (r.box.lo[k] + r.box.hi[k]); cbox.lo[k] = std::min(cbox.lo[k], c); cbox.hi[k] = std::max(cbox.hi[k], c); }
    }
    nodes[idx].box = box;
    const uint32_t count = (uint32_t) refs.size();
    if (count <= max_leaf) return make_leaf(idx, refs);

    // ---- object split: binned SAH over the references' centroids ----
    float obj_cost = FLT_MAX; int obj_axis = -1, obj_bin = -1; Aabb obj_left = empty_box(), obj_right = empty_box();
    for (int a = 0; a < 3; a++) {
      const float ext = cbox.hi[a] - cbox.lo[a];
      if (!(ext > 0.0f)) continue;
      Aabb bin_box[kMaxBins]; uint32_t bin_cnt[kMaxBins];
      for (int b = 0; b < kBins; b++) { bin_box[b] = empty_box(); bin_cnt[b] = 0; }
      const float scale = kBins / ext;
      for (const Ref& r : refs) {
        int b = (int) ((0.5f * (r.box.lo[a] + r.box.hi[a]) - cbox.lo[a]) * scale);
        b = std::min(std::max(b, 0), kBins - 1);
        grow(bin_box[b], r.box); bin_cnt[b]++;
      }
      Aabb right_box[kMaxBins]; uint32_t right_cnt[kMaxBins];
      Aabb acc = empty_box(); uint32_t c = 0;
      for (int b = kBins - 1; b > 0; b--) { grow(acc, bin_box[b]); c += bin_cnt[b]; right_box[b] = acc; right_cnt[b] = c; }
      acc = empty_box(); c = 0;
      for (int b = 0; b < kBins - 1; b++) {
        grow(acc, bin_box[b]); c += bin_cnt[b];
        if (c == 0 || right_cnt[b + 1] == 0) continue;
        const float cost = half_area(acc) * c + half_area(right_box[b + 1]) * right_cnt[b + 1];
        if (cost < obj_cost) { obj_cost = cost; obj_axis = a; obj_bin = b; obj_left = acc; obj_right = right_box[b + 1]; }
      }
    }

    // ---- spatial split: only where the object split's children overlap noticeably, and while the reference budget lasts ----
    float sp_cost = FLT_MAX; int sp_axis = -1; float sp_plane = 0.0f;
    bool try_spatial = budget > 0 && depth < 48;
    if (try_spatial && obj_axis >= 0) {
      const Aabb ov = intersect_boxes(obj_left, obj_right);
      try_spatial = valid_box(ov) && half_area(ov) > alpha * root_area;
    }
    if (try_spatial) {
      for (int a = 0; a < 3; a++) {
        const float ext = box.hi[a] - box.lo[a];
        if (!(ext > 0.0f)) continue;
        Aabb bin_box[kMaxBins]; uint32_t enter[kMaxBins], leave[kMaxBins];
        for (int b = 0; b < kSpatialBins; b++) { bin_box[b] = empty_box(); enter[b] = leave[b] = 0; }
        const float scale = kSpatialBins / ext, width = ext / kSpatialBins;
        for (const Ref& r : refs) {
          int b0 = std::min(std::max((int) ((r.box.lo[a] - box.lo[a]) * scale), 0), kSpatialBins - 1);
          int b1 = std::min(std::max((int) ((r.box.hi[a] - box.lo[a]) * scale), 0), kSpatialBins - 1);
          if (!splittable[r.prim]) {  // kept whole: counted where its centroid lies, with its whole box
            const int bc = std::min(std::max((int) ((0.5f * (r.box.lo[a] + r.box.hi[a]) - box.lo[a]) * scale), 0), kSpatialBins - 1);
            grow(bin_box[bc], r.box); enter[bc]++; leave[bc]++;
            continue;
          }
          if (b0 == b1) grow(bin_box[b0], r.box);
          else {
            const float* p = vertices + (size_t) r.prim * 12;
            for (int b = b0; b <= b1; b++) {
              const Aabb piece = clip_triangle(p, a, box.lo[a] + b * width, (b == kSpatialBins - 1) ? box.hi[a] : box.lo[a] + (b + 1) * width, r.box);
              if (valid_box(piece)) grow(bin_box[b], piece);
            }
          }
          enter[b0]++; leave[b1]++;
        }
        Aabb right_box[kMaxBins]; uint32_t right_cnt[kMaxBins];
        Aabb acc = empty_box(); uint32_t c = 0;
        for (int b = kSpatialBins - 1; b > 0; b--) { grow(acc, bin_box[b]); c += leave[b]; right_box[b] = acc; right_cnt[b] = c; }
        acc = empty_box(); c = 0;
        for (int b = 0; b < kSpatialBins - 1; b++) {
          grow(acc, bin_box[b]); c += enter[b];
          if (c == 0 || right_cnt[b + 1] == 0) continue;
          const float cost = half_area(acc) * c + half_area(right_box[b + 1]) * right_cnt[b + 1];
          if (cost < sp_cost) { sp_cost = cost; sp_axis = a; sp_plane = box.lo[a] + (b + 1) * width; }
        }
      }
    }

    std::vector<Ref> left, right;
    if (sp_axis >= 0 && sp_cost < obj_cost) {
      left.reserve(count); right.reserve(count);
      for (const Ref& r : refs) {
        const bool whole = !splittable[r.prim];
        if (r.box.hi[sp_axis] <= sp_plane || (whole && 0.5f * (r.box.lo[sp_axis] + r.box.hi[sp_axis]) <= sp_plane)) left.push_back(r);
        else if (r.box.lo[sp_axis] >= sp_plane || whole) right.push_back(r);
        else {
          const float* p = vertices + (size_t) r.prim * 12;
          const Aabb lb = clip_triangle(p, sp_axis, r.box.lo[sp_axis], sp_plane, r.box), rb = clip_triangle(p, sp_axis, sp_plane, r.box.hi[sp_axis], r.box);
          const bool lv = valid_box(lb), rv = valid_box(rb);
          if (lv) left.push_back(Ref{r.prim, lb});
          if (rv) right.push_back(Ref{r.prim, rb});
          if (!lv && !rv) left.push_back(r);
          if (lv && rv && budget > 0) budget--;
        }
      }
      if (left.empty() || right.empty() || left.size() == count || right.size() == count) { left.clear(); right.clear(); }  // no progress: fall back to the object split
    }
    if (left.empty()) {
      if (obj_axis >= 0) {
        const float ext = cbox.hi[obj_axis] - cbox.lo[obj_axis], scale = kBins / ext, lo = cbox.lo[obj_axis];
        for (const Ref& r : refs) {
          int b = (int) ((0.5f * (r.box.lo[obj_axis] + r.box.hi[obj_axis]) - lo) * scale);
          b = std::min(std::max(b, 0), kBins - 1);
          (b <= obj_bin ? left : right).push_back(r);
        }
      }
      if (left.empty() || right.empty()) {  // coincident centroids: halve the list
        left.assign(refs.begin(), refs.begin() + count / 2);
        right.assign(refs.begin() + count / 2, refs.end());
      }
    }
    std::vector<Ref>().swap(refs);  // the parent's list is not needed below this point
    const uint32_t l = build(left, depth + 1);
    const uint32_t r = build(right, depth + 1);
    nodes[idx].left = l; nodes[idx].right = r;
    return idx;
  }
};

inline void set_child_box(Bvh4Node& n, int k, const Aabb& b) {
  // conservative padding: a triangle accepted by the exact intersection test must never be culled by box rounding
  float lo[3], hi[3];
  for (int a = 0; a < 3; a++) {
    const float pad = 1e-5f * std::max(std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])), 1e-20f) + 1e-30f;
    lo[a] = b.lo[a] - pad; hi[a] = b.hi[a] + pad;
  }
  n.lo_x[k] = lo[0]; n.lo_y[k] = lo[1]; n.lo_z[k] = lo[2];
  n.hi_x[k] = hi[0]; n.hi_y[k] = hi[1]; n.hi_z[k] = hi[2];
}

// Which binary nodes become 4-wide nodes. The greedy rule (rounds 1-5) opens the child with the largest box until four are reached - or until only leaves
// are left, which on the hall gave 3.03 children per node: 48 % more nodes than a full 4-ary tree over the same leaves. The optimal rule (round 6;
// the dynamic programme of Wald, Benthin, Boulos 2008 / Ylitie, Karras, Laine 2017 for wide nodes): a visit costs the same whatever the number of
// occupied slots, and a node is visited in proportion to its box area, so the cost of a 4-wide tree is the sum of its inner nodes' areas; T(n) = area(n) +
// the cheapest cut of n's binary subtree into at most four pieces, a piece being a leaf (free: the leaf sets are fixed) or a binary node m at T(m).
// F[n][k - 1] = the cheapest way to hand n's subtree to a parent that has k slots for it.
struct CollapsePlan {
  std::vector<float> F;  // 4 per binary node
  template <class B>
  explicit CollapsePlan(const B& b) : F(b.nodes.size() * 4, 0.0f) {
    // children before parents: an explicit post-order (the reinsertion pass may have moved subtrees, so indices say nothing)
    std::vector<uint32_t> order; order.reserve(b.nodes.size());
    std::vector<uint32_t> stack{0u};
    while (!stack.empty()) {
      const uint32_t n = stack.back(); stack.pop_back();
      order.push_back(n);
      if (b.nodes[n].count == 0) { stack.push_back(b.nodes[n].left); stack.push_back(b.nodes[n].right); }
    }
    for (size_t i = order.size(); i-- > 0;) {
      const uint32_t n = order[i];
      const BinNode& bn = b.nodes[n];
      if (bn.count > 0) continue;  // leaves: 0 for every k
      const float* L = &F[(size_t) bn.left * 4];
      const float* R = &F[(size_t) bn.right * 4];
      float G[5] = {0, 0, FLT_MAX, FLT_MAX, FLT_MAX};  // G[k]: n's two subtrees cut into at most k pieces
      for (int k = 2; k <= 4; k++)
        for (int a = 1; a < k; a++) G[k] = std::min(G[k], L[a - 1] + R[k - a - 1]);
      float* f = &F[(size_t) n * 4];
      f[0] = half_area(bn.box) + G[4];
      for (int k = 2; k <= 4; k++) f[k - 1] = std::min(f[0], G[k]);
    }
  }
  // the pieces of binary node n's subtree for a parent with k slots, appended to kids
  template <class B>
  void cut(const B& b, uint32_t n, int k, uint32_t* kids, int& nk) const {
    const BinNode& bn = b.nodes[n];
    const float* f = &F[(size_t) n * 4];
    if (bn.count > 0 || k == 1 || f[k - 1] == f[0]) { kids[nk++] = n; return; }  // (ties: the node stays whole - fewer, fuller nodes)
    split(b, n, k, kids, nk);
  }
  template <class B>
  void split(const B& b, uint32_t n, int k, uint32_t* kids, int& nk) const {
    const BinNode& bn = b.nodes[n];
    const float* L = &F[(size_t) bn.left * 4];
    const float* R = &F[(size_t) bn.right * 4];
    int best_a = 1; float best = FLT_MAX;
    for (int a = 1; a < k; a++) { const float c = L[a - 1] + R[k - a - 1]; if (c < best) { best = c; best_a = a; } }
    cut(b, bn.left, best_a, kids, nk);
    cut(b, bn.right, k - best_a, kids, nk);
  }
};

// Test support: the cheapest 4-wide tree over the binary tree by exhaustion - every subset of the inner nodes (the root always) is tried as the set of surviving
// nodes; a surviving node's children are the nearest surviving nodes or leaves below it, and the subset is valid when none has more than four. Shares nothing
// with CollapsePlan but the cost's definition. -1 when the tree has more than 20 inner nodes.
template <class B>
double brute_force_collapse_cost(const B& b) {
  std::vector<uint32_t> inner;
  for (uint32_t i = 0; i < b.nodes.size(); i++) if (b.nodes[i].count == 0) inner.push_back(i);
  // (nodes that cannot be reached from the root do not exist in the builders' arrays)
  if (inner.empty()) return 0.0;
  if (inner.size() > 20 || inner[0] != 0u) return -1.0;
  std::vector<int> slot(b.nodes.size(), -1);
  for (size_t k = 0; k < inner.size(); k++) slot[inner[k]] = (int) k;
  double best = -1.0;
  const uint32_t subsets = 1u << (inner.size() - 1);
  for (uint32_t mask = 0; mask < subsets; mask++) {
    const uint32_t alive = (mask << 1) | 1u;  // bit k: inner[k] survives; the root does
    bool valid = true;
    double cost = 0.0;
    for (size_t k = 0; k < inner.size() && valid; k++) {
      if (!((alive >> k) & 1u)) continue;
      cost += (double) half_area(b.nodes[inner[k]].box);
      uint32_t children = 0;
      std::vector<uint32_t> st{b.nodes[inner[k]].left, b.nodes[inner[k]].right};
      while (!st.empty()) {
        const uint32_t n = st.back(); st.pop_back();
        if (b.nodes[n].count > 0 || ((alive >> slot[n]) & 1u)) children++;
        else { st.push_back(b.nodes[n].left); st.push_back(b.nodes[n].right); }
      }
      valid = children <= 4;
    }
    if (valid && (best < 0.0 || cost < best)) best = cost;
  }
  return best;
}

template <class B>
Bvh4 collapse(const B& b) {
  Bvh4 out;
  out.prims = b.order;
  if (b.nodes.empty()) return out;
  const bool optimal = env_int("LUM_BVH_COLLAPSE", 1) != 0;  // 0: the greedy rule of rounds 1-5
  std::unique_ptr<CollapsePlan> plan;
  if (optimal) {
    plan.reset(new CollapsePlan(b));
    out.plan_cost = b.nodes[0].count > 0 ? 0.0 : (double) plan->F[0];
    if (env_int("LUM_BVH_COLLAPSE_BRUTE", 0)) out.brute_cost = brute_force_collapse_cost(b);
  }
  struct Job { uint32_t bin, node4; uint32_t depth; };
  std::vector<Job> jobs;
  out.nodes.emplace_back();
  jobs.push_back({0, 0, 1});
  for (size_t j = 0; j < jobs.size(); j++) {
    const Job job = jobs[j];
    out.max_depth = std::max(out.max_depth, job.depth);
    uint32_t kids[4];
    int nk = 0;
    const BinNode& root = b.nodes[job.bin];
    if (root.count > 0) kids[nk++] = job.bin;  // whole set fits one leaf: root with a single leaf child
    else if (optimal) plan->split(b, job.bin, 4, kids, nk);
    else { kids[nk++] = root.left; kids[nk++] = root.right; }
    while (!optimal && nk < 4) {
      int pick = -1; float best = -1.0f;
      for (int k = 0; k < nk; k++) {
        const BinNode& c = b.nodes[kids[k]];
        if (c.count > 0) continue;
        const float a = half_area(c.box);
        if (a > best) { best = a; pick = k; }
      }
      if (pick < 0) break;
      const BinNode c = b.nodes[kids[pick]];
      kids[pick] = c.left;
      kids[nk++] = c.right;
    }
    Bvh4Node n;
    std::memset(&n, 0, sizeof(n));
    for (int k = 0; k < 4; k++) {
      n.child[k] = kBvhEmpty;
      n.lo_x[k] = n.lo_y[k] = n.lo_z[k] = FLT_MAX;
      n.hi_x[k] = n.hi_y[k] = n.hi_z[k] = -FLT_MAX;
    }
    for (int k = 0; k < nk; k++) {
      const BinNode& c = b.nodes[kids[k]];
      set_child_box(n, k, c.box);
      if (c.count > 0) n.child[k] = kBvhLeafBit | ((c.count - 1) << 28) | c.first;
      else {
        const uint32_t id = (uint32_t) out.nodes.size();
        out.nodes.emplace_back();
        n.child[k] = id;
        jobs.push_back({kids[k], id, job.depth + 1});
      }
    }
    out.nodes[job.node4] = n;
  }
  return out;
}

}  // namespace

Bvh4 build_bvh4_triangles(const float* vertices, const Aabb* boxes, const uint8_t* splittable, uint32_t count, uint32_t max_leaf, uint32_t max_depth) {
  // Off unless asked for (LUM_BVH_SPATIAL=1). Measured with tools/bvh_quality.cpp on the benchmark hall (1.43 M evenly tessellated triangles): 1.5 % of
  // the triangles get a second reference, node visits per closest-hit ray 17.50 -> 17.52, build time x 2 - that mesh has no long thin triangles
  // for a plane to cut. Meshes that do (architectural models with unsubdivided walls and beams) are what it is kept for.
  if (count == 0 || !vertices || env_int("LUM_BVH_SPATIAL", 0) == 0) return build_bvh4(boxes, count, max_leaf, max_depth);
  std::vector<uint8_t> all;
  if (!splittable) { all.assign(count, 1); splittable = all.data(); }
  SplitBuilder b;
  b.vertices = vertices;
  b.splittable = splittable;
  b.max_leaf = std::min<uint32_t>(std::max<uint32_t>(max_leaf, 1), kBvhLeafMaxTri);
  b.kBins = std::min(std::max(env_int("LUM_BVH_BINS", 16), 4), kMaxBins);
  b.kSpatialBins = std::min(std::max(env_int("LUM_BVH_SPATIAL_BINS", 32), 4), kMaxBins);
  b.alpha = env_float("LUM_BVH_SPATIAL_ALPHA", 1e-5f);
  b.budget = (size_t) (env_float("LUM_BVH_SPATIAL_BUDGET", 0.5f) * count);  // at most this many extra references
  std::vector<Ref> refs(count);
  Aabb root = empty_box();
  for (uint32_t i = 0; i < count; i++) { refs[i] = Ref{i, boxes[i]}; grow(root, boxes[i]); }
  b.root_area = std::max(half_area(root), 1e-30f);
  b.order.reserve(count + b.budget);
  b.nodes.reserve(2 * (size_t) count);
  b.build(refs, 0);
  Bvh4 out = collapse(b);
  if (out.max_depth <= max_depth && out.prims.size() < (1u << 28)) return out;
  return build_bvh4(boxes, count, max_leaf, max_depth);  // too deep with splits: the plain tree (which has its own median-split fallback)
}

Bvh4 build_bvh4(const Aabb* boxes, uint32_t count, uint32_t max_leaf, uint32_t max_depth) {
  if (count == 0) {
    Bvh4 out;
    Bvh4Node n;
    std::memset(&n, 0, sizeof(n));
    for (int k = 0; k < 4; k++) {
      n.child[k] = kBvhEmpty;
      n.lo_x[k] = n.lo_y[k] = n.lo_z[k] = FLT_MAX;
      n.hi_x[k] = n.hi_y[k] = n.hi_z[k] = -FLT_MAX;
    }
    out.nodes.push_back(n);
    out.max_depth = 1;
    return out;
  }
  for (int attempt = 0; attempt < 2; attempt++) {
    Builder b;
    b.balanced = attempt == 1;
    b.max_leaf = max_leaf < 1 ? 1 : (max_leaf > kBvhLeafMaxTri ? kBvhLeafMaxTri : max_leaf);
    if (max_leaf > 1) b.max_leaf = (uint32_t) std::min<int>(std::max(env_int("LUM_BVH_MAX_LEAF", (int) b.max_leaf), 1), (int) kBvhLeafMaxTri);  // not the top level (one instance per leaf)
    b.kBins = std::min(std::max(env_int("LUM_BVH_BINS", 16), 4), kMaxBins);
    if (max_leaf > 1) b.sah_leaf_traversal_cost = env_float("LUM_BVH_SAH_LEAF", -1.0f);
    b.items.resize(count);
    for (uint32_t i = 0; i < count; i++) { std::memcpy(b.items[i].lo, boxes[i].lo, 12); std::memcpy(b.items[i].hi, boxes[i].hi, 12); b.items[i].prim = i; }
    const auto t0 = std::chrono::steady_clock::now();
    b.build(count);
    if (max_leaf > 1 && attempt == 0) {
      const int passes = env_int("LUM_BVH_REINSERT", 0);
      if (passes > 0) { Reinserter r(b.nodes); r.run(passes, env_float("LUM_BVH_REINSERT_FRACTION", 1.0f)); }
    }
    const auto t1 = std::chrono::steady_clock::now();
    Bvh4 out = collapse(b);
    if (env_int("LUM_BVH_TIMING", 0))
      std::fprintf(stderr, "[bvh] %u primitives: binary tree %.3f s (%u threads), collapse %.3f s\n", count, std::chrono::duration<double>(t1 - t0).count(), build_threads(),
                   std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
    if (out.max_depth <= max_depth) return out;
  }
  return Bvh4();
}

}  // namespace lum

// lum_core.h lumc_host_bvh_probe: the host builder's tree in numbers (tests/test_bvh_collapse.py)
extern "C" uint32_t lumc_leaf_max_triangles(void) { return lum::kBvhLeafMaxTri; }
extern "C" int lumc_host_bvh_probe(const float* boxes, uint32_t count, uint32_t max_leaf, uint64_t out[6], double* inner_area) {
  using namespace lum;
  if (!boxes || !out || count == 0) return 1;
  static_assert(sizeof(Aabb) == 24, "six floats per box");
  const Aabb* in = reinterpret_cast<const Aabb*>(boxes);
  const Bvh4 t = build_bvh4(in, count, max_leaf, 64);
  if (t.nodes.empty()) return 1;
  std::vector<uint32_t> seen(count, 0);
  uint64_t leaves = 0, largest = 0, slots = 0;
  bool sound = t.prims.size() == count;
  double area = 0.0;
  struct Job { uint32_t node; Aabb box; bool has_box; };
  std::vector<Job> stack{{0u, empty_box(), false}};
  std::vector<Aabb> below(t.nodes.size(), empty_box());
  // boxes of what lies below every child, bottom-up over a depth-first order
  std::vector<uint32_t> order;
  { std::vector<uint32_t> st{0u}; while (!st.empty()) { const uint32_t n = st.back(); st.pop_back(); order.push_back(n); for (int k = 0; k < 4; k++) { const uint32_t c = t.nodes[n].child[k]; if (c != kBvhEmpty && !(c & kBvhLeafBit)) st.push_back(c); } } }
  for (size_t i = order.size(); i-- > 0;) {
    const Bvh4Node& n = t.nodes[order[i]];
    Aabb all = empty_box();
    for (int k = 0; k < 4; k++) {
      const uint32_t c = n.child[k];
      if (c == kBvhEmpty) continue;
      slots++;
      Aabb content = empty_box();
      if (c & kBvhLeafBit) {
        const uint32_t first = c & 0x0FFFFFFFu, cnt = ((c >> 28) & 7u) + 1u;
        leaves++; largest = std::max<uint64_t>(largest, cnt);
        for (uint32_t j = 0; j < cnt; j++) {
          if (first + j >= t.prims.size() || t.prims[first + j] >= count) { sound = false; continue; }
          seen[t.prims[first + j]]++;
          grow(content, in[t.prims[first + j]]);
        }
      }
      else content = below[c];
      const Aabb cb{{n.lo_x[k], n.lo_y[k], n.lo_z[k]}, {n.hi_x[k], n.hi_y[k], n.hi_z[k]}};
      for (int a = 0; a < 3; a++) if (cb.lo[a] > content.lo[a] || cb.hi[a] < content.hi[a]) sound = false;  // (the stored boxes are padded outwards)
      grow(all, cb);
    }
    below[order[i]] = all;
    area += half_area(all);
  }
  for (uint32_t i = 0; i < count; i++) if (seen[i] != 1u) sound = false;
  out[0] = t.nodes.size(); out[1] = leaves; out[2] = largest; out[3] = t.max_depth; out[4] = slots; out[5] = sound ? 1u : 0u;
  if (inner_area) { inner_area[0] = area; inner_area[1] = t.plan_cost; inner_area[2] = t.brute_cost; }
  return 0;
}
