#include "bvh_build.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

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

struct Builder {
  const Aabb* boxes;
  std::vector<float> centroid;  // 3 per prim
  std::vector<uint32_t> order;
  std::vector<BinNode> nodes;
  bool balanced;
  uint32_t max_leaf;
  int kBins = 16;
  float sah_leaf_traversal_cost = -1.0f;  // < 0: every set that fits a leaf becomes one

  uint32_t build(uint32_t first, uint32_t count, int depth) {
    const uint32_t idx = (uint32_t) nodes.size();
    nodes.emplace_back();
    Aabb box = empty_box();
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = first; i < first + count; i++) {
      grow(box, boxes[order[i]]);
      for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], centroid[3 * order[i] + k]); chi[k] = std::max(chi[k], centroid[3 * order[i] + k]); }
    }
    nodes[idx].box = box;
    nodes[idx].first = first;
    const bool fits_leaf = count <= max_leaf;
    if (fits_leaf && (count <= 1 || sah_leaf_traversal_cost < 0.0f || balanced)) { nodes[idx].count = count; return idx; }

    int axis = 0;
    if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
    if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
    uint32_t mid = first + count / 2;
    bool use_median = balanced || depth > 40 || !(chi[axis] > clo[axis]);

    if (!use_median) {
      float best_cost = FLT_MAX;
      int best_axis = -1, best_bin = -1;
      for (int a = 0; a < 3; a++) {
        const float ext = chi[a] - clo[a];
        if (!(ext > 0.0f)) continue;
        Aabb bin_box[kMaxBins];
        uint32_t bin_cnt[kMaxBins];
        for (int b = 0; b < kBins; b++) { bin_box[b] = empty_box(); bin_cnt[b] = 0; }
        const float scale = kBins / ext;
        for (uint32_t i = first; i < first + count; i++) {
          int b = (int) ((centroid[3 * order[i] + a] - clo[a]) * scale);
          b = std::min(std::max(b, 0), kBins - 1);
          grow(bin_box[b], boxes[order[i]]);
          bin_cnt[b]++;
        }
        float right_area[kMaxBins];
        uint32_t right_cnt[kMaxBins];
        Aabb acc = empty_box();
        uint32_t c = 0;
        for (int b = kBins - 1; b > 0; b--) { grow(acc, bin_box[b]); c += bin_cnt[b]; right_area[b] = half_area(acc); right_cnt[b] = c; }
        acc = empty_box(); c = 0;
        for (int b = 0; b < kBins - 1; b++) {
          grow(acc, bin_box[b]); c += bin_cnt[b];
          if (c == 0 || right_cnt[b + 1] == 0) continue;
          const float cost = half_area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = b; }
        }
      }
      if (fits_leaf) {  // SAH leaf termination: split only if cheaper than testing the whole set
        const float area = half_area(box);
        if (best_axis < 0 || !(area > 0.0f) || sah_leaf_traversal_cost + best_cost / area >= (float) count) { nodes[idx].count = count; return idx; }
      }
      if (best_axis < 0) use_median = true;
      else {
        const float ext = chi[best_axis] - clo[best_axis], scale = kBins / ext, lo = clo[best_axis];
        auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t p) {
          int b = (int) ((centroid[3 * p + best_axis] - lo) * scale);
          b = std::min(std::max(b, 0), kBins - 1);
          return b <= best_bin;
        });
        mid = (uint32_t) (it - order.begin());
        if (mid == first || mid == first + count) use_median = true;
      }
    }
    if (use_median) {
      mid = first + count / 2;
      std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count, [&](uint32_t a, uint32_t b) {
        const float ca = centroid[3 * a + axis], cb = centroid[3 * b + axis];
        return ca < cb || (ca == cb && a < b);
      });
    }
    const uint32_t l = build(first, mid - first, depth + 1);
    const uint32_t r = build(mid, first + count - mid, depth + 1);
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

Bvh4 collapse(const Builder& b) {
  Bvh4 out;
  out.prims = b.order;
  if (b.nodes.empty()) return out;
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
    else { kids[nk++] = root.left; kids[nk++] = root.right; }
    while (nk < 4) {
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
    b.boxes = boxes;
    b.balanced = attempt == 1;
    b.max_leaf = max_leaf < 1 ? 1 : (max_leaf > kBvhLeafMaxTri ? kBvhLeafMaxTri : max_leaf);
    if (max_leaf > 1) b.max_leaf = (uint32_t) std::min<int>(std::max(env_int("LUM_BVH_MAX_LEAF", (int) b.max_leaf), 1), (int) kBvhLeafMaxTri);  // not the top level (one instance per leaf)
    b.kBins = std::min(std::max(env_int("LUM_BVH_BINS", 16), 4), kMaxBins);
    if (max_leaf > 1) b.sah_leaf_traversal_cost = env_float("LUM_BVH_SAH_LEAF", -1.0f);
    b.centroid.resize(3 * (size_t) count);
    for (uint32_t i = 0; i < count; i++)
      for (int k = 0; k < 3; k++) b.centroid[3 * (size_t) i + k] = 0.5f * (boxes[i].lo[k] + boxes[i].hi[k]);
    b.order.resize(count);
    std::iota(b.order.begin(), b.order.end(), 0u);
    b.nodes.reserve(2 * (size_t) count);
    b.build(0, count, 0);
    Bvh4 out = collapse(b);
    if (out.max_depth <= max_depth) return out;
  }
  return Bvh4();
}

}  // namespace lum
