// Host-side BVH4 builder for the software traversal that replaces the reference's OptiX acceleration structures
// (reference: src/luminary/device/optix_bvh.c:150-684 builds GAS/IAS through optixAccelBuild).
// Round-1 builder: binned-SAH binary tree on the CPU, collapsed into 128-byte 4-wide nodes. A GPU LBVH builder is the
// planned replacement; the node/triangle layout consumed by the kernels does not depend on the builder.
#pragma once

#include <cstdint>
#include <vector>

#include "../device/dev_scene.h"

namespace lum {

struct Aabb { float lo[3], hi[3]; };

struct Bvh4 {
  std::vector<Bvh4Node> nodes;   // nodes[0] is the root and always an inner node
  std::vector<uint32_t> prims;   // primitive ids in leaf order; leaves reference ranges of this array
  uint32_t max_depth = 0;
  // test support (lumc_host_bvh_probe): the cost the collapse plan reports for the root (sum of the surviving binary nodes' half areas), and - for trees of at
  // most 20 binary inner nodes, LUM_BVH_COLLAPSE_BRUTE=1 - the cheapest of ALL valid choices of surviving nodes, found by trying every subset
  double plan_cost = 0.0, brute_cost = -1.0;
};

// Builds a BVH4 over `count` boxes. Leaves hold at most `max_leaf` (<= kBvhLeafMaxTri) primitives. The tree has at most
// `max_depth` BVH4 levels (the kernels' traversal stack is sized for that): a SAH tree that is deeper is rebuilt with median
// splits; if that is still too deep the result is empty (nodes.empty()).
Bvh4 build_bvh4(const Aabb* boxes, uint32_t count, uint32_t max_leaf = kBvhLeafMaxTri, uint32_t max_depth = 20);

// The same tree with spatial splits (Stich, Friedrich, Dietrich: "Spatial Splits in Bounding Volume Hierarchies", HPG 2009): where the best object
// split leaves the children's boxes overlapping, the set is also tried cut by a plane, triangles that straddle it referenced on both sides with their
// boxes clipped to each side. `vertices` = 3 x float4 per triangle (the device scene's layout); `splittable[i] == 0` keeps triangle i in one leaf (a
// visibility ray multiplies the transparency of every triangle reference it crosses, so only opaque triangles may be referenced twice - for those,
// and for closest hits, a second reference changes nothing: same distance, same ids). prims then holds one entry per REFERENCE (prims.size() >= count).
// Opt-in (LUM_BVH_SPATIAL=1): the benchmark meshes are evenly tessellated and gain nothing (tools/bvh_quality.cpp), long thin triangles do.
Bvh4 build_bvh4_triangles(const float* vertices, const Aabb* boxes, const uint8_t* splittable, uint32_t count, uint32_t max_leaf = kBvhLeafMaxTri, uint32_t max_depth = 20);

// Same contract, built on the current HIP device (lbvh.hip): Morton-ordered binary radix tree collapsed to 4-wide nodes. Much faster to
// build, somewhat slower to trace. Returns an empty result when the tree is deeper than `max_depth` or a HIP call fails; the caller
// then falls back to build_bvh4.
Bvh4 build_bvh4_lbvh(const Aabb* boxes, uint32_t count, uint32_t max_leaf = kBvhLeafMaxTri, uint32_t max_depth = 20);
// The same on the device with parallel locally-ordered clustering (bottom-up merges of nearest neighbours in Morton order) instead of the radix
// tree: two to three times the LBVH's build time, trees between its quality and the SAH builder's.
Bvh4 build_bvh4_ploc(const Aabb* boxes, uint32_t count, uint32_t max_leaf = kBvhLeafMaxTri, uint32_t max_depth = 20);

// The host builder's binned SAH, level by level on the device (lbvh.hip): for meshes without degenerate sets the same binary tree and leaf order as
// build_bvh4, in tens of milliseconds. Empty result when the tree is deeper than `max_depth` 4-wide levels or a HIP call fails (the caller falls back).
Bvh4 build_bvh4_sah_gpu(const Aabb* boxes, uint32_t count, uint32_t max_leaf = kBvhLeafMaxTri, uint32_t max_depth = 20);

}  // namespace lum
