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
};

// Builds a BVH4 over `count` boxes. Leaves hold at most `max_leaf` (<= kBvhLeafMaxTri) primitives. The tree has at most
// `max_depth` BVH4 levels (the kernels' traversal stack is sized for that): a SAH tree that is deeper is rebuilt with median
// splits; if that is still too deep the result is empty (nodes.empty()).
Bvh4 build_bvh4(const Aabb* boxes, uint32_t count, uint32_t max_leaf = kBvhLeafMaxTri, uint32_t max_depth = 20);

// Same contract, built on the current HIP device (lbvh.hip): Morton-ordered binary radix tree collapsed to 4-wide nodes. Much faster to
// build, somewhat slower to trace. Returns an empty result when the tree is deeper than `max_depth` or a HIP call fails; the caller
// then falls back to build_bvh4.
Bvh4 build_bvh4_lbvh(const Aabb* boxes, uint32_t count, uint32_t max_leaf = kBvhLeafMaxTri, uint32_t max_depth = 20);

}  // namespace lum
