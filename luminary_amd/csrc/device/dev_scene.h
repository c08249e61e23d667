// Layout of the scene as the kernels see it (all pointers are HBM addresses).
// Encodings follow the reference's device formats so that host conversion code and kernels agree with it:
//   vertices       device_structs.h:270-273 (16 B: position + oct-packed normal)
//   tri_tex        device_structs.h:275-281 (16 B: 3 packed UVs + material id)
//   materials      device_structs.h:202-223 (32 B)
//   transforms     device_structs.h:295-300 (32 B)
//   light tree     device_utils.h:283-327
// Added by this implementation: the BVH4 arrays that replace the OptiX acceleration structures (optix_bvh.c).
#pragma once

#include <hip/hip_vector_types.h>
#include <stdint.h>

namespace lum {

// One 128-byte BVH4 node = one L2 cache line. Child boxes are stored SoA so that a lane reads them with 8 x 16-byte loads.
struct alignas(128) Bvh4Node {
  float lo_x[4], lo_y[4], lo_z[4];
  float hi_x[4], hi_y[4], hi_z[4];
  uint32_t child[4];  // kBvhEmpty | leaf: kBvhLeafBit | (count-1) << 28 | first primitive | inner: node index
  uint32_t pad[4];
};
static_assert(sizeof(Bvh4Node) == 128, "BVH4 node must be one cache line");

constexpr uint32_t kBvhEmpty      = 0xFFFFFFFFu;
constexpr uint32_t kBvhLeafBit    = 0x80000000u;
constexpr uint32_t kBvhLeafMaxTri = 4;

// Triangle in traversal order, 48 bytes: v0.xyz + id | e1.xyz | e2.xyz (edges precomputed with the same float
// subtraction the reference's intersection code performs, so hit distances are identical).
struct BvhTri { float p0[3]; uint32_t id; float e1[3]; uint32_t pad0; float e2[3]; uint32_t pad1; };
static_assert(sizeof(BvhTri) == 48, "48 bytes per triangle");

struct DeviceScene {
  // geometry
  const uint32_t* mesh_tri_offset;
  const float4* vertices;
  const uint4* tri_tex;
  const uint32_t* instance_mesh_ids;
  const float4* instance_transforms;  // 2 x float4 per instance
  const uint4* materials;             // 2 x uint4 per material
  // light tree
  const uint4* light_tree_root;   // header, then 3 x 16 B per section; nullptr without lights
  const uint4* light_tree_nodes;  // 4 x 16 B per node
  const uint2* light_tri_handles;
  // sampler and LUTs
  const uint32_t* bluenoise_2d;
  const uint16_t* lut_conductor;
  const uint16_t* lut_glossy;
  const uint16_t* lut_dielectric;
  const uint16_t* lut_dielectric_inv;
  // acceleration structures
  const Bvh4Node* blas_nodes;      // all meshes, concatenated
  const BvhTri* blas_tris;         // all meshes, traversal order, id = triangle id inside its mesh
  const uint32_t* mesh_node_offset;  // root node of mesh m = blas_nodes[mesh_node_offset[m]]
  const uint32_t* mesh_bvhtri_offset;
  const Bvh4Node* tlas_nodes;      // leaves index tlas_prims
  const uint32_t* tlas_prims;      // instance ids in traversal order
  const float4* instance_inv;      // 3 x float4 per instance: rows of the world->object matrix, .w = translation component
  const Bvh4Node* light_nodes;     // leaves index light_tris
  const BvhTri* light_tris;        // world space, id = light id
  uint32_t num_meshes, num_instances, num_materials, num_lights;
  uint32_t tlas_num_nodes, light_num_nodes;
  // settings / camera / sky (device_structs.h:8-124)
  uint32_t width, height, max_ray_depth, shading_mode;
  float cam_pos[3];
  float cam_rotation[4];
  float cam_fov, cam_aperture_size, cam_object_distance, cam_scale, cam_rr_threshold;
  uint32_t cam_aperture_shape, cam_aperture_blade_count;
  uint32_t sky_mode;
  float sky_constant_color[3];
};

// Path state, one entry per live path, structure-of-arrays of 16-byte words (coalesced 16 B/lane accesses).
struct PathQueue {
  float4* origin_t;   // origin.xyz, hit distance (written by the trace kernel)
  float4* dir_slot;   // direction.xyz, result slot (uint bits)
  uint4* aux;         // throughput record x,y | medium IOR stack | state flags
  uint4* hit_id;      // hit (or ignore) instance, triangle | pixel x | y << 16 | sample id
};

// Next-event-estimation work produced by the shade kernel and consumed by the shadow kernel (80 B per vertex).
struct NeeQueue {
  float4* geo_ray_dist;     // direction to the sampled light, distance
  float4* geo_color_light;  // weighted radiance, light id (uint bits)
  float4* bsdf_ray_prob;    // BSDF-sampled direction, its sampling probability
  float4* bsdf_weight_sum;  // BSDF weight, light tree root sum
  uint4* ambient;           // packed colour x,y | packed direction x,y
};

enum Counter : uint32_t { kCntTrace = 0, kCntShadow = 1, kCntLightBvh = 2, kCntVertices = 3, kCntNodes = 4, kCntTris = 5, kCntNodesShadow = 6, kCntTrisShadow = 7, kCntCount = 8 };

}  // namespace lum
