// Host scene store and its conversion to the device format.
// Reference: src/luminary/scene.c (entity store), src/luminary/mesh.h:8-36, src/luminary/device/device_structs.c:11-412
// (entity/material/vertex/transform encoders), src/luminary/device/device_packing.c:6-82, src/luminary/device/device_light.c
// (light tree). Needs no GPU.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/lum_core.h"
#include "../../../include/luminary_amd.h"

namespace lum {

struct HostMesh {
  std::string name;
  std::vector<float> positions;  // 9 per triangle
  std::vector<float> normals;    // 9 per triangle
  std::vector<float> uvs;        // 6 per triangle
  std::vector<uint16_t> material_ids;
  uint32_t triangle_count() const { return (uint32_t) material_ids.size(); }
};

struct HostInstance {  // mesh.h:23-30
  uint32_t mesh_id = 0xFFFFFFFFu;
  LuminaryVec3 translation{0, 0, 0}, scale{1, 1, 1}, rotation{0, 0, 0};
  bool active = true;
};

// RGBA8 texture (r in the low byte of a texel word) sampled normalised, wrapped and linearly filtered; `gamma` is applied to r, g, b
// (texture.h:20-40: PNG files set it to 100000 / gAMA, everything else keeps 1).
struct HostTexture { uint32_t width = 0, height = 0; float gamma = 1.0f; std::vector<uint32_t> texels; };

struct HostScene {
  LuminaryRendererSettings settings;
  LuminaryCamera camera;
  LuminaryOcean ocean;
  LuminarySky sky;
  LuminaryCloud cloud;
  LuminaryFog fog;
  LuminaryParticles particles;
  std::vector<LuminaryMaterial> materials;
  std::vector<HostMesh> meshes;
  std::vector<HostInstance> instances;
  std::vector<HostTexture> textures;
  float hdri_origin[3] = {0.0f, 0.0f, 0.0f};  // where the sky panorama of HDRI mode is baked from: the camera position when the sky last changed
  HostScene();
};

// Defaults (settings.c:6-28, camera.c:7-66, sky.c:6-41, material.c:5-29 and the out-of-scope entities' own files).
void default_settings(LuminaryRendererSettings* s);
void default_camera(LuminaryCamera* c);
void default_sky(LuminarySky* s);
void default_material(LuminaryMaterial* m);
void default_ocean(LuminaryOcean* o);
void default_cloud(LuminaryCloud* c);
void default_fog(LuminaryFog* f);
void default_particles(LuminaryParticles* p);

// Encoders (device_packing.c:6-44, device_structs.c:243-311, :388-412, host_math.c:6-21).
uint32_t pack_normal(const float n[3]);
uint32_t pack_uv(float u, float v);
void encode_material(const LuminaryMaterial& m, uint16_t out[16]);
void euler_to_quaternion(const LuminaryVec3& r, float q[4]);
void encode_transform(const HostInstance& inst, float out[8]);

// Device-format scene plus the storage behind its pointers.
struct DeviceSceneBuffers {
  LumDeviceSceneView view;
  std::vector<uint32_t> mesh_tri_offset;
  std::vector<float> vertices;
  std::vector<uint32_t> tri_tex;
  std::vector<uint32_t> instance_mesh_ids;
  std::vector<float> instance_transforms;
  std::vector<uint16_t> materials;
  std::vector<uint8_t> light_tree_root, light_tree_nodes;
  std::vector<uint32_t> light_tri_handles;
  std::vector<float> light_bvh_tris;
  std::vector<uint32_t> bluenoise;
  std::vector<uint32_t> texture_table, texels;
  std::vector<float> sky_stars;            // 4 floats per star, grid order
  std::vector<uint32_t> sky_stars_offsets; // 64 x 32 + 1
  std::vector<float> particle_vertices;    // 24 floats per particle
  std::vector<float> particle_normals;     // 4 floats per particle
  // texture pool bookkeeping that outlives a partial update
  uint32_t num_textures = 0, moon_albedo_tex = 0xFFFFFFFFu, moon_normal_tex = 0xFFFFFFFFu;
  bool moon_in_pool = false;
  uint32_t rebuilt = 0;  // LUMC_DIRTY_* parts the last update_device_scene rebuilt (what the core has to take over)
};

// Fills `out` from the scene. `bluenoise` must hold 65536 texels. Returns an empty string or an error message.
// update_device_scene re-encodes only the parts named by `dirty` (LUMC_DIRTY_*, include/lum_core.h) and refreshes the scalar fields; everything
// else in `out` must be what an earlier call built from the same scene.
std::string update_device_scene(const HostScene& scene, const std::vector<uint32_t>& bluenoise, uint32_t dirty, DeviceSceneBuffers* out);
std::string build_device_scene(const HostScene& scene, const std::vector<uint32_t>& bluenoise, DeviceSceneBuffers* out);

// Light tree build (device_light.c:2236-2265). Exposed for tests.
struct LightTreeOutput {
  std::vector<uint8_t> root, nodes;
  std::vector<uint32_t> tri_handles;  // 2 per light
  std::vector<float> bvh_tris;        // 12 per light
};
void build_light_tree(const HostScene& scene, LightTreeOutput* out);

}  // namespace lum
