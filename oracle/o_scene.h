/*
 * ORACLE (test infrastructure, not product): an INDEPENDENT restatement of the reference's host -> device scene encoders and of its
 * light-tree build, written from the reference's text (device/device_structs.c, device/device_packing.c, device/device_light.c,
 * host_math.c, host_intrinsics.h, device/cuda/light.cuh) and NOT from the product's luminary_amd/csrc/host/scene.cpp.
 *
 * Why it exists: the rendering oracle (o_render.c) and the HIP kernels both consume the device-format scene the
 * product's scene.cpp builds, so a wrong rounding in an encoder or a light tree that differs from device_light.c's is decoded identically by
 * both and no render-parity test can see it. This file takes the HOST-level scene (the loaders' meshes, the public API's materials and
 * instances, the raw textures) and produces the same device-format arrays a second time; tests/test_scene_encoders.py compares bytes.
 *
 * Only tests/ may load this (through liboracle.so). Parity status: "unpinned" against a running reference like the rest of the device
 * layer - device_structs.c / device_light.c reach cuda.h through device_utils.h and cannot be built here - but pinned against the
 * product by construction of a second, separately written implementation.
 */
#ifndef ORACLE_O_SCENE_H
#define ORACLE_O_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* host-level material: the fields of the reference's Material (include/luminary/structs.h:314-336) */
typedef struct OSceneMaterial {
  int32_t base_substrate; /* 0 opaque, 1 translucent */
  float albedo[4];
  float emission[3];
  float emission_scale, roughness, roughness_clamp, refraction_index;
  uint8_t emission_active, thin_walled, metallic, colored_transparency, roughness_as_smoothness, normal_map_is_compressed, bidirectional_emission, pad;
  uint16_t albedo_tex, luminance_tex, roughness_tex, metallic_tex, normal_tex, pad2;
} OSceneMaterial;

/* host-level mesh: flat per-triangle arrays (mesh.h:8-14 TriangleGeomData) */
typedef struct OSceneMesh {
  const float* positions;       /* 9 per triangle */
  const float* normals;         /* 9 per triangle */
  const float* uvs;             /* 6 per triangle */
  const uint16_t* material_ids; /* 1 per triangle */
  uint32_t triangle_count;
  uint32_t pad;
} OSceneMesh;

/* host-level instance (mesh.h:23-30 MeshInstance) */
typedef struct OSceneInstance {
  uint32_t mesh_id;
  uint32_t active;
  float translation[3], rotation[3] /* euler angles */, scale[3];
} OSceneInstance;

typedef struct OSceneInput {
  uint32_t num_meshes, num_instances, num_materials, num_textures;
  const OSceneMesh* meshes;
  const OSceneInstance* instances;
  const OSceneMaterial* materials;
  const uint32_t* texture_table; /* 4 words per texture: first texel, width, height, gamma bits (raw host textures, no encoding) */
  const uint32_t* texels;        /* RGBA8 */
} OSceneInput;

typedef struct OSceneOutput {
  uint32_t total_triangles;
  uint32_t num_lights;
  uint32_t light_tree_root_bytes;
  uint32_t num_light_tree_nodes;
  uint32_t* mesh_tri_offset;    /* num_meshes + 1 */
  float* vertices;              /* 12 floats per triangle (3 x DeviceTriangleVertex) */
  uint32_t* tri_tex;            /* 4 words per triangle (DeviceTriangleTexture) */
  uint32_t* instance_mesh_ids;  /* the product's container convention: 0xFFFFFFFF for an inactive instance or an invalid mesh id */
  float* instance_transforms;   /* 8 words per instance (DeviceTransform) */
  uint16_t* materials;          /* 16 halfwords per material (DeviceMaterialCompressed) */
  uint8_t* light_tree_root;     /* DeviceLightTreeRootHeader + sections */
  uint8_t* light_tree_nodes;    /* DeviceLightTreeNode x num_light_tree_nodes */
  uint32_t* light_tri_handles;  /* TriangleHandle {instance_id, tri_id} x num_lights */
  float* light_bvh_tris;        /* LightTreeBVHTriangle: 3 x Vec128 per light; the w lanes hold what the reference's arithmetic leaves there */
  float* light_intensities;     /* per light: the average_intensity the fragment was weighted with (diagnosis) */
} OSceneOutput;

/* ---- the scalar half of the conversion (device_structs.c:11-250) and the parameters the product derives on the host where the reference derives them per ray ---- */
typedef struct OSceneEntities {
  /* settings.c / camera.c / sky.c / fog.c / particles.c / ocean.c / cloud.c values as the public API returns them */
  uint32_t width, height, supersampling;
  float cam_rotation[3];                        /* euler angles */
  float sky_azimuth, sky_altitude, sky_moon_azimuth, sky_moon_altitude, sky_geometry_offset[3], sky_mie_diameter;
  float fog_droplet_diameter;
  float particles_direction_altitude, particles_direction_azimuth, particles_phase_diameter;
  uint32_t ocean_water_type, ocean_caustics_ris_sample_count;
  float cloud_droplet_diameter, cloud_wind_angle[3]; /* low, mid, top */
} OSceneEntities;

typedef struct OSceneConstants {
  uint32_t width, height;                       /* internal resolution: << supersampling (device_structs.c:21-22) */
  float cam_rotation[4];                        /* quaternion x, y, z, w (device_structs.c:76, host_math.c:6-21) */
  float sky_sun_pos[3], sky_moon_pos[3];        /* device_structs.c:131-171, in double */
  float sky_mie_phase[4], fog_phase[4], particles_phase[4], cloud_phase[4]; /* jendersie_eon_phase_parameters, cuda/math.cuh:1189-1232 */
  float particles_direction[3];                 /* angles_to_direction, cuda/math.cuh:781-788 */
  float ocean_scattering[3], ocean_absorption[3], ocean_molecular_weight; /* cuda/ocean_utils.cuh:300-385 */
  uint32_t ocean_caustics_ris_sample_count;     /* max(n, 1) - 1 (device_structs.c:94) */
  float cloud_wind[3][2];                       /* cos, sin of the layers' wind angles (device_structs.c:184-185) */
} OSceneConstants;

void oracle_scene_constants(const OSceneEntities* in, OSceneConstants* out);

int oracle_scene_encode(const OSceneInput* in, OSceneOutput* out);
void oracle_scene_free(OSceneOutput* out);

#ifdef __cplusplus
}
#endif
#endif
