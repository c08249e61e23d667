/*
 * ORACLE (test infrastructure, not product): public interface of the CPU restatement.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. The product
 * (libluminary_amd.so) never links, loads or calls it.
 *
 * The scene is handed over in the *device format* (what the kernels see): vertices with packed normals,
 * 32-byte compressed materials, 32-byte instance transforms, the 8-wide quantised light tree, the blue-noise
 * mask and the four BSDF energy LUTs. Those buffers are produced by the product's host layer
 * (reference: device/device_structs.c, device/device_light.c) and are inputs here.
 * Layout must stay identical to `LumDeviceSceneView` in include/lum_core.h (checked by tests/test_layouts.py).
 */
#ifndef ORACLE_ORACLE_H
#define ORACLE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleScene {
  /* geometry: all meshes concatenated; mesh m owns triangles [mesh_tri_offset[m], mesh_tri_offset[m+1]) */
  uint32_t num_meshes;
  uint32_t num_instances;
  uint32_t num_materials;
  uint32_t num_lights;
  const uint32_t* mesh_tri_offset;   /* num_meshes + 1 */
  const float* vertices;             /* 3 per triangle, 16 B each: x,y,z, packed normal (as float bits) */
  const uint32_t* tri_tex;           /* 16 B per triangle: uv0, uv1, uv2, material_id (low 16 bits) */
  const uint32_t* instance_mesh_ids; /* num_instances */
  const float* instance_transforms;  /* 32 B each: translation, scale, 4 x u16 quaternion */
  const uint16_t* materials;         /* 32 B each */
  /* light tree (device_utils.h:283-327) */
  const uint8_t* light_tree_root;    /* 16-byte header + 48 bytes per section; NULL when there are no lights */
  const uint8_t* light_tree_nodes;   /* 64 bytes per node */
  const uint32_t* light_tri_handles; /* 2 per light: instance_id, tri_id */
  const float* light_bvh_tris;       /* 12 floats per light: 3 x (x,y,z,pad) world space */
  uint32_t num_light_tree_nodes;
  uint32_t num_textures;
  /* sampler + LUTs */
  const uint32_t* bluenoise_2d;      /* 65536 */
  const uint16_t* lut_conductor;     /* 1024 */
  const uint16_t* lut_glossy;        /* 1024 */
  const uint16_t* lut_dielectric;    /* 32768 */
  const uint16_t* lut_dielectric_inv;/* 32768 */
  const uint32_t* texture_table;      /* 4 words per texture: first texel, width, height, gamma (float bits) */
  const uint32_t* texels;             /* RGBA8, r in the low byte; all textures back to back */
  /* settings (device_structs.h:8-22), internal resolution */
  uint32_t width, height, max_ray_depth, shading_mode;
  /* camera (device_structs.h:38-82), thin lens only */
  float cam_pos[3];
  float cam_rotation[4]; /* quaternion x,y,z,w */
  float cam_fov, cam_aperture_size, cam_object_distance, cam_scale, cam_rr_threshold;
  uint32_t cam_aperture_shape, cam_aperture_blade_count;
  /* sky (device_structs.h:100-124) */
  uint32_t sky_mode;
  float sky_constant_color[3];
  /* procedural sky (device_structs.h:101-124), read when sky_mode == DEFAULT */
  uint32_t sky_steps, sky_ozone_absorption;
  float sky_geometry_offset[3];
  float sky_sun_strength, sky_base_density, sky_rayleigh_density, sky_mie_density, sky_ozone_density, sky_rayleigh_falloff, sky_mie_falloff,
    sky_ground_visibility, sky_ozone_layer_thickness, sky_multiscattering_factor;
  float sky_sun_pos[3];   /* sky space, kilometres (device_structs.c:135-150) */
  float sky_mie_phase[4]; /* Jendersie-Eon g_hg, g_d, alpha, w_d of sky.mie_diameter (math.cuh:1189-1232) */
  const float* sky_lut_transmittance;   /* 2 x 64 x 256 float4 (low wavelengths plane, high plane); inputs: oracle_sky_generate_luts makes them */
  const float* sky_lut_multiscattering; /* 2 x 32 x 32 float4 */
  /* the other celestial bodies of the procedural sky (sky.cuh:447-505) */
  float sky_moon_pos[3];
  float sky_moon_tex_offset;
  uint32_t sky_moon_albedo_tex, sky_moon_normal_tex; /* texture ids in texture_table (appended by the host layer); 0xFFFFFFFF = no moon */
  float sky_stars_intensity;
  uint32_t sky_stars_count;
  const float* sky_stars;            /* 4 floats per star in grid order: altitude, azimuth, radius, intensity (utils.h:115-121) */
  const uint32_t* sky_stars_offsets; /* 64 x 32 + 1 cell offsets (device_sky.c:469-547) */
  /* sky mode HDRI: a dim x dim equirectangular panorama, 4 floats per texel (device_sky.c:344-366). NULL = baked from the procedural sky
     * at upload, seen from sky_hdri_origin with sky_hdri_samples samples per texel (sky.hdri_dim / hdri_samples, the camera position) */
  const float* sky_hdri;
  uint32_t sky_hdri_dim, sky_hdri_samples;
  float sky_hdri_origin[3];
  uint32_t sky_aerial_perspective; /* sky.aerial_perspective: in-scattering and extinction along rays that hit geometry (not in constant-colour mode) */
  /* fog volume (device_structs.c:219-231, cuda/volume_utils.cuh:8-27): a homogeneous scattering volume bounded by a disk of radius fog_dist around
   * the camera and by y <= fog_height; scattering coefficient 0.001 * fog_density, no absorption */
  uint32_t fog_active;
  float fog_density, fog_dist, fog_height;
  float fog_phase[4];               /* Jendersie-Eon g_hg, g_d, alpha, w_d of fog.droplet_diameter (math.cuh:1189-1232) */
  const float* bridge_lut;          /* 64 x 21 floats (data/bridge): vertex-count importance of the bridge sampler (light_bridges.cuh:67-108) */
  uint32_t bridge_max_num_vertices; /* settings.bridge_max_num_vertices (a 4-bit field on the device, device_structs.h:11) */
  /* particles (device_particle.c, cuda/particle.cuh:165-211): `particles_count` camera-facing-agnostic quads scattered in the unit cube, tiled 25 x 25 x 25
   * times around the ray's start in a space scaled by particles_scale; hit only by delta paths. Generated by the host layer. */
  uint32_t particles_active, particles_count;
  float particles_scale, particles_speed;
  float particles_albedo[3];
  float particles_direction[3];       /* angles_to_direction(direction_altitude, direction_azimuth) (math.cuh:781-788) */
  float particles_phase[4];           /* Jendersie-Eon parameters of phase_diameter */
  const float* particle_vertices;     /* 6 x float4 per particle: two triangles (a00, a01, a10), (a11, a01, a10), w = 1 */
  const float* particle_normals;      /* float4 per particle: the quad's normal */
  /* ocean (device_structs.c ocean convert; cuda/ocean_utils.cuh): a procedural height field at y = ocean_height, water below it (the second volume type).
   * The Jerlov water type's coefficients (ocean_utils.cuh:291-385) are looked up by the host layer. */
  uint32_t ocean_active;
  float ocean_height, ocean_amplitude, ocean_frequency, ocean_refractive_index;
  float ocean_scattering[3], ocean_absorption[3];
  float ocean_molecular_weight;
  uint32_t ocean_caustics_active, ocean_caustics_ris_sample_count;
  float ocean_caustics_domain_scale;
  uint32_t ocean_multiscattering, ocean_triangle_light_contribution;
  /* clouds (device_structs.c:173-217; cuda/cloud.cuh): three layers of ray-marched noise-density clouds, rendered in sky mode DEFAULT only
   * (device_manager.c:474). cloud_layers[l] (l = low, mid, top): active, height_max, height_min, coverage, coverage_min, type, type_min,
   * wind_speed, cos and sin of wind_angle. The noise textures (RGBA8: shape 128^3, detail 32^3, weather 1024^2, device_cloud.c:9-11) are taken
   * from here when all three are given, otherwise generated by the core from cloud_seed. */
  uint32_t cloud_active, cloud_atmosphere_scattering, cloud_steps, cloud_shadow_steps, cloud_octaves, cloud_seed;
  float cloud_offset_x, cloud_offset_z, cloud_density, cloud_noise_shape_scale, cloud_noise_detail_scale, cloud_noise_weather_scale;
  float cloud_phase[4];               /* Jendersie-Eon parameters of droplet_diameter */
  float cloud_layers[3][10];
  const uint8_t* cloud_noise_shape;
  const uint8_t* cloud_noise_detail;
  const uint8_t* cloud_noise_weather;
} OracleScene;

/* counters[0] closest-hit rays, [1] shadow rays executed, [2] light-BVH queries executed, [3] path vertices shaded */
enum { ORACLE_CNT_TRACE = 0, ORACLE_CNT_SHADOW = 1, ORACLE_CNT_LIGHT_BVH = 2, ORACLE_CNT_VERTICES = 3, ORACLE_CNT_COUNT = 4 };

/*
 * Renders samples [first_sample, first_sample + num_samples) of the pixels listed in `pixels` (index = x + y*width;
 * NULL = every pixel) and ADDS them into planar accumulators of `num_pixels` floats each:
 * first_moment = [R | G | B], second_moment = luminance of the squared sample (accumulation.cuh:63-84).
 * `threads` <= 0 uses every core. `use_bvh` = 0 intersects by brute force (small scenes only). Returns 0 on success.
 */
int oracle_render(
  const OracleScene* scene, const uint32_t* pixels, uint32_t num_pixels, uint32_t first_sample, uint32_t num_samples, int use_bvh,
  int threads, float* first_moment, float* second_moment, uint64_t* counters);

/* Closest-hit query used by the traversal parity tests: out = instance_id, tri_id, t bits per ray (HIT_TYPE_SKY on miss). */
int oracle_trace_closest(
  const OracleScene* scene, uint32_t num_rays, const float* origins, const float* dirs, const uint32_t* ignore_handles, int use_bvh,
  uint32_t* out_hits);

/* BSDF energy LUT generation (bsdf_lut.cuh). texel range [first, first+count) of the named table:
 * 0 conductor, 1 glossy (needs conductor), 2 dielectric, 3 dielectric_inv. */
int oracle_generate_lut(const uint32_t* bluenoise_2d, int table, uint32_t first, uint32_t count, const uint16_t* conductor, uint16_t* dst);

/* Unit-level entry points for known-answer tests. */
uint32_t oracle_squares32(uint32_t key, uint32_t counter);
void oracle_sobol(uint32_t offset, uint32_t dimension, uint32_t out[2]);
void oracle_random_2d(const uint32_t* bn, uint32_t target, uint32_t px, uint32_t py, uint32_t sample, uint32_t depth, uint32_t out[2]);
void oracle_record_roundtrip(const float in[3], uint32_t packed[2], float out[3]);
void oracle_ray_roundtrip(const float in[3], uint32_t packed[2], float out[3]);
uint32_t oracle_normal_pack(const float in[3]);
void oracle_normal_unpack(uint32_t packed, float out[3]);
void oracle_sincos(float x, float out[2]);
float oracle_atan2(float y, float x);
void oracle_camera_ray(const OracleScene* scene, uint32_t x, uint32_t y, uint32_t sample_id, float out[6]);
uint32_t oracle_scene_sizeof(void);

/* Output chain (o_output.h): planar first moment [3 * src pixels] -> display-referred planes `frame_output` and
 * ARGB8 words [dst pixels]. Field-for-field the same as LumOutputParams (include/lum_core.h). */
typedef struct {
  uint32_t src_width, src_height, dst_width, dst_height;
  float inv_sample_count, exposure;
  uint32_t tonemap, filter, dithering, purkinje, use_color_correction, passthrough;
  float purkinje_kappa1, purkinje_kappa2;
  float cc_h, cc_s, cc_v;
  float film_grain;
  float agx_slope, agx_power, agx_saturation;
  uint32_t supersampling, undersampling_stage;
} OracleOutputParamsAbi;
/* input: planes of (src >> undersampling_stage) pixels; frame_output: planes of (src >> max(stage, supersampling)) pixels */
void oracle_generate_output(const OracleOutputParamsAbi* params, const float* first_moment, const uint16_t* bluenoise_1d, float* frame_output, uint32_t* argb8);
/* bloom (device/device_post.c:56-139), in place on the planar image [3][(full_width >> stage) * (full_height >> stage)] */
void oracle_post_bloom(float* image, uint32_t full_width, uint32_t full_height, uint32_t stage, float blend);
/* compact preview image of the undersampling iteration (stage, iteration): 3 planes of (width >> stage) x (height >> stage) */
void oracle_result_undersampled(const float* first_moment, uint32_t width, uint32_t height, uint32_t stage, uint32_t iteration, float* result);
/* Adaptive sampling (o_adaptive.h). Blocks are 4x4 pixels, row-major over ceil(width/4) x ceil(height/4); executions[s] = completed
 * executions of stage s; stage_counts[block] holds count-1 of stages 1..4 in its four bytes. */
int oracle_render_counts(
  const OracleScene* scene, const uint32_t* first_sample, const uint32_t* num_samples, int use_bvh, int threads, float* first_moment,
  float* second_moment, uint64_t* counters); /* per-pixel sample ranges over the whole frame, added like oracle_render */
void oracle_adaptive_build_stage(
  uint32_t width, uint32_t height, const uint32_t executions[5], uint32_t current_stage, uint32_t max_rate, uint32_t avg_rate, float exposure,
  const OracleOutputParamsAbi* op, const float* first_moment, const float* second_moment, uint32_t* stage_counts, float* block_variance, float* total);
void oracle_adaptive_block_variance(
  uint32_t width, uint32_t height, const uint32_t executions[5], uint32_t current_stage, float exposure, const OracleOutputParamsAbi* op,
  const float* first_moment, const float* second_moment, const uint32_t* stage_counts, float* block_variance);
float oracle_adaptive_counts_from(
  uint32_t width, uint32_t height, uint32_t current_stage, uint32_t max_rate, uint32_t avg_rate, const float* block_variance, uint32_t* stage_counts);
void oracle_pixel_samples(uint32_t width, uint32_t height, const uint32_t executions[5], const uint32_t* stage_counts, uint32_t* out);
void oracle_generate_result(
  uint32_t width, uint32_t height, uint32_t mode, uint32_t local_error_minimization, uint32_t uniform_samples, float exposure, const uint32_t executions[5],
  uint32_t stage_id, const uint32_t* stage_counts, const OracleOutputParamsAbi* op, const float* first_moment, const float* second_moment, float* frame_result);
/* Procedural sky (o_sky.h): the two look-up tables of the scene's atmosphere (2 x 64 x 256 x 4 and 2 x 32 x 32 x 4 floats), and the colour
 * of the sky seen from a world-space point along `ray` with the given ray-march offset in [0, 1). */
void oracle_sky_generate_luts(const OracleScene* scene, float* transmittance, float* multiscattering);
void oracle_sky_color(const OracleScene* scene, const float origin_world[3], const float ray[3], int include_sun, float random_offset, float out[3]);
/* what a ray that leaves the scene sees in sky mode HDRI: the panorama's texel, plus the sun disk when `state` holds ST_CAMERA_DIRECTION (2) or
 * ST_ALLOW_EMISSION (8) (sky_color_main's HDRI branch, cuda/sky.cuh:579-595) */
void oracle_sky_hdri_color(const OracleScene* scene, const float origin_world[3], const float ray[3], uint32_t state, float out[3]);
void oracle_sky_hdri(const OracleScene* scene, const float origin_world[3], uint32_t dim, uint32_t samples, float* rgba); /* sky_hdri.cuh:58-160 */
float oracle_log2(float x);
float oracle_exp2(float x);
float oracle_pow(float x, float y);

#ifdef __cplusplus
}
#endif

/* ---- probes for the analytic checks (tests/test_oracle_analytic.py); see o_render.c ---- */
typedef struct { float albedo[3]; float opacity, roughness, ior_ratio; uint32_t flags; /* 1 translucent, 2 inside, 4 metallic, 8 coloured transparency */ } OracleProbeMaterial;
void oracle_probe_bsdf_sample(const OracleScene* s, const OracleProbeMaterial* m, const float normal[3], const float V[3], uint32_t px, uint32_t py, uint32_t first,
                              uint32_t count, float* rays, float* weights, uint32_t* flags);
void oracle_probe_bsdf_eval(const OracleScene* s, const OracleProbeMaterial* m, const float normal[3], const float V[3], uint32_t count, const float* L, float inv_pdf,
                            float* values);
void oracle_probe_microfacet_pdf(const float V[3], float roughness, uint32_t count, const float* L, float* pdf);
void oracle_probe_triangle_sample(const float origin[3], const float tri[9], int bidirectional, uint32_t count, const float* rnd, float* rays, float* solid_angles,
                                  uint32_t* ok);
void oracle_probe_light_tree(const OracleScene* s, const OracleProbeMaterial* m, const float position[3], const float normal[3], const float V[3], uint32_t px, uint32_t py,
                             uint32_t first, uint32_t count, uint32_t* light_ids, float* weights, float* root_sums);
void oracle_probe_light_sample(const OracleScene* s, const OracleProbeMaterial* m, const float position[3], const float normal[3], const float V[3], uint32_t px, uint32_t py,
                               uint32_t first, uint32_t count, uint32_t* light_ids, float* rays, float* colors, float* dists);

/* fog (o_volume.h) */
void oracle_probe_volume_path(const float cam_pos[3], float dist, float height, uint32_t count, const float* origins, const float* dirs, const float* limits, float* out);
void oracle_probe_fog_phase(const OracleScene* s, uint32_t count, const float* cos_angle, float* out);
void oracle_probe_fog_phase_sample(const OracleScene* s, uint32_t count, const float* rnd, float* out);
void oracle_probe_particle_trace(const OracleScene* s, uint32_t count, const float* pos, const float* dir, const float* tmax, float* out_t, uint32_t* out_tri);
void oracle_probe_volume_sampling(float scattering, float max_length, uint32_t count, const float* rnd, float* t, float* pdf);

/* clouds (o_cloud.h) */
void oracle_cloud_noise(uint32_t seed, uint32_t* shape, uint32_t* detail, uint32_t* weather);
void oracle_probe_cloud_noise(uint32_t count, const float* p, float scale, int octaves, float seed, float persistence, float* out_perlin, float* out_worley);
void oracle_probe_cloud_density(const OracleScene* s, int layer, uint32_t count, const float* sky_pos, float* out_height, float* out_density);

/* ocean (o_ocean.h): heights at (x, z) pairs; per ray the intersection distance, the height of the end point above the surface and the normal there;
 * the Fresnel reflection coefficient of the flat surface per incident direction (from above: index_in_over_out = 1 / ior) */
void oracle_probe_ocean_height(const OracleScene* s, uint32_t count, const float* xz, float* out);
void oracle_probe_ocean_trace(const OracleScene* s, uint32_t count, const float* origins, const float* dirs, const float* limits, float* out_t, float* out_residual,
                              float* out_normal);
void oracle_probe_ocean_fresnel(float ior, uint32_t count, const float* dirs, float* out_reflection, float* out_refracted);

#endif
