/*
 * luminary_amd.h - the Luminary C host API as exported by libluminary_amd.so.
 *
 * Drop-in boundary: every type and function below has the name, field order, argument meaning and error behaviour of
 * the reference's public headers, so a frontend written against <luminary/luminary.h> (e.g. Mandarin Duck) links against
 * this library unchanged. Source of each declaration (paths under /root/reference/include/luminary/):
 *   vectors / colours          api_utils.h:27-51
 *   result codes               error.h:24-101
 *   settings ... instance PODs structs.h:29-391
 *   path                       path.h:24-31
 *   host functions             host.h:29-129
 *   init / shutdown            luminary.h:44-49
 * A frontend's own includes - <luminary/luminary.h>, <luminary/host.h>, <luminary/structs.h>, ... - are served by the forwarding headers in
 * include/luminary/, which all lead here: compile the frontend with -I<this repo>/include and link it with -lluminary_amd.
 * Scope of this implementation: the triangle / BSDF / NEE path with the thin-lens camera under every sky mode
 * (constant colour, procedural atmosphere with sun, moon and stars, baked panorama), the fog volume (scattering events, light scattered in
 * from sun, sky and - over multi-vertex bridges - emissive triangles), particles, the ocean (ray-marched height field, Jerlov water volume,
 * sun and sky light through the surface with caustics), clouds (three ray-marched layers over generated noise textures, procedural sky
 * mode, baked into the panorama in HDRI mode; their shadow in the aerial perspective), textures, adaptive sampling, the undersampling preview, the display chain with bloom,
 * and the debug shading modes. Rendering runs on the library's own "Device" thread once luminary_host_start_new_render was called,
 * like the reference's. The physical camera is stored and returned unchanged but does not influence the image yet.
 *
 * Additive extension (the reference only returns tone-mapped ARGB8): the luminary_ext_* functions at the
 * end give access to float radiance, ray counters and batch rendering. Existing symbols are untouched.
 */
#ifndef LUMINARY_AMD_H
#define LUMINARY_AMD_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LUMINARY_API

/* ---- api_utils.h ---- */
typedef struct LuminaryVec3 { float x, y, z; } LuminaryVec3;
typedef struct LuminaryRGBF { float r, g, b; } LuminaryRGBF;
typedef struct LuminaryRGBAF { float r, g, b, a; } LuminaryRGBAF;
typedef struct LuminaryARGB8 { uint8_t b, g, r, a; } LuminaryARGB8;

/* ---- error.h ---- */
typedef uint64_t LuminaryResult;
#define LUMINARY_SUCCESS (0ull)
#define LUMINARY_ERROR_ARGUMENT_NULL (1ull)
#define LUMINARY_ERROR_NOT_IMPLEMENTED (2ull)
#define LUMINARY_ERROR_INVALID_API_ARGUMENT (3ull)
#define LUMINARY_ERROR_MEMORY_LEAK (4ull)
#define LUMINARY_ERROR_OUT_OF_MEMORY (5ull)
#define LUMINARY_ERROR_C_STD (6ull)
#define LUMINARY_ERROR_API_EXCEPTION (7ull)
#define LUMINARY_ERROR_CUDA (8ull)  /* kept for ABI compatibility: raised for HIP runtime failures */
#define LUMINARY_ERROR_OPTIX (9ull) /* kept for ABI compatibility: raised for acceleration-structure failures */
#define LUMINARY_ERROR_PREVIOUS_ERROR (10ull)
#define LUMINARY_ERROR_DEBUG_ASSERT (11ull)
#define LUMINARY_ERROR_MISSING_DATA (12ull)
#define LUMINARY_ERROR_INVALID_DEVICE (13ull)
#define LUMINARY_ERROR_PROPAGATED (0x8000000000000000ull)
LUMINARY_API const char* luminary_result_to_string(LuminaryResult result);

/* ---- structs.h ---- */
#define LUMINARY_HOST_CREATE_INFO_DEVICE_MASK_ALL_DEVICES (0xFFFFFFFF)
typedef struct LuminaryHostCreateInfo { uint32_t device_mask; } LuminaryHostCreateInfo;

typedef enum LuminaryShadingMode {
  LUMINARY_SHADING_MODE_DEFAULT = 0, LUMINARY_SHADING_MODE_ALBEDO = 1, LUMINARY_SHADING_MODE_DEPTH = 2, LUMINARY_SHADING_MODE_NORMAL = 3,
  LUMINARY_SHADING_MODE_IDENTIFICATION = 4, LUMINARY_SHADING_MODE_LIGHTS = 5, LUMINARY_SHADING_MODE_COUNT
} LuminaryShadingMode;
typedef enum LuminaryAdaptiveSamplingOutputMode {
  LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_BEAUTY = 0, LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_VARIANCE = 1,
  LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_ERROR = 2, LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_SAMPLE_DISTRIBUTION = 3,
  LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_COUNT
} LuminaryAdaptiveSamplingOutputMode;

typedef struct LuminaryRendererSettings {
  uint32_t width, height, max_ray_depth, bridge_max_num_vertices, undersampling, supersampling;
  bool enable_adaptive_sampling;
  uint32_t adaptive_sampling_max_sampling_rate, adaptive_sampling_avg_sampling_rate, adaptive_sampling_update_interval;
  bool adaptive_sampling_exposure_aware;
  LuminaryAdaptiveSamplingOutputMode adaptive_sampling_output_mode;
  LuminaryShadingMode shading_mode;
  float region_x, region_y, region_width, region_height;
} LuminaryRendererSettings;

typedef struct LuminaryDeviceInfo {
  bool is_main_device, is_unavailable, is_enabled;
  char name[256];
  size_t memory_size, allocated_memory_size;
} LuminaryDeviceInfo;

typedef struct LuminaryOutputProperties { bool enabled; uint32_t width, height; } LuminaryOutputProperties;
#define LUMINARY_OUTPUT_HANDLE_INVALID 0xFFFFFFFF
typedef uint32_t LuminaryOutputHandle;
typedef struct LuminaryOutputRequestProperties { uint32_t sample_count, width, height; } LuminaryOutputRequestProperties;
typedef uint32_t LuminaryOutputPromiseHandle;
typedef struct LuminaryPixelQueryResult {
  bool pixel_query_is_valid;
  uint32_t instance_id;
  uint16_t material_id;
  float depth;
  LuminaryVec3 rel_hit_pos;
} LuminaryPixelQueryResult;
typedef struct LuminaryImage {
  uint8_t* buffer;
  uint32_t width, height;
  size_t ld;
  struct { float time; uint32_t sample_count; } meta_data;
} LuminaryImage;

typedef enum LuminaryFilter {
  LUMINARY_FILTER_NONE = 0, LUMINARY_FILTER_GRAY = 1, LUMINARY_FILTER_SEPIA = 2, LUMINARY_FILTER_GAMEBOY = 3, LUMINARY_FILTER_2BITGRAY = 4,
  LUMINARY_FILTER_CRT = 5, LUMINARY_FILTER_BLACKWHITE = 6, LUMINARY_FILTER_COUNT
} LuminaryFilter;
typedef enum LuminaryToneMap {
  LUMINARY_TONEMAP_NONE = 0, LUMINARY_TONEMAP_ACES = 1, LUMINARY_TONEMAP_REINHARD = 2, LUMINARY_TONEMAP_UNCHARTED2 = 3, LUMINARY_TONEMAP_AGX = 4,
  LUMINARY_TONEMAP_AGX_PUNCHY = 5, LUMINARY_TONEMAP_AGX_CUSTOM = 6, LUMINARY_TONEMAP_COUNT
} LuminaryToneMap;
typedef enum LuminaryApertureShape { LUMINARY_APERTURE_ROUND = 0, LUMINARY_APERTURE_BLADED = 1, LUMINARY_APERTURE_COUNT } LuminaryApertureShape;

typedef struct LuminaryCamera {
  LuminaryVec3 pos, rotation;
  LuminaryApertureShape aperture_shape;
  uint32_t aperture_blade_count;
  float exposure;
  LuminaryToneMap tonemap;
  float agx_custom_slope, agx_custom_power, agx_custom_saturation;
  LuminaryFilter filter;
  bool use_local_error_minimization;
  float bloom_blend;
  bool dithering, purkinje;
  float purkinje_kappa1, purkinje_kappa2, wasd_speed, mouse_speed;
  bool smooth_movement;
  float smoothing_factor, russian_roulette_threshold;
  bool use_color_correction;
  LuminaryRGBF color_correction;
  float film_grain, camera_scale, object_distance;
  bool use_physical_camera;
  struct { float fov, aperture_size; } thin_lens;
  struct {
    bool allow_reflections, use_spectral_rendering;
    float focal_length, front_focal_point, back_focal_point, front_principal_point, back_principal_point, aperture_point, aperture_diameter,
      exit_pupil_point, exit_pupil_diameter, image_plane_distance, sensor_width;
  } physical;
} LuminaryCamera;

typedef enum LuminaryJerlovWaterType {
  LUMINARY_JERLOV_WATER_TYPE_I = 0, LUMINARY_JERLOV_WATER_TYPE_IA = 1, LUMINARY_JERLOV_WATER_TYPE_IB = 2, LUMINARY_JERLOV_WATER_TYPE_II = 3,
  LUMINARY_JERLOV_WATER_TYPE_III = 4, LUMINARY_JERLOV_WATER_TYPE_1C = 5, LUMINARY_JERLOV_WATER_TYPE_3C = 6, LUMINARY_JERLOV_WATER_TYPE_5C = 7,
  LUMINARY_JERLOV_WATER_TYPE_7C = 8, LUMINARY_JERLOV_WATER_TYPE_9C = 9, LUMINARY_JERLOV_WATER_TYPE_COUNT
} LuminaryJerlovWaterType;
typedef struct LuminaryOcean {
  bool active;
  float height, amplitude, frequency, refractive_index;
  LuminaryJerlovWaterType water_type;
  bool caustics_active;
  uint32_t caustics_ris_sample_count;
  float caustics_domain_scale;
  bool multiscattering, triangle_light_contribution;
} LuminaryOcean;

typedef enum LuminarySkyMode { LUMINARY_SKY_MODE_DEFAULT = 0, LUMINARY_SKY_MODE_HDRI = 1, LUMINARY_SKY_MODE_CONSTANT_COLOR = 2, LUMINARY_SKY_MODE_COUNT } LuminarySkyMode;
typedef struct LuminarySky {
  LuminaryVec3 geometry_offset;
  float azimuth, altitude, moon_azimuth, moon_altitude, moon_tex_offset, sun_strength, base_density;
  bool ozone_absorption;
  uint32_t steps, stars_count, stars_seed;
  float stars_intensity, rayleigh_density, mie_density, ozone_density, rayleigh_falloff, mie_falloff, mie_diameter, ground_visibility,
    ozone_layer_thickness, multiscattering_factor;
  uint32_t hdri_dim, hdri_samples;
  bool aerial_perspective;
  LuminaryRGBF constant_color;
  LuminarySkyMode mode;
} LuminarySky;

typedef struct LuminaryCloudLayer {
  bool active;
  float height_max, height_min, coverage, coverage_min, type, type_min, wind_speed, wind_angle;
} LuminaryCloudLayer;
typedef struct LuminaryCloud {
  bool active, initialized, atmosphere_scattering;
  LuminaryCloudLayer low, mid, top;
  float offset_x, offset_z, density;
  uint32_t seed;
  float droplet_diameter;
  uint32_t steps, shadow_steps;
  float noise_shape_scale, noise_detail_scale, noise_weather_scale, mipmap_bias;
  uint32_t octaves;
} LuminaryCloud;
typedef struct LuminaryFog { bool active; float density, droplet_diameter, height, dist; } LuminaryFog;
typedef struct LuminaryParticles {
  bool active;
  uint32_t seed, count;
  LuminaryRGBF albedo;
  float speed, direction_altitude, direction_azimuth, phase_diameter, scale, size, size_variation;
} LuminaryParticles;

typedef enum LuminaryMaterialBaseSubstrate {
  LUMINARY_MATERIAL_BASE_SUBSTRATE_OPAQUE, LUMINARY_MATERIAL_BASE_SUBSTRATE_TRANSLUCENT, LUMINARY_MATERIAL_BASE_SUBSTRATE_COUNT
} LuminaryMaterialBaseSubstrate;
typedef struct LuminaryMaterial {
  uint32_t id;
  LuminaryMaterialBaseSubstrate base_substrate;
  LuminaryRGBAF albedo;
  LuminaryRGBF emission;
  float emission_scale, roughness, roughness_clamp, refraction_index;
  bool emission_active, thin_walled, metallic, colored_transparency, roughness_as_smoothness, normal_map_is_compressed, bidirectional_emission;
  uint16_t albedo_tex, luminance_tex, roughness_tex, metallic_tex, normal_tex;
} LuminaryMaterial;
typedef struct LuminaryInstance { uint32_t id, mesh_id; LuminaryVec3 position, rotation, scale; } LuminaryInstance;

/* ---- path.h ---- */
typedef struct LuminaryPath LuminaryPath;
LUMINARY_API LuminaryResult luminary_path_create(LuminaryPath** path);
LUMINARY_API LuminaryResult luminary_path_set_from_string(LuminaryPath* path, const char* string);
LUMINARY_API LuminaryResult luminary_path_destroy(LuminaryPath** path);

/* ---- luminary.h ---- */
LUMINARY_API void luminary_init(void);
LUMINARY_API void luminary_shutdown(void);

/* ---- host.h ---- */
typedef struct LuminaryHost LuminaryHost;
LUMINARY_API LuminaryResult luminary_host_create(LuminaryHost** host, LuminaryHostCreateInfo info);
LUMINARY_API LuminaryResult luminary_host_destroy(LuminaryHost** host);
LUMINARY_API LuminaryResult luminary_host_start_new_render(LuminaryHost* host);
LUMINARY_API LuminaryResult luminary_host_get_device_count(LuminaryHost* host, uint32_t* device_count);
LUMINARY_API LuminaryResult luminary_host_get_device_info(LuminaryHost* host, uint32_t device_id, LuminaryDeviceInfo* info);
LUMINARY_API LuminaryResult luminary_host_set_device_enable(LuminaryHost* host, uint32_t device_id, bool enable);
LUMINARY_API LuminaryResult luminary_host_start_device(LuminaryHost* host, uint32_t index);
LUMINARY_API LuminaryResult luminary_host_shutdown_device(LuminaryHost* host, uint32_t index);
LUMINARY_API LuminaryResult luminary_host_load_lum_file(LuminaryHost* host, LuminaryPath* path);
LUMINARY_API LuminaryResult luminary_host_load_obj_file(LuminaryHost* host, LuminaryPath* path);
LUMINARY_API LuminaryResult luminary_host_get_current_sample_time(LuminaryHost* host, double* time);
LUMINARY_API LuminaryResult luminary_host_get_num_queue_workers(const LuminaryHost* host, uint32_t* num_queue_workers);
LUMINARY_API LuminaryResult luminary_host_get_queue_worker_name(const LuminaryHost* host, uint32_t queue_worker_id, const char** string);
LUMINARY_API LuminaryResult luminary_host_get_queue_worker_string(const LuminaryHost* host, uint32_t queue_worker_id, const char** string);
LUMINARY_API LuminaryResult luminary_host_get_queue_worker_time(const LuminaryHost* host, uint32_t queue_worker_id, double* time);
LUMINARY_API LuminaryResult luminary_host_set_output_properties(LuminaryHost* host, LuminaryOutputProperties properties);
LUMINARY_API LuminaryResult luminary_host_request_output(LuminaryHost* host, LuminaryOutputRequestProperties properties, LuminaryOutputPromiseHandle* handle);
LUMINARY_API LuminaryResult luminary_host_try_await_output(LuminaryHost* host, LuminaryOutputPromiseHandle handle, LuminaryOutputHandle* output_handle);
LUMINARY_API LuminaryResult luminary_host_acquire_output(LuminaryHost* host, LuminaryOutputHandle* output_handle);
LUMINARY_API LuminaryResult luminary_host_get_image(LuminaryHost* host, LuminaryOutputHandle output_handle, LuminaryImage* image);
LUMINARY_API LuminaryResult luminary_host_release_output(LuminaryHost* host, LuminaryOutputHandle output_handle);
LUMINARY_API LuminaryResult luminary_host_get_pixel_info(LuminaryHost* host, uint16_t x, uint16_t y, LuminaryPixelQueryResult* result);
LUMINARY_API LuminaryResult luminary_host_get_settings(LuminaryHost* host, LuminaryRendererSettings* settings);
LUMINARY_API LuminaryResult luminary_host_set_settings(LuminaryHost* host, const LuminaryRendererSettings* settings);
LUMINARY_API LuminaryResult luminary_host_get_camera(LuminaryHost* host, LuminaryCamera* camera);
LUMINARY_API LuminaryResult luminary_host_set_camera(LuminaryHost* host, const LuminaryCamera* camera);
LUMINARY_API LuminaryResult luminary_host_get_ocean(LuminaryHost* host, LuminaryOcean* ocean);
LUMINARY_API LuminaryResult luminary_host_set_ocean(LuminaryHost* host, const LuminaryOcean* ocean);
LUMINARY_API LuminaryResult luminary_host_get_sky(LuminaryHost* host, LuminarySky* sky);
LUMINARY_API LuminaryResult luminary_host_set_sky(LuminaryHost* host, const LuminarySky* sky);
LUMINARY_API LuminaryResult luminary_host_get_cloud(LuminaryHost* host, LuminaryCloud* cloud);
LUMINARY_API LuminaryResult luminary_host_set_cloud(LuminaryHost* host, const LuminaryCloud* cloud);
LUMINARY_API LuminaryResult luminary_host_get_fog(LuminaryHost* host, LuminaryFog* fog);
LUMINARY_API LuminaryResult luminary_host_set_fog(LuminaryHost* host, const LuminaryFog* fog);
LUMINARY_API LuminaryResult luminary_host_get_particles(LuminaryHost* host, LuminaryParticles* particles);
LUMINARY_API LuminaryResult luminary_host_set_particles(LuminaryHost* host, const LuminaryParticles* particles);
LUMINARY_API LuminaryResult luminary_host_get_material(LuminaryHost* host, uint16_t id, LuminaryMaterial* material);
LUMINARY_API LuminaryResult luminary_host_set_material(LuminaryHost* host, uint16_t id, const LuminaryMaterial* material);
LUMINARY_API LuminaryResult luminary_host_get_instance(LuminaryHost* host, uint32_t id, LuminaryInstance* instance);
LUMINARY_API LuminaryResult luminary_host_set_instance(LuminaryHost* host, const LuminaryInstance* instance);
LUMINARY_API LuminaryResult luminary_host_new_instance(LuminaryHost* host, LuminaryInstance* instance);
LUMINARY_API LuminaryResult luminary_host_get_num_meshes(LuminaryHost* host, uint32_t* num_meshes);
LUMINARY_API LuminaryResult luminary_host_get_num_materials(LuminaryHost* host, uint32_t* num_materials);
LUMINARY_API LuminaryResult luminary_host_get_num_instances(LuminaryHost* host, uint32_t* num_instances);
LUMINARY_API LuminaryResult luminary_host_save_png(LuminaryHost* host, LuminaryOutputHandle handle, LuminaryPath* path);
LUMINARY_API LuminaryResult luminary_host_request_sky_hdri_build(LuminaryHost* host);

/* ---- additive extension (not in the reference) ---- */
struct LumDeviceSceneView;
/* ---- "extra utils" frontends link against: reference include/luminary/{host_memory,array,queue,ringbuffer,thread_status,log,name_strings}.h ---- */
#define host_malloc(ptr, size) _host_malloc((void**) (ptr), (size), (const char*) #ptr, (const char*) __func__, __LINE__)
#define host_realloc(ptr, size) _host_realloc((void**) (ptr), (size), (const char*) #ptr, (const char*) __func__, __LINE__)
#define host_free(ptr) _host_free((void**) (ptr), (const char*) #ptr, (const char*) __func__, __LINE__)
LUMINARY_API LuminaryResult _host_malloc(void** ptr, size_t size, const char* buf_name, const char* func, uint32_t line);   /* host_memory.h:28 */
LUMINARY_API LuminaryResult _host_realloc(void** ptr, size_t size, const char* buf_name, const char* func, uint32_t line);  /* host_memory.h:29 */
LUMINARY_API LuminaryResult _host_free(void** ptr, const char* buf_name, const char* func, uint32_t line);                  /* host_memory.h:30 */

#define array_create(array, size_of_element, num_elements) \
  _array_create((void**) (array), (size_of_element), (num_elements), (const char*) #array, (const char*) __func__, __LINE__)
#define array_resize(array, size) _array_resize((void**) (array), (size), (const char*) #array, (const char*) __func__, __LINE__)
#define array_push(array, object) _array_push((void**) (array), (void*) (object), (const char*) #array, (const char*) __func__, __LINE__)
#define array_copy(dst, src) _array_copy((void**) (dst), (void**) (src), (const char*) #dst, (const char*) __func__, __LINE__)
#define array_append(dst, src) _array_append((void**) (dst), (const void*) (src), (const char*) #dst, (const char*) __func__, __LINE__)
#define array_set_num_elements(array, num_elements) \
  _array_set_num_elements((void**) (array), (num_elements), (const char*) #array, (const char*) __func__, __LINE__)
#define array_destroy(array) _array_destroy((void**) (array), (const char*) #array, (const char*) __func__, __LINE__)
LUMINARY_API LuminaryResult _array_create(void** array, size_t size_of_element, uint32_t num_elements, const char* buf_name, const char* func, uint32_t line); /* array.h:34 */
LUMINARY_API LuminaryResult _array_resize(void** array, size_t size, const char* buf_name, const char* func, uint32_t line);
LUMINARY_API LuminaryResult _array_push(void** array, void* object, const char* buf_name, const char* func, uint32_t line);
LUMINARY_API LuminaryResult _array_copy(void** dst, const void* src, const char* buf_name, const char* func, uint32_t line);
LUMINARY_API LuminaryResult _array_append(void** dst, const void* src, const char* buf_name, const char* func, uint32_t line);
LUMINARY_API LuminaryResult _array_destroy(void** array, const char* buf_name, const char* func, uint32_t line);
LUMINARY_API LuminaryResult array_clear(void* array);
LUMINARY_API LuminaryResult array_get_size(const void* array, size_t* size);
LUMINARY_API LuminaryResult array_get_num_elements(const void* array, uint32_t* num_elements);
LUMINARY_API LuminaryResult _array_set_num_elements(void** array, uint32_t num_elements, const char* buf_name, const char* func, uint32_t line);

typedef struct LuminaryQueue LuminaryQueue;          /* queue.h:25-40 */
typedef bool (*LuminaryEqOp)(void* lhs, void* rhs);
#define queue_create(queue, size_of_element, num_elements) \
  _queue_create((queue), (size_of_element), (num_elements), (const char*) #queue, (const char*) __func__, __LINE__)
#define queue_destroy(queue) _queue_destroy((queue), (const char*) #queue, (const char*) __func__, __LINE__)
LUMINARY_API LuminaryResult _queue_create(LuminaryQueue** queue, size_t size_of_element, size_t num_elements, const char* buf_name, const char* func, uint32_t line);
LUMINARY_API LuminaryResult queue_push(LuminaryQueue* queue, void* object);
LUMINARY_API LuminaryResult queue_push_unique(LuminaryQueue* queue, void* object, LuminaryEqOp equal_operator, bool* already_queued);
LUMINARY_API LuminaryResult queue_pop(LuminaryQueue* queue, void* object, bool* success);
LUMINARY_API LuminaryResult queue_pop_blocking(LuminaryQueue* queue, void* object, bool* success);
LUMINARY_API LuminaryResult queue_set_is_blocking(LuminaryQueue* queue, bool is_blocking);
LUMINARY_API LuminaryResult _queue_destroy(LuminaryQueue** queue, const char* buf_name, const char* func, uint32_t line);

typedef struct LuminaryRingBuffer LuminaryRingBuffer;  /* ringbuffer.h:25-34 */
#define ringbuffer_create(buffer, size) _ringbuffer_create((buffer), (size), (const char*) #buffer, (const char*) __func__, __LINE__)
#define ringbuffer_destroy(buffer) _ringbuffer_destroy((buffer), (const char*) #buffer, (const char*) __func__, __LINE__)
LUMINARY_API LuminaryResult _ringbuffer_create(LuminaryRingBuffer** buffer, size_t size, const char* buf_name, const char* func, uint32_t line);
LUMINARY_API LuminaryResult ringbuffer_allocate_entry(LuminaryRingBuffer* buffer, size_t entry_size, void** entry);
LUMINARY_API LuminaryResult ringbuffer_release_entry(LuminaryRingBuffer* buffer, size_t entry_size);
LUMINARY_API LuminaryResult _ringbuffer_destroy(LuminaryRingBuffer** buffer, const char* buf_name, const char* func, uint32_t line);

typedef struct LuminaryThreadStatus LuminaryThreadStatus;  /* thread_status.h:25-34 */
LUMINARY_API LuminaryResult thread_status_create(LuminaryThreadStatus** thread_status);
LUMINARY_API LuminaryResult thread_status_set_worker_name(LuminaryThreadStatus* thread_status, const char* name);
LUMINARY_API LuminaryResult thread_status_get_worker_name(LuminaryThreadStatus* thread_status, const char** name);
LUMINARY_API LuminaryResult thread_status_start(LuminaryThreadStatus* thread_status, const char* string);
LUMINARY_API LuminaryResult thread_status_get_time(LuminaryThreadStatus* thread_status, double* time);
LUMINARY_API LuminaryResult thread_status_get_string(LuminaryThreadStatus* thread_status, const char** string);
LUMINARY_API LuminaryResult thread_status_stop(LuminaryThreadStatus* thread_status);
LUMINARY_API LuminaryResult thread_status_destroy(LuminaryThreadStatus** thread_status);

#define log_message(fmt, ...) luminary_print_log("[%s:%d] " fmt, __func__, __LINE__, ##__VA_ARGS__)   /* log.h:23-31 */
#define warn_message(fmt, ...) luminary_print_warn("[%s:%d] " fmt, __func__, __LINE__, ##__VA_ARGS__)
#define error_message(fmt, ...) luminary_print_error("[%s:%d] " fmt, __func__, __LINE__, ##__VA_ARGS__)
#define crash_message(fmt, ...) luminary_print_crash("[%s:%d] " fmt, __func__, __LINE__, ##__VA_ARGS__)
LUMINARY_API void luminary_print_log(const char* format, ...);
LUMINARY_API void luminary_print_info(bool log, const char* format, ...);
LUMINARY_API void luminary_print_info_inline(bool log, const char* format, ...);
LUMINARY_API void luminary_print_warn(const char* format, ...);
LUMINARY_API void luminary_print_error(const char* format, ...);
LUMINARY_API void luminary_print_crash(const char* format, ...);
LUMINARY_API void luminary_write_log(void);

extern const char* const luminary_strings_shading_mode[LUMINARY_SHADING_MODE_COUNT];   /* name_strings.h:22-29 */
extern const char* const luminary_strings_adaptive_sampling_output_mode[LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_COUNT];
extern const char* const luminary_strings_filter[LUMINARY_FILTER_COUNT];
extern const char* const luminary_strings_tonemap[LUMINARY_TONEMAP_COUNT];
extern const char* const luminary_strings_aperture[LUMINARY_APERTURE_COUNT];
extern const char* const luminary_strings_jerlov_water_type[LUMINARY_JERLOV_WATER_TYPE_COUNT];
extern const char* const luminary_strings_sky_mode[LUMINARY_SKY_MODE_COUNT];
extern const char* const luminary_strings_material_base_substrate[LUMINARY_MATERIAL_BASE_SUBSTRATE_COUNT];

/* Adds an RGBA8 texture (byte order r, g, b, a; rows top to bottom as in a PNG) and returns the id materials refer to through
 * albedo_tex / roughness_tex / normal_tex (luminance and metallic textures are not evaluated, the latter like in the reference). `gamma` is
 * applied to r, g, b on fetch (1 = none; PNG files carry 100000 / gAMA). */
LUMINARY_API LuminaryResult luminary_ext_add_texture(LuminaryHost* host, const uint8_t* rgba8, uint32_t width, uint32_t height, float gamma, uint16_t* texture_id);

/* Embedded data files by name, as the reference's Ceb-generated accessor (device/device_embedded.c:1075-1093): "bluenoise_1D.bin",
 * "bluenoise_2D.bin"; *info = 0 on success, non-zero for an unknown name (frontend assets such as fonts are not part of this library). */
LUMINARY_API void ceb_access(const char* name, void** ptr, int64_t* lmem, uint64_t* info);

/* additive: RGBA8 PNG of an ARGB8 image (words b | g << 8 | r << 16 | a << 24; `ld` = words per row), as luminary_host_save_png writes */
LUMINARY_API LuminaryResult luminary_ext_write_png(const char* path, const uint32_t* argb8, uint32_t width, uint32_t height, size_t ld);
/* additive: bytes currently held through _host_malloc, and the text luminary_write_log would write */
LUMINARY_API LuminaryResult luminary_ext_host_memory_in_use(uint64_t* bytes);
LUMINARY_API LuminaryResult luminary_ext_get_log(const char** text, size_t* length);

/* Adds a mesh from flat per-triangle arrays (reference Mesh layout, mesh.h:8-14): 9 position floats, 9 normal floats, 6 uv floats and one
 * material id per triangle. Materials are added with luminary_ext_add_material. Returns the new ids. */
LUMINARY_API LuminaryResult luminary_ext_add_mesh(
  LuminaryHost* host, const float* positions, const float* normals, const float* uvs, const uint16_t* material_ids, uint32_t triangle_count, uint32_t* mesh_id);
LUMINARY_API LuminaryResult luminary_ext_add_material(LuminaryHost* host, const LuminaryMaterial* material, uint16_t* material_id);
/* The host-level mesh back (borrowed pointers into the host's store, valid until the mesh list changes): what an independent encoder starts from. */
LUMINARY_API LuminaryResult luminary_ext_get_mesh(LuminaryHost* host, uint32_t mesh_id, const float** positions, const float** normals, const float** uvs,
                                                  const uint16_t** material_ids, uint32_t* triangle_count);
/* Converts the current scene to the device format (device_structs.c conversions + light tree build). The view and everything it points
 * to stay valid until the next call or host destruction. Needs no GPU. */
/* Whether replacing `old` by `input` restarts the integration (camera.c:80-147, settings.c:45-72) or only changes the outputs.
 * entity 0: LuminaryRendererSettings, 1: LuminaryCamera. luminary_host_set_camera / set_settings apply this rule. */
LUMINARY_API LuminaryResult luminary_ext_change_restarts_integration(int entity, const void* input, const void* old, bool* restarts);
/* The path of a file named inside `base_file` (mesh files of a .lum, material libraries of an .obj, maps of an .mtl): path_extend +
 * path_apply of src/luminary/path.c */
LUMINARY_API LuminaryResult luminary_ext_path_extend(const char* base_file, const char* name, char* out, size_t out_size);
/* rotation_euler_angles_to_quaternion (src/luminary/host_math.c:6-21), x y z w */
LUMINARY_API LuminaryResult luminary_ext_euler_to_quaternion(const float rotation[3], float quaternion[4]);
LUMINARY_API LuminaryResult luminary_ext_build_device_scene(LuminaryHost* host, const struct LumDeviceSceneView** view);
/* Synchronous batch rendering of sample ids [first_sample, first_sample + num_samples) of `pixels` (NULL = all) on this process' GPU. */
LUMINARY_API LuminaryResult luminary_ext_render_samples(
  LuminaryHost* host, const uint32_t* pixels, uint32_t num_pixels, uint32_t first_sample, uint32_t num_samples, uint32_t samples_per_pass);
/* The reference's render loop, synchronous: `num_samples` more sample allocations of the whole frame. With
 * settings.enable_adaptive_sampling an allocation is one execution of the adaptive sampler's current stage (rates per 4x4 block, stages
 * built after adaptive_sampling_update_interval << stage allocations); otherwise one sample id per pixel. Outputs (recurring and
 * requested) are produced through the result image: local error minimisation and adaptive_sampling_output_mode apply. */
LUMINARY_API LuminaryResult luminary_ext_render(LuminaryHost* host, uint32_t num_samples);
/* Asynchronous rendering is started by luminary_host_start_new_render, as in the reference (host.c:406-414): the host's "Device" worker
 * then renders until the host is destroyed, restarting the accumulation whenever an edit dirties the integration, and the frontend polls
 * luminary_host_try_await_output / luminary_host_acquire_output. Additive controls: stop that worker (returns once the running iteration
 * has ended; the accumulated frame stays and the luminary_ext_render* calls can drive the loop synchronously), and query it. */
LUMINARY_API LuminaryResult luminary_ext_stop_render(LuminaryHost* host);
LUMINARY_API LuminaryResult luminary_ext_is_rendering(LuminaryHost* host, bool* rendering, uint32_t* accumulated_samples);
/* Planar float accumulators of the pixels given to luminary_ext_render_samples: first moment [R|G|B] and luminance second moment. */
LUMINARY_API LuminaryResult luminary_ext_get_accumulators(LuminaryHost* host, float* first_moment, float* second_moment, uint32_t* num_pixels);
/* Radiance = first moment / sample_count for the full frame (rgb interleaved, width*height*3 floats). */
LUMINARY_API LuminaryResult luminary_ext_get_radiance(LuminaryHost* host, float* rgb, uint32_t* sample_count, uint32_t width, uint32_t height);
/* out[0] closest-hit rays, [1] shadow rays, [2] light-BVH queries, [3] shaded vertices, [4] BVH nodes visited, [5] triangles tested. */
LUMINARY_API LuminaryResult luminary_ext_get_ray_counters(LuminaryHost* host, uint64_t out[8]);
/* The lumc context of this host (include/lum_core.h), for callers that drive passes themselves. */
LUMINARY_API void* luminary_ext_get_core_context(LuminaryHost* host);

#ifdef __cplusplus
}
#endif

#endif
