/*
 * lum_core.h - C ABI of the MI355X path-tracing core (libluminary_amd.so), hot path only.
 *
 * This is the boundary a host layer binds to (cgo / ctypes / plain C). Plain pointers and sizes, no C++ or torch types.
 * It replaces, for the triangle/BSDF/NEE path, the `device_*` surface the reference's device manager drives
 * (/root/reference paths):
 *   lumc_context_create/destroy     src/luminary/device/device.c:520-640 (device_create), :1700-1787 (device_destroy)
 *   lumc_scene_upload               device/device_manager.c:281-513 (scene -> device sync), device/device_mesh.c:19-51,
 *                                   device/optix_bvh.c:150-684 (acceleration structures, rebuilt here as BVH4),
 *                                   device/device_light.c:2363-2428 (light tree upload), device/device_bsdf.c:64-130 (LUTs)
 *   lumc_set_pixels                 replaces the sample partition of device/device_result_interface.c:107-175 by an
 *                                   image partition: each process renders the pixels it is given
 *   lumc_render                     device/device_renderer.c:488-575 (device_renderer_continue: the per-sample kernel queue)
 *   lumc_download_accumulators      device/device_result_interface.c:154-162 (download of the four moment planes)
 *   lumc_counters                   no reference equivalent (the reference does not count rays)
 * Every function returns 0 on success; on failure a message is available from lumc_last_error().
 */
#ifndef LUM_CORE_H
#define LUM_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct LumContext LumContext;

/*
 * Scene in the device format of the reference (what its kernels read). All pointers are HOST pointers; the library
 * copies what it needs. Layout is shared with the test oracle (oracle/oracle.h, checked by tests/test_layouts.py).
 */
typedef struct LumDeviceSceneView {
  uint32_t num_meshes;
  uint32_t num_instances;
  uint32_t num_materials;
  uint32_t num_lights;
  const uint32_t* mesh_tri_offset;    /* num_meshes + 1 */
  const float* vertices;              /* 3 per triangle, 16 B each (device_structs.h:270-273) */
  const uint32_t* tri_tex;            /* 16 B per triangle (device_structs.h:275-281) */
  const uint32_t* instance_mesh_ids;  /* num_instances */
  const float* instance_transforms;   /* 32 B each (device_structs.h:295-300) */
  const uint16_t* materials;          /* 32 B each (device_structs.h:202-223) */
  const uint8_t* light_tree_root;     /* device_utils.h:304-327; NULL without lights */
  const uint8_t* light_tree_nodes;    /* device_utils.h:283-302 */
  const uint32_t* light_tri_handles;  /* 2 per light */
  const float* light_bvh_tris;        /* 12 floats per light (device_light.h LightTreeBVHTriangle) */
  uint32_t num_light_tree_nodes;
  uint32_t num_textures;
  const uint32_t* bluenoise_2d;       /* 65536 texels */
  const uint16_t* lut_conductor;      /* 4 BSDF energy tables; pass NULL to have the library generate them on the GPU */
  const uint16_t* lut_glossy;
  const uint16_t* lut_dielectric;
  const uint16_t* lut_dielectric_inv;
  const uint32_t* texture_table;      /* 4 words per texture: first texel, width, height, gamma (float bits) */
  const uint32_t* texels;             /* RGBA8, r in the low byte; all textures back to back */
  uint32_t width, height, max_ray_depth, shading_mode;
  float cam_pos[3];
  float cam_rotation[4];
  float cam_fov, cam_aperture_size, cam_object_distance, cam_scale, cam_rr_threshold;
  uint32_t cam_aperture_shape, cam_aperture_blade_count;
  uint32_t sky_mode;
  float sky_constant_color[3];
} LumDeviceSceneView;

enum { LUMC_CNT_TRACE = 0, LUMC_CNT_SHADOW = 1, LUMC_CNT_LIGHT_BVH = 2, LUMC_CNT_VERTICES = 3, LUMC_CNT_NODES = 4, LUMC_CNT_TRIS = 5, LUMC_CNT_NODES_SHADOW = 6, LUMC_CNT_TRIS_SHADOW = 7,
       LUMC_CNT_NODES_LIGHT = 8, LUMC_CNT_TRIS_LIGHT = 9, LUMC_CNT_NODES_LDS = 10, LUMC_CNT_NODES_LDS_SHADOW = 11, LUMC_CNT_COUNT = 12 };
enum { LUMC_KERNEL_GENERATE = 0, LUMC_KERNEL_TRACE = 1, LUMC_KERNEL_SHADE = 2, LUMC_KERNEL_SHADOW = 3, LUMC_KERNEL_ACCUMULATE = 4, LUMC_KERNEL_LIGHT_QUERY = 5,
       LUMC_KERNEL_RESOLVE = 6, LUMC_KERNEL_OUTPUT = 7, LUMC_KERNEL_COUNT = 8 };

int lumc_context_create(int device_ordinal, LumContext** out);
void lumc_context_destroy(LumContext* ctx);
const char* lumc_last_error(const LumContext* ctx);

int lumc_scene_upload(LumContext* ctx, const LumDeviceSceneView* scene);
/* Copies the four energy tables (1024, 1024, 32768, 32768 u16) the context renders with to host memory. */
int lumc_download_luts(LumContext* ctx, uint16_t* conductor, uint16_t* glossy, uint16_t* dielectric, uint16_t* dielectric_inv);

/* Pixels (x + y*width) this context renders; NULL = the whole frame. (Re)allocates and zeroes internal accumulators. */
int lumc_set_pixels(LumContext* ctx, const uint32_t* pixels, uint32_t num_pixels);

/*
 * Renders sample ids [first_sample, first_sample + num_samples) of the context's pixels, `samples_per_pass` sample ids per
 * wavefront pass, and adds them into planar accumulators (first moment [R|G|B], second moment of the luminance).
 * d_first_moment/d_second_moment are DEVICE pointers owned by the caller (3*num_pixels and num_pixels floats) or NULL to use
 * the context's own. `stream` is a hipStream_t (NULL = default stream). Asynchronous: returns after enqueueing.
 */
int lumc_render(LumContext* ctx, uint32_t first_sample, uint32_t num_samples, uint32_t samples_per_pass, float* d_first_moment, float* d_second_moment,
                void* stream);
int lumc_synchronize(LumContext* ctx);
int lumc_clear_accumulators(LumContext* ctx);
int lumc_download_accumulators(LumContext* ctx, float* first_moment, float* second_moment);

int lumc_counters(LumContext* ctx, uint64_t out[LUMC_CNT_COUNT]);
int lumc_reset_counters(LumContext* ctx);
/* With profiling on, every kernel launch of lumc_render is bracketed by HIP events on its own stream. */
int lumc_set_profiling(LumContext* ctx, int enabled);
int lumc_kernel_times(LumContext* ctx, double total_ms[LUMC_KERNEL_COUNT], uint32_t launches[LUMC_KERNEL_COUNT]);

/*
 * Output chain (replaces device/device_output.c:178-343 + cuda/kernels.cuh:503-644 generate_final_image / convert_RGBF_to_ARGB8 and
 * cuda/tonemap.cuh): planar first moment [3 * src pixels] of a full frame -> ARGB8 words (b | g << 8 | r << 16 | 0xFF << 24) of the
 * requested size. The values are the reference's device-side camera/output state (device_structs.c:40-88): `exposure` is already
 * exp(camera.exposure), `passthrough` is set when the shading mode is not DEFAULT.
 */
typedef struct LumOutputParams {
  uint32_t src_width, src_height, dst_width, dst_height;
  float inv_sample_count, exposure;
  uint32_t tonemap, filter, dithering, purkinje, use_color_correction, passthrough;
  float purkinje_kappa1, purkinje_kappa2;
  float cc_h, cc_s, cc_v;
  float film_grain;
  float agx_slope, agx_power, agx_saturation;
} LumOutputParams;
/* d_first_moment NULL = the context's own accumulators (needs a full-frame pixel set). d_argb8: DEVICE buffer of dst pixels. */
int lumc_generate_output(LumContext* ctx, const LumOutputParams* params, const float* d_first_moment, uint32_t* d_argb8, void* stream);
/* Same, result copied to host memory; optionally also the display-referred float planes [3 * src pixels] (NULL to skip). */
int lumc_generate_output_host(LumContext* ctx, const LumOutputParams* params, const float* d_first_moment, uint32_t* argb8, float* frame_output);
/* Same with the first moment in HOST memory (e.g. a frame assembled from several GPUs); it is uploaded to a temporary buffer. */
int lumc_generate_output_from_host(LumContext* ctx, const LumOutputParams* params, const float* first_moment, uint32_t* argb8, float* frame_output);

/*
 * Adaptive sampling (replaces device/device_adaptive_sampler.c, cuda/adaptive_sampling.cuh, cuda/kernels.cuh:195-355 and the stage
 * schedule of device/device_renderer.c:350-375). Works on the whole frame with the context's own accumulators.
 * An *execution* is one sample allocation step of the reference: at stage 0 one sample of every pixel, at stage s >= 1
 * rate_s(block) samples of every pixel of a 4x4 block. Stage s lasts exactly `update_interval << s` executions; then the rates of
 * stage s+1 are computed from the measured variance (at most four such builds).
 */
typedef struct LumAdaptiveParams {
  uint32_t max_sampling_rate, avg_sampling_rate, update_interval; /* LuminaryRendererSettings::adaptive_sampling_* */
  float exposure;        /* exp(camera.exposure) when exposure aware, 0 otherwise (device_adaptive_sampler.c:57) */
  LumOutputParams tone;  /* tone curve of the compression factor: only tonemap and agx_* are read */
} LumAdaptiveParams;
typedef struct LumAdaptiveInfo {
  uint32_t stage_id, executions[5], num_blocks, blocks_x, blocks_y;
  uint32_t tasks_per_execution; /* paths one execution of the current stage generates (pixels of partial edge blocks included) */
  float variance_total;         /* sum of the block variances of the last stage build */
} LumAdaptiveInfo;
/* Starts (or restarts) adaptive rendering: needs a full-frame pixel set; clears the accumulators and the stage state. */
int lumc_adaptive_begin(LumContext* ctx, const LumAdaptiveParams* params);
/* Runs `executions` executions, building stages when they are due. Asynchronous like lumc_render except for one small download per
 * stage build (the task prefix, used to cut an execution into passes). */
int lumc_adaptive_render(LumContext* ctx, uint32_t executions, void* stream);
int lumc_adaptive_info(LumContext* ctx, LumAdaptiveInfo* out);
/* Per-block packed rates (byte s-1 = rate of stage s, minus one) and optionally the block variances of the last build. */
int lumc_adaptive_download(LumContext* ctx, uint32_t* stage_counts, float* block_variance);
/* Leaves adaptive mode (lumc_set_pixels does so too). */
int lumc_adaptive_end(LumContext* ctx);

/*
 * Result image (replaces accumulation_generate_result, cuda/accumulation.cuh:86-200): planar mean radiance [3 * W * H] from the
 * context's accumulators. mode: 0 beauty (with optional local error minimisation), 1 variance, 2 error, 3 sample distribution
 * (LuminaryAdaptiveSamplingOutputMode). While adaptive mode is active every pixel is normalised by its own sample count, otherwise by
 * `uniform_samples`. `exposure` = exp(camera.exposure) and `tone` feed the error image. d_result: DEVICE buffer or NULL for the
 * context's own (what lumc_generate_output reads when params->inv_sample_count == 1 and d_first_moment is that buffer).
 */
int lumc_generate_result(LumContext* ctx, uint32_t mode, uint32_t local_error_minimization, uint32_t uniform_samples, float exposure,
                         const LumOutputParams* tone, float* d_result, void* stream);
/* Same, copied to host memory [3 * W * H]. */
int lumc_generate_result_host(LumContext* ctx, uint32_t mode, uint32_t local_error_minimization, uint32_t uniform_samples, float exposure,
                              const LumOutputParams* tone, float* result);
/* Device address of the context's result image (valid after lumc_generate_result with d_result == NULL). */
const float* lumc_result_image(LumContext* ctx);

/* Closest-hit query on device buffers (float3 origins/dirs, optional uint2 ignore handles, uint3 out: instance, triangle, t bits). */
int lumc_trace_closest(LumContext* ctx, uint32_t num_rays, const float* d_origins, const float* d_dirs, const uint32_t* d_ignore, uint32_t* d_out, void* stream);
/* Closest hit of the camera ray of pixel (x, y) at sample id `sample_id`: out = instance id (0xFFFFFFFE = sky), triangle id, t bits, then the ray
 * direction x, y, z bits. Serves luminary_host_get_pixel_info (the reference fills a G-buffer during undersampled previews instead,
 * optix/optix_kernel_raytrace.cu:18-76). */
int lumc_pixel_query(LumContext* ctx, uint32_t x, uint32_t y, uint32_t sample_id, uint32_t out[6]);
/* Same with host buffers (copies in and out); used by the parity tests. */
int lumc_trace_closest_host(LumContext* ctx, uint32_t num_rays, const float* origins, const float* dirs, const uint32_t* ignore, uint32_t* out);

/* Builder of the per-mesh BVHs at the next lumc_scene_upload: 0 = binned SAH on the host (default: best trees), 1 = LBVH on the GPU
 * (Morton order + Karras hierarchy, lbvh.hip: fastest build, for scene edits and very large meshes). Images do not depend on it.
 * An environment variable LUM_BVH_BUILDER=lbvh selects 1 for new contexts. */
int lumc_set_bvh_builder(LumContext* ctx, int builder);
/* Seconds the per-mesh builds of the last lumc_scene_upload took (host wall clock, transfers included). */
double lumc_bvh_build_seconds(const LumContext* ctx);
/* Meshes of the last upload built by the SAH builder (out[0]; includes LBVH trees that came out too deep) and by LBVH (out[1]). */
int lumc_bvh_meshes_by_builder(const LumContext* ctx, uint32_t out[2]);
/* Sizes of the acceleration structures built by lumc_scene_upload: out[0] BLAS nodes, [1] BLAS triangles, [2] TLAS nodes, [3] light nodes. */
int lumc_bvh_stats(LumContext* ctx, uint64_t out[4]);
uint32_t lumc_scene_view_sizeof(void);

#ifdef __cplusplus
}
#endif

#endif
