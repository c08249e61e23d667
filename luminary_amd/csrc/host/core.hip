// C ABI of the path-tracing core (include/lum_core.h): context, scene upload, BVH construction, pass scheduling.
// Host-side counterpart of the reference's device layer for the hot path only:
//   device/device.c (context, streams, constant memory), device/device_work_buffers.c:54-117 (task/result buffers),
//   device/device_renderer.c:53-134, :488-575 (per-depth kernel queue), device/device_result_interface.c (moments).
// There is deliberately no CPU fallback: every entry point fails with an error string when HIP is unavailable.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cfloat>
#include <queue>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../../include/lum_core.h"
#include "../device/kernels.h"
#include "../device/dev_output.h"
#include "../device/dev_adaptive.h"
#include "../device/wavefront_table_impl.h"  // this translation unit holds the exact flavour; the fast one is csrc/device/wavefront_fast.hip
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>
#include "bvh_build.h"

using namespace lum;

// A mesh's bottom-level tree (node indices relative to the mesh, leaf ranges relative to its first triangle). The contexts of one process share them: a host
// with several devices hands every context the same meshes, the first one builds a mesh's tree, the others - and a later upload of the same mesh - take it
// from find_mesh_tree (below), and it is released with the last context that holds it.
struct MeshTree { Bvh4 bvh; bool built_on_gpu = false; };

struct LumContext {
  int device = 0;
  std::string error;
  const WavefrontKernels* wf = wavefront_kernels_fast();  // flavour of the wavefront kernels (lumc_set_flavour; LUM_FLAVOUR=exact|fast overrides the default)
  // device allocations of the scene by the part of it they belong to (lumc_scene_update frees and rebuilds a part at a time)
  enum AllocGroup { kGrpMesh = 0, kGrpInst, kGrpMat, kGrpLight, kGrpTex, kGrpConst, kGrpPart, kGrpOnce, kGrpCount };
  std::vector<void*> scene_allocs[kGrpCount];
  int alloc_group = kGrpOnce;
  // what a partial update needs again: the per-mesh trees (node indices relative to the mesh, leaf ranges relative to its first triangle) and boxes
  std::vector<std::shared_ptr<const MeshTree>> mesh_bvh;
  std::vector<Aabb> mesh_box;
  std::vector<uint32_t> sky_lut_key;  // the sky parameters the two sky tables were generated from
  float* d_bridge_lut = nullptr;      // the bridge sampler's vertex-count table (context-owned: scene.bridge_lut points here while bridges are possible)
  std::vector<float> bridge_lut_host; // its content, to notice a caller that hands over another table
  float4* d_sky_lut[2] = {nullptr, nullptr};
  DeviceScene scene{};
  bool has_scene = false;
  uint64_t bvh_stats[4] = {0, 0, 0, 0};
  int ambient_reuse = -1;         // -1 by flavour (fast: on), 0 off, 1 on (lumc_set_ambient_reuse; LUM_AMBIENT_REUSE)
  uint32_t shade_grid_rounds = LUM_SHADE_DYNAMIC ? 2 : 8;  // k_shade's grid as a multiple of its resident set (0: the common 2048-workgroup cap); LUM_SHADE_GRID
  int fused_resolve = 1;          // with the fast flavour's ambient reuse: k_shade resolves the previous depth's vertices itself (lumc_set_fused_resolve; LUM_FUSED_RESOLVE)
  void* fused_block = nullptr;    // what that needs beyond the usual work buffers: a third path queue, the parent words, a second set of NEE records, the fallback rays' items
  bool fused_records_stale = false;  // a queue's planes changed places (ray-sorting mode 3) since the records were written
  uint32_t fused_capacity = 0, fused_refused_capacity = 0;  // (the capacity its allocation last failed for: not tried again)
  uint2* d_sobol = nullptr;         // the pass's Sobol / Owen table (dev_sampler.h LUM_SOBOL_TABLE; wavefront_depths fills it)
  size_t sobol_entries = 0;
  int sobol_table = 1;              // LUM_SOBOL_TABLE_RT=0: the sampler hashes every number itself
  uint32_t* d_ended[2] = {nullptr, nullptr};  // a depth's vertices that no entry continues, by the depth's parity (k_shade lists them; the next depth's k_shade resolves them, or k_resolve_ended)
  // ... the next depth's k_shade (1) or k_resolve_ended after the depth's visibility pass (0; LUM_FUSED_ENDED=0). k_shade only takes the listed vertices
  // when its input comes through the cursor (kernels.h: LUM_SHADE_DYNAMIC): a build without it keeps the separate kernel, whatever is asked for.
  int fused_ended = LUM_SHADE_DYNAMIC ? 1 : 0;
  int fused_ended_default = LUM_SHADE_DYNAMIC ? 1 : 0;  // what lumc_set_fused_resolve(1) goes back to (the environment's choice, if any)
  FusedResolve* d_fused = nullptr;  // six records in device memory: the previous depth's queue (three buffers) and NEE records (two) by depth % 6
  NeeQueue nee2{};
  ShadowQueue fallback{};
  int bvh_builder = 3;            // 0 binned SAH on the host, 1 LBVH on the GPU, 2 PLOC on the GPU, 3 binned SAH on the GPU (default since round 4: the host builder's trees in a fifth of its time; a mesh it cannot take falls back to 0) (lumc_set_bvh_builder)
  bool top_order_by_area = false; // which nodes count as the top of the tree (staged in LDS): breadth first, or best first by box area (LUM_TOP_ORDER=area; measured: mixed)
  double bvh_build_seconds = 0.0; // bottom-level builds of the last lumc_scene_upload
  uint32_t bvh_meshes_by_builder[2] = {0, 0};  // meshes of the last upload built by SAH / by LBVH
  uint32_t lds_nodes = 0;         // nodes of the tree top every ray-kernel workgroup stages in LDS
  uint32_t trace_blocks = 256;    // persistent grid of the ray kernels
  // LUTs owned by the context when generated here
  uint16_t* d_luts[4] = {nullptr, nullptr, nullptr, nullptr};
  // pixels and accumulators
  uint32_t* d_pixels = nullptr;
  uint32_t num_pixels = 0;
  uint64_t pixels_hash = 0;       // of the pixel list in its order (lumc_set_pixels): the tile gather checks that the set IS the share of the deal it assumes
  float* d_first_moment = nullptr;
  float* d_second_moment = nullptr;
  // work buffers (sized for capacity paths)
  uint32_t capacity = 0;
  void* work_block = nullptr;
  PathQueue queue[3]{};           // [2]: only with the fused resolve (ensure_fused)
  NeeQueue nee{};
  ShadowQueue shadow{};
  uint32_t particle_lds_nodes = 0;
  VolumeQueue volume{};           // fog (dev_volume.h); allocated with the work block when the scene's fog is active
  CloudQueue cloud{};             // the cloud marches of a depth (kernels.h k_clouds_*); allocated with the work block when clouds are marched
  uint32_t work_shadow_kinds = 0; // visibility-ray kinds per path the work block was sized for (4, or 17 with fog)
  float4* d_results = nullptr;
  float* d_frame_output = nullptr;  // display-referred planes of the output chain [3 * W * H]
  uint32_t frame_output_pixels = 0;
  uint16_t* d_bluenoise_1d = nullptr;
  uint32_t* d_argb8 = nullptr;
  uint32_t argb8_pixels = 0;
  // adaptive sampling (dev_adaptive.h)
  struct Adaptive {
    bool active = false;
    LumAdaptiveParams params{};
    uint32_t blocks_x = 0, blocks_y = 0, num_blocks = 0;
    uint32_t stage_id = 0;
    uint32_t executions[kAdaptiveStages + 1] = {0, 0, 0, 0, 0};
    uint32_t* d_stage_counts = nullptr;
    uint32_t* d_block_tasks = nullptr;
    uint32_t* d_block_task_end = nullptr;
    float* d_block_variance = nullptr;
    float* d_partial = nullptr;   // chunk sums, then the total in the last element
    void* d_scan_temp = nullptr;
    size_t scan_temp_bytes = 0;
    std::vector<uint32_t> task_end;  // host copy of d_block_task_end: passes are cut at block boundaries
    float variance_total = 0.0f;
    uint8_t* d_block_mask = nullptr; // image-tile partition over GPUs: blocks this context renders (nullptr = all)
    bool build_pending = false;      // partitioned: a stage is due and waits for the block variances of all ranks
  } adaptive;
  uint32_t* d_cloud_noise[3] = {nullptr, nullptr, nullptr};  // the clouds' shape / detail / weather textures generated here (kept across scene uploads)
  bool cloud_noise_static = false;  // shape and detail do not depend on the seed
  uint32_t cloud_noise_seed = 0;
  bool cloud_noise_weather_valid = false;
  float4* d_sky_hdri = nullptr;     // baked sky (lumc_sky_hdri_build): dim x dim equirectangular, rgb + 0
  uint32_t sky_hdri_dim = 0;
  std::vector<float*> bloom_mips;  // mip chain of lumc_post_bloom, level i of (width >> (i + 1)) x (height >> (i + 1))
  uint32_t bloom_width = 0, bloom_height = 0;
  uint32_t* d_undersampling_pixels = nullptr;  // pixel list of the current undersampling iteration (lumc_render_undersampled)
  uint32_t undersampling_capacity = 0;
  std::vector<uint32_t> sky_hdri_key;  // what the bake was made from (sky parameters, origin, dim, samples): an unchanged key reuses it
  float* d_frame_result = nullptr;  // mean radiance planes of lumc_generate_result [3 * W * H]
  uint32_t frame_result_pixels = 0;
  // ray ordering (N1): keys + permutation, double-buffered for hipcub's radix sort; sized for the visibility items (4 per path)
  bool sync_debug = false;
  int sort_mode = 0;              // 0 queue order, 1 closest-hit rays of depth >= 1 traced through a sorted permutation, 2 visibility rays too, 3 the path queue physically reordered (lumc_set_ray_sorting, LUM_SORT)
  PathQueue sort_queue{};         // mode 3: the four state planes the reorder pass writes; swapped with the queue's own afterwards
  void* sort_planes[4] = {nullptr, nullptr, nullptr, nullptr};  // what was allocated for them (after swaps sort_queue may point into the work block)
  uint32_t sort_queue_capacity = 0;
  int sort_key = 0;               // 0 position-major (Morton cell | direction octant), 1 direction-major
  uint32_t* d_sort_keys[2] = {nullptr, nullptr};
  uint32_t* d_sort_vals[2] = {nullptr, nullptr};
  void* d_sort_temp = nullptr;
  size_t sort_temp_bytes = 0;
  uint32_t sort_capacity = 0;
  float world_lo[3] = {0, 0, 0}, world_hi[3] = {1, 1, 1};  // bounds of the top-level BVH
  // image-tile multi-GPU (lumc_comm_*, lumc_frame_assemble*): this rank's communicator and its [4][frame pixels] assembly buffer
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  float* d_frame = nullptr;
  uint32_t frame_capacity = 0;
  // tile gather (lumc_frame_gather*): this rank's padded [4][gather_stride] send buffer; on the root the [world][4][gather_stride] receive buffer and every
  // rank's pixel list [world][gather_stride] (0xFFFFFFFF = padding), keyed by (width, height, world)
  float* d_gather_send = nullptr;
  float* d_gather_recv = nullptr;
  uint32_t* d_gather_pixels = nullptr;
  uint32_t gather_stride = 0, gather_key[3] = {0, 0, 0};
  size_t gather_recv_floats = 0;
  bool use_frame = false;         // the result / output entry points read the assembled frame instead of this context's own accumulators
  uint32_t* d_ctrl = nullptr;     // kCtlStride control words per depth (+1 row), zeroed per pass; last row: cursor of lumc_trace_closest
  uint64_t* d_counters = nullptr;
  // profiling
  bool profiling = false;
  struct Stamp { hipEvent_t a, b; int kernel; };
  std::vector<Stamp> stamps;
  double kernel_ms[LUMC_KERNEL_COUNT] = {};
  uint32_t kernel_launches[LUMC_KERNEL_COUNT] = {};
};

namespace {

#define HIP_TRY(ctx, expr)                                                                                          \
  do {                                                                                                              \
    const hipError_t e__ = (expr);                                                                                  \
    if (e__ != hipSuccess) {                                                                                        \
      (ctx)->error = std::string(#expr) + " failed: " + hipGetErrorString(e__);                                     \
      return 1;                                                                                                     \
    }                                                                                                               \
  } while (0)

// [0, n) in contiguous chunks over the host's cores (per-triangle loops of the scene upload: 10 M triangles are 100 ms each on one core)
template <class F>
void host_parallel_for(size_t n, F&& fn) {
  const unsigned hc = std::thread::hardware_concurrency();
  const unsigned threads = (unsigned) std::min<size_t>(std::min(std::max(hc, 1u), 32u), std::max<size_t>(n / 65536, 1));
  if (threads <= 1) { fn((size_t) 0, n); return; }
  const size_t chunk = (n + threads - 1) / threads;
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < threads; t++) pool.emplace_back([&, t] { const size_t b = std::min(n, t * chunk), e = std::min(n, b + chunk); if (b < e) fn(b, e); });
  fn((size_t) 0, std::min(n, chunk));
  for (auto& th : pool) th.join();
}

// ---- the process's bottom-level trees by what they were built from: the mesh's triangles (two 64-bit hashes of the vertex words, chunk by chunk so that the
// value does not depend on the number of threads), the builder asked for and every LUM_* variable of the environment (the builders' knobs) ----
struct MeshTreeKey {
  uint64_t h0, h1, env;
  uint32_t tris; int builder;
  bool operator<(const MeshTreeKey& o) const { return std::tie(h0, h1, env, tris, builder) < std::tie(o.h0, o.h1, o.env, o.tris, o.builder); }
};
static std::mutex g_mesh_tree_mutex;
static std::map<MeshTreeKey, std::weak_ptr<const MeshTree>> g_mesh_trees;
extern "C" char** environ;

static MeshTreeKey mesh_tree_key(const float* tri_vertices, uint32_t tris, int builder) {
  constexpr size_t kChunk = 65536;  // triangles (12 floats each)
  const size_t chunks = ((size_t) tris + kChunk - 1) / kChunk;
  std::vector<uint64_t> part(2 * chunks);
  host_parallel_for(chunks, [&](size_t b, size_t e) {
    for (size_t c = b; c < e; c++) {
      const size_t first = c * kChunk, last = std::min<size_t>((size_t) tris, first + kChunk);
      uint64_t a = 0x9E3779B97F4A7C15ull ^ c, z = 0xC2B2AE3D27D4EB4Full + c;
      for (size_t w = first * 6; w < last * 6; w++) {
        uint64_t x;
        std::memcpy(&x, reinterpret_cast<const char*>(tri_vertices) + w * 8, 8);
        a = (a ^ x) * 0x100000001B3ull; a ^= a >> 29;
        z = (z + x) * 0xFF51AFD7ED558CCDull; z ^= z >> 32;
      }
      part[2 * c] = a; part[2 * c + 1] = z;
    }
  });
  MeshTreeKey k{0xCBF29CE484222325ull, 0x84222325CBF29CE4ull, 0xCBF29CE484222325ull, tris, builder};
  for (size_t c = 0; c < chunks; c++) { k.h0 = (k.h0 ^ part[2 * c]) * 0x100000001B3ull; k.h1 = (k.h1 + part[2 * c + 1]) * 0xFF51AFD7ED558CCDull; k.h1 ^= k.h1 >> 32; }
  for (char** e = environ; e && *e; e++)
    if (std::strncmp(*e, "LUM_", 4) == 0) for (const char* p = *e; *p; p++) k.env = (k.env ^ (uint8_t) *p) * 0x100000001B3ull;
  return k;
}
static std::shared_ptr<const MeshTree> find_mesh_tree(const MeshTreeKey& k) {
  if (const char* e = getenv("LUM_BVH_SHARE")) if (atoi(e) == 0) return nullptr;  // every upload builds (tools/lbvh_bench.py times the builders this way)
  std::lock_guard<std::mutex> lock(g_mesh_tree_mutex);
  auto it = g_mesh_trees.find(k);
  if (it == g_mesh_trees.end()) return nullptr;
  std::shared_ptr<const MeshTree> t = it->second.lock();
  if (!t) g_mesh_trees.erase(it);
  return t;
}
static void keep_mesh_tree(const MeshTreeKey& k, const std::shared_ptr<const MeshTree>& t) {
  std::lock_guard<std::mutex> lock(g_mesh_tree_mutex);
  for (auto it = g_mesh_trees.begin(); it != g_mesh_trees.end();) it = it->second.expired() ? g_mesh_trees.erase(it) : std::next(it);
  g_mesh_trees[k] = t;
}

template <typename T>
int upload(LumContext* ctx, const T* host, size_t count, const T** out, bool scene_owned = true) {
  *out = nullptr;
  if (count == 0 || host == nullptr) return 0;
  void* d = nullptr;
  HIP_TRY(ctx, hipMalloc(&d, sizeof(T) * count));
  if (scene_owned) ctx->scene_allocs[ctx->alloc_group].push_back(d);
  HIP_TRY(ctx, hipMemcpy(d, host, sizeof(T) * count, hipMemcpyHostToDevice));
  *out = (const T*) d;
  return 0;
}

void free_group(LumContext* ctx, int group) {
  for (void* p : ctx->scene_allocs[group]) (void) hipFree(p);
  ctx->scene_allocs[group].clear();
}
void free_scene(LumContext* ctx) {
  for (int g = 0; g < LumContext::kGrpCount; g++) free_group(ctx, g);
  for (int i = 0; i < 4; i++) { if (ctx->d_luts[i]) (void) hipFree(ctx->d_luts[i]); ctx->d_luts[i] = nullptr; }
  for (int i = 0; i < 2; i++) { if (ctx->d_sky_lut[i]) (void) hipFree(ctx->d_sky_lut[i]); ctx->d_sky_lut[i] = nullptr; }
  ctx->sky_lut_key.clear();
  if (ctx->d_bridge_lut) (void) hipFree(ctx->d_bridge_lut);
  ctx->d_bridge_lut = nullptr; ctx->bridge_lut_host.clear();
  ctx->mesh_bvh.clear(); ctx->mesh_box.clear();
  ctx->has_scene = false;
}

void free_adaptive(LumContext* ctx) {
  LumContext::Adaptive& a = ctx->adaptive;
  void* bufs[] = {a.d_stage_counts, a.d_block_tasks, a.d_block_task_end, a.d_block_variance, a.d_partial, a.d_scan_temp, a.d_block_mask};
  for (void* b : bufs) if (b) (void) hipFree(b);
  a = LumContext::Adaptive();
}

void free_work(LumContext* ctx) {
  if (ctx->work_block) (void) hipFree(ctx->work_block);
  ctx->work_block = nullptr;
  if (ctx->fused_block) (void) hipFree(ctx->fused_block);
  ctx->fused_block = nullptr; ctx->fused_capacity = 0; ctx->d_fused = nullptr;
  ctx->fused_refused_capacity = 0;  // memory may have been freed since the refusal: the next pass asks again
  ctx->queue[2] = PathQueue{}; ctx->nee2 = NeeQueue{}; ctx->fallback = ShadowQueue{};
  for (int k = 0; k < 3; k++) ctx->queue[k].parent = nullptr;
  ctx->capacity = 0;
  ctx->work_shadow_kinds = 0;
  ctx->cloud = CloudQueue{};
  // the reorder pass's planes (ray-sorting mode 3) trade places with the queues' own: they go with them
  for (int k = 0; k < 4; k++) { if (ctx->sort_planes[k]) (void) hipFree(ctx->sort_planes[k]); ctx->sort_planes[k] = nullptr; }
  ctx->sort_queue = PathQueue{}; ctx->sort_queue_capacity = 0;
  if (ctx->d_sobol) (void) hipFree(ctx->d_sobol);
  ctx->d_sobol = nullptr; ctx->sobol_entries = 0;
}

int ensure_work(LumContext* ctx, uint32_t paths) {
  // in a volume a path can ask for 19 visibility rays in the in-scattering pass (15 bridge segments, sun and ambient in up to two segments each)
  // instead of 4 at a surface (6 with an ocean: the second segments of the sun and ambient samples of a vertex under water)
  const bool volumes = ctx->scene.fog_active || ctx->scene.ocean_active;
  const uint32_t kinds = volumes ? kVolumeShadowKinds : 4u;
  const bool clouds = ctx->scene.cloud_active && ctx->scene.sky_mode == kSkyDefault;
  if (clouds && paths >= (1u << 30)) { ctx->error = "pass too large for the cloud march list (2^30 paths)"; return 1; }
  if (paths <= ctx->capacity && kinds <= ctx->work_shadow_kinds && (!clouds || ctx->cloud.items)) return 0;
  if (paths < ctx->capacity) paths = ctx->capacity;
  free_work(ctx);
  // per path: 2 queues x 68 B + NEE 84 B + result 16 B + up to `kinds` visibility rays x (48 B + 16 B result) + 4 B light-query index
  // (+ the volumes' 96 B of in-scattering records, 4 B scattering-event index and 48 B of water-surface factors of the surface vertices)
  const size_t n = paths;
  const size_t bytes = n * (2 * 68 + 84 + 16 + (size_t) kinds * 64 + 4 + (kinds > 4u ? 100 + 48 : 0) + (clouds ? 3 * (4 + 16 + 4) : 0)) + 56 * 256;
  HIP_TRY(ctx, hipMalloc(&ctx->work_block, bytes));
  char* p = (char*) ctx->work_block;
  auto take = [&](size_t sz) { char* r = p; p += (sz + 255) & ~(size_t) 255; return r; };  // keeps every array 256-byte aligned
  for (int k = 0; k < 2; k++) {
    ctx->queue[k].origin_t = (float4*) take(n * 16);
    ctx->queue[k].dir_slot = (float4*) take(n * 16);
    ctx->queue[k].aux      = (uint4*) take(n * 16);
    ctx->queue[k].hit_id   = (uint4*) take(n * 16);
    ctx->queue[k].hit_scene_tri = (uint32_t*) take(n * 4);
  }
  ctx->nee.geo_color_light = (float4*) take(n * 16);
  ctx->nee.bsdf_ray_prob   = (float4*) take(n * 16);
  ctx->nee.bsdf_weight_sum = (float4*) take(n * 16);
  ctx->nee.ambient         = (uint4*) take(n * 16);
  ctx->nee.sun             = (uint4*) take(n * 16);
  ctx->nee.amb_path        = (uint32_t*) take(n * 4);
  ctx->d_results           = (float4*) take(n * 16);
  ctx->shadow.origin_dist  = (float4*) take(kinds * n * 16);
  ctx->shadow.dir_out      = (float4*) take(kinds * n * 16);
  ctx->shadow.ids          = (uint4*) take(kinds * n * 16);
  ctx->shadow.vis          = (float4*) take(kinds * n * 16);
  ctx->shadow.light_items  = (uint32_t*) take(n * 4);
  ctx->shadow.capacity     = paths;
  ctx->volume = VolumeQueue{};
  ctx->nee.sun_water = nullptr; ctx->nee.amb_t1 = nullptr; ctx->nee.amb_t2 = nullptr;
  if (kinds > 4u) {
    ctx->volume.bridge = (float4*) take(n * 16);
    ctx->volume.sky    = (uint4*) take(n * 16);
    ctx->volume.weight = (float4*) take(n * 16);
    ctx->volume.sun_water = (float4*) take(n * 16);
    ctx->volume.amb_t1 = (float4*) take(n * 16);
    ctx->volume.amb_t2 = (float4*) take(n * 16);
    ctx->volume.items  = (uint32_t*) take(n * 4);
    ctx->nee.sun_water = (float4*) take(n * 16);
    ctx->nee.amb_t1    = (float4*) take(n * 16);
    ctx->nee.amb_t2    = (float4*) take(n * 16);
  }
  ctx->cloud = CloudQueue{};
  if (clouds) {  // per path up to three marches: list entry, result, distance of the first cloud
    ctx->cloud.items    = (uint32_t*) take(3 * n * 4);
    ctx->cloud.result   = (float4*) take(3 * n * 16);
    ctx->cloud.hit_dist = (float*) take(3 * n * 4);
    ctx->cloud.capacity = paths;
  }
  ctx->work_shadow_kinds = kinds;
  ctx->capacity = paths;
  return 0;
}

// The fused resolve's own buffers (FusedResolve, kernels.h), sized like the work buffers: per path a third queue entry (68 B), three parent words, a second
// set of NEE records (84 B) and one fallback ray (48 B + its vertex's index).
// The six records k_shade reads the previous depth through (device memory): rewritten whenever a queue's planes move.
int upload_fused_records(LumContext* ctx, hipStream_t stream) {
  FusedResolve by_depth[6];
  for (int d = 0; d < 6; d++) {  // depth d is shaded from queue d % 3 with the records d & 1: the depth before it lives in queue (d + 2) % 3 and the other record set
    by_depth[d].prev = ctx->queue[(d + 2) % 3];
    by_depth[d].nee_prev = (d & 1) ? ctx->nee : ctx->nee2;
    by_depth[d].fallback = ctx->fallback;
    by_depth[d].ended = ctx->d_ended[d & 1];
    by_depth[d].ended_prev = ctx->d_ended[(d & 1) ^ 1];
  }
  HIP_TRY(ctx, hipStreamSynchronize(stream));  // a pass still reading the old records on a non-blocking stream is not ordered against the null-stream copy below
  HIP_TRY(ctx, hipMemcpy(ctx->d_fused, by_depth, sizeof(by_depth), hipMemcpyHostToDevice));
  ctx->fused_records_stale = false;
  return 0;
}
int ensure_fused(LumContext* ctx, hipStream_t stream) {
  if (ctx->fused_block && ctx->fused_capacity == ctx->capacity) return ctx->fused_records_stale ? upload_fused_records(ctx, stream) : 0;
  if (ctx->fused_refused_capacity == ctx->capacity) return 1;
  if (ctx->fused_block) (void) hipFree(ctx->fused_block);
  ctx->fused_block = nullptr; ctx->fused_capacity = 0;
  const size_t n = ctx->capacity;
  const size_t bytes = n * (68 + 3 * 4 + 84 + 48 + 4 + 2 * 4) + 27 * 256 + 6 * sizeof(FusedResolve);
  if (hipMalloc(&ctx->fused_block, bytes) != hipSuccess) { ctx->fused_block = nullptr; ctx->fused_refused_capacity = ctx->capacity; return 1; }
  char* p = (char*) ctx->fused_block;
  auto take = [&](size_t sz) { char* r = p; p += (sz + 255) & ~(size_t) 255; return r; };
  PathQueue& q = ctx->queue[2];
  q.origin_t = (float4*) take(n * 16); q.dir_slot = (float4*) take(n * 16); q.aux = (uint4*) take(n * 16); q.hit_id = (uint4*) take(n * 16);
  q.hit_scene_tri = (uint32_t*) take(n * 4);
  for (int k = 0; k < 3; k++) ctx->queue[k].parent = (uint32_t*) take(n * 4);
  NeeQueue& e = ctx->nee2;
  e = NeeQueue{};
  e.geo_color_light = (float4*) take(n * 16); e.bsdf_ray_prob = (float4*) take(n * 16); e.bsdf_weight_sum = (float4*) take(n * 16);
  e.ambient = (uint4*) take(n * 16); e.sun = (uint4*) take(n * 16); e.amb_path = (uint32_t*) take(n * 4);
  ShadowQueue& f = ctx->fallback;
  f.origin_dist = (float4*) take(n * 16); f.dir_out = (float4*) take(n * 16); f.ids = (uint4*) take(n * 16);
  f.light_items = (uint32_t*) take(n * 4);
  f.vis = ctx->shadow.vis;  // the undecided samples' answers go where the depth's own ambient answers went: kind 2 of the previous depth's words
  f.capacity = ctx->shadow.capacity;
  ctx->d_ended[0] = (uint32_t*) take(n * 4); ctx->d_ended[1] = (uint32_t*) take(n * 4);
  ctx->d_fused = (FusedResolve*) take(6 * sizeof(FusedResolve));
  ctx->fused_capacity = ctx->capacity;
  return upload_fused_records(ctx, stream);
}

constexpr uint32_t kCtrlRows = 68;  // depths 0..63, one row past the last depth, spare, lumc_trace_closest

inline uint32_t grid_for(uint32_t n) {
  const uint32_t blocks = (n + kBlock - 1) / kBlock;
  return blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);  // 256 CUs x 8 resident blocks, grid-stride beyond that
}

// k_shade: its workgroups are grid-stride loops of equal length, three of them resident per CU (3 waves per SIMD). With the common cap of 2048 workgroups that
// was 2.67 rounds of the 768 resident places, paid as 3; the grid is now a whole number of rounds, and eight of them: the shorter a workgroup, the shorter the
// kernel's tail (hall, k_shade per 3 steps: 2048 workgroups 380 ms | 1 round 399 | 2: 383 | 3: 376 | 4: 372 | 6: 369 | 8: 368 | 12: 366 | 24: 371; the Example-class
// scene is best at 6-8). LUM_SHADE_GRID=<rounds> (0: the 2048 cap).
inline uint32_t shade_grid(const LumContext* ctx, uint32_t n) {
  const uint32_t blocks = (n + kBlock - 1) / kBlock;
  const uint32_t resident = ctx->trace_blocks * 3u;  // trace_blocks = the device's CUs (one persistent ray workgroup each)
  // (not with an ocean: k_shade<.., ocean> keeps a scratch frame and its workgroups cost more to start - Example-class scene with an ocean, 8 rounds: shade +2 %)
#if LUM_SHADE_DYNAMIC
  // input by cursor (kernels.h): twice the resident set. The second half only starts when the queue is used up and leaves at once; what it buys is that every
  // place is taken from the start (hall, k_shade per 3 steps: fixed shares 345 ms | cursor, 1 x resident 337 | 2 x: 333 | 3 x: 331 | 4 x: 332; the scan and
  // the Example-class scene, whose launches are short, are level at 1-2 x and lose 3-5 % at 3-4 x: profiles/r05_ab_experiments.txt). LUM_SHADE_GRID=<rounds>.
  const uint32_t rounds = ctx->shade_grid_rounds;
  const uint32_t cap = resident * (rounds ? rounds : 1u);
#else
  const uint32_t rounds = ctx->scene.ocean_active ? 0u : ctx->shade_grid_rounds;
  const uint32_t cap = rounds ? resident * rounds : 2048u;
#endif
  return blocks < 1 ? 1 : std::min(blocks, cap);
}

// Persistent ray kernels: one workgroup per CU (kTraceBlock threads, its own LDS copy of the tree top); waves pull work from a cursor.
inline uint32_t grid_persistent(const LumContext* ctx, uint32_t n) {
  const uint32_t tb = ctx->wf->trace_block;
  const uint32_t blocks = (n + tb - 1) / tb;
  return blocks < 1 ? 1 : (blocks > ctx->trace_blocks ? ctx->trace_blocks : blocks);
}

struct Launch {
  LumContext* ctx;
  hipStream_t stream;
  int kernel;
  size_t idx = (size_t) -1;
  Launch(LumContext* c, hipStream_t s, int k) : ctx(c), stream(s), kernel(k) {
    if (!ctx->profiling) return;
    LumContext::Stamp st;
    st.kernel = k;
    if (hipEventCreate(&st.a) != hipSuccess || hipEventCreate(&st.b) != hipSuccess) return;
    (void) hipEventRecord(st.a, stream);
    ctx->stamps.push_back(st);
    idx = ctx->stamps.size() - 1;
  }
  ~Launch() {
    if (idx != (size_t) -1) (void) hipEventRecord(ctx->stamps[idx].b, stream);
    if (ctx->sync_debug) {  // LUM_SYNC_DEBUG=1: name the launch group a device fault belongs to
      const hipError_t e = hipStreamSynchronize(stream);
      std::fprintf(stderr, "[lum] launch group %d: %s\n", kernel, hipGetErrorString(e));
    }
  }
};

int resolve_stamps(LumContext* ctx) {
  for (auto& st : ctx->stamps) {
    float ms = 0.0f;
    if (hipEventSynchronize(st.b) == hipSuccess && hipEventElapsedTime(&ms, st.a, st.b) == hipSuccess) {
      ctx->kernel_ms[st.kernel] += ms;
      ctx->kernel_launches[st.kernel]++;
    }
    (void) hipEventDestroy(st.a);
    (void) hipEventDestroy(st.b);
  }
  ctx->stamps.clear();
  return 0;
}

Aabb tri_box(const float* a, const float* b, const float* c) {
  Aabb box;
  for (int k = 0; k < 3; k++) { box.lo[k] = std::min(a[k], std::min(b[k], c[k])); box.hi[k] = std::max(a[k], std::max(b[k], c[k])); }
  return box;
}

// Host twin of the device transform (cuda/math.cuh:393-489) used to bound instances for the top-level BVH.
// World->object matrix of one instance, float arithmetic in the exact operation order of dev_math.h's xf_rel_inv applied to the
// unit vectors (column j = xf_rel_inv(e_j)); rows[i] = (m_i0, m_i1, m_i2, translation_i). The kernels map rays with these rows
// (dev_trace.h traverse_scene); the oracle derives the same 12 numbers on its own (oracle/o_trace.h tracer_init).
void instance_inverse_rows(const float* p, float4 rows[3]) {
  uint32_t a, b;
  std::memcpy(&a, p + 6, 4); std::memcpy(&b, p + 7, 4);
  const float ux = 1.0f - ((a & 0xFFFFu) * (1.0f / 0x7FFF)), uy = 1.0f - ((a >> 16) * (1.0f / 0x7FFF));
  const float uz = 1.0f - ((b & 0xFFFFu) * (1.0f / 0x7FFF)), s = ((b >> 16) * (1.0f / 0x7FFF)) - 1.0f;
  const float inv_scale[3] = {1.0f / p[3], 1.0f / p[4], 1.0f / p[5]};
  float col[3][3];
  for (int j = 0; j < 3; j++) {
    const float vx = (j == 0 ? 1.0f : 0.0f) * inv_scale[0], vy = (j == 1 ? 1.0f : 0.0f) * inv_scale[1], vz = (j == 2 ? 1.0f : 0.0f) * inv_scale[2];
    const float duv = ux * vx + uy * vy + uz * vz, duu = ux * ux + uy * uy + uz * uz;
    const float cx = uy * vz - uz * vy, cy = uz * vx - ux * vz, cz = ux * vy - uy * vx;
    const float k0 = 2.0f * duv, k1 = s * s - duu, k2 = 2.0f * s;
    col[j][0] = (ux * k0 + vx * k1) + cx * k2;
    col[j][1] = (uy * k0 + vy * k1) + cy * k2;
    col[j][2] = (uz * k0 + vz * k1) + cz * k2;
  }
  for (int i = 0; i < 3; i++) rows[i] = make_float4(col[0][i], col[1][i], col[2][i], p[i]);
}

// World box of an object-space box under the inverse of the map above (double precision, then padded): the top-level BVH must
// bound the geometry exactly where the ray mapping puts it.
bool instance_world_box(const float4 rows[3], const Aabb& ob, Aabb& wb) {
  const double m[3][3] = {{rows[0].x, rows[0].y, rows[0].z}, {rows[1].x, rows[1].y, rows[1].z}, {rows[2].x, rows[2].y, rows[2].z}};
  const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                     m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
  if (!(std::fabs(det) > 0.0) || !std::isfinite(det)) return false;
  double f[3][3];
  f[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; f[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det; f[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
  f[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det; f[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; f[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
  f[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; f[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det; f[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
  const double t[3] = {rows[0].w, rows[1].w, rows[2].w};
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (int c = 0; c < 8; c++) {
    const double v[3] = {(c & 1) ? ob.hi[0] : ob.lo[0], (c & 2) ? ob.hi[1] : ob.lo[1], (c & 4) ? ob.hi[2] : ob.lo[2]};
    for (int k = 0; k < 3; k++) {
      const double w = f[k][0] * v[0] + f[k][1] * v[1] + f[k][2] * v[2] + t[k];
      lo[k] = std::min(lo[k], w); hi[k] = std::max(hi[k], w);
    }
  }
  for (int k = 0; k < 3; k++) {
    // float rounding of the ray mapping (a few ulp of the coordinates involved) is covered by a relative pad
    const double pad = 4e-6 * std::max(std::fabs(lo[k]), std::fabs(hi[k])) + 4e-6 * std::fabs(t[k]) + 1e-6 * (hi[k] - lo[k]) + 1e-30;
    wb.lo[k] = (float) (lo[k] - pad); wb.hi[k] = (float) (hi[k] + pad);
    wb.lo[k] = std::nextafter(wb.lo[k], -FLT_MAX); wb.hi[k] = std::nextafter(wb.hi[k], FLT_MAX);
  }
  return true;
}

// ---- ray ordering (north star: "ray-sorted wavefront"; the reference sorts its tasks by hit type every depth, cuda/kernels.cuh:391-484) ----
// Key = Morton code of the ray origin's cell in a 64^3 grid over the scene bounds (18 bits) combined with the direction's octant (3 bits).
// Flavour-neutral: the order in which a queue is traced never changes a result (every path owns its slots), it only decides which rays
// share a wave, a CU's L1 and an XCD's L2.
struct SortGrid { float lo[3], scale[3]; uint32_t direction_major; };

__device__ __forceinline__ uint32_t spread6(uint32_t v) {  // 6 bits -> every third bit
  v &= 0x3Fu;
  v = (v | (v << 8)) & 0x300Fu;
  v = (v | (v << 4)) & 0x30C3u;
  v = (v | (v << 2)) & 0x9249u;
  return v;
}

__global__ __launch_bounds__(256) void k_ray_sort_keys(const float4* __restrict__ origin, const float4* __restrict__ dir, const uint32_t* __restrict__ count, uint32_t capacity,
                                                       SortGrid g, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t n = min(*count, capacity);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < capacity; i += gridDim.x * 256u) {
    uint32_t key = 0x200000u;  // beyond the live items: above every live key (21 bits), so they sort to the end whether or not the sort is stable
    if (i < n) {
      const float4 o = origin[i], d = dir[i];
      const uint32_t cx = (uint32_t) fminf(fmaxf((o.x - g.lo[0]) * g.scale[0], 0.0f), 63.0f), cy = (uint32_t) fminf(fmaxf((o.y - g.lo[1]) * g.scale[1], 0.0f), 63.0f),
                     cz = (uint32_t) fminf(fmaxf((o.z - g.lo[2]) * g.scale[2], 0.0f), 63.0f);
      const uint32_t morton = spread6(cx) | (spread6(cy) << 1) | (spread6(cz) << 2);
      const uint32_t octant = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
      key = g.direction_major ? ((octant << 18) | morton) : ((morton << 3) | octant);
    }
    keys[i] = key;
    vals[i] = i;
  }
}

int ensure_sort(LumContext* ctx, uint32_t items) {
  if (items <= ctx->sort_capacity) return 0;
  for (int k = 0; k < 2; k++) {
    if (ctx->d_sort_keys[k]) (void) hipFree(ctx->d_sort_keys[k]);
    if (ctx->d_sort_vals[k]) (void) hipFree(ctx->d_sort_vals[k]);
    ctx->d_sort_keys[k] = ctx->d_sort_vals[k] = nullptr;
  }
  if (ctx->d_sort_temp) (void) hipFree(ctx->d_sort_temp);
  ctx->d_sort_temp = nullptr; ctx->sort_capacity = 0;
  for (int k = 0; k < 2; k++) {
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_sort_keys[k], sizeof(uint32_t) * (size_t) items));
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_sort_vals[k], sizeof(uint32_t) * (size_t) items));
  }
  hipcub::DoubleBuffer<uint32_t> keys(ctx->d_sort_keys[0], ctx->d_sort_keys[1]), vals(ctx->d_sort_vals[0], ctx->d_sort_vals[1]);
  HIP_TRY(ctx, hipcub::DeviceRadixSort::SortPairs(nullptr, ctx->sort_temp_bytes, keys, vals, (int) items, 0, 22, (hipStream_t) 0));
  HIP_TRY(ctx, hipMalloc(&ctx->d_sort_temp, std::max<size_t>(ctx->sort_temp_bytes, 16)));
  ctx->sort_capacity = items;
  return 0;
}

// Mode 3: the path state of the live paths gathered through the sorted permutation into a second set of planes, written in order - the pass every
// later kernel of the depth then reads coherently (trace, shade, and through the order of the appends the visibility rays and the next depth).
__global__ __launch_bounds__(256) void k_permute_queue(PathQueue src, PathQueue dst, const uint32_t* __restrict__ order, const uint32_t* __restrict__ count, uint32_t capacity) {
  const uint32_t n = min(*count, capacity);
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const uint32_t j = order[i];
    const float4 o = src.origin_t[j], d = src.dir_slot[j];
    const uint4 a = src.aux[j], h = src.hit_id[j];
    dst.origin_t[i] = o; dst.dir_slot[i] = d; dst.aux[i] = a; dst.hit_id[i] = h;
  }
}

int ensure_sort_queue(LumContext* ctx, uint32_t items) {
  if (items == ctx->sort_queue_capacity) return 0;
  for (int k = 0; k < 4; k++) { if (ctx->sort_planes[k]) (void) hipFree(ctx->sort_planes[k]); ctx->sort_planes[k] = nullptr; }
  ctx->sort_queue = PathQueue{}; ctx->sort_queue_capacity = 0;
  for (int k = 0; k < 4; k++) HIP_TRY(ctx, hipMalloc(&ctx->sort_planes[k], 16 * (size_t) items));
  ctx->sort_queue.origin_t = (float4*) ctx->sort_planes[0]; ctx->sort_queue.dir_slot = (float4*) ctx->sort_planes[1];
  ctx->sort_queue.aux = (uint4*) ctx->sort_planes[2]; ctx->sort_queue.hit_id = (uint4*) ctx->sort_planes[3];
  ctx->sort_queue_capacity = items;
  return 0;
}

// Sorted order of the first *count items of (origin, dir); returns the permutation (device pointer) or nullptr on failure.
const uint32_t* sort_rays(LumContext* ctx, hipStream_t stream, const float4* origin, const float4* dir, const uint32_t* count, uint32_t capacity) {
  if (ensure_sort(ctx, capacity)) return nullptr;
  SortGrid g;
  for (int k = 0; k < 3; k++) { g.lo[k] = ctx->world_lo[k]; const float e = ctx->world_hi[k] - ctx->world_lo[k]; g.scale[k] = e > 0.0f ? 64.0f / e : 0.0f; }
  g.direction_major = ctx->sort_key == 1 ? 1u : 0u;
  Launch l(ctx, stream, LUMC_KERNEL_SORT);
  const uint32_t blocks = std::min<uint32_t>((capacity + 255u) / 256u, 4096u);
  hipLaunchKernelGGL(k_ray_sort_keys, dim3(blocks ? blocks : 1), dim3(256), 0, stream, origin, dir, count, capacity, g, ctx->d_sort_keys[0], ctx->d_sort_vals[0]);
  hipcub::DoubleBuffer<uint32_t> keys(ctx->d_sort_keys[0], ctx->d_sort_keys[1]), vals(ctx->d_sort_vals[0], ctx->d_sort_vals[1]);
  size_t bytes = ctx->sort_temp_bytes;
  if (hipcub::DeviceRadixSort::SortPairs(ctx->d_sort_temp, bytes, keys, vals, (int) capacity, 0, 22, stream) != hipSuccess) return nullptr;
  return vals.Current();
}

}  // namespace

extern "C" {

int lumc_context_create(int device_ordinal, LumContext** out) {
  if (!out) return 1;
  *out = nullptr;
  LumContext* ctx = new LumContext();
  ctx->device = device_ordinal;
  if (const char* b = getenv("LUM_BVH_BUILDER")) ctx->bvh_builder = (std::strcmp(b, "lbvh") == 0) ? 1 : (std::strcmp(b, "ploc") == 0) ? 2 : (std::strcmp(b, "sah") == 0 || std::strcmp(b, "host") == 0) ? 0 : 3;
  if (const char* o = getenv("LUM_TOP_ORDER")) ctx->top_order_by_area = std::strcmp(o, "area") == 0;
  if (const char* e = getenv("LUM_SORT")) ctx->sort_mode = atoi(e);
  if (const char* e = getenv("LUM_SYNC_DEBUG")) ctx->sync_debug = atoi(e) != 0;
  if (const char* e = getenv("LUM_SORT_KEY")) ctx->sort_key = atoi(e);
  if (const char* e = getenv("LUM_AMBIENT_REUSE")) ctx->ambient_reuse = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("LUM_FUSED_RESOLVE")) ctx->fused_resolve = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("LUM_SHADE_GRID")) ctx->shade_grid_rounds = (uint32_t) atoi(e);
  if (const char* e = getenv("LUM_SOBOL_TABLE_RT")) ctx->sobol_table = atoi(e) != 0 ? 1 : 0;
  if (const char* e = getenv("LUM_FUSED_ENDED")) ctx->fused_ended = ctx->fused_ended_default = (LUM_SHADE_DYNAMIC && atoi(e) != 0) ? 1 : 0;
  if (const char* f = getenv("LUM_FLAVOUR")) ctx->wf = (std::strcmp(f, "exact") == 0) ? wavefront_kernels_exact() : wavefront_kernels_fast();
  *out = ctx;
  int count = 0;
  HIP_TRY(ctx, hipGetDeviceCount(&count));
  if (device_ordinal < 0 || device_ordinal >= count) { ctx->error = "no such HIP device"; return 1; }
  HIP_TRY(ctx, hipSetDevice(device_ordinal));
  HIP_TRY(ctx, (hipError_t) wavefront_kernels_exact()->init_sampler_seeds());  // per device: module globals live on each GPU
  HIP_TRY(ctx, (hipError_t) wavefront_kernels_fast()->init_sampler_seeds());
  HIP_TRY(ctx, hipMalloc((void**) &ctx->d_ctrl, sizeof(uint32_t) * kCtlStride * kCtrlRows));
  HIP_TRY(ctx, hipMemset(ctx->d_ctrl, 0, sizeof(uint32_t) * kCtlStride * kCtrlRows));
  HIP_TRY(ctx, hipMalloc((void**) &ctx->d_counters, sizeof(uint64_t) * LUMC_CNT_COUNT));
  HIP_TRY(ctx, hipMemset(ctx->d_counters, 0, sizeof(uint64_t) * LUMC_CNT_COUNT));
  return 0;
}

void lumc_context_destroy(LumContext* ctx) {
  if (!ctx) return;
  (void) hipSetDevice(ctx->device);
  (void) hipDeviceSynchronize();
  resolve_stamps(ctx);
  free_scene(ctx);
  free_work(ctx);
  if (ctx->d_pixels) (void) hipFree(ctx->d_pixels);
  if (ctx->d_first_moment) (void) hipFree(ctx->d_first_moment);
  if (ctx->d_second_moment) (void) hipFree(ctx->d_second_moment);
  if (ctx->comm) { (void) ncclCommDestroy(ctx->comm); ctx->comm = nullptr; }
  if (ctx->d_frame) (void) hipFree(ctx->d_frame);
  if (ctx->d_ctrl) (void) hipFree(ctx->d_ctrl);
  for (int k = 0; k < 2; k++) { if (ctx->d_sort_keys[k]) (void) hipFree(ctx->d_sort_keys[k]); if (ctx->d_sort_vals[k]) (void) hipFree(ctx->d_sort_vals[k]); }
  if (ctx->d_sort_temp) (void) hipFree(ctx->d_sort_temp);
  if (ctx->d_frame_output) (void) hipFree(ctx->d_frame_output);
  if (ctx->d_bluenoise_1d) (void) hipFree(ctx->d_bluenoise_1d);
  if (ctx->d_argb8) (void) hipFree(ctx->d_argb8);
  if (ctx->d_counters) (void) hipFree(ctx->d_counters);
  if (ctx->d_frame_result) (void) hipFree(ctx->d_frame_result);
  if (ctx->d_gather_send) (void) hipFree(ctx->d_gather_send);
  if (ctx->d_gather_recv) (void) hipFree(ctx->d_gather_recv);
  if (ctx->d_gather_pixels) (void) hipFree(ctx->d_gather_pixels);
  if (ctx->d_sky_hdri) (void) hipFree(ctx->d_sky_hdri);
  for (uint32_t*& t : ctx->d_cloud_noise) { if (t) (void) hipFree(t); t = nullptr; }
  if (ctx->d_undersampling_pixels) (void) hipFree(ctx->d_undersampling_pixels);
  for (float* m : ctx->bloom_mips) (void) hipFree(m);
  free_adaptive(ctx);
  delete ctx;
}

const char* lumc_last_error(const LumContext* ctx) { return ctx ? ctx->error.c_str() : "null context"; }
uint32_t lumc_scene_view_sizeof(void) { return (uint32_t) sizeof(LumDeviceSceneView); }

// ---- 8-wide quantised nodes (Bvh8Node, dev_scene.h) from a finished 4-wide tree ----
// Every surviving node absorbs inner children, largest surface area first, while at most eight children result; the absorbed nodes
// disappear. `keep`: nodes referenced from outside (the root, the roots of the meshes), which survive by construction since nothing has them as
// a child. Survivors keep their relative order (the 4-wide array is already "top of the tree first"). Returns false when the traversal stack
// (dev_trace.h kStackSize, up to seven pushes per level) could overflow.
struct WideChild { float lo[3], hi[3]; uint32_t ref; };
static float box_area(const WideChild& c) {
  const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
  return dx * dy + dy * dz + dz * dx;
}
static uint32_t node_children(const Bvh4Node& n, WideChild* out) {
  uint32_t m = 0;
  for (int k = 0; k < 4; k++) {
    if (n.child[k] == kBvhEmpty) continue;
    out[m] = WideChild{{n.lo_x[k], n.lo_y[k], n.lo_z[k]}, {n.hi_x[k], n.hi_y[k], n.hi_z[k]}, n.child[k]};
    m++;
  }
  return m;
}
static bool widen_to_bvh8(const std::vector<Bvh4Node>& in, const std::vector<uint32_t>& keep, std::vector<Bvh8Node>& out, std::vector<uint32_t>& old_to_new,
                          std::vector<uint32_t>* levels_out) {  // levels_out[r]: levels of the 8-wide tree below keep[r]
  const size_t n = in.size();
  std::vector<std::vector<WideChild>> wide(n);
  std::vector<uint8_t> survives(n, 0);
  std::vector<uint32_t> level(n, 0), root_of(n, 0);
  std::vector<uint32_t> queue;
  for (size_t r = 0; r < keep.size(); r++) if (keep[r] < n && !survives[keep[r]]) { survives[keep[r]] = 1; level[keep[r]] = 1; root_of[keep[r]] = (uint32_t) r; queue.push_back(keep[r]); }
  std::vector<uint32_t> max_level(keep.size(), 0);
  for (size_t head = 0; head < queue.size(); head++) {
    const uint32_t i = queue[head];
    max_level[root_of[i]] = std::max(max_level[root_of[i]], level[i]);
    WideChild list[8];
    uint32_t m = node_children(in[i], list);
    for (;;) {
      int best = -1;
      float best_area = -1.0f;
      uint32_t best_m = 0;
      for (uint32_t k = 0; k < m; k++) {
        const uint32_t ref = list[k].ref;
        if (ref & kBvhLeafBit) continue;
        uint32_t cm = 0;
        for (int j = 0; j < 4; j++) if (in[ref].child[j] != kBvhEmpty) cm++;
        if (cm == 0 || m - 1 + cm > 8) continue;
        const float a = box_area(list[k]);
        if (a > best_area) { best_area = a; best = (int) k; best_m = cm; }
      }
      if (best < 0) break;
      WideChild sub[4];
      const uint32_t cm = node_children(in[list[best].ref], sub);
      (void) best_m;
      list[best] = sub[0];
      for (uint32_t j = 1; j < cm; j++) list[m++] = sub[j];
    }
    wide[i].assign(list, list + m);
    for (uint32_t k = 0; k < m; k++) {
      const uint32_t ref = list[k].ref;
      if (!(ref & kBvhLeafBit) && !survives[ref]) { survives[ref] = 1; level[ref] = level[i] + 1; root_of[ref] = root_of[i]; queue.push_back(ref); }
    }
  }
  old_to_new.assign(n, 0xFFFFFFFFu);
  uint32_t count = 0;
  for (size_t i = 0; i < n; i++) if (survives[i]) old_to_new[i] = count++;
  out.assign(count, Bvh8Node{});
  for (size_t i = 0; i < n; i++) {
    if (!survives[i]) continue;
    Bvh8Node& o = out[old_to_new[i]];
    std::memset(&o, 0, sizeof(o));
    const std::vector<WideChild>& ch = wide[i];
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (const WideChild& c : ch) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], c.lo[a]); hi[a] = std::max(hi[a], c.hi[a]); }
    if (ch.empty()) for (int a = 0; a < 3; a++) { lo[a] = 0.0f; hi[a] = 0.0f; }
    float scale[3];
    for (int a = 0; a < 3; a++) {
      o.origin[a] = lo[a];
      const float extent = hi[a] - lo[a];
      int e = -126;
      if (extent > 0.0f && std::isfinite(extent)) {
        int ex;
        std::frexp(extent / 255.0f, &ex);  // extent / 255 = f * 2^ex with f in [0.5, 1): 2^ex >= extent / 255
        e = std::max(-126, std::min(127, ex));
      }
      // the quotient must stay below 256 after the float subtraction's rounding: one step coarser when it does not
      while (e < 127 && std::ceil((hi[a] - lo[a]) / std::ldexp(1.0f, e)) > 255.0f) e++;
      o.exp[a] = (uint8_t) (e + 127);
      scale[a] = std::ldexp(1.0f, e);
    }
    uint8_t* qlo[3] = {o.lo_x, o.lo_y, o.lo_z};
    uint8_t* qhi[3] = {o.hi_x, o.hi_y, o.hi_z};
    for (uint32_t k = 0; k < 8; k++) {
      if (k >= ch.size()) {
        o.child[k] = kBvhEmpty;
        for (int a = 0; a < 3; a++) { qlo[a][k] = 255; qhi[a][k] = 0; }
        continue;
      }
      const WideChild& c = ch[k];
      o.child[k] = (c.ref & kBvhLeafBit) ? c.ref : old_to_new[c.ref];
      for (int a = 0; a < 3; a++) {
        const float l = std::floor((c.lo[a] - lo[a]) / scale[a]), h = std::ceil((c.hi[a] - lo[a]) / scale[a]);
        qlo[a][k] = (uint8_t) std::max(0.0f, std::min(255.0f, l));
        qhi[a][k] = (uint8_t) std::max(0.0f, std::min(255.0f, h));
      }
    }
  }
  if (levels_out) *levels_out = max_level;
  return true;
}

// ---- 8-wide nodes with children in octant slots (Bvh8oNode, dev_scene.h) from a finished 4-wide tree ----
// Widening as above (a node absorbs its largest inner children while at most eight result). Then, per node: the children take octant slots (greedy:
// the (child, slot) pair with the largest projection of the child's offset from the node's centre onto the slot's diagonal first, as the reference's
// bvh.c:1093-1145), the inner children become consecutive nodes in slot order (nodes are numbered in the order the widening queue meets them: the top
// of the tree first), and the leaf slots' primitives become one consecutive run per node: `leaf_order` is the new order of the leaf items (triangles of
// the bottom level / records of the top level) as indices into the old one. `is_top[i]`: node i belongs to the top level (its leaves are instance
// records). Returns false when a node's leaves do not fit the 5-bit offsets (cannot happen with <= 4 primitives per leaf: 8 x 4 = 32).
struct Bvh8oResult {
  std::vector<Bvh8oNode> nodes;
  std::vector<uint32_t> old_to_new;        // for the `keep` roots (others: 0xFFFFFFFF when absorbed)
  std::vector<uint32_t> tri_order, top_leaf_order;
  std::vector<uint32_t> levels;            // per keep root
  uint32_t top_nodes = 0;
};
static bool build_bvh8o(const std::vector<Bvh4Node>& in, const std::vector<uint8_t>& is_top, const std::vector<uint32_t>& keep, size_t num_tris, size_t num_top_leaves, Bvh8oResult& out) {
  const size_t n = in.size();
  std::vector<std::vector<WideChild>> wide(n);
  std::vector<uint8_t> survives(n, 0);
  std::vector<uint32_t> level(n, 0), root_of(n, 0), queue;
  out.old_to_new.assign(n, 0xFFFFFFFFu);
  for (size_t r = 0; r < keep.size(); r++) if (keep[r] < n && !survives[keep[r]]) { survives[keep[r]] = 1; level[keep[r]] = 1; root_of[keep[r]] = (uint32_t) r; out.old_to_new[keep[r]] = (uint32_t) queue.size(); queue.push_back(keep[r]); }
  out.levels.assign(keep.size(), 0);
  out.tri_order.clear(); out.top_leaf_order.clear();
  out.tri_order.reserve(num_tris); out.top_leaf_order.reserve(num_top_leaves);
  out.nodes.clear();
  out.top_nodes = 0;
  for (size_t head = 0; head < queue.size(); head++) {
    const uint32_t i = queue[head];
    out.levels[root_of[i]] = std::max(out.levels[root_of[i]], level[i]);
    WideChild list[8];
    uint32_t m = node_children(in[i], list);
    for (;;) {
      int best = -1;
      float best_area = -1.0f;
      for (uint32_t k = 0; k < m; k++) {
        const uint32_t ref = list[k].ref;
        if (ref & kBvhLeafBit) continue;
        uint32_t cm = 0;
        for (int j = 0; j < 4; j++) if (in[ref].child[j] != kBvhEmpty) cm++;
        if (cm == 0 || m - 1 + cm > 8) continue;
        const float a = box_area(list[k]);
        if (a > best_area) { best_area = a; best = (int) k; }
      }
      if (best < 0) break;
      WideChild sub[4];
      const uint32_t cm = node_children(in[list[best].ref], sub);
      list[best] = sub[0];
      for (uint32_t j = 1; j < cm; j++) list[m++] = sub[j];
    }
    // the node's box, slots
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t k = 0; k < m; k++) for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], list[k].lo[a]); hi[a] = std::max(hi[a], list[k].hi[a]); }
    if (m == 0) for (int a = 0; a < 3; a++) { lo[a] = 0.0f; hi[a] = 0.0f; }
    int slot_of[8];
    {
      float cost[8][8];
      for (uint32_t k = 0; k < m; k++)
        for (int sl = 0; sl < 8; sl++) {
          float c = 0.0f;
          for (int a = 0; a < 3; a++) c += (((sl >> a) & 1) ? 1.0f : -1.0f) * (0.5f * (list[k].lo[a] + list[k].hi[a]) - 0.5f * (lo[a] + hi[a]));
          cost[k][sl] = c;
        }
      bool used_k[8] = {false, false, false, false, false, false, false, false}, used_s[8] = {false, false, false, false, false, false, false, false};
      for (uint32_t it = 0; it < m; it++) {
        int bk = -1, bs = -1;
        float best = -FLT_MAX;
        for (uint32_t k = 0; k < m; k++) if (!used_k[k]) for (int sl = 0; sl < 8; sl++) if (!used_s[sl] && (bk < 0 || cost[k][sl] > best)) { best = cost[k][sl]; bk = (int) k; bs = sl; }
        used_k[bk] = used_s[bs] = true;
        slot_of[bk] = bs;
      }
    }
    int child_in_slot[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    for (uint32_t k = 0; k < m; k++) child_in_slot[slot_of[k]] = (int) k;
    Bvh8oNode o;
    std::memset(&o, 0, sizeof(o));
    float scale[3];
    for (int a = 0; a < 3; a++) {
      o.origin[a] = lo[a];
      const float extent = hi[a] - lo[a];
      int e = -126;
      if (extent > 0.0f && std::isfinite(extent)) {
        int ex;
        std::frexp(extent / 255.0f, &ex);
        e = std::max(-126, std::min(127, ex));
      }
      while (e < 127 && std::ceil((hi[a] - lo[a]) / std::ldexp(1.0f, e)) > 255.0f) e++;  // the quotient must stay below 256 after the subtraction's rounding
      o.exp[a] = (uint8_t) (e + 127);
      scale[a] = std::ldexp(1.0f, e);
    }
    uint8_t* qlo[3] = {o.lo_x, o.lo_y, o.lo_z};
    uint8_t* qhi[3] = {o.hi_x, o.hi_y, o.hi_z};
    const bool top = is_top[i] != 0;
    if (top) out.top_nodes++;
    std::vector<uint32_t>& leaf_order = top ? out.top_leaf_order : out.tri_order;
    o.leaf_base = (uint32_t) leaf_order.size();
    o.child_base = (uint32_t) queue.size();  // where this node's first inner child is about to be numbered
    for (int sl = 0; sl < 8; sl++) {
      const int k = child_in_slot[sl];
      if (k < 0) { for (int a = 0; a < 3; a++) { qlo[a][sl] = 255; qhi[a][sl] = 0; } continue; }
      const WideChild& c = list[k];
      for (int a = 0; a < 3; a++) {
        const float l = std::floor((c.lo[a] - lo[a]) / scale[a]), h = std::ceil((c.hi[a] - lo[a]) / scale[a]);
        qlo[a][sl] = (uint8_t) std::max(0.0f, std::min(255.0f, l));
        qhi[a][sl] = (uint8_t) std::max(0.0f, std::min(255.0f, h));
      }
      if (c.ref & kBvhLeafBit) {
        const uint32_t first = c.ref & 0x0FFFFFFFu, count = ((c.ref >> 28) & 7u) + 1u;
        const uint32_t offset = (uint32_t) leaf_order.size() - o.leaf_base;
        if (offset > 31u || count > 4u) return false;
        o.meta[sl] = (uint8_t) (offset | ((count - 1u) << 5));
        for (uint32_t j = 0; j < count; j++) leaf_order.push_back(first + j);
      }
      else {
        o.imask |= (uint8_t) (1u << sl);
        survives[c.ref] = 1; level[c.ref] = level[i] + 1; root_of[c.ref] = root_of[i];
        out.old_to_new[c.ref] = (uint32_t) queue.size();
        queue.push_back(c.ref);
      }
    }
    out.nodes.push_back(o);
  }
  return out.tri_order.size() == num_tris && out.top_leaf_order.size() == num_top_leaves;
}

// ---- 64-byte quantised nodes (Bvh4QNode, dev_scene.h) from a finished 4-wide tree: node for node, same indices ----
static void quantise_bvh4(std::vector<Bvh4Node>& nodes) {
  std::vector<Bvh4QNode> out(nodes.size());
  for (size_t i = 0; i < nodes.size(); i++) {
    const Bvh4Node& n = nodes[i];
    Bvh4QNode& o = out[i];
    std::memset(&o, 0, sizeof(o));
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    bool any = false;
    for (int k = 0; k < 4; k++) {
      if (n.child[k] == kBvhEmpty) continue;
      any = true;
      const float clo[3] = {n.lo_x[k], n.lo_y[k], n.lo_z[k]}, chi[3] = {n.hi_x[k], n.hi_y[k], n.hi_z[k]};
      for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], clo[a]); hi[a] = std::max(hi[a], chi[a]); }
    }
    if (!any) for (int a = 0; a < 3; a++) { lo[a] = 0.0f; hi[a] = 0.0f; }
    float scale[3];
    for (int a = 0; a < 3; a++) {
      o.origin[a] = lo[a];
      const float extent = hi[a] - lo[a];
      int e = -126;
      if (extent > 0.0f && std::isfinite(extent)) {
        int ex;
        std::frexp(extent / 255.0f, &ex);  // extent / 255 = f * 2^ex with f in [0.5, 1): 2^ex >= extent / 255
        e = std::max(-126, std::min(127, ex));
      }
      while (e < 127 && std::ceil((hi[a] - lo[a]) / std::ldexp(1.0f, e)) > 255.0f) e++;  // the quotient must stay below 256 after the subtraction's rounding
      o.exp[a] = (uint8_t) (e + 127);
      scale[a] = std::ldexp(1.0f, e);
    }
    uint8_t* qlo[3] = {o.lo_x, o.lo_y, o.lo_z};
    uint8_t* qhi[3] = {o.hi_x, o.hi_y, o.hi_z};
    for (int k = 0; k < 4; k++) {
      o.child[k] = n.child[k];
      if (n.child[k] == kBvhEmpty) { for (int a = 0; a < 3; a++) { qlo[a][k] = 255; qhi[a][k] = 0; } continue; }
      const float clo[3] = {n.lo_x[k], n.lo_y[k], n.lo_z[k]}, chi[3] = {n.hi_x[k], n.hi_y[k], n.hi_z[k]};
      for (int a = 0; a < 3; a++) {
        const float l = std::floor((clo[a] - lo[a]) / scale[a]), h = std::ceil((chi[a] - lo[a]) / scale[a]);
        qlo[a][k] = (uint8_t) std::max(0.0f, std::min(255.0f, l));
        qhi[a][k] = (uint8_t) std::max(0.0f, std::min(255.0f, h));
      }
    }
  }
  // the array keeps its element type for the upload: the first half of it now holds the 64-byte nodes
  std::memcpy(nodes.data(), out.data(), out.size() * sizeof(Bvh4QNode));
}

// The clouds' noise textures (device_cloud.c:62-101): shape and detail once per context, the weather map per seed.
static int ensure_cloud_noise(LumContext* ctx, uint32_t seed) {
  const size_t counts[3] = {(size_t) kCloudShapeRes * kCloudShapeRes * kCloudShapeRes, (size_t) kCloudDetailRes * kCloudDetailRes * kCloudDetailRes,
                            (size_t) kCloudWeatherRes * kCloudWeatherRes};
  for (int k = 0; k < 3; k++)
    if (!ctx->d_cloud_noise[k]) HIP_TRY(ctx, hipMalloc((void**) &ctx->d_cloud_noise[k], sizeof(uint32_t) * counts[k]));
  if (!ctx->cloud_noise_static) {
    hipLaunchKernelGGL(exact::k_cloud_noise_shape, dim3(2048), dim3(256), 0, 0, ctx->d_cloud_noise[0], (uint32_t) kCloudShapeRes);
    hipLaunchKernelGGL(exact::k_cloud_noise_detail, dim3(128), dim3(256), 0, 0, ctx->d_cloud_noise[1], (uint32_t) kCloudDetailRes);
    HIP_TRY(ctx, hipGetLastError());
    ctx->cloud_noise_static = true;
  }
  if (!ctx->cloud_noise_weather_valid || ctx->cloud_noise_seed != seed) {
    hipLaunchKernelGGL(exact::k_cloud_noise_weather, dim3(2048), dim3(256), 0, 0, ctx->d_cloud_noise[2], (uint32_t) kCloudWeatherRes, (float) seed);
    HIP_TRY(ctx, hipGetLastError());
    ctx->cloud_noise_seed = seed; ctx->cloud_noise_weather_valid = true;
  }
  HIP_TRY(ctx, hipDeviceSynchronize());
  return 0;
}
int lumc_cloud_noise_generate(LumContext* ctx, uint32_t seed, uint32_t* shape, uint32_t* detail, uint32_t* weather) {
  if (!ctx) return 1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ensure_cloud_noise(ctx, seed)) return 1;
  uint32_t* out[3] = {shape, detail, weather};
  const size_t counts[3] = {(size_t) kCloudShapeRes * kCloudShapeRes * kCloudShapeRes, (size_t) kCloudDetailRes * kCloudDetailRes * kCloudDetailRes,
                            (size_t) kCloudWeatherRes * kCloudWeatherRes};
  for (int k = 0; k < 3; k++)
    if (out[k]) HIP_TRY(ctx, hipMemcpy(out[k], ctx->d_cloud_noise[k], sizeof(uint32_t) * counts[k], hipMemcpyDeviceToHost));
  return 0;
}

// The particle tree (device_particle.c:23-131, optix_bvh.c's particle GAS / IAS): one bottom-level tree over the 2 x count triangles of the unit
// cell, and a top level over its 25 x 25 x 25 integer translations, instance id = (xi * 25 + yi) * 25 + zi as the reference numbers them. The
// top-level leaves hold the translation as an exact affine map (rows of the identity), so entering an instance is one subtraction per axis.
static int build_particle_tree(LumContext* ctx, const LumDeviceSceneView* v, DeviceScene& sc) {
  sc.particle_bvh_nodes = nullptr; sc.particle_tris = nullptr; sc.particle_leaves = nullptr; sc.particle_tlas_num_nodes = 0; sc.particle_normals = nullptr;
  if (!sc.particles_active) return 0;
  if (!v->particle_vertices || !v->particle_normals) { ctx->error = "lumc_scene_upload: active particles without particle_vertices / particle_normals"; return 1; }
  const uint32_t nt = 2u * sc.particles_count;
  std::vector<Aabb> tri_boxes(nt);
  Aabb cell{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
  for (uint32_t t = 0; t < nt; t++) {
    const float* p = v->particle_vertices + (size_t) t * 12;
    tri_boxes[t] = tri_box(p, p + 4, p + 8);
    for (int k = 0; k < 3; k++) { cell.lo[k] = std::min(cell.lo[k], tri_boxes[t].lo[k]); cell.hi[k] = std::max(cell.hi[k], tri_boxes[t].hi[k]); }
  }
  constexpr int kDim = 25;  // PARTICLES_BLOCK_DIM
  std::vector<Aabb> boxes((size_t) kDim * kDim * kDim);
  std::vector<float> offsets(3 * boxes.size());
  uint32_t id = 0;
  for (int xi = 0; xi < kDim; xi++)
    for (int yi = 0; yi < kDim; yi++)
      for (int zi = 0; zi < kDim; zi++, id++) {
        const float off[3] = {(float) (xi - (kDim >> 1)), (float) (yi - (kDim >> 1)), (float) (zi - (kDim >> 1))};
        for (int k = 0; k < 3; k++) { offsets[3 * id + k] = off[k]; boxes[id].lo[k] = cell.lo[k] + off[k] - 1e-5f; boxes[id].hi[k] = cell.hi[k] + off[k] + 1e-5f; }
      }
  Bvh4 tlas = build_bvh4(boxes.data(), (uint32_t) boxes.size(), 1, 16);
  Bvh4 blas = build_bvh4(tri_boxes.data(), nt, kBvhLeafMaxTri, 26);
  if (tlas.nodes.empty() || blas.nodes.empty()) { ctx->error = "particle BVH exceeds the traversal's depth limits"; return 1; }
  std::vector<Bvh4Node> nodes = tlas.nodes;
  const uint32_t base = (uint32_t) nodes.size();
  for (Bvh4Node n : blas.nodes) {
    for (int k = 0; k < 4; k++) if (n.child[k] != kBvhEmpty && !(n.child[k] & kBvhLeafBit)) n.child[k] += base;
    nodes.push_back(n);
  }
  std::vector<BvhTri> tris((size_t) nt + 1);
  std::memset(tris.data(), 0, sizeof(BvhTri) * tris.size());
  for (uint32_t i = 0; i < nt; i++) {
    const uint32_t t = blas.prims[i];
    const float* p = v->particle_vertices + (size_t) t * 12;
    for (int k = 0; k < 3; k++) { tris[i].p0[k] = p[k]; tris[i].e1[k] = p[4 + k] - p[k]; tris[i].e2[k] = p[8 + k] - p[k]; }
    tris[i].id = t; tris[i].scene_index = t; tris[i].albedo_tex = kBvhTriNoTexture;
  }
  std::vector<float4> leaves(4 * tlas.prims.size() + 4);
  for (size_t i = 0; i < tlas.prims.size(); i++) {
    const uint32_t inst = tlas.prims[i];
    leaves[4 * i + 0] = make_float4(1.0f, 0.0f, 0.0f, offsets[3 * inst + 0]);
    leaves[4 * i + 1] = make_float4(0.0f, 1.0f, 0.0f, offsets[3 * inst + 1]);
    leaves[4 * i + 2] = make_float4(0.0f, 0.0f, 1.0f, offsets[3 * inst + 2]);
    const uint32_t words[4] = {inst, base, 0u, 0u};
    std::memcpy(&leaves[4 * i + 3], words, 16);
  }
  uint32_t mesh_root_index = base;
#if LUM_BVH8
  {
    std::vector<Bvh8Node> wide;
    std::vector<uint32_t> map;
    std::vector<uint32_t> levels;
    widen_to_bvh8(nodes, {0u, base}, wide, map, &levels);
    if (7u * (levels[0] + levels[1]) + 4u > (uint32_t) kStackSize) { ctx->error = "particle BVH too deep for the traversal stack"; return 1; }
    mesh_root_index = map[base];
    static_assert(sizeof(Bvh8Node) == sizeof(Bvh4Node), "same 128-byte slot");
    nodes.resize(wide.size());
    std::memcpy(nodes.data(), wide.data(), wide.size() * sizeof(Bvh8Node));
    for (size_t i = 0; i < tlas.prims.size(); i++) { const uint32_t words[4] = {tlas.prims[i], mesh_root_index, 0u, 0u}; std::memcpy(&leaves[4 * i + 3], words, 16); }
  }
#endif
#if LUM_BVH8O
  {
    std::vector<uint8_t> is_top(nodes.size(), 0);
    for (uint32_t i = 0; i < base; i++) is_top[i] = 1;
    Bvh8oResult r8;
    if (!build_bvh8o(nodes, is_top, {0u, base}, nt, tlas.prims.size(), r8)) { ctx->error = "8-wide conversion of the particle tree failed"; return 1; }
    if (r8.levels[0] + r8.levels[1] + 8u > (uint32_t) kStackSize) { ctx->error = "particle BVH too deep for the traversal stack"; return 1; }
    mesh_root_index = r8.old_to_new[base];
    std::vector<BvhTri> reordered(tris.size());
    std::memset(reordered.data(), 0, sizeof(BvhTri) * reordered.size());
    for (size_t i = 0; i < r8.tri_order.size(); i++) reordered[i] = tris[r8.tri_order[i]];
    tris.swap(reordered);
    std::vector<float4> moved(leaves.size());
    for (size_t i = 0; i < r8.top_leaf_order.size(); i++) {
      for (int k = 0; k < 3; k++) moved[4 * i + k] = leaves[4 * (size_t) r8.top_leaf_order[i] + k];
      const uint32_t words[4] = {tlas.prims[r8.top_leaf_order[i]], mesh_root_index, 0u, 0u};
      std::memcpy(&moved[4 * i + 3], words, 16);
    }
    leaves.swap(moved);
    nodes.resize(r8.nodes.size());
    std::memcpy(nodes.data(), r8.nodes.data(), r8.nodes.size() * sizeof(Bvh8oNode));
  }
#endif
#if LUM_BVH4Q
  quantise_bvh4(nodes);
#endif
  if (upload(ctx, nodes.data(), nodes.size(), &sc.particle_bvh_nodes)) return 1;
  if (upload(ctx, tris.data(), tris.size(), &sc.particle_tris)) return 1;
  if (upload(ctx, leaves.data(), leaves.size(), &sc.particle_leaves)) return 1;
  if (upload(ctx, (const float4*) v->particle_normals, (size_t) sc.particles_count, &sc.particle_normals)) return 1;
  sc.particle_tlas_num_nodes = (uint32_t) tlas.nodes.size();
  sc.particle_num_leaves = (uint32_t) (leaves.size() / 4);
  ctx->particle_lds_nodes = (uint32_t) std::min<size_t>(ctx->lds_nodes, nodes.size());
  return 0;
}

// The scene on the device, part by part (lumc_scene_update). Every part frees what it allocated before; parts that are not dirty keep their device
// arrays and the fields of ctx->scene that point at them.
static int scene_update(LumContext* ctx, const LumDeviceSceneView* v, unsigned dirty) {
  DeviceScene& sc = ctx->scene;
  if (!v->bluenoise_2d) { ctx->error = "scene has no blue-noise mask"; return 1; }
  if (v->max_ray_depth > 63) { ctx->error = "max_ray_depth exceeds 63 (6-bit field, device_structs.h:9)"; return 1; }
  const uint32_t total_tris = v->num_meshes ? v->mesh_tri_offset[v->num_meshes] : 0;
  if (dirty & LUMC_DIRTY_MESHES) dirty |= LUMC_DIRTY_INSTANCES;  // the assembled node array holds the per-mesh trees
#if LUM_BVH8O
  if (dirty & LUMC_DIRTY_INSTANCES) dirty |= LUMC_DIRTY_MESHES;  // (experiment) the 8-wide conversion reorders the triangles by the assembled tree: an instance edit rebuilds all of it
#endif
  if (dirty & LUMC_DIRTY_PARTICLES) dirty |= LUMC_DIRTY_CONSTANTS;
  const bool dirty_meshes = (dirty & LUMC_DIRTY_MESHES) != 0, dirty_instances = (dirty & LUMC_DIRTY_INSTANCES) != 0, dirty_lights = (dirty & LUMC_DIRTY_LIGHTS) != 0;
  ctx->has_scene = false;  // until this update has gone through

  if (dirty_meshes) {
    free_group(ctx, LumContext::kGrpMesh); ctx->alloc_group = LumContext::kGrpMesh;
    if (upload(ctx, v->mesh_tri_offset, (size_t) v->num_meshes + 1, &sc.mesh_tri_offset)) return 1;
    if (upload(ctx, (const float4*) v->vertices, (size_t) total_tris * 3, &sc.vertices)) return 1;
    if (upload(ctx, (const uint4*) v->tri_tex, (size_t) total_tris, &sc.tri_tex)) return 1;
  }
  if (dirty_instances) {
    free_group(ctx, LumContext::kGrpInst); ctx->alloc_group = LumContext::kGrpInst;
    if (upload(ctx, v->instance_mesh_ids, v->num_instances, &sc.instance_mesh_ids)) return 1;
    if (upload(ctx, (const float4*) v->instance_transforms, (size_t) v->num_instances * 2, &sc.instance_transforms)) return 1;
  }
  if (dirty & LUMC_DIRTY_MATERIALS) {
    free_group(ctx, LumContext::kGrpMat); ctx->alloc_group = LumContext::kGrpMat;
    if (upload(ctx, (const uint4*) v->materials, (size_t) v->num_materials * 2, &sc.materials)) return 1;
  }
  if (dirty_lights) {
    free_group(ctx, LumContext::kGrpLight); ctx->alloc_group = LumContext::kGrpLight;
    sc.light_tree_root = nullptr; sc.light_root_children = nullptr; sc.light_tree_nodes = nullptr; sc.light_tri_handles = nullptr; sc.light_tri_table = nullptr;
    sc.light_nodes = nullptr; sc.light_tris = nullptr; sc.light_num_nodes = 0;
  }
  if (dirty_lights && v->light_tree_root && v->num_lights) {
    const uint32_t sections = v->light_tree_root[10];
    if (upload(ctx, (const uint4*) v->light_tree_root, (size_t) 1 + 3 * sections, &sc.light_tree_root)) return 1;
    {
      // The root's children as floats (dev_light.h tree_prepass): mean = byte * 2^e + base per axis, sigma = byte * 2^e_sigma, power = the 16-bit
      // integer - the operations the kernels used to perform per vertex (cuda/light_tree.cuh:133-161, :203-205), every one exact or a single
      // binary32 rounding, so the table holds the same bits (this translation unit is compiled without contraction).
      const uint32_t* h = (const uint32_t*) v->light_tree_root;
      auto bf = [](uint32_t v16) { const uint32_t b = (v16 & 0xFFFFu) << 16; float f; std::memcpy(&f, &b, 4); return f; };
      const float base[3] = {bf(h[0]), bf(h[0] >> 16), bf(h[1])};
      const float ex[3] = {std::ldexp(1.0f, (int8_t) (h[3] & 0xFF)), std::ldexp(1.0f, (int8_t) ((h[3] >> 8) & 0xFF)), std::ldexp(1.0f, (int8_t) ((h[3] >> 16) & 0xFF))};
      const float ev = std::ldexp(1.0f, (int8_t) (h[3] >> 24));
      std::vector<float> table((size_t) sections * 8 * 8 + 16, 0.0f);  // + one pair of zeros: the pass reads two children per step
      for (uint32_t s = 0; s < sections; s++) {
        const uint8_t* sec = (const uint8_t*) (h + 4 + 12 * s);  // 8 x rel mean x, y, z, rel std dev, then 8 x u16 power
        for (uint32_t c = 0; c < 8; c++) {
          float* e = &table[((size_t) s * 8 + c) * 8];
          for (int a = 0; a < 3; a++) { const float q = (float) sec[8 * a + c]; const float scaled = q * ex[a]; e[a] = scaled + base[a]; }
          e[3] = (float) sec[24 + c] * ev;
          uint16_t pw; std::memcpy(&pw, sec + 32 + 2 * c, 2);
          e[4] = (float) pw;
        }
      }
      if (upload(ctx, table.data(), table.size(), &sc.light_root_children)) return 1;
    }
    if (upload(ctx, (const uint4*) v->light_tree_nodes, (size_t) v->num_light_tree_nodes * 4, &sc.light_tree_nodes)) return 1;
    if (upload(ctx, (const uint2*) v->light_tri_handles, v->num_lights, &sc.light_tri_handles)) return 1;
  }
  if (!sc.bluenoise_2d) { ctx->alloc_group = LumContext::kGrpOnce; if (upload(ctx, v->bluenoise_2d, 65536, &sc.bluenoise_2d)) return 1; }
  if (dirty & LUMC_DIRTY_TEXTURES) {
    free_group(ctx, LumContext::kGrpTex); ctx->alloc_group = LumContext::kGrpTex;
    sc.num_textures = 0; sc.texture_table = nullptr; sc.texels = nullptr;
  }
  if ((dirty & LUMC_DIRTY_TEXTURES) && v->num_textures && v->texture_table && v->texels) {
    size_t texel_count = 0;
    for (uint32_t t = 0; t < v->num_textures; t++)
      texel_count = std::max(texel_count, (size_t) v->texture_table[4 * t] + (size_t) v->texture_table[4 * t + 1] * v->texture_table[4 * t + 2]);
    if (upload(ctx, (const uint4*) v->texture_table, v->num_textures, &sc.texture_table)) return 1;
    if (upload(ctx, v->texels, texel_count, &sc.texels)) return 1;
    sc.num_textures = v->num_textures;
  }

  // ---- top-level BVH over the instances' world boxes + one bottom-level BVH per mesh, in ONE node array with absolute indices ----
  // Depth caps keep the traversal stack bounded (dev_trace.h kStackSize): top level <= 16, bottom levels <= 26 BVH4 levels.
  if (dirty_instances) {
  std::vector<Aabb>& mesh_box = ctx->mesh_box;
  std::vector<std::vector<Aabb>> tri_boxes(v->num_meshes);
  if (dirty_meshes) mesh_box.assign(v->num_meshes, Aabb{});
  if (mesh_box.size() != v->num_meshes || (!dirty_meshes && ctx->mesh_bvh.size() != v->num_meshes)) { ctx->error = "lumc_scene_update: the meshes changed but LUMC_DIRTY_MESHES is not set"; return 1; }
  for (uint32_t m = 0; dirty_meshes && m < v->num_meshes; m++) {
    const uint32_t t0 = v->mesh_tri_offset[m], nt = v->mesh_tri_offset[m + 1] - t0;
    tri_boxes[m].resize(nt);
    Aabb mb{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
    std::mutex mb_mutex;
    host_parallel_for(nt, [&](size_t b, size_t e) {
      Aabb part{{FLT_MAX, FLT_MAX, FLT_MAX}, {-FLT_MAX, -FLT_MAX, -FLT_MAX}};
      for (size_t t = b; t < e; t++) {
        const float* p = v->vertices + ((size_t) t0 + t) * 12;
        tri_boxes[m][t] = tri_box(p, p + 4, p + 8);
        for (int k = 0; k < 3; k++) { part.lo[k] = std::min(part.lo[k], tri_boxes[m][t].lo[k]); part.hi[k] = std::max(part.hi[k], tri_boxes[m][t].hi[k]); }
      }
      std::lock_guard<std::mutex> lock(mb_mutex);
      for (int k = 0; k < 3; k++) { mb.lo[k] = std::min(mb.lo[k], part.lo[k]); mb.hi[k] = std::max(mb.hi[k], part.hi[k]); }
    });
    mesh_box[m] = mb;
  }
  std::vector<float4> inv_rows(3 * (size_t) v->num_instances + 3);
  for (uint32_t i = 0; i < v->num_instances; i++) instance_inverse_rows(v->instance_transforms + (size_t) i * 8, &inv_rows[3 * (size_t) i]);
  ctx->alloc_group = LumContext::kGrpInst;  // NOT the group of whatever was uploaded before: a TEXTURES / MATERIALS / LIGHTS-only update frees those groups
  if (upload(ctx, inv_rows.data(), inv_rows.size(), &sc.instance_rows)) return 1;  // by instance id: the exact flavour's ambient reuse re-tests a ray against a hit's triangle (k_resolve_reuse)

  std::vector<Bvh4Node> nodes;
  std::vector<uint32_t> tlas_order;  // instance id of every top-level leaf
  {
    std::vector<Aabb> boxes;
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < v->num_instances; i++) {
      const uint32_t m = v->instance_mesh_ids[i];
      if (m >= v->num_meshes || v->mesh_tri_offset[m + 1] == v->mesh_tri_offset[m]) continue;
      Aabb wb;
      if (!instance_world_box(&inv_rows[3 * (size_t) i], mesh_box[m], wb)) continue;  // degenerate transform: nothing to hit
      boxes.push_back(wb);
      ids.push_back(i);
    }
    for (int k = 0; k < 3; k++) { ctx->world_lo[k] = FLT_MAX; ctx->world_hi[k] = -FLT_MAX; }
    for (const Aabb& b : boxes)
      for (int k = 0; k < 3; k++) { ctx->world_lo[k] = std::min(ctx->world_lo[k], b.lo[k]); ctx->world_hi[k] = std::max(ctx->world_hi[k], b.hi[k]); }
    if (boxes.empty()) for (int k = 0; k < 3; k++) { ctx->world_lo[k] = 0.0f; ctx->world_hi[k] = 1.0f; }
    Bvh4 tlas = build_bvh4(boxes.data(), (uint32_t) boxes.size(), 1, 16);  // one instance per top-level leaf (dev_trace.h)
    if (tlas.nodes.empty()) { ctx->error = "top-level BVH exceeds 16 levels"; return 1; }
    tlas_order.resize(tlas.prims.size());
    for (size_t i = 0; i < tlas_order.size(); i++) tlas_order[i] = ids[tlas.prims[i]];
    nodes = tlas.nodes;  // root at index 0, child indices already absolute
    sc.tlas_num_nodes = (uint32_t) tlas.nodes.size();
    ctx->bvh_stats[2] = tlas.nodes.size();
  }
  if (dirty_meshes) {
    ctx->bvh_build_seconds = 0.0;
    ctx->bvh_meshes_by_builder[0] = ctx->bvh_meshes_by_builder[1] = 0;
    ctx->mesh_bvh.assign(v->num_meshes, nullptr);
  }
  std::vector<BvhTri> blas_tris(dirty_meshes ? (size_t) total_tris + 1 : 0);
  if (dirty_meshes) std::memset(blas_tris.data(), 0, sizeof(BvhTri) * blas_tris.size());
  std::vector<uint32_t> mesh_root(v->num_meshes + 1, 0);
  for (uint32_t m = 0; m < v->num_meshes; m++) {
    const uint32_t t0 = v->mesh_tri_offset[m], nt = v->mesh_tri_offset[m + 1] - t0;
    if (dirty_meshes) {  // the only part of an upload that takes long: an instance edit reuses the trees, another device of the host the first one's
      const auto t_build = std::chrono::steady_clock::now();
      const MeshTreeKey key = mesh_tree_key(v->vertices + (size_t) t0 * 12, nt, ctx->bvh_builder);
      std::shared_ptr<const MeshTree> tree = find_mesh_tree(key);
      if (!tree) {
        auto built = std::make_shared<MeshTree>();
        if (ctx->bvh_builder == 1) built->bvh = build_bvh4_lbvh(tri_boxes[m].data(), nt, kBvhLeafMaxTri, 26);
        else if (ctx->bvh_builder == 2) built->bvh = build_bvh4_ploc(tri_boxes[m].data(), nt, kBvhLeafMaxTri, 26);
        else if (ctx->bvh_builder == 3) built->bvh = build_bvh4_sah_gpu(tri_boxes[m].data(), nt, kBvhLeafMaxTri, 26);
        built->built_on_gpu = !built->bvh.nodes.empty();
        if (!built->built_on_gpu) built->bvh = build_bvh4(tri_boxes[m].data(), nt, kBvhLeafMaxTri, 26);  // the host builder: asked for, or the fallback for a mesh the GPU builders cannot take
        if (built->bvh.nodes.empty()) { ctx->error = "mesh BVH exceeds 26 levels"; return 1; }
        tree = built;
        keep_mesh_tree(key, tree);
      }
      ctx->bvh_meshes_by_builder[tree->built_on_gpu ? 1 : 0]++;
      ctx->bvh_build_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build).count();
      ctx->mesh_bvh[m] = std::move(tree);
    }
    const Bvh4& bvh = ctx->mesh_bvh[m]->bvh;
    const uint32_t base = (uint32_t) nodes.size();
    mesh_root[m] = base;
    for (Bvh4Node n : bvh.nodes) {
      for (int k = 0; k < 4; k++) {
        if (n.child[k] == kBvhEmpty) continue;
        if (n.child[k] & kBvhLeafBit) n.child[k] += t0;  // leaf ranges index blas_tris directly (28 bits)
        else n.child[k] += base;
      }
      nodes.push_back(n);
    }
    if (dirty_meshes) host_parallel_for(nt, [&](size_t b, size_t e) {
      for (size_t i = b; i < e; i++) {
        const uint32_t t = bvh.prims[i];
        const float* p = v->vertices + (size_t) (t0 + t) * 12;
        BvhTri& bt = blas_tris[(size_t) t0 + i];
        for (int k = 0; k < 3; k++) { bt.p0[k] = p[k]; bt.e1[k] = p[4 + k] - p[k]; bt.e2[k] = p[8 + k] - p[k]; }
        bt.id = t; bt.scene_index = t0 + t;
        bt.albedo_tex = kBvhTriNoTexture;  // k_tri_opacity writes the word from the triangle's material (below; again after a material edit)
      }
    });
    if (dirty_meshes) { tri_boxes[m].clear(); tri_boxes[m].shrink_to_fit(); }
  }
  // ---- renumber: the top of the tree first, in breadth-first order across both levels (top-level leaves continue into the root of
  // their mesh), so that "node index < K" selects the K most visited nodes; the ray kernels stage those in LDS ----
  std::vector<uint8_t> node_is_top;  // after the renumbering: the node belongs to the top level (LUM_BVH8O)
  {
    const size_t n = nodes.size();
    std::vector<uint32_t> order;
    std::vector<uint8_t> seen(n, 0);
    order.reserve(n);
    const size_t top_budget = std::min<size_t>(n, 4096);
    auto next_of = [&](uint32_t id, int k, uint32_t& next) {
      const uint32_t c = nodes[id].child[k];
      if (c == kBvhEmpty) return false;
      if (c & kBvhLeafBit) {
        if (id >= sc.tlas_num_nodes) return false;
        next = mesh_root[v->instance_mesh_ids[tlas_order[c & 0x0FFFFFFFu]]];
      }
      else next = c;
      return true;
    };
    if (ctx->top_order_by_area) {
      // best first: a ray that enters a node's box enters a child's with probability area(child) / area(box) (convex boxes, uniformly
      // distributed lines), so the product of those ratios down from the root estimates how often a node is visited; a child never
      // outranks its parent, the order stays top-down
      auto half_area = [](float dx, float dy, float dz) { return (double) dx * dy + (double) dy * dz + (double) dz * dx; };
      auto child_shares = [&](const Bvh4Node& node, double share[4]) {
        float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int k = 0; k < 4; k++) {
          if (node.child[k] == kBvhEmpty) continue;
          lo[0] = std::min(lo[0], node.lo_x[k]); lo[1] = std::min(lo[1], node.lo_y[k]); lo[2] = std::min(lo[2], node.lo_z[k]);
          hi[0] = std::max(hi[0], node.hi_x[k]); hi[1] = std::max(hi[1], node.hi_y[k]); hi[2] = std::max(hi[2], node.hi_z[k]);
        }
        const double whole = half_area(hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
        for (int k = 0; k < 4; k++)
          share[k] = (node.child[k] == kBvhEmpty) ? 0.0
                   : (whole > 0.0) ? std::min(half_area(node.hi_x[k] - node.lo_x[k], node.hi_y[k] - node.lo_y[k], node.hi_z[k] - node.lo_z[k]) / whole, 1.0) : 1.0;
      };
      // a mesh is entered through every instance of it: its root's estimate is the sum over the top-level leaves that lead to it
      std::vector<double> root_estimate(n, 0.0);
      {
        std::vector<std::pair<uint32_t, double>> walk{{0u, 1.0}};
        for (size_t head = 0; head < walk.size() && sc.tlas_num_nodes > 0; head++) {
          const uint32_t id = walk[head].first;
          double share[4];
          child_shares(nodes[id], share);
          for (int k = 0; k < 4; k++) {
            const uint32_t c = nodes[id].child[k];
            if (c == kBvhEmpty) continue;
            uint32_t next;
            next_of(id, k, next);
            if (c & kBvhLeafBit) root_estimate[next] += walk[head].second * share[k];
            else walk.push_back({next, walk[head].second * share[k]});
          }
        }
      }
      std::priority_queue<std::pair<double, uint32_t>> open;
      std::vector<uint8_t> queued(n, 0);
      open.push({DBL_MAX, 0u}); queued[0] = 1;
      while (!open.empty() && order.size() < top_budget) {
        const double p = (open.top().first == DBL_MAX) ? 1.0 : open.top().first;
        const uint32_t id = open.top().second;
        open.pop();
        order.push_back(id); seen[id] = 1;
        double share[4];
        child_shares(nodes[id], share);
        for (int k = 0; k < 4; k++) {
          uint32_t next;
          if (!next_of(id, k, next) || queued[next]) continue;
          queued[next] = 1;
          const bool enters_mesh = (nodes[id].child[k] & kBvhLeafBit) != 0;
          open.push({enters_mesh ? root_estimate[next] : p * share[k], next});
        }
      }
    }
    else {
      order.push_back(0); seen[0] = 1;
      for (size_t head = 0; head < order.size() && order.size() < top_budget; head++) {
        const uint32_t id = order[head];
        for (int k = 0; k < 4; k++) {
          uint32_t next;
          if (next_of(id, k, next) && !seen[next]) { seen[next] = 1; order.push_back(next); }
        }
      }
    }
    for (uint32_t i = 0; i < n; i++) if (!seen[i]) order.push_back(i);
    std::vector<uint32_t> new_index(n);
    for (uint32_t i = 0; i < n; i++) new_index[order[i]] = i;
    std::vector<Bvh4Node> renum(n);
    for (uint32_t i = 0; i < n; i++) {
      Bvh4Node node = nodes[order[i]];
      for (int k = 0; k < 4; k++)
        if (node.child[k] != kBvhEmpty && !(node.child[k] & kBvhLeafBit)) node.child[k] = new_index[node.child[k]];
      renum[i] = node;
    }
    nodes.swap(renum);
    for (uint32_t m = 0; m < v->num_meshes; m++) mesh_root[m] = new_index[mesh_root[m]];
    node_is_top.resize(n);
    for (uint32_t i = 0; i < n; i++) node_is_top[i] = order[i] < sc.tlas_num_nodes ? 1 : 0;
  }
  if (total_tris >= (1u << 28) || nodes.size() >= (1u << 25)) { ctx->error = "scene too large for 28-bit leaf ranges / 32-bit node offsets"; return 1; }
#if LUM_BVH8
  {
    std::vector<uint32_t> keep;
    keep.push_back(0u);
    for (uint32_t m = 0; m < v->num_meshes; m++) if (v->mesh_tri_offset[m + 1] > v->mesh_tri_offset[m]) keep.push_back(mesh_root[m]);
    // the two levels are walked one after the other: their depths add up on the stack
    std::vector<Bvh8Node> wide;
    std::vector<uint32_t> map;
    std::vector<uint32_t> levels;
    widen_to_bvh8(nodes, keep, wide, map, &levels);
    uint32_t deepest_mesh = 0;
    for (size_t r = 1; r < levels.size(); r++) deepest_mesh = std::max(deepest_mesh, levels[r]);
    if (7u * (levels[0] + deepest_mesh) + 4u > (uint32_t) kStackSize) { ctx->error = "BVH too deep for the traversal stack (8-wide nodes)"; return 1; }
    uint32_t new_tlas_nodes = 0;
    for (uint32_t i = 0; i < sc.tlas_num_nodes; i++) if (map[i] != 0xFFFFFFFFu) new_tlas_nodes++;
    sc.tlas_num_nodes = new_tlas_nodes;
    for (uint32_t m = 0; m < v->num_meshes; m++) if (v->mesh_tri_offset[m + 1] > v->mesh_tri_offset[m]) mesh_root[m] = map[mesh_root[m]];
    nodes.resize(wide.size());
    std::memcpy(nodes.data(), wide.data(), wide.size() * sizeof(Bvh8Node));
    ctx->bvh_stats[2] = new_tlas_nodes;
  }
#endif
#if LUM_BVH8O
  {
    std::vector<uint32_t> keep;
    keep.push_back(0u);
    for (uint32_t m = 0; m < v->num_meshes; m++) if (v->mesh_tri_offset[m + 1] > v->mesh_tri_offset[m]) keep.push_back(mesh_root[m]);
    Bvh8oResult r8;
    if (!build_bvh8o(nodes, node_is_top, keep, total_tris, tlas_order.size(), r8)) { ctx->error = "8-wide conversion failed (leaf offsets / unreferenced leaves)"; return 1; }
    uint32_t deepest_mesh = 0;
    for (size_t r = 1; r < r8.levels.size(); r++) deepest_mesh = std::max(deepest_mesh, r8.levels[r]);
    if (r8.levels[0] + deepest_mesh + 8u > (uint32_t) kStackSize) { ctx->error = "BVH too deep for the traversal stack (8-wide nodes)"; return 1; }  // one group entry per level + an instance's two
    {  // the triangles and the top-level leaf records in the order the nodes refer to them
      std::vector<BvhTri> reordered(blas_tris.size());
      std::memset(reordered.data(), 0, sizeof(BvhTri) * reordered.size());
      for (size_t i = 0; i < r8.tri_order.size(); i++) reordered[i] = blas_tris[r8.tri_order[i]];
      blas_tris.swap(reordered);
      std::vector<uint32_t> tl(tlas_order.size());
      for (size_t i = 0; i < tl.size(); i++) tl[i] = tlas_order[r8.top_leaf_order[i]];
      tlas_order.swap(tl);
    }
    for (uint32_t m = 0; m < v->num_meshes; m++) if (v->mesh_tri_offset[m + 1] > v->mesh_tri_offset[m]) mesh_root[m] = r8.old_to_new[mesh_root[m]];
    sc.tlas_num_nodes = r8.top_nodes;
    ctx->bvh_stats[2] = r8.top_nodes;
    static_assert(sizeof(Bvh8oNode) == sizeof(Bvh4Node), "same 128-byte slot");
    nodes.resize(r8.nodes.size());
    std::memcpy(nodes.data(), r8.nodes.data(), r8.nodes.size() * sizeof(Bvh8oNode));
  }
#endif
#if LUM_BVH4Q
  quantise_bvh4(nodes);
#endif
  ctx->alloc_group = LumContext::kGrpInst;
  if (upload(ctx, nodes.data(), nodes.size(), &sc.bvh_nodes)) return 1;
  if (dirty_meshes) { ctx->alloc_group = LumContext::kGrpMesh; if (upload(ctx, blas_tris.data(), blas_tris.size(), &sc.blas_tris)) return 1; ctx->alloc_group = LumContext::kGrpInst; }
  {
    std::vector<float4> leaves(4 * tlas_order.size() + 4);
    for (size_t i = 0; i < tlas_order.size(); i++) {
      const uint32_t inst = tlas_order[i];
      for (int k = 0; k < 3; k++) leaves[4 * i + k] = inv_rows[3 * (size_t) inst + k];
      const uint32_t words[4] = {inst, mesh_root[v->instance_mesh_ids[inst]], 0u, 0u};
      std::memcpy(&leaves[4 * i + 3], words, 16);
    }
    if (upload(ctx, leaves.data(), leaves.size(), &sc.tlas_leaves)) return 1;
    sc.tlas_num_leaves = (uint32_t) (leaves.size() / 4);  // records that exist (one of padding included): what a workgroup may stage in LDS
  }
  ctx->bvh_stats[0] = nodes.size() - sc.tlas_num_nodes;
  {
    // resident workgroups per CU share the LDS: what the device offers minus a margin, 128 B per node
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    size_t lds_bytes = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : prop.sharedMemPerBlock;
    // the ray kernels are compiled for 128 VGPRs: 4 waves per SIMD = 16 waves per CU = one workgroup of kTraceBlock = 1024 threads (both flavours since round 4)
#ifndef LUM_TRACE_BLOCKS_PER_CU
#define LUM_TRACE_BLOCKS_PER_CU 1  // experiment: more, smaller workgroups per CU (each with its own, smaller LDS copy of the tree top)
#endif
    const int blocks_per_cu = LUM_TRACE_BLOCKS_PER_CU;  // one workgroup of kTraceBlock threads per CU
    lds_bytes = std::min<size_t>(lds_bytes, 160 * 1024) / blocks_per_cu;
    lds_bytes = lds_bytes > 16384 ? lds_bytes - 8192 : 0;  // margin: the ray kernels' static LDS (the prefetch experiment's sink) and the runtime's own
    lds_bytes = lds_bytes > LUM_LDS_STACK_BYTES ? lds_bytes - LUM_LDS_STACK_BYTES : 0;  // the stacks' share (dev_trace.h, TraversalStack)
    ctx->lds_nodes = (uint32_t) std::min<size_t>(lds_bytes / kNodeBytes, nodes.size());
    if (const char* e = getenv("LUM_LDS_NODES")) ctx->lds_nodes = std::min<uint32_t>((uint32_t) atoi(e), ctx->lds_nodes);
    ctx->trace_blocks = (uint32_t) prop.multiProcessorCount * blocks_per_cu;
#if LUM_PHASE_QUEUES
    // dev_trace_pool.h: per persistent workgroup and pool slot 4 x 16 bytes of query state and the stack entries beyond the LDS ones (once per context)
    if (!sc.pool_state) {
      const size_t slots = (size_t) ctx->trace_blocks * LUM_POOL_SLOTS * (LUM_TRACE_BLOCK / 64u);
      void* a = nullptr; void* b = nullptr;
      HIP_TRY(ctx, hipMalloc(&a, slots * 4u * sizeof(uint4)));
      ctx->scene_allocs[LumContext::kGrpOnce].push_back(a);
      HIP_TRY(ctx, hipMalloc(&b, slots * (size_t) kStackSize * sizeof(unsigned long long)));
      ctx->scene_allocs[LumContext::kGrpOnce].push_back(b);
      sc.pool_state = (uint4*) a; sc.pool_stack = (unsigned long long*) b;
    }
#endif
    // The attribute is a property of the kernel, not of a context: it is set to what the largest scene may ask for (the whole budget computed
    // above), never to this scene's need - a second context with a small scene must not lower the cap a first one launches with.
    const size_t dyn = lds_bytes + LUM_LDS_STACK_BYTES;
    HIP_TRY(ctx, (hipError_t) wavefront_kernels_exact()->set_ray_kernel_lds(dyn));
    HIP_TRY(ctx, (hipError_t) wavefront_kernels_fast()->set_ray_kernel_lds(dyn));
  }
  }  // dirty_instances
  // ---- light-only BVH (world-space triangles; reference: optix_bvh.c:382-478) ----
  if (dirty_lights) {
    ctx->alloc_group = LumContext::kGrpLight;
    const uint32_t nl = (v->light_tree_root && v->light_bvh_tris) ? v->num_lights : 0;
    std::vector<Aabb> boxes(nl);
    for (uint32_t l = 0; l < nl; l++) { const float* p = v->light_bvh_tris + (size_t) l * 12; boxes[l] = tri_box(p, p + 4, p + 8); }
    Bvh4 lb = build_bvh4(boxes.data(), nl, kBvhLeafMaxTri, 40);
    if (lb.nodes.empty()) { ctx->error = "light BVH exceeds 40 levels"; return 1; }
    std::vector<BvhTri> tris(nl ? nl : 1);
    std::memset(tris.data(), 0, sizeof(BvhTri) * tris.size());
    for (uint32_t i = 0; i < nl; i++) {
      const uint32_t l = lb.prims[i];
      const float* p = v->light_bvh_tris + (size_t) l * 12;
      for (int k = 0; k < 3; k++) { tris[i].p0[k] = p[k]; tris[i].e1[k] = p[4 + k] - p[k]; tris[i].e2[k] = p[8 + k] - p[k]; }
      tris[i].id = l;
    }
    if (upload(ctx, lb.nodes.data(), lb.nodes.size(), &sc.light_nodes)) return 1;
    if (upload(ctx, tris.data(), tris.size(), &sc.light_tris)) return 1;
    sc.light_num_nodes = (uint32_t) lb.nodes.size();
    ctx->bvh_stats[3] = lb.nodes.size();
  }
  ctx->bvh_stats[1] = total_tris;

  { const uint32_t nt = sc.num_textures; sc.num_meshes = v->num_meshes; sc.num_instances = v->num_instances; sc.num_materials = v->num_materials; sc.num_lights = v->num_lights; sc.num_textures = nt; }
  if (total_tris && (dirty_meshes || (dirty & LUMC_DIRTY_MATERIALS))) {  // the triangles' material words: texture id, or whether they stop a visibility ray on their own
    hipLaunchKernelGGL(k_tri_opacity, dim3((total_tris + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, sc, const_cast<BvhTri*>(sc.blas_tris), total_tris);
    HIP_TRY(ctx, hipGetLastError());
  }
  if ((dirty_lights || ((dirty & (LUMC_DIRTY_MATERIALS | LUMC_DIRTY_INSTANCES | LUMC_DIRTY_MESHES)) && sc.light_tri_table)) && sc.light_tree_root && sc.num_lights) {  // the emissive triangles in world space with what their material says, one record per light (load_tri_light_table)
    float4* table = const_cast<float4*>(sc.light_tri_table);  // a material edit alone refills the table in place (same lights)
    if (dirty_lights) { HIP_TRY(ctx, hipMalloc((void**) &table, sizeof(float4) * 4 * (size_t) sc.num_lights)); ctx->scene_allocs[LumContext::kGrpLight].push_back(table); }
    hipLaunchKernelGGL(k_light_table, dim3((sc.num_lights + kBlock - 1) / kBlock), dim3(kBlock), 0, 0, sc, table);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipDeviceSynchronize());
    sc.light_tri_table = table;
  }
  if (dirty & (LUMC_DIRTY_CONSTANTS | LUMC_DIRTY_PARTICLES)) {
  if (dirty & LUMC_DIRTY_CONSTANTS) { free_group(ctx, LumContext::kGrpConst); }
  ctx->alloc_group = LumContext::kGrpConst;
  sc.width = v->width; sc.height = v->height; sc.max_ray_depth = v->max_ray_depth; sc.shading_mode = v->shading_mode;
  std::memcpy(sc.cam_pos, v->cam_pos, sizeof(sc.cam_pos));
  std::memcpy(sc.cam_rotation, v->cam_rotation, sizeof(sc.cam_rotation));
  sc.cam_fov = v->cam_fov; sc.cam_aperture_size = v->cam_aperture_size; sc.cam_object_distance = v->cam_object_distance;
  sc.cam_scale = v->cam_scale; sc.cam_rr_threshold = v->cam_rr_threshold;
  sc.cam_aperture_shape = v->cam_aperture_shape; sc.cam_aperture_blade_count = v->cam_aperture_blade_count;
  sc.sky_mode = v->sky_mode;
  std::memcpy(sc.sky_constant_color, v->sky_constant_color, sizeof(sc.sky_constant_color));
  sc.sky_steps = v->sky_steps; sc.sky_ozone_absorption = v->sky_ozone_absorption;
  std::memcpy(sc.sky_geometry_offset, v->sky_geometry_offset, sizeof(sc.sky_geometry_offset));
  sc.sky_sun_strength = v->sky_sun_strength; sc.sky_base_density = v->sky_base_density; sc.sky_rayleigh_density = v->sky_rayleigh_density;
  sc.sky_mie_density = v->sky_mie_density; sc.sky_ozone_density = v->sky_ozone_density; sc.sky_rayleigh_falloff = v->sky_rayleigh_falloff;
  sc.sky_mie_falloff = v->sky_mie_falloff; sc.sky_ground_visibility = v->sky_ground_visibility; sc.sky_ozone_layer_thickness = v->sky_ozone_layer_thickness;
  sc.sky_multiscattering_factor = v->sky_multiscattering_factor;
  std::memcpy(sc.sky_sun_pos, v->sky_sun_pos, sizeof(sc.sky_sun_pos));
  std::memcpy(sc.sky_mie_phase, v->sky_mie_phase, sizeof(sc.sky_mie_phase));
  std::memcpy(sc.sky_moon_pos, v->sky_moon_pos, sizeof(sc.sky_moon_pos));
  sc.sky_moon_tex_offset = v->sky_moon_tex_offset; sc.sky_stars_intensity = v->sky_stars_intensity;
  sc.sky_moon_albedo_tex = v->sky_moon_albedo_tex; sc.sky_moon_normal_tex = v->sky_moon_normal_tex;
  sc.sky_stars_count = 0; sc.sky_stars = nullptr; sc.sky_stars_offsets = nullptr;
  if (v->sky_stars && v->sky_stars_offsets && v->sky_stars_count) {
    if (upload(ctx, (const float4*) v->sky_stars, (size_t) v->sky_stars_count, &sc.sky_stars)) return 1;
    if (upload(ctx, v->sky_stars_offsets, (size_t) 64 * 32 + 1, &sc.sky_stars_offsets)) return 1;
    sc.sky_stars_count = v->sky_stars_count;
  }
  // ---- sky look-up tables: taken from the caller or generated here (device/device_sky.c:64-200), only for the procedural sky ----
  sc.sky_lut_transmittance = nullptr; sc.sky_lut_multiscattering = nullptr;
  sc.sky_hdri = nullptr; sc.sky_hdri_dim = 0;
  sc.sky_aerial_perspective = v->sky_aerial_perspective;
  // ---- fog ----
  sc.fog_active = v->fog_active ? 1u : 0u;
  sc.fog_density = v->fog_density; sc.fog_dist = v->fog_dist; sc.fog_height = v->fog_height;
  std::memcpy(sc.fog_phase, v->fog_phase, sizeof(sc.fog_phase));
  sc.bridge_max_num_vertices = v->bridge_max_num_vertices;
  if (sc.fog_active && !(sc.fog_density > 0.0f)) { ctx->error = "lumc_scene_upload: fog needs a positive density"; return 1; }
  // ---- ocean ----
  sc.ocean_active = v->ocean_active ? 1u : 0u;
  sc.ocean_height = v->ocean_height; sc.ocean_amplitude = v->ocean_amplitude; sc.ocean_frequency = v->ocean_frequency;
  sc.ocean_refractive_index = v->ocean_refractive_index;
  std::memcpy(sc.ocean_scattering, v->ocean_scattering, sizeof(sc.ocean_scattering));
  std::memcpy(sc.ocean_absorption, v->ocean_absorption, sizeof(sc.ocean_absorption));
  sc.ocean_molecular_weight = v->ocean_molecular_weight;
  sc.ocean_caustics_active = v->ocean_caustics_active ? 1u : 0u;
  sc.ocean_caustics_ris_sample_count = v->ocean_caustics_ris_sample_count;
  sc.ocean_caustics_domain_scale = v->ocean_caustics_domain_scale;
  sc.ocean_multiscattering = v->ocean_multiscattering ? 1u : 0u;
  sc.ocean_triangle_light_contribution = v->ocean_triangle_light_contribution ? 1u : 0u;
  if (sc.ocean_active) {
    if (!(sc.ocean_refractive_index >= 1.0f)) { ctx->error = "lumc_scene_upload: the ocean needs a refractive index of at least 1"; return 1; }
  }
  // ---- clouds ----
  sc.cloud_active = v->cloud_active ? 1u : 0u;
  sc.cloud_atmosphere_scattering = v->cloud_atmosphere_scattering ? 1u : 0u;
  sc.cloud_steps = v->cloud_steps & 0x3FFu; sc.cloud_shadow_steps = v->cloud_shadow_steps & 0x3FFu; sc.cloud_octaves = v->cloud_octaves & 0xFu;  // DeviceCloud's bit fields
  sc.cloud_offset_x = v->cloud_offset_x; sc.cloud_offset_z = v->cloud_offset_z; sc.cloud_density = v->cloud_density;
  sc.cloud_noise_shape_scale = v->cloud_noise_shape_scale; sc.cloud_noise_detail_scale = v->cloud_noise_detail_scale; sc.cloud_noise_weather_scale = v->cloud_noise_weather_scale;
  std::memcpy(sc.cloud_phase, v->cloud_phase, sizeof(sc.cloud_phase));
  std::memcpy(sc.cloud_layers, v->cloud_layers, sizeof(sc.cloud_layers));
  sc.cloud_noise_shape = nullptr; sc.cloud_noise_detail = nullptr; sc.cloud_noise_weather = nullptr;
  if (sc.cloud_active) {
    if (sc.cloud_steps == 0 || sc.cloud_shadow_steps == 0) { ctx->error = "lumc_scene_upload: clouds need positive step counts"; return 1; }
    if (v->cloud_noise_shape && v->cloud_noise_detail && v->cloud_noise_weather) {
      if (upload(ctx, (const uint32_t*) v->cloud_noise_shape, (size_t) kCloudShapeRes * kCloudShapeRes * kCloudShapeRes, &sc.cloud_noise_shape)) return 1;
      if (upload(ctx, (const uint32_t*) v->cloud_noise_detail, (size_t) kCloudDetailRes * kCloudDetailRes * kCloudDetailRes, &sc.cloud_noise_detail)) return 1;
      if (upload(ctx, (const uint32_t*) v->cloud_noise_weather, (size_t) kCloudWeatherRes * kCloudWeatherRes, &sc.cloud_noise_weather)) return 1;
    }
    else {
      if (ensure_cloud_noise(ctx, v->cloud_seed)) return 1;
      sc.cloud_noise_shape = ctx->d_cloud_noise[0]; sc.cloud_noise_detail = ctx->d_cloud_noise[1]; sc.cloud_noise_weather = ctx->d_cloud_noise[2];
    }
  }
  // ---- particles ----
  sc.particles_active = (v->particles_active && v->particles_count) ? 1u : 0u;
  sc.particles_count = sc.particles_active ? v->particles_count : 0u;
  sc.particles_scale = v->particles_scale; sc.particles_speed = v->particles_speed;
  std::memcpy(sc.particles_albedo, v->particles_albedo, sizeof(sc.particles_albedo));
  std::memcpy(sc.particles_direction, v->particles_direction, sizeof(sc.particles_direction));
  std::memcpy(sc.particles_phase, v->particles_phase, sizeof(sc.particles_phase));
  if (dirty & LUMC_DIRTY_PARTICLES) {
    free_group(ctx, LumContext::kGrpPart); ctx->alloc_group = LumContext::kGrpPart;
    if (build_particle_tree(ctx, v, sc)) return 1;
    ctx->alloc_group = LumContext::kGrpConst;
  }
  if (sc.sky_mode != kSkyConstantColor) {  // HDRI mode bakes from them and samples the sun through them
    const size_t tm_texels = 2 * (size_t) kSkyTmWidth * kSkyTmHeight, ms_texels = 2 * (size_t) kSkyMsSize * kSkyMsSize;
    if (v->sky_lut_transmittance && v->sky_lut_multiscattering) {
      if (upload(ctx, (const float4*) v->sky_lut_transmittance, tm_texels, &sc.sky_lut_transmittance)) return 1;
      if (upload(ctx, (const float4*) v->sky_lut_multiscattering, ms_texels, &sc.sky_lut_multiscattering)) return 1;
      ctx->sky_lut_key.clear();
    }
    else {
      // the two tables are functions of the atmosphere's parameters alone (sky.cuh:110-176, :186-332): a camera move or a sun move keeps them
      std::vector<uint32_t> key;
      auto put = [&](const void* p, size_t bytes) { const size_t at = key.size(); key.resize(at + (bytes + 3) / 4, 0u); std::memcpy(key.data() + at, p, bytes); };
      put(&sc.sky_ozone_absorption, sizeof(sc.sky_ozone_absorption));
      const float params[] = {sc.sky_base_density, sc.sky_rayleigh_density, sc.sky_mie_density, sc.sky_ozone_density, sc.sky_rayleigh_falloff, sc.sky_mie_falloff,
                              sc.sky_ground_visibility, sc.sky_ozone_layer_thickness, sc.sky_multiscattering_factor, sc.sky_sun_strength};
      put(params, sizeof(params)); put(sc.sky_mie_phase, sizeof(sc.sky_mie_phase)); put(sc.sky_sun_pos, sizeof(sc.sky_sun_pos)); put(sc.sky_geometry_offset, sizeof(sc.sky_geometry_offset));
      if (!ctx->d_sky_lut[0]) {
        HIP_TRY(ctx, hipMalloc((void**) &ctx->d_sky_lut[0], sizeof(float4) * tm_texels));
        HIP_TRY(ctx, hipMalloc((void**) &ctx->d_sky_lut[1], sizeof(float4) * ms_texels));
        ctx->sky_lut_key.clear();
      }
      if (key != ctx->sky_lut_key) {
        hipLaunchKernelGGL(k_sky_transmittance_lut, dim3((kSkyTmWidth * kSkyTmHeight + 63) / 64), dim3(64), 0, 0, sc, ctx->d_sky_lut[0]);
        sc.sky_lut_transmittance = ctx->d_sky_lut[0];  // the multiscattering integration reads the finished transmittance table
        hipLaunchKernelGGL(k_sky_multiscattering_lut, dim3(kSkyMsSize, kSkyMsSize), dim3(kSkyMsIter), 0, 0, sc, ctx->d_sky_lut[1]);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipDeviceSynchronize());
        ctx->sky_lut_key = key;
      }
      sc.sky_lut_transmittance = ctx->d_sky_lut[0];
      sc.sky_lut_multiscattering = ctx->d_sky_lut[1];
    }
  }
  // ---- BSDF energy tables: taken from the caller or generated here (device/device_bsdf.c:64-130) ----
  const uint16_t* host_luts[4] = {v->lut_conductor, v->lut_glossy, v->lut_dielectric, v->lut_dielectric_inv};
  const uint32_t lut_count[4] = {1024, 1024, 32768, 32768};
  const bool have_luts = ctx->d_luts[0] != nullptr;
  for (int t = 0; t < 4 && !have_luts; t++) HIP_TRY(ctx, hipMalloc((void**) &ctx->d_luts[t], sizeof(uint16_t) * lut_count[t]));
  if (have_luts && dirty != LUMC_DIRTY_ALL) { /* a partial update keeps the tables the context renders with */ }
  else if (host_luts[0] && host_luts[1] && host_luts[2] && host_luts[3]) {
    for (int t = 0; t < 4; t++) HIP_TRY(ctx, hipMemcpy(ctx->d_luts[t], host_luts[t], sizeof(uint16_t) * lut_count[t], hipMemcpyHostToDevice));
  }
  else {
    // The tables are a function of the embedded blue-noise mask alone (65 536 samples per texel, one thread per texel: 0.29 s of GPU time):
    // generated once per process, every later upload copies them.
    static std::mutex lut_mutex;
    static std::vector<uint16_t> lut_cache[4];
    static std::vector<uint32_t> lut_cache_mask;
    std::lock_guard<std::mutex> lock(lut_mutex);
    const bool cached = !lut_cache[0].empty() && lut_cache_mask.size() == 65536 && std::memcmp(lut_cache_mask.data(), v->bluenoise_2d, sizeof(uint32_t) * 65536) == 0;
    if (cached) {
      for (int t = 0; t < 4; t++) HIP_TRY(ctx, hipMemcpy(ctx->d_luts[t], lut_cache[t].data(), sizeof(uint16_t) * lut_count[t], hipMemcpyHostToDevice));
    }
    else {
      // the two big tables and the conductor table are independent: side by side on three streams; the glossy table divides by the conductor's
      hipStream_t streams[3];
      for (auto& st : streams) HIP_TRY(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      const int first_wave[3] = {0, 2, 3};
      for (int k = 0; k < 3; k++) {
        const int t = first_wave[k];
        hipLaunchKernelGGL(k_generate_lut, dim3((lut_count[t] + 63) / 64), dim3(64), 0, streams[k], sc.bluenoise_2d, t, lut_count[t], ctx->d_luts[0], ctx->d_luts[t]);
      }
      hipLaunchKernelGGL(k_generate_lut, dim3((lut_count[1] + 63) / 64), dim3(64), 0, streams[0], sc.bluenoise_2d, 1, lut_count[1], ctx->d_luts[0], ctx->d_luts[1]);
      HIP_TRY(ctx, hipGetLastError());
      for (auto& st : streams) { HIP_TRY(ctx, hipStreamSynchronize(st)); (void) hipStreamDestroy(st); }
      for (int t = 0; t < 4; t++) {
        lut_cache[t].resize(lut_count[t]);
        HIP_TRY(ctx, hipMemcpy(lut_cache[t].data(), ctx->d_luts[t], sizeof(uint16_t) * lut_count[t], hipMemcpyDeviceToHost));
      }
      lut_cache_mask.assign(v->bluenoise_2d, v->bluenoise_2d + 65536);
    }
  }
  sc.lut_conductor = ctx->d_luts[0]; sc.lut_glossy = ctx->d_luts[1]; sc.lut_dielectric = ctx->d_luts[2]; sc.lut_dielectric_inv = ctx->d_luts[3];
  // ---- sky panorama (HDRI mode): the caller's, or baked here from the procedural sky as the reference's device manager does when the
  // sky changes (device_manager.c:351-366, device_sky.c:249-366); lumc_sky_hdri_build re-bakes on request ----
  if (sc.sky_mode == kSkyHdri) {
    if (v->sky_hdri && v->sky_hdri_dim) {
      if (upload(ctx, (const float4*) v->sky_hdri, (size_t) v->sky_hdri_dim * v->sky_hdri_dim, &sc.sky_hdri)) return 1;
      sc.sky_hdri_dim = v->sky_hdri_dim;
    }
    else {
      ctx->has_scene = true;  // the bake renders this scene's sky
      if (lumc_sky_hdri_build(ctx, v->sky_hdri_origin, v->sky_hdri_dim, v->sky_hdri_samples ? v->sky_hdri_samples : 1u)) { ctx->has_scene = false; return 1; }
    }
  }
  }  // constants
  // ---- bridges to emissive triangles (fog, or an ocean with triangle_light_contribution): the vertex-count table. Decided after EVERY update, not
  // only when the constants are dirty: a material that becomes emissive (MATERIALS | LIGHTS) gives a fogged scene its first light, and
  // bridges_vertex_count_importance reads the table without a check. The table lives in the context (5 KB, uploaded once per content). ----
  {
    const bool need_bridges = (sc.fog_active || (sc.ocean_active && sc.ocean_triangle_light_contribution)) && sc.num_lights > 0 && sc.light_tree_root;
    sc.bridge_lut = nullptr;
    if (need_bridges) {
      if (!v->bridge_lut) { ctx->error = sc.fog_active ? "lumc_scene_upload: fog with emissive triangles needs bridge_lut" : "lumc_scene_upload: an ocean lit by emissive triangles needs bridge_lut"; return 1; }
      if (sc.bridge_max_num_vertices == 0) { ctx->error = "lumc_scene_upload: bridge_max_num_vertices must be at least 1"; return 1; }
      const size_t n = (size_t) 64 * 21;
      if (!ctx->d_bridge_lut || ctx->bridge_lut_host.size() != n || std::memcmp(ctx->bridge_lut_host.data(), v->bridge_lut, n * sizeof(float)) != 0) {
        if (!ctx->d_bridge_lut) HIP_TRY(ctx, hipMalloc((void**) &ctx->d_bridge_lut, n * sizeof(float)));
        HIP_TRY(ctx, hipMemcpy(ctx->d_bridge_lut, v->bridge_lut, n * sizeof(float), hipMemcpyHostToDevice));
        ctx->bridge_lut_host.assign(v->bridge_lut, v->bridge_lut + n);
      }
      sc.bridge_lut = ctx->d_bridge_lut;
    }
  }
  // the moon's texture ids follow the texture pool (the host layer appends the two moon textures behind the scene's own): an added texture moves them
  sc.sky_moon_albedo_tex = v->sky_moon_albedo_tex; sc.sky_moon_normal_tex = v->sky_moon_normal_tex;
  ctx->has_scene = true;
  return 0;
}

int lumc_scene_upload(LumContext* ctx, const LumDeviceSceneView* v) {
  if (!ctx || !v) return 1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  free_scene(ctx);
  std::memset(&ctx->scene, 0, sizeof(ctx->scene));
  return scene_update(ctx, v, LUMC_DIRTY_ALL);
}

int lumc_scene_update(LumContext* ctx, const LumDeviceSceneView* v, unsigned int dirty) {
  if (!ctx || !v) return 1;
  if (!ctx->has_scene || (dirty & LUMC_DIRTY_ALL) == LUMC_DIRTY_ALL) return lumc_scene_upload(ctx, v);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());  // nothing renders from the arrays that are about to be freed
  if (scene_update(ctx, v, dirty & LUMC_DIRTY_ALL)) { free_scene(ctx); return 1; }  // a failed partial update leaves no half-updated scene behind
  return 0;
}

int lumc_download_luts(LumContext* ctx, uint16_t* conductor, uint16_t* glossy, uint16_t* dielectric, uint16_t* dielectric_inv) {
  if (!ctx || !ctx->has_scene) return 1;
  uint16_t* dst[4] = {conductor, glossy, dielectric, dielectric_inv};
  const uint32_t lut_count[4] = {1024, 1024, 32768, 32768};
  for (int t = 0; t < 4; t++)
    if (dst[t]) HIP_TRY(ctx, hipMemcpy(dst[t], ctx->d_luts[t], sizeof(uint16_t) * lut_count[t], hipMemcpyDeviceToHost));
  return 0;
}

int lumc_download_sky_luts(LumContext* ctx, float* transmittance, float* multiscattering) {
  if (!ctx || !ctx->has_scene || !ctx->scene.sky_lut_transmittance) { if (ctx) ctx->error = "lumc_download_sky_luts: the scene has no procedural sky"; return 1; }
  if (transmittance) HIP_TRY(ctx, hipMemcpy(transmittance, ctx->scene.sky_lut_transmittance, sizeof(float4) * 2 * kSkyTmWidth * kSkyTmHeight, hipMemcpyDeviceToHost));
  if (multiscattering) HIP_TRY(ctx, hipMemcpy(multiscattering, ctx->scene.sky_lut_multiscattering, sizeof(float4) * 2 * kSkyMsSize * kSkyMsSize, hipMemcpyDeviceToHost));
  return 0;
}

// Everything the bake reads: the sky's parameters (not its tables: they are functions of the parameters), the star field's size, the moon.
static std::vector<uint32_t> sky_hdri_key(const DeviceScene& sc, uint32_t ctx_cloud_seed, const float origin[3], uint32_t dim, uint32_t samples) {
  std::vector<uint32_t> key;
  auto put = [&](const void* p, size_t bytes) { const size_t at = key.size(); key.resize(at + (bytes + 3) / 4, 0u); std::memcpy(key.data() + at, p, bytes); };
  put(&sc.sky_steps, sizeof(sc.sky_steps)); put(&sc.sky_ozone_absorption, sizeof(sc.sky_ozone_absorption));
  put(sc.sky_geometry_offset, sizeof(sc.sky_geometry_offset));
  const float params[] = {sc.sky_sun_strength, sc.sky_base_density, sc.sky_rayleigh_density, sc.sky_mie_density, sc.sky_ozone_density, sc.sky_rayleigh_falloff, sc.sky_mie_falloff,
                          sc.sky_ground_visibility, sc.sky_ozone_layer_thickness, sc.sky_multiscattering_factor, sc.sky_moon_tex_offset, sc.sky_stars_intensity};
  put(params, sizeof(params));
  put(sc.sky_sun_pos, sizeof(sc.sky_sun_pos)); put(sc.sky_mie_phase, sizeof(sc.sky_mie_phase)); put(sc.sky_moon_pos, sizeof(sc.sky_moon_pos));
  put(&sc.sky_stars_count, sizeof(sc.sky_stars_count));
  put(&sc.cloud_active, sizeof(sc.cloud_active));
  if (sc.cloud_active) {  // the clouds are baked in (sky_hdri.cuh:88-92)
    const uint32_t ints[] = {sc.cloud_atmosphere_scattering, sc.cloud_steps, sc.cloud_shadow_steps, sc.cloud_octaves};
    const float floats[] = {sc.cloud_offset_x, sc.cloud_offset_z, sc.cloud_density, sc.cloud_noise_shape_scale, sc.cloud_noise_detail_scale, sc.cloud_noise_weather_scale};
    put(ints, sizeof(ints)); put(floats, sizeof(floats)); put(sc.cloud_phase, sizeof(sc.cloud_phase)); put(sc.cloud_layers, sizeof(sc.cloud_layers));
    const uint64_t tex[] = {(uint64_t) (uintptr_t) sc.cloud_noise_shape, (uint64_t) (uintptr_t) sc.cloud_noise_weather, (uint64_t) ctx_cloud_seed};
    put(tex, sizeof(tex));
  }
  put(origin, 3 * sizeof(float)); put(&dim, sizeof(dim)); put(&samples, sizeof(samples));
  return key;
}

int lumc_sky_hdri_build(LumContext* ctx, const float origin[3], uint32_t dim, uint32_t samples) {
  if (!ctx || !origin || !ctx->has_scene || !ctx->scene.sky_lut_transmittance) { if (ctx) ctx->error = "lumc_sky_hdri_build: the scene has no atmosphere (constant-colour sky)"; return 1; }
  if (dim < 2 || dim > 16384 || samples == 0) { ctx->error = "lumc_sky_hdri_build: dim must be in [2, 16384] and samples positive"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> key = sky_hdri_key(ctx->scene, ctx->cloud_noise_seed, origin, dim, samples);
  if (ctx->d_sky_hdri && key == ctx->sky_hdri_key) {
    if (ctx->scene.sky_mode == kSkyHdri) { ctx->scene.sky_hdri = ctx->d_sky_hdri; ctx->scene.sky_hdri_dim = dim; }
    return 0;
  }
  ctx->sky_hdri_key.clear();
  if (ctx->sky_hdri_dim != dim) {
    if (ctx->d_sky_hdri) (void) hipFree(ctx->d_sky_hdri);
    ctx->d_sky_hdri = nullptr; ctx->sky_hdri_dim = 0;
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_sky_hdri, sizeof(float4) * (size_t) dim * dim));
    ctx->sky_hdri_dim = dim;
  }
  const uint64_t threads = (uint64_t) dim * dim * 32u;
  hipLaunchKernelGGL(k_sky_hdri, dim3((uint32_t) ((threads + 255) / 256)), dim3(256), 0, 0, ctx->scene, origin[0], origin[1], origin[2], dim, samples, ctx->d_sky_hdri);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  ctx->sky_hdri_key = std::move(key);
  if (ctx->scene.sky_mode == kSkyHdri) { ctx->scene.sky_hdri = ctx->d_sky_hdri; ctx->scene.sky_hdri_dim = dim; }
  return 0;
}

int lumc_sky_hdri_download(LumContext* ctx, float* rgba, uint32_t* dim) {
  if (!ctx || !ctx->d_sky_hdri) { if (ctx) ctx->error = "lumc_sky_hdri_download: no baked sky"; return 1; }
  if (dim) *dim = ctx->sky_hdri_dim;
  if (rgba) HIP_TRY(ctx, hipMemcpy(rgba, ctx->d_sky_hdri, sizeof(float4) * (size_t) ctx->sky_hdri_dim * ctx->sky_hdri_dim, hipMemcpyDeviceToHost));
  return 0;
}

// FNV-1a over a pixel list (null: 0, 1, 2 ... n - 1): the identity of a context's pixel set and of its ORDER
static uint64_t pixel_list_hash(const uint32_t* pixels, uint32_t n) {
  uint64_t h = 1469598103934665603ull;
  for (uint32_t i = 0; i < n; i++) { h ^= pixels ? pixels[i] : i; h *= 1099511628211ull; }
  return h;
}

int lumc_set_pixels(LumContext* ctx, const uint32_t* pixels, uint32_t num_pixels) {
  if (!ctx || !ctx->has_scene) { if (ctx) ctx->error = "lumc_set_pixels: no scene"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!pixels) num_pixels = ctx->scene.width * ctx->scene.height;
  free_adaptive(ctx);
  if (ctx->d_pixels) { (void) hipFree(ctx->d_pixels); ctx->d_pixels = nullptr; }
  if (ctx->d_first_moment) { (void) hipFree(ctx->d_first_moment); ctx->d_first_moment = nullptr; }
  if (ctx->d_second_moment) { (void) hipFree(ctx->d_second_moment); ctx->d_second_moment = nullptr; }
  ctx->num_pixels = num_pixels;
  ctx->pixels_hash = pixel_list_hash(pixels, num_pixels);  // what lumc_frame_gather checks the set against (a null list = the frame in row-major order)
  if (num_pixels == 0) return 0;
  if (pixels) {
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_pixels, sizeof(uint32_t) * num_pixels));
    HIP_TRY(ctx, hipMemcpy(ctx->d_pixels, pixels, sizeof(uint32_t) * num_pixels, hipMemcpyHostToDevice));
  }
  HIP_TRY(ctx, hipMalloc((void**) &ctx->d_first_moment, sizeof(float) * 3 * (size_t) num_pixels));
  HIP_TRY(ctx, hipMalloc((void**) &ctx->d_second_moment, sizeof(float) * (size_t) num_pixels));
  return lumc_clear_accumulators(ctx);
}

int lumc_clear_accumulators(LumContext* ctx) {
  if (!ctx || !ctx->d_first_moment) return 1;
  HIP_TRY(ctx, hipMemset(ctx->d_first_moment, 0, sizeof(float) * 3 * (size_t) ctx->num_pixels));
  HIP_TRY(ctx, hipMemset(ctx->d_second_moment, 0, sizeof(float) * (size_t) ctx->num_pixels));
  return 0;
}

// The particle pass of the closest-hit kernel: the same traversal on the particle tree (the scene copy carries it in place of the surfaces' tree).
static void trace_particles(LumContext* ctx, hipStream_t stream, const PathQueue& q, uint32_t* ctrl, uint32_t N) {
  DeviceScene tree = ctx->scene;
  tree.bvh_nodes = tree.particle_bvh_nodes; tree.blas_tris = tree.particle_tris; tree.tlas_leaves = tree.particle_leaves; tree.tlas_num_nodes = tree.particle_tlas_num_nodes; tree.tlas_num_leaves = tree.particle_num_leaves;
  Launch l(ctx, stream, LUMC_KERNEL_TRACE);
  ctx->wf->trace_particles(grid_persistent(ctx, N), (size_t) ctx->particle_lds_nodes * kNodeBytes + LUM_LDS_STACK_BYTES, stream, tree, q, ctrl, ctx->particle_lds_nodes);
}

// Does the next pass reuse the closest-hit rays for the ambient visibility (lumc_set_ambient_reuse)? Plain scenes only: with fog the vertex's ambient term
// is dimmed along the packed direction, an ocean ends the ambient ray at the water surface, particles and the ocean replace closest hits after the
// pass that would answer, clouds and the procedural sky have no ambient sample; the reorder of sort mode 3 does not move hit_scene_tri.
static bool ambient_reuse_active(const LumContext* ctx) {
  const DeviceScene& sc = ctx->scene;
  // -1: by flavour. The fast flavour: on. The exact flavour: off - asked for (1), it takes only the answers it can prove for the ambient ray itself
  // (k_resolve_reuse re-tests that ray against the hit's triangle) and stays bit-identical to the oracle, but the proof's gathers cost more than the
  // cheap rays they save (hall: visibility kernel -29 ms, resolve +61 ms per step), so it is not its default.
  const bool wanted = ctx->ambient_reuse < 0 ? (ctx->wf == wavefront_kernels_fast()) : ctx->ambient_reuse != 0;
  return wanted && ctx->has_scene && sc.sky_mode != kSkyDefault && !sc.fog_active && !sc.ocean_active && !sc.particles_active && !sc.cloud_active &&
         !sc.sky_aerial_perspective && ctx->sort_mode == 0 && sc.shading_mode == 0u;
}

// The depth loop of one wavefront pass over the paths k_generate* left in queue[0] (at most N of them, counted on the device).
// `first_sample`, `sample_count`: the pass's sample ids when they are one contiguous range for every pixel (lumc_render), 0 otherwise.
static int wavefront_depths(LumContext* ctx, hipStream_t stream, uint32_t N, uint32_t first_sample = 0, uint32_t sample_count = 0) {
  DeviceScene sc = ctx->scene;
  const uint32_t max_depth = sc.max_ray_depth;
  sc.sobol_table = nullptr;
  if (ctx->sobol_table && sample_count > 0 && sample_count <= kSobolTableMaxSamples && sc.shading_mode == 0u) {
    // the Sobol / Owen pairs of this pass's sample ids for every dimension k_shade can ask for (dev_sampler.h LUM_SOBOL_TABLE): 1.3 MB at 32 ids and 8 bounces
    const uint32_t stride = (sample_count + 15u) & ~15u, dims = (max_depth + 1u) * kRndTargetCount;
    const size_t entries = (size_t) stride * dims;
    if (ctx->sobol_entries < entries) {
      if (ctx->d_sobol) (void) hipFree(ctx->d_sobol);
      ctx->d_sobol = nullptr; ctx->sobol_entries = 0;
      if (hipMalloc((void**) &ctx->d_sobol, entries * sizeof(uint2)) == hipSuccess) ctx->sobol_entries = entries;
      else (void) hipGetLastError();  // no room: the sampler hashes
    }
    if (ctx->d_sobol) {
      ctx->wf->sobol_table(stream, ctx->d_sobol, first_sample, sample_count, stride, dims);
      sc.sobol_table = ctx->d_sobol; sc.sobol_first = first_sample; sc.sobol_count = sample_count; sc.sobol_stride = stride;
    }
  }
  const size_t lds_dyn = (size_t) ctx->lds_nodes * kNodeBytes + LUM_LDS_STACK_BYTES;
  int cur = 0;
  const WavefrontKernels& wf = *ctx->wf;
  const bool render_volumes = sc.fog_active || sc.ocean_active;  // device_manager.c:478
  if (sc.shading_mode != 0u) {  // debug shading modes: one closest-hit pass and a colour per path (device_renderer.c:136-181)
    {
      Launch l(ctx, stream, LUMC_KERNEL_TRACE);
      wf.trace(grid_persistent(ctx, N), lds_dyn, stream, sc, ctx->queue[0], nullptr, ctx->d_ctrl, ctx->d_counters, ctx->lds_nodes);
    }
    if (sc.particles_active) trace_particles(ctx, stream, ctx->queue[0], ctx->d_ctrl, N);
    if (sc.ocean_active) {
      Launch l(ctx, stream, LUMC_KERNEL_TRACE);
      wf.trace_ocean(grid_for(N), stream, sc, ctx->queue[0], (const uint32_t*) ctx->d_ctrl);
    }
    if (render_volumes) {  // the debug queue keeps volume_process_events (device_renderer.c:145-147)
      Launch l(ctx, stream, LUMC_KERNEL_VOLUME);
      wf.volume_events(grid_for(N), stream, sc, ctx->queue[0], ctx->volume, ctx->d_results, ctx->d_ctrl, 0u);
    }
    if (sc.sky_aerial_perspective && sc.sky_mode != kSkyConstantColor) {  // the debug queue keeps the in-scattering events (device_renderer.c:150-154)
      Launch l(ctx, stream, LUMC_KERNEL_SKY);
      wf.sky_inscattering(grid_for(N), stream, sc, ctx->queue[0], ctx->d_results, (const uint32_t*) ctx->d_ctrl, 0u);
    }
    Launch l(ctx, stream, LUMC_KERNEL_SHADE);
    wf.shade_debug(grid_for(N), stream, sc, ctx->queue[0], ctx->d_results, (const uint32_t*) ctx->d_ctrl);
    return 0;
  }
  // Ambient-visibility reuse (lumc_set_ambient_reuse; AmbientReuse in kernels.h): the vertices of depth d leave their ambient sample to the closest-hit
  // pass of depth d + 1, which is followed by a second, small visibility pass (what the closest hit could not decide) and only then by the resolve of
  // depth d - still before k_shade of depth d + 1 touches the result slots, so the order of the sums is the usual one.
  const bool reuse = ambient_reuse_active(ctx);
  // Fused resolve (FusedResolve, kernels.h): with the fast flavour's reuse the resolve of depth d is done by k_shade of depth d + 1 for the vertices an entry
  // continues, by k_resolve_ended for the others; the queues rotate through three buffers and the NEE records through two, so that depth d is intact while
  // depth d + 1 is shaded. The exact flavour's (provable) reuse keeps its own kernel: its sums must land in the reference's order.
  bool fused = reuse && wf.fused_resolve && ctx->fused_resolve != 0 && ctx->wf == wavefront_kernels_fast() && max_depth > 0;
  if (fused && ensure_fused(ctx, stream)) {  // no room for its buffers (a third of the work buffers again): the separate resolve kernel does the same sums
    (void) hipGetLastError();
    ctx->error.clear();
    fused = false;
  }
  // (cur == depth % 3 and the record set == depth & 1 below: what the six device records assume)
  bool resolve_pending = false;  // the previous depth's resolve waits for this depth's closest-hit pass
  for (uint32_t depth = 0; depth <= max_depth; depth++) {
    // the sampler's depth constant is not advanced before the last pass (device_renderer.c:126-130)
    const uint32_t depth_const = (depth == max_depth && depth > 0) ? depth - 1 : depth;
    uint32_t* ctrl = ctx->d_ctrl + kCtlStride * depth;
    // camera rays leave k_generate in pixel order, which is as coherent as rays get; later depths are sorted on request
    const uint32_t* order = nullptr;
    if (ctx->sort_mode >= 1 && depth >= 1) {
      order = sort_rays(ctx, stream, ctx->queue[cur].origin_t, ctx->queue[cur].dir_slot, ctrl + kCtlPaths, N);
      if (!order) { ctx->error = "ray sorting failed"; return 1; }
      if (ctx->sort_mode == 3) {  // physical reorder: the planes change places, the permutation is spent
        if (ensure_sort_queue(ctx, ctx->capacity)) return 1;
        Launch l(ctx, stream, LUMC_KERNEL_SORT);
        PathQueue& q = ctx->queue[cur];
        hipLaunchKernelGGL(k_permute_queue, dim3(std::min<uint32_t>((N + 255u) / 256u, 65536u)), dim3(256), 0, stream, q, ctx->sort_queue, order, ctrl + kCtlPaths, N);
        std::swap(q.origin_t, ctx->sort_queue.origin_t); std::swap(q.dir_slot, ctx->sort_queue.dir_slot);
        std::swap(q.aux, ctx->sort_queue.aux); std::swap(q.hit_id, ctx->sort_queue.hit_id);
        ctx->fused_records_stale = true;  // (the fused resolve does not run with ray sorting; a later pass without it must not read the old planes)
        order = nullptr;
      }
    }
    {
      Launch l(ctx, stream, LUMC_KERNEL_TRACE);
      wf.trace(grid_persistent(ctx, N), lds_dyn, stream, sc, ctx->queue[cur], order, ctrl, ctx->d_counters, ctx->lds_nodes);
    }
    const int next_q = fused ? (cur + 1) % 3 : (cur ^ 1), prev_q = fused ? (cur + 2) % 3 : (cur ^ 1);
    NeeQueue& nee = (fused && (depth & 1u)) ? ctx->nee2 : ctx->nee;
    NeeQueue& nee_before = (fused && (depth & 1u)) ? ctx->nee : ctx->nee2;
    if (resolve_pending) {  // the previous depth's resolve: ambient samples answered by the pass above; what it cannot answer is traced (the control words of the fog's visibility pass: no fog here) and resolved after
      uint32_t* prev = ctrl - kCtlStride;
      {
        Launch l(ctx, stream, LUMC_KERNEL_RESOLVE);
        wf.resolve_reuse(grid_for(N), stream, sc, ctx->queue[cur ^ 1], ctx->queue[cur], ctx->nee, ctx->shadow, ctx->d_results, prev, ctx->d_counters);
      }
      {
        Launch l(ctx, stream, LUMC_KERNEL_SHADOW);
        wf.shadow_rays(ctx->trace_blocks, lds_dyn, stream, sc, ctx->shadow, nullptr, prev + kCtlVolumeShift, ctx->d_counters, ctx->lds_nodes);
      }
      Launch l(ctx, stream, LUMC_KERNEL_RESOLVE);
      wf.resolve_listed(std::min<uint32_t>(grid_for(N), 1024u), stream, sc, ctx->queue[cur ^ 1], ctx->nee, ctx->shadow, ctx->d_results, (const uint32_t*) prev);
      resolve_pending = false;
    }
    if (sc.particles_active) trace_particles(ctx, stream, ctx->queue[cur], ctrl, N);  // optix_kernel_raytrace.cu:171
    if (sc.ocean_active) {  // optix_kernel_raytrace.cu:134-144, :172
      Launch l(ctx, stream, LUMC_KERNEL_TRACE);
      wf.trace_ocean(grid_for(N), stream, sc, ctx->queue[cur], (const uint32_t*) ctrl);
    }
    if (render_volumes) {  // device_renderer.c:64-76: in-scattering with its own visibility pass, then the scattering events
      {
        Launch l(ctx, stream, LUMC_KERNEL_VOLUME);
        wf.volume_inscatter(grid_for(N), stream, sc, ctx->queue[cur], ctx->volume, ctx->shadow, ctrl, depth_const);
      }
      {
        Launch l(ctx, stream, LUMC_KERNEL_SHADOW);
        wf.shadow_rays(grid_persistent(ctx, N), lds_dyn, stream, sc, ctx->shadow, nullptr, ctrl + kCtlVolumeShift, ctx->d_counters, ctx->lds_nodes);
      }
      Launch l(ctx, stream, LUMC_KERNEL_VOLUME);
      wf.volume_resolve(grid_for(N), stream, sc, ctx->queue[cur], ctx->volume, ctx->shadow, ctx->d_results, (const uint32_t*) ctrl);
      wf.volume_events(grid_for(N), stream, sc, ctx->queue[cur], ctx->volume, ctx->d_results, ctrl, depth_const);
    }
    if (sc.cloud_active && sc.sky_mode == kSkyDefault && sc.cloud_noise_shape) {  // device_manager.c:474, device_renderer.c:78-82
      Launch l(ctx, stream, LUMC_KERNEL_SKY);
#if LUM_CLOUD_PERSISTENT
      wf.clouds_list(grid_for(N), stream, sc, ctx->queue[cur], ctx->cloud, ctrl);
      wf.clouds_march(ctx->trace_blocks * 4u, stream, sc, ctx->queue[cur], ctx->cloud, ctrl, depth_const);  // persistent: 4 workgroups of 256 per CU
#endif
      wf.clouds(grid_for(N), stream, sc, ctx->queue[cur], ctx->cloud, ctx->d_results, (const uint32_t*) ctrl, depth_const);
    }
    if (sc.sky_aerial_perspective && sc.sky_mode != kSkyConstantColor) {  // device_manager.c:475, device_renderer.c:84-88
      Launch l(ctx, stream, LUMC_KERNEL_SKY);
      wf.sky_inscattering(grid_for(N), stream, sc, ctx->queue[cur], ctx->d_results, (const uint32_t*) ctrl, depth_const);
    }
    {
      Launch l(ctx, stream, LUMC_KERNEL_SHADE);
      wf.shade(shade_grid(ctx, N), stream, sc, ctx->queue[cur], ctx->queue[next_q], nee, ctx->shadow, ctx->d_results, ctrl, depth_const, ctx->d_counters,
               (reuse && depth < max_depth) ? 1u : 0u, fused ? ctx->d_fused + depth % 6u : nullptr,
               fused ? ((depth > 0 ? 1u : 0u) | (depth < max_depth ? 2u : 0u) | (ctx->fused_ended ? 4u : 0u)) : 0u);
    }
    if (fused && depth > 0) {  // the samples of depth - 1 their paths' closest hits could not decide: traced now, their vertices resolved (before this depth's visibility pass reuses the words)
      {
        Launch l(ctx, stream, LUMC_KERNEL_SHADOW);
        wf.shadow_rays(ctx->trace_blocks, lds_dyn, stream, sc, ctx->fallback, nullptr, ctrl + kCtlVolumeShift, ctx->d_counters, ctx->lds_nodes);
      }
      Launch l(ctx, stream, LUMC_KERNEL_RESOLVE);
      wf.resolve_listed(std::min<uint32_t>(grid_for(N), 1024u), stream, sc, ctx->queue[prev_q], nee_before, ctx->fallback, ctx->d_results, (const uint32_t*) ctrl);
    }
    if (sc.particles_active) {  // device_renderer.c:99-103
      Launch l(ctx, stream, LUMC_KERNEL_SHADE);
      wf.particle_shade(grid_for(N), stream, sc, ctx->queue[cur], ctx->queue[cur ^ 1], ctx->nee, ctx->shadow, ctrl, depth_const);
    }
    if (sc.ocean_active) {  // device_renderer.c:104-108
      Launch l(ctx, stream, LUMC_KERNEL_SHADE);
      wf.ocean_shade(grid_for(N), stream, sc, ctx->queue[cur], ctx->queue[cur ^ 1], ctx->nee, ctx->shadow, ctrl, depth_const);
    }
    if (sc.sky_mode == kSkyDefault) {  // paths that left the scene into the procedural sky (listed by k_shade)
      Launch l(ctx, stream, LUMC_KERNEL_SKY);
      wf.sky(grid_for(N), stream, sc, ctx->queue[cur], ctx->shadow, ctx->d_results, (const uint32_t*) ctrl, depth_const);
    }
    {
      Launch l(ctx, stream, LUMC_KERNEL_LIGHT_QUERY);
      // (one resident round of its workgroups - four per CU: the kernel's workgroups are dear to start (a 1 KB stack per lane in scratch); 2 rounds, the common cap:
      //  Example-class 4.6 -> 4.0 ms per 3 steps, scan 3.9 -> 3.6, hall equal; 4 / 8 / 16 rounds on the hall: 30.2 / 32.8 / 44.5 ms against 29.9)
      wf.light_query(std::min<uint32_t>(grid_for(N), ctx->trace_blocks * 4u), stream, sc, ctx->queue[cur], nee, ctx->shadow, ctrl, depth_const, ctx->d_counters);
    }
    const uint32_t* shadow_order = nullptr;
    if (ctx->sort_mode == 2) {
      shadow_order = sort_rays(ctx, stream, ctx->shadow.origin_dist, ctx->shadow.dir_out, ctrl + kCtlShadowItems,
                               (sc.ocean_active ? kSurfaceShadowKindsWater : 4u) * (ctx->shadow.capacity < N ? ctx->shadow.capacity : N));
      if (!shadow_order) { ctx->error = "ray sorting failed"; return 1; }
    }
    {
      Launch l(ctx, stream, LUMC_KERNEL_SHADOW);
      wf.shadow_rays(grid_persistent(ctx, N), lds_dyn, stream, sc, ctx->shadow, shadow_order, ctrl, ctx->d_counters, ctx->lds_nodes);
    }
    if (fused && depth < max_depth) {  // the vertices no entry of the next depth continues; the others are resolved by those entries, in k_shade - and so are these, as its last input (fused_flags & 4)
      if (!ctx->fused_ended) {
        Launch l(ctx, stream, LUMC_KERNEL_RESOLVE);
        wf.resolve_ended(std::min<uint32_t>(grid_for(N), 4096u), stream, sc, ctx->queue[cur], nee, ctx->shadow, ctx->d_results, (const uint32_t*) ctrl, ctx->d_ended[depth & 1u]);
      }
    }
    else if (reuse && depth < max_depth) resolve_pending = true;
    else {
      Launch l(ctx, stream, LUMC_KERNEL_RESOLVE);
      wf.resolve(grid_for(N), stream, sc, ctx->queue[cur], nee, ctx->shadow, ctx->d_results, (const uint32_t*) ctrl);
    }
    if (render_volumes && depth != max_depth) {  // device_renderer.c:114-118
      Launch l(ctx, stream, LUMC_KERNEL_VOLUME);
      wf.volume_bounce(grid_for(N), stream, sc, ctx->queue[cur], ctx->queue[cur ^ 1], ctx->volume, ctrl, depth_const);
    }
    cur = next_q;
  }
  return 0;
}

int lumc_render(LumContext* ctx, uint32_t first_sample, uint32_t num_samples, uint32_t samples_per_pass, float* d_fm, float* d_sm, void* stream_) {
  if (!ctx || !ctx->has_scene) { if (ctx) ctx->error = "lumc_render: no scene"; return 1; }
  if (ctx->num_pixels == 0) return 0;
  hipStream_t stream = (hipStream_t) stream_;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!d_fm) { d_fm = ctx->d_first_moment; d_sm = ctx->d_second_moment; }
  if (samples_per_pass == 0) samples_per_pass = 1;
  // sample ids beyond 2^20 would duplicate earlier ones (cuda/kernels.cuh:103-105)
  if (first_sample >= kMaxGlobalSamples) return 0;
  if (first_sample + (uint64_t) num_samples > kMaxGlobalSamples) num_samples = kMaxGlobalSamples - first_sample;
  const uint32_t P = ctx->num_pixels;
  const uint64_t want = (uint64_t) P * samples_per_pass;
  if (want > 0x7FFFFFFFull) { ctx->error = "pass too large"; return 1; }
  if (ensure_work(ctx, (uint32_t) want)) return 1;
  const DeviceScene& sc = ctx->scene;
  const uint32_t max_depth = sc.max_ray_depth;

  for (uint32_t done = 0; done < num_samples; done += samples_per_pass) {
    const uint32_t batch = std::min(samples_per_pass, num_samples - done);
    const uint32_t N = P * batch;
    PassParams pp{ctx->d_pixels, P, batch, first_sample + done};
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_ctrl, 0, sizeof(uint32_t) * kCtlStride * (max_depth + 2), stream));
    {
      Launch l(ctx, stream, LUMC_KERNEL_GENERATE);
      ctx->wf->generate(grid_for(N), stream, sc, pp, ctx->queue[0], ctx->d_results, ctx->d_ctrl + kCtlPaths);
    }
    if (wavefront_depths(ctx, stream, N, first_sample + done, batch)) return 1;
    {
      Launch l(ctx, stream, LUMC_KERNEL_ACCUMULATE);
      hipLaunchKernelGGL(k_accumulate, dim3(grid_for(P)), dim3(kBlock), 0, stream, (const float4*) ctx->d_results, P, batch, d_fm, d_sm);
    }
    HIP_TRY(ctx, hipGetLastError());
  }
  return 0;
}

// The pixels of one iteration of the undersampling preview (tasks_create, cuda/kernels.cuh:47-95, whole-frame window): one per block of
// 2^stage pixels, at the block's corner or half a block in, by the iteration's two bits.
static std::vector<uint32_t> undersampling_pixels(uint32_t width, uint32_t height, uint32_t stage, uint32_t iteration) {
  const uint32_t scale = 1u << stage;
  const uint32_t uw = (width + scale - 1) >> stage, uh = (height + scale - 1) >> stage;
  std::vector<uint32_t> px;
  px.reserve((size_t) uw * uh);
  for (uint32_t id = 0; id < uw * uh; id++) {
    uint32_t y = id / uw, x = id - y * uw;
    if (scale > 1) {
      x = x * scale + ((iteration & 1u) ? 0u : scale >> 1);
      y = y * scale + ((iteration & 2u) ? 0u : scale >> 1);
    }
    if (x >= width || y >= height) continue;
    px.push_back(x + y * width);
  }
  return px;
}

int lumc_render_undersampled(LumContext* ctx, uint32_t stage, uint32_t iteration, void* stream_) {
  if (!ctx || !ctx->has_scene || !ctx->d_first_moment || ctx->d_pixels || ctx->num_pixels != ctx->scene.width * ctx->scene.height) {
    if (ctx) ctx->error = "lumc_render_undersampled: needs the full-frame accumulators";
    return 1;
  }
  if (stage == 0 || stage > 15 || iteration > 3) { ctx->error = "lumc_render_undersampled: stage in [1, 15], iteration in [0, 3]"; return 1; }
  hipStream_t stream = (hipStream_t) stream_;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const DeviceScene& sc = ctx->scene;
  const std::vector<uint32_t> px = undersampling_pixels(sc.width, sc.height, stage, iteration);
  const uint32_t n = (uint32_t) px.size();
  if (n == 0) return 0;
  if (ctx->undersampling_capacity < n) {
    if (ctx->d_undersampling_pixels) (void) hipFree(ctx->d_undersampling_pixels);
    ctx->d_undersampling_pixels = nullptr; ctx->undersampling_capacity = 0;
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_undersampling_pixels, sizeof(uint32_t) * (size_t) n));
    ctx->undersampling_capacity = n;
  }
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_undersampling_pixels, px.data(), sizeof(uint32_t) * (size_t) n, hipMemcpyHostToDevice, stream));
  HIP_TRY(ctx, hipStreamSynchronize(stream));  // the list leaves scope
  if (ensure_work(ctx, n)) return 1;
  PassParams pp{ctx->d_undersampling_pixels, n, 1u, 0u};  // every pixel's first sample
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_ctrl, 0, sizeof(uint32_t) * kCtlStride * (sc.max_ray_depth + 2), stream));
  {
    Launch l(ctx, stream, LUMC_KERNEL_GENERATE);
    ctx->wf->generate(grid_for(n), stream, sc, pp, ctx->queue[0], ctx->d_results, ctx->d_ctrl + kCtlPaths);
  }
  if (wavefront_depths(ctx, stream, n)) return 1;
  {
    Launch l(ctx, stream, LUMC_KERNEL_ACCUMULATE);
    hipLaunchKernelGGL(k_accumulate_scatter, dim3(grid_for(n)), dim3(kBlock), 0, stream, (const float4*) ctx->d_results, (const uint32_t*) ctx->d_undersampling_pixels, n,
                       ctx->num_pixels, ctx->d_first_moment, ctx->d_second_moment);
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// ---- adaptive sampling ----
namespace {

AdaptiveView adaptive_view(const LumContext* ctx) {
  const LumContext::Adaptive& a = ctx->adaptive;
  AdaptiveView v;
  v.stage_counts = a.d_stage_counts; v.block_task_end = a.d_block_task_end;
  v.blocks_x = a.blocks_x; v.blocks_y = a.blocks_y; v.num_blocks = a.num_blocks;
  for (uint32_t s = 0; s <= kAdaptiveStages; s++) v.executions[s] = a.executions[s];
  v.stage_id = a.stage_id;
  return v;
}

OutputParams tone_params(const LumOutputParams* p) {
  OutputParams op;
  std::memset(&op, 0, sizeof(op));
  if (p) std::memcpy(&op, p, sizeof(op));
  return op;
}

// adaptive_sampler_compute_next_stage (device_adaptive_sampler.c:105-215) in two halves: the block variances measured so far, then the
// rates of stage `stage_id + 1` from them. Between the halves a partitioned render exchanges the variances of the ranks' blocks.
int adaptive_compute_variance(LumContext* ctx, hipStream_t stream) {
  LumContext::Adaptive& a = ctx->adaptive;
  const DeviceScene& sc = ctx->scene;
  const AdaptiveView view = adaptive_view(ctx);
  const OutputParams op = tone_params(&a.params.tone);
  hipLaunchKernelGGL(k_adaptive_block_variance, dim3((a.num_blocks * 16 + 255) / 256), dim3(256), 0, stream, view, op, sc.width, sc.height, a.params.exposure,
                     (const float*) ctx->d_first_moment, (const float*) ctx->d_second_moment, a.d_block_variance);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// Inclusive prefix over d_block_tasks and its host copy (passes are cut at block boundaries).
int adaptive_task_prefix(LumContext* ctx, hipStream_t stream) {
  LumContext::Adaptive& a = ctx->adaptive;
  const uint32_t nb = a.num_blocks;
  HIP_TRY(ctx, hipcub::DeviceScan::InclusiveSum(a.d_scan_temp, a.scan_temp_bytes, a.d_block_tasks, a.d_block_task_end, (int) nb, stream));
  a.task_end.resize(nb);
  HIP_TRY(ctx, hipMemcpyAsync(a.task_end.data(), a.d_block_task_end, sizeof(uint32_t) * nb, hipMemcpyDeviceToHost, stream));
  HIP_TRY(ctx, hipStreamSynchronize(stream));
  return 0;
}

int adaptive_finish_build(LumContext* ctx, hipStream_t stream) {
  LumContext::Adaptive& a = ctx->adaptive;
  const uint32_t nb = a.num_blocks, chunks = (nb + kAdaptiveSumChunk - 1) / kAdaptiveSumChunk;
  hipLaunchKernelGGL(k_adaptive_sum_chunks, dim3((chunks + 63) / 64), dim3(64), 0, stream, (const float*) a.d_block_variance, nb, a.d_partial);
  hipLaunchKernelGGL(k_adaptive_sum_total, dim3(1), dim3(1), 0, stream, (const float*) a.d_partial, chunks, a.d_partial + chunks);
  hipLaunchKernelGGL(k_adaptive_stage_counts, dim3((nb + 255) / 256), dim3(256), 0, stream, (const float*) a.d_block_variance, (const float*) (a.d_partial + chunks), nb,
                     a.stage_id, a.params.max_sampling_rate, a.params.avg_sampling_rate, a.d_stage_counts, a.d_block_tasks, (const uint8_t*) a.d_block_mask);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(&a.variance_total, a.d_partial + chunks, sizeof(float), hipMemcpyDeviceToHost, stream));
  if (adaptive_task_prefix(ctx, stream)) return 1;
  a.stage_id++;
  a.build_pending = false;
  return 0;
}

int adaptive_build_stage(LumContext* ctx, hipStream_t stream) {
  if (adaptive_compute_variance(ctx, stream)) return 1;
  return adaptive_finish_build(ctx, stream);
}

// `merged` consecutive executions of stage >= 1 as one set of passes of whole blocks (tasks_create_adaptive_sampling + the usual depth
// loop + accumulation). Merging keeps the passes large enough to fill the GPU when the rates are low.
constexpr uint32_t kAdaptiveTasksPerPass = 16u << 20;

int adaptive_execute(LumContext* ctx, hipStream_t stream, uint32_t merged) {
  LumContext::Adaptive& a = ctx->adaptive;
  const DeviceScene& sc = ctx->scene;
  const uint32_t nb = a.num_blocks;
  const AdaptiveView view = adaptive_view(ctx);
  uint32_t block = 0;
  while (block < nb) {
    AdaptivePass pass;
    pass.executions = merged;
    pass.block_begin = block;
    pass.task_begin = (block ? a.task_end[block - 1] : 0u) * merged;
    // as many whole blocks as fit the pass (a single block has at most 16 * 256 tasks per execution)
    const uint32_t limit = (pass.task_begin + kAdaptiveTasksPerPass) / merged;
    uint32_t end = (uint32_t) (std::upper_bound(a.task_end.begin() + block, a.task_end.end(), limit) - a.task_end.begin());
    if (end == block) end = block + 1;
    pass.block_end = end;
    pass.task_end = a.task_end[end - 1] * merged;
    const uint32_t N = pass.task_end - pass.task_begin;
    if (ensure_work(ctx, N)) return 1;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_ctrl, 0, sizeof(uint32_t) * kCtlStride * (sc.max_ray_depth + 2), stream));
    {
      Launch l(ctx, stream, LUMC_KERNEL_GENERATE);
      ctx->wf->generate_adaptive(grid_for(N), stream, sc, view, pass, ctx->queue[0], ctx->d_results, ctx->d_ctrl + kCtlPaths);
    }
    if (wavefront_depths(ctx, stream, N)) return 1;
    {
      Launch l(ctx, stream, LUMC_KERNEL_ACCUMULATE);
      hipLaunchKernelGGL(k_accumulate_adaptive, dim3(grid_for((end - block) * 16)), dim3(kBlock), 0, stream, view, pass, sc.width, sc.height, (const float4*) ctx->d_results,
                         ctx->d_first_moment, ctx->d_second_moment);
    }
    HIP_TRY(ctx, hipGetLastError());
    block = end;
  }
  a.executions[a.stage_id] += merged;
  return 0;
}

}  // namespace

int lumc_adaptive_begin(LumContext* ctx, const LumAdaptiveParams* params) {
  if (!ctx || !params) { if (ctx) ctx->error = "lumc_adaptive_begin: null argument"; return 1; }
  if (!ctx->has_scene || ctx->d_pixels || ctx->num_pixels != ctx->scene.width * ctx->scene.height || ctx->num_pixels == 0) {
    ctx->error = "lumc_adaptive_begin: needs a scene and the full-frame pixel set (lumc_set_pixels(ctx, NULL, 0))";
    return 1;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  free_adaptive(ctx);
  LumContext::Adaptive& a = ctx->adaptive;
  a.params = *params;
  // adaptive_sampler_setup, device_adaptive_sampler.c:40-58
  a.params.max_sampling_rate = std::min(std::max(params->max_sampling_rate, 1u), kAdaptiveMaxRate);
  a.params.avg_sampling_rate = std::min(std::max(params->avg_sampling_rate, 1u), a.params.max_sampling_rate);
  a.params.update_interval = std::max(params->update_interval, 1u);
  a.blocks_x = (ctx->scene.width + 3u) >> kAdaptiveBlockLog;
  a.blocks_y = (ctx->scene.height + 3u) >> kAdaptiveBlockLog;
  a.num_blocks = a.blocks_x * a.blocks_y;
  const uint32_t nb = a.num_blocks, chunks = (nb + kAdaptiveSumChunk - 1) / kAdaptiveSumChunk;
  HIP_TRY(ctx, hipMalloc((void**) &a.d_stage_counts, sizeof(uint32_t) * nb));
  HIP_TRY(ctx, hipMalloc((void**) &a.d_block_tasks, sizeof(uint32_t) * nb));
  HIP_TRY(ctx, hipMalloc((void**) &a.d_block_task_end, sizeof(uint32_t) * nb));
  HIP_TRY(ctx, hipMalloc((void**) &a.d_block_variance, sizeof(float) * nb));
  HIP_TRY(ctx, hipMalloc((void**) &a.d_partial, sizeof(float) * (chunks + 1)));
  HIP_TRY(ctx, hipMemset(a.d_stage_counts, 0, sizeof(uint32_t) * nb));
  HIP_TRY(ctx, hipMemset(a.d_block_variance, 0, sizeof(float) * nb));
  HIP_TRY(ctx, hipcub::DeviceScan::InclusiveSum(nullptr, a.scan_temp_bytes, a.d_block_tasks, a.d_block_task_end, (int) nb, (hipStream_t) 0));
  HIP_TRY(ctx, hipMalloc(&a.d_scan_temp, std::max<size_t>(a.scan_temp_bytes, 16)));
  a.active = true;
  return lumc_clear_accumulators(ctx);
}

int lumc_adaptive_render(LumContext* ctx, uint32_t executions, void* stream_) {
  if (!ctx || !ctx->adaptive.active) { if (ctx) ctx->error = "lumc_adaptive_render: call lumc_adaptive_begin first"; return 1; }
  hipStream_t stream = (hipStream_t) stream_;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  LumContext::Adaptive& a = ctx->adaptive;
  if (a.build_pending) { ctx->error = "lumc_adaptive_render: a stage build is pending (lumc_adaptive_variance / lumc_adaptive_build_from)"; return 1; }
  while (executions > 0) {
    const uint32_t s = a.stage_id;
    // stage s lasts update_interval << s executions (device_renderer.c:364-371); the last stage never ends
    uint32_t run = executions;
    if (s < kAdaptiveStages) {
      const uint64_t due = (uint64_t) a.params.update_interval << s;
      run = (uint32_t) std::min<uint64_t>(run, due > a.executions[s] ? due - a.executions[s] : 0);
    }
    if (s == 0 && !a.d_block_mask) {
      // one sample id for every pixel per execution: the uniform wavefront pass, several executions per pass
      if (run && lumc_render(ctx, a.executions[0], run, std::min(run, 8u), nullptr, nullptr, stream_)) return 1;
      a.executions[0] += run;
    }
    else {
      // merge executions while a merged pass stays within the usual pass size
      const uint32_t per_execution = std::max(a.task_end.empty() ? 1u : a.task_end.back(), 1u);
      const uint32_t merge_max = std::max(1u, std::min(kAdaptiveTasksPerPass / per_execution, 64u));
      for (uint32_t e = 0; e < run;) {
        const uint32_t merged = std::min(merge_max, run - e);
        if (adaptive_execute(ctx, stream, merged)) return 1;
        e += merged;
      }
    }
    executions -= run;
    if (s < kAdaptiveStages && a.executions[s] >= ((uint64_t) a.params.update_interval << s)) {
      // partitioned: the rates need the block variances of every rank; stop here and let the caller exchange them. The exchange entry
      // points (lumc_adaptive_variance / _build_from) work on the null stream: everything queued on the caller's stream is finished first.
      if (a.d_block_mask) { a.build_pending = true; HIP_TRY(ctx, hipStreamSynchronize(stream)); return 0; }
      if (adaptive_build_stage(ctx, stream)) return 1;
    }
  }
  return 0;
}

int lumc_adaptive_note_first_sample(LumContext* ctx, void* stream_) {
  if (!ctx || !ctx->adaptive.active) { if (ctx) ctx->error = "lumc_adaptive_note_first_sample: call lumc_adaptive_begin first"; return 1; }
  LumContext::Adaptive& a = ctx->adaptive;
  if (a.stage_id != 0 || a.executions[0] != 0) { ctx->error = "lumc_adaptive_note_first_sample: the first execution is already done"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  a.executions[0] = 1;
  if (a.executions[0] >= (uint64_t) a.params.update_interval) {
    if (a.d_block_mask) { a.build_pending = true; HIP_TRY(ctx, hipStreamSynchronize((hipStream_t) stream_)); return 0; }
    if (adaptive_build_stage(ctx, (hipStream_t) stream_)) return 1;
  }
  return 0;
}

int lumc_adaptive_set_partition(LumContext* ctx, const uint8_t* block_mask) {
  if (!ctx || !ctx->adaptive.active || !block_mask) { if (ctx) ctx->error = "lumc_adaptive_set_partition: adaptive mode is not active or null mask"; return 1; }
  LumContext::Adaptive& a = ctx->adaptive;
  for (uint32_t s = 0; s <= kAdaptiveStages; s++)
    if (a.executions[s]) { ctx->error = "lumc_adaptive_set_partition: call it before the first execution"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!a.d_block_mask) HIP_TRY(ctx, hipMalloc((void**) &a.d_block_mask, a.num_blocks));
  HIP_TRY(ctx, hipMemcpy(a.d_block_mask, block_mask, a.num_blocks, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_adaptive_uniform_tasks, dim3((a.num_blocks + 255) / 256), dim3(256), 0, 0, (const uint8_t*) a.d_block_mask, a.num_blocks, a.d_block_tasks);
  HIP_TRY(ctx, hipGetLastError());
  return adaptive_task_prefix(ctx, (hipStream_t) 0);
}

int lumc_adaptive_variance(LumContext* ctx, float* block_variance) {
  if (!ctx || !ctx->adaptive.active || !block_variance) { if (ctx) ctx->error = "lumc_adaptive_variance: adaptive mode is not active or null buffer"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (adaptive_compute_variance(ctx, (hipStream_t) 0)) return 1;
  HIP_TRY(ctx, hipMemcpy(block_variance, ctx->adaptive.d_block_variance, sizeof(float) * ctx->adaptive.num_blocks, hipMemcpyDeviceToHost));
  return 0;
}

int lumc_adaptive_build_from(LumContext* ctx, const float* block_variance) {
  if (!ctx || !ctx->adaptive.active || !block_variance) { if (ctx) ctx->error = "lumc_adaptive_build_from: adaptive mode is not active or null buffer"; return 1; }
  LumContext::Adaptive& a = ctx->adaptive;
  if (a.stage_id >= kAdaptiveStages) { ctx->error = "lumc_adaptive_build_from: the last stage is already running"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemcpy(a.d_block_variance, block_variance, sizeof(float) * a.num_blocks, hipMemcpyHostToDevice));
  return adaptive_finish_build(ctx, (hipStream_t) 0);
}

int lumc_adaptive_info(LumContext* ctx, LumAdaptiveInfo* out) {
  if (!ctx || !out || !ctx->adaptive.active) { if (ctx) ctx->error = "lumc_adaptive_info: adaptive mode is not active"; return 1; }
  const LumContext::Adaptive& a = ctx->adaptive;
  out->stage_id = a.stage_id;
  for (uint32_t s = 0; s <= kAdaptiveStages; s++) out->executions[s] = a.executions[s];
  out->num_blocks = a.num_blocks; out->blocks_x = a.blocks_x; out->blocks_y = a.blocks_y;
  out->tasks_per_execution = a.task_end.empty() ? a.num_blocks * 16u : a.task_end.back();
  out->variance_total = a.variance_total;
  out->build_pending = a.build_pending ? 1u : 0u;
  return 0;
}

int lumc_adaptive_download(LumContext* ctx, uint32_t* stage_counts, float* block_variance) {
  if (!ctx || !ctx->adaptive.active) { if (ctx) ctx->error = "lumc_adaptive_download: adaptive mode is not active"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  const LumContext::Adaptive& a = ctx->adaptive;
  if (stage_counts) HIP_TRY(ctx, hipMemcpy(stage_counts, a.d_stage_counts, sizeof(uint32_t) * a.num_blocks, hipMemcpyDeviceToHost));
  if (block_variance) HIP_TRY(ctx, hipMemcpy(block_variance, a.d_block_variance, sizeof(float) * a.num_blocks, hipMemcpyDeviceToHost));
  return 0;
}

int lumc_adaptive_end(LumContext* ctx) {
  if (!ctx) return 1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  free_adaptive(ctx);
  return 0;
}

int lumc_generate_result(LumContext* ctx, uint32_t mode, uint32_t local_error_minimization, uint32_t uniform_samples, float exposure, const LumOutputParams* tone,
                         float* d_result, void* stream_) {
  const bool framed = ctx && ctx->use_frame && ctx->d_frame && ctx->has_scene && ctx->frame_capacity == ctx->scene.width * ctx->scene.height;
  if (!ctx || !ctx->has_scene || (!framed && (!ctx->d_first_moment || ctx->d_pixels || ctx->num_pixels != ctx->scene.width * ctx->scene.height))) {
    if (ctx) ctx->error = "lumc_generate_result: needs the full-frame accumulators";
    return 1;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t) stream_;
  const uint32_t n = ctx->scene.width * ctx->scene.height;
  const float* src_fm = framed ? ctx->d_frame : ctx->d_first_moment;
  const float* src_sm = framed ? ctx->d_frame + 3 * (size_t) n : ctx->d_second_moment;
  if (!d_result) {
    if (ctx->frame_result_pixels != n) {
      if (ctx->d_frame_result) (void) hipFree(ctx->d_frame_result);
      ctx->d_frame_result = nullptr; ctx->frame_result_pixels = 0;
      HIP_TRY(ctx, hipMalloc((void**) &ctx->d_frame_result, sizeof(float) * 3 * (size_t) n));
      ctx->frame_result_pixels = n;
    }
    d_result = ctx->d_frame_result;
  }
  AdaptiveView view;
  std::memset(&view, 0, sizeof(view));
  if (ctx->adaptive.active) view = adaptive_view(ctx);
  else { view.blocks_x = (ctx->scene.width + 3u) >> kAdaptiveBlockLog; view.blocks_y = (ctx->scene.height + 3u) >> kAdaptiveBlockLog; view.num_blocks = view.blocks_x * view.blocks_y; }
  if (!ctx->adaptive.active && uniform_samples == 0) { ctx->error = "lumc_generate_result: no samples"; return 1; }
  ResultParams rp{ctx->scene.width, ctx->scene.height, mode, local_error_minimization, uniform_samples, exposure};
  const OutputParams op = tone_params(tone);
  {
    Launch l(ctx, stream, LUMC_KERNEL_OUTPUT);
    hipLaunchKernelGGL(k_generate_result, dim3(grid_for(n)), dim3(256), 0, stream, view, rp, op, src_fm, src_sm, d_result);
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int lumc_generate_result_host(LumContext* ctx, uint32_t mode, uint32_t local_error_minimization, uint32_t uniform_samples, float exposure, const LumOutputParams* tone,
                              float* result) {
  if (!ctx || !result) { if (ctx) ctx->error = "lumc_generate_result_host: null argument"; return 1; }
  if (lumc_generate_result(ctx, mode, local_error_minimization, uniform_samples, exposure, tone, nullptr, nullptr)) return 1;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(result, ctx->d_frame_result, sizeof(float) * 3 * (size_t) ctx->scene.width * ctx->scene.height, hipMemcpyDeviceToHost));
  return 0;
}

int lumc_generate_result_undersampled(LumContext* ctx, uint32_t stage, uint32_t iteration, float* d_result, void* stream_) {
  if (!ctx || !ctx->has_scene || !ctx->d_first_moment || ctx->d_pixels || ctx->num_pixels != ctx->scene.width * ctx->scene.height) {
    if (ctx) ctx->error = "lumc_generate_result_undersampled: needs the full-frame accumulators";
    return 1;
  }
  if (stage == 0 || stage > 15 || iteration > 3) { ctx->error = "lumc_generate_result_undersampled: stage in [1, 15], iteration in [0, 3]"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t) stream_;
  const uint32_t n = ctx->num_pixels;
  if (!d_result) {
    if (ctx->frame_result_pixels != n) {
      if (ctx->d_frame_result) (void) hipFree(ctx->d_frame_result);
      ctx->d_frame_result = nullptr; ctx->frame_result_pixels = 0;
      HIP_TRY(ctx, hipMalloc((void**) &ctx->d_frame_result, sizeof(float) * 3 * (size_t) n));
      ctx->frame_result_pixels = n;
    }
    d_result = ctx->d_frame_result;
  }
  const uint32_t compact = (ctx->scene.width >> stage) * (ctx->scene.height >> stage);
  if (compact == 0) return 0;
  {
    Launch l(ctx, stream, LUMC_KERNEL_OUTPUT);
    hipLaunchKernelGGL(k_result_undersampled, dim3(grid_for(compact)), dim3(256), 0, stream, (const float*) ctx->d_first_moment, ctx->scene.width, ctx->scene.height, stage, iteration,
                       d_result);
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int lumc_generate_result_undersampled_host(LumContext* ctx, uint32_t stage, uint32_t iteration, float* result) {
  if (!ctx || !result) { if (ctx) ctx->error = "lumc_generate_result_undersampled_host: null argument"; return 1; }
  if (lumc_generate_result_undersampled(ctx, stage, iteration, nullptr, nullptr)) return 1;
  HIP_TRY(ctx, hipDeviceSynchronize());
  const size_t compact = (size_t) (ctx->scene.width >> stage) * (ctx->scene.height >> stage);
  HIP_TRY(ctx, hipMemcpy(result, ctx->d_frame_result, sizeof(float) * 3 * compact, hipMemcpyDeviceToHost));
  return 0;
}

const float* lumc_result_image(LumContext* ctx) { return ctx ? ctx->d_frame_result : nullptr; }

// _device_post_bloom_apply, device/device_post.c:56-139
int lumc_post_bloom(LumContext* ctx, float* d_image, uint32_t full_width, uint32_t full_height, uint32_t undersampling_stage, float blend, void* stream_) {
  if (!ctx) return 1;
  if (!d_image) d_image = ctx->d_frame_result;
  if (!d_image || full_width == 0 || full_height == 0) { ctx->error = "lumc_post_bloom: no image"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t) stream_;
  uint32_t chain = 0;  // _device_post_bloom_mip_count: floor(log2(min dimension))
  for (uint32_t m = std::min(full_width, full_height); m > 1; m >>= 1) chain++;
  if (undersampling_stage + 1 >= chain) return 0;  // too coarse for a mip chain (device_post.c:62-64)
  if (ctx->bloom_width != full_width || ctx->bloom_height != full_height) {
    for (float* m : ctx->bloom_mips) (void) hipFree(m);
    ctx->bloom_mips.clear(); ctx->bloom_width = ctx->bloom_height = 0;
    for (uint32_t i = 0; i < chain; i++) {
      float* m = nullptr;
      HIP_TRY(ctx, hipMalloc((void**) &m, sizeof(float) * (size_t) (full_width >> (i + 1)) * (full_height >> (i + 1))));
      ctx->bloom_mips.push_back(m);
    }
    ctx->bloom_width = full_width; ctx->bloom_height = full_height;
  }
  const uint32_t width = full_width >> undersampling_stage, height = full_height >> undersampling_stage, mips = chain - undersampling_stage;
  const size_t plane = (size_t) width * height;
  Launch l(ctx, stream, LUMC_KERNEL_OUTPUT);
  for (uint32_t c = 0; c < 3; c++) {
    float* image = d_image + c * plane;
    std::vector<float*>& mip = ctx->bloom_mips;
    hipLaunchKernelGGL(k_post_downsample, dim3(grid_for((width >> 1) * (height >> 1))), dim3(256), 0, stream, (const float*) image, width, height, mip[0], width >> 1, height >> 1);
    for (uint32_t i = 0; i + 1 < mips; i++)
      hipLaunchKernelGGL(k_post_downsample, dim3(grid_for((width >> (i + 2)) * (height >> (i + 2)))), dim3(256), 0, stream, (const float*) mip[i], width >> (i + 1), height >> (i + 1),
                         mip[i + 1], width >> (i + 2), height >> (i + 2));
    for (uint32_t i = mips - 1; i > 0; i--)
      hipLaunchKernelGGL(k_post_upsample, dim3(grid_for((width >> i) * (height >> i))), dim3(256), 0, stream, (const float*) mip[i], width >> (i + 1), height >> (i + 1), mip[i - 1],
                         width >> i, height >> i, 1.0f, 1.0f);
    hipLaunchKernelGGL(k_post_upsample, dim3(grid_for(width * height)), dim3(256), 0, stream, (const float*) mip[0], width >> 1, height >> 1, image, width, height, blend / mips,
                       1.0f - blend);
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int lumc_post_bloom_host(LumContext* ctx, float* image, uint32_t full_width, uint32_t full_height, uint32_t undersampling_stage, float blend) {
  if (!ctx || !image) { if (ctx) ctx->error = "lumc_post_bloom_host: null argument"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t bytes = sizeof(float) * 3 * (size_t) (full_width >> undersampling_stage) * (full_height >> undersampling_stage);
  float* d = nullptr;
  HIP_TRY(ctx, hipMalloc((void**) &d, bytes));
  int rc = 1;
  if (hipMemcpy(d, image, bytes, hipMemcpyHostToDevice) == hipSuccess && lumc_post_bloom(ctx, d, full_width, full_height, undersampling_stage, blend, nullptr) == 0 &&
      hipDeviceSynchronize() == hipSuccess && hipMemcpy(image, d, bytes, hipMemcpyDeviceToHost) == hipSuccess)
    rc = 0;
  else if (ctx->error.empty()) ctx->error = "lumc_post_bloom_host: transfer failed";
  (void) hipFree(d);
  return rc;
}

int lumc_synchronize(LumContext* ctx) {
  if (!ctx) return 1;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  return resolve_stamps(ctx);
}

int lumc_download_accumulators(LumContext* ctx, float* first_moment, float* second_moment) {
  if (!ctx || !ctx->d_first_moment) return 1;
  HIP_TRY(ctx, hipDeviceSynchronize());
  if (first_moment) HIP_TRY(ctx, hipMemcpy(first_moment, ctx->d_first_moment, sizeof(float) * 3 * (size_t) ctx->num_pixels, hipMemcpyDeviceToHost));
  if (second_moment) HIP_TRY(ctx, hipMemcpy(second_moment, ctx->d_second_moment, sizeof(float) * (size_t) ctx->num_pixels, hipMemcpyDeviceToHost));
  return 0;
}

int lumc_counters(LumContext* ctx, uint64_t out[LUMC_CNT_COUNT]) {
  if (!ctx) return 1;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, ctx->d_counters, sizeof(uint64_t) * LUMC_CNT_COUNT, hipMemcpyDeviceToHost));
  return 0;
}
int lumc_reset_counters(LumContext* ctx) {
  if (!ctx) return 1;
  HIP_TRY(ctx, hipMemset(ctx->d_counters, 0, sizeof(uint64_t) * LUMC_CNT_COUNT));
  return 0;
}
int lumc_set_profiling(LumContext* ctx, int enabled) {
  if (!ctx) return 1;
  ctx->profiling = enabled != 0;
  for (int k = 0; k < LUMC_KERNEL_COUNT; k++) { ctx->kernel_ms[k] = 0.0; ctx->kernel_launches[k] = 0; }
  return 0;
}
int lumc_kernel_times(LumContext* ctx, double total_ms[LUMC_KERNEL_COUNT], uint32_t launches[LUMC_KERNEL_COUNT]) {
  if (!ctx) return 1;
  if (lumc_synchronize(ctx)) return 1;
  for (int k = 0; k < LUMC_KERNEL_COUNT; k++) { total_ms[k] = ctx->kernel_ms[k]; launches[k] = ctx->kernel_launches[k]; }
  return 0;
}

extern "C" const unsigned char lum_embedded_bluenoise_1d[];
extern "C" const unsigned char lum_embedded_bluenoise_1d_end[];

int lumc_generate_output(LumContext* ctx, const LumOutputParams* params, const float* d_first_moment, uint32_t* d_argb8, void* stream_) {
  if (!ctx || !params || !d_argb8) { if (ctx) ctx->error = "lumc_generate_output: null argument"; return 1; }
  static_assert(sizeof(LumOutputParams) == sizeof(OutputParams), "output parameter structs must match");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t stream = (hipStream_t) stream_;
  OutputParams p;
  std::memcpy(&p, params, sizeof(p));
  if (p.src_width == 0 || p.src_height == 0 || p.dst_width < 2 || p.dst_height < 2) { ctx->error = "lumc_generate_output: image sizes must be at least 2x2"; return 1; }
  if (p.supersampling > 3 || p.undersampling_stage > 15) { ctx->error = "lumc_generate_output: supersampling at most 3, undersampling stage at most 15"; return 1; }
  const uint32_t ns = p.src_width * p.src_height;
  const uint32_t uo = std::max(p.undersampling_stage, p.supersampling);
  const uint32_t n_out = (p.src_width >> uo) * (p.src_height >> uo);
  if (n_out == 0) { ctx->error = "lumc_generate_output: the frame is smaller than one output pixel"; return 1; }
  if (!d_first_moment) {
    if (p.undersampling_stage) { ctx->error = "lumc_generate_output: an undersampled image must be passed explicitly (lumc_result_image)"; return 1; }
    if (!ctx->d_first_moment || ctx->d_pixels || ctx->num_pixels != ns) { ctx->error = "lumc_generate_output: the context does not hold a full frame of this size"; return 1; }
    d_first_moment = ctx->d_first_moment;
  }
  if (!ctx->d_bluenoise_1d) {
    const size_t bytes = (size_t) (lum_embedded_bluenoise_1d_end - lum_embedded_bluenoise_1d);
    if (bytes != 65536 * sizeof(uint16_t)) { ctx->error = "embedded 1D blue-noise mask has the wrong size"; return 1; }
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_bluenoise_1d, bytes));
    HIP_TRY(ctx, hipMemcpy(ctx->d_bluenoise_1d, lum_embedded_bluenoise_1d, bytes, hipMemcpyHostToDevice));
  }
  if (ctx->frame_output_pixels < ns) {
    if (ctx->d_frame_output) (void) hipFree(ctx->d_frame_output);
    ctx->d_frame_output = nullptr;
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_frame_output, sizeof(float) * 3 * (size_t) ns));
    ctx->frame_output_pixels = ns;
  }
  {
    Launch l(ctx, stream, LUMC_KERNEL_OUTPUT);
    hipLaunchKernelGGL(k_final_image, dim3(grid_for(n_out)), dim3(256), 0, stream, p, d_first_moment, ctx->d_frame_output);
    hipLaunchKernelGGL(k_to_argb8, dim3(grid_for(p.dst_width * p.dst_height)), dim3(256), 0, stream, p, (const float*) ctx->d_frame_output,
                       (const uint16_t*) ctx->d_bluenoise_1d, d_argb8);
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int lumc_generate_output_host(LumContext* ctx, const LumOutputParams* params, const float* d_first_moment, uint32_t* argb8, float* frame_output) {
  if (!ctx || !params || !argb8) { if (ctx) ctx->error = "lumc_generate_output_host: null argument"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint32_t n = params->dst_width * params->dst_height;
  if (ctx->argb8_pixels < n) {
    if (ctx->d_argb8) (void) hipFree(ctx->d_argb8);
    ctx->d_argb8 = nullptr;
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_argb8, sizeof(uint32_t) * (size_t) n));
    ctx->argb8_pixels = n;
  }
  if (lumc_generate_output(ctx, params, d_first_moment, ctx->d_argb8, nullptr)) return 1;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(argb8, ctx->d_argb8, sizeof(uint32_t) * (size_t) n, hipMemcpyDeviceToHost));
  if (frame_output) {
    const uint32_t uo = std::max(params->undersampling_stage, params->supersampling);
    HIP_TRY(ctx, hipMemcpy(frame_output, ctx->d_frame_output, sizeof(float) * 3 * (size_t) (params->src_width >> uo) * (params->src_height >> uo), hipMemcpyDeviceToHost));
  }
  return 0;
}

int lumc_generate_output_from_host(LumContext* ctx, const LumOutputParams* params, const float* first_moment, uint32_t* argb8, float* frame_output) {
  if (!ctx || !params || !first_moment || !argb8) { if (ctx) ctx->error = "lumc_generate_output_from_host: null argument"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t bytes = sizeof(float) * 3 * (size_t) (params->src_width >> params->undersampling_stage) * (params->src_height >> params->undersampling_stage);
  float* d = nullptr;
  HIP_TRY(ctx, hipMalloc((void**) &d, bytes));
  int rc = 1;
  if (hipMemcpy(d, first_moment, bytes, hipMemcpyHostToDevice) == hipSuccess) rc = lumc_generate_output_host(ctx, params, d, argb8, frame_output);
  else ctx->error = "lumc_generate_output_from_host: upload failed";
  (void) hipFree(d);
  return rc;
}

int lumc_trace_closest(LumContext* ctx, uint32_t n, const float* d_origins, const float* d_dirs, const uint32_t* d_ignore, uint32_t* d_out, void* stream_) {
  if (!ctx || !ctx->has_scene) { if (ctx) ctx->error = "lumc_trace_closest: no scene"; return 1; }
  if (n == 0) return 0;
  hipStream_t stream = (hipStream_t) stream_;
  uint32_t* cursor = ctx->d_ctrl + kCtlStride * (kCtrlRows - 1);
  HIP_TRY(ctx, hipMemsetAsync(cursor, 0, sizeof(uint32_t) * 8, stream));  // up to 8 work cursors (dev_trace.h LUM_XCD_RANGES)
  Launch l(ctx, stream, LUMC_KERNEL_TRACE);
  ctx->wf->trace_rays(grid_persistent(ctx, n), (size_t) ctx->lds_nodes * kNodeBytes + LUM_LDS_STACK_BYTES, stream, ctx->scene, n, d_origins, d_dirs, d_ignore, d_out, cursor, ctx->d_counters,
                      ctx->lds_nodes);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int lumc_trace_closest_host(LumContext* ctx, uint32_t n, const float* origins, const float* dirs, const uint32_t* ignore, uint32_t* out) {
  if (!ctx || !ctx->has_scene) { if (ctx) ctx->error = "lumc_trace_closest_host: no scene"; return 1; }
  if (n == 0) return 0;
  float *d_o = nullptr, *d_d = nullptr;
  uint32_t *d_i = nullptr, *d_out = nullptr;
  HIP_TRY(ctx, hipMalloc((void**) &d_o, sizeof(float) * 3 * (size_t) n));
  HIP_TRY(ctx, hipMalloc((void**) &d_d, sizeof(float) * 3 * (size_t) n));
  HIP_TRY(ctx, hipMalloc((void**) &d_out, sizeof(uint32_t) * 3 * (size_t) n));
  HIP_TRY(ctx, hipMemcpy(d_o, origins, sizeof(float) * 3 * (size_t) n, hipMemcpyHostToDevice));
  HIP_TRY(ctx, hipMemcpy(d_d, dirs, sizeof(float) * 3 * (size_t) n, hipMemcpyHostToDevice));
  if (ignore) {
    HIP_TRY(ctx, hipMalloc((void**) &d_i, sizeof(uint32_t) * 2 * (size_t) n));
    HIP_TRY(ctx, hipMemcpy(d_i, ignore, sizeof(uint32_t) * 2 * (size_t) n, hipMemcpyHostToDevice));
  }
  int rc = lumc_trace_closest(ctx, n, d_o, d_d, d_i, d_out, nullptr);
  if (!rc) {
    const hipError_t e = hipMemcpy(out, d_out, sizeof(uint32_t) * 3 * (size_t) n, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { ctx->error = hipGetErrorString(e); rc = 1; }
  }
  (void) hipFree(d_o); (void) hipFree(d_d); (void) hipFree(d_out);
  if (d_i) (void) hipFree(d_i);
  return rc;
}

int lumc_pixel_query(LumContext* ctx, uint32_t x, uint32_t y, uint32_t sample_id, uint32_t out[6]) {
  if (!ctx || !ctx->has_scene || !out) { if (ctx) ctx->error = "lumc_pixel_query: no scene"; return 1; }
  if (x >= ctx->scene.width || y >= ctx->scene.height) { ctx->error = "lumc_pixel_query: pixel outside the frame"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  float* d = nullptr;
  HIP_TRY(ctx, hipMalloc((void**) &d, sizeof(float) * 9));
  hipLaunchKernelGGL(k_pixel_ray, dim3(1), dim3(64), 0, 0, ctx->scene, x, y, sample_id, d, d + 3);
  int rc = lumc_trace_closest(ctx, 1, d, d + 3, nullptr, (uint32_t*) (d + 6), nullptr);
  if (!rc && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out, d + 6, 12, hipMemcpyDeviceToHost) != hipSuccess ||
              hipMemcpy(out + 3, d + 3, 12, hipMemcpyDeviceToHost) != hipSuccess)) { ctx->error = "lumc_pixel_query: device error"; rc = 1; }
  (void) hipFree(d);
  return rc;
}

#ifdef LUM_PHASE_STATS
extern "C" int lumc_debug_phase_stats(uint64_t out[16], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(uint64_t) * 16) != hipSuccess) return 1;
  if (reset) { const uint64_t zero[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), zero, sizeof(zero)) != hipSuccess) return 1; }
  return 0;
}
#endif

// ---- multi-GPU: the image is dealt to the GPUs in 32x32 tiles, every GPU accumulates its own pixels, and ONE reduce per output assembles
// the four moment planes on the display GPU. Replaces the reference's sample partition with host-staged sums
// (device/device_result_interface.c:107-299, at most four devices). Transport: RCCL over xGMI - one communicator rank per context, created
// either per process (lumc_comm_init_rank, launched as one process per GPU) or for all GPUs of one process (lumc_comm_init_all). ----
namespace {
__global__ __launch_bounds__(256) void k_frame_scatter(const float* __restrict__ fm, const float* __restrict__ sm, const uint32_t* __restrict__ pixels, uint32_t n,
                                                       uint32_t frame_pixels, float* __restrict__ frame) {
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const uint32_t index = pixels ? pixels[p] : p;
    if (index >= frame_pixels) continue;
    frame[index] = fm[p]; frame[frame_pixels + index] = fm[n + p]; frame[2u * frame_pixels + index] = fm[2u * n + p];
    frame[3u * frame_pixels + index] = sm[p];
  }
}
__global__ __launch_bounds__(256) void k_frame_add(const float4* __restrict__ src, float4* __restrict__ dst, uint32_t count4) {  // buffer_add, cuda/kernels.cuh:646-675
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count4; i += gridDim.x * 256u) {
    const float4 a = src[i]; float4 b = dst[i];
    b.x += a.x; b.y += a.y; b.z += a.z; b.w += a.w;
    dst[i] = b;
  }
}
#define NCCL_TRY(ctx, expr)                                                                          \
  do {                                                                                               \
    const ncclResult_t r__ = (expr);                                                                 \
    if (r__ != ncclSuccess) { (ctx)->error = std::string(#expr) + " failed: " + ncclGetErrorString(r__); return 1; } \
  } while (0)

// this context's pixels scattered into its zeroed [4][frame_pixels] frame buffer
int frame_scatter(LumContext* ctx, uint32_t frame_pixels, hipStream_t stream) {
  if (!ctx->d_first_moment || ctx->num_pixels == 0) { ctx->error = "lumc_frame_assemble: no accumulators (lumc_set_pixels)"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const uint32_t padded = frame_pixels;  // 4 planes of n floats = n float4s: the add kernel walks the whole buffer, planes keep stride n
  if (ctx->frame_capacity != padded) {
    if (ctx->d_frame) (void) hipFree(ctx->d_frame);
    ctx->d_frame = nullptr; ctx->frame_capacity = 0;
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_frame, sizeof(float) * 4 * (size_t) padded));
    ctx->frame_capacity = padded;
  }
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_frame, 0, sizeof(float) * 4 * (size_t) ctx->frame_capacity, stream));
  hipLaunchKernelGGL(k_frame_scatter, dim3(grid_for(ctx->num_pixels)), dim3(256), 0, stream, (const float*) ctx->d_first_moment, (const float*) ctx->d_second_moment,
                     (const uint32_t*) ctx->d_pixels, ctx->num_pixels, ctx->frame_capacity, ctx->d_frame);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}
}  // namespace

int lumc_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// The tile deal ("block -> GPU by interleaved round-robin for load balance"). Round 5: a rank-1 lattice instead of t % world over the
// row-major grid. The old deal is periodic in x with period `world` tiles whenever the tile row length is a multiple of `world` - at 3840 px (120 tiles)
// and 8 ranks every rank owned vertical 32-pixel stripes. Now tile (x, y) belongs to rank (x + k * y) % world, with k chosen among the steps COPRIME to
// `world` so that a rank's tiles form the most isotropic lattice: k maximises the shortest distance between two tiles of one rank (world 8: k = 3, nearest
// own tiles at (2, 2) and (1, -3); world 2: the checkerboard; world 4 and 6: k = 1, the diagonals). Coprime (round 6, advisor): the row offset k * y then
// runs through every residue, so the tiles a row has beyond a multiple of `world` go to every rank in turn - with k = 2 at 4 ranks (round 5's choice, more
// isotropic) they always went to the same half (1376 x 1080: max / mean share 1.023). Balance bound: over any `world` consecutive tile rows every rank owns
// the same number of tiles; a frame's shares differ by at most (tiles_y % world) tiles (+ the clipped tiles of the right and bottom edge).
// LUM_TILE_DEAL=rowmajor restores t % world (A/B of the load-balance table, profiles/r05_load_balance.json).
uint32_t lumc_tile_lattice_step(uint32_t world) {
  if (world < 2) return 0;
  auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t t = a % b; a = b; b = t; } return a; };
  uint32_t best_k = 1; int64_t best = -1;
  for (uint32_t k = 1; k < world; k++) {
    if (gcd(k, world) != 1u) continue;
    int64_t shortest = INT64_MAX;
    for (int64_t b = -(int64_t) world; b <= (int64_t) world; b++)
      for (int64_t a = -(int64_t) world; a <= (int64_t) world; a++) {
        if ((a == 0 && b == 0) || ((a + (int64_t) k * b) % (int64_t) world) != 0) continue;
        shortest = std::min(shortest, a * a + b * b);
      }
    if (shortest > best) { best = shortest; best_k = k; }
  }
  return best_k;
}

static bool tile_deal_rowmajor() {
  static const bool v = [] { const char* e = std::getenv("LUM_TILE_DEAL"); return e && std::strcmp(e, "rowmajor") == 0; }();
  return v;
}

// the step of a world size, computed once per size (the search is cubic in `world`; the host's render threads call this concurrently)
static uint32_t tile_lattice_step_cached(uint32_t world) {
  static uint32_t step_of[65];
  static std::once_flag once;
  std::call_once(once, [] { for (uint32_t w = 0; w <= 64; w++) step_of[w] = lumc_tile_lattice_step(w); });
  if (world <= 64) return step_of[world];
  static std::mutex m;
  static std::map<uint32_t, uint32_t> beyond;
  std::lock_guard<std::mutex> lock(m);
  auto it = beyond.find(world);
  if (it == beyond.end()) it = beyond.emplace(world, lumc_tile_lattice_step(world)).first;
  return it->second;
}

uint32_t lumc_tile_owner(uint32_t tile_x, uint32_t tile_y, uint32_t tiles_x, uint32_t world) {
  if (world < 2) return 0;
  if (tile_deal_rowmajor()) return (uint32_t) (((uint64_t) tile_y * tiles_x + tile_x) % world);
  return (uint32_t) (((uint64_t) tile_x + (uint64_t) tile_lattice_step_cached(world) * tile_y) % world);
}

// Writes the rank's pixel indices (x + y * width): its tiles in row-major tile order, rows within a tile; `out` may be NULL to query the count.
int lumc_tile_pixels(uint32_t width, uint32_t height, uint32_t rank, uint32_t world, uint32_t tile, uint32_t* out, uint32_t* count) {
  if (!count || world == 0 || rank >= world || tile == 0) return 1;
  const uint32_t tx = (width + tile - 1) / tile, ty = (height + tile - 1) / tile;
  uint32_t n = 0;
  for (uint32_t j = 0; j < ty; j++)
    for (uint32_t i = 0; i < tx; i++) {
      if (lumc_tile_owner(i, j, tx, world) != rank) continue;
      const uint32_t x0 = i * tile, y0 = j * tile;
      for (uint32_t y = y0; y < std::min(y0 + tile, height); y++)
        for (uint32_t x = x0; x < std::min(x0 + tile, width); x++) { if (out) out[n] = x + y * width; n++; }
    }
  *count = n;
  return 0;
}

int lumc_comm_unique_id(uint8_t id[LUMC_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) <= LUMC_COMM_ID_BYTES, "unique id does not fit");
  if (!id) return 1;
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return 1;
  std::memset(id, 0, LUMC_COMM_ID_BYTES);
  std::memcpy(id, &u, sizeof(u));
  return 0;
}

int lumc_comm_init_rank(LumContext* ctx, int world, int rank, const uint8_t id[LUMC_COMM_ID_BYTES]) {
  if (!ctx || !id || world < 1 || rank < 0 || rank >= world) { if (ctx) ctx->error = "lumc_comm_init_rank: bad arguments"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (ctx->comm) { (void) ncclCommDestroy(ctx->comm); ctx->comm = nullptr; }
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  NCCL_TRY(ctx, ncclCommInitRank(&ctx->comm, world, u, rank));
  ctx->comm_rank = rank; ctx->comm_world = world;
  return 0;
}

int lumc_comm_init_all(LumContext** ctxs, int n) {
  if (!ctxs || n < 1) return 1;
  std::vector<int> devices(n);
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return 1;
    devices[i] = ctxs[i]->device;
    for (int j = 0; j < i; j++)
      if (devices[j] == devices[i]) { ctxs[0]->error = "lumc_comm_init_all: two contexts on one device (RCCL needs one GPU per rank)"; return 1; }
  }
  std::vector<ncclComm_t> comms(n, nullptr);
  NCCL_TRY(ctxs[0], ncclCommInitAll(comms.data(), n, devices.data()));
  for (int i = 0; i < n; i++) {
    if (ctxs[i]->comm) (void) ncclCommDestroy(ctxs[i]->comm);
    ctxs[i]->comm = comms[i]; ctxs[i]->comm_rank = i; ctxs[i]->comm_world = n;
  }
  return 0;
}

void lumc_comm_destroy(LumContext* ctx) {
  if (ctx && ctx->comm) { (void) hipSetDevice(ctx->device); (void) ncclCommDestroy(ctx->comm); ctx->comm = nullptr; ctx->comm_world = 1; ctx->comm_rank = 0; }
}

// One process per GPU: this rank's pixels into its frame buffer, then ncclReduce(SUM) to `root` (every pixel has one owner, so the sum is
// a gather: 16 bytes per pixel and rank, once per output). Without a communicator (single GPU) the scatter alone is the frame.
int lumc_frame_assemble(LumContext* ctx, uint32_t frame_pixels, int root, void* stream_, float** d_frame_out) {
  if (!ctx) return 1;
  hipStream_t stream = (hipStream_t) stream_;
  if (frame_scatter(ctx, frame_pixels, stream)) return 1;
  if (ctx->comm) {
    if (root < 0 || root >= ctx->comm_world) { ctx->error = "lumc_frame_assemble: bad root"; return 1; }
    NCCL_TRY(ctx, ncclReduce(ctx->d_frame, ctx->d_frame, 4 * (size_t) ctx->frame_capacity, ncclFloat, ncclSum, root, ctx->comm, stream));
  }
  if (d_frame_out) *d_frame_out = ctx->d_frame;
  return 0;
}

// One process, several GPUs: all contexts scatter, then one grouped ncclReduce to `root` (lumc_comm_init_all). Contexts without a common
// communicator (RCCL unavailable, or test set-ups with two contexts on one device) are summed through peer copies and the add kernel
// instead - the reference's own transport (device_result_interface.c:177-215), kept as the fallback.
int lumc_frame_assemble_all(LumContext** ctxs, int n, uint32_t frame_pixels, int root, float** d_frame_root) {
  if (!ctxs || n < 1 || root < 0 || root >= n) return 1;
  bool rccl = n > 1;
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return 1;
    if (frame_scatter(ctxs[i], frame_pixels, (hipStream_t) 0)) { if (i) ctxs[0]->error = ctxs[i]->error; return 1; }
    rccl = rccl && ctxs[i]->comm && ctxs[i]->comm_world == n && ctxs[i]->comm_rank == i;
  }
  LumContext* r = ctxs[root];
  if (rccl) {
    NCCL_TRY(r, ncclGroupStart());
    for (int i = 0; i < n; i++) {
      (void) hipSetDevice(ctxs[i]->device);
      const ncclResult_t e = ncclReduce(ctxs[i]->d_frame, ctxs[i]->d_frame, 4 * (size_t) ctxs[i]->frame_capacity, ncclFloat, ncclSum, root, ctxs[i]->comm, (hipStream_t) 0);
      if (e != ncclSuccess) { (void) ncclGroupEnd(); r->error = std::string("ncclReduce failed: ") + ncclGetErrorString(e); return 1; }
    }
    NCCL_TRY(r, ncclGroupEnd());
    for (int i = 0; i < n; i++) { HIP_TRY(r, hipSetDevice(ctxs[i]->device)); HIP_TRY(r, hipDeviceSynchronize()); }
  }
  else if (n > 1) {
    float* staging = nullptr;
    HIP_TRY(r, hipSetDevice(r->device));
    const size_t bytes = sizeof(float) * 4 * (size_t) r->frame_capacity;
    HIP_TRY(r, hipMalloc((void**) &staging, bytes));
    for (int i = 0; i < n; i++) {
      if (i == root) continue;
      HIP_TRY(r, hipSetDevice(ctxs[i]->device));
      HIP_TRY(r, hipDeviceSynchronize());
      HIP_TRY(r, hipSetDevice(r->device));
      HIP_TRY(r, hipMemcpyPeer(staging, r->device, ctxs[i]->d_frame, ctxs[i]->device, bytes));
      hipLaunchKernelGGL(k_frame_add, dim3(grid_for(r->frame_capacity)), dim3(256), 0, 0, (const float4*) staging, (float4*) r->d_frame, r->frame_capacity);
      HIP_TRY(r, hipGetLastError());
    }
    HIP_TRY(r, hipDeviceSynchronize());
    (void) hipFree(staging);
  }
  if (d_frame_root) *d_frame_root = r->d_frame;
  return 0;
}

// ---- tile gather: the frame assembled from the ranks' own pixels instead of a reduce over whole frames ----
// Every pixel has one owner, so summing the ranks' zero-padded full frames (lumc_frame_assemble: 16 bytes per FRAME pixel from every rank, 133 MB per
// rank at 4K) moves `world` times what is needed: a rank's contribution is the 16 bytes of each pixel it OWNS. Where the ranks' pixel sets are the tile
// deal of lumc_tile_pixels (32 x 32 tiles dealt by lumc_tile_owner's lattice - what bench.py and the host API's tiled render loop use), every rank can compute every other
// rank's pixel list, so nothing but the sums travels: each rank packs its [3][P] + [P] accumulators into a [4][M] buffer (M = the largest tile share,
// zero padded: the deal's shares differ by at most tiles_y % world tiles, see lumc_tile_lattice_step), ONE ncclGather brings the `world` buffers to the root, and a scatter kernel on the root puts every
// value at its pixel. Reference: device_result_interface.c:107-299 (sample partition, sums staged through pinned host memory).
namespace {
__global__ __launch_bounds__(256) void k_gather_pack(const float* __restrict__ fm, const float* __restrict__ sm, uint32_t n, uint32_t stride, float* __restrict__ send) {
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < stride; p += gridDim.x * 256u) {
    const bool in = p < n;
    send[p] = in ? fm[p] : 0.0f; send[stride + p] = in ? fm[n + p] : 0.0f; send[2u * stride + p] = in ? fm[2u * n + p] : 0.0f; send[3u * stride + p] = in ? sm[p] : 0.0f;
  }
}
__global__ __launch_bounds__(256) void k_gather_unpack(const float* __restrict__ recv, const uint32_t* __restrict__ pixels, uint32_t world, uint32_t stride, uint32_t frame_pixels,
                                                       float* __restrict__ frame) {
  const uint32_t total = world * stride;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const uint32_t index = pixels[i];
    if (index >= frame_pixels) continue;  // padding
    const uint32_t r = i / stride, p = i - r * stride;
    const float* src = recv + (size_t) r * 4u * stride;
    frame[index] = src[p]; frame[frame_pixels + index] = src[stride + p]; frame[2u * frame_pixels + index] = src[2u * stride + p]; frame[3u * frame_pixels + index] = src[3u * stride + p];
  }
}

constexpr uint32_t kGatherTile = 32u;  // the deal bench.py, luminary_amd/distributed.py and the host API use

// Sizes this context's gather buffers for (width, height, world): the send buffer on every rank, the receive buffer and the pixel lists on the root.
// Fails when this context's pixel count is not its share of the deal (the gather is only for the standard deal; anything else reduces).
int gather_prepare(LumContext* ctx, uint32_t width, uint32_t height, int world, int rank, bool is_root) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  uint32_t share = 0, stride = 0;
  for (int r = 0; r < world; r++) {
    uint32_t c = 0;
    if (lumc_tile_pixels(width, height, (uint32_t) r, (uint32_t) world, kGatherTile, nullptr, &c)) { ctx->error = "lumc_frame_gather: bad deal"; return 1; }
    if (r == rank) share = c;
    stride = std::max(stride, c);
  }
  bool is_share = ctx->d_first_moment && ctx->num_pixels == share;
  if (is_share) {  // ... and the same pixels in the same order: the root scatters the rank's sums through the list IT derives from the deal
    std::vector<uint32_t> mine(share ? share : 1);
    uint32_t c = 0;
    (void) lumc_tile_pixels(width, height, (uint32_t) rank, (uint32_t) world, kGatherTile, mine.data(), &c);
    is_share = pixel_list_hash(mine.data(), share) == ctx->pixels_hash;
  }
  if (!is_share) {
    ctx->error = "lumc_frame_gather: this context's pixel set is not its share of the 32x32 tile deal, in the deal's order (use lumc_frame_assemble for other partitions)";
    return 1;
  }
  stride = (stride + 3u) & ~3u;
  const bool same = ctx->gather_key[0] == width && ctx->gather_key[1] == height && ctx->gather_key[2] == (uint32_t) world && ctx->gather_stride == stride && ctx->d_gather_send;
  if (!same) {
    if (ctx->d_gather_send) (void) hipFree(ctx->d_gather_send);
    if (ctx->d_gather_pixels) (void) hipFree(ctx->d_gather_pixels);
    ctx->d_gather_send = nullptr; ctx->d_gather_pixels = nullptr;
    HIP_TRY(ctx, hipMalloc((void**) &ctx->d_gather_send, sizeof(float) * 4 * (size_t) stride));
    ctx->gather_stride = stride; ctx->gather_key[0] = width; ctx->gather_key[1] = height; ctx->gather_key[2] = (uint32_t) world;
  }
  if (is_root) {
    const size_t need = (size_t) world * 4 * stride;
    if (ctx->gather_recv_floats < need) {
      if (ctx->d_gather_recv) (void) hipFree(ctx->d_gather_recv);
      ctx->d_gather_recv = nullptr; ctx->gather_recv_floats = 0;
      HIP_TRY(ctx, hipMalloc((void**) &ctx->d_gather_recv, sizeof(float) * need));
      ctx->gather_recv_floats = need;
    }
    if (!ctx->d_gather_pixels) {
      std::vector<uint32_t> lists((size_t) world * stride, 0xFFFFFFFFu);
      for (int r = 0; r < world; r++) { uint32_t c = 0; (void) lumc_tile_pixels(width, height, (uint32_t) r, (uint32_t) world, kGatherTile, lists.data() + (size_t) r * stride, &c); }
      HIP_TRY(ctx, hipMalloc((void**) &ctx->d_gather_pixels, sizeof(uint32_t) * lists.size()));
      HIP_TRY(ctx, hipMemcpy(ctx->d_gather_pixels, lists.data(), sizeof(uint32_t) * lists.size(), hipMemcpyHostToDevice));
    }
    const uint32_t frame_pixels = width * height;
    if (ctx->frame_capacity != frame_pixels) {
      if (ctx->d_frame) (void) hipFree(ctx->d_frame);
      ctx->d_frame = nullptr; ctx->frame_capacity = 0;
      HIP_TRY(ctx, hipMalloc((void**) &ctx->d_frame, sizeof(float) * 4 * (size_t) frame_pixels));
      ctx->frame_capacity = frame_pixels;
    }
  }
  return 0;
}
int gather_pack(LumContext* ctx, hipStream_t stream) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_gather_pack, dim3(grid_for(ctx->gather_stride)), dim3(256), 0, stream, (const float*) ctx->d_first_moment, (const float*) ctx->d_second_moment, ctx->num_pixels,
                     ctx->gather_stride, ctx->d_gather_send);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}
int gather_unpack(LumContext* root, int world, hipStream_t stream) {
  HIP_TRY(root, hipSetDevice(root->device));
  hipLaunchKernelGGL(k_gather_unpack, dim3(grid_for((uint32_t) world * root->gather_stride)), dim3(256), 0, stream, (const float*) root->d_gather_recv, (const uint32_t*) root->d_gather_pixels,
                     (uint32_t) world, root->gather_stride, root->frame_capacity, root->d_frame);
  HIP_TRY(root, hipGetLastError());
  return 0;
}
}  // namespace

// One process per GPU (lumc_comm_init_rank): pack, one ncclGather to `root`, scatter on the root. Without a communicator (one rank) the pack is copied.
int lumc_frame_gather(LumContext* ctx, uint32_t width, uint32_t height, int root, void* stream_, float** d_frame_out) {
  if (!ctx) return 1;
  hipStream_t stream = (hipStream_t) stream_;
  const int world = ctx->comm ? ctx->comm_world : 1, rank = ctx->comm ? ctx->comm_rank : 0;
  if (root < 0 || root >= world) { ctx->error = "lumc_frame_gather: bad root"; return 1; }
  if (gather_prepare(ctx, width, height, world, rank, rank == root)) return 1;
  if (gather_pack(ctx, stream)) return 1;
  const size_t count = 4 * (size_t) ctx->gather_stride;
  if (ctx->comm) NCCL_TRY(ctx, ncclGather(ctx->d_gather_send, rank == root ? ctx->d_gather_recv : nullptr, count, ncclFloat, root, ctx->comm, stream));
  else HIP_TRY(ctx, hipMemcpyAsync(ctx->d_gather_recv, ctx->d_gather_send, sizeof(float) * count, hipMemcpyDeviceToDevice, stream));
  if (rank == root && gather_unpack(ctx, world, stream)) return 1;
  if (d_frame_out) *d_frame_out = rank == root ? ctx->d_frame : nullptr;
  return 0;
}

// One process, several GPUs (ctxs[i] holds share i of the deal over n): a grouped ncclGather when the contexts share a communicator
// (lumc_comm_init_all), peer copies of the packed buffers into the root's receive buffer otherwise.
int lumc_frame_gather_all(LumContext** ctxs, int n, uint32_t width, uint32_t height, int root, float** d_frame_root) {
  if (!ctxs || n < 1 || root < 0 || root >= n) return 1;
  bool rccl = n > 1;
  for (int i = 0; i < n; i++) {
    if (!ctxs[i]) return 1;
    if (gather_prepare(ctxs[i], width, height, n, i, i == root) || gather_pack(ctxs[i], (hipStream_t) 0)) { if (i) ctxs[0]->error = ctxs[i]->error; return 1; }
    rccl = rccl && ctxs[i]->comm && ctxs[i]->comm_world == n && ctxs[i]->comm_rank == i;
  }
  LumContext* r = ctxs[root];
  const size_t count = 4 * (size_t) r->gather_stride;
  if (rccl) {
    NCCL_TRY(r, ncclGroupStart());
    for (int i = 0; i < n; i++) {
      (void) hipSetDevice(ctxs[i]->device);
      const ncclResult_t e = ncclGather(ctxs[i]->d_gather_send, i == root ? r->d_gather_recv : nullptr, count, ncclFloat, root, ctxs[i]->comm, (hipStream_t) 0);
      if (e != ncclSuccess) { (void) ncclGroupEnd(); r->error = std::string("ncclGather failed: ") + ncclGetErrorString(e); return 1; }
    }
    NCCL_TRY(r, ncclGroupEnd());
    for (int i = 0; i < n; i++) { HIP_TRY(r, hipSetDevice(ctxs[i]->device)); HIP_TRY(r, hipDeviceSynchronize()); }
  }
  else {
    for (int i = 0; i < n; i++) {
      HIP_TRY(r, hipSetDevice(ctxs[i]->device));
      HIP_TRY(r, hipDeviceSynchronize());
      HIP_TRY(r, hipSetDevice(r->device));
      HIP_TRY(r, hipMemcpyPeer(r->d_gather_recv + (size_t) i * count, r->device, ctxs[i]->d_gather_send, ctxs[i]->device, sizeof(float) * count));
    }
  }
  if (gather_unpack(r, n, (hipStream_t) 0)) return 1;
  HIP_TRY(r, hipDeviceSynchronize());
  if (d_frame_root) *d_frame_root = r->d_frame;
  return 0;
}

// ---- one process, several GPUs: what the tiled render loop of the host API needs beyond the frame assembly ----
namespace {
// this context's accumulators <- the frame's values at its pixels; with an adaptive partition only inside the blocks it owns
__global__ __launch_bounds__(256) void k_accumulators_from_frame(const float* __restrict__ frame, uint32_t frame_pixels, const uint32_t* __restrict__ pixels, uint32_t n,
                                                                 const uint8_t* __restrict__ block_mask, uint32_t width, uint32_t blocks_x, float* __restrict__ fm, float* __restrict__ sm) {
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n; p += gridDim.x * 256u) {
    const uint32_t index = pixels ? pixels[p] : p;
    bool mine = index < frame_pixels;
    if (mine && block_mask) { const uint32_t y = index / width, x = index - y * width; mine = block_mask[(y >> 2) * blocks_x + (x >> 2)] != 0; }
    fm[p] = mine ? frame[index] : 0.0f; fm[n + p] = mine ? frame[frame_pixels + index] : 0.0f; fm[2u * n + p] = mine ? frame[2u * frame_pixels + index] : 0.0f;
    sm[p] = mine ? frame[3u * frame_pixels + index] : 0.0f;
  }
}
}  // namespace

// The accumulators of `dst` (whatever its pixel set: a tile list, or the full frame with an adaptive partition) take the values the frame buffer of
// `src` holds at dst's pixels (lumc_frame_assemble on src first: its own full-frame accumulators, scattered). This is how the first sample of a
// frame, rendered coarse to fine on the main device alone (the undersampling preview, device.c:392-420), is handed to the devices that go on with
// the frame's tiles: every pixel's sums continue where the preview left them, so the tiled frame equals the single-device frame bit for bit.
int lumc_accumulators_from_frame(LumContext* dst, LumContext* src) {
  if (!dst || !src || !src->d_frame || !dst->d_first_moment || dst->num_pixels == 0) { if (dst) dst->error = "lumc_accumulators_from_frame: no frame on the source or no accumulators on the destination"; return 1; }
  const uint32_t frame_pixels = src->frame_capacity;
  const float* frame = src->d_frame;
  float* staging = nullptr;
  HIP_TRY(dst, hipSetDevice(src->device));
  HIP_TRY(dst, hipDeviceSynchronize());
  HIP_TRY(dst, hipSetDevice(dst->device));
  if (dst != src) {  // another context (another GPU, or the same one in test set-ups): a copy of the frame on dst's device
    const size_t bytes = sizeof(float) * 4 * (size_t) frame_pixels;
    HIP_TRY(dst, hipMalloc((void**) &staging, bytes));
    HIP_TRY(dst, hipMemcpyPeer(staging, dst->device, src->d_frame, src->device, bytes));
    frame = staging;
  }
  const LumContext::Adaptive& a = dst->adaptive;
  hipLaunchKernelGGL(k_accumulators_from_frame, dim3(grid_for(dst->num_pixels)), dim3(256), 0, 0, frame, frame_pixels, (const uint32_t*) dst->d_pixels, dst->num_pixels,
                     a.active ? (const uint8_t*) a.d_block_mask : nullptr, dst->scene.width, a.active ? a.blocks_x : 0u, dst->d_first_moment, dst->d_second_moment);
  HIP_TRY(dst, hipGetLastError());
  HIP_TRY(dst, hipDeviceSynchronize());
  if (staging) (void) hipFree(staging);
  return 0;
}

// A stage build of adaptive rendering tiled over the contexts of one process (lumc_adaptive_set_partition on each): every context computes the
// variances of its blocks, ONE all-reduce of 4 bytes per block makes the array complete everywhere (every block has one owner: the sum is a gather and
// exact), every context derives the same rates. Grouped ncclAllReduce when the contexts share a communicator (lumc_comm_init_all); otherwise - two
// contexts on one device in tests, or no RCCL - the arrays are summed on the host in context order.
int lumc_adaptive_exchange_all(LumContext** ctxs, int n) {
  if (!ctxs || n < 1) return 1;
  bool rccl = n > 1;
  for (int i = 0; i < n; i++) {
    if (!ctxs[i] || !ctxs[i]->adaptive.active) { if (ctxs[0]) ctxs[0]->error = "lumc_adaptive_exchange_all: adaptive mode is not active on every context"; return 1; }
    if (ctxs[i]->adaptive.num_blocks != ctxs[0]->adaptive.num_blocks) { ctxs[0]->error = "lumc_adaptive_exchange_all: contexts of different frames"; return 1; }
    HIP_TRY(ctxs[0], hipSetDevice(ctxs[i]->device));
    if (adaptive_compute_variance(ctxs[i], (hipStream_t) 0)) { ctxs[0]->error = ctxs[i]->error; return 1; }
    rccl = rccl && ctxs[i]->comm && ctxs[i]->comm_world == n && ctxs[i]->comm_rank == i;
  }
  const uint32_t nb = ctxs[0]->adaptive.num_blocks;
  if (rccl) {
    NCCL_TRY(ctxs[0], ncclGroupStart());
    for (int i = 0; i < n; i++) {
      (void) hipSetDevice(ctxs[i]->device);
      const ncclResult_t e = ncclAllReduce(ctxs[i]->adaptive.d_block_variance, ctxs[i]->adaptive.d_block_variance, nb, ncclFloat, ncclSum, ctxs[i]->comm, (hipStream_t) 0);
      if (e != ncclSuccess) { (void) ncclGroupEnd(); ctxs[0]->error = std::string("ncclAllReduce failed: ") + ncclGetErrorString(e); return 1; }
    }
    NCCL_TRY(ctxs[0], ncclGroupEnd());
  }
  else if (n > 1) {
    std::vector<float> sum(nb, 0.0f), part(nb);
    for (int i = 0; i < n; i++) {
      HIP_TRY(ctxs[0], hipSetDevice(ctxs[i]->device));
      HIP_TRY(ctxs[0], hipMemcpy(part.data(), ctxs[i]->adaptive.d_block_variance, sizeof(float) * nb, hipMemcpyDeviceToHost));
      for (uint32_t b = 0; b < nb; b++) sum[b] += part[b];
    }
    for (int i = 0; i < n; i++) {
      HIP_TRY(ctxs[0], hipSetDevice(ctxs[i]->device));
      HIP_TRY(ctxs[0], hipMemcpy(ctxs[i]->adaptive.d_block_variance, sum.data(), sizeof(float) * nb, hipMemcpyHostToDevice));
    }
  }
  for (int i = 0; i < n; i++) {
    HIP_TRY(ctxs[0], hipSetDevice(ctxs[i]->device));
    if (ctxs[i]->adaptive.stage_id >= kAdaptiveStages) { ctxs[0]->error = "lumc_adaptive_exchange_all: the last stage is already running"; return 1; }
    if (adaptive_finish_build(ctxs[i], (hipStream_t) 0)) { ctxs[0]->error = ctxs[i]->error; return 1; }
  }
  return 0;
}
// Ranks of the communicator this context belongs to (1 without one): what a launcher prints to show that RCCL saw every GPU.
int lumc_comm_count(const LumContext* ctx) {
  if (!ctx || !ctx->comm) return 1;
  int count = 1;
  return ncclCommCount(ctx->comm, &count) == ncclSuccess ? count : 1;
}

// The assembled frame of this context (valid on the root after lumc_frame_assemble*): planar first moment [3][frame_pixels] and second moment.
int lumc_frame_download(LumContext* ctx, uint32_t frame_pixels, float* first_moment, float* second_moment) {
  if (!ctx || !ctx->d_frame || frame_pixels > ctx->frame_capacity) { if (ctx) ctx->error = "lumc_frame_download: no assembled frame"; return 1; }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  const size_t cap = ctx->frame_capacity;
  if (first_moment)
    for (int c = 0; c < 3; c++) HIP_TRY(ctx, hipMemcpy(first_moment + (size_t) c * frame_pixels, ctx->d_frame + (size_t) c * cap, sizeof(float) * frame_pixels, hipMemcpyDeviceToHost));
  if (second_moment) HIP_TRY(ctx, hipMemcpy(second_moment, ctx->d_frame + 3 * cap, sizeof(float) * frame_pixels, hipMemcpyDeviceToHost));
  return 0;
}
uint32_t lumc_frame_plane_stride(const LumContext* ctx) { return ctx ? ctx->frame_capacity : 0; }
// The display entry points of this context (lumc_generate_result*, and through them the output chain) read the assembled full frame instead
// of the context's own accumulators: what the display GPU of a tiled render shows.
int lumc_use_assembled_frame(LumContext* ctx, int on) {
  if (!ctx) return 1;
  if (on && (!ctx->d_frame || !ctx->has_scene || ctx->frame_capacity != ctx->scene.width * ctx->scene.height)) { ctx->error = "lumc_use_assembled_frame: no assembled frame of this scene's size"; return 1; }
  ctx->use_frame = on != 0;
  return 0;
}
int lumc_device_name(int ordinal, char* out, size_t size) {
  if (!out || size == 0) return 1;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) { out[0] = 0; return 1; }
  std::snprintf(out, size, "%s", prop.name);
  return 0;
}

int lumc_set_ray_sorting(LumContext* ctx, int mode) {
  if (!ctx || mode < 0 || mode > 3) { if (ctx) ctx->error = "lumc_set_ray_sorting: 0 (queue order), 1 (closest-hit rays sorted), 2 (visibility rays too), 3 (path queue physically reordered)"; return 1; }
  ctx->sort_mode = mode;
  return 0;
}
int lumc_get_ray_sorting(const LumContext* ctx) { return ctx ? ctx->sort_mode : 0; }

int lumc_set_flavour(LumContext* ctx, int flavour) {
  if (!ctx || flavour < 0 || flavour > 1) { if (ctx) ctx->error = "lumc_set_flavour: 0 (exact) or 1 (fast)"; return 1; }
  ctx->wf = flavour == LUMC_FLAVOUR_FAST ? wavefront_kernels_fast() : wavefront_kernels_exact();
  return 0;
}
int lumc_set_fused_resolve(LumContext* ctx, int on) {
  if (!ctx) return 1;
  ctx->fused_resolve = on != 0 ? 1 : 0;
  ctx->fused_ended = on == 2 ? 0 : ctx->fused_ended_default;  // 2: the vertices whose path ended keep their own kernel (k_resolve_ended) - for comparison; 1: the context's default (LUM_FUSED_ENDED, LUM_SHADE_DYNAMIC)
  return 0;
}
int lumc_set_sobol_table(LumContext* ctx, int on) {
  if (!ctx) return 1;
  ctx->sobol_table = on != 0 ? 1 : 0;
  return 0;
}
int lumc_set_ambient_reuse(LumContext* ctx, int mode) {
  if (!ctx || mode < -1 || mode > 1) { if (ctx) ctx->error = "lumc_set_ambient_reuse: -1 (by flavour), 0 (off) or 1 (on)"; return 1; }
  ctx->ambient_reuse = mode;
  return 0;
}
int lumc_get_ambient_reuse(const LumContext* ctx) { return (ctx && ambient_reuse_active(ctx)) ? 1 : 0; }
int lumc_get_flavour(const LumContext* ctx) { return (ctx && ctx->wf == wavefront_kernels_exact()) ? LUMC_FLAVOUR_EXACT : LUMC_FLAVOUR_FAST; }

unsigned int lumc_lds_stack_bytes(void) { return LUM_LDS_STACK_BYTES; }

int lumc_set_bvh_builder(LumContext* ctx, int builder) {
  if (!ctx || builder < 0 || builder > 3) { if (ctx) ctx->error = "lumc_set_bvh_builder: 0 (SAH, host), 1 (LBVH, GPU), 2 (PLOC, GPU) or 3 (SAH, GPU)"; return 1; }
  ctx->bvh_builder = builder;
  return 0;
}
double lumc_bvh_build_seconds(const LumContext* ctx) { return ctx ? ctx->bvh_build_seconds : 0.0; }
int lumc_bvh_meshes_by_builder(const LumContext* ctx, uint32_t out[2]) {
  if (!ctx || !out) return 1;
  out[0] = ctx->bvh_meshes_by_builder[0]; out[1] = ctx->bvh_meshes_by_builder[1];
  return 0;
}

int lumc_bvh_stats(LumContext* ctx, uint64_t out[4]) {
  if (!ctx) return 1;
  for (int k = 0; k < 4; k++) out[k] = ctx->bvh_stats[k];
  return 0;
}

}  // extern "C"
