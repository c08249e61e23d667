// Implementation of the Luminary C host API (include/luminary_amd.h) on top of the host scene store and the HIP core.
// Reference behaviour: src/luminary/host/host.c (function by function, cited below), src/luminary/luminary.c:7-31,
// src/luminary/path.c, src/luminary/error.c. The reference runs these calls through a Host queue thread and a Device queue
// thread. Here scene edits are applied on the caller's thread under the host's mutex, and rendering happens either
//   * asynchronously, as in the reference: luminary_host_start_new_render starts the host's "Device" worker thread, which renders sample
//     allocation after sample allocation (restarting by itself whenever an edit dirties the integration) and produces the outputs that
//     are due, while the frontend only polls luminary_host_try_await_output / luminary_host_acquire_output (mandarin_duck.c:140-244); or
//   * synchronously inside the additive luminary_ext_render* calls (tests, batch tools), when no render was started.
// (DESIGN.md "Threading".)
#include <atomic>
#include <cmath>
#include <cfloat>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <type_traits>
#include <string>
#include <vector>

#include "../../../include/lum_core.h"
#include "../../../include/luminary_amd.h"
#include "loaders.h"
#include "scene.h"

extern "C" const unsigned char lum_embedded_bluenoise_2d[];
extern "C" const unsigned char lum_embedded_bluenoise_2d_end[];

#include "output.h"

struct LuminaryPath { std::string value; };

struct LuminaryHost {
  lum::HostScene scene;
  lum::DeviceSceneBuffers device_scene;
  // Which parts of the scene edits have touched since the device scene / a device's copy of it was last brought up to date (LUMC_DIRTY_*; the
  // reference keeps such flags per entity, scene.h:42-63, and its device manager uploads by them, device_manager.c:311-320, :424-450): a camera
  // move re-encodes and re-uploads nothing but constants, a material edit the material array, an instance edit the top-level tree.
  uint32_t scene_dirty = LUMC_DIRTY_ALL;  // parts of `device_scene` that are out of date
  uint32_t core_dirty = LUMC_DIRTY_ALL;   // parts the main device's context has not taken over yet
  bool device_scene_valid = false;
  bool core_scene_valid = false;
  uint32_t core_width = 0, core_height = 0;  // frame size the main context's pixel set was made for
  LumContext* core = nullptr;        // context of the main device: renders (its tiles) and produces every output
  int device_ordinal = 0;            // HIP ordinal of the main device
  // every visible device (device_manager.c:776-862 creates one Device per masked id); the frame is tiled over the enabled ones
  struct DeviceSlot { int ordinal = 0; bool enabled = false; LumContext* core = nullptr; bool scene_valid = false; uint32_t dirty = LUMC_DIRTY_ALL; };
  std::vector<DeviceSlot> devices;
  uint32_t main_slot = 0;
  uint32_t partition_n = 0;          // devices the current accumulation is tiled over (0 or 1: the main device renders every pixel)
  bool comm_ready = false;           // the slots of the current partition share an RCCL communicator
  bool handoff_preview = false;      // the frame's first sample sits on the main device (undersampling preview): the tiles take it over when they are dealt
  std::vector<uint32_t> pixels;  // pixel set of the current accumulation
  bool pixels_all = true;
  uint32_t num_pixels = 0;
  uint32_t accumulated_samples = 0;  // uniform rendering: samples per pixel; adaptive rendering: executions (the reference's sample count)
  bool adaptive_active = false;      // the accumulation is driven by lumc_adaptive_* (luminary_ext_render with adaptive sampling enabled)
  bool hdri_origin_pending = true;   // the next scene build re-bakes the sky panorama from the camera's position (SCENE_DIRTY_FLAG_HDRI)
  lum::OutputStore outputs;
  double render_seconds = 0.0;
  double last_sample_ms = 0.0;  // wall time of the most recent render chunk per sample allocation, in milliseconds (device_sampletime.c)
  std::vector<std::string> log;
  std::mutex mutex;
  // ---- asynchronous rendering: the "Device" queue worker (device_manager.c:127-148, :874) ----
  std::thread worker;
  std::mutex worker_mutex;             // guards the four flags below
  std::condition_variable worker_cv;
  bool worker_started = false, worker_stop = false;
  bool async_active = false;           // a render was started and not stopped
  bool async_failed = false;           // the last iteration failed: wait for the next edit or start instead of spinning
  std::atomic<int> api_waiting{0};     // callers waiting for `mutex`: the worker lets them in between two iterations
  LuminaryThreadStatus* status_host = nullptr;    // queue worker 0 "Host" (edits are applied on the caller's thread: always idle)
  LuminaryThreadStatus* status_device = nullptr;  // queue worker 1 "Device"
};

// Lock of the host's mutex taken by API calls: announces itself so that the render worker, which would otherwise re-take the mutex at
// once, yields between two iterations (std::mutex makes no fairness promise).
struct ApiLock {
  LuminaryHost* h;
  explicit ApiLock(LuminaryHost* host) : h(host) { h->api_waiting.fetch_add(1); h->mutex.lock(); h->api_waiting.fetch_sub(1); }
  ~ApiLock() { h->mutex.unlock(); }
  ApiLock(const ApiLock&) = delete;
  ApiLock& operator=(const ApiLock&) = delete;
};

namespace {

std::vector<uint32_t> embedded_bluenoise() {
  const size_t bytes = (size_t) (lum_embedded_bluenoise_2d_end - lum_embedded_bluenoise_2d);
  std::vector<uint32_t> v(bytes / 4);
  std::memcpy(v.data(), lum_embedded_bluenoise_2d, v.size() * 4);
  return v;
}

void invalidate(LuminaryHost* h, uint32_t dirty = LUMC_DIRTY_ALL) {
  h->scene_dirty |= dirty; h->core_dirty |= dirty;
  // An adaptive accumulation ends with the edit. The context leaves adaptive mode in lumc_set_pixels only, and an unchanged frame size would keep the
  // pixel set (ensure_core): without this, a uniform render after `enable_adaptive_sampling = false` ran on a context whose result image still
  // normalised by the stale per-block sample counts.
  if (h->adaptive_active) h->num_pixels = 0;
  h->device_scene_valid = false; h->core_scene_valid = false; h->accumulated_samples = 0; h->adaptive_active = false;
  for (auto& slot : h->devices) { slot.scene_valid = false; slot.dirty |= dirty; }
  { std::lock_guard<std::mutex> l(h->worker_mutex); h->async_failed = false; }  // the edit may have repaired what failed
  h->worker_cv.notify_all();
}

#define CHECK_NULL(p) do { if (!(p)) return LUMINARY_ERROR_ARGUMENT_NULL; } while (0)

// Does a change of the camera / the renderer settings restart the integration (SCENE_DIRTY_FLAG_INTEGRATION), or does it only change how
// the accumulated frame is shown (SCENE_DIRTY_FLAG_OUTPUT)? camera_check_for_dirty (camera.c:80-147), settings_check_for_dirty
// (settings.c:45-72); checked against both in tests/test_reference_host.py.
bool camera_change_restarts(const LuminaryCamera& in, const LuminaryCamera& old) {
  bool d = in.pos.x != old.pos.x || in.pos.y != old.pos.y || in.pos.z != old.pos.z || in.rotation.x != old.rotation.x || in.rotation.y != old.rotation.y ||
           in.rotation.z != old.rotation.z || in.russian_roulette_threshold != old.russian_roulette_threshold || in.camera_scale != old.camera_scale ||
           in.object_distance != old.object_distance || in.use_physical_camera != old.use_physical_camera || in.aperture_shape != old.aperture_shape;
  if (in.aperture_shape != LUMINARY_APERTURE_ROUND) d = d || in.aperture_blade_count != old.aperture_blade_count;
  if (in.use_physical_camera) {
    const auto& a = in.physical; const auto& b = old.physical;
    d = d || a.allow_reflections != b.allow_reflections || a.use_spectral_rendering != b.use_spectral_rendering || a.focal_length != b.focal_length ||
        a.front_focal_point != b.front_focal_point || a.back_focal_point != b.back_focal_point || a.front_principal_point != b.front_principal_point ||
        a.back_principal_point != b.back_principal_point || a.aperture_point != b.aperture_point || a.aperture_diameter != b.aperture_diameter ||
        a.exit_pupil_point != b.exit_pupil_point || a.exit_pupil_diameter != b.exit_pupil_diameter || a.image_plane_distance != b.image_plane_distance ||
        a.sensor_width != b.sensor_width;
  }
  else d = d || in.thin_lens.fov != old.thin_lens.fov || in.thin_lens.aperture_size != old.thin_lens.aperture_size;
  return d;
}
bool settings_change_restarts(const LuminaryRendererSettings& in, const LuminaryRendererSettings& old) {
  bool d = in.width != old.width || in.height != old.height || in.supersampling != old.supersampling || in.bridge_max_num_vertices != old.bridge_max_num_vertices ||
           in.undersampling != old.undersampling || in.shading_mode != old.shading_mode || in.enable_adaptive_sampling != old.enable_adaptive_sampling ||
           in.region_x != old.region_x || in.region_y != old.region_y || in.region_width != old.region_width || in.region_height != old.region_height;
  if (in.enable_adaptive_sampling)
    d = d || in.adaptive_sampling_max_sampling_rate != old.adaptive_sampling_max_sampling_rate || in.adaptive_sampling_avg_sampling_rate != old.adaptive_sampling_avg_sampling_rate ||
        in.adaptive_sampling_update_interval != old.adaptive_sampling_update_interval || in.adaptive_sampling_exposure_aware != old.adaptive_sampling_exposure_aware;
  if (in.shading_mode == LUMINARY_SHADING_MODE_DEFAULT) d = d || in.max_ray_depth != old.max_ray_depth;
  return d;
}
template <typename T> bool change_restarts(const T&, const T&) { return true; }  // sky, ocean, cloud, fog, particles: every field feeds the integration
inline bool change_restarts(const LuminaryCamera& in, const LuminaryCamera& old) { return camera_change_restarts(in, old); }
inline bool change_restarts(const LuminaryRendererSettings& in, const LuminaryRendererSettings& old) { return settings_change_restarts(in, old); }

LuminaryResult ensure_device_scene(LuminaryHost* h) {
  if (h->device_scene_valid) return LUMINARY_SUCCESS;
  if (h->hdri_origin_pending) {  // sky_hdri_update (device/device_sky.c:249-266): the panorama follows the camera only when the sky is dirty
    const LuminaryVec3 p = h->scene.camera.pos;
    h->scene.hdri_origin[0] = p.x; h->scene.hdri_origin[1] = p.y; h->scene.hdri_origin[2] = p.z;
    h->hdri_origin_pending = false;
  }
  static const std::vector<uint32_t> bluenoise = embedded_bluenoise();
  const std::string err = lum::update_device_scene(h->scene, bluenoise, h->scene_dirty, &h->device_scene);
  if (!err.empty()) { h->log.push_back(err); std::fprintf(stderr, "[luminary_amd] %s\n", err.c_str()); h->scene_dirty = h->core_dirty = LUMC_DIRTY_ALL; return LUMINARY_ERROR_API_EXCEPTION; }
  // what was re-encoded is what the devices have to take over (the encoder may add a part: the texture pool when the sky mode moves the moon in or out)
  h->core_dirty |= h->device_scene.rebuilt;
  for (auto& slot : h->devices) slot.dirty |= h->device_scene.rebuilt;
  h->scene_dirty = 0;
  h->device_scene_valid = true;
  h->core_scene_valid = false;
  return LUMINARY_SUCCESS;
}

LuminaryResult ensure_core(LuminaryHost* h) {
  if (!h->core) {
    if (lumc_context_create(h->device_ordinal, &h->core)) {
      std::fprintf(stderr, "[luminary_amd] no usable HIP device: %s\n", lumc_last_error(h->core));
      lumc_context_destroy(h->core);
      h->core = nullptr;
      return LUMINARY_ERROR_CUDA;
    }
  }
  const LuminaryResult r = ensure_device_scene(h);
  if (r) return r;
  if (!h->core_scene_valid) {
    if (lumc_scene_update(h->core, &h->device_scene.view, h->core_dirty)) { std::fprintf(stderr, "[luminary_amd] %s\n", lumc_last_error(h->core)); h->core_dirty = LUMC_DIRTY_ALL; return LUMINARY_ERROR_CUDA; }
    h->core_dirty = 0;
    h->core_scene_valid = true;
    // The accumulation starts over. With the frame size unchanged the pixel sets of the devices stay as they are and only their accumulators
    // are cleared; a new size makes the render loop set them up again (num_pixels = 0).
    const LumDeviceSceneView& v = h->device_scene.view;
    const bool same_frame = h->num_pixels != 0 && h->core_width == v.width && h->core_height == v.height;
    h->core_width = v.width; h->core_height = v.height;
    if (!same_frame || h->partition_n > 1 || lumc_clear_accumulators(h->core)) h->num_pixels = 0;
  }
  return LUMINARY_SUCCESS;
}

// The enabled devices, main device first. Rendering is tiled over them when the whole frame is rendered uniformly; adaptive sampling, render
// regions / pixel subsets and the undersampling preview run on the main device alone.
std::vector<LuminaryHost::DeviceSlot*> enabled_slots(LuminaryHost* h) {
  std::vector<LuminaryHost::DeviceSlot*> out;
  if (h->main_slot < h->devices.size()) out.push_back(&h->devices[h->main_slot]);
  for (uint32_t i = 0; i < h->devices.size(); i++)
    if (i != h->main_slot && h->devices[i].enabled) out.push_back(&h->devices[i]);
  return out;
}

// Contexts with the current scene on every device of the partition; slot 0 of the result is the main device (h->core).
LuminaryResult ensure_partition_cores(LuminaryHost* h, std::vector<LumContext*>* cores) {
  const LuminaryResult r = ensure_core(h);
  if (r) return r;
  cores->clear();
  for (LuminaryHost::DeviceSlot* slot : enabled_slots(h)) {
    if (slot == &h->devices[h->main_slot]) { slot->core = h->core; slot->scene_valid = true; slot->dirty = 0; cores->push_back(h->core); continue; }
    if (!slot->core) {
      if (lumc_context_create(slot->ordinal, &slot->core)) {
        std::fprintf(stderr, "[luminary_amd] device %d: %s\n", slot->ordinal, lumc_last_error(slot->core));
        lumc_context_destroy(slot->core);
        slot->core = nullptr;
        return LUMINARY_ERROR_CUDA;
      }
      lumc_set_flavour(slot->core, lumc_get_flavour(h->core));
      slot->scene_valid = false;
    }
    if (!slot->scene_valid) {
      if (lumc_scene_update(slot->core, &h->device_scene.view, slot->dirty)) { std::fprintf(stderr, "[luminary_amd] device %d: %s\n", slot->ordinal, lumc_last_error(slot->core)); slot->dirty = LUMC_DIRTY_ALL; return LUMINARY_ERROR_CUDA; }
      slot->dirty = 0;
      slot->scene_valid = true;
      h->partition_n = 0;  // the accumulation restarts: the render loop deals the tiles again (which also clears every device's accumulators)
    }
    cores->push_back(slot->core);
  }
  return LUMINARY_SUCCESS;
}

}  // namespace

extern "C" {

// ---- error.c ---- (the reference's texts, checked against its own error.c in tests/test_reference_host.py; codes 8 and 9 keep their
// wording for frontends that print or match them, although the errors come from HIP and the BVH builders here)
const char* luminary_result_to_string(LuminaryResult result) {
  switch (result & ~LUMINARY_ERROR_PROPAGATED) {
    case LUMINARY_SUCCESS: return "Success";
    case LUMINARY_ERROR_ARGUMENT_NULL: return "Encountered NULL argument";
    case LUMINARY_ERROR_NOT_IMPLEMENTED: return "Encountered a section that is not implemented";
    case LUMINARY_ERROR_INVALID_API_ARGUMENT: return "Encountered an invalid argument";
    case LUMINARY_ERROR_MEMORY_LEAK: return "Identified a memory leak";
    case LUMINARY_ERROR_OUT_OF_MEMORY: return "Ran out of memory";
    case LUMINARY_ERROR_C_STD: return "Encountered an error in a call to a C stdlib function";
    case LUMINARY_ERROR_API_EXCEPTION: return "Encountered an internal error";
    case LUMINARY_ERROR_CUDA: return "Encountered an error reported by CUDA";
    case LUMINARY_ERROR_OPTIX: return "Encountered an error reported by OptiX";
    case LUMINARY_ERROR_PREVIOUS_ERROR: return "Encountered an unstable state due to a previous error";
    case LUMINARY_ERROR_DEBUG_ASSERT: return "Encountered an invalid state during debugging";
    case LUMINARY_ERROR_MISSING_DATA: return "Missing necessary embedded data";
    case LUMINARY_ERROR_INVALID_DEVICE: return "Specified invalid device";
    default: return "Unknown";
  }
}

// ---- luminary.c:7-31 ----
void luminary_init(void) {}
void luminary_shutdown(void) {}

// ---- path.c ----
LuminaryResult luminary_path_create(LuminaryPath** path) { CHECK_NULL(path); *path = new LuminaryPath(); return LUMINARY_SUCCESS; }
LuminaryResult luminary_path_set_from_string(LuminaryPath* path, const char* string) { CHECK_NULL(path); CHECK_NULL(string); path->value = string; return LUMINARY_SUCCESS; }
LuminaryResult luminary_path_destroy(LuminaryPath** path) { CHECK_NULL(path); CHECK_NULL(*path); delete *path; *path = nullptr; return LUMINARY_SUCCESS; }

// ---- host.c:292-404 ----
LuminaryResult luminary_host_create(LuminaryHost** host, LuminaryHostCreateInfo info) {
  CHECK_NULL(host);
  LuminaryHost* h = new LuminaryHost();
  // one process per GPU: the lowest set bit of the mask selects the ordinal (torch.distributed sets LOCAL_RANK for the launcher)
  int ordinal = 0;
  if (info.device_mask != 0) while (!((info.device_mask >> ordinal) & 1u) && ordinal < 31) ordinal++;
  if (const char* lr = std::getenv("LOCAL_RANK")) { if (info.device_mask == LUMINARY_HOST_CREATE_INFO_DEVICE_MASK_ALL_DEVICES) ordinal = std::atoi(lr); }
  h->device_ordinal = ordinal;
  // One slot per visible device; the mask enables them (device_manager.c:791-822), the lowest enabled one is the main device. Under a
  // one-process-per-GPU launcher (LOCAL_RANK set, default mask) the process drives its own GPU only. LUM_MAX_DEVICES caps the number of
  // devices a host uses; LUM_FAKE_DEVICES=n (tests on a single-GPU box) presents device 0 n times.
  {
    int count = lumc_device_count();
    const char* fake = std::getenv("LUM_FAKE_DEVICES");
    const int fake_n = fake ? std::atoi(fake) : 0;
    if (fake_n > 1 && count >= 1) count = fake_n;
    int cap = 32;
    if (const char* e = std::getenv("LUM_MAX_DEVICES")) cap = std::max(1, std::atoi(e));
    const bool own_gpu_only = std::getenv("LOCAL_RANK") && info.device_mask == LUMINARY_HOST_CREATE_INFO_DEVICE_MASK_ALL_DEVICES;
    int enabled = 0;
    for (int i = 0; i < count && i < 32; i++) {
      LuminaryHost::DeviceSlot slot;
      slot.ordinal = fake_n > 1 ? 0 : i;
      slot.enabled = own_gpu_only ? (i == ordinal) : (((info.device_mask >> i) & 1u) != 0 && enabled < cap);
      if (slot.enabled) enabled++;
      h->devices.push_back(slot);
    }
    if (h->devices.empty()) { LuminaryHost::DeviceSlot slot; slot.ordinal = ordinal; slot.enabled = true; h->devices.push_back(slot); }  // no GPU visible: calls that need one fail later
    h->main_slot = 0;
    bool found = false;
    for (uint32_t i = 0; i < h->devices.size() && !found; i++) if (h->devices[i].enabled) { h->main_slot = i; found = true; }
    if (!found) { h->devices[0].enabled = true; h->main_slot = 0; }
    h->device_ordinal = h->devices[h->main_slot].ordinal;
  }
  // the reference names its queue workers "Host", "Device" and "Worker n" (host.c:318-330, device_manager.c:874)
  if (thread_status_create(&h->status_host) || thread_status_create(&h->status_device)) { delete h; return LUMINARY_ERROR_OUT_OF_MEMORY; }
  thread_status_set_worker_name(h->status_host, "Host");
  thread_status_set_worker_name(h->status_device, "Device");
  *host = h;
  return LUMINARY_SUCCESS;
}

namespace {
LuminaryResult render_locked(LuminaryHost* host, uint32_t num_samples, uint32_t samples_per_pass);
uint32_t pass_size(LuminaryHost* h);
void stop_worker(LuminaryHost* h, bool join) {
  {
    std::lock_guard<std::mutex> l(h->worker_mutex);
    h->async_active = false;
    if (join) h->worker_stop = true;
  }
  h->worker_cv.notify_all();
  if (join && h->worker_started) { h->worker.join(); h->worker_started = false; }
  else { ApiLock wait_for_the_running_iteration(h); }
}

// The "Device" worker: one iteration = the next sample allocations of the running accumulation (luminary_ext_render, which also rebuilds
// the scene after an edit and produces the outputs that are due). Small chunks first, so that the first images of a new accumulation
// appear quickly, then whole wavefront passes of pass_size() sample ids.
void worker_main(LuminaryHost* h) {
  for (;;) {
    {
      std::unique_lock<std::mutex> l(h->worker_mutex);
      h->worker_cv.wait(l, [&] { return h->worker_stop || (h->async_active && !h->async_failed); });
      if (h->worker_stop) break;
    }
    while (h->api_waiting.load() > 0) std::this_thread::yield();  // edits first
    LuminaryResult r = LUMINARY_SUCCESS;
    bool idle = false;
    {
      // One lock from the decision to the last kernel of the iteration: the flag is looked at again under the host's mutex (a stop that set it
      // and then took and released this mutex has been seen by now, so nothing renders after luminary_ext_stop_render returned), and the
      // first sample id is computed under the same lock the passes run under (no start / edit / synchronous render can slip in between).
      ApiLock lock(h);
      bool active;
      { std::lock_guard<std::mutex> l(h->worker_mutex); active = h->async_active && !h->async_failed && !h->worker_stop; }
      if (!active) continue;
      const uint32_t accumulated = h->accumulated_samples;
      if (accumulated >= (1u << 20)) idle = true;  // every sample id is used up (MAX_NUM_GLOBAL_SAMPLES, device_utils.h:39): idle until the next edit
      else {
        thread_status_start(h->status_device, h->core_scene_valid ? "Rendering" : "Updating scene");
        // small allocations first so that the first images of a new accumulation appear quickly, then whole passes (pass_size)
        const uint32_t cap = pass_size(h);
        const uint32_t chunk = accumulated < 1u ? 1u : (accumulated < cap ? accumulated : cap);
        r = render_locked(h, chunk, cap);
      }
    }
    if (idle) {
      std::unique_lock<std::mutex> l(h->worker_mutex);
      h->worker_cv.wait_for(l, std::chrono::milliseconds(50));
      continue;
    }
    thread_status_stop(h->status_device);
    if (r != LUMINARY_SUCCESS) {
      std::fprintf(stderr, "[luminary_amd] render worker: %s; waiting for the next scene edit or luminary_host_start_new_render\n", luminary_result_to_string(r));
      std::lock_guard<std::mutex> l(h->worker_mutex);
      h->async_failed = true;
    }
  }
}
}  // namespace

LuminaryResult luminary_host_destroy(LuminaryHost** host) {
  CHECK_NULL(host); CHECK_NULL(*host);
  stop_worker(*host, true);
  for (uint32_t i = 0; i < (*host)->devices.size(); i++)
    if (i != (*host)->main_slot && (*host)->devices[i].core) lumc_context_destroy((*host)->devices[i].core);
  if ((*host)->core) lumc_context_destroy((*host)->core);
  thread_status_destroy(&(*host)->status_host);
  thread_status_destroy(&(*host)->status_device);
  delete *host;
  *host = nullptr;
  return LUMINARY_SUCCESS;
}

// host.c:406-414: marks the integration dirty, i.e. the accumulation restarts; the reference's Device thread then renders until the host
// is destroyed. Here this call is what starts the "Device" worker (both modes of Mandarin Duck call it before polling for outputs,
// mandarin_duck.c:153, :200).
LuminaryResult luminary_host_start_new_render(LuminaryHost* host) {
  CHECK_NULL(host);
  {
    ApiLock lock(host);
    host->accumulated_samples = 0;
    host->render_seconds = 0.0;
    host->adaptive_active = false;
    if (host->core && host->num_pixels) lumc_clear_accumulators(host->core);
  }
  {
    std::lock_guard<std::mutex> l(host->worker_mutex);
    host->async_active = true;
    host->async_failed = false;
    if (!host->worker_started) { host->worker_stop = false; host->worker = std::thread(worker_main, host); host->worker_started = true; }
  }
  host->worker_cv.notify_all();
  return LUMINARY_SUCCESS;
}
// Additive: stops the asynchronous rendering started by luminary_host_start_new_render (returns once the running iteration has ended); the
// accumulated frame stays. The luminary_ext_render* calls drive the same loop synchronously afterwards.
LuminaryResult luminary_ext_stop_render(LuminaryHost* host) {
  CHECK_NULL(host);
  stop_worker(host, false);
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_ext_is_rendering(LuminaryHost* host, bool* rendering, uint32_t* accumulated_samples) {
  CHECK_NULL(host);
  { std::lock_guard<std::mutex> l(host->worker_mutex); if (rendering) *rendering = host->async_active && !host->async_failed; }
  if (accumulated_samples) { ApiLock lock(host); *accumulated_samples = host->accumulated_samples; }
  return LUMINARY_SUCCESS;
}

// host.c:416-470, device_manager.c:529-572: the visible devices, which of them render, and the main device (renders its tiles and produces
// every output; re-elected as the lowest enabled device when it is disabled)
LuminaryResult luminary_host_get_device_count(LuminaryHost* host, uint32_t* device_count) {
  CHECK_NULL(host); CHECK_NULL(device_count);
  *device_count = (uint32_t) host->devices.size();
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_host_get_device_info(LuminaryHost* host, uint32_t device_id, LuminaryDeviceInfo* info) {
  CHECK_NULL(host); CHECK_NULL(info);
  if (device_id >= host->devices.size()) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  ApiLock lock(host);
  std::memset(info, 0, sizeof(*info));
  info->is_main_device = device_id == host->main_slot; info->is_enabled = host->devices[device_id].enabled; info->is_unavailable = false;
  char name[200];
  if (lumc_device_name(host->devices[device_id].ordinal, name, sizeof(name))) { std::snprintf(name, sizeof(name), "HIP device %d", host->devices[device_id].ordinal); info->is_unavailable = true; }
  std::snprintf(info->name, sizeof(info->name), "%s", name);
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_host_set_device_enable(LuminaryHost* host, uint32_t device_id, bool enable) {
  CHECK_NULL(host);
  if (device_id >= host->devices.size()) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  ApiLock lock(host);
  if (host->devices[device_id].enabled == enable) return LUMINARY_SUCCESS;
  if (!enable) {
    uint32_t others = 0;
    for (uint32_t i = 0; i < host->devices.size(); i++) if (i != device_id && host->devices[i].enabled) others++;
    if (others == 0) return LUMINARY_ERROR_INVALID_API_ARGUMENT;  // "No device could be selected as the main device" (device_manager.c:44-46)
  }
  host->devices[device_id].enabled = enable;
  if (!enable && device_id == host->main_slot) {  // the main device changes: its context (accumulators, outputs in flight) goes with it
    if (host->core) { lumc_context_destroy(host->core); host->core = nullptr; }
    host->devices[device_id].core = nullptr;
    for (uint32_t i = 0; i < host->devices.size(); i++) if (host->devices[i].enabled) { host->main_slot = i; break; }
    LuminaryHost::DeviceSlot& m = host->devices[host->main_slot];
    host->device_ordinal = m.ordinal;
    if (m.core) { host->core = m.core; }  // its render context becomes the main context
    host->core_scene_valid = false;
    host->core_dirty = LUMC_DIRTY_ALL;  // the new main context takes the whole scene again
    host->num_pixels = 0;
  }
  host->partition_n = 0;
  host->comm_ready = false;
  invalidate(host, 0);  // the integration restarts with the new partition; the scene itself did not change
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_host_start_device(LuminaryHost* host, uint32_t index) { return luminary_host_set_device_enable(host, index, true); }
LuminaryResult luminary_host_shutdown_device(LuminaryHost* host, uint32_t index) { return luminary_host_set_device_enable(host, index, false); }

// host.c:35-100 (+ :472-532)
LuminaryResult luminary_host_load_obj_file(LuminaryHost* host, LuminaryPath* path) {
  CHECK_NULL(host); CHECK_NULL(path);
  ApiLock lock(host);
  lum::HostMesh mesh;
  std::vector<LuminaryMaterial> mats;
  std::vector<std::string> warnings;
  std::string err;
  std::vector<lum::HostTexture> textures;
  if (!lum::load_obj(path->value, lum::ObjLoadArgs(), (uint32_t) host->scene.materials.size(), &mesh, &mats, &warnings, &err, &textures, (uint32_t) host->scene.textures.size())) {
    std::fprintf(stderr, "[luminary_amd] %s\n", err.c_str());
    return LUMINARY_ERROR_API_EXCEPTION;
  }
  for (auto& w : warnings) std::fprintf(stderr, "[luminary_amd] warning: %s\n", w.c_str());
  if (mesh.triangle_count() == 0 && mesh.name.empty()) return LUMINARY_SUCCESS;
  host->scene.materials.insert(host->scene.materials.end(), mats.begin(), mats.end());
  for (auto& t : textures) host->scene.textures.push_back(std::move(t));
  host->scene.meshes.push_back(std::move(mesh));
  invalidate(host, LUMC_DIRTY_MESHES | LUMC_DIRTY_INSTANCES | LUMC_DIRTY_MATERIALS | LUMC_DIRTY_TEXTURES | LUMC_DIRTY_LIGHTS);
  return LUMINARY_SUCCESS;
}

// host.c:534-605
LuminaryResult luminary_host_load_lum_file(LuminaryHost* host, LuminaryPath* path) {
  CHECK_NULL(host); CHECK_NULL(path);
  ApiLock lock(host);
  lum::LumFileContent content;
  lum::default_settings(&content.settings); lum::default_camera(&content.camera); lum::default_ocean(&content.ocean); lum::default_sky(&content.sky);
  lum::default_cloud(&content.cloud); lum::default_fog(&content.fog); lum::default_particles(&content.particles);
  std::vector<std::string> warnings;
  std::string err;
  if (!lum::load_lum_v4(path->value, &content, &warnings, &err)) { std::fprintf(stderr, "[luminary_amd] %s\n", err.c_str()); return LUMINARY_ERROR_API_EXCEPTION; }
  // every mesh file is read before anything is committed: a file that fails to load leaves the host's scene exactly as it was
  std::vector<lum::HostMesh> new_meshes;
  std::vector<LuminaryMaterial> new_materials;
  std::vector<lum::HostTexture> new_textures;
  for (const std::string& obj : content.obj_files) {
    lum::HostMesh mesh;
    std::vector<LuminaryMaterial> mats;
    std::vector<lum::HostTexture> textures;
    if (!lum::load_obj(lum::extend_path(path->value, obj), content.obj_args, (uint32_t) (host->scene.materials.size() + new_materials.size()), &mesh, &mats, &warnings, &err, &textures,
                       (uint32_t) (host->scene.textures.size() + new_textures.size()))) {
      std::fprintf(stderr, "[luminary_amd] %s\n", err.c_str());
      return LUMINARY_ERROR_API_EXCEPTION;
    }
    new_materials.insert(new_materials.end(), mats.begin(), mats.end());
    for (auto& t : textures) new_textures.push_back(std::move(t));
    new_meshes.push_back(std::move(mesh));
  }
  host->scene.materials.insert(host->scene.materials.end(), new_materials.begin(), new_materials.end());
  for (auto& t : new_textures) host->scene.textures.push_back(std::move(t));
  for (auto& m : new_meshes) {
    const uint32_t mesh_id = (uint32_t) host->scene.meshes.size();
    host->scene.meshes.push_back(std::move(m));
    lum::HostInstance inst;
    inst.mesh_id = mesh_id;
    host->scene.instances.push_back(inst);
  }
  for (auto& w : warnings) std::fprintf(stderr, "[luminary_amd] warning: %s\n", w.c_str());
  host->scene.settings = content.settings; host->scene.camera = content.camera; host->scene.ocean = content.ocean; host->scene.sky = content.sky;
  host->scene.cloud = content.cloud; host->scene.fog = content.fog; host->scene.particles = content.particles;
  host->hdri_origin_pending = true;
  invalidate(host);
  return LUMINARY_SUCCESS;
}

// host.c:607-613 -> sample_time_get_time (device_sampletime.c:32-45): with one device, the time its latest sample took (milliseconds; 0 before the first)
LuminaryResult luminary_host_get_current_sample_time(LuminaryHost* host, double* time) { CHECK_NULL(host); CHECK_NULL(time); *time = host->last_sample_ms; return LUMINARY_SUCCESS; }
// host.c:615-703. Two queue workers exist here: 0 "Host" (scene edits are applied on the caller's thread, so it never reports a task) and
// 1 "Device", the render worker, with what it is doing and for how long (thread_status.h). The reference adds 16 "Worker n" threads that
// load mesh files in parallel (host.c:15-20); files are loaded on the caller's thread here.
LuminaryResult luminary_host_get_num_queue_workers(const LuminaryHost* host, uint32_t* n) { CHECK_NULL(host); CHECK_NULL(n); *n = 2; return LUMINARY_SUCCESS; }
static LuminaryThreadStatus* queue_worker_status(const LuminaryHost* host, uint32_t id) { return id == 0 ? host->status_host : id == 1 ? host->status_device : nullptr; }
LuminaryResult luminary_host_get_queue_worker_name(const LuminaryHost* host, uint32_t id, const char** string) {
  CHECK_NULL(host); CHECK_NULL(string);
  LuminaryThreadStatus* st = queue_worker_status(host, id);
  if (!st) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  return thread_status_get_worker_name(st, string);
}
LuminaryResult luminary_host_get_queue_worker_string(const LuminaryHost* host, uint32_t id, const char** string) {
  CHECK_NULL(host); CHECK_NULL(string);
  LuminaryThreadStatus* st = queue_worker_status(host, id);
  if (!st) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  return thread_status_get_string(st, string);
}
LuminaryResult luminary_host_get_queue_worker_time(const LuminaryHost* host, uint32_t id, double* time) {
  CHECK_NULL(host); CHECK_NULL(time);
  LuminaryThreadStatus* st = queue_worker_status(host, id);
  if (!st) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  return thread_status_get_time(st, time);
}

// ---- output chain (host.c:930-1075, host_output_handler.c, device_output.c:178-343) ----
// Rendering is synchronous here (luminary_ext_render_samples), so outputs are produced on the caller's thread right after the pass
// that reaches the sample count they are due at; the handle/promise semantics are the reference's.
LuminaryResult luminary_host_set_output_properties(LuminaryHost* host, LuminaryOutputProperties p) { CHECK_NULL(host); host->outputs.set_properties(p); return LUMINARY_SUCCESS; }
LuminaryResult luminary_host_request_output(LuminaryHost* host, LuminaryOutputRequestProperties p, LuminaryOutputPromiseHandle* handle) {
  CHECK_NULL(host); CHECK_NULL(handle);
  if (p.width < 2 || p.height < 2) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  *handle = host->outputs.add_request(p);
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_host_try_await_output(LuminaryHost* host, LuminaryOutputPromiseHandle handle, LuminaryOutputHandle* out) {
  CHECK_NULL(host); CHECK_NULL(out);
  return host->outputs.acquire_from_promise(handle, out);
}
LuminaryResult luminary_host_acquire_output(LuminaryHost* host, LuminaryOutputHandle* out) { CHECK_NULL(host); CHECK_NULL(out); return host->outputs.acquire_recurring(out); }
LuminaryResult luminary_host_get_image(LuminaryHost* host, LuminaryOutputHandle handle, LuminaryImage* image) { CHECK_NULL(host); CHECK_NULL(image); return host->outputs.get_image(handle, image); }
LuminaryResult luminary_host_release_output(LuminaryHost* host, LuminaryOutputHandle handle) { CHECK_NULL(host); return host->outputs.release(handle); }
// host.c:997-1014. The reference reads a G-buffer that its trace kernel fills during undersampled previews (optix_kernel_raytrace.cu:18-76);
// here the pixel's first-sample camera ray is traced on demand, which gives the same fields at any time.
LuminaryResult luminary_host_get_pixel_info(LuminaryHost* host, uint16_t x, uint16_t y, LuminaryPixelQueryResult* result) {
  CHECK_NULL(host); CHECK_NULL(result);
  ApiLock lock(host);
  std::memset(result, 0, sizeof(*result));
  result->instance_id = 0xFFFFFFFFu; result->material_id = 0xFFFF; result->depth = -1.0f;  // DEPTH_INVALID / MATERIAL_ID_INVALID, utils.h:30-33
  if (ensure_core(host)) return LUMINARY_SUCCESS;  // no device: the query has no data, like a G-buffer that is not ready
  const LumDeviceSceneView& v = host->device_scene.view;
  if (x >= v.width || y >= v.height) return LUMINARY_SUCCESS;
  uint32_t q[6];
  if (lumc_pixel_query(host->core, x, y, 0, q)) return LUMINARY_ERROR_CUDA;
  float depth, dir[3];
  std::memcpy(&depth, &q[2], 4); std::memcpy(dir, &q[3], 12);
  result->depth = depth;
  if (q[0] < 0x7FFFFFFFu && q[0] < v.num_instances) {  // HIT_TYPE_TRIANGLE_ID_LIMIT
    const uint32_t mesh = v.instance_mesh_ids[q[0]];
    result->instance_id = q[0];
    result->material_id = (uint16_t) (v.tri_tex[(size_t) (v.mesh_tri_offset[mesh] + q[1]) * 4 + 3] & 0xFFFFu);
  }
  if (depth < FLT_MAX) {  // rel_hit_pos = ray * depth, kept at bfloat16 precision like the G-buffer
    float rel[3] = {dir[0] * depth, dir[1] * depth, dir[2] * depth};
    for (int k = 0; k < 3; k++) { uint32_t b; std::memcpy(&b, &rel[k], 4); b &= 0xFFFF0000u; std::memcpy(&rel[k], &b, 4); }
    result->rel_hit_pos.x = rel[0]; result->rel_hit_pos.y = rel[1]; result->rel_hit_pos.z = rel[2];
  }
  result->pixel_query_is_valid = (result->depth != -1.0f) || (result->instance_id != 0xFFFFFFFFu) || (result->material_id != 0xFFFF);
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_host_save_png(LuminaryHost* host, LuminaryOutputHandle handle, LuminaryPath* path) {
  CHECK_NULL(host); CHECK_NULL(path);
  if (handle == LUMINARY_OUTPUT_HANDLE_INVALID) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  LuminaryResult r = host->outputs.acquire(handle);
  if (r) return r;
  LuminaryImage image;
  r = host->outputs.get_image(handle, &image);
  if (!r) r = lum::write_png(path->value.c_str(), reinterpret_cast<const uint32_t*>(image.buffer), image.width, image.height, image.ld);
  host->outputs.release(handle);
  return r;
}
LuminaryResult luminary_ext_add_texture(LuminaryHost* host, const uint8_t* rgba8, uint32_t width, uint32_t height, float gamma, uint16_t* texture_id) {
  CHECK_NULL(host); CHECK_NULL(rgba8); CHECK_NULL(texture_id);
  if (width == 0 || height == 0 || host->scene.textures.size() >= 0xFFFF) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  ApiLock lock(host);
  lum::HostTexture t;
  t.width = width; t.height = height; t.gamma = gamma;
  t.texels.resize((size_t) width * height);
  std::memcpy(t.texels.data(), rgba8, t.texels.size() * 4);
  host->scene.textures.push_back(std::move(t));
  *texture_id = (uint16_t) (host->scene.textures.size() - 1);
  invalidate(host, LUMC_DIRTY_TEXTURES | LUMC_DIRTY_LIGHTS);  // emissive textures weigh the light tree
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_ext_write_png(const char* path, const uint32_t* argb8, uint32_t width, uint32_t height, size_t ld) { return lum::write_png(path, argb8, width, height, ld); }
// host.c:1077-1084 -> scene_set_hdri_dirty (scene.c:712-722): the panorama is baked again, seen from the camera's current position
LuminaryResult luminary_host_request_sky_hdri_build(LuminaryHost* host) {
  CHECK_NULL(host);
  ApiLock lock(host);
  host->hdri_origin_pending = true;
  invalidate(host, LUMC_DIRTY_CONSTANTS);
  return LUMINARY_SUCCESS;
}

// ---- entity getters / setters (host.c:705-900) ----
#define ENTITY_ACCESSORS(NAME, TYPE, FIELD)                                                             \
  LuminaryResult luminary_host_get_##NAME(LuminaryHost* host, TYPE* out) {                              \
    CHECK_NULL(host); CHECK_NULL(out);                                                                  \
    ApiLock lock(host);                                                      \
    *out = host->scene.FIELD;                                                                           \
    return LUMINARY_SUCCESS;                                                                            \
  }                                                                                                     \
  LuminaryResult luminary_host_set_##NAME(LuminaryHost* host, const TYPE* in) {                         \
    CHECK_NULL(host); CHECK_NULL(in);                                                                   \
    ApiLock lock(host);                                                      \
    if (std::memcmp(&host->scene.FIELD, in, sizeof(TYPE)) != 0) {                                       \
      const bool restarts = change_restarts(*in, host->scene.FIELD);                                    \
      host->scene.FIELD = *in;                                                                          \
      if (std::is_same<TYPE, LuminarySky>::value) host->hdri_origin_pending = true; /* sky.c:45: every sky change dirties the panorama */ \
      /* these entities reach the kernels as constants (and tables derived from them): no mesh, instance or material array is touched */ \
      if (restarts) invalidate(host, LUMC_DIRTY_CONSTANTS | (std::is_same<TYPE, LuminaryParticles>::value ? LUMC_DIRTY_PARTICLES : 0u)); \
      /* else: only the outputs change; the accumulated frame stays (SCENE_DIRTY_FLAG_OUTPUT) */ \
    }                                                                                                   \
    return LUMINARY_SUCCESS;                                                                            \
  }
ENTITY_ACCESSORS(settings, LuminaryRendererSettings, settings)
ENTITY_ACCESSORS(camera, LuminaryCamera, camera)
ENTITY_ACCESSORS(ocean, LuminaryOcean, ocean)
ENTITY_ACCESSORS(sky, LuminarySky, sky)
ENTITY_ACCESSORS(cloud, LuminaryCloud, cloud)
ENTITY_ACCESSORS(fog, LuminaryFog, fog)
ENTITY_ACCESSORS(particles, LuminaryParticles, particles)

LuminaryResult luminary_host_get_material(LuminaryHost* host, uint16_t id, LuminaryMaterial* material) {
  CHECK_NULL(host); CHECK_NULL(material);
  ApiLock lock(host);
  if (id >= host->scene.materials.size()) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  *material = host->scene.materials[id];
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_host_set_material(LuminaryHost* host, uint16_t id, const LuminaryMaterial* material) {
  CHECK_NULL(host); CHECK_NULL(material);
  ApiLock lock(host);
  if (id >= host->scene.materials.size()) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  host->scene.materials[id] = *material;
  host->scene.materials[id].id = id;
  invalidate(host, LUMC_DIRTY_MATERIALS | LUMC_DIRTY_LIGHTS);  // geometry and acceleration structures stay; emission feeds the light tree
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_host_get_instance(LuminaryHost* host, uint32_t id, LuminaryInstance* instance) {
  CHECK_NULL(host); CHECK_NULL(instance);
  ApiLock lock(host);
  if (id >= host->scene.instances.size()) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  const lum::HostInstance& i = host->scene.instances[id];
  instance->id = id; instance->mesh_id = i.mesh_id; instance->position = i.translation; instance->rotation = i.rotation; instance->scale = i.scale;
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_host_set_instance(LuminaryHost* host, const LuminaryInstance* instance) {
  CHECK_NULL(host); CHECK_NULL(instance);
  ApiLock lock(host);
  if (instance->id >= host->scene.instances.size()) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  lum::HostInstance& i = host->scene.instances[instance->id];
  i.mesh_id = instance->mesh_id; i.translation = instance->position; i.rotation = instance->rotation; i.scale = instance->scale; i.active = true;
  invalidate(host, LUMC_DIRTY_INSTANCES | LUMC_DIRTY_LIGHTS);  // top-level tree and light tree; the per-mesh trees are reused
  return LUMINARY_SUCCESS;
}
// host.c:902-930: writes defaults and the new id back to the caller
LuminaryResult luminary_host_new_instance(LuminaryHost* host, LuminaryInstance* instance) {
  CHECK_NULL(host); CHECK_NULL(instance);
  ApiLock lock(host);
  lum::HostInstance i;
  i.mesh_id = 0;
  host->scene.instances.push_back(i);
  instance->id = (uint32_t) host->scene.instances.size() - 1; instance->mesh_id = i.mesh_id;
  instance->position = i.translation; instance->rotation = i.rotation; instance->scale = i.scale;
  invalidate(host, LUMC_DIRTY_INSTANCES | LUMC_DIRTY_LIGHTS);
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_host_get_num_meshes(LuminaryHost* host, uint32_t* n) { CHECK_NULL(host); CHECK_NULL(n); *n = (uint32_t) host->scene.meshes.size(); return LUMINARY_SUCCESS; }
LuminaryResult luminary_host_get_num_materials(LuminaryHost* host, uint32_t* n) { CHECK_NULL(host); CHECK_NULL(n); *n = (uint32_t) host->scene.materials.size(); return LUMINARY_SUCCESS; }
LuminaryResult luminary_host_get_num_instances(LuminaryHost* host, uint32_t* n) { CHECK_NULL(host); CHECK_NULL(n); *n = (uint32_t) host->scene.instances.size(); return LUMINARY_SUCCESS; }

// ---- additive extension ----
LuminaryResult luminary_ext_add_mesh(LuminaryHost* host, const float* positions, const float* normals, const float* uvs, const uint16_t* material_ids,
                                     uint32_t triangle_count, uint32_t* mesh_id) {
  CHECK_NULL(host); CHECK_NULL(positions); CHECK_NULL(material_ids);
  ApiLock lock(host);
  lum::HostMesh m;
  m.positions.assign(positions, positions + 9 * (size_t) triangle_count);
  m.material_ids.assign(material_ids, material_ids + triangle_count);
  if (uvs) m.uvs.assign(uvs, uvs + 6 * (size_t) triangle_count); else m.uvs.assign(6 * (size_t) triangle_count, 0.0f);
  if (normals) m.normals.assign(normals, normals + 9 * (size_t) triangle_count);
  else {  // face normals, as the .obj path does for files without `vn` (wavefront.c:918-929)
    m.normals.resize(9 * (size_t) triangle_count);
    for (uint32_t t = 0; t < triangle_count; t++) {
      const float* p = positions + 9 * (size_t) t;
      const float e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
      float n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
      const float rl = 1.0f / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (!std::isnan(rl) && !std::isinf(rl)) { n[0] *= rl; n[1] *= rl; n[2] *= rl; }
      for (int k = 0; k < 3; k++) std::memcpy(&m.normals[9 * (size_t) t + 3 * k], n, sizeof(n));
    }
  }
  host->scene.meshes.push_back(std::move(m));
  if (mesh_id) *mesh_id = (uint32_t) host->scene.meshes.size() - 1;
  invalidate(host, LUMC_DIRTY_MESHES | LUMC_DIRTY_INSTANCES | LUMC_DIRTY_LIGHTS);
  return LUMINARY_SUCCESS;
}
// The host-level mesh as the loaders left it (the reference's Mesh / TriangleGeomData, mesh.h:8-14): borrowed pointers, valid until the mesh list changes. What an
// independent encoder (the test oracle's o_scene.c) starts from.
LuminaryResult luminary_ext_get_mesh(LuminaryHost* host, uint32_t mesh_id, const float** positions, const float** normals, const float** uvs,
                                     const uint16_t** material_ids, uint32_t* triangle_count) {
  CHECK_NULL(host); CHECK_NULL(triangle_count);
  ApiLock lock(host);
  if (mesh_id >= host->scene.meshes.size()) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  const lum::HostMesh& m = host->scene.meshes[mesh_id];
  if (positions) *positions = m.positions.data();
  if (normals) *normals = m.normals.data();
  if (uvs) *uvs = m.uvs.data();
  if (material_ids) *material_ids = m.material_ids.data();
  *triangle_count = m.triangle_count();
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_ext_add_material(LuminaryHost* host, const LuminaryMaterial* material, uint16_t* material_id) {
  CHECK_NULL(host); CHECK_NULL(material);
  ApiLock lock(host);
  if (host->scene.materials.size() >= 0xFFFF) return LUMINARY_ERROR_API_EXCEPTION;
  host->scene.materials.push_back(*material);
  host->scene.materials.back().id = (uint32_t) host->scene.materials.size() - 1;
  if (material_id) *material_id = (uint16_t) (host->scene.materials.size() - 1);
  invalidate(host, LUMC_DIRTY_MATERIALS | LUMC_DIRTY_LIGHTS);
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_ext_build_device_scene(LuminaryHost* host, const LumDeviceSceneView** view) {
  CHECK_NULL(host); CHECK_NULL(view);
  ApiLock lock(host);
  const LuminaryResult r = ensure_device_scene(host);
  if (r) return r;
  *view = &host->device_scene.view;
  return LUMINARY_SUCCESS;
}
namespace {
// camera / settings -> device-side output state (device_structs.c:40-88: exposure is stored as exp(exposure))
LumOutputParams output_params(const LuminaryHost* h, uint32_t dst_width, uint32_t dst_height) {
  const LuminaryCamera& c = h->scene.camera;
  const LumDeviceSceneView& v = h->device_scene.view;
  LumOutputParams p;
  std::memset(&p, 0, sizeof(p));
  p.src_width = v.width; p.src_height = v.height; p.dst_width = dst_width; p.dst_height = dst_height;
  p.inv_sample_count = 1.0f / (float) h->accumulated_samples;
  p.exposure = std::exp(c.exposure);
  p.tonemap = (uint32_t) c.tonemap; p.filter = (uint32_t) c.filter; p.dithering = c.dithering ? 1u : 0u; p.purkinje = c.purkinje ? 1u : 0u;
  p.use_color_correction = c.use_color_correction ? 1u : 0u;
  // tonemap_apply leaves the pixel alone for debug shading modes and for the adaptive-sampling diagnostic images (tonemap.cuh:206-211)
  p.passthrough = (h->scene.settings.shading_mode != LUMINARY_SHADING_MODE_DEFAULT ||
                   h->scene.settings.adaptive_sampling_output_mode != LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_BEAUTY) ? 1u : 0u;
  p.purkinje_kappa1 = c.purkinje_kappa1; p.purkinje_kappa2 = c.purkinje_kappa2;
  p.cc_h = c.color_correction.r; p.cc_s = c.color_correction.g; p.cc_v = c.color_correction.b;
  p.film_grain = c.film_grain;
  p.agx_slope = c.agx_custom_slope; p.agx_power = c.agx_custom_power; p.agx_saturation = c.agx_custom_saturation;
  p.supersampling = h->scene.settings.supersampling;  // the frame is rendered at output size << supersampling (device_structs.c:20-21)
  return p;
}

// One render iteration of the undersampling preview (device.c:392-420): stage > 0 while the frame's first sample is rendered coarse to fine.
struct PreviewState { uint32_t stage = 0, iteration = 0; };

// device_setup_undersampling, device.c:1281-1310: the preview runs only when somebody watches (recurring outputs), never for render regions.
// Schedule: stage N with iterations 3..0, stages N-1..1 with iterations 2..0.
std::vector<PreviewState> preview_schedule(LuminaryHost* h) {
  std::vector<PreviewState> out;
  const LuminaryRendererSettings& st = h->scene.settings;
  if (!h->outputs.properties().enabled || !(st.region_width >= 1.0f && st.region_height >= 1.0f)) return out;
  // (with several devices the main device renders the coarse-to-fine first sample alone; the tiles then take its sums over: lumc_accumulators_from_frame)
  for (uint32_t stage = st.undersampling & 31u; stage > 0; stage--)
    for (uint32_t it = (stage == (st.undersampling & 31u)) ? 4u : 3u; it-- > 0;) { PreviewState ps; ps.stage = stage; ps.iteration = it; out.push_back(ps); }
  return out;
}

// accumulation_generate_result (accumulation.cuh:86-200) into the core's result image; the output chain then reads that image with a
// sample count of one. Returns the parameters to hand to lumc_generate_output*.
// device_post_apply (device/device_post.c:204-226), run between the result image and the display chain: bloom, for beauty images of
// the whole frame in the default shading mode
int post_process(LuminaryHost* h, PreviewState preview) {
  const LuminaryRendererSettings& st = h->scene.settings;
  if (st.shading_mode != LUMINARY_SHADING_MODE_DEFAULT || st.adaptive_sampling_output_mode != LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_BEAUTY) return 0;
  if (!(st.region_width >= 1.0f && st.region_height >= 1.0f)) return 0;
  const float blend = h->scene.camera.bloom_blend;
  if (!(blend > 0.0f)) return 0;  // device_post_update, device_post.c:186-202
  const LumDeviceSceneView& v = h->device_scene.view;
  return lumc_post_bloom(h->core, nullptr, v.width, v.height, preview.stage, blend, nullptr);
}

int result_image(LuminaryHost* h, LumOutputParams* p, PreviewState preview) {
  if (preview.stage) {  // accumulation_generate_result_undersampling (device_renderer.c:410-420): the coarse image of the pixels that exist so far
    if (lumc_generate_result_undersampled(h->core, preview.stage, preview.iteration, nullptr, nullptr)) return 1;
    p->inv_sample_count = 1.0f;
    p->undersampling_stage = preview.stage;
    return post_process(h, preview);
  }
  const uint32_t mode = (uint32_t) h->scene.settings.adaptive_sampling_output_mode;
  const uint32_t lem = h->scene.camera.use_local_error_minimization ? 1u : 0u;
  if (lumc_generate_result(h->core, mode, lem, h->adaptive_active ? 0u : h->accumulated_samples, p->exposure, p, nullptr, nullptr)) return 1;
  p->inv_sample_count = 1.0f;
  return post_process(h, preview);
}

// device_output_generate_output, device_output.c:203-270: the recurring output if enabled, then every request that is due now
// `preview`: the state of the iteration just rendered (device.c:1509-1536 generates the output before the state advances).
// Tiled render: the moments of all devices on the main device (one grouped RCCL reduce; peer copies without a communicator), which the
// result / output entry points of the main context then read.
LuminaryResult assemble_partition(LuminaryHost* h) {
  if (h->partition_n <= 1) return LUMINARY_SUCCESS;
  std::vector<LumContext*> cores;
  for (LuminaryHost::DeviceSlot* slot : enabled_slots(h)) if (slot->core) cores.push_back(slot->core);
  if (cores.size() != h->partition_n || cores[0] != h->core) return LUMINARY_ERROR_API_EXCEPTION;
  const LumDeviceSceneView& v = h->device_scene.view;
  // uniform tiled rendering: the devices hold the shares of the 32x32 tile deal, so the frame is a gather of their own pixels (1 / n of a reduce's bytes);
  // adaptive rendering keeps full-frame accumulators with a block mask per device: those frames are summed
  const bool gathered = !h->adaptive_active && lumc_frame_gather_all(cores.data(), (int) cores.size(), v.width, v.height, 0, nullptr) == 0;
  if ((!gathered && lumc_frame_assemble_all(cores.data(), (int) cores.size(), v.width * v.height, 0, nullptr)) || lumc_use_assembled_frame(h->core, 1)) {
    std::fprintf(stderr, "[luminary_amd] %s\n", lumc_last_error(h->core));
    return LUMINARY_ERROR_CUDA;
  }
  return LUMINARY_SUCCESS;
}

LuminaryResult produce_outputs(LuminaryHost* h, PreviewState preview = PreviewState()) {
  if (!h->pixels_all || (h->accumulated_samples == 0 && preview.stage == 0)) return LUMINARY_SUCCESS;
  const LuminaryOutputProperties props = h->outputs.properties();
  const bool wanted = (props.enabled && props.width >= 2 && props.height >= 2) || !h->outputs.pending_requests().empty();
  if (wanted) { const LuminaryResult ra = assemble_partition(h); if (ra) return ra; }
  lum::OutputMeta meta;
  meta.sample_count = h->accumulated_samples;
  meta.time = (float) h->render_seconds;
  if (props.enabled && props.width >= 2 && props.height >= 2) {
    meta.width = props.width; meta.height = props.height;
    const uint32_t handle = h->outputs.begin_recurring(meta);
    LumOutputParams p = output_params(h, meta.width, meta.height);
    if (result_image(h, &p, preview) || lumc_generate_output_host(h->core, &p, lumc_result_image(h->core), h->outputs.data(handle), nullptr)) {
      h->outputs.publish(handle);
      return LUMINARY_ERROR_CUDA;
    }
    h->outputs.publish(handle);
  }
  for (const LuminaryOutputRequestProperties& req : h->outputs.pending_requests()) {
    if (req.sample_count > 0 && req.sample_count != h->accumulated_samples) continue;
    meta.width = req.width; meta.height = req.height;
    uint32_t handle;
    if (h->outputs.begin_for_request(meta, &handle)) continue;
    LumOutputParams p = output_params(h, meta.width, meta.height);
    const int rc = result_image(h, &p, preview) || lumc_generate_output_host(h->core, &p, lumc_result_image(h->core), h->outputs.data(handle), nullptr);
    h->outputs.publish(handle);
    if (rc) return LUMINARY_ERROR_CUDA;
  }
  return LUMINARY_SUCCESS;
}
}  // namespace

namespace {
// Sample ids per wavefront pass of the render loop. Deep bounces keep few paths alive, so many ids share a pass to keep 256 CUs busy (hall +5 %,
// Example-class +26 %, scan +35 % from 8 to 32, profiles/r02_ab_experiments.txt) - but a pass is also the latency of the next image, so a frame
// that is being watched (recurring outputs) keeps 8. Round 6: 64 ids for a frame nobody watches (same box, 32 -> 64: hall +2.1 %, Example-class +10.7 %,
// scan +14 % samples/s, profiles/r06_ab_experiments.txt). Bounded by the work buffers: at most 128 M paths (about 64 GB of queues, a fifth of the 288 GB) per pass.
uint32_t pass_size(LuminaryHost* h) {
  const uint32_t want = h->outputs.properties().enabled ? 8u : 64u;
  const LuminaryRendererSettings& st = h->scene.settings;
  const uint64_t pixels = std::max<uint64_t>((uint64_t) (st.width << st.supersampling) * (st.height << st.supersampling), 1u);
  const uint64_t fit = std::max<uint64_t>((128ull << 20) / pixels, 1u);
  return (uint32_t) std::min<uint64_t>(want, fit);
}

// luminary_ext_render_samples with the host's mutex held by the caller
LuminaryResult render_samples_locked(LuminaryHost* host, const uint32_t* pixels, uint32_t num_pixels, uint32_t first_sample, uint32_t num_samples,
                                     uint32_t samples_per_pass) {
  std::vector<LumContext*> cores;
  LuminaryResult r = ensure_partition_cores(host, &cores);
  if (r) return r;
  const bool all = pixels == nullptr;
  // the whole frame is tiled over the enabled devices (32x32 tiles dealt by lumc_tile_owner's lattice, lumc_tile_pixels); pixel subsets stay on the main device
  const uint32_t want_n = (all && cores.size() > 1) ? (uint32_t) cores.size() : 1u;
  bool same = (host->num_pixels != 0) && (all == host->pixels_all) && (host->partition_n == want_n);
  if (same && !all) same = (host->pixels.size() == num_pixels) && std::memcmp(host->pixels.data(), pixels, sizeof(uint32_t) * num_pixels) == 0;
  if (host->adaptive_active) same = false;  // leaving adaptive mode restarts the accumulation
  if (!same) {
    const LumDeviceSceneView& v = host->device_scene.view;
    // the first sample of this frame was rendered on the main device as the preview: its sums go to whoever owns the pixels from now on
    const bool handoff = host->handoff_preview && all && first_sample == 1 && host->accumulated_samples == 1 && host->pixels_all && host->num_pixels == v.width * v.height;
    host->handoff_preview = false;
    if (handoff && lumc_frame_assemble(host->core, v.width * v.height, 0, nullptr, nullptr)) { std::fprintf(stderr, "[luminary_amd] %s\n", lumc_last_error(host->core)); return LUMINARY_ERROR_CUDA; }
    if (want_n > 1) {
      std::vector<uint32_t> tiles;
      for (uint32_t k = 0; k < want_n; k++) {
        uint32_t count = 0;
        lumc_tile_pixels(v.width, v.height, k, want_n, 32, nullptr, &count);
        tiles.resize(count ? count : 1);
        lumc_tile_pixels(v.width, v.height, k, want_n, 32, tiles.data(), &count);
        if (lumc_set_pixels(cores[k], tiles.data(), count)) { std::fprintf(stderr, "[luminary_amd] %s\n", lumc_last_error(cores[k])); return LUMINARY_ERROR_CUDA; }
      }
      if (!host->comm_ready) {  // one RCCL communicator over the partition's GPUs; without one the frame is assembled through peer copies
        if (lumc_comm_init_all(cores.data(), (int) want_n) == 0) host->comm_ready = true;
        else std::fprintf(stderr, "[luminary_amd] no RCCL communicator (%s): frames are assembled through peer copies\n", lumc_last_error(cores[0]));
      }
    }
    else {
      lumc_use_assembled_frame(host->core, 0);
      if (lumc_set_pixels(host->core, pixels, num_pixels)) return LUMINARY_ERROR_CUDA;
    }
    host->partition_n = want_n;
    host->adaptive_active = false;
    host->pixels_all = all;
    if (!all) host->pixels.assign(pixels, pixels + num_pixels);
    host->num_pixels = all ? v.width * v.height : num_pixels;
    host->accumulated_samples = 0;
    if (handoff) {
      for (uint32_t k = 0; k < want_n; k++)
        if (lumc_accumulators_from_frame(cores[k], host->core)) { std::fprintf(stderr, "[luminary_amd] %s\n", lumc_last_error(cores[k])); return LUMINARY_ERROR_CUDA; }
      host->accumulated_samples = 1;
    }
    else host->render_seconds = 0.0;
  }
  // stop at every sample count a pending request is keyed to (device_output.c:160-168), produce the outputs due there, go on
  uint32_t done = 0;
  while (done < num_samples) {
    uint32_t chunk = num_samples - done;
    for (const LuminaryOutputRequestProperties& req : host->outputs.pending_requests())
      if (req.sample_count > host->accumulated_samples && req.sample_count - host->accumulated_samples < chunk) chunk = req.sample_count - host->accumulated_samples;
    const auto t0 = std::chrono::steady_clock::now();
    // every device gets its kernels enqueued before the first one is waited for: the GPUs run side by side
    for (uint32_t k = 0; k < want_n; k++)
      if (lumc_render(cores[k], first_sample + done, chunk, samples_per_pass, nullptr, nullptr, nullptr)) { std::fprintf(stderr, "[luminary_amd] %s\n", lumc_last_error(cores[k])); return LUMINARY_ERROR_CUDA; }
    for (uint32_t k = 0; k < want_n; k++)
      if (lumc_synchronize(cores[k])) { std::fprintf(stderr, "[luminary_amd] %s\n", lumc_last_error(cores[k])); return LUMINARY_ERROR_CUDA; }
    { const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); host->render_seconds += dt; host->last_sample_ms = 1e3 * dt / (chunk ? chunk : 1u); }
    host->accumulated_samples += chunk;
    done += chunk;
    r = produce_outputs(host);
    if (r) return r;
  }
  return LUMINARY_SUCCESS;
}
}  // namespace

LuminaryResult luminary_ext_render_samples(LuminaryHost* host, const uint32_t* pixels, uint32_t num_pixels, uint32_t first_sample, uint32_t num_samples,
                                           uint32_t samples_per_pass) {
  CHECK_NULL(host);
  ApiLock lock(host);
  return render_samples_locked(host, pixels, num_pixels, first_sample, num_samples, samples_per_pass);
}
// The frame's first sample as the undersampling preview: every iteration renders one pixel per block, then the outputs are produced
// from the coarse image; the last iteration completes sample 0 of every pixel. Returns through *ran whether the preview applied.
static LuminaryResult render_first_sample_as_preview(LuminaryHost* host, bool* ran) {
  *ran = false;
  const std::vector<PreviewState> schedule = preview_schedule(host);
  if (schedule.empty() || host->accumulated_samples != 0 || !host->pixels_all) return LUMINARY_SUCCESS;
  for (size_t k = 0; k < schedule.size(); k++) {
    const auto t0 = std::chrono::steady_clock::now();
    if (lumc_render_undersampled(host->core, schedule[k].stage, schedule[k].iteration, nullptr) || lumc_synchronize(host->core)) {
      std::fprintf(stderr, "[luminary_amd] %s\n", lumc_last_error(host->core));
      return LUMINARY_ERROR_CUDA;
    }
    { const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); host->render_seconds += dt; host->last_sample_ms = 1e3 * dt; }
    if (k + 1 == schedule.size()) host->accumulated_samples = 1;  // device_renderer_finish_iteration counts the sample with the last iteration
    const LuminaryResult r = produce_outputs(host, schedule[k]);
    if (r) return r;
  }
  *ran = true;
  return LUMINARY_SUCCESS;
}

namespace {
// luminary_ext_render with the host's mutex held by the caller (the render worker and the API call share it): the first sample id of the
// allocation is read under the same lock the passes run under.
LuminaryResult render_locked(LuminaryHost* host, uint32_t num_samples, uint32_t samples_per_pass) {
  const LuminaryRendererSettings settings = host->scene.settings;
  if (!settings.enable_adaptive_sampling) {
    uint32_t first = host->pixels_all && !host->adaptive_active ? host->accumulated_samples : 0;
    if (first == 0 && num_samples > 0) {  // a new accumulation: its first sample may be due as the undersampling preview
      const LuminaryResult r = ensure_core(host);
      if (r) return r;
      if (!preview_schedule(host).empty()) {
        lumc_use_assembled_frame(host->core, 0);
        if (lumc_set_pixels(host->core, nullptr, 0)) return LUMINARY_ERROR_CUDA;
        host->partition_n = 1;
        host->adaptive_active = false;
        host->pixels_all = true;
        host->num_pixels = host->device_scene.view.width * host->device_scene.view.height;
        host->accumulated_samples = 0;
        host->render_seconds = 0.0;
        bool ran = false;
        const LuminaryResult rp = render_first_sample_as_preview(host, &ran);
        if (rp) return rp;
        if (ran) { first = 1; num_samples--; host->handoff_preview = enabled_slots(host).size() > 1; }
      }
    }
    if (num_samples == 0 && !host->handoff_preview) return LUMINARY_SUCCESS;
    return render_samples_locked(host, nullptr, 0, first, num_samples, samples_per_pass);
  }
  // Adaptive sampling over the enabled devices: the reference's main device computes the rates for all devices (device_adaptive_sampler.c:60-74,
  // :205-213; device_manager.c:452-469). Here every device owns the 4x4 blocks of its 32x32 tiles (lumc_adaptive_set_partition), renders their
  // tasks and, when a stage is due, contributes their variances to ONE all-reduce of 4 bytes per block (lumc_adaptive_exchange_all), after which every
  // device derives the same rates: rates, variances and frame equal the single-device run bit for bit.
  std::vector<LumContext*> cores;
  LuminaryResult r = ensure_partition_cores(host, &cores);
  if (r) return r;
  const uint32_t n = (uint32_t) cores.size();
  auto fail = [&](LumContext* c) { std::fprintf(stderr, "[luminary_amd] %s\n", lumc_last_error(c)); return LUMINARY_ERROR_CUDA; };
  if (!host->adaptive_active || host->partition_n != n) {
    const LumDeviceSceneView& v = host->device_scene.view;
    lumc_use_assembled_frame(host->core, 0);
    LumAdaptiveParams ap;
    std::memset(&ap, 0, sizeof(ap));
    ap.max_sampling_rate = settings.adaptive_sampling_max_sampling_rate;
    ap.avg_sampling_rate = settings.adaptive_sampling_avg_sampling_rate;
    ap.update_interval = settings.adaptive_sampling_update_interval;
    ap.tone = output_params(host, v.width, v.height);
    ap.exposure = settings.adaptive_sampling_exposure_aware ? ap.tone.exposure : 0.0f;
    const uint32_t blocks_x = (v.width + 3) / 4, blocks_y = (v.height + 3) / 4, tiles_x = (v.width + 31) / 32;
    std::vector<uint8_t> mask((size_t) blocks_x * blocks_y);
    for (uint32_t k = 0; k < n; k++) {
      if (lumc_set_pixels(cores[k], nullptr, 0) || lumc_adaptive_begin(cores[k], &ap)) return fail(cores[k]);
      if (n > 1) {  // the blocks of a tile belong to the tile's device, like the pixels of a uniform tiled render (lumc_tile_owner)
        for (uint32_t by = 0; by < blocks_y; by++)
          for (uint32_t bx = 0; bx < blocks_x; bx++) mask[(size_t) by * blocks_x + bx] = (lumc_tile_owner(bx / 8, by / 8, tiles_x, n) == k) ? 1 : 0;
        if (lumc_adaptive_set_partition(cores[k], mask.data())) return fail(cores[k]);
      }
    }
    if (n > 1 && !host->comm_ready) {
      if (lumc_comm_init_all(cores.data(), (int) n) == 0) host->comm_ready = true;
      else std::fprintf(stderr, "[luminary_amd] no RCCL communicator (%s): block variances and frames go through the host / peer copies\n", lumc_last_error(cores[0]));
    }
    host->partition_n = n;
    host->pixels_all = true;
    host->num_pixels = v.width * v.height;
    host->accumulated_samples = 0;
    host->render_seconds = 0.0;
    host->handoff_preview = false;
    host->adaptive_active = true;
  }
  uint32_t done = 0;
  if (host->accumulated_samples == 0 && num_samples > 0) {  // execution 0 of stage 0 as the undersampling preview, when one is due
    bool ran = false;
    const uint32_t partition = host->partition_n;
    host->partition_n = 1;  // the preview's images come from the main device's own accumulators
    r = render_first_sample_as_preview(host, &ran);
    host->partition_n = partition;
    if (r) return r;
    if (ran) {
      if (n > 1) {  // every device takes over the first sample of the blocks it owns
        if (lumc_frame_assemble(host->core, host->num_pixels, 0, nullptr, nullptr)) return fail(host->core);
        for (uint32_t k = 0; k < n; k++) if (lumc_accumulators_from_frame(cores[k], host->core)) return fail(cores[k]);
      }
      for (uint32_t k = 0; k < n; k++) if (lumc_adaptive_note_first_sample(cores[k], nullptr)) return fail(cores[k]);
      done = 1;
    }
  }
  auto executions_done = [&](LumContext* c, uint32_t* pending) {
    LumAdaptiveInfo info;
    std::memset(&info, 0, sizeof(info));
    lumc_adaptive_info(c, &info);
    uint32_t total = 0;
    for (uint32_t e : info.executions) total += e;
    if (pending) *pending = info.build_pending;
    return total;
  };
  // a stage may be due right away (the preview was its last execution)
  { uint32_t pending = 0; executions_done(cores[0], &pending); if (n > 1 && pending && lumc_adaptive_exchange_all(cores.data(), (int) n)) return fail(cores[0]); }
  while (done < num_samples) {
    uint32_t chunk = num_samples - done;
    for (const LuminaryOutputRequestProperties& req : host->outputs.pending_requests())
      if (req.sample_count > host->accumulated_samples && req.sample_count - host->accumulated_samples < chunk) chunk = req.sample_count - host->accumulated_samples;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t left = chunk;
    while (left > 0) {  // a partitioned context stops where a stage is due: exchange, go on
      const uint32_t before = executions_done(cores[0], nullptr);
      for (uint32_t k = 0; k < n; k++) if (lumc_adaptive_render(cores[k], left, nullptr)) return fail(cores[k]);
      for (uint32_t k = 0; k < n; k++) if (lumc_synchronize(cores[k])) return fail(cores[k]);
      uint32_t pending = 0;
      const uint32_t ran_now = executions_done(cores[0], &pending) - before;
      if (pending && lumc_adaptive_exchange_all(cores.data(), (int) n)) return fail(cores[0]);
      if (ran_now == 0 && !pending) { std::fprintf(stderr, "[luminary_amd] adaptive rendering made no progress\n"); return LUMINARY_ERROR_API_EXCEPTION; }
      left -= std::min(left, ran_now);
    }
    { const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); host->render_seconds += dt; host->last_sample_ms = 1e3 * dt / (chunk ? chunk : 1u); }
    host->accumulated_samples += chunk;
    done += chunk;
    r = produce_outputs(host);
    if (r) return r;
  }
  return LUMINARY_SUCCESS;
}
}  // namespace

// The reference's render loop for `num_samples` more sample allocations of the whole frame (device_renderer.c:488-575): with
// settings.enable_adaptive_sampling the stage schedule of the adaptive sampler, otherwise one sample id per pixel and allocation.
LuminaryResult luminary_ext_render(LuminaryHost* host, uint32_t num_samples) {
  CHECK_NULL(host);
  ApiLock lock(host);
  return render_locked(host, num_samples, 8);
}
LuminaryResult luminary_ext_get_accumulators(LuminaryHost* host, float* first_moment, float* second_moment, uint32_t* num_pixels) {
  CHECK_NULL(host);
  ApiLock lock(host);
  if (num_pixels) *num_pixels = host->num_pixels;
  if (!host->core || host->num_pixels == 0) return LUMINARY_ERROR_API_EXCEPTION;
  if (host->partition_n > 1) {  // tiled render: the assembled frame
    if (!(first_moment || second_moment)) return LUMINARY_SUCCESS;
    const LuminaryResult ra = assemble_partition(host);
    if (ra) return ra;
    return lumc_frame_download(host->core, host->num_pixels, first_moment, second_moment) ? LUMINARY_ERROR_CUDA : LUMINARY_SUCCESS;
  }
  if ((first_moment || second_moment) && lumc_download_accumulators(host->core, first_moment, second_moment)) return LUMINARY_ERROR_CUDA;
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_ext_get_radiance(LuminaryHost* host, float* rgb, uint32_t* sample_count, uint32_t width, uint32_t height) {
  CHECK_NULL(host); CHECK_NULL(rgb);
  ApiLock lock(host);
  if (!host->core || !host->pixels_all || host->num_pixels == 0) return LUMINARY_ERROR_API_EXCEPTION;
  const LumDeviceSceneView& v = host->device_scene.view;
  if (width != v.width || height != v.height) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  if (host->adaptive_active) {  // every pixel has its own sample count: the beauty result image is the radiance
    if (host->partition_n > 1) { const LuminaryResult ra = assemble_partition(host); if (ra) return ra; }
    std::vector<float> planes(3 * (size_t) host->num_pixels);
    if (lumc_generate_result_host(host->core, 0, 0, 0, 1.0f, nullptr, planes.data())) return LUMINARY_ERROR_CUDA;
    for (size_t p = 0; p < host->num_pixels; p++)
      for (int c = 0; c < 3; c++) rgb[3 * p + c] = planes[(size_t) c * host->num_pixels + p];
    if (sample_count) *sample_count = host->accumulated_samples;
    return LUMINARY_SUCCESS;
  }
  std::vector<float> fm(3 * (size_t) host->num_pixels);
  if (host->partition_n > 1) {
    const LuminaryResult ra = assemble_partition(host);
    if (ra) return ra;
    if (lumc_frame_download(host->core, host->num_pixels, fm.data(), nullptr)) return LUMINARY_ERROR_CUDA;
  }
  else if (lumc_download_accumulators(host->core, fm.data(), nullptr)) return LUMINARY_ERROR_CUDA;
  const float norm = host->accumulated_samples ? 1.0f / host->accumulated_samples : 0.0f;  // accumulation.cuh:149-153
  for (size_t p = 0; p < host->num_pixels; p++)
    for (int c = 0; c < 3; c++) rgb[3 * p + c] = fm[(size_t) c * host->num_pixels + p] * norm;
  if (sample_count) *sample_count = host->accumulated_samples;
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_ext_get_ray_counters(LuminaryHost* host, uint64_t out[8]) {
  CHECK_NULL(host); CHECK_NULL(out);
  if (!host->core) return LUMINARY_ERROR_API_EXCEPTION;
  ApiLock lock(host);
  uint64_t all[LUMC_CNT_COUNT] = {0};  // the core keeps more counters than this call returns
  if (lumc_counters(host->core, all)) return LUMINARY_ERROR_CUDA;
  if (host->partition_n > 1)  // rays of every device of the tiled render
    for (LuminaryHost::DeviceSlot* slot : enabled_slots(host)) {
      if (!slot->core || slot->core == host->core) continue;
      uint64_t c[LUMC_CNT_COUNT] = {0};
      if (lumc_counters(slot->core, c)) return LUMINARY_ERROR_CUDA;
      for (int k = 0; k < LUMC_CNT_COUNT; k++) all[k] += c[k];
    }
  for (int k = 0; k < 8; k++) out[k] = all[k];
  return LUMINARY_SUCCESS;
}
// host_math.c:6-21 as the instance transforms use it (exposed so that it can be checked against the reference's own function)
LuminaryResult luminary_ext_euler_to_quaternion(const float rotation[3], float quaternion[4]) {
  CHECK_NULL(rotation); CHECK_NULL(quaternion);
  LuminaryVec3 r; r.x = rotation[0]; r.y = rotation[1]; r.z = rotation[2];
  lum::euler_to_quaternion(r, quaternion);
  return LUMINARY_SUCCESS;
}
// path_extend + path_apply (path.c) as the loaders use it; exposed for the check against the reference
LuminaryResult luminary_ext_path_extend(const char* base_file, const char* name, char* out, size_t out_size) {
  CHECK_NULL(base_file); CHECK_NULL(name); CHECK_NULL(out);
  const std::string r = lum::extend_path(base_file, name);
  if (r.size() + 1 > out_size) return LUMINARY_ERROR_OUT_OF_MEMORY;
  std::memcpy(out, r.c_str(), r.size() + 1);
  return LUMINARY_SUCCESS;
}
// 0: renderer settings, 1: camera. Exposed so that the rule can be checked against the reference's *_check_for_dirty.
LuminaryResult luminary_ext_change_restarts_integration(int entity, const void* input, const void* old, bool* restarts) {
  CHECK_NULL(input); CHECK_NULL(old); CHECK_NULL(restarts);
  if (entity == 0) *restarts = settings_change_restarts(*(const LuminaryRendererSettings*) input, *(const LuminaryRendererSettings*) old);
  else if (entity == 1) *restarts = camera_change_restarts(*(const LuminaryCamera*) input, *(const LuminaryCamera*) old);
  else return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  return LUMINARY_SUCCESS;
}
void* luminary_ext_get_core_context(LuminaryHost* host) {
  if (!host) return nullptr;
  ApiLock lock(host);
  if (ensure_core(host)) return nullptr;
  return host->core;
}

}  // extern "C"
