// Test support: a C face on lum::OutputStore (luminary_amd/csrc/host/output.cpp), so that tests/test_reference_host.py can drive it and
// the reference's host_output_handler.c (oracle/_ref) with the same operations. Compiled by the test together with output.cpp.
#include "../../luminary_amd/csrc/host/output.h"

extern "C" {
void* os_create() { return new lum::OutputStore(); }
void os_destroy(void* s) { delete (lum::OutputStore*) s; }
void os_set_properties(void* s, int enabled, uint32_t w, uint32_t h) { LuminaryOutputProperties p; p.enabled = enabled != 0; p.width = w; p.height = h; ((lum::OutputStore*) s)->set_properties(p); }
uint32_t os_add_request(void* s, uint32_t sample_count, uint32_t w, uint32_t h) {
  LuminaryOutputRequestProperties p; p.sample_count = sample_count; p.width = w; p.height = h;
  return ((lum::OutputStore*) s)->add_request(p);
}
static lum::OutputMeta meta(uint32_t w, uint32_t h, uint32_t sc) { lum::OutputMeta m; m.width = w; m.height = h; m.sample_count = sc; m.time = 1.0f; return m; }
uint32_t os_begin_recurring(void* s, uint32_t w, uint32_t h, uint32_t sc) { return ((lum::OutputStore*) s)->begin_recurring(meta(w, h, sc)); }
uint64_t os_begin_for_request(void* s, uint32_t w, uint32_t h, uint32_t sc, uint32_t* handle) { return ((lum::OutputStore*) s)->begin_for_request(meta(w, h, sc), handle); }
uint64_t os_publish(void* s, uint32_t handle) { return ((lum::OutputStore*) s)->publish(handle); }
uint64_t os_acquire_recurring(void* s, uint32_t* handle) { return ((lum::OutputStore*) s)->acquire_recurring(handle); }
uint64_t os_acquire_from_promise(void* s, uint32_t promise, uint32_t* handle) { return ((lum::OutputStore*) s)->acquire_from_promise(promise, handle); }
uint64_t os_acquire(void* s, uint32_t handle) { return ((lum::OutputStore*) s)->acquire(handle); }
uint64_t os_release(void* s, uint32_t handle) { return ((lum::OutputStore*) s)->release(handle); }
uint64_t os_get_image(void* s, uint32_t handle, uint32_t out[3]) {
  LuminaryImage img;
  const uint64_t rc = ((lum::OutputStore*) s)->get_image(handle, &img);
  if (rc == 0) { out[0] = img.width; out[1] = img.height; out[2] = img.meta_data.sample_count; }
  return rc;
}
}
