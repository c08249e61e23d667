// Host side of the output chain: ARGB8 image store with reference counts and promises, and a PNG writer.
// Behaviour follows the reference's output handler (src/luminary/host/host_output_handler.c): images are produced either for the
// recurring output (size set by luminary_host_set_output_properties) or for a request (promise) keyed by an exact sample count (0 =
// the next output); a handle stays valid while its reference count is non-zero; at least four recurring images of the current size
// are kept before the oldest unreferenced one is overwritten; an image produced for a request is neither handed out by
// luminary_host_acquire_output nor overwritten until its promise was awaited. Checked operation by operation against the reference's
// own handler (tests/test_reference_host.py); two deliberate differences: an image still being written is not handed to an awaiting
// promise, and a promise that already holds an image is not given a second one.
#pragma once

#include <cstdint>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/luminary_amd.h"

namespace lum {

struct OutputMeta { uint32_t width = 0, height = 0, sample_count = 0; float time = 0.0f; };

class OutputStore {
 public:
  static constexpr uint32_t kInvalid = LUMINARY_OUTPUT_HANDLE_INVALID;

  void set_properties(LuminaryOutputProperties p);
  LuminaryOutputProperties properties();
  uint32_t add_request(LuminaryOutputRequestProperties p);
  // requests that an output with these properties would fulfil (width, height, sample count rule of host_output_handler.c:263-276)
  std::vector<LuminaryOutputRequestProperties> pending_requests();

  // producer side: get a slot, fill data(), publish
  uint32_t begin_recurring(const OutputMeta& meta);
  LuminaryResult begin_for_request(const OutputMeta& meta, uint32_t* handle);
  uint32_t* data(uint32_t handle);
  LuminaryResult publish(uint32_t handle);

  // consumer side
  LuminaryResult acquire_recurring(uint32_t* handle);
  LuminaryResult acquire_from_promise(uint32_t promise, uint32_t* handle);
  LuminaryResult acquire(uint32_t handle);
  LuminaryResult release(uint32_t handle);
  LuminaryResult get_image(uint32_t handle, LuminaryImage* image);

 private:
  struct Object {
    std::vector<uint32_t> pixels;
    OutputMeta meta;
    bool recurring = false, populated = false, allocated = false;
    uint32_t reference_count = 0, promise_reference = kInvalid;
    uint64_t time_stamp = 0;
  };
  struct Promise { bool pending = false; LuminaryOutputRequestProperties props{0, 0, 0}; uint32_t handle = kInvalid; };
  uint32_t slot_for_write(const OutputMeta& meta, bool recurring);
  void prepare(uint32_t handle, const OutputMeta& meta, bool recurring, uint32_t promise);

  std::mutex mutex_;
  LuminaryOutputProperties props_{false, 0, 0};
  std::vector<Object> objects_;
  std::vector<Promise> promises_;
  uint64_t clock_ = 0;
};

// RGBA8 truecolour PNG of an ARGB8 image (words b | g << 8 | r << 16 | a << 24), like png_store_image (host/png.c:757-784).
LuminaryResult write_png(const char* path, const uint32_t* argb8, uint32_t width, uint32_t height, size_t ld);

// PNG reader for material textures (host/png.c:25-720 handles the same subset): non-interlaced, colour types 0, 2, 3, 4, 6 at 8 or 16
// bits (16-bit samples keep their high byte; 1/2/4-bit grey and palette images are expanded), all five scanline filters, tRNS for
// palettes, gAMA (gamma = 100000 / gAMA, png.c:541). Result: RGBA8 words (r in the low byte), rows top to bottom.
bool read_png(const std::string& path, uint32_t* width, uint32_t* height, float* gamma, std::vector<uint32_t>* rgba8, std::string* err);
// Same from memory (`name` only labels error messages): the embedded moon textures (device/device_embedded_data.c:62-92).
bool read_png_memory(const uint8_t* data, size_t size, const std::string& name, uint32_t* width, uint32_t* height, float* gamma, std::vector<uint32_t>* rgba8,
                     std::string* err);

}  // namespace lum
