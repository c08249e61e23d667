#include "output.h"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lum {

void OutputStore::set_properties(LuminaryOutputProperties p) { std::lock_guard<std::mutex> l(mutex_); props_ = p; }
LuminaryOutputProperties OutputStore::properties() { std::lock_guard<std::mutex> l(mutex_); return props_; }

uint32_t OutputStore::add_request(LuminaryOutputRequestProperties p) {
  std::lock_guard<std::mutex> l(mutex_);
  uint32_t id = kInvalid;
  for (uint32_t i = 0; i < promises_.size(); i++)
    if (!promises_[i].pending) { id = i; break; }
  if (id == kInvalid) { promises_.emplace_back(); id = (uint32_t) promises_.size() - 1; }
  promises_[id].pending = true;
  promises_[id].props = p;
  promises_[id].handle = kInvalid;
  return id;
}

std::vector<LuminaryOutputRequestProperties> OutputStore::pending_requests() {
  std::lock_guard<std::mutex> l(mutex_);
  std::vector<LuminaryOutputRequestProperties> out;
  for (const Promise& p : promises_)
    if (p.pending && p.handle == kInvalid) out.push_back(p.props);
  return out;
}

// host_output_handler.c:195-236: the oldest unreferenced slot; slots of another size go first; keep four valid recurring images
uint32_t OutputStore::slot_for_write(const OutputMeta& meta, bool recurring) {
  uint32_t selected = kInvalid, valid = 0;
  uint64_t earliest = UINT64_MAX;
  bool selected_is_valid = true;
  for (uint32_t i = 0; i < objects_.size(); i++) {
    const Object& o = objects_[i];
    if (o.reference_count || o.promise_reference != kInvalid) continue;
    const bool still_valid = (o.meta.width == meta.width && o.meta.height == meta.height) || !recurring;
    if (still_valid) {
      valid++;
      if (o.time_stamp >= earliest) continue;
    }
    selected = i;
    earliest = o.time_stamp;
    if (!still_valid) { selected_is_valid = false; break; }
  }
  if (valid < 4 && selected_is_valid) selected = kInvalid;
  if (selected == kInvalid) { objects_.emplace_back(); selected = (uint32_t) objects_.size() - 1; }
  return selected;
}

void OutputStore::prepare(uint32_t handle, const OutputMeta& meta, bool recurring, uint32_t promise) {
  Object& o = objects_[handle];
  o.pixels.resize((size_t) meta.width * meta.height);
  o.meta = meta;
  o.recurring = recurring;
  o.populated = false;
  o.allocated = true;
  o.reference_count = 1;  // held by the producer until publish()
  o.promise_reference = promise;
}

uint32_t OutputStore::begin_recurring(const OutputMeta& meta) {
  std::lock_guard<std::mutex> l(mutex_);
  const uint32_t h = slot_for_write(meta, true);
  prepare(h, meta, true, kInvalid);
  return h;
}

LuminaryResult OutputStore::begin_for_request(const OutputMeta& meta, uint32_t* handle) {
  std::lock_guard<std::mutex> l(mutex_);
  uint32_t promise = kInvalid;
  for (uint32_t i = 0; i < promises_.size(); i++) {
    const Promise& p = promises_[i];
    if (!p.pending || p.handle != kInvalid) continue;
    if (p.props.width != meta.width || p.props.height != meta.height) continue;
    if (p.props.sample_count > 0 && p.props.sample_count != meta.sample_count) continue;
    promise = i;
    break;
  }
  if (promise == kInvalid) return LUMINARY_ERROR_API_EXCEPTION;  // "Tried to create an output for a request that has no promise."
  const uint32_t h = slot_for_write(meta, false);
  prepare(h, meta, false, promise);
  promises_[promise].handle = h;
  *handle = h;
  return LUMINARY_SUCCESS;
}

uint32_t* OutputStore::data(uint32_t handle) { std::lock_guard<std::mutex> l(mutex_); return handle < objects_.size() ? objects_[handle].pixels.data() : nullptr; }

LuminaryResult OutputStore::publish(uint32_t handle) {
  std::lock_guard<std::mutex> l(mutex_);
  if (handle >= objects_.size() || objects_[handle].reference_count == 0) return LUMINARY_ERROR_API_EXCEPTION;
  objects_[handle].populated = true;
  objects_[handle].reference_count--;
  objects_[handle].time_stamp = ++clock_;
  return LUMINARY_SUCCESS;
}

LuminaryResult OutputStore::acquire_recurring(uint32_t* handle) {
  std::lock_guard<std::mutex> l(mutex_);
  uint32_t latest = kInvalid;
  uint64_t stamp = 0;
  for (uint32_t i = 0; i < objects_.size(); i++) {
    const Object& o = objects_[i];
    if (!o.populated || o.time_stamp <= stamp) continue;
    if (o.meta.width != props_.width || o.meta.height != props_.height) continue;
    if (o.promise_reference != kInvalid) continue;  // an output made for a request is not handed out here until its promise was awaited (host_output_handler.c:104-106)
    latest = i;
    stamp = o.time_stamp;
  }
  if (latest != kInvalid) objects_[latest].reference_count++;
  *handle = latest;
  return LUMINARY_SUCCESS;
}

LuminaryResult OutputStore::acquire_from_promise(uint32_t promise, uint32_t* handle) {
  if (promise == kInvalid) { *handle = kInvalid; return LUMINARY_SUCCESS; }
  std::lock_guard<std::mutex> l(mutex_);
  if (promise >= promises_.size()) return LUMINARY_ERROR_API_EXCEPTION;
  Promise& p = promises_[promise];
  uint32_t h = p.handle;
  if (h != kInvalid && !objects_[h].populated) h = kInvalid;
  if (h != kInvalid) {
    objects_[h].reference_count++;
    objects_[h].promise_reference = kInvalid;
    p.pending = false;
  }
  *handle = h;
  return LUMINARY_SUCCESS;
}

LuminaryResult OutputStore::acquire(uint32_t handle) {
  if (handle == kInvalid) return LUMINARY_SUCCESS;
  std::lock_guard<std::mutex> l(mutex_);
  if (handle >= objects_.size() || objects_[handle].reference_count == 0) return LUMINARY_ERROR_API_EXCEPTION;
  objects_[handle].reference_count++;
  return LUMINARY_SUCCESS;
}

LuminaryResult OutputStore::release(uint32_t handle) {
  if (handle == kInvalid) return LUMINARY_SUCCESS;
  std::lock_guard<std::mutex> l(mutex_);
  if (handle >= objects_.size() || objects_[handle].reference_count == 0) return LUMINARY_ERROR_API_EXCEPTION;
  objects_[handle].reference_count--;
  return LUMINARY_SUCCESS;
}

LuminaryResult OutputStore::get_image(uint32_t handle, LuminaryImage* image) {
  if (handle == kInvalid) { std::memset(image, 0, sizeof(*image)); return LUMINARY_SUCCESS; }
  std::lock_guard<std::mutex> l(mutex_);
  if (handle >= objects_.size() || objects_[handle].reference_count == 0) return LUMINARY_ERROR_API_EXCEPTION;
  Object& o = objects_[handle];
  image->buffer = reinterpret_cast<uint8_t*>(o.pixels.data());
  image->width = o.meta.width;
  image->height = o.meta.height;
  image->ld = o.meta.width;
  image->meta_data.time = o.meta.time;
  image->meta_data.sample_count = o.meta.sample_count;
  return LUMINARY_SUCCESS;
}

// ---- PNG: signature, IHDR, one IDAT (zlib stream of filter-type-0 scanlines), IEND ----
namespace {
void put_u32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void put_chunk(std::vector<uint8_t>& file, const char type[4], const uint8_t* payload, size_t n) {
  put_u32(file, (uint32_t) n);
  const size_t start = file.size();
  file.insert(file.end(), type, type + 4);
  if (n) file.insert(file.end(), payload, payload + n);
  put_u32(file, (uint32_t) crc32(0L, file.data() + start, (uInt) (n + 4)));
}
}  // namespace

LuminaryResult write_png(const char* path, const uint32_t* argb8, uint32_t width, uint32_t height, size_t ld) {
  if (!path || !argb8) return LUMINARY_ERROR_ARGUMENT_NULL;
  if (width == 0 || height == 0) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  std::vector<uint8_t> raw((size_t) height * (1 + 4 * (size_t) width));
  for (uint32_t y = 0; y < height; y++) {
    uint8_t* row = raw.data() + (size_t) y * (1 + 4 * (size_t) width);
    row[0] = 0;  // filter type None
    for (uint32_t x = 0; x < width; x++) {
      const uint32_t w = argb8[x + (size_t) y * ld];
      row[1 + 4 * x + 0] = (w >> 16) & 0xFF;
      row[1 + 4 * x + 1] = (w >> 8) & 0xFF;
      row[1 + 4 * x + 2] = w & 0xFF;
      row[1 + 4 * x + 3] = w >> 24;
    }
  }
  uLongf bound = compressBound((uLong) raw.size());
  std::vector<uint8_t> z(bound);
  if (compress2(z.data(), &bound, raw.data(), (uLong) raw.size(), 6) != Z_OK) return LUMINARY_ERROR_C_STD;
  std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  std::vector<uint8_t> ihdr;
  put_u32(ihdr, width); put_u32(ihdr, height);
  ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8 bit, truecolour + alpha
  put_chunk(file, "IHDR", ihdr.data(), ihdr.size());
  put_chunk(file, "IDAT", z.data(), bound);
  put_chunk(file, "IEND", nullptr, 0);
  FILE* f = std::fopen(path, "wb");
  if (!f) return LUMINARY_ERROR_C_STD;
  const size_t written = std::fwrite(file.data(), 1, file.size(), f);
  std::fclose(f);
  return written == file.size() ? LUMINARY_SUCCESS : LUMINARY_ERROR_C_STD;
}

namespace {
uint32_t be32(const uint8_t* p) { return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3]; }
uint8_t paeth(int a, int b, int c) {
  const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
  return (uint8_t) ((pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c));
}
}  // namespace

bool read_png(const std::string& path, uint32_t* width, uint32_t* height, float* gamma, std::vector<uint32_t>* rgba8, std::string* err) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) { *err = "File " + path + " could not be opened!"; return false; }
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(f);
  return read_png_memory(file.data(), file.size(), path, width, height, gamma, rgba8, err);
}

bool read_png_memory(const uint8_t* data, size_t size, const std::string& path, uint32_t* width, uint32_t* height, float* gamma, std::vector<uint32_t>* rgba8,
                     std::string* err) {
  const std::vector<uint8_t> file(data, data + size);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (file.size() < 33 || std::memcmp(file.data(), sig, 8) != 0) { *err = path + " is not a PNG file"; return false; }
  uint32_t w = 0, h = 0, depth = 0, colour = 0, interlace = 0;
  bool have_ihdr = false;
  std::vector<uint8_t> idat, palette, trns;
  *gamma = 1.0f;
  size_t pos = 8;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const char* type = (const char*) &file[pos + 4];
    if (pos + 12 + (size_t) len > file.size()) { *err = path + ": truncated chunk"; return false; }
    const uint8_t* body = &file[pos + 8];
    if (be32(body + len) != (uint32_t) crc32(0L, &file[pos + 4], (uInt) (len + 4))) { *err = path + ": chunk checksum mismatch"; return false; }
    if (!std::memcmp(type, "IHDR", 4) && len >= 13) { w = be32(body); h = be32(body + 4); depth = body[8]; colour = body[9]; interlace = body[12]; have_ihdr = true; }
    else if (!std::memcmp(type, "PLTE", 4)) palette.assign(body, body + len);
    else if (!std::memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
    else if (!std::memcmp(type, "gAMA", 4) && len == 4 && be32(body) != 0) *gamma = 100000.0f / (float) be32(body);
    else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if (!std::memcmp(type, "IEND", 4)) break;
    pos += 12 + (size_t) len;
  }
  static const int channels_of[7] = {1, 0, 3, 1, 2, 0, 4};
  // bit depths the PNG specification allows per colour type (section 11.2.2): 1, 2, 4 only for greyscale and palette images, 16 not for palettes
  const bool depth_ok = depth == 8 || (depth == 16 && colour != 3) || ((depth == 1 || depth == 2 || depth == 4) && (colour == 0 || colour == 3));
  if (!have_ihdr || w == 0 || h == 0 || colour > 6 || channels_of[colour] == 0 || interlace != 0 || !depth_ok) {
    *err = path + ": unsupported PNG layout (no header, interlaced or a bit depth the format does not allow)";
    return false;
  }
  // textures named in a scene's .mtl are untrusted input: bound the size before anything is allocated from it (16384 is also the limit
  // of the device texture table's 14-bit coordinates elsewhere in the pipeline)
  if (w > 16384 || h > 16384) { *err = path + ": image larger than 16384 x 16384"; return false; }
  const uint32_t channels = (uint32_t) channels_of[colour];
  const size_t bits_per_pixel = (size_t) channels * depth, bpp = std::max<size_t>(1, bits_per_pixel / 8), stride = (bits_per_pixel * w + 7) / 8;
  std::vector<uint8_t> raw;
  try { raw.resize((stride + 1) * (size_t) h); rgba8->assign((size_t) w * h, 0); }
  catch (const std::bad_alloc&) { *err = path + ": out of memory for a " + std::to_string(w) + " x " + std::to_string(h) + " image"; return false; }
  uLongf raw_len = (uLongf) raw.size();
  if (uncompress(raw.data(), &raw_len, idat.data(), (uLong) idat.size()) != Z_OK || raw_len != raw.size()) { *err = path + ": corrupt image data"; return false; }
  // undo the scanline filters in place (PNG specification, section 9)
  std::vector<uint8_t> prev(stride, 0);
  for (uint32_t y = 0; y < h; y++) {
    uint8_t* row = &raw[(stride + 1) * (size_t) y];
    const uint8_t filter = row[0];
    uint8_t* px = row + 1;
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? px[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      switch (filter) {
        case 1: px[i] = (uint8_t) (px[i] + a); break;
        case 2: px[i] = (uint8_t) (px[i] + b); break;
        case 3: px[i] = (uint8_t) (px[i] + ((a + b) >> 1)); break;
        case 4: px[i] = (uint8_t) (px[i] + paeth(a, b, c)); break;
        default: break;
      }
    }
    std::memcpy(prev.data(), px, stride);
  }
  for (uint32_t y = 0; y < h; y++) {
    const uint8_t* px = &raw[(stride + 1) * (size_t) y + 1];
    for (uint32_t x = 0; x < w; x++) {
      uint8_t s[4] = {0, 0, 0, 255};
      if (depth < 8) {  // packed grey or palette index
        const size_t bit = (size_t) x * depth;
        const uint32_t v = (px[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
        s[0] = (colour == 3) ? (uint8_t) v : (uint8_t) (v * 255u / ((1u << depth) - 1u));
      }
      else for (uint32_t c = 0; c < channels; c++) s[c] = px[((size_t) x * channels + c) * (depth / 8)];  // high byte of 16-bit samples
      uint8_t r, g, b, a = 255;
      if (colour == 0) { r = g = b = s[0]; }
      else if (colour == 4) { r = g = b = s[0]; a = s[1]; }
      else if (colour == 3) {
        const size_t idx = s[0];
        if (3 * idx + 2 >= palette.size()) { *err = path + ": palette index out of range"; return false; }
        r = palette[3 * idx]; g = palette[3 * idx + 1]; b = palette[3 * idx + 2];
        if (idx < trns.size()) a = trns[idx];
      }
      else { r = s[0]; g = s[1]; b = s[2]; if (colour == 6) a = s[3]; }
      (*rgba8)[(size_t) y * w + x] = (uint32_t) r | ((uint32_t) g << 8) | ((uint32_t) b << 16) | ((uint32_t) a << 24);
    }
  }
  *width = w; *height = h;
  return true;
}

}  // namespace lum
