// The "extra utils" part of the public API that frontends such as Mandarin Duck link against: counted host
// allocations, header-prefixed dynamic arrays, a bounded blocking queue, a ring allocator, per-thread status/timing, logging and the
// enum name tables. Behaviour follows the reference headers include/luminary/{host_memory,array,queue,ringbuffer,thread_status,log,
// name_strings}.h and what src/luminary/{host_memory,array,queue,ringbuffer,thread_status,log,name_strings}.c do with them
// (argument checks, result codes, growth policy, wrap-around rules); the implementation is C++ (std::mutex / condition_variable /
// atomic / chrono) rather than the reference's C11 threads.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <csignal>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

#include "../../../include/luminary_amd.h"

namespace {

constexpr uint64_t kMemoryMagic = 0x4D54534F484D554Cull;  // "LUMHOSTM", host_memory.c:22
constexpr uint64_t kMemoryFreed = 1337ull;
constexpr uint64_t kArrayMagic  = 0x59415252414D554Cull;  // "LUMARRAY", array.c:17
constexpr uint64_t kArrayFreed  = 420ull;

struct MemoryHeader { uint64_t magic, size, pad[6]; };
static_assert(sizeof(MemoryHeader) == 64, "allocations keep 64-byte alignment behind the header");
struct ArrayHeader { uint64_t magic, size_of_element; uint32_t num_elements, allocated; uint64_t pad[5]; };
static_assert(sizeof(ArrayHeader) == 64, "array header is 64 bytes (array.c:7-14)");

std::atomic<uint64_t> g_total_allocation{0};

#define NULL_CHECK(p) do { if (!(p)) return LUMINARY_ERROR_ARGUMENT_NULL; } while (0)
#define TRY(expr) do { const LuminaryResult r__ = (expr); if (r__ != LUMINARY_SUCCESS) return r__ | LUMINARY_ERROR_PROPAGATED; } while (0)

ArrayHeader* array_header(void* data) { return data ? reinterpret_cast<ArrayHeader*>(data) - 1 : nullptr; }
const ArrayHeader* array_header(const void* data) { return data ? reinterpret_cast<const ArrayHeader*>(data) - 1 : nullptr; }

// ---- logging state (log.c) ----
std::mutex g_log_mutex;
std::string g_log_text;
bool g_volatile_line = false;

std::string vformat(const char* format, va_list args) {
  va_list copy;
  va_copy(copy, args);
  const int n = vsnprintf(nullptr, 0, format, copy);
  va_end(copy);
  std::string out(n > 0 ? (size_t) n : 0, '\0');
  if (n > 0) vsnprintf(out.data(), (size_t) n + 1, format, args);
  return out;
}
void log_line(const char* tag, const std::string& text) { g_log_text += tag; g_log_text += text; g_log_text += '\n'; }
void console(const char* colour, const std::string& text, bool newline) {
  if (g_volatile_line) fputs("\33[2K\r", stdout);
  fprintf(stdout, "%s%s\033[0m%s", colour, text.c_str(), newline ? "\n" : "");
  fflush(stdout);
  g_volatile_line = !newline;
}

}  // namespace

struct LuminaryQueue {
  std::vector<uint8_t> buffer;
  size_t element_count = 0, element_size = 0, read_ptr = 0, write_ptr = 0, elements_in_queue = 0;
  std::mutex mutex;
  std::condition_variable cond;
  bool is_blocking = true;
};
struct LuminaryRingBuffer { void* memory; size_t size, allocated, ptr; };
struct LuminaryThreadStatus { const char* name; const char* string; uint64_t time_point; double time; };

extern "C" {

// ---- host_memory.h ----
LuminaryResult _host_malloc(void** ptr, size_t size, const char*, const char*, uint32_t) {
  NULL_CHECK(ptr);
  MemoryHeader* h = (MemoryHeader*) malloc(size + sizeof(MemoryHeader));
  if (!h) return LUMINARY_ERROR_OUT_OF_MEMORY;
  memset(h, 0, sizeof(MemoryHeader));
  h->magic = kMemoryMagic;
  h->size = size;
  g_total_allocation.fetch_add(size);
  *ptr = h + 1;
  return LUMINARY_SUCCESS;
}
LuminaryResult _host_realloc(void** ptr, size_t size, const char*, const char*, uint32_t) {
  NULL_CHECK(ptr);
  NULL_CHECK(*ptr);
  MemoryHeader* h = reinterpret_cast<MemoryHeader*>(*ptr) - 1;
  if (h->magic != kMemoryMagic) return LUMINARY_ERROR_API_EXCEPTION;
  if (h->size > g_total_allocation.load()) return LUMINARY_ERROR_MEMORY_LEAK;
  g_total_allocation.fetch_sub(h->size);
  h = (MemoryHeader*) realloc(h, size + sizeof(MemoryHeader));
  if (!h) return LUMINARY_ERROR_OUT_OF_MEMORY;
  h->size = size;
  g_total_allocation.fetch_add(size);
  *ptr = h + 1;
  return LUMINARY_SUCCESS;
}
LuminaryResult _host_free(void** ptr, const char*, const char*, uint32_t) {
  NULL_CHECK(ptr);
  NULL_CHECK(*ptr);
  MemoryHeader* h = reinterpret_cast<MemoryHeader*>(*ptr) - 1;
  if (h->magic != kMemoryMagic) return LUMINARY_ERROR_API_EXCEPTION;
  if (h->size > g_total_allocation.load()) return LUMINARY_ERROR_MEMORY_LEAK;
  h->magic = kMemoryFreed;
  g_total_allocation.fetch_sub(h->size);
  free(h);
  *ptr = nullptr;
  return LUMINARY_SUCCESS;
}
LuminaryResult luminary_ext_host_memory_in_use(uint64_t* bytes) { NULL_CHECK(bytes); *bytes = g_total_allocation.load(); return LUMINARY_SUCCESS; }

// ---- array.h ----
LuminaryResult _array_create(void** array, size_t size_of_element, uint32_t num_elements, const char* n, const char* f, uint32_t l) {
  NULL_CHECK(array);
  void* block;
  TRY(_host_malloc(&block, size_of_element * num_elements + sizeof(ArrayHeader), n, f, l));
  ArrayHeader* h = (ArrayHeader*) block;
  memset(h, 0, sizeof(ArrayHeader));
  h->magic = kArrayMagic;
  h->size_of_element = size_of_element;
  h->allocated = num_elements;
  h->num_elements = 0;
  *array = h + 1;
  return LUMINARY_SUCCESS;
}
LuminaryResult _array_resize(void** array, size_t num_elements, const char* n, const char* f, uint32_t l) {
  NULL_CHECK(array);
  NULL_CHECK(*array);
  ArrayHeader* h = array_header(*array);
  if (h->magic != kArrayMagic) return LUMINARY_ERROR_API_EXCEPTION;
  void* block = h;
  TRY(_host_realloc(&block, h->size_of_element * num_elements + sizeof(ArrayHeader), n, f, l));
  h = (ArrayHeader*) block;
  h->allocated = (uint32_t) num_elements;
  if (h->num_elements > num_elements) h->num_elements = (uint32_t) num_elements;
  *array = h + 1;
  return LUMINARY_SUCCESS;
}
LuminaryResult _array_destroy(void** array, const char* n, const char* f, uint32_t l) {
  NULL_CHECK(array);
  NULL_CHECK(*array);
  ArrayHeader* h = array_header(*array);
  if (h->magic != kArrayMagic) return LUMINARY_ERROR_API_EXCEPTION;
  h->magic = kArrayFreed;
  void* block = h;
  TRY(_host_free(&block, n, f, l));
  *array = nullptr;
  return LUMINARY_SUCCESS;
}
LuminaryResult _array_push(void** array, void* object, const char* n, const char* f, uint32_t l) {
  NULL_CHECK(array);
  NULL_CHECK(object);
  NULL_CHECK(*array);
  ArrayHeader* h = array_header(*array);
  if (h->magic != kArrayMagic) return LUMINARY_ERROR_API_EXCEPTION;
  if (h->num_elements == 0xFFFFFFFFu) return LUMINARY_ERROR_API_EXCEPTION;
  if (h->num_elements == h->allocated) {
    TRY(_array_resize(array, (size_t) h->allocated * 2, n, f, l));  // doubling, as array.c:84-87 (an array created with 0 slots cannot grow)
    h = array_header(*array);
    if (h->num_elements == h->allocated) return LUMINARY_ERROR_OUT_OF_MEMORY;
  }
  memcpy(reinterpret_cast<uint8_t*>(h + 1) + (size_t) h->num_elements * h->size_of_element, object, h->size_of_element);
  h->num_elements++;
  return LUMINARY_SUCCESS;
}
LuminaryResult array_clear(void* array) {
  NULL_CHECK(array);
  ArrayHeader* h = array_header(array);
  if (h->magic != kArrayMagic) return LUMINARY_ERROR_API_EXCEPTION;
  h->num_elements = 0;
  return LUMINARY_SUCCESS;
}
LuminaryResult _array_append(void** dst, const void* src, const char* n, const char* f, uint32_t l) {
  NULL_CHECK(src);
  NULL_CHECK(dst);
  NULL_CHECK(*dst);
  const ArrayHeader* sh = array_header(src);
  ArrayHeader* dh = array_header(*dst);
  if (sh->magic != kArrayMagic || dh->magic != kArrayMagic) return LUMINARY_ERROR_API_EXCEPTION;
  if (sh->size_of_element != dh->size_of_element) return LUMINARY_ERROR_API_EXCEPTION;
  if (dh->allocated < dh->num_elements + sh->num_elements) {
    TRY(_array_resize(dst, (size_t) dh->num_elements + sh->num_elements, n, f, l));
    dh = array_header(*dst);
  }
  memcpy(reinterpret_cast<uint8_t*>(*dst) + dh->size_of_element * dh->num_elements, src, sh->size_of_element * sh->num_elements);
  dh->num_elements += sh->num_elements;
  return LUMINARY_SUCCESS;
}
LuminaryResult _array_copy(void** dst, const void* src, const char* n, const char* f, uint32_t l) {
  NULL_CHECK(src);
  NULL_CHECK(dst);
  TRY(array_clear(*dst));
  TRY(_array_append(dst, src, n, f, l));
  return LUMINARY_SUCCESS;
}
LuminaryResult array_get_size(const void* array, size_t* size) {
  NULL_CHECK(array);
  NULL_CHECK(size);
  const ArrayHeader* h = array_header(array);
  if (h->magic != kArrayMagic) return LUMINARY_ERROR_API_EXCEPTION;
  *size = h->allocated;
  return LUMINARY_SUCCESS;
}
LuminaryResult array_get_num_elements(const void* array, uint32_t* num_elements) {
  NULL_CHECK(array);
  NULL_CHECK(num_elements);
  const ArrayHeader* h = array_header(array);
  if (h->magic != kArrayMagic) return LUMINARY_ERROR_API_EXCEPTION;
  *num_elements = h->num_elements;
  return LUMINARY_SUCCESS;
}
LuminaryResult _array_set_num_elements(void** array, uint32_t num_elements, const char* n, const char* f, uint32_t l) {
  NULL_CHECK(array);
  NULL_CHECK(*array);
  ArrayHeader* h = array_header(*array);
  if (h->magic != kArrayMagic) return LUMINARY_ERROR_API_EXCEPTION;
  if (num_elements > h->allocated) {
    TRY(_array_resize(array, num_elements, n, f, l));
    h = array_header(*array);
  }
  if (num_elements > h->num_elements)  // new slots read as zero
    memset(reinterpret_cast<uint8_t*>(*array) + h->size_of_element * h->num_elements, 0, h->size_of_element * (num_elements - h->num_elements));
  h->num_elements = num_elements;
  return LUMINARY_SUCCESS;
}

// ---- queue.h: bounded FIFO; pop_blocking waits unless blocking was switched off ----
LuminaryResult _queue_create(LuminaryQueue** queue, size_t size_of_element, size_t num_elements, const char*, const char*, uint32_t) {
  NULL_CHECK(queue);
  if (size_of_element == 0 || num_elements == 0) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  LuminaryQueue* q = new LuminaryQueue();
  q->buffer.resize(size_of_element * num_elements);
  q->element_count = num_elements;
  q->element_size = size_of_element;
  *queue = q;
  return LUMINARY_SUCCESS;
}
LuminaryResult queue_push(LuminaryQueue* q, void* object) {
  NULL_CHECK(q);
  NULL_CHECK(object);
  std::lock_guard<std::mutex> lock(q->mutex);
  if (q->elements_in_queue == q->element_count) return LUMINARY_ERROR_OUT_OF_MEMORY;
  memcpy(q->buffer.data() + q->write_ptr * q->element_size, object, q->element_size);
  q->write_ptr = (q->write_ptr + 1 >= q->element_count) ? 0 : q->write_ptr + 1;
  q->elements_in_queue++;
  q->cond.notify_one();
  return LUMINARY_SUCCESS;
}
LuminaryResult queue_push_unique(LuminaryQueue* q, void* object, LuminaryEqOp equal_operator, bool* already_queued) {
  NULL_CHECK(q);
  NULL_CHECK(object);
  NULL_CHECK(equal_operator);
  NULL_CHECK(already_queued);
  std::lock_guard<std::mutex> lock(q->mutex);
  if (q->elements_in_queue == q->element_count) return LUMINARY_ERROR_OUT_OF_MEMORY;
  bool found = false;
  size_t p = q->read_ptr;
  for (size_t k = 0; k < q->elements_in_queue && !found; k++) {
    found = equal_operator(q->buffer.data() + p * q->element_size, object);
    p = (p + 1 >= q->element_count) ? 0 : p + 1;
  }
  *already_queued = found;
  if (!found) {
    memcpy(q->buffer.data() + q->write_ptr * q->element_size, object, q->element_size);
    q->write_ptr = (q->write_ptr + 1 >= q->element_count) ? 0 : q->write_ptr + 1;
    q->elements_in_queue++;
  }
  q->cond.notify_one();
  return LUMINARY_SUCCESS;
}
static void queue_take(LuminaryQueue* q, void* object) {
  memcpy(object, q->buffer.data() + q->read_ptr * q->element_size, q->element_size);
  q->read_ptr = (q->read_ptr + 1 >= q->element_count) ? 0 : q->read_ptr + 1;
  q->elements_in_queue--;
}
LuminaryResult queue_pop(LuminaryQueue* q, void* object, bool* success) {
  NULL_CHECK(q);
  NULL_CHECK(object);
  NULL_CHECK(success);
  std::lock_guard<std::mutex> lock(q->mutex);
  *success = q->elements_in_queue != 0;
  if (*success) queue_take(q, object);
  return LUMINARY_SUCCESS;
}
LuminaryResult queue_pop_blocking(LuminaryQueue* q, void* object, bool* success) {
  NULL_CHECK(q);
  NULL_CHECK(object);
  NULL_CHECK(success);
  std::unique_lock<std::mutex> lock(q->mutex);
  q->cond.wait(lock, [q] { return q->elements_in_queue != 0 || !q->is_blocking; });
  *success = q->elements_in_queue != 0;
  if (*success) queue_take(q, object);
  return LUMINARY_SUCCESS;
}
LuminaryResult queue_set_is_blocking(LuminaryQueue* q, bool is_blocking) {
  NULL_CHECK(q);
  { std::lock_guard<std::mutex> lock(q->mutex); q->is_blocking = is_blocking; }
  q->cond.notify_all();
  return LUMINARY_SUCCESS;
}
LuminaryResult _queue_destroy(LuminaryQueue** queue, const char*, const char*, uint32_t) {
  NULL_CHECK(queue);
  NULL_CHECK(*queue);
  if ((*queue)->elements_in_queue != 0) return LUMINARY_ERROR_API_EXCEPTION;  // "Queue is not empty."
  queue_set_is_blocking(*queue, false);
  delete *queue;
  *queue = nullptr;
  return LUMINARY_SUCCESS;
}

// ---- ringbuffer.h: entries are handed out in order and released by size; an entry never wraps ----
LuminaryResult _ringbuffer_create(LuminaryRingBuffer** buffer, size_t size, const char* n, const char* f, uint32_t l) {
  NULL_CHECK(buffer);
  if (size == 0) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  LuminaryRingBuffer* b = nullptr;
  TRY(_host_malloc((void**) &b, sizeof(LuminaryRingBuffer), n, f, l));
  memset(b, 0, sizeof(*b));
  TRY(_host_malloc(&b->memory, size, n, f, l));
  b->size = size;
  *buffer = b;
  return LUMINARY_SUCCESS;
}
LuminaryResult ringbuffer_allocate_entry(LuminaryRingBuffer* b, size_t entry_size, void** entry) {
  NULL_CHECK(b);
  NULL_CHECK(entry);
  if (entry_size == 0) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  if (b->allocated + entry_size > b->size) return LUMINARY_ERROR_OUT_OF_MEMORY;
  if (b->ptr + entry_size > b->size) {
    // the tail that is skipped counts as used while this entry lives
    if (b->allocated + (b->size - b->ptr) + entry_size > b->size) return LUMINARY_ERROR_OUT_OF_MEMORY;
    *entry = b->memory;
    b->ptr = entry_size;
  }
  else {
    *entry = (uint8_t*) b->memory + b->ptr;
    b->ptr += entry_size;
  }
  b->allocated += entry_size;
  return LUMINARY_SUCCESS;
}
LuminaryResult ringbuffer_release_entry(LuminaryRingBuffer* b, size_t entry_size) {
  NULL_CHECK(b);
  if (entry_size == 0 || b->allocated < entry_size) return LUMINARY_ERROR_INVALID_API_ARGUMENT;
  b->allocated -= entry_size;
  return LUMINARY_SUCCESS;
}
LuminaryResult _ringbuffer_destroy(LuminaryRingBuffer** buffer, const char* n, const char* f, uint32_t l) {
  NULL_CHECK(buffer);
  NULL_CHECK(*buffer);
  if ((*buffer)->allocated > 0) return LUMINARY_ERROR_MEMORY_LEAK;
  TRY(_host_free(&(*buffer)->memory, n, f, l));
  TRY(_host_free((void**) buffer, n, f, l));
  return LUMINARY_SUCCESS;
}

// ---- thread_status.h: what a worker is doing and for how long (CPU clock, like the reference's clock()) ----
LuminaryResult thread_status_create(LuminaryThreadStatus** s) {
  NULL_CHECK(s);
  TRY(_host_malloc((void**) s, sizeof(LuminaryThreadStatus), "thread_status", __func__, __LINE__));
  memset(*s, 0, sizeof(LuminaryThreadStatus));
  return LUMINARY_SUCCESS;
}
LuminaryResult thread_status_set_worker_name(LuminaryThreadStatus* s, const char* name) { NULL_CHECK(s); s->name = name; return LUMINARY_SUCCESS; }
LuminaryResult thread_status_get_worker_name(LuminaryThreadStatus* s, const char** name) { NULL_CHECK(s); NULL_CHECK(name); *name = s->name; return LUMINARY_SUCCESS; }
LuminaryResult thread_status_get_string(LuminaryThreadStatus* s, const char** string) { NULL_CHECK(s); NULL_CHECK(string); *string = s->string; return LUMINARY_SUCCESS; }
static double seconds_since(uint64_t t0) { return (double) ((uint64_t) clock() - t0) / CLOCKS_PER_SEC; }
LuminaryResult thread_status_start(LuminaryThreadStatus* s, const char* string) {
  NULL_CHECK(s);
  s->time_point = (uint64_t) clock();
  if (s->time_point == 0) s->time_point = 1;  // 0 means "not running"
  s->string = string;
  return LUMINARY_SUCCESS;
}
LuminaryResult thread_status_get_time(LuminaryThreadStatus* s, double* time) {
  NULL_CHECK(s);
  NULL_CHECK(time);
  if (s->time_point != 0) s->time = seconds_since(s->time_point);
  *time = s->time;
  return LUMINARY_SUCCESS;
}
LuminaryResult thread_status_stop(LuminaryThreadStatus* s) {
  NULL_CHECK(s);
  s->time = (s->time_point != 0) ? seconds_since(s->time_point) : 0.0;
  s->time_point = 0;
  s->string = nullptr;
  return LUMINARY_SUCCESS;
}
LuminaryResult thread_status_destroy(LuminaryThreadStatus** s) { NULL_CHECK(s); NULL_CHECK(*s); TRY(_host_free((void**) s, "thread_status", __func__, __LINE__)); return LUMINARY_SUCCESS; }

// ---- log.h: everything is kept for luminary_write_log; info/warn/error/crash also go to the console ----
void luminary_print_log(const char* format, ...) {
  va_list a; va_start(a, format); const std::string t = vformat(format, a); va_end(a);
  std::lock_guard<std::mutex> lock(g_log_mutex);
  log_line("[LOG] ", t);
}
void luminary_print_info(bool log, const char* format, ...) {
  va_list a; va_start(a, format); const std::string t = vformat(format, a); va_end(a);
  std::lock_guard<std::mutex> lock(g_log_mutex);
  if (log) log_line("[INFO] ", t);
  console("\x1B[1m", t, true);
}
void luminary_print_info_inline(bool log, const char* format, ...) {
  va_list a; va_start(a, format); const std::string t = vformat(format, a); va_end(a);
  std::lock_guard<std::mutex> lock(g_log_mutex);
  if (log) log_line("[INFO] ", t);
  console("\x1B[1m", t, false);
}
void luminary_print_warn(const char* format, ...) {
  va_list a; va_start(a, format); const std::string t = vformat(format, a); va_end(a);
  std::lock_guard<std::mutex> lock(g_log_mutex);
  log_line("[WARN] ", t);
  console("\x1B[93m\x1B[1m", t, true);
}
void luminary_print_error(const char* format, ...) {
  va_list a; va_start(a, format); const std::string t = vformat(format, a); va_end(a);
  std::lock_guard<std::mutex> lock(g_log_mutex);
  log_line("[ERR] ", t);
  console("\x1B[91m\x1B[1m", t, true);
}
void luminary_print_crash(const char* format, ...) {
  va_list a; va_start(a, format); const std::string t = vformat(format, a); va_end(a);
  {
    std::lock_guard<std::mutex> lock(g_log_mutex);
    log_line("[CRASH] ", t);
    console("\x1B[95m\x1B[1m", t, true);
  }
  luminary_write_log();
  exit(SIGABRT);  // log.c:111-116 (without the interactive "press enter")
}
void luminary_write_log(void) {
  std::lock_guard<std::mutex> lock(g_log_mutex);
  FILE* f = fopen("luminary.log", "wb");
  if (!f) return;
  fwrite(g_log_text.data(), 1, g_log_text.size(), f);
  fclose(f);
}
LuminaryResult luminary_ext_get_log(const char** text, size_t* length) {
  NULL_CHECK(text);
  std::lock_guard<std::mutex> lock(g_log_mutex);
  *text = g_log_text.c_str();
  if (length) *length = g_log_text.size();
  return LUMINARY_SUCCESS;
}

// ---- embedded files: the reference embeds its data with the Ceb tool and frontends fetch them by name
// (device/device_embedded.c:10-14, :1075-1093; src/mandarin_duck/display.c:219). info = 0 on success, non-zero = unknown name. ----
extern const unsigned char lum_embedded_bluenoise_2d[];
extern const unsigned char lum_embedded_bluenoise_2d_end[];
extern const unsigned char lum_embedded_bluenoise_1d[];
extern const unsigned char lum_embedded_bluenoise_1d_end[];
void ceb_access(const char* name, void** ptr, int64_t* lmem, uint64_t* info) {
  struct Entry { const char* name; const unsigned char* begin; const unsigned char* end; };
  const Entry files[] = {{"bluenoise_2D.bin", lum_embedded_bluenoise_2d, lum_embedded_bluenoise_2d_end},
                         {"bluenoise_1D.bin", lum_embedded_bluenoise_1d, lum_embedded_bluenoise_1d_end}};
  if (info) *info = 1;
  if (ptr) *ptr = nullptr;
  if (lmem) *lmem = 0;
  if (!name) return;
  for (const Entry& e : files) {
    if (strcmp(name, e.name) != 0) continue;
    if (ptr) *ptr = const_cast<unsigned char*>(e.begin);
    if (lmem) *lmem = (int64_t) (e.end - e.begin);
    if (info) *info = 0;
    return;
  }
}

// ---- name_strings.h ----
const char* const luminary_strings_shading_mode[LUMINARY_SHADING_MODE_COUNT] = {"None", "Albedo", "Depth", "Normal", "Identification", "Lights"};
const char* const luminary_strings_adaptive_sampling_output_mode[LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_COUNT] = {"Beauty", "Rel Variance", "Rel Error",
                                                                                                                  "Sample Distribution"};
const char* const luminary_strings_filter[LUMINARY_FILTER_COUNT] = {"None", "Gray", "Sepia", "Gameboy", "2 Bit Gray", "CRT", "Black & White"};
const char* const luminary_strings_tonemap[LUMINARY_TONEMAP_COUNT] = {"None", "ACES", "Reinhard", "Uncharted 2", "Agx", "Agx Punchy", "Agx Custom"};
const char* const luminary_strings_aperture[LUMINARY_APERTURE_COUNT] = {"Round", "Bladed"};
const char* const luminary_strings_jerlov_water_type[LUMINARY_JERLOV_WATER_TYPE_COUNT] = {"I", "IA", "IB", "II", "III", "1C", "3C", "5C", "7C", "9C"};
const char* const luminary_strings_sky_mode[LUMINARY_SKY_MODE_COUNT] = {"Default", "HDRI", "Constant Color"};
const char* const luminary_strings_material_base_substrate[LUMINARY_MATERIAL_BASE_SUBSTRATE_COUNT] = {"Opaque", "Translucent"};

}  // extern "C"
