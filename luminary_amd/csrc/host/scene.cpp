#include "scene.h"

#include <mutex>
#include <stdlib.h>

#include "output.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>

namespace lum {

// ---------------------------------------------------------------------------------------------------------------------
// Defaults
// ---------------------------------------------------------------------------------------------------------------------

void default_settings(LuminaryRendererSettings* s) {  // settings.c:6-28
  std::memset(s, 0, sizeof(*s));
  s->width = 2560; s->height = 1440; s->max_ray_depth = 4; s->bridge_max_num_vertices = 15;
  s->undersampling = 2; s->supersampling = 1; s->enable_adaptive_sampling = true;
  s->adaptive_sampling_max_sampling_rate = 256; s->adaptive_sampling_avg_sampling_rate = 2; s->adaptive_sampling_update_interval = 64;
  s->adaptive_sampling_exposure_aware = true;
  s->adaptive_sampling_output_mode = LUMINARY_ADAPTIVE_SAMPLING_OUTPUT_MODE_BEAUTY;
  s->shading_mode = LUMINARY_SHADING_MODE_DEFAULT;
  s->region_x = 0.0f; s->region_y = 0.0f; s->region_width = 1.0f; s->region_height = 1.0f;
}

void default_camera(LuminaryCamera* c) {  // camera.c:7-66
  std::memset(c, 0, sizeof(*c));
  c->aperture_shape = LUMINARY_APERTURE_ROUND; c->aperture_blade_count = 7; c->exposure = 0.0f; c->bloom_blend = 0.01f;
  c->dithering = true; c->tonemap = LUMINARY_TONEMAP_AGX; c->use_local_error_minimization = false;
  c->agx_custom_slope = 1.0f; c->agx_custom_power = 1.0f; c->agx_custom_saturation = 1.0f;
  c->filter = LUMINARY_FILTER_NONE; c->wasd_speed = 1.0f; c->mouse_speed = 1.0f; c->smooth_movement = false; c->smoothing_factor = 0.1f;
  c->purkinje = true; c->purkinje_kappa1 = 0.2f; c->purkinje_kappa2 = 0.29f; c->russian_roulette_threshold = 0.1f;
  c->use_color_correction = false; c->film_grain = 0.0f; c->camera_scale = 1.0f; c->object_distance = 1.0f; c->use_physical_camera = false;
  c->thin_lens.fov = 1.0f; c->thin_lens.aperture_size = 0.0f;
  const float scale = 50.53f / 100.0f;
  const float last_vertex_point = 88.18f * scale;
  c->physical.focal_length = 50.53f;
  c->physical.front_focal_point = last_vertex_point - (-22.69f);
  c->physical.back_focal_point = last_vertex_point - 65.18f;
  c->physical.front_principal_point = last_vertex_point - 27.84f;
  c->physical.back_principal_point = last_vertex_point - 14.65f;
  c->physical.aperture_point = last_vertex_point - 28.02f;
  c->physical.aperture_diameter = 21.411f;
  c->physical.exit_pupil_point = 0.0f;
  c->physical.exit_pupil_diameter = 28.0f;
  c->physical.image_plane_distance = 65.18f - last_vertex_point;
  c->physical.sensor_width = 20.0f;
}

// jendersie_eon_phase_parameters, cuda/math.cuh:1189-1232: the four parameters depend on the droplet diameter only, so they are
// evaluated once here (the reference evaluates them per ray on the device) and travel with the scene.
void jendersie_eon_parameters(float d, float out[4]) {
  float g_hg, g_d, alpha, w_d;
  if (d >= 5.0f && d <= 50.0f) {
    g_hg = std::exp(-0.0990567f / (d - 1.67154f));
    g_d = std::exp(-(2.20679f / (d + 3.91029f)) - 0.428934f);
    alpha = std::exp(3.62489f - (8.29288f / (d + 5.52825f)));
    w_d = std::exp(-(0.599085f / (d - 0.641583f)) - 0.665888f);
  }
  else if (d >= 1.5f && d < 5.0f) {
    const float l = std::log(d), ll = std::log(l);
    g_hg = 0.0604931f * ll + 0.940256f;
    g_d = 0.500411f - (0.081287f / (-2.0f * l + std::tan(l) + 1.27551f));
    alpha = 7.30354f * l + 6.31675f;
    w_d = 0.026914f * (l - std::cos(5.68947f * (ll - 0.0292149f))) + 0.376475f;
  }
  else if (d >= 0.1f && d < 1.5f) {
    const float l = std::log(d);
    g_hg = 0.862f - 0.143f * l * l;
    g_d = 0.379685f * std::cos(1.19692f * std::cos(((l - 0.238604f) * (l + 1.00667f)) / (0.507522f - 0.15677f * l)) + 1.37932f * l + 0.0625835f) + 0.344213f;
    alpha = 250.0f;
    w_d = 0.146209f * std::cos(3.38707f * l + 2.11193f) + 0.316072f + 0.0778917f * l;
  }
  else if (d < 0.1f) {
    g_hg = 13.8f * d * d;
    g_d = 1.1456f * d * std::sin(9.29044f * d);
    alpha = 250.0f;
    w_d = 0.252977f - 312.983f * std::pow(d, 4.3f);
  }
  else { g_hg = 0.0f; g_d = 0.0f; alpha = 0.0f; w_d = 0.0f; }  // beyond 50 um the reference leaves the values unset
  out[0] = g_hg; out[1] = g_d; out[2] = alpha; out[3] = w_d;
}

// _sky_stars_generate, device/device_sky.c:484-547: `count` stars from the C library's generator seeded with `seed`, bucketed into a
// 64 x 32 grid over (azimuth, altitude) in steps of 0.1 rad. srand()/rand() are glibc's random(); random_r with a private state
// yields the same sequence without touching the process-wide generator.
bool generate_stars(uint32_t seed, uint32_t count, std::vector<float>* stars, std::vector<uint32_t>* offsets) {
  constexpr uint32_t kGridX = 64, kGridY = 32;
  constexpr float kRefPi = 3.141592653589f;  // utils.h:14
  struct random_data state;
  char state_buf[128];
  std::memset(&state, 0, sizeof(state));
  std::memset(state_buf, 0, sizeof(state_buf));
  if (initstate_r(seed, state_buf, sizeof(state_buf), &state) != 0) return false;
  auto rnd = [&]() { int32_t r = 0; random_r(&state, &r); return (float) (((double) r) / RAND_MAX); };
  struct Star { float altitude, azimuth, radius, intensity; };
  std::vector<Star> buffer(count);
  std::vector<uint32_t> counts(kGridX * kGridY, 0);
  auto cell = [&](const Star& s) { return (uint32_t) (s.azimuth * 10.0f) + (uint32_t) ((s.altitude + kRefPi * 0.5f) * 10.0f) * kGridX; };
  for (uint32_t i = 0; i < count; i++) {
    Star s;
    s.altitude = -kRefPi * 0.5f + kRefPi * (1.0f - std::sqrt(rnd()));
    s.azimuth = 2.0f * kRefPi * rnd();
    s.radius = 0.0001f + 0.0004f * (1.0f - std::sqrt(rnd()));
    s.intensity = 0.0001f + 0.0015f * (0.1f + 0.9f * (1.0f - std::sqrt(rnd())));
    const uint32_t x = (uint32_t) (s.azimuth * 10.0f), y = (uint32_t) ((s.altitude + kRefPi * 0.5f) * 10.0f);
    if (x >= kGridX || y >= kGridY) return false;  // "Star generation exception." in the reference
    counts[cell(s)]++;
    buffer[i] = s;
  }
  offsets->assign(kGridX * kGridY + 1, 0);
  uint32_t offset = 0;
  for (uint32_t i = 0; i < kGridX * kGridY; i++) { (*offsets)[i] = offset; offset += counts[i]; counts[i] = 0; }
  (*offsets)[kGridX * kGridY] = offset;
  stars->assign(4 * (size_t) count + 4, 0.0f);
  for (uint32_t i = 0; i < count; i++) {
    const uint32_t p = cell(buffer[i]);
    const uint32_t o = (*offsets)[p] + counts[p]++;
    std::memcpy(&(*stars)[4 * (size_t) o], &buffer[i], sizeof(Star));
  }
  return true;
}

extern "C" const unsigned char lum_embedded_bridge_lut[];  // 16-byte aligned (embed.S)
extern "C" const unsigned char lum_embedded_moon_albedo[];
extern "C" const unsigned char lum_embedded_moon_albedo_end[];
extern "C" const unsigned char lum_embedded_moon_normal[];
extern "C" const unsigned char lum_embedded_moon_normal_end[];
// The embedded moon textures, decoded once.
bool moon_textures(const HostTexture* out[2]) {
  static HostTexture tex[2];
  static int state = 0;  // 0 untried, 1 ok, -1 failed
  static std::mutex mutex;
  std::lock_guard<std::mutex> lock(mutex);
  if (state == 0) {
    std::string err;
    const bool a = read_png_memory(lum_embedded_moon_albedo, (size_t) (lum_embedded_moon_albedo_end - lum_embedded_moon_albedo), "moon_albedo.png", &tex[0].width,
                                   &tex[0].height, &tex[0].gamma, &tex[0].texels, &err);
    const bool n = read_png_memory(lum_embedded_moon_normal, (size_t) (lum_embedded_moon_normal_end - lum_embedded_moon_normal), "moon_normal.png", &tex[1].width,
                                   &tex[1].height, &tex[1].gamma, &tex[1].texels, &err);
    state = (a && n) ? 1 : -1;
  }
  out[0] = &tex[0]; out[1] = &tex[1];
  return state == 1;
}

void default_sky(LuminarySky* s) {  // sky.c:6-41
  std::memset(s, 0, sizeof(*s));
  s->geometry_offset.y = 0.1f; s->altitude = 0.5f; s->azimuth = 3.141f; s->moon_altitude = -0.5f;
  s->sun_strength = 1.0f; s->base_density = 1.0f; s->rayleigh_density = 1.0f; s->mie_density = 1.0f; s->ozone_density = 1.0f;
  s->ground_visibility = 60.0f; s->mie_diameter = 2.0f; s->ozone_layer_thickness = 15.0f; s->rayleigh_falloff = 8.0f; s->mie_falloff = 1.7f;
  s->multiscattering_factor = 1.0f; s->steps = 40; s->ozone_absorption = true; s->aerial_perspective = false;
  s->hdri_dim = 2048; s->hdri_samples = 32; s->stars_seed = 0; s->stars_count = 10000; s->stars_intensity = 1.0f;
  s->constant_color.r = 1.0f; s->constant_color.g = 1.0f; s->constant_color.b = 1.0f;
  s->mode = LUMINARY_SKY_MODE_DEFAULT;
}

void default_material(LuminaryMaterial* m) {  // material.c:5-29
  std::memset(m, 0, sizeof(*m));
  m->base_substrate = LUMINARY_MATERIAL_BASE_SUBSTRATE_OPAQUE;
  m->albedo.r = 0.9f; m->albedo.g = 0.9f; m->albedo.b = 0.9f; m->albedo.a = 0.9f;
  m->emission_scale = 1.0f; m->roughness = 0.7f; m->roughness_clamp = 0.25f; m->refraction_index = 1.0f;
  m->normal_map_is_compressed = true;
  m->albedo_tex = m->luminance_tex = m->roughness_tex = m->metallic_tex = m->normal_tex = 0xFFFF;
}

void default_ocean(LuminaryOcean* o) {  // ocean.c:6-22
  std::memset(o, 0, sizeof(*o));
  o->amplitude = 0.2f; o->frequency = 0.12f; o->refractive_index = 1.333f; o->water_type = LUMINARY_JERLOV_WATER_TYPE_IB;
  o->caustics_ris_sample_count = 32; o->caustics_domain_scale = 0.5f;
}

void default_cloud(LuminaryCloud* c) {  // cloud.c:6-52
  std::memset(c, 0, sizeof(*c));
  c->steps = 96; c->shadow_steps = 8; c->atmosphere_scattering = true; c->seed = 1;
  c->noise_shape_scale = 1.0f; c->noise_detail_scale = 1.0f; c->noise_weather_scale = 1.0f; c->octaves = 9;
  c->droplet_diameter = 25.0f; c->density = 1.0f;
  auto layer = [](LuminaryCloudLayer& l, float hmax, float hmin, float wind) {
    l.active = true; l.height_max = hmax; l.height_min = hmin; l.coverage = 1.0f; l.coverage_min = 0.0f; l.type = 1.0f; l.type_min = 0.0f;
    l.wind_speed = wind; l.wind_angle = 0.0f;
  };
  layer(c->low, 5.0f, 1.5f, 2.5f); layer(c->mid, 6.0f, 5.5f, 2.5f); layer(c->top, 8.0f, 7.95f, 1.0f);
}

void default_fog(LuminaryFog* f) {  // fog.c:6-16
  f->active = false; f->density = 1.0f; f->droplet_diameter = 10.0f; f->height = 500.0f; f->dist = 500.0f;
}

void default_particles(LuminaryParticles* p) {  // particles.c:6-24
  std::memset(p, 0, sizeof(*p));
  p->scale = 10.0f; p->albedo.r = p->albedo.g = p->albedo.b = 1.0f; p->direction_altitude = 1.234f; p->phase_diameter = 50.0f;
  p->count = 8192; p->size = 1.0f; p->size_variation = 0.1f;
}

HostScene::HostScene() {
  default_settings(&settings); default_camera(&camera); default_ocean(&ocean); default_sky(&sky); default_cloud(&cloud);
  default_fog(&fog); default_particles(&particles);
}

// ---------------------------------------------------------------------------------------------------------------------
// Encoders
// ---------------------------------------------------------------------------------------------------------------------

uint32_t pack_normal(const float n[3]) {  // device_packing.c:6-35 (double precision octahedral encoding)
  double x = n[0], y = n[1], z = n[2];
  const double rn = 1.0 / (std::fabs(x) + std::fabs(y) + std::fabs(z));
  x *= rn; y *= rn; z *= rn;
  const double t = std::fmax(std::fmin(-z, 1.0), 0.0);
  x += (x >= 0.0) ? t : -t;
  y += (y >= 0.0) ? t : -t;
  x = std::fmax(std::fmin(x, 1.0), -1.0);
  y = std::fmax(std::fmin(y, 1.0), -1.0);
  x = (x + 1.0) * 0.5; y = (y + 1.0) * 0.5;
  const uint32_t xu = (uint32_t) (x * 0xFFFF + 0.5), yu = (uint32_t) (y * 0xFFFF + 0.5);
  return (yu << 16) | xu;
}

static uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float bits_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

uint32_t pack_uv(float u, float v) { return (float_bits(u) & 0xFFFF0000u) | (float_bits(v) >> 16); }  // device_packing.c:37-44

static uint16_t unorm16(float f) { return (uint16_t) (f * 0xFFFFu + 0.5f); }                 // device_structs.c:252-254
static uint16_t float_to_u16(float f) { return (uint16_t) ((float_bits(f) >> 15) & 0xFFFF); }  // device_structs.c:256-261

void encode_material(const LuminaryMaterial& m, uint16_t w[16]) {  // device_structs.c:263-311, word order of device_structs.h:202-223
  uint8_t flags = 0;
  flags |= m.emission_active ? 0x02 : 0;
  flags |= m.thin_walled ? 0x04 : 0;
  flags |= m.metallic ? 0x08 : 0;
  flags |= m.colored_transparency ? 0x10 : 0;
  flags |= m.roughness_as_smoothness ? 0x20 : 0;
  flags |= m.normal_map_is_compressed ? 0x40 : 0;
  flags |= m.bidirectional_emission ? 0x80 : 0;
  if (m.base_substrate == LUMINARY_MATERIAL_BASE_SUBSTRATE_TRANSLUCENT) flags |= 0x01;
  const uint8_t clamp8 = (uint8_t) (unorm16(m.roughness_clamp) >> 8);
  w[0] = (uint16_t) (flags | (clamp8 << 8));
  w[1] = m.metallic_tex;
  w[2] = unorm16(m.roughness);
  w[3] = unorm16(0.5f * (m.refraction_index - 1.0f));
  float er = m.emission.r, eg = m.emission.g, eb = m.emission.b;
  const float norm = 1.0f / std::fmin(std::fmax(std::fmax(er, eg), eb) + 1.0f, (float) 0xFFFFu);
  er *= norm; eg *= norm; eb *= norm;
  w[4] = unorm16(m.albedo.r); w[5] = unorm16(m.albedo.g); w[6] = unorm16(m.albedo.b); w[7] = unorm16(m.albedo.a);
  w[8] = unorm16(er); w[9] = unorm16(eg); w[10] = unorm16(eb);
  w[11] = float_to_u16(m.emission_scale / norm);
  w[12] = m.albedo_tex; w[13] = m.luminance_tex; w[14] = m.roughness_tex; w[15] = m.normal_tex;
}

void euler_to_quaternion(const LuminaryVec3& r, float q[4]) {  // host_math.c:6-21
  const float cr = std::cos(r.x * 0.5f), sr = std::sin(r.x * 0.5f), cp = std::cos(r.y * 0.5f), sp = std::sin(r.y * 0.5f);
  const float cy = std::cos(r.z * 0.5f), sy = std::sin(r.z * 0.5f);
  q[3] = cr * cp * cy + sr * sp * sy;
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
}

void encode_transform(const HostInstance& inst, float out[8]) {  // device_structs.c:388-412 (inverse quaternion, 16 bit per component)
  float q[4];
  euler_to_quaternion(inst.rotation, q);
  const uint16_t x = (uint16_t) (((1.0f - q[0]) * 0x7FFF) + 0.5f), y = (uint16_t) (((1.0f - q[1]) * 0x7FFF) + 0.5f);
  const uint16_t z = (uint16_t) (((1.0f - q[2]) * 0x7FFF) + 0.5f), w = (uint16_t) (((1.0f + q[3]) * 0x7FFF) + 0.5f);
  out[0] = inst.translation.x; out[1] = inst.translation.y; out[2] = inst.translation.z;
  out[3] = inst.scale.x; out[4] = inst.scale.y; out[5] = inst.scale.z;
  out[6] = bits_float((uint32_t) x | ((uint32_t) y << 16));
  out[7] = bits_float((uint32_t) z | ((uint32_t) w << 16));
}

// ---------------------------------------------------------------------------------------------------------------------
// Light tree (device_light.c). Binned-SAH binary tree over emissive triangles weighted by power, per-node power-weighted
// mean and spatial variance, collapse into a <=128-child root plus 8-wide quantised nodes.
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct F3 { float x, y, z; };
inline F3 f3(float x, float y, float z) { return F3{x, y, z}; }
inline F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
inline F3 fmin3(F3 a, F3 b) { return f3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)); }
inline F3 fmax3(F3 a, F3 b) { return f3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)); }
inline F3 cross3(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float dot3(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float axis_of(F3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

constexpr float kMaxValue = 1e10f;
constexpr int kLightBins  = 32;
constexpr uint32_t kNull  = 0xFFFFFFFFu;

struct Fragment {  // device_light.h LightTreeFragment
  F3 low, high, middle, v0, v1, v2;
  // the fourth lane of v0, v1, v2 and of `middle`. The reference's vectors have four lanes and vec128_rotate_quaternion (host_intrinsics.h:221-233) scales q.w along
  // with q.xyz, so a vertex of a ROTATED instance leaves it with w = 2 q.w dot(q.xyz, a) instead of 0; nothing clears it: the four-lane dot products of the node
  // variance (device_light.c:538-547) add (w_k - w_mean)^2, and the light BVH's vertex buffer carries it. Found by the independent encoder oracle/o_scene.c in round 5
  // (rounds 1-4 computed three lanes: the zoo's rotated emitters got 111 nodes instead of the reference's 109).
  float w0 = 0.0f, w1 = 0.0f, w2 = 0.0f, wm = 0.0f;
  float power;
  uint32_t instance_id, tri_id;
};

struct BinaryNode {  // device_light.c:40-58 (fields that are used)
  uint32_t triangle_count = 0, triangles_address = 0, child_address = 0;
  bool internal = false;
  float left_power = 0.0f, right_power = 0.0f;
};
struct TreeNode {  // device_light.c:60-76
  F3 left_mean{0, 0, 0}, right_mean{0, 0, 0};
  float left_variance = 0.0f, right_variance = 0.0f, left_power = 0.0f, right_power = 0.0f;
  uint32_t child_ptr = kNull, light_ptr = 0, light_count = 0;
};
struct ChildNode { F3 mean{0, 0, 0}; float variance = 0.0f, power = 0.0f; bool is_leaf = false; };

// vec128_rotate_quaternion, host_intrinsics.h:202-213
F3 rotate_q(F3 a, const float q[4]) {
  const float dqa = a.x * q[0] + a.y * q[1] + a.z * q[2], dqq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  const F3 qv = f3(q[0], q[1], q[2]);
  const F3 cr = cross3(qv, a);
  F3 r = qv * (2.0f * dqa);
  r = r + a * (q[3] * q[3] - dqq);
  r = r + cr * (2.0f * q[3]);
  return r;
}
// ... and its fourth lane: q.w * (2 dot_qa) + a.w * (...) + cross.w * (...) with a.w = cross.w = +0, then * scale.w (1) + offset.w (0) (device_light.c:2029-2031, :2075-2077)
float rotate_q_w(F3 a, const float q[4]) {
  const float dqa = a.x * q[0] + a.y * q[1] + a.z * q[2], dqq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
  float w = q[3] * (2.0f * dqa);
  w = w + 0.0f * (q[3] * q[3] - dqq);
  w = w + 0.0f * (2.0f * q[3]);
  return w * 1.0f + 0.0f;
}

void fit_bounds(const Fragment* f, uint32_t n, F3* high, F3* low) {
  F3 h = f3(-kMaxValue, -kMaxValue, -kMaxValue), l = f3(kMaxValue, kMaxValue, kMaxValue);
  for (uint32_t i = 0; i < n; i++) { h = fmax3(h, f[i].high); l = fmin3(l, f[i].low); }
  *high = h; *low = l;
}
// vec128_box_area (host_intrinsics.h:208-216): the products x y, x z, y z, w w (w = 0) through vec128_hsum, i.e. (xy + yz) + (xz + 0) - the order matters in the last bit
inline float box_area(F3 d) { return (d.x * d.y + d.y * d.z) + (d.x * d.z + 0.0f); }

struct Bin { F3 high, low; int32_t entry; float power; };

// device_light.c:197-243
double construct_bins(Bin* bins, const Fragment* frags, uint32_t n, int axis, double* offset) {
  F3 high, low;
  fit_bounds(frags, n, &high, &low);
  const double high_axis = axis_of(high, axis), low_axis = axis_of(low, axis);
  const double span = high_axis - low_axis;
  const double interval = span / kLightBins;
  if (interval <= (FLT_EPSILON * 16.0f) * std::fabs(low_axis)) return 0.0;
  *offset = low_axis;
  for (int b = 0; b < kLightBins; b++) {
    bins[b].high = f3(-kMaxValue, -kMaxValue, -kMaxValue); bins[b].low = f3(kMaxValue, kMaxValue, kMaxValue);
    bins[b].entry = 0; bins[b].power = 0.0f;
  }
  const double inv_interval = 1.0 / interval;
  for (uint32_t i = 0; i < n; i++) {
    const double value = axis_of(frags[i].middle, axis);
    int32_t pos = ((int32_t) std::ceil((value - low_axis) * inv_interval)) - 1;
    if (pos < 0) pos = 0;
    if (pos >= kLightBins) pos = kLightBins - 1;
    bins[pos].entry++;
    bins[pos].power += frags[i].power;
    bins[pos].high = fmax3(bins[pos].high, frags[i].high);
    bins[pos].low = fmin3(bins[pos].low, frags[i].low);
  }
  return interval;
}

// device_light.c:245-268
void divide_along_axis(double split, int axis, Fragment* frags, uint32_t n) {
  uint32_t left = 0, right = 0;
  while (left + right < n) {
    const Fragment frag = frags[left];
    if ((double) axis_of(frag.middle, axis) > split) {
      const uint32_t swap_index = n - 1 - right;
      frags[left] = frags[swap_index];
      frags[swap_index] = frag;
      right++;
    }
    else left++;
  }
}

// device_light.c:270-486
void build_binary(std::vector<Fragment>& frags, std::vector<BinaryNode>& nodes) {
  nodes.clear();
  if (frags.empty()) return;
  BinaryNode root;
  root.triangle_count = (uint32_t) frags.size();
  nodes.push_back(root);
  Bin bins[kLightBins];
  uint32_t begin = 0, end = 1;
  while (begin != end) {
    for (uint32_t ptr = begin; ptr < end; ptr++) {
      BinaryNode node = nodes[ptr];
      const uint32_t fptr = node.triangles_address, fcount = node.triangle_count;
      if (fcount == 1) continue;
      Fragment* f = frags.data() + fptr;
      F3 hp, lp;
      fit_bounds(f, fcount, &hp, &lp);
      const F3 diff = hp - lp;
      const float max_axis_interval = std::fmax(std::fmax(diff.x, diff.y), std::fmax(diff.z, 0.0f));
      double optimal_cost = DBL_MAX, optimal_plane = 0.0;
      int axis = 0;
      bool found = false;
      uint32_t optimal_split = 0;
      float opt_left_power = 0.0f, opt_right_power = 0.0f;
      for (int a = 0; a < 3; a++) {
        double low_split = 0.0;
        const double interval = construct_bins(bins, f, fcount, a, &low_split);
        if (interval == 0.0) continue;
        const double interval_cost = max_axis_interval / interval;
        uint32_t left = 0;
        float left_power = 0.0f, right_power = 0.0f;
        for (int k = 0; k < kLightBins; k++) right_power += bins[k].power;
        F3 high_left = f3(-kMaxValue, -kMaxValue, -kMaxValue), low_left = f3(kMaxValue, kMaxValue, kMaxValue);
        for (int k = 1; k < kLightBins; k++) {
          high_left = fmax3(high_left, bins[k - 1].high); low_left = fmin3(low_left, bins[k - 1].low);
          F3 high_right = f3(-kMaxValue, -kMaxValue, -kMaxValue), low_right = f3(kMaxValue, kMaxValue, kMaxValue);
          for (int j = k; j < kLightBins; j++) { high_right = fmax3(high_right, bins[j].high); low_right = fmin3(low_right, bins[j].low); }
          left_power += bins[k - 1].power;
          right_power -= bins[k - 1].power;
          const float left_area = box_area(high_left - low_left), right_area = box_area(high_right - low_right);
          const double total_cost = interval_cost * (left_power * left_area + right_power * right_area);
          left += bins[k - 1].entry;
          if (left == 0 || left == fcount) continue;
          if (total_cost < optimal_cost) {
            optimal_cost = total_cost; optimal_split = left; optimal_plane = low_split + k * interval; found = true; axis = a;
            opt_left_power = left_power; opt_right_power = right_power;
          }
        }
      }
      if (found) divide_along_axis(optimal_plane, axis, f, fcount);
      else {
        optimal_split = fcount / 2;
        opt_left_power = 0.0f; opt_right_power = 0.0f;
        uint32_t i = 0;
        for (; i < optimal_split; i++) opt_left_power += f[i].power;
        for (; i < fcount; i++) opt_right_power += f[i].power;
      }
      node.left_power = opt_left_power; node.right_power = opt_right_power;
      node.child_address = (uint32_t) nodes.size();
      BinaryNode l, r;
      l.triangle_count = optimal_split; l.triangles_address = fptr;
      r.triangle_count = node.triangle_count - optimal_split; r.triangles_address = fptr + optimal_split;
      nodes.push_back(l); nodes.push_back(r);
      node.internal = true;
      nodes[ptr] = node;
    }
    begin = end;
    end = (uint32_t) nodes.size();
  }
}

// device_light.c:488-584: power-weighted mean of triangle centres, spatial variance over the three vertices of every member
void mean_and_variance(const std::vector<Fragment>& frags, const BinaryNode& node, float parent_power, float* power, F3* mean, float* variance) {
  if (*power < parent_power * 1e-5f) {
    float np = 0.0f;
    for (uint32_t i = 0; i < node.triangle_count; i++) np += frags[node.triangles_address + i].power;
    *power = np;
  }
  const float inv_total = 1.0f / *power;
  F3 p = f3(0.0f, 0.0f, 0.0f);
  float pw = 0.0f;  // the mean's fourth lane (Fragment::wm)
  for (uint32_t i = 0; i < node.triangle_count; i++) {
    const Fragment& fr = frags[node.triangles_address + i];
    const float w = fr.power * inv_total;
    p = f3(std::fma(fr.middle.x, w, p.x), std::fma(fr.middle.y, w, p.y), std::fma(fr.middle.z, w, p.z));
    pw = std::fma(fr.wm, w, pw);
  }
  // vec128_dot = vec128_hsum of the four products: (x x + z z) + (y y + w w) (host_intrinsics.h:107-117, :201-203)
  auto dot4 = [](F3 d, float dw) { return (d.x * d.x + d.z * d.z) + (d.y * d.y + dw * dw); };
  float var = 0.0f;
  for (uint32_t i = 0; i < node.triangle_count; i++) {
    const Fragment& fr = frags[node.triangles_address + i];
    const float w = (1.0f / 3.0f) * fr.power * inv_total;
    const F3 d0 = fr.v0 - p, d1 = fr.v1 - p, d2 = fr.v2 - p;
    var += w * dot4(d0, fr.w0 - pw);
    var += w * dot4(d1, fr.w1 - pw);
    var += w * dot4(d2, fr.w2 - pw);
  }
  *mean = p; *variance = var;
}

struct Collapse {
  const std::vector<TreeNode>* bn;
  std::vector<uint32_t> jobs;
  std::vector<uint32_t> new_fragments;
  uint32_t triangles_ptr = 0;

  // device_light.c:664-848
  void collapse_node(const TreeNode& base, ChildNode* children, uint32_t* cbi, uint32_t max_children, uint32_t* light_ptr, uint32_t* child_count_out,
                     uint32_t* leaf_count_out) {
    const std::vector<TreeNode>& nodes = *bn;
    uint32_t child_count = 0;
    bool work = false;
    if (base.light_count > 1) {
      ChildNode l; l.mean = base.left_mean; l.variance = base.left_variance; l.power = base.left_power;
      cbi[child_count] = base.child_ptr; children[child_count++] = l;
      ChildNode r; r.mean = base.right_mean; r.variance = base.right_variance; r.power = base.right_power;
      cbi[child_count] = base.child_ptr + 1; children[child_count++] = r;
      work = child_count < max_children;
    }
    else {  // single light in the scene
      ChildNode c; c.is_leaf = true; c.power = 1.0f;
      cbi[child_count] = 0; children[child_count++] = c;
    }
    while (work) {
      work = false;
      float best_cost = 0.0f;
      uint32_t selected = 0;
      for (uint32_t c = 0; c < max_children; c++) {
        if (cbi[c] == kNull) continue;
        const TreeNode& n = nodes[cbi[c]];
        if (n.light_count == 1) continue;
        const float cost = (n.left_power + n.right_power) * (n.left_variance + n.right_variance);
        if (cost > best_cost) { best_cost = cost; selected = c; work = true; }
      }
      if (!work) break;
      const TreeNode n = nodes[cbi[selected]];
      ChildNode l; l.mean = n.left_mean; l.variance = n.left_variance; l.power = n.left_power;
      cbi[selected] = n.child_ptr; children[selected] = l;
      ChildNode r; r.mean = n.right_mean; r.variance = n.right_variance; r.power = n.right_power;
      uint32_t slot = 0;
      for (; slot < max_children; slot++) if (cbi[slot] == kNull) break;
      cbi[slot] = n.child_ptr + 1; children[slot] = r;
      child_count++;
      if (child_count == max_children) break;
    }
    if (child_count < max_children) {
      for (uint32_t c = 0; c < child_count; c++) {
        if (cbi[c] == kNull) {
          uint32_t s = child_count;
          for (; s < max_children; s++) if (cbi[s] != kNull) break;
          std::swap(children[c], children[s]); std::swap(cbi[c], cbi[s]);
        }
      }
    }
    uint32_t leaves = 0;
    for (uint32_t c = 0; c < child_count; c++) {
      const TreeNode& n = nodes[cbi[c]];
      if (n.light_count == 1) {
        if (*light_ptr == kNull) *light_ptr = triangles_ptr;
        new_fragments[triangles_ptr++] = n.light_ptr;
        children[c].is_leaf = true;
        cbi[c] = kNull;
        leaves++;
      }
    }
    for (uint32_t c = 0; c < leaves; c++) {
      if (!children[c].is_leaf) {
        uint32_t s = c + 1;
        for (; s < child_count; s++) if (children[s].is_leaf) break;
        std::swap(children[c], children[s]); std::swap(cbi[c], cbi[s]);
      }
    }
    *child_count_out = child_count; *leaf_count_out = leaves;
  }
};

uint16_t pack_float_mode(float v, int mode) {  // device_packing.c:46-70; mode 0 floor, 1 ceil
  uint32_t b = float_bits(v);
  if (mode == 1) { if (v >= 0.0f) b += (1u << 16) - 1; }
  else { if (v < 0.0f) b += (1u << 16) - 1; }
  return (uint16_t) (b >> 16);
}
float unpack_float16(uint16_t v) { return bits_float(((uint32_t) v) << 16); }

// ---- emission textures: the per-triangle intensity the light tree weighs a textured emitter with (device_light.c:1902-2018 runs
// light_compute_intensity, cuda/light.cuh:191-270, on the GPU; here it runs where the tree is built). A texture fetch must return the
// bits the kernels' fetch returns, so log2/exp2/pow are the sequences of csrc/device/dev_math.h and the filter is the one of
// texture_load in csrc/device/dev_bsdf.h (this file is compiled with -ffp-contract=off like the kernels). ----
float host_log2_det(float x) {
  const uint32_t bits = float_bits(x);
  int e = (int) ((bits >> 23) & 0xFFu) - 127;
  float m = bits_float((bits & 0x007FFFFFu) | 0x3F800000u);
  if (m > 1.41421356f) { m = m * 0.5f; e = e + 1; }
  const float q = (m - 1.0f) / (m + 1.0f);
  const float z = q * q;
  float p = 0.0909090909f;
  p = p * z + 0.111111111f;
  p = p * z + 0.142857143f;
  p = p * z + 0.2f;
  p = p * z + 0.333333333f;
  p = p * z;
  const float ln_m = 2.0f * q + (2.0f * q) * p;
  return (float) e + ln_m * 1.44269504f;
}
float host_exp2_det(float x) {
  x = std::fmin(std::fmax(x, -126.0f), 127.0f);
  const float n = std::rint(x);
  const float f = x - n;
  float p = 1.52527338e-5f;
  p = p * f + 1.54035304e-4f;
  p = p * f + 1.33335581e-3f;
  p = p * f + 9.61812911e-3f;
  p = p * f + 5.55041087e-2f;
  p = p * f + 2.40226507e-1f;
  p = p * f + 6.93147181e-1f;
  p = p * f + 1.0f;
  return std::ldexp(p, (int) n);
}
float host_pow_det(float x, float y) { return (x > 0.0f) ? host_exp2_det(y * host_log2_det(x)) : 0.0f; }

// rgb of a fetch with the default arguments (flip_v, gamma; cuda/texture_utils.cuh:12-45)
void host_texture_rgb(const HostTexture& t, float u_in, float v_in, float rgb[3]) {
  const int w = (int) t.width, h = (int) t.height;
  const float u = u_in, v = 1.0f - v_in;
  const float xb = (u - std::floor(u)) * (float) w - 0.5f, yb = (v - std::floor(v)) * (float) h - 0.5f;
  const float xf = std::floor(xb), yf = std::floor(yb);
  const float ax = xb - xf, ay = yb - yf;
  int x0 = (int) xf, y0 = (int) yf, x1 = x0 + 1, y1 = y0 + 1;
  if (x0 < 0) x0 += w;
  if (y0 < 0) y0 += h;
  if (x1 >= w) x1 -= w;
  if (y1 >= h) y1 -= h;
  const uint32_t t00 = t.texels[x0 + (size_t) y0 * w], t10 = t.texels[x1 + (size_t) y0 * w], t01 = t.texels[x0 + (size_t) y1 * w], t11 = t.texels[x1 + (size_t) y1 * w];
  for (int c = 0; c < 3; c++) {
    const float c00 = ((t00 >> (8 * c)) & 0xFFu) * (1.0f / 255.0f), c10 = ((t10 >> (8 * c)) & 0xFFu) * (1.0f / 255.0f);
    const float c01 = ((t01 >> (8 * c)) & 0xFFu) * (1.0f / 255.0f), c11 = ((t11 >> (8 * c)) & 0xFFu) * (1.0f / 255.0f);
    const float top = c00 + ax * (c10 - c00), bot = c01 + ax * (c11 - c01);
    float r = top + ay * (bot - top);
    if (t.gamma != 1.0f) r = host_pow_det(r, t.gamma);
    rgb[c] = r;
  }
}

// light_microtriangle_id_to_bary, cuda/light_microtriangle.cuh:8-61: 64 micro-triangles in 8 rows of 15, 13, ... 1
void microtriangle_bary(uint32_t id, float b0[2], float b1[2], float b2[2]) {
  uint32_t row;
  // the reference's chain of comparisons, kept literally (its row bounds are inclusive: ids 15, 28, ... belong to the earlier row)
  uint32_t col;
  if (id <= 15) { row = 0; col = id >> 1; }
  else if (id <= 15 + 13) { row = 1; col = (id - 15) >> 1; }
  else if (id <= 15 + 13 + 11) { row = 2; col = (id - 15 - 13) >> 1; }
  else if (id <= 15 + 13 + 11 + 9) { row = 3; col = (id - 15 - 13 - 11) >> 1; }
  else if (id <= 15 + 13 + 11 + 9 + 7) { row = 4; col = (id - 15 - 13 - 11 - 9) >> 1; }
  else if (id <= 15 + 13 + 11 + 9 + 7 + 5) { row = 5; col = (id - 15 - 13 - 11 - 9 - 7) >> 1; }
  else if (id <= 15 + 13 + 11 + 9 + 7 + 5 + 3) { row = 6; col = (id - 15 - 13 - 11 - 9 - 7 - 5) >> 1; }
  else { row = 7; col = 0; }
  const bool is_top = (id & 1u) == (row & 1u);
  b0[0] = (float) row; b0[1] = (float) (col + 1);
  b1[0] = (float) (row + 1); b1[1] = (float) col;
  b2[0] = is_top ? (float) row : (float) (row + 1); b2[1] = is_top ? (float) col : (float) (col + 1);
  for (int k = 0; k < 2; k++) { b0[k] *= 1.0f / 8.0f; b1[k] *= 1.0f / 8.0f; b2[k] *= 1.0f / 8.0f; }
}

// lights_get_max_emission, cuda/light.cuh:191-237: the brightest texel found on a grid over the micro-triangle, one step per texel
float microtriangle_max_emission(const HostTexture& tex, const float vertex[2], const float edge1[2], const float edge2[2], uint32_t id) {
  float b0[2], b1[2], b2[2];
  microtriangle_bary(id, b0, b1, b2);
  const float uv0[2] = {vertex[0] + b0[0] * edge1[0] + b0[1] * edge2[0], vertex[1] + b0[0] * edge1[1] + b0[1] * edge2[1]};
  const float uv1[2] = {vertex[0] + b1[0] * edge1[0] + b1[1] * edge2[0], vertex[1] + b1[0] * edge1[1] + b1[1] * edge2[1]};
  const float uv2[2] = {vertex[0] + b2[0] * edge1[0] + b2[1] * edge2[0], vertex[1] + b2[0] * edge1[1] + b2[1] * edge2[1]};
  const float me1[2] = {uv1[0] - uv0[0], uv1[1] - uv0[1]}, me2[2] = {uv2[0] - uv0[0], uv2[1] - uv0[1]};
  const float steps_u = std::fmax(std::fabs(me1[0]), std::fabs(me2[0])) * (float) tex.width;
  const float steps_v = std::fmax(std::fabs(me1[1]), std::fabs(me2[1])) * (float) tex.height;
  const float steps = std::ceil(std::fmax(steps_u, steps_v));
  const float step_size = 1.0f / steps;
  float best[3] = {0.0f, 0.0f, 0.0f};
  for (float a = 0.0f; a < 1.0f; a += step_size) {
    for (float b = 0.0f; a + b < 1.0f; b += step_size) {
      float rgb[3];
      host_texture_rgb(tex, uv0[0] + a * me1[0] + b * me2[0], uv0[1] + a * me1[1] + b * me2[1], rgb);
      for (int c = 0; c < 3; c++) best[c] = std::fmax(best[c], rgb[c]);
    }
  }
  return std::fmax(best[0], std::fmax(best[1], best[2]));  // color_importance, cuda/math.cuh:1066-1068
}

// light_compute_intensity, cuda/light.cuh:239-270: the maximum over the 64 micro-triangles, from the bf16 texture coordinates the device holds
float triangle_emission_intensity(const HostTexture& tex, const float* uvs /* 6 */) {
  float c[3][2];
  for (int k = 0; k < 3; k++) {
    const uint32_t packed = pack_uv(uvs[2 * k], uvs[2 * k + 1]);
    c[k][0] = bits_float(packed & 0xFFFF0000u); c[k][1] = bits_float(packed << 16);
  }
  const float e1[2] = {c[1][0] - c[0][0], c[1][1] - c[0][1]}, e2[2] = {c[2][0] - c[0][0], c[2][1] - c[0][1]};
  float best = 0.0f;
  for (uint32_t id = 0; id < 64; id++) best = std::fmax(best, microtriangle_max_emission(tex, c[0], e1, e2, id));
  return best;
}

struct Quantiser { F3 min_mean; int8_t ex, ey, ez, es; float cx, cy, cz, cv; uint16_t bx, by, bz; };

int8_t exponent_for(float range) { return (int8_t) std::ceil(std::log2(range * 1.0f / 255.0f)); }

Quantiser make_quantiser(const ChildNode* children, uint32_t count, float* max_power_out) {
  F3 mn = f3(kMaxValue, kMaxValue, kMaxValue), mx = f3(-kMaxValue, -kMaxValue, -kMaxValue);
  float max_variance = 0.0f, max_power = 0.0f;
  for (uint32_t c = 0; c < count; c++) {
    mn = fmin3(mn, children[c].mean); mx = fmax3(mx, children[c].mean);
    max_variance = std::fmax(max_variance, children[c].variance);
    max_power = std::fmax(max_power, children[c].power);
  }
  const float max_std_dev = std::sqrt(max_variance);
  Quantiser q;
  q.bx = pack_float_mode(mn.x, 0); q.by = pack_float_mode(mn.y, 0); q.bz = pack_float_mode(mn.z, 0);
  mn = f3(unpack_float16(q.bx), unpack_float16(q.by), unpack_float16(q.bz));
  q.min_mean = mn;
  q.ex = (mx.x != mn.x) ? exponent_for(mx.x - mn.x) : 0;
  q.ey = (mx.y != mn.y) ? exponent_for(mx.y - mn.y) : 0;
  q.ez = (mx.z != mn.z) ? exponent_for(mx.z - mn.z) : 0;
  // a lone light has zero variance; log2(0) has no int8 image, the reference leaves that to the compiler (device_light.c:939)
  q.es = (max_std_dev > 0.0f) ? exponent_for(max_std_dev) : 0;
  q.cx = 1.0f / std::exp2((float) q.ex); q.cy = 1.0f / std::exp2((float) q.ey); q.cz = 1.0f / std::exp2((float) q.ez);
  q.cv = 1.0f / std::exp2((float) q.es);
  *max_power_out = max_power;
  return q;
}

struct QuantChild { uint8_t mx, my, mz, sd; uint32_t power; };
QuantChild quantise_child(const Quantiser& q, const ChildNode& c, float max_power, uint32_t power_scale) {
  QuantChild o;
  o.mx = (uint8_t) (uint64_t) std::floor((c.mean.x - q.min_mean.x) * q.cx + 0.5f);
  o.my = (uint8_t) (uint64_t) std::floor((c.mean.y - q.min_mean.y) * q.cy + 0.5f);
  o.mz = (uint8_t) (uint64_t) std::floor((c.mean.z - q.min_mean.z) * q.cz + 0.5f);
  uint64_t sd = (uint64_t) (std::sqrt(c.variance) * q.cv + 0.5f);
  uint64_t pw = (uint64_t) std::floor(power_scale * c.power / max_power + 0.5f);
  sd = std::max<uint64_t>(sd, 1); pw = std::max<uint64_t>(pw, 1);
  o.sd = (uint8_t) sd; o.power = (uint32_t) pw;
  return o;
}

}  // namespace

void build_light_tree(const HostScene& scene, LightTreeOutput* out) {
  out->root.clear(); out->nodes.clear(); out->tri_handles.clear(); out->bvh_tris.clear();
  // ---- fragments (device_light.c:2020-2113): world-space emissive triangles, grouped per instance by material slot ----
  std::vector<Fragment> frags;
  std::map<std::pair<uint64_t, uint32_t>, float> intensity_cache;  // (mesh, triangle), texture -> integrated maximum; instances share it
  auto textured_intensity = [&](uint32_t mesh_id, uint32_t tri, uint32_t tex) {
    const auto key = std::make_pair(((uint64_t) mesh_id << 32) | tri, tex);
    auto it = intensity_cache.find(key);
    if (it != intensity_cache.end()) return it->second;
    const float v = triangle_emission_intensity(scene.textures[tex], scene.meshes[mesh_id].uvs.data() + 6 * (size_t) tri);
    intensity_cache.emplace(key, v);
    return v;
  };
  for (uint32_t inst_id = 0; inst_id < scene.instances.size(); inst_id++) {
    const HostInstance& inst = scene.instances[inst_id];
    if (!inst.active || inst.mesh_id >= scene.meshes.size()) continue;
    const HostMesh& mesh = scene.meshes[inst.mesh_id];
    float qf[4];
    euler_to_quaternion(inst.rotation, qf);
    const float q[4] = {-qf[0], -qf[1], -qf[2], qf[3]};
    const F3 offset = f3(inst.translation.x, inst.translation.y, inst.translation.z), scale = f3(inst.scale.x, inst.scale.y, inst.scale.z);
    // material slots in order of first appearance (device_light.c:1648-1672)
    std::vector<uint16_t> slots;
    for (uint32_t t = 0; t < mesh.triangle_count(); t++)
      if (std::find(slots.begin(), slots.end(), mesh.material_ids[t]) == slots.end()) slots.push_back(mesh.material_ids[t]);
    for (uint16_t mat_id : slots) {
      if (mat_id >= scene.materials.size()) continue;
      const LuminaryMaterial& mat = scene.materials[mat_id];
      float intensity = 0.0f;  // device_light.c:1828-1846
      if (mat.emission_active) intensity = (mat.luminance_tex != 0xFFFF) ? mat.emission_scale : std::fmax(mat.emission.r, std::fmax(mat.emission.g, mat.emission.b));
      if (!(intensity > 0.0f)) continue;
      const bool textured = mat.emission_active && mat.luminance_tex != 0xFFFF;
      for (uint32_t t = 0; t < mesh.triangle_count(); t++) {
        if (mesh.material_ids[t] != mat_id) continue;
        // device_light.c:1686, :2014: 1 for constant emission, the integrated texture maximum otherwise (0 for a missing texture: light.cuh:195-196)
        float average_intensity = 1.0f;
        if (textured) average_intensity = mat.luminance_tex < scene.textures.size() ? textured_intensity(inst.mesh_id, t, mat.luminance_tex) : 0.0f;
        if (average_intensity == 0.0f) continue;
        const float* p = mesh.positions.data() + 9 * (size_t) t;
        const F3 a = rotate_q(f3(p[0], p[1], p[2]), q) * scale + offset, b = rotate_q(f3(p[3], p[4], p[5]), q) * scale + offset,
                 c = rotate_q(f3(p[6], p[7], p[8]), q) * scale + offset;
        const F3 cr = cross3(b - a, c - a);
        const float area = 0.5f * std::sqrt((cr.x * cr.x + cr.z * cr.z) + (cr.y * cr.y + 0.0f));  // vec128_norm2: sqrt of vec128_hsum = (x + z) + (y + w)
        if (area == 0.0f) continue;
        Fragment f;
        f.low = fmin3(a, fmin3(b, c)); f.high = fmax3(a, fmax3(b, c));
        f.middle = (a + (b + c)) * (1.0f / 3.0f);
        f.v0 = a; f.v1 = b; f.v2 = c;
        f.w0 = rotate_q_w(f3(p[0], p[1], p[2]), q); f.w1 = rotate_q_w(f3(p[3], p[4], p[5]), q); f.w2 = rotate_q_w(f3(p[6], p[7], p[8]), q);
        f.wm = (f.w0 + (f.w1 + f.w2)) * (1.0f / 3.0f);
        f.power = intensity * area * average_intensity;
        f.instance_id = inst_id; f.tri_id = t;
        frags.push_back(f);
      }
    }
  }
  if (frags.empty()) return;

  std::vector<BinaryNode> binary;
  build_binary(frags, binary);

  // ---- traversal structure (device_light.c:586-649) ----
  std::vector<TreeNode> tree(binary.size());
  for (size_t i = 0; i < binary.size(); i++) {
    const BinaryNode& b = binary[i];
    TreeNode n;
    n.left_power = b.left_power; n.right_power = b.right_power; n.light_count = b.triangle_count; n.light_ptr = b.triangles_address;
    if (b.internal) {
      const float parent_power = b.left_power + b.right_power;
      n.child_ptr = b.child_address;
      mean_and_variance(frags, binary[b.child_address], parent_power, &n.left_power, &n.left_mean, &n.left_variance);
      mean_and_variance(frags, binary[b.child_address + 1], parent_power, &n.right_power, &n.right_mean, &n.right_variance);
    }
    tree[i] = n;
  }

  // ---- collapse (device_light.c:850-1224) ----
  Collapse cw;
  cw.bn = &tree;
  cw.new_fragments.assign(frags.size(), kNull);
  {
    ChildNode children[128];
    uint32_t cbi[128];
    for (auto& c : cbi) c = kNull;
    uint32_t light_ptr = kNull, child_count = 0, num_lights = 0;
    cw.collapse_node(tree[0], children, cbi, 128, &light_ptr, &child_count, &num_lights);
    float max_power;
    const Quantiser q = make_quantiser(children, child_count, &max_power);
    const uint32_t num_sections = (child_count + 7) / 8;
    out->root.assign(16 + 48 * (size_t) num_sections, 0);
    uint8_t* h = out->root.data();
    const uint16_t h16[5] = {q.bx, q.by, q.bz, (uint16_t) num_lights, pack_float_mode(max_power, 1)};
    std::memcpy(h, h16, 10);
    h[10] = (uint8_t) num_sections; h[11] = 0;
    h[12] = (uint8_t) q.ex; h[13] = (uint8_t) q.ey; h[14] = (uint8_t) q.ez; h[15] = (uint8_t) q.es;
    for (uint32_t c = 0; c < child_count; c++) {
      const QuantChild qc = quantise_child(q, children[c], max_power, 0xFFFF);
      uint8_t* s = h + 16 + 48 * (size_t) (c / 8);
      const uint32_t k = c % 8;
      s[k] = qc.mx; s[8 + k] = qc.my; s[16 + k] = qc.mz; s[24 + k] = qc.sd;
      const uint16_t pw = (uint16_t) qc.power;
      std::memcpy(s + 32 + 2 * k, &pw, 2);
    }
    for (uint32_t c = 0; c < child_count; c++) if (cbi[c] != kNull) cw.jobs.push_back(cbi[c]);
  }
  for (size_t job = 0; job < cw.jobs.size(); job++) {
    ChildNode children[8];
    uint32_t cbi[8];
    for (auto& c : cbi) c = kNull;
    uint32_t light_ptr = kNull, child_count = 0, num_lights = 0;
    cw.collapse_node(tree[cw.jobs[job]], children, cbi, 8, &light_ptr, &child_count, &num_lights);
    float max_power;
    const Quantiser q = make_quantiser(children, child_count, &max_power);
    uint8_t n[64];
    std::memset(n, 0, 64);
    const uint16_t b16[3] = {q.bx, q.by, q.bz};
    std::memcpy(n, b16, 6);
    n[8] = (uint8_t) q.ex; n[9] = (uint8_t) q.ey; n[10] = (uint8_t) q.ez; n[11] = (uint8_t) q.es;
    n[12] = (uint8_t) num_lights;
    const uint32_t child_ptr = (uint32_t) cw.jobs.size();
    std::memcpy(n + 16, &child_ptr, 4);
    std::memcpy(n + 20, &light_ptr, 4);
    for (uint32_t c = 0; c < child_count; c++) {
      const QuantChild qc = quantise_child(q, children[c], max_power, 0xFF);
      n[24 + c] = qc.mx; n[32 + c] = qc.my; n[40 + c] = qc.mz; n[48 + c] = qc.sd; n[56 + c] = (uint8_t) qc.power;
    }
    out->nodes.insert(out->nodes.end(), n, n + 64);
    for (uint32_t c = 0; c < child_count; c++) if (cbi[c] != kNull) cw.jobs.push_back(cbi[c]);
  }

  // ---- finalize (device_light.c:1226-1288): apply the permutation found by the collapse ----
  const size_t nl = frags.size();
  out->tri_handles.resize(2 * nl);
  out->bvh_tris.assign(12 * nl, 0.0f);
  for (size_t id = 0; id < nl; id++) {
    const Fragment& f = frags[cw.new_fragments[id]];
    out->tri_handles[2 * id] = f.instance_id; out->tri_handles[2 * id + 1] = f.tri_id;
    float* t = out->bvh_tris.data() + 12 * id;
    t[0] = f.v0.x; t[1] = f.v0.y; t[2] = f.v0.z; t[4] = f.v1.x; t[5] = f.v1.y; t[6] = f.v1.z; t[8] = f.v2.x; t[9] = f.v2.y; t[10] = f.v2.z;
    t[3] = f.w0; t[7] = f.w1; t[11] = f.w2;  // LightTreeBVHTriangle is three Vec128: the lanes no kernel reads (vertex stride 16, optix_bvh.c:382-478) hold what the arithmetic left
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Scene -> device format
// ---------------------------------------------------------------------------------------------------------------------

// ---- particles: particle_generate (cuda/particle.cuh:165-211) evaluated by the host layer (the reference runs it on the GPU at scene update,
// device_particle.c:99-131). white_noise_offset = the 16-bit Squares generator (random.cuh:196-211, :297-307). ----
static float particle_white_noise(uint32_t offset) {
  const uint32_t key = 0xfcbd6e15u, counter = offset;
  uint32_t x = counter * key, y = counter * key, z = y + key;
  x = x * x + y; x = (x >> 16) | (x << 16);
  x = x * x + z; x = (x >> 16) | (x << 16);
  const uint32_t v = ((x * x + y) >> 16) & 0xFFFFu;
  return bits_float(0x3F800000u | (v << 7)) - 1.0f;
}
void generate_particles(const LuminaryParticles& p, std::vector<float>* vertices, std::vector<float>* normals) {
  const uint32_t count = p.count;
  vertices->assign((size_t) count * 24, 0.0f);
  normals->assign((size_t) count * 4, 0.0f);
  const float size = p.size * 0.001f;
  for (uint32_t id = 0; id < count; id++) {
    const uint32_t base = p.seed + id * 6;
    const float px = particle_white_noise(base + 0), py = particle_white_noise(base + 1), pz = particle_white_noise(base + 2);
    const float r1 = 2.0f * particle_white_noise(base + 3) - 1.0f, r2 = particle_white_noise(base + 4);
    float n[3];  // sample_ray_sphere, math.cuh:326-344
    if (std::fabs(r1) > 1.0f - FLT_EPSILON) { n[0] = 0.0f; n[1] = 0.0f; n[2] = std::copysign(1.0f, r1); }
    else {
      const float a = std::sqrt(1.0f - r1 * r1), b = 2.0f * 3.14159265358979323846f * r2;
      n[0] = a * std::cos(b); n[1] = a * std::sin(b); n[2] = r1;
    }
    // create_basis, math.cuh:301-321
    const float sign = std::copysign(1.0f, n[2]);
    const float a = -1.0f / (sign + n[2]);
    const float b = n[0] * n[1] * a;
    const float u1[3] = {1.0f + sign * n[0] * n[0] * a, sign * b, -sign * n[0]};
    const float u2[3] = {b, sign + n[1] * n[1] * a, -n[1]};
    const float random_size = 2.0f * particle_white_noise(base + 5) - 1.0f;
    const float s = size * (1.0f + random_size * p.size_variation);
    auto corner = [&](float cx, float cy, float* o) {  // p + transform_vec3(basis, (cx, cy, 0)), math.cuh:445-453
      o[0] = px + ((u1[0] * cx + u2[0] * cy) + n[0] * 0.0f);
      o[1] = py + ((u1[1] * cx + u2[1] * cy) + n[1] * 0.0f);
      o[2] = pz + ((u1[2] * cx + u2[2] * cy) + n[2] * 0.0f);
    };
    float a00[3], a01[3], a10[3], a11[3];
    corner(s, s, a00); corner(s, -s, a01); corner(-s, s, a10); corner(-s, -s, a11);
    const float* order[6] = {a00, a01, a10, a11, a01, a10};
    float* v = vertices->data() + (size_t) id * 24;
    for (int k = 0; k < 6; k++) { v[4 * k] = order[k][0]; v[4 * k + 1] = order[k][1]; v[4 * k + 2] = order[k][2]; v[4 * k + 3] = 1.0f; }
    const float e1[3] = {a01[0] - a00[0], a01[1] - a00[1], a01[2] - a00[2]}, e2[3] = {a10[0] - a00[0], a10[1] - a00[1], a10[2] - a00[2]};
    float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const float inv = 1.0f / std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    float* q = normals->data() + (size_t) id * 4;
    q[0] = c[0] * inv; q[1] = c[1] * inv; q[2] = c[2] * inv;
  }
}

// Which parts of the device scene an edit touched (scene.h:42-63 keeps such flags per entity; device_manager.c:311-320, :424-450 re-uploads by
// them). LUMC_DIRTY_* of include/lum_core.h, shared with the core's lumc_scene_update.
std::string update_device_scene(const HostScene& scene, const std::vector<uint32_t>& bluenoise, uint32_t dirty, DeviceSceneBuffers* out) {
  if (bluenoise.size() != 65536) return "blue-noise mask must hold 65536 texels";
  const LuminaryRendererSettings& st = scene.settings;
  if (st.supersampling > 3) return "supersampling exceeds its 2-bit field";
  const uint32_t width = st.width << st.supersampling, height = st.height << st.supersampling;  // device_structs.c:20-21
  if (width == 0 || height == 0 || width >= 16384 || height >= 16384) return "internal resolution must be in [1, 16383] (14-bit pixel ids)";
  if (scene.camera.use_physical_camera) return "physical camera is outside the supported scope (thin lens only)";
  if (scene.materials.size() > 0xFFFF) return "too many materials";

  DeviceSceneBuffers& b = *out;
  if (b.bluenoise.size() != 65536) { b.bluenoise = bluenoise; dirty = LUMC_DIRTY_ALL; }  // first build
  if (dirty & LUMC_DIRTY_MESHES) {
  b.mesh_tri_offset.assign(scene.meshes.size() + 1, 0);
  size_t total = 0;
  for (size_t m = 0; m < scene.meshes.size(); m++) { b.mesh_tri_offset[m] = (uint32_t) total; total += scene.meshes[m].triangle_count(); }
  b.mesh_tri_offset[scene.meshes.size()] = (uint32_t) total;
  b.vertices.resize(total * 12);
  b.tri_tex.resize(total * 4);
  for (size_t m = 0; m < scene.meshes.size(); m++) {
    const HostMesh& mesh = scene.meshes[m];
    for (uint32_t t = 0; t < mesh.triangle_count(); t++) {
      const size_t g = (size_t) b.mesh_tri_offset[m] + t;
      for (int k = 0; k < 3; k++) {  // device_structs.c:351-361
        float* v = b.vertices.data() + (g * 3 + k) * 4;
        v[0] = mesh.positions[9 * (size_t) t + 3 * k]; v[1] = mesh.positions[9 * (size_t) t + 3 * k + 1]; v[2] = mesh.positions[9 * (size_t) t + 3 * k + 2];
        v[3] = bits_float(pack_normal(mesh.normals.data() + 9 * (size_t) t + 3 * k));
      }
      uint32_t* tt = b.tri_tex.data() + g * 4;  // device_structs.c:363-374
      tt[0] = pack_uv(mesh.uvs[6 * (size_t) t + 0], mesh.uvs[6 * (size_t) t + 1]);
      tt[1] = pack_uv(mesh.uvs[6 * (size_t) t + 2], mesh.uvs[6 * (size_t) t + 3]);
      tt[2] = pack_uv(mesh.uvs[6 * (size_t) t + 4], mesh.uvs[6 * (size_t) t + 5]);
      tt[3] = mesh.material_ids[t];
    }
  }
  }
  if (dirty & (LUMC_DIRTY_INSTANCES | LUMC_DIRTY_MESHES)) {
  // inactive instances keep their slot (ids are stable) but point at no mesh
  b.instance_mesh_ids.resize(scene.instances.size());
  b.instance_transforms.resize(scene.instances.size() * 8);
  for (size_t i = 0; i < scene.instances.size(); i++) {
    const HostInstance& inst = scene.instances[i];
    b.instance_mesh_ids[i] = (inst.active && inst.mesh_id < scene.meshes.size()) ? inst.mesh_id : 0xFFFFFFFFu;
    encode_transform(inst, b.instance_transforms.data() + 8 * i);
  }
  }
  if (dirty & LUMC_DIRTY_MATERIALS) {
    b.materials.resize(scene.materials.size() * 16);
    for (size_t i = 0; i < scene.materials.size(); i++) encode_material(scene.materials[i], b.materials.data() + 16 * i);
  }
  if (dirty & LUMC_DIRTY_LIGHTS) {  // the tree depends on the emissive materials, the instances and the meshes (device_manager.c:439-450: rebuilt when its build id changes)
    LightTreeOutput lt;
    build_light_tree(scene, &lt);
    b.light_tree_root = lt.root; b.light_tree_nodes = lt.nodes; b.light_tri_handles = lt.tri_handles; b.light_bvh_tris = lt.bvh_tris;
  }

  LumDeviceSceneView& v = b.view;
  std::memset(&v, 0, sizeof(v));
  v.num_meshes = (uint32_t) scene.meshes.size(); v.num_instances = (uint32_t) scene.instances.size();
  v.num_materials = (uint32_t) scene.materials.size(); v.num_lights = (uint32_t) (b.light_tri_handles.size() / 2);
  // the texture pool: the scene's textures, then the embedded moon textures while the procedural sky is on (a change of the sky mode moves them in or out)
  const bool moon_in_pool = (scene.sky.mode & 3u) == LUMINARY_SKY_MODE_DEFAULT;
  if (b.moon_in_pool != moon_in_pool) dirty |= LUMC_DIRTY_TEXTURES;
  b.rebuilt = dirty;
  if (dirty & LUMC_DIRTY_TEXTURES) {
  b.moon_in_pool = moon_in_pool;
  b.moon_albedo_tex = b.moon_normal_tex = 0xFFFFFFFFu;
  b.texture_table.clear(); b.texels.clear();
  for (const HostTexture& t : scene.textures) {
    float g = t.gamma;
    uint32_t gbits;
    std::memcpy(&gbits, &g, 4);
    b.texture_table.push_back((uint32_t) b.texels.size()); b.texture_table.push_back(t.width); b.texture_table.push_back(t.height); b.texture_table.push_back(gbits);
    b.texels.insert(b.texels.end(), t.texels.begin(), t.texels.end());
  }
  b.num_textures = (uint32_t) scene.textures.size();
  if (moon_in_pool) {
    // the embedded moon textures join the pool behind the scene's own (device_embedded_data.c:62-92)
    const HostTexture* moon[2];
    if (moon_textures(moon)) {
      for (int k = 0; k < 2; k++) {
        float g = moon[k]->gamma;
        uint32_t gbits;
        std::memcpy(&gbits, &g, 4);
        b.texture_table.push_back((uint32_t) b.texels.size()); b.texture_table.push_back(moon[k]->width); b.texture_table.push_back(moon[k]->height);
        b.texture_table.push_back(gbits);
        b.texels.insert(b.texels.end(), moon[k]->texels.begin(), moon[k]->texels.end());
      }
      b.moon_albedo_tex = b.num_textures; b.moon_normal_tex = b.num_textures + 1;
      b.num_textures += 2;
    }
  }
  }
  v.num_textures = b.num_textures; v.sky_moon_albedo_tex = b.moon_albedo_tex; v.sky_moon_normal_tex = b.moon_normal_tex;
  v.texture_table = b.texture_table.empty() ? nullptr : b.texture_table.data();
  v.texels = b.texels.empty() ? nullptr : b.texels.data();
  v.mesh_tri_offset = b.mesh_tri_offset.data(); v.vertices = b.vertices.data(); v.tri_tex = b.tri_tex.data();
  v.instance_mesh_ids = b.instance_mesh_ids.data(); v.instance_transforms = b.instance_transforms.data(); v.materials = b.materials.data();
  v.light_tree_root = b.light_tree_root.empty() ? nullptr : b.light_tree_root.data();
  v.light_tree_nodes = b.light_tree_nodes.empty() ? nullptr : b.light_tree_nodes.data();
  v.light_tri_handles = b.light_tri_handles.empty() ? nullptr : b.light_tri_handles.data();
  v.light_bvh_tris = b.light_bvh_tris.empty() ? nullptr : b.light_bvh_tris.data();
  v.num_light_tree_nodes = (uint32_t) (b.light_tree_nodes.size() / 64);
  v.bluenoise_2d = b.bluenoise.data();
  v.width = width; v.height = height; v.max_ray_depth = st.max_ray_depth & 63u; v.shading_mode = st.shading_mode;
  // camera (device_structs.c:40-88)
  v.cam_pos[0] = scene.camera.pos.x; v.cam_pos[1] = scene.camera.pos.y; v.cam_pos[2] = scene.camera.pos.z;
  euler_to_quaternion(scene.camera.rotation, v.cam_rotation);
  v.cam_fov = scene.camera.thin_lens.fov; v.cam_aperture_size = scene.camera.thin_lens.aperture_size;
  v.cam_object_distance = scene.camera.object_distance; v.cam_scale = scene.camera.camera_scale;
  v.cam_rr_threshold = scene.camera.russian_roulette_threshold;
  v.cam_aperture_shape = scene.camera.aperture_shape & 1u; v.cam_aperture_blade_count = scene.camera.aperture_blade_count & 7u;
  v.sky_mode = scene.sky.mode & 3u;
  v.sky_constant_color[0] = scene.sky.constant_color.r; v.sky_constant_color[1] = scene.sky.constant_color.g; v.sky_constant_color[2] = scene.sky.constant_color.b;
  // procedural sky: device_struct_sky_convert, device_structs.c:106-150 (sun position in double, as there)
  const LuminarySky& sky = scene.sky;
  v.sky_steps = sky.steps & 1023u; v.sky_ozone_absorption = sky.ozone_absorption ? 1u : 0u;
  v.sky_geometry_offset[0] = sky.geometry_offset.x; v.sky_geometry_offset[1] = sky.geometry_offset.y; v.sky_geometry_offset[2] = sky.geometry_offset.z;
  v.sky_sun_strength = sky.sun_strength; v.sky_base_density = sky.base_density; v.sky_rayleigh_density = sky.rayleigh_density; v.sky_mie_density = sky.mie_density;
  v.sky_ozone_density = sky.ozone_density; v.sky_rayleigh_falloff = sky.rayleigh_falloff; v.sky_mie_falloff = sky.mie_falloff;
  v.sky_ground_visibility = sky.ground_visibility; v.sky_ozone_layer_thickness = sky.ozone_layer_thickness; v.sky_multiscattering_factor = sky.multiscattering_factor;
  {
    double sx = std::cos((double) sky.azimuth) * std::cos((double) sky.altitude), sy = std::sin((double) sky.altitude), sz = std::sin((double) sky.azimuth) * std::cos((double) sky.altitude);
    const double scale = 1.0 / std::sqrt(sx * sx + sy * sy + sz * sz);
    const double sun_distance = 149597870.0f, earth_radius = 6371.0f;  // sky_defines.h:4-6
    sx *= scale * sun_distance; sy *= scale * sun_distance; sz *= scale * sun_distance;
    sy -= earth_radius;
    sx -= sky.geometry_offset.x; sy -= sky.geometry_offset.y; sz -= sky.geometry_offset.z;
    v.sky_sun_pos[0] = (float) sx; v.sky_sun_pos[1] = (float) sy; v.sky_sun_pos[2] = (float) sz;
  }
  jendersie_eon_parameters(sky.mie_diameter, v.sky_mie_phase);
  {
    double mx = std::cos((double) sky.moon_azimuth) * std::cos((double) sky.moon_altitude), my = std::sin((double) sky.moon_altitude),
           mz = std::sin((double) sky.moon_azimuth) * std::cos((double) sky.moon_altitude);
    const double scale = 1.0 / std::sqrt(mx * mx + my * my + mz * mz);
    const double moon_distance = 384399.0f, earth_radius = 6371.0f;  // sky_defines.h:4, :8
    mx *= scale * moon_distance; my *= scale * moon_distance; mz *= scale * moon_distance;
    my -= earth_radius;
    mx -= sky.geometry_offset.x; my -= sky.geometry_offset.y; mz -= sky.geometry_offset.z;
    v.sky_moon_pos[0] = (float) mx; v.sky_moon_pos[1] = (float) my; v.sky_moon_pos[2] = (float) mz;
  }
  v.sky_moon_tex_offset = sky.moon_tex_offset;
  v.sky_stars_intensity = sky.stars_intensity;
  v.sky_stars_count = 0; v.sky_stars = nullptr; v.sky_stars_offsets = nullptr;
  if ((sky.mode & 3u) == LUMINARY_SKY_MODE_DEFAULT && generate_stars(sky.stars_seed, sky.stars_count, &b.sky_stars, &b.sky_stars_offsets)) {
    v.sky_stars_count = sky.stars_count; v.sky_stars = b.sky_stars.data(); v.sky_stars_offsets = b.sky_stars_offsets.data();
  }
  v.sky_lut_transmittance = nullptr; v.sky_lut_multiscattering = nullptr;  // generated on the GPU at upload
  // HDRI mode: baked on the GPU at upload (sky_hdri_update, device/device_sky.c:249-281: dim and samples at least 1, origin = camera)
  v.sky_hdri = nullptr;
  v.sky_hdri_dim = sky.hdri_dim ? sky.hdri_dim : 1u; v.sky_hdri_samples = sky.hdri_samples ? sky.hdri_samples : 1u;
  std::memcpy(v.sky_hdri_origin, scene.hdri_origin, sizeof(v.sky_hdri_origin));
  v.sky_aerial_perspective = sky.aerial_perspective ? 1u : 0u;
  // fog (device_struct_fog_convert, device_structs.c:219-231); the phase function's parameters depend on the droplet diameter only
  v.fog_active = scene.fog.active ? 1u : 0u;
  v.fog_density = scene.fog.density; v.fog_dist = scene.fog.dist; v.fog_height = scene.fog.height;
  jendersie_eon_parameters(scene.fog.droplet_diameter, v.fog_phase);
  v.bridge_lut = reinterpret_cast<const float*>(lum_embedded_bridge_lut);  // device_embedded_data.c:50-60
  v.bridge_max_num_vertices = st.bridge_max_num_vertices & 15u;             // device_structs.h:11
  // particles (device_particle.c:84-131: nothing exists while they are inactive)
  const LuminaryParticles& pt = scene.particles;
  v.particles_active = (pt.active && pt.count > 0) ? 1u : 0u;
  v.particles_count = v.particles_active ? pt.count : 0u;
  v.particles_scale = pt.scale; v.particles_speed = pt.speed;
  v.particles_albedo[0] = pt.albedo.r; v.particles_albedo[1] = pt.albedo.g; v.particles_albedo[2] = pt.albedo.b;
  v.particles_direction[0] = std::cos(pt.direction_azimuth) * std::cos(pt.direction_altitude);  // angles_to_direction, math.cuh:781-788
  v.particles_direction[1] = std::sin(pt.direction_altitude);
  v.particles_direction[2] = std::sin(pt.direction_azimuth) * std::cos(pt.direction_altitude);
  jendersie_eon_parameters(pt.phase_diameter, v.particles_phase);
  if (dirty & LUMC_DIRTY_PARTICLES) {
    b.particle_vertices.clear(); b.particle_normals.clear();
    if (v.particles_active) {
      if (pt.count > (1u << 22)) return "particles: more than 4 M particles";
      if (!(pt.scale > 0.0f)) return "particles: scale must be positive";
      generate_particles(pt, &b.particle_vertices, &b.particle_normals);
    }
  }
  v.particle_vertices = b.particle_vertices.empty() ? nullptr : b.particle_vertices.data();
  v.particle_normals = b.particle_normals.empty() ? nullptr : b.particle_normals.data();
  // ocean (device_struct_ocean_convert, device_structs.c:87-105); Jerlov coefficients: ocean_utils.cuh:291-385
  const LuminaryOcean& oc = scene.ocean;
  v.ocean_active = oc.active ? 1u : 0u;
  v.ocean_height = oc.height; v.ocean_amplitude = oc.amplitude; v.ocean_frequency = oc.frequency; v.ocean_refractive_index = oc.refractive_index;
  {
    static const float scattering[10][3] = {{0.001f, 0.002f, 0.004f}, {0.002f, 0.004f, 0.007f}, {0.045f, 0.054f, 0.07f}, {0.27f, 0.365f, 0.516f}, {0.737f, 0.998f, 1.413f},
                                            {0.274f, 0.372f, 0.526f}, {0.904f, 1.071f, 1.532f}, {3.589f, 1.382f, 1.857f}, {1.772f, 2.394f, 3.376f}, {2.347f, 3.18f, 4.496f}};
    static const float absorption[10][3] = {{0.309f, 0.053f, 0.009f}, {0.309f, 0.054f, 0.014f}, {0.309f, 0.054f, 0.015f}, {0.31f, 0.054f, 0.016f}, {0.31f, 0.056f, 0.031f},
                                            {0.316f, 0.067f, 0.105f}, {0.508f, 0.052f, 0.161f}, {4.638f, 0.222f, 0.216f}, {0.351f, 0.188f, 0.574f}, {0.398f, 0.349f, 0.995f}};
    static const float molecular_weight[10] = {0.93f, 0.44f, 0.06f, 0.007f, 0.003f, 0.005f, 0.003f, 0.001f, 0.0f, 0.0f};
    const uint32_t type = (uint32_t) oc.water_type;
    for (int k = 0; k < 3; k++) { v.ocean_scattering[k] = type < 10 ? scattering[type][k] : 0.0f; v.ocean_absorption[k] = type < 10 ? absorption[type][k] : 0.0f; }
    v.ocean_molecular_weight = type < 10 ? molecular_weight[type] : 0.0f;
  }
  v.ocean_caustics_active = oc.caustics_active ? 1u : 0u;
  v.ocean_caustics_ris_sample_count = std::max(oc.caustics_ris_sample_count, 1u) - 1u;  // device_structs.c:94
  v.ocean_caustics_domain_scale = oc.caustics_domain_scale;
  v.ocean_multiscattering = oc.multiscattering ? 1u : 0u; v.ocean_triangle_light_contribution = oc.triangle_light_contribution ? 1u : 0u;
  // clouds (device_struct_cloud_convert, device_structs.c:173-217)
  const LuminaryCloud& cl = scene.cloud;
  v.cloud_active = cl.active ? 1u : 0u; v.cloud_atmosphere_scattering = cl.atmosphere_scattering ? 1u : 0u;
  v.cloud_steps = cl.steps; v.cloud_shadow_steps = cl.shadow_steps; v.cloud_octaves = cl.octaves; v.cloud_seed = cl.seed;
  v.cloud_offset_x = cl.offset_x; v.cloud_offset_z = cl.offset_z; v.cloud_density = cl.density;
  v.cloud_noise_shape_scale = cl.noise_shape_scale; v.cloud_noise_detail_scale = cl.noise_detail_scale; v.cloud_noise_weather_scale = cl.noise_weather_scale;
  jendersie_eon_parameters(cl.droplet_diameter, v.cloud_phase);
  const LuminaryCloudLayer* layers[3] = {&cl.low, &cl.mid, &cl.top};
  for (int l = 0; l < 3; l++) {
    const LuminaryCloudLayer& y = *layers[l];
    const float row[10] = {y.active ? 1.0f : 0.0f, y.height_max, y.height_min, y.coverage, y.coverage_min, y.type, y.type_min, y.wind_speed, std::cos(y.wind_angle), std::sin(y.wind_angle)};
    std::memcpy(v.cloud_layers[l], row, sizeof(row));
  }
  v.cloud_noise_shape = nullptr; v.cloud_noise_detail = nullptr; v.cloud_noise_weather = nullptr;
  return std::string();
}

std::string build_device_scene(const HostScene& scene, const std::vector<uint32_t>& bluenoise, DeviceSceneBuffers* out) {
  return update_device_scene(scene, bluenoise, LUMC_DIRTY_ALL, out);
}

}  // namespace lum
