// Scene file readers: Wavefront .obj/.mtl and Luminary .lum version 4.
// Semantics follow the reference (format knowledge, not code): src/luminary/host/wavefront.c:25-48 (material defaults),
// :285-423 (.mtl keys), :425-564 (face forms, quads as fans), :566-756 (.obj keys, `usemtl` by name, one mesh per file,
// an `o` line is required), :758-824 (material mapping), :828-996 (flat per-triangle mesh, degenerate removal, face normals
// when `vn` is missing); src/luminary/host/lum.c:51-128 (header), lum_v4.c:18-757 (8-character keys per section).
#include "loaders.h"
#include "output.h"

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace lum {
namespace {

struct ObjMaterial {
  std::string name;
  float kd[3] = {0.9f, 0.9f, 0.9f}, dissolve = 1.0f, ks[3] = {0, 0, 0}, ns = 300.0f, ke[3] = {0, 0, 0}, ni = 1.0f;
  std::string map[5];  // albedo (map_Kd), luminance (map_Ke), roughness (map_Ns), metallic (map_refl), normal (map_Bump); paths relative to the .mtl
};
struct ObjTri { int32_t v[3], vt[3], vn[3]; uint16_t material; };

std::string dirname_of(const std::string& p) {
  const size_t s = p.find_last_of("/\\");
  return s == std::string::npos ? std::string() : p.substr(0, s + 1);
}
}  // namespace

// path_extend + path_apply (path.c:8-17, :60-115): a file named inside another file. An absolute name (leading '/', or a drive designator)
// stands for itself; a relative one is looked up next to the file that names it, its directory part read with either separator.
std::string extend_path(const std::string& base_file, const std::string& name) {
  if (!name.empty() && (name[0] == '/' || name.find(':') != std::string::npos)) return name;
  std::string rel = name;
  const size_t last = rel.find_last_of("/\\");
  if (last != std::string::npos)
    for (size_t i = 0; i < last; i++)
      if (rel[i] == '\\') rel[i] = '/';
  if (last != std::string::npos && rel[last] == '\\') rel[last] = '/';
  const size_t s = base_file.find_last_of("/\\");
  return (s == std::string::npos ? std::string() : base_file.substr(0, s + 1)) + rel;
}

namespace {
std::string trim(const std::string& s) {
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t')) a++;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r' || s[b - 1] == '\n')) b--;
  return s.substr(a, b - a);
}
uint32_t read_floats(const char* str, uint32_t n, float* dst) {
  const char* p = str;
  uint32_t got = 0;
  for (uint32_t i = 0; i < n; i++) {
    char* end = nullptr;
    const float v = std::strtof(p, &end);
    if (end == p) break;
    dst[i] = v; got++; p = end;
  }
  return got;
}

// Face forms accepted by the reference: v, v/vt, v/vt/vn with 3 or 4 corners (wavefront.c:425-564). "v//vn" yields three
// numbers per corner only when vt is present, exactly like the reference's digit scanner, so it is parsed the same way:
// every run of digits (optionally signed) terminated by '/', ' ' or end of line is one number.
uint32_t parse_face(const char* str, ObjTri* f1, ObjTri* f2) {
  int32_t data[12];
  uint32_t n = 0;
  const char* p = str;
  while (*p && !((*p >= '0' && *p <= '9') || *p == '-')) { if (*p == '\r' || *p == '\n') break; p++; }
  int32_t sign = 1;
  while (*p && *p != '\r' && *p != '\n') {
    if (*p == '-') sign = -1;
    int32_t value = 0;
    bool digits = false;
    while (*p >= '0' && *p <= '9') { value = value * 10 + (*p - '0'); p++; digits = true; }
    (void) digits;
    if (*p == '/' || *p == ' ' || *p == '\0' || *p == '\r' || *p == '\n') {
      if (n < 12) data[n] = value * sign;
      n++;
      sign = 1;
    }
    if (*p == '\0' || *p == '\r' || *p == '\n') break;
    p++;
  }
  std::memset(f1, 0, sizeof(*f1)); std::memset(f2, 0, sizeof(*f2));
  auto corner = [&](ObjTri* f, int k, int stride, int idx) {
    f->v[k] = data[idx * stride];
    if (stride >= 2) f->vt[k] = data[idx * stride + 1];
    if (stride >= 3) f->vn[k] = data[idx * stride + 2];
  };
  int stride = 0, corners = 0;
  switch (n) {
    case 3: stride = 1; corners = 3; break;
    case 4: stride = 1; corners = 4; break;
    case 6: stride = 2; corners = 3; break;
    case 8: stride = 2; corners = 4; break;
    case 9: stride = 3; corners = 3; break;
    case 12: stride = 3; corners = 4; break;
    default: return 0;
  }
  corner(f1, 0, stride, 0); corner(f1, 1, stride, 1); corner(f1, 2, stride, 2);
  if (corners == 4) { corner(f2, 0, stride, 0); corner(f2, 1, stride, 2); corner(f2, 2, stride, 3); return 2; }
  return 1;
}

bool read_mtl(const std::string& path, float emission_scale, std::vector<ObjMaterial>* mats, std::string* err) {
  std::ifstream in(path);
  if (!in) { *err = "Failed to open *.mtl file (" + path + ")"; return false; }
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    const char* l = line.c_str();
    ObjMaterial& cur = mats->back();
    float v[3];
    if (line.compare(0, 6, "newmtl") == 0) { ObjMaterial m; m.name = trim(line.size() > 7 ? line.substr(7) : std::string()); mats->push_back(m); }
    else if (l[0] == 'K' && l[1] == 'd') { if (read_floats(l + 3, 3, v) == 3) { cur.kd[0] = v[0]; cur.kd[1] = v[1]; cur.kd[2] = v[2]; } }
    else if (l[0] == 'd') { if (read_floats(l + 2, 1, v)) cur.dissolve = v[0]; }
    else if (l[0] == 'K' && l[1] == 's') { if (read_floats(l + 3, 3, v) == 3) { cur.ks[0] = v[0]; cur.ks[1] = v[1]; cur.ks[2] = v[2]; } }
    else if (l[0] == 'N' && l[1] == 's') { if (read_floats(l + 3, 1, v)) cur.ns = v[0]; }
    else if (l[0] == 'K' && l[1] == 'e') {  // scaled here and again through emission_scale, as the reference does (wavefront.c:385-396, :808)
      if (read_floats(l + 3, 3, v) == 3) { cur.ke[0] = v[0] * emission_scale; cur.ke[1] = v[1] * emission_scale; cur.ke[2] = v[2] * emission_scale; }
    }
    else if (l[0] == 'N' && l[1] == 'i') { if (read_floats(l + 3, 1, v)) cur.ni = v[0]; }
    else if (line.compare(0, 4, "map_") == 0 && line.size() >= 8) {
      // wavefront.c:159-245: the map kind, optional "-option args..." groups, then the path
      int kind = -1;
      size_t off = 7;
      if (line.compare(4, 2, "Kd") == 0) kind = 0;
      else if (line.compare(4, 2, "Ke") == 0) kind = 1;
      else if (line.compare(4, 2, "Ns") == 0) kind = 2;
      else if (line.compare(4, 4, "refl") == 0) { kind = 3; off = 9; }
      else if (line.compare(4, 4, "Bump") == 0) { kind = 4; off = 9; }
      if (kind >= 0 && off < line.size()) {
        std::string rest = trim(line.substr(off));
        while (!rest.empty() && rest[0] == '-') {
          const char c0 = rest.size() > 1 ? rest[1] : ' ';
          uint32_t num_args = 0;
          if (c0 == 'o' || c0 == 's') num_args = 3;
          else if (c0 == 't') num_args = (rest.size() > 2 && rest[2] == ' ') ? 3 : 1;
          else if (c0 == 'm') num_args = 2;
          else if (c0 == 'c' || c0 == 'b') num_args = 1;
          for (uint32_t a = 0; a <= num_args && !rest.empty(); a++) {
            const size_t sp = rest.find(' ');
            rest = (sp == std::string::npos) ? std::string() : trim(rest.substr(sp + 1));
          }
        }
        if (!rest.empty()) cur.map[kind] = rest;
      }
    }
  }
  return true;
}

}  // namespace

bool load_obj(const std::string& path, const ObjLoadArgs& args, uint32_t material_offset, HostMesh* mesh_out, std::vector<LuminaryMaterial>* materials_out,
              std::vector<std::string>* warnings, std::string* err, std::vector<HostTexture>* textures_out, uint32_t texture_offset) {
  std::ifstream in(path);
  if (!in) { *err = "File " + path + " could not be opened!"; return false; }
  std::vector<float> verts, normals, uvs;
  std::vector<ObjTri> tris;
  std::vector<ObjMaterial> mats(1);  // material 0 of every file is the default one (wavefront.c:64-68)
  std::vector<std::string> loaded_mtls, object_names;
  std::string mtl_dir = dirname_of(path);
  uint16_t current_material = 0;
  std::string line;
  while (std::getline(in, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    const char* l = line.c_str();
    float v[3] = {0, 0, 0};
    if (l[0] == 'v' && l[1] == ' ') { read_floats(l + 2, 3, v); verts.insert(verts.end(), v, v + 3); }
    else if (l[0] == 'v' && l[1] == 'n') { read_floats(l + 3, 3, v); normals.insert(normals.end(), v, v + 3); }
    else if (l[0] == 'v' && l[1] == 't') { read_floats(l + 3, 2, v); uvs.insert(uvs.end(), v, v + 2); }
    else if (l[0] == 'f') {
      ObjTri a, b;
      const uint32_t n = parse_face(l, &a, &b);
      if (n == 0) warnings->push_back("A face is of unsupported format. " + line);
      if (n >= 1) { a.material = current_material; tris.push_back(a); }
      if (n >= 2) { b.material = current_material; tris.push_back(b); }
    }
    else if (l[0] == 'o') { object_names.push_back(trim(line.size() > 1 ? line.substr(1) : std::string())); }
    else if (line.compare(0, 6, "mtllib") == 0) {
      const std::string name = trim(line.substr(6));
      bool seen = false;
      for (auto& s : loaded_mtls) seen |= (s == name);
      if (!seen) {
        loaded_mtls.push_back(name);
        if (!read_mtl(extend_path(path, name), args.emission_scale, &mats, err)) return false;
        mtl_dir = dirname_of(extend_path(path, name));
      }
    }
    else if (line.compare(0, 6, "usemtl") == 0) {
      const std::string name = trim(line.substr(6));
      current_material = 0;
      for (size_t m = 1; m < mats.size(); m++) if (mats[m].name == name) { current_material = (uint16_t) m; break; }
    }
  }
  if (object_names.empty()) {  // wavefront.c:843-848
    warnings->push_back("Wavefront file contained no objects.");
    mesh_out->material_ids.clear();
    return true;
  }
  // materials (wavefront.c:758-824)
  std::vector<std::string> texture_files;
  for (size_t m = 0; m < mats.size(); m++) {
    const ObjMaterial& w = mats[m];
    LuminaryMaterial mat;
    default_material(&mat);
    mat.id = material_offset + (uint32_t) m;
    mat.base_substrate = LUMINARY_MATERIAL_BASE_SUBSTRATE_OPAQUE;
    mat.albedo.r = w.kd[0]; mat.albedo.g = w.kd[1]; mat.albedo.b = w.kd[2]; mat.albedo.a = w.dissolve;
    mat.emission.r = w.ke[0]; mat.emission.g = w.ke[1]; mat.emission.b = w.ke[2];
    mat.emission_scale = args.emission_scale;
    mat.refraction_index = w.ni;
    mat.roughness = 1.0f - w.ns / 1000.0f;
    mat.roughness_clamp = 0.25f;
    mat.roughness_as_smoothness = args.legacy_smoothness;
    mat.emission_active = (w.ke[0] > 0.0f) || (w.ke[1] > 0.0f) || (w.ke[2] > 0.0f);
    mat.thin_walled = false;
    mat.normal_map_is_compressed = true;
    mat.bidirectional_emission = args.force_bidirectional_emission;
    mat.metallic = w.ks[0] > 0.5f;
    // wavefront.c:787-818: one texture per distinct file. A metallic map is stored but, as in the reference (geometry_utils.cuh:152-160), not
    // evaluated: its presence alone makes the material non-metallic.
    uint16_t* slot[5] = {&mat.albedo_tex, &mat.luminance_tex, &mat.roughness_tex, &mat.metallic_tex, &mat.normal_tex};
    for (int k = 0; k < 5; k++) {
      if (w.map[k].empty()) continue;
      if (!textures_out) { warnings->push_back("texture " + w.map[k] + " ignored: no texture store"); continue; }
      uint32_t id = 0xFFFF;
      for (size_t t = 0; t < texture_files.size(); t++) if (texture_files[t] == w.map[k]) id = (uint32_t) t;
      if (id == 0xFFFF) {
        HostTexture tex;
        std::string terr;
        if (!read_png(extend_path(mtl_dir + "x", w.map[k]), &tex.width, &tex.height, &tex.gamma, &tex.texels, &terr)) { warnings->push_back("texture ignored: " + terr); continue; }
        if (texture_offset + textures_out->size() >= 0xFFFF) { warnings->push_back("Exceeded limit of 65535 textures."); continue; }
        textures_out->push_back(std::move(tex));
        texture_files.push_back(w.map[k]);
        id = (uint32_t) texture_files.size() - 1;
      }
      *slot[k] = (uint16_t) (texture_offset + id);
    }
    if (mat.luminance_tex != 0xFFFF) mat.emission_active = true;  // wavefront.c:809
    materials_out->push_back(mat);
  }
  // mesh (wavefront.c:828-996)
  const uint32_t vertex_count = (uint32_t) (verts.size() / 3), uv_count = (uint32_t) (uvs.size() / 2), normal_count = (uint32_t) (normals.size() / 3);
  HostMesh& mesh = *mesh_out;
  mesh.name = object_names[0];
  mesh.positions.clear(); mesh.normals.clear(); mesh.uvs.clear(); mesh.material_ids.clear();
  auto resolve = [](int32_t idx, uint32_t count) -> uint32_t { return (idx > 0) ? (uint32_t) (idx - 1) : (uint32_t) (idx + (int32_t) count); };
  for (const ObjTri& t : tris) {
    const uint32_t i1 = resolve(t.v[0], vertex_count), i2 = resolve(t.v[1], vertex_count), i3 = resolve(t.v[2], vertex_count);
    if (i1 >= vertex_count || i2 >= vertex_count || i3 >= vertex_count) continue;
    const float* v1 = &verts[3 * (size_t) i1]; const float* v2 = &verts[3 * (size_t) i2]; const float* v3 = &verts[3 * (size_t) i3];
    const float e1[3] = {v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2]}, e2[3] = {v3[0] - v1[0], v3[1] - v1[1], v3[2] - v1[2]};
    if (std::fabs(e1[0]) < FLT_EPSILON && std::fabs(e1[1]) < FLT_EPSILON && std::fabs(e1[2]) < FLT_EPSILON && std::fabs(e2[0]) < FLT_EPSILON
        && std::fabs(e2[1]) < FLT_EPSILON && std::fabs(e2[2]) < FLT_EPSILON)
      continue;
    mesh.positions.insert(mesh.positions.end(), v1, v1 + 3); mesh.positions.insert(mesh.positions.end(), v2, v2 + 3);
    mesh.positions.insert(mesh.positions.end(), v3, v3 + 3);
    float fn[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    const float frl = 1.0f / std::sqrt(fn[0] * fn[0] + fn[1] * fn[1] + fn[2] * fn[2]);
    if (!std::isnan(frl) && !std::isinf(frl)) { fn[0] *= frl; fn[1] *= frl; fn[2] *= frl; }
    for (int k = 0; k < 3; k++) {
      const uint32_t ti = resolve(t.vt[k], uv_count);
      if (ti < uv_count) { mesh.uvs.push_back(uvs[2 * (size_t) ti]); mesh.uvs.push_back(uvs[2 * (size_t) ti + 1]); }
      else { mesh.uvs.push_back(0.0f); mesh.uvs.push_back(0.0f); }
    }
    for (int k = 0; k < 3; k++) {
      const uint32_t ni = resolve(t.vn[k], normal_count);
      float n[3] = {fn[0], fn[1], fn[2]};
      if (ni < normal_count) { n[0] = normals[3 * (size_t) ni]; n[1] = normals[3 * (size_t) ni + 1]; n[2] = normals[3 * (size_t) ni + 2]; }
      const float rl = 1.0f / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
      if (std::isnan(rl) || std::isinf(rl)) { n[0] = fn[0]; n[1] = fn[1]; n[2] = fn[2]; }
      else { n[0] *= rl; n[1] *= rl; n[2] *= rl; }
      mesh.normals.insert(mesh.normals.end(), n, n + 3);
    }
    mesh.material_ids.push_back((uint16_t) (material_offset + t.material));
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// .lum v4
// ---------------------------------------------------------------------------------------------------------------------
namespace {

enum ValKind { kU32, kF32, kF32x2, kF32x3, kBool, kIgnore };
struct KeyDesc { const char* section; const char* key; ValKind kind; size_t offset; size_t offset2; size_t offset3; };

#define OFF(T, f) offsetof(T, f)
// One row per key the version-4 parser understands (lum_v4.c:25-668). Legacy keys that the reference accepts and ignores are kIgnore.
const KeyDesc kSettingsKeys[] = {
  {"G", "WIDTH___", kU32, OFF(LuminaryRendererSettings, width), 0, 0}, {"G", "HEIGHT__", kU32, OFF(LuminaryRendererSettings, height), 0, 0},
  {"G", "BOUNCES_", kU32, OFF(LuminaryRendererSettings, max_ray_depth), 0, 0}, {"G", "NUMLIGHT", kIgnore, 0, 0, 0}};
const KeyDesc kCameraKeys[] = {
  {"CA", "POSITION", kF32x3, OFF(LuminaryCamera, pos.x), OFF(LuminaryCamera, pos.y), OFF(LuminaryCamera, pos.z)},
  {"CA", "ROTATION", kF32x3, OFF(LuminaryCamera, rotation.x), OFF(LuminaryCamera, rotation.y), OFF(LuminaryCamera, rotation.z)},
  {"CA", "FOV_____", kF32, OFF(LuminaryCamera, thin_lens.fov), 0, 0}, {"CA", "FOCALLEN", kF32, OFF(LuminaryCamera, object_distance), 0, 0},
  {"CA", "APERTURE", kF32, OFF(LuminaryCamera, thin_lens.aperture_size), 0, 0}, {"CA", "APESHAPE", kU32, OFF(LuminaryCamera, aperture_shape), 0, 0},
  {"CA", "APEBLACO", kU32, OFF(LuminaryCamera, aperture_blade_count), 0, 0}, {"CA", "AUTOEXP_", kIgnore, 0, 0, 0},
  {"CA", "MINEXPOS", kIgnore, 0, 0, 0}, {"CA", "MAXEXPOS", kIgnore, 0, 0, 0}, {"CA", "BLOOMBLE", kF32, OFF(LuminaryCamera, bloom_blend), 0, 0},
  {"CA", "LENSFLAR", kIgnore, 0, 0, 0}, {"CA", "LENSFTHR", kIgnore, 0, 0, 0}, {"CA", "DITHER__", kBool, OFF(LuminaryCamera, dithering), 0, 0},
  {"CA", "TONEMAP_", kU32, OFF(LuminaryCamera, tonemap), 0, 0}, {"CA", "AGXSLOPE", kF32, OFF(LuminaryCamera, agx_custom_slope), 0, 0},
  {"CA", "AGXPOWER", kF32, OFF(LuminaryCamera, agx_custom_power), 0, 0}, {"CA", "AGXSATUR", kF32, OFF(LuminaryCamera, agx_custom_saturation), 0, 0},
  {"CA", "FILTER__", kU32, OFF(LuminaryCamera, filter), 0, 0}, {"CA", "PURKINJE", kBool, OFF(LuminaryCamera, purkinje), 0, 0},
  {"CA", "RUSSIANR", kF32, OFF(LuminaryCamera, russian_roulette_threshold), 0, 0}, {"CA", "FIREFLYC", kIgnore, 0, 0, 0},
  {"CA", "FILMGRAI", kF32, OFF(LuminaryCamera, film_grain), 0, 0}};
const KeyDesc kSkyKeys[] = {
  {"S", "MODE____", kU32, OFF(LuminarySky, mode), 0, 0},
  {"S", "OFFSET__", kF32x3, OFF(LuminarySky, geometry_offset.x), OFF(LuminarySky, geometry_offset.y), OFF(LuminarySky, geometry_offset.z)},
  {"S", "MOONALTI", kF32, OFF(LuminarySky, moon_altitude), 0, 0}, {"S", "MOONAZIM", kF32, OFF(LuminarySky, moon_azimuth), 0, 0},
  {"S", "MOONTEXO", kF32, OFF(LuminarySky, moon_tex_offset), 0, 0}, {"S", "SUNSTREN", kF32, OFF(LuminarySky, sun_strength), 0, 0},
  {"S", "OZONEABS", kBool, OFF(LuminarySky, ozone_absorption), 0, 0}, {"S", "STEPS___", kU32, OFF(LuminarySky, steps), 0, 0},
  {"S", "STARSEED", kU32, OFF(LuminarySky, stars_seed), 0, 0}, {"S", "STARINTE", kF32, OFF(LuminarySky, stars_intensity), 0, 0},
  {"S", "STARNUM_", kU32, OFF(LuminarySky, stars_count), 0, 0}, {"S", "AZIMUTH_", kF32, OFF(LuminarySky, azimuth), 0, 0},
  {"S", "ALTITUDE", kF32, OFF(LuminarySky, altitude), 0, 0}, {"S", "DENSITY_", kF32, OFF(LuminarySky, base_density), 0, 0},
  {"S", "RAYLEDEN", kF32, OFF(LuminarySky, rayleigh_density), 0, 0}, {"S", "MIEDENSI", kF32, OFF(LuminarySky, mie_density), 0, 0},
  {"S", "OZONEDEN", kF32, OFF(LuminarySky, ozone_density), 0, 0}, {"S", "RAYLEFAL", kF32, OFF(LuminarySky, rayleigh_falloff), 0, 0},
  {"S", "MIEFALLO", kF32, OFF(LuminarySky, mie_falloff), 0, 0}, {"S", "GROUNDVI", kF32, OFF(LuminarySky, ground_visibility), 0, 0},
  {"S", "DIAMETER", kF32, OFF(LuminarySky, mie_diameter), 0, 0}, {"S", "OZONETHI", kF32, OFF(LuminarySky, ozone_layer_thickness), 0, 0},
  {"S", "MSFACTOR", kF32, OFF(LuminarySky, multiscattering_factor), 0, 0}, {"S", "AERIALPE", kBool, OFF(LuminarySky, aerial_perspective), 0, 0},
  {"S", "HDRISAMP", kU32, OFF(LuminarySky, hdri_samples), 0, 0}, {"S", "HDRIORIG", kIgnore, 0, 0, 0}, {"S", "HDRIMIPB", kIgnore, 0, 0, 0},
  {"S", "COLORCON", kF32x3, OFF(LuminarySky, constant_color.r), OFF(LuminarySky, constant_color.g), OFF(LuminarySky, constant_color.b)}};
#define CLOUD_LAYER(P, L)                                                                                                            \
  {"CL", P "ACTIV", kBool, OFF(LuminaryCloud, L.active), 0, 0},                                                                       \
    {"CL", P "COVER", kF32x2, OFF(LuminaryCloud, L.coverage_min), OFF(LuminaryCloud, L.coverage), 0},                                \
    {"CL", P "TYPE_", kF32x2, OFF(LuminaryCloud, L.type_min), OFF(LuminaryCloud, L.type), 0},                                        \
    {"CL", P "HEIGH", kF32x2, OFF(LuminaryCloud, L.height_min), OFF(LuminaryCloud, L.height_max), 0},                                \
    {"CL", P "WIND_", kF32x2, OFF(LuminaryCloud, L.wind_speed), OFF(LuminaryCloud, L.wind_angle), 0}
const KeyDesc kCloudKeys[] = {
  {"CL", "ACTIVE__", kBool, OFF(LuminaryCloud, active), 0, 0}, {"CL", "INSCATTE", kBool, OFF(LuminaryCloud, atmosphere_scattering), 0, 0},
  {"CL", "MIPMAPBI", kF32, OFF(LuminaryCloud, mipmap_bias), 0, 0}, {"CL", "SEED____", kU32, OFF(LuminaryCloud, seed), 0, 0},
  {"CL", "OFFSET__", kF32x2, OFF(LuminaryCloud, offset_x), OFF(LuminaryCloud, offset_z), 0},
  {"CL", "SHASCALE", kF32, OFF(LuminaryCloud, noise_shape_scale), 0, 0}, {"CL", "DETSCALE", kF32, OFF(LuminaryCloud, noise_detail_scale), 0, 0},
  {"CL", "WEASCALE", kF32, OFF(LuminaryCloud, noise_weather_scale), 0, 0}, {"CL", "DIAMETER", kF32, OFF(LuminaryCloud, droplet_diameter), 0, 0},
  {"CL", "SHASTEPS", kU32, OFF(LuminaryCloud, shadow_steps), 0, 0}, {"CL", "STEPS___", kU32, OFF(LuminaryCloud, steps), 0, 0},
  {"CL", "DENSITY_", kF32, OFF(LuminaryCloud, density), 0, 0},
  CLOUD_LAYER("LOW", low), CLOUD_LAYER("MID", mid), CLOUD_LAYER("TOP", top)};
const KeyDesc kFogKeys[] = {
  {"F", "ACTIVE__", kBool, OFF(LuminaryFog, active), 0, 0}, {"F", "DENSITY_", kF32, OFF(LuminaryFog, density), 0, 0},
  {"F", "DIAMETER", kF32, OFF(LuminaryFog, droplet_diameter), 0, 0}, {"F", "DISTANCE", kF32, OFF(LuminaryFog, dist), 0, 0},
  {"F", "HEIGHT__", kF32, OFF(LuminaryFog, height), 0, 0}};
const KeyDesc kOceanKeys[] = {
  {"O", "ACTIVE__", kBool, OFF(LuminaryOcean, active), 0, 0}, {"O", "HEIGHT__", kF32, OFF(LuminaryOcean, height), 0, 0},
  {"O", "AMPLITUD", kF32, OFF(LuminaryOcean, amplitude), 0, 0}, {"O", "FREQUENC", kF32, OFF(LuminaryOcean, frequency), 0, 0},
  {"O", "CHOPPY__", kIgnore, 0, 0, 0}, {"O", "REFRACT_", kF32, OFF(LuminaryOcean, refractive_index), 0, 0},
  {"O", "WATERTYP", kU32, OFF(LuminaryOcean, water_type), 0, 0}, {"O", "CAUSACTI", kBool, OFF(LuminaryOcean, caustics_active), 0, 0},
  {"O", "CAUSRISS", kU32, OFF(LuminaryOcean, caustics_ris_sample_count), 0, 0}, {"O", "CAUSSCAL", kF32, OFF(LuminaryOcean, caustics_domain_scale), 0, 0},
  {"O", "MULTISCA", kBool, OFF(LuminaryOcean, multiscattering), 0, 0}, {"O", "LIGHTSON", kBool, OFF(LuminaryOcean, triangle_light_contribution), 0, 0}};
const KeyDesc kParticleKeys[] = {
  {"P", "ACTIVE__", kBool, OFF(LuminaryParticles, active), 0, 0}, {"P", "SCALE___", kF32, OFF(LuminaryParticles, scale), 0, 0},
  {"P", "ALBEDO__", kF32x3, OFF(LuminaryParticles, albedo.r), OFF(LuminaryParticles, albedo.g), OFF(LuminaryParticles, albedo.b)},
  {"P", "DIRECTIO", kF32x2, OFF(LuminaryParticles, direction_altitude), OFF(LuminaryParticles, direction_azimuth), 0},
  {"P", "SPEED___", kF32, OFF(LuminaryParticles, speed), 0, 0}, {"P", "PHASEDIA", kF32, OFF(LuminaryParticles, phase_diameter), 0, 0},
  {"P", "SEED____", kU32, OFF(LuminaryParticles, seed), 0, 0}, {"P", "COUNT___", kU32, OFF(LuminaryParticles, count), 0, 0},
  {"P", "SIZE____", kF32, OFF(LuminaryParticles, size), 0, 0}, {"P", "SIZEVARI", kF32, OFF(LuminaryParticles, size_variation), 0, 0}};

template <size_t N>
bool apply_key(const KeyDesc (&table)[N], void* base, const char* key, const char* value) {
  for (const KeyDesc& d : table) {
    if (std::strncmp(d.key, key, 8) != 0) continue;
    char* b = (char*) base;
    switch (d.kind) {
      case kU32: std::sscanf(value, "%u", (uint32_t*) (b + d.offset)); break;
      case kF32: std::sscanf(value, "%f", (float*) (b + d.offset)); break;
      case kF32x2: std::sscanf(value, "%f %f", (float*) (b + d.offset), (float*) (b + d.offset2)); break;
      case kF32x3: std::sscanf(value, "%f %f %f", (float*) (b + d.offset), (float*) (b + d.offset2), (float*) (b + d.offset3)); break;
      case kBool: { uint32_t u = 0; std::sscanf(value, "%u", &u); *(bool*) (b + d.offset) = u != 0; } break;
      case kIgnore: break;
    }
    return true;
  }
  return false;
}

}  // namespace

bool load_lum_v4(const std::string& path, LumFileContent* content, std::vector<std::string>* warnings, std::string* err) {
  FILE* file = std::fopen(path.c_str(), "rb");
  if (!file) { *err = "File " + path + " could not be opened."; return false; }
  char line[4096];
  if (!std::fgets(line, sizeof(line), file) || std::strncmp(line, "Luminary", 8) != 0) { std::fclose(file); *err = "File is not a Luminary file."; return false; }
  uint32_t version = 0;
  if (!std::fgets(line, sizeof(line), file) || !(line[0] == 'v' || line[0] == 'V')) { std::fclose(file); *err = "Luminary file has no version information."; return false; }
  std::sscanf(line, "%*s %u", &version);
  if (version != 4) {
    std::fclose(file);
    *err = (version == 5) ? "Luminary file version 5 is not supported (the reference only prints it, lum_v5.c:38-43)."
                          : "Luminary file version is not supported (supported: 4).";
    return false;
  }
  content->camera.use_physical_camera = false;  // lum_v4.c:699
  bool force_no_bloom = false;
  content->obj_args = ObjLoadArgs();
  content->obj_args.force_bidirectional_emission = true;  // lum_v4.c:752
  while (std::fgets(line, sizeof(line), file)) {
    const char c0 = line[0], c1 = line[1];
    if (c0 == '#' || c0 == '\n' || c0 == '\r' || c0 == 'T') continue;
    const char* space = std::strchr(line, ' ');
    if (!space || std::strlen(space + 1) < 8) { warnings->push_back(std::string("Scene file contains unknown line! Content: ") + line); continue; }
    const char* key = space + 1;
    const char* value = (std::strlen(key) > 9) ? key + 9 : "";
    bool ok = true;
    if (c0 == 'G') {
      if (std::strncmp(key, "MESHFILE", 8) == 0) {
        char name[4096]; name[0] = '\0';
        std::sscanf(value, "%4095s", name);
        content->obj_files.push_back(name);  // each mesh file also gets an identity instance (lum_v4.c:31-39)
      }
      else ok = apply_key(kSettingsKeys, &content->settings, key, value);
    }
    else if (c0 == 'M') {
      uint32_t u = 0;
      if (std::strncmp(key, "EMISSION", 8) == 0) std::sscanf(value, "%f", &content->obj_args.emission_scale);
      else if (std::strncmp(key, "COLORTRA", 8) == 0) { std::sscanf(value, "%u", &u); content->obj_args.force_transparency_cutout = u != 0; }
      else if (std::strncmp(key, "IORSHADO", 8) == 0) {}
      // the constant the reference compares with spells INVERTRO (its comment says INTERTRO, lum_v4.c:125-129); found by the pin against lum_v4.c
      else if (std::strncmp(key, "INVERTRO", 8) == 0) { std::sscanf(value, "%u", &u); content->obj_args.legacy_smoothness = u != 0; }
      else ok = false;
    }
    else if (c0 == 'C' && c1 == 'A') {
      if (std::strncmp(key, "EXPOSURE", 8) == 0) { std::sscanf(value, "%f", &content->camera.exposure); content->camera.exposure = std::log(content->camera.exposure); }
      else if (std::strncmp(key, "BLOOM___", 8) == 0) { uint32_t u = 0; std::sscanf(value, "%u", &u); force_no_bloom = (u == 0); }
      else ok = apply_key(kCameraKeys, &content->camera, key, value);
    }
    else if (c0 == 'S') {
      if (std::strncmp(key, "HDRIDIM_", 8) == 0) { std::sscanf(value, "%u", &content->sky.hdri_dim); if (content->sky.hdri_dim == 0) content->sky.hdri_dim = 1; }
      else ok = apply_key(kSkyKeys, &content->sky, key, value);
    }
    else if (c0 == 'C' && c1 == 'L') ok = apply_key(kCloudKeys, &content->cloud, key, value);
    else if (c0 == 'F') ok = apply_key(kFogKeys, &content->fog, key, value);
    else if (c0 == 'O') ok = apply_key(kOceanKeys, &content->ocean, key, value);
    else if (c0 == 'P') ok = apply_key(kParticleKeys, &content->particles, key, value);
    else { warnings->push_back(std::string("Scene file contains unknown line! Content: ") + line); continue; }
    if (!ok) warnings->push_back(std::string(key, 8) + " is not a valid setting of its section.");
  }
  std::fclose(file);
  if (force_no_bloom) content->camera.bloom_blend = 0.0f;
  return true;
}

}  // namespace lum
