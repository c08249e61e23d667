// Scene file readers (.obj/.mtl, .lum v4). See loaders.cpp for the reference locations of the formats.
#pragma once

#include <string>
#include <vector>

#include "scene.h"

namespace lum {

struct ObjLoadArgs {  // wavefront.h WavefrontArguments
  bool legacy_smoothness = false;
  bool force_transparency_cutout = false;
  float emission_scale = 1.0f;
  bool force_bidirectional_emission = false;
};

// Reads one .obj (+ its .mtl files) into a single mesh and the list of materials it defines (material 0 = default material).
// `material_offset` is added to the material ids stored in the mesh. An .obj without an `o` line yields an empty mesh and a warning.
bool load_obj(const std::string& path, const ObjLoadArgs& args, uint32_t material_offset, HostMesh* mesh, std::vector<LuminaryMaterial>* materials,
              std::vector<std::string>* warnings, std::string* err, std::vector<HostTexture>* textures = nullptr, uint32_t texture_offset = 0);

struct LumFileContent {
  LuminaryRendererSettings settings;
  LuminaryCamera camera;
  LuminaryOcean ocean;
  LuminarySky sky;
  LuminaryCloud cloud;
  LuminaryFog fog;
  LuminaryParticles particles;
  std::vector<std::string> obj_files;  // relative to the .lum file; each also gets an identity instance
  ObjLoadArgs obj_args;
};

bool load_lum_v4(const std::string& path, LumFileContent* content, std::vector<std::string>* warnings, std::string* err);
// A file named inside another file (path_extend + path_apply, path.c): absolute names stand for themselves, relative ones are looked up
// next to the naming file, with either separator in their directory part.
std::string extend_path(const std::string& base_file, const std::string& name);

}  // namespace lum
