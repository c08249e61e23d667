/*
 * Headless frontend on the public C API only (include/luminary_amd.h), in the manner of Mandarin Duck's benchmark mode
 * (`LuminaryMD scene.lum -b <samples> <name> -o <dir>`, reference src/mandarin_duck/main.c + mandarin_duck.c): load a .lum / .obj
 * scene, request the output of a given sample count, render, save the PNG, report the time. It is what "drop-in" means in practice:
 * the same calls a frontend makes against the reference library, linked against libluminary_amd.so instead.
 *
 *   gcc -std=c11 -I include examples/luminary_cli.c -L luminary_amd/lib -lluminary_amd -Wl,-rpath,$PWD/luminary_amd/lib -o luminary_cli
 *   ./luminary_cli scene.lum 64 out.png [width height]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "luminary_amd.h"

#define CHECK(call)                                                                                   \
  do {                                                                                                \
    const LuminaryResult r__ = (call);                                                                \
    if (r__ != LUMINARY_SUCCESS) {                                                                    \
      fprintf(stderr, "%s failed: %s\n", #call, luminary_result_to_string(r__));                     \
      return 2;                                                                                       \
    }                                                                                                 \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s <scene.lum|scene.obj> <samples> <out.png> [width height]\n", argv[0]);
    return 1;
  }
  const char* scene = argv[1];
  const uint32_t samples = (uint32_t) strtoul(argv[2], NULL, 10);

  luminary_init();
  LuminaryHost* host;
  LuminaryHostCreateInfo info;
  memset(&info, 0, sizeof(info));
  info.device_mask = LUMINARY_HOST_CREATE_INFO_DEVICE_MASK_ALL_DEVICES;
  CHECK(luminary_host_create(&host, info));

  LuminaryPath* path;
  CHECK(luminary_path_create(&path));
  CHECK(luminary_path_set_from_string(path, scene));
  const size_t len = strlen(scene);
  if (len > 4 && strcmp(scene + len - 4, ".obj") == 0) CHECK(luminary_host_load_obj_file(host, path));
  else CHECK(luminary_host_load_lum_file(host, path));

  LuminaryRendererSettings settings;
  CHECK(luminary_host_get_settings(host, &settings));
  if (argc >= 6) {
    settings.width = (uint32_t) strtoul(argv[4], NULL, 10);
    settings.height = (uint32_t) strtoul(argv[5], NULL, 10);
  }
  settings.undersampling = 0; /* no preview stages in a batch render */
  CHECK(luminary_host_set_settings(host, &settings));

  LuminaryOutputRequestProperties request;
  memset(&request, 0, sizeof(request));
  request.sample_count = samples;
  request.width = settings.width;
  request.height = settings.height;
  LuminaryOutputPromiseHandle promise;
  CHECK(luminary_host_request_output(host, request, &promise));

  /* the reference renders on its own threads once a scene is loaded; here the additive luminary_ext_render drives the same loop */
  CHECK(luminary_ext_render(host, samples));

  LuminaryOutputHandle output = LUMINARY_OUTPUT_HANDLE_INVALID;
  CHECK(luminary_host_try_await_output(host, promise, &output));
  if (output == LUMINARY_OUTPUT_HANDLE_INVALID) {
    fprintf(stderr, "the requested output was not produced\n");
    return 3;
  }
  LuminaryImage image;
  CHECK(luminary_host_get_image(host, output, &image));
  CHECK(luminary_path_set_from_string(path, argv[3]));
  CHECK(luminary_host_save_png(host, output, path));
  printf("%s: %ux%u, %u samples, %.3f s -> %s\n", scene, image.width, image.height, image.meta_data.sample_count, (double) image.meta_data.time, argv[3]);
  CHECK(luminary_host_release_output(host, output));
  CHECK(luminary_path_destroy(&path));
  CHECK(luminary_host_destroy(&host));
  luminary_shutdown();
  return 0;
}
