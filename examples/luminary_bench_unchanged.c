/*
 * What Mandarin Duck's benchmark mode does (`LuminaryMD scene.lum -b <log2 samples> <name> -o <dir>`; reference
 * src/mandarin_duck/mandarin_duck.c:53-98 queues the outputs, :186-244 runs the loop), written against the REFERENCE's public headers
 * only: no luminary_ext_* call, includes spelled <luminary/...> as a frontend spells them. The library renders on its own thread after
 * luminary_host_start_new_render; this program only polls luminary_host_try_await_output, like the original.
 *
 *   gcc -std=c11 -I include examples/luminary_bench_unchanged.c -L luminary_amd/lib -lluminary_amd -Wl,-rpath,$PWD/luminary_amd/lib -o lum_bench
 *   ./lum_bench scene.lum 5 name outdir [width height]
 */
#include <luminary/luminary.h>
#include <luminary/host.h>
#include <luminary/structs.h>
#include <luminary/path.h>
#include <luminary/error.h>
#include <luminary/array.h>
#include <luminary/thread_status.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LUM_FAILURE_HANDLE(command)                                                        \
  {                                                                                        \
    LuminaryResult __lum_func_err = command;                                               \
    if (__lum_func_err != LUMINARY_SUCCESS) {                                              \
      fprintf(stderr, "%s failed: %s\n", #command, luminary_result_to_string(__lum_func_err)); \
      exit(2);                                                                             \
    }                                                                                      \
  }

int main(int argc, char** argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s <scene.lum> <log2 samples> <name> <outdir> [width height]\n", argv[0]);
    return 1;
  }
  const uint32_t num_outputs = (uint32_t) strtoul(argv[2], NULL, 10);
  const char* name = argv[3];
  const char* outdir = argv[4];

  luminary_init();
  LuminaryHost* host;
  LuminaryHostCreateInfo info;
  info.device_mask = LUMINARY_HOST_CREATE_INFO_DEVICE_MASK_ALL_DEVICES;
  LUM_FAILURE_HANDLE(luminary_host_create(&host, info));

  LuminaryPath* path;
  LUM_FAILURE_HANDLE(luminary_path_create(&path));
  LUM_FAILURE_HANDLE(luminary_path_set_from_string(path, argv[1]));
  LUM_FAILURE_HANDLE(luminary_host_load_lum_file(host, path));
  LUM_FAILURE_HANDLE(luminary_path_destroy(&path));

  LuminaryRendererSettings settings;
  LUM_FAILURE_HANDLE(luminary_host_get_settings(host, &settings));
  if (argc >= 7) {
    settings.width = (uint32_t) strtoul(argv[5], NULL, 10);
    settings.height = (uint32_t) strtoul(argv[6], NULL, 10);
    LUM_FAILURE_HANDLE(luminary_host_set_settings(host, &settings));
  }

  /* _mandarin_duck_queue_benchmark_outputs: 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, then every 32 samples up to 2^n */
  LuminaryOutputPromiseHandle* promises;
  LUM_FAILURE_HANDLE(array_create(&promises, sizeof(LuminaryOutputPromiseHandle), 16));
  const uint32_t num_exponential = (num_outputs < 5) ? num_outputs : 5;
  for (uint32_t id = 0; id <= num_exponential; id++) {
    LuminaryOutputRequestProperties props;
    LuminaryOutputPromiseHandle handle;
    props.sample_count = 1u << id;
    props.width = settings.width;
    props.height = settings.height;
    LUM_FAILURE_HANDLE(luminary_host_request_output(host, props, &handle));
    LUM_FAILURE_HANDLE(array_push(&promises, &handle));
    if (id >= 2) {
      props.sample_count = (1u << (id - 1)) + (1u << (id - 2));
      LUM_FAILURE_HANDLE(luminary_host_request_output(host, props, &handle));
      LUM_FAILURE_HANDLE(array_push(&promises, &handle));
    }
  }
  if (num_outputs > 5) {
    for (uint32_t sample_count = 1u << 6; sample_count <= (1u << num_outputs); sample_count += 32) {
      LuminaryOutputRequestProperties props;
      LuminaryOutputPromiseHandle handle;
      props.sample_count = sample_count;
      props.width = settings.width;
      props.height = settings.height;
      LUM_FAILURE_HANDLE(luminary_host_request_output(host, props, &handle));
      LUM_FAILURE_HANDLE(array_push(&promises, &handle));
    }
  }
  LuminaryOutputProperties recurring;
  memset(&recurring, 0, sizeof(recurring));
  recurring.enabled = false;
  LUM_FAILURE_HANDLE(luminary_host_set_output_properties(host, recurring));

  char file[4096];
  snprintf(file, sizeof(file), "%s/BenchResults-%s.txt", outdir, name);
  FILE* times = fopen(file, "wb");
  if (!times) { fprintf(stderr, "cannot open %s\n", file); return 3; }

  uint32_t num_promises;
  LUM_FAILURE_HANDLE(array_get_num_elements(promises, &num_promises));
  LUM_FAILURE_HANDLE(luminary_host_start_new_render(host));

  uint32_t obtained = 0;
  while (obtained != num_promises) {
    for (uint32_t id = 0; id < num_promises; id++) {
      const LuminaryOutputPromiseHandle promise = promises[id];
      if (promise == LUMINARY_OUTPUT_HANDLE_INVALID) continue;
      LuminaryOutputHandle output;
      luminary_host_try_await_output(host, promise, &output);
      if (output == LUMINARY_OUTPUT_HANDLE_INVALID) continue;
      LuminaryImage image;
      LUM_FAILURE_HANDLE(luminary_host_get_image(host, output, &image));
      printf("[%07.1fs] %05u Samples\n", (double) image.meta_data.time, image.meta_data.sample_count);
      obtained++;
      LuminaryPath* image_path;
      LUM_FAILURE_HANDLE(luminary_path_create(&image_path));
      snprintf(file, sizeof(file), "%s/Bench-%05u-%s.png", outdir, image.meta_data.sample_count, name);
      fprintf(times, "%u, %f\n", image.meta_data.sample_count, (double) image.meta_data.time);
      LUM_FAILURE_HANDLE(luminary_path_set_from_string(image_path, file));
      LUM_FAILURE_HANDLE(luminary_host_save_png(host, output, image_path));
      LUM_FAILURE_HANDLE(luminary_path_destroy(&image_path));
      LUM_FAILURE_HANDLE(luminary_host_release_output(host, output));
      promises[id] = LUMINARY_OUTPUT_HANDLE_INVALID;
    }
  }
  fclose(times);

  /* the status window's data (windows/renderer_status.c:15-61) */
  uint32_t num_workers;
  LUM_FAILURE_HANDLE(luminary_host_get_num_queue_workers(host, &num_workers));
  for (uint32_t w = 0; w < num_workers; w++) {
    const char* worker_name;
    const char* string;
    double time;
    LUM_FAILURE_HANDLE(luminary_host_get_queue_worker_name(host, w, &worker_name));
    LUM_FAILURE_HANDLE(luminary_host_get_queue_worker_string(host, w, &string));
    LUM_FAILURE_HANDLE(luminary_host_get_queue_worker_time(host, w, &time));
    printf("queue worker %u: %s | %s | %.2fs\n", w, worker_name ? worker_name : "(null)", string ? string : "(idle)", time);
  }
  LUM_FAILURE_HANDLE(array_destroy(&promises));
  LUM_FAILURE_HANDLE(luminary_host_destroy(&host));
  luminary_shutdown();
  return 0;
}
