// Microbenchmark: cost of divergent 16-byte gathers on gfx950, to size the BVH node format.
// Every lane walks a pseudo-random chain of 128-byte lines and reads n x 16 bytes of each line.
//   hipcc --offload-arch=gfx950 -O3 -o gather tools/microbench/gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int N, int DEP, int MODE>
__global__ __launch_bounds__(256) void k_gather(const float4* __restrict__ buf, uint32_t lines_mask, int iters, float* out) {
  // MODE 0: every lane its own line; 1: 4 adjacent lanes share a line; 2: 16 lanes share; 3: only every 4th lane active
  uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  if (MODE == 1) tid >>= 2;
  if (MODE == 2) tid >>= 4;
  if (MODE == 3 && (threadIdx.x & 3)) return;
  uint32_t idx = tid * 2654435761u;
  float acc = 0.0f;
  for (int i = 0; i < iters; i++) {
    idx = idx * 1664525u + 1013904223u;
    const uint32_t line = (idx >> 8) & lines_mask;
    const float4* p = buf + (size_t) line * 8;
    if (MODE == 1 || MODE == 2) p += 0;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < N; k++) { const float4 v = p[k]; s += v.x + v.y + v.z + v.w; }
    acc += s;
    if (DEP) idx += (uint32_t) (s != 12345.0f ? 0 : 1);  // make the next address depend on the loaded data
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int N, int DEP, int MODE = 0>
void run(const float4* buf, uint32_t lines, float* out, int blocks, int iters, const char* label) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k_gather<N, DEP, MODE>), dim3(blocks), dim3(256), 0, 0, buf, lines - 1, iters, out);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_gather<N, DEP, MODE>), dim3(blocks), dim3(256), 0, 0, buf, lines - 1, iters, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double visits = (double) blocks * 256 * iters;
  printf("%-10s mode=%d N=%d dep=%d lines=%8u (%6.1f MB): %7.3f ms  %7.2f G line-visits/s  %8.1f GB/s useful  %6.1f clk/wave-visit/CU\n", label, MODE, N, DEP, lines, lines * 128.0 / 1e6, ms,
         visits / ms / 1e6, visits * N * 16 / ms / 1e6, ms * 1e-3 * 2.4e9 / (visits / 64 / 256));
}

int main() {
  const size_t max_lines = 1u << 22;  // 512 MB
  float4* buf; float* out;
  hipMalloc(&buf, max_lines * 128);
  hipMemset(buf, 0, max_lines * 128);
  const int blocks = 256 * 8;
  hipMalloc(&out, (size_t) blocks * 256 * 4);
  for (uint32_t lines : {1u << 15}) {
    run<7, 1, 1>(buf, lines, out, blocks, 256, "share4");
    run<7, 1, 2>(buf, lines, out, blocks, 256, "share16");
    run<7, 1, 3>(buf, lines, out, blocks, 256, "quarter");
    run<1, 1, 1>(buf, lines, out, blocks, 256, "share4");
    run<1, 1, 3>(buf, lines, out, blocks, 256, "quarter");
  }
  for (uint32_t lines : {1u << 15, 1u << 18, 1u << 21}) {
    run<1, 1>(buf, lines, out, blocks, 256, "gather");
    run<2, 1>(buf, lines, out, blocks, 256, "gather");
    run<4, 1>(buf, lines, out, blocks, 256, "gather");
    run<7, 1>(buf, lines, out, blocks, 256, "gather");
    run<8, 1>(buf, lines, out, blocks, 256, "gather");
    run<4, 0>(buf, lines, out, blocks, 256, "gather");
    run<7, 0>(buf, lines, out, blocks, 256, "gather");
  }
  return 0;
}
