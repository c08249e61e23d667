// Calibration of rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ / TCP_TCC_READ_REQ for the access pattern of the ray kernels: every lane reads
// n x 16 bytes of its own pseudo-random 128-byte line (a BVH4 node visit reads 7 x 16 B of one line), next to the coalesced streaming read
// MI355X_MICROARCH.md's factor 2 was measured on. Every line of the table is read exactly once per launch (line = odd multiplier x id mod
// 2^k is a bijection), so the bytes that have to come from memory are known: lines x 128 B if a miss fetches the whole line, lines x 64 B
// x (halves touched) if it fetches 64-byte halves.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/microbench/fetch_calib.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./fetch_calib
// Kernel names carry the mode: k_calib<N16, HOT> with N16 = 16-byte loads per line (0 = coalesced stream); HOT = 1 reads a 1 MiB table
// that stays in every XCD's L2 instead (calibrates the L2-side request counters; nothing should reach memory).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int N16, int HOT>
__global__ __launch_bounds__(256) void k_calib(const float4* __restrict__ buf, uint32_t lines_mask, float* out) {
  const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
  float s = 0.0f;
  if (N16 == 0) {  // coalesced: 16 B per lane, consecutive lanes consecutive addresses, 8 loads per lane one table-eighth apart
    const size_t total16 = ((size_t) lines_mask + 1) * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) { const float4 v = buf[(size_t) tid + (size_t) k * (total16 / 8)]; s += v.x + v.y + v.z + v.w; }
  }
  else {
    const uint32_t line = (tid * 2654435761u) & lines_mask;  // odd multiplier: a permutation of the lines
    const float4* p = buf + (size_t) line * 8;
#pragma unroll
    for (int k = 0; k < N16; k++) { const float4 v = p[k]; s += v.x + v.y + v.z + v.w; }
  }
  if (s == 12345.678f) out[tid] = s;  // keeps the loads alive, writes nothing
}

template <int N16, int HOT>
void run(const float4* buf, uint32_t lines, float* out) {
  // one thread per line (gather) or per 8 x 16 B (stream: lines threads x 8 loads x 16 B = lines x 128 B)
  const uint32_t threads = lines;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int reps = HOT ? 64 : 1;
  hipEventRecord(a);
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((k_calib<N16, HOT>), dim3(HOT ? 65536 : threads / 256), dim3(256), 0, 0, buf, lines - 1, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double lane_loads = (double) (HOT ? 65536u * 256u : threads) * (N16 ? N16 : 8) * reps;
  printf("k_calib<%d,%d> lines=%u table=%.1f MiB launches=%d threads/launch=%u  16B-lane-loads=%.0f  useful bytes=%.0f  line bytes=%.0f  %.3f ms  %.1f G lane-loads/s\n", N16, HOT, lines,
         lines * 128.0 / 1048576.0, reps, HOT ? 65536u * 256u : threads, lane_loads, lane_loads * 16, HOT ? 0.0 : lines * 128.0, ms, lane_loads / ms / 1e6);
}

int main() {
  const uint32_t cold_lines = 1u << 25;  // 4 GiB: far beyond the 256 MiB Infinity Cache, every line touched once per launch
  const uint32_t hot_lines = 1u << 13;   // 1 MiB: resident in every XCD's L2
  float4* buf; float* out;
  if (hipMalloc(&buf, (size_t) cold_lines * 128) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  hipMemset(buf, 0, (size_t) cold_lines * 128);
  hipMalloc(&out, (size_t) cold_lines * 4);
  hipDeviceSynchronize();
  run<0, 0>(buf, cold_lines, out);
  run<1, 0>(buf, cold_lines, out);
  run<2, 0>(buf, cold_lines, out);
  run<4, 0>(buf, cold_lines, out);
  run<7, 0>(buf, cold_lines, out);
  run<8, 0>(buf, cold_lines, out);
  run<1, 1>(buf, hot_lines, out);
  run<7, 1>(buf, hot_lines, out);
  hipDeviceSynchronize();
  return 0;
}
