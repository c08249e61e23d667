// What a chain of dependent divergent line gathers can reach on this part: the honest ceiling of a BVH traversal's node fetches.
// Every lane (GROUP = 1) or every group of 4 adjacent lanes (GROUP = 4) owns one "ray": per step it reads LOADS x 16 bytes of one
// pseudo-random 128-byte line of a table and - DEP = 1 - derives the next line from what it read, as a traversal derives the next node from the
// node it has just tested (DEP = 0: the next line is a function of the lane's own counter, loads stay in flight together).
//   GROUP = 1, LOADS = 7   a Bvh4Node visit of dev_trace.h: one lane reads 7 of the line's 8 sixteen-byte words
//   GROUP = 1, LOADS = 4   a 64-byte quantised node
//   GROUP = 4, LOADS = 2   four lanes share a ray: lane j reads words j and j + 4 of the line (the wave's 16 rays touch 16 lines per load
//                          instruction, each with 64 contiguous bytes)
// ACTIVE: lanes (or groups) per wave that hold a ray at all (the ray kernels run at a lane utilisation of about 0.5).
// Launch shape: 256 CUs x 1 workgroup of `block` threads (1024 = 16 waves per CU, as the fast flavour's ray kernels) unless --blocks-per-cu.
// Output per configuration: line visits per second, useful bytes (LOADS x 16 x visits) per second, and line bytes (128 x visits) per second - the
// last one is what the memory side moves when every visit misses the caches, i.e. what rocprofv3's FETCH_SIZE x 2 would tally.
//   hipcc --offload-arch=gfx950 -O3 -o gather tools/microbench/gather.hip && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int LOADS, int GROUP, int DEP>
__global__ __launch_bounds__(1024) void k_gather(const uint4* __restrict__ table, uint32_t lines_mask, uint32_t steps, uint32_t active, uint32_t* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t gid = (blockIdx.x * blockDim.x + threadIdx.x) / GROUP;
  const uint32_t sub = lane % GROUP;
  uint32_t cur = mix(gid * 2654435761u + 12345u) & lines_mask;
  uint32_t acc = 0, counter = gid;
  if (lane / GROUP >= active) { if (out && cur == 0xFFFFFFFFu) out[gid] = 1; return; }
  for (uint32_t s = 0; s < steps; s++) {
    const uint4* p = table + (size_t) cur * 8u;
    uint32_t h = 0;
    if (GROUP == 1) {
#pragma unroll
      for (int k = 0; k < LOADS; k++) { const uint4 v = p[k]; h ^= v.x + v.y + v.z + v.w; }
    }
    else {
#pragma unroll
      for (int k = 0; k < LOADS; k++) { const uint4 v = p[sub + GROUP * k]; h ^= v.x + v.y + v.z + v.w; }
      // the group's lanes agree on the next line: xor over the 4 lanes (DPP row shuffles, as a cooperative traversal would combine its child tests)
      h ^= __shfl_xor(h, 1);
      h ^= __shfl_xor(h, 2);
    }
    acc += h;
    counter += 0x9E3779B9u;
    cur = (DEP ? mix(h ^ counter) : mix(counter)) & lines_mask;
  }
  if (out && acc == 0x12345678u) out[gid] = acc;
}

struct Cfg { int loads, group, dep; };

template <int LOADS, int GROUP, int DEP>
static double run(const uint4* table, uint32_t lines, uint32_t steps, uint32_t active, int block, int blocks_per_cu, uint32_t* out, int cus) {
  const dim3 grid(cus * blocks_per_cu), blk(block);
  hipLaunchKernelGGL((k_gather<LOADS, GROUP, DEP>), grid, blk, 0, 0, table, lines - 1, steps / 8 + 1, active, out);  // warm-up
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a);
  hipLaunchKernelGGL((k_gather<LOADS, GROUP, DEP>), grid, blk, 0, 0, table, lines - 1, steps, active, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  const double rays = (double) grid.x * (block / 64) * active;
  const double visits = rays * steps;
  const double sec = ms * 1e-3;
  printf("{\"loads\": %d, \"group\": %d, \"dep\": %d, \"table_mib\": %.0f, \"waves_per_cu\": %d, \"active_per_wave\": %u, \"rays_in_flight\": %.0f, \"ms\": %.3f, "
         "\"gvisits_per_s\": %.2f, \"useful_tb_s\": %.3f, \"line_tb_s\": %.3f, \"ns_per_step\": %.0f}\n",
         LOADS, GROUP, DEP, lines * 128.0 / 1048576.0, block / 64 * blocks_per_cu, active, rays, ms, visits / sec / 1e9, visits * LOADS * 16.0 * (GROUP == 1 ? 1 : GROUP) / sec / 1e12,
         visits * 128.0 / sec / 1e12, sec * 1e9 / steps);
  fflush(stdout);
  return visits / sec;
}

int main(int argc, char** argv) {
  int block = 1024, blocks_per_cu = 1;
  uint32_t steps = 2000;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--block") && i + 1 < argc) block = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--blocks-per-cu") && i + 1 < argc) blocks_per_cu = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--steps") && i + 1 < argc) steps = (uint32_t) atoi(argv[++i]);
  }
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const uint32_t max_lines = 1u << 23;  // 1 GiB
  uint4* table; uint32_t* out;
  if (hipMalloc(&table, (size_t) max_lines * 128) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  {  // random content (the next line depends on it)
    std::vector<uint32_t> h((size_t) 1 << 22);
    uint32_t x = 0x12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    for (size_t off = 0; off < (size_t) max_lines * 128; off += h.size() * 4) hipMemcpy((char*) table + off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  }
  hipMalloc(&out, (size_t) cus * 4 * 1024 * 4);
  hipDeviceSynchronize();
  printf("# %s, %d CUs; block %d x %d per CU; %u steps per ray\n", prop.name, cus, block, blocks_per_cu, steps);
  const uint32_t sizes[] = {1u << 14 /* 2 MiB: every XCD's L2 */, 1u << 20 /* 128 MiB: Infinity Cache, the hall's working set */, 1u << 23 /* 1 GiB: HBM, the scan's */};
  for (uint32_t lines : sizes) {
    for (uint32_t active : {64u, 32u}) {
      run<7, 1, 1>(table, lines, steps, active, block, blocks_per_cu, out, cus);
      run<4, 1, 1>(table, lines, steps, active, block, blocks_per_cu, out, cus);
      run<2, 1, 1>(table, lines, steps, active, block, blocks_per_cu, out, cus);
      run<1, 1, 1>(table, lines, steps, active, block, blocks_per_cu, out, cus);
      run<2, 4, 1>(table, lines, steps, active / 4, block, blocks_per_cu, out, cus);
      run<1, 4, 1>(table, lines, steps, active / 4, block, blocks_per_cu, out, cus);
    }
    run<7, 1, 0>(table, lines, steps, 64, block, blocks_per_cu, out, cus);
    run<2, 4, 0>(table, lines, steps, 16, block, blocks_per_cu, out, cus);
  }
  hipDeviceSynchronize();
  return 0;
}
