// The north star's MFMA clause, measured: "MFMA used only for the dense ray-packet/AABB slab test" (VERDICT round 4, item 7).
//
// A slab test needs, per ray and child box, six plane distances t = plane * inv + noi (inv = 1 / direction, noi = -origin * inv): 24 multiply-adds per visit of
// a 4-wide node, then max3 / min3 and a comparison per child. Only a PACKET - rays that visit the same node - gives the multiply-adds a matrix shape:
//   D[p][i] = A[p][0] * B[0][i] + A[p][1] * B[1][i]   with A = [plane_p, 1] (32 planes x 2), B = [inv_i ; noi_i] (2 x 32 rays)
// is one v_mfma_f32_32x32x2_f32 per axis: 32 planes (the lo / hi planes of the 4 children of FOUR nodes) x 32 rays = 1024 distances in 64 cycles of a SIMD's
// matrix pipe, i.e. 16 useful multiply-adds per cycle against the vector pipe's 32 (f32 MFMA runs at the vector rate and K = 2 spends half of it on the "+ noi"),
// and a wave then holds 32 rays in 64 lanes (the accumulator layout gives lanes i and i + 32 half of ray i's planes each).
// The matrix pipe issues beside the vector pipe (MI355X_MICROARCH.md, "Wave scheduling"), so the question is whether moving the 24 multiply-adds there while the
// vector pipe does the max3 / min3 / compare part makes the staged tree top's visits cheaper. Two kernels on nodes resident in LDS (the tree top, as in
// dev_trace.h), depth-0-like coherent packets, endless loop over the staged nodes:
//   valu   one ray per lane (64 rays per wave), the product's child_entry arithmetic, every lane reads the same node (LDS broadcast)
//   mfma   32 rays per wave; per group of 4 nodes 3 MFMAs form the 96 plane distances per ray, the lanes then do min / max / compare for their half
// Output: ray-box tests per second per CU and chip-wide. Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/microbench/mfma_slab tools/microbench/mfma_slab.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float float16v __attribute__((ext_vector_type(16)));

struct Node { float lo_x[4], lo_y[4], lo_z[4], hi_x[4], hi_y[4], hi_z[4]; uint32_t child[4], pad[4]; };  // the product's 128-byte BVH4 node (dev_scene.h)
static_assert(sizeof(Node) == 128, "one line");

__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

constexpr int kNodes = 256;  // 32 KB of staged nodes

// one ray per lane; the packet walks the staged nodes in order (every lane the same node: the LDS reads broadcast)
__global__ __launch_bounds__(1024) void k_valu(const Node* __restrict__ g_nodes, uint32_t rounds, unsigned long long* out) {
  __shared__ Node nodes[kNodes];
  for (uint32_t i = threadIdx.x; i < kNodes * 8u; i += blockDim.x) reinterpret_cast<float4*>(nodes)[i] = reinterpret_cast<const float4*>(g_nodes)[i];
  __syncthreads();
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  const float ox = 0.5f + 1e-3f * (gid & 63u), oy = 0.5f, oz = -2.0f;
  const float dx = 1e-3f * ((gid & 7u) + 1.0f), dy = 1e-3f * (((gid >> 3) & 7u) + 1.0f), dz = 1.0f;  // a coherent bundle: all direction signs positive
  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz, nx = -ox * ix, ny = -oy * iy, nz = -oz * iz;
  const float tmax = 1e30f;
  uint32_t hits = 0;
  for (uint32_t r = 0; r < rounds; r++) {
#pragma unroll 2
    for (int n = 0; n < kNodes; n++) {
      const float4 lx = *reinterpret_cast<const float4*>(nodes[n].lo_x), ly = *reinterpret_cast<const float4*>(nodes[n].lo_y), lz = *reinterpret_cast<const float4*>(nodes[n].lo_z);
      const float4 hx = *reinterpret_cast<const float4*>(nodes[n].hi_x), hy = *reinterpret_cast<const float4*>(nodes[n].hi_y), hz = *reinterpret_cast<const float4*>(nodes[n].hi_z);
      auto entry = [&](float ax, float ay, float az, float bx, float by, float bz) {
        const float tn = vmax3(__builtin_fmaf(ax, ix, nx), __builtin_fmaf(ay, iy, ny), fmaxf(__builtin_fmaf(az, iz, nz), 0.0f));
        const float tf = vmin3(__builtin_fmaf(bx, ix, nx), __builtin_fmaf(by, iy, ny), fminf(__builtin_fmaf(bz, iz, nz), tmax));
        return tn <= tf ? 1u : 0u;
      };
      hits += entry(lx.x, ly.x, lz.x, hx.x, hy.x, hz.x) + entry(lx.y, ly.y, lz.y, hx.y, hy.y, hz.y) + entry(lx.z, ly.z, lz.z, hx.z, hy.z, hz.z) +
              entry(lx.w, ly.w, lz.w, hx.w, hy.w, hz.w);
    }
  }
  if (hits == 0xFFFFFFFFu) out[1] = hits;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = 0;
  atomicAdd(out + 2, (unsigned long long) hits);
}

// 32 rays per wave (lane i and lane i + 32 belong to ray i). Per group of 4 nodes: the A operand of axis a holds that axis' 32 planes
// (plane p = node (p >> 3), lo / hi (p >> 2) & 1, child p & 3) in lanes 0-31 and 1.0 in lanes 32-63; the B operand inv_a in lanes 0-31 and noi_a in lanes
// 32-63. The accumulator gives lane l (ray l & 31) the rows {0-3, 8-11, 16-19, 24-27} + 4 * (l >> 5): with the plane order above, lanes < 32 hold the lo planes
// of the four nodes' children and lanes >= 32 the hi planes, so the near / far halves meet through one ds_swizzle / DPP row exchange per value... which would cost
// what was saved; instead the planes are ordered so that a lane holds BOTH planes of two children: p = node * 8 + half * 4 + (child & 1) * 2 + lohi with
// half = child >> 1 -> rows 4 h .. 4 h + 3 of node n are (child 2h, lo) (child 2h, hi) (child 2h + 1, lo) (child 2h + 1, hi): lane l < 32 tests children 0, 1
// of every node, lane l + 32 children 2, 3.
__global__ __launch_bounds__(1024) void k_mfma(const Node* __restrict__ g_nodes, uint32_t rounds, unsigned long long* out) {
  __shared__ float planes[3][kNodes / 4][32];  // per axis and node group: the 32 plane values in MFMA row order
  for (uint32_t i = threadIdx.x; i < 3u * (kNodes / 4) * 32u; i += blockDim.x) {
    const uint32_t a = i / ((kNodes / 4) * 32u), rem = i % ((kNodes / 4) * 32u), grp = rem / 32u, p = rem % 32u;
    const uint32_t node = grp * 4u + (p >> 3), half = (p >> 2) & 1u, child = half * 2u + ((p >> 1) & 1u), hi = p & 1u;
    const Node& nd = g_nodes[node];
    const float* src = a == 0 ? (hi ? nd.hi_x : nd.lo_x) : a == 1 ? (hi ? nd.hi_y : nd.lo_y) : (hi ? nd.hi_z : nd.lo_z);
    planes[a][grp][p] = src[child];
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63u, ray = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32u + (lane & 31u);
  const float ox = 0.5f + 1e-3f * (ray & 63u), oy = 0.5f, oz = -2.0f;
  const float dx = 1e-3f * ((ray & 7u) + 1.0f), dy = 1e-3f * (((ray >> 3) & 7u) + 1.0f), dz = 1.0f;
  const float ix = 1.0f / dx, iy = 1.0f / dy, iz = 1.0f / dz;
  const bool upper = lane >= 32u;
  const float bx = upper ? -ox * ix : ix, by = upper ? -oy * iy : iy, bz = upper ? -oz * iz : iz;  // B: row 0 = inv, row 1 = noi
  const float tmax = 1e30f;
  uint32_t hits = 0;
  const float16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t r = 0; r < rounds; r++) {
#pragma unroll 2
    for (int grp = 0; grp < kNodes / 4; grp++) {
      const float ax = upper ? 1.0f : planes[0][grp][lane & 31u], ay = upper ? 1.0f : planes[1][grp][lane & 31u], az = upper ? 1.0f : planes[2][grp][lane & 31u];
      const float16v tx = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bx, zero, 0, 0, 0);
      const float16v ty = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, by, zero, 0, 0, 0);
      const float16v tz = __builtin_amdgcn_mfma_f32_32x32x2f32(az, bz, zero, 0, 0, 0);
      // registers 4 n .. 4 n + 3 = node n: (child a, lo) (child a, hi) (child b, lo) (child b, hi) - all direction signs positive: lo = near
#pragma unroll
      for (int n = 0; n < 4; n++) {
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const float tn = vmax3(tx[4 * n + 2 * c], ty[4 * n + 2 * c], fmaxf(tz[4 * n + 2 * c], 0.0f));
          const float tf = vmin3(tx[4 * n + 2 * c + 1], ty[4 * n + 2 * c + 1], fminf(tz[4 * n + 2 * c + 1], tmax));
          hits += tn <= tf ? 1u : 0u;
        }
      }
    }
  }
  if (hits == 0xFFFFFFFFu) out[1] = hits;
  atomicAdd(out + 3, (unsigned long long) hits);
}

int main(int argc, char** argv) {
  uint32_t rounds = 200;
  for (int i = 1; i < argc; i++) if (!strcmp(argv[i], "--rounds") && i + 1 < argc) rounds = (uint32_t) atoi(argv[++i]);
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  std::vector<Node> nodes(kNodes);
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> U(0.0f, 1.0f);
  for (auto& nd : nodes)
    for (int k = 0; k < 4; k++) {
      const float c[3] = {U(rng), U(rng), U(rng)}, e = 0.1f + 0.3f * U(rng);
      nd.lo_x[k] = c[0] - e; nd.hi_x[k] = c[0] + e; nd.lo_y[k] = c[1] - e; nd.hi_y[k] = c[1] + e; nd.lo_z[k] = c[2] - e; nd.hi_z[k] = c[2] + e;
      nd.child[k] = 0; nd.pad[k] = 0;
    }
  Node* d_nodes;
  unsigned long long* out;
  hipMalloc(&d_nodes, sizeof(Node) * kNodes);
  hipMalloc(&out, 64);
  hipMemcpy(d_nodes, nodes.data(), sizeof(Node) * kNodes, hipMemcpyHostToDevice);
  printf("# %s, %d CUs, one workgroup of 1024 threads per CU, %d staged nodes, %u rounds\n", prop.name, cus, kNodes, rounds);
  for (int kind = 0; kind < 2; kind++) {
    for (int rep = 0; rep < 3; rep++) {
      hipMemset(out, 0, 64);
      hipEvent_t a, b;
      hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      if (kind == 0) hipLaunchKernelGGL(k_valu, dim3(cus), dim3(1024), 0, 0, d_nodes, rounds, out);
      else hipLaunchKernelGGL(k_mfma, dim3(cus), dim3(1024), 0, 0, d_nodes, rounds, out);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms = 0;
      hipEventElapsedTime(&ms, a, b);
      unsigned long long h[4] = {0, 0, 0, 0};
      hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
      // ray-box tests: rays x nodes x 4 children x rounds; the vector kernel holds 1024 rays per workgroup, the matrix kernel 512
      const double rays = (double) cus * (kind == 0 ? 1024.0 : 512.0);
      const double tests = rays * kNodes * 4.0 * rounds;
      if (rep == 2)
        printf("{\"kernel\": \"%s\", \"rays_per_workgroup\": %d, \"ms\": %.3f, \"ray_box_tests_per_s\": %.4g, \"per_cu\": %.4g, \"hits\": %llu}\n",
               kind == 0 ? "valu (one ray per lane)" : "mfma + valu (32-ray packets, 3 v_mfma_f32_32x32x2_f32 per 4 nodes)", kind == 0 ? 1024 : 512, ms, tests / (ms * 1e-3),
               tests / (ms * 1e-3) / cus, kind == 0 ? h[2] : h[3]);
      hipEventDestroy(a); hipEventDestroy(b);
    }
  }
  return 0;
}
