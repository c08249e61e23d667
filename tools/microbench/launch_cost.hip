// What an empty launch of the persistent ray kernels' shape costs: 256 workgroups x 1024 threads, with / without 152 KB of dynamic LDS, with / without
// a 1 KB private array per lane (scratch), with / without a 128-register allocation. The kernels read one word and return.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_cost tools/microbench/launch_cost.hip && /tmp/launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int kScratchWords, int kLiveRegs>
__global__ __launch_bounds__(1024) void k_empty(const unsigned* flag, unsigned* out) {
  extern __shared__ float4 lds[];
  if (*flag == 0u) return;  // always taken; what follows only sizes the kernel
  unsigned priv[kScratchWords > 0 ? kScratchWords : 1];
  for (int i = 0; i < (kScratchWords > 0 ? kScratchWords : 1); i++) priv[i] = flag[i + threadIdx.x];
  unsigned acc = 0;
  float live[kLiveRegs];
  for (int i = 0; i < kLiveRegs; i++) live[i] = (float) flag[i * 7 + threadIdx.x];
  for (int k = 0; k < 64; k++) for (int i = 0; i < kLiveRegs; i++) live[i] = live[i] * live[(i + 1) % kLiveRegs] + 1.0f;
  for (int i = 0; i < kLiveRegs; i++) acc += (unsigned) live[i];
  for (int i = 0; i < (kScratchWords > 0 ? kScratchWords : 1); i++) acc += priv[(i * 17 + flag[3]) % (kScratchWords > 0 ? kScratchWords : 1)];
  lds[threadIdx.x] = make_float4((float) acc, 0.f, 0.f, 0.f);
  __syncthreads();
  out[blockIdx.x * 1024 + threadIdx.x] = acc + (unsigned) lds[(threadIdx.x + 1) & 1023].x;
}

template <class K>
static int timeit(const char* name, K kernel, size_t lds, int grid, int block, const unsigned* flag, unsigned* out) {
  CHECK(hipFuncSetAttribute((const void*) kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  for (int i = 0; i < 20; i++) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, 0, flag, out);
  CHECK(hipDeviceSynchronize());
  const int reps = 500;
  CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; i++) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), lds, 0, flag, out);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  std::printf("%-58s grid %4d x %4d  lds %6zu B  %7.2f us per launch (back to back)\n", name, grid, block, lds, ms * 1000.0f / reps);
  return 0;
}

int main() {
  unsigned *flag, *out;
  CHECK(hipMalloc(&flag, 1 << 20)); CHECK(hipMemset(flag, 0, 1 << 20));
  CHECK(hipMalloc(&out, 256 * 1024 * 4));
  int rc = 0;
  rc |= timeit("no scratch, few registers", k_empty<0, 4>, 0, 256, 1024, flag, out);
  rc |= timeit("no scratch, few registers", k_empty<0, 4>, 152 * 1024, 256, 1024, flag, out);
  rc |= timeit("1 KB scratch per lane", k_empty<256, 4>, 0, 256, 1024, flag, out);
  rc |= timeit("1 KB scratch per lane", k_empty<256, 4>, 152 * 1024, 256, 1024, flag, out);
  rc |= timeit("64 B scratch per lane", k_empty<16, 4>, 152 * 1024, 256, 1024, flag, out);
  rc |= timeit("no scratch, ~100 live registers", k_empty<0, 100>, 152 * 1024, 256, 1024, flag, out);
  rc |= timeit("1 KB scratch, ~100 live registers", k_empty<256, 100>, 152 * 1024, 256, 1024, flag, out);
  rc |= timeit("1 KB scratch, ~100 live registers, 256 threads x 1024", k_empty<256, 100>, 0, 1024, 256, flag, out);
  rc |= timeit("no scratch, few registers, 256 threads x 768", k_empty<0, 4>, 20 * 1024, 768, 256, flag, out);
  return rc;
}
