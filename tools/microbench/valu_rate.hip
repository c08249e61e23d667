// Issue rate of the vector ALU for the instruction kinds k_shade is made of: how many wave64 instructions per second the chip retires when nothing
// but the VALU is in the way. bench.py prices k_shade against this figure ("roofline_shade").
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/microbench/valu_rate.hip && ./valu_rate
// Every kernel runs a loop of 64 independent instructions of one kind on 16 registers (no dependency closer than 16 instructions), `waves` waves per
// SIMD on every CU. Output: G wave-instructions per second chip-wide and cycles per wave-instruction and SIMD at the clock the run reached.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

// The clock the run reached, measured inside the kernel: the shader-clock counter (s_memtime) over the constant 100 MHz counter (s_memrealtime), both
// read by one wave at the start and at the end of its loop. (Round 3 divided by an assumed 2.4 GHz.)
__device__ unsigned long long g_clock_samples[4];
__device__ __forceinline__ unsigned long long real_time() { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, float seed) {
  const bool sampler = blockIdx.x == 0 && threadIdx.x == 0;
  unsigned long long c0 = 0, r0 = 0;
  if (sampler) { c0 = __builtin_readcyclecounter(); r0 = real_time(); }
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  float b0 = seed * 2, b1 = seed * 3, b2 = seed * 4, b3 = seed * 5, b4 = seed * 6, b5 = seed * 7, b6 = seed * 8, b7 = seed * 9;
  const float m = 0.999f, c = 0.001f;
  for (int i = 0; i < (KIND == 11 ? 0 : iters); i++) {
    if (KIND == 0) {  // v_fma_f32, 16 independent chains
      REP4(asm volatile("v_fma_f32 %0, %0, %16, %17\n v_fma_f32 %1, %1, %16, %17\n v_fma_f32 %2, %2, %16, %17\n v_fma_f32 %3, %3, %16, %17\n"
                        "v_fma_f32 %4, %4, %16, %17\n v_fma_f32 %5, %5, %16, %17\n v_fma_f32 %6, %6, %16, %17\n v_fma_f32 %7, %7, %16, %17\n"
                        "v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                        "v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5),
                          "+v"(b6), "+v"(b7)
                        : "v"(m), "v"(c));)
    }
    else if (KIND == 1) {  // v_pk_fma_f32 on 8 register pairs: 64 wave-instructions = 128 fused multiply-adds per lane
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {b0, b1}, p5 = {b2, b3}, p6 = {b4, b5}, p7 = {b6, b7};
      const f2 mm = {m, m}, cc = {c, c};
      REP4(REP4(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                             : "v"(mm), "v"(cc));))
      a0 = p0.x + p4.x; a1 = p0.y + p4.y; a2 = p1.x + p5.x; a3 = p1.y + p5.y; a4 = p2.x + p6.x; a5 = p2.y + p6.y; a6 = p3.x + p7.x; a7 = p3.y + p7.y;
    }
    else if (KIND == 2) {  // v_cmp_lt_f32 + v_cndmask_b32 pairs (the reservoir's accept / select)
      REP4(REP4(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m), "v"(c)
                             : "vcc");))
    }
    else if (KIND == 3) {  // v_rcp_f32 (quarter-rate transcendental unit?)
      REP4(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                        "v_rcp_f32 %8, %8\n v_rcp_f32 %9, %9\n v_rcp_f32 %10, %10\n v_rcp_f32 %11, %11\n v_rcp_f32 %12, %12\n v_rcp_f32 %13, %13\n v_rcp_f32 %14, %14\n v_rcp_f32 %15, %15\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5),
                          "+v"(b6), "+v"(b7));)
    }
    else if (KIND == 4) {  // v_mul_f32 + v_med3_f32 (rescale and clamp)
      REP4(REP4(asm volatile("v_mul_f32 %0, %0, %8\n v_med3_f32 %0, %0, 0, %9\n v_mul_f32 %1, %1, %8\n v_med3_f32 %1, %1, 0, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m), "v"(c));))
    }
    else if (KIND == 5) {  // v_pk_mul_f32
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {b0, b1}, p5 = {b2, b3}, p6 = {b4, b5}, p7 = {b6, b7};
      const f2 mm = {m, m};
      REP4(REP4(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7)
                             : "v"(mm));))
      a0 = p0.x + p4.x; a1 = p0.y + p4.y; a2 = p1.x + p5.x; a3 = p1.y + p5.y; a4 = p2.x + p6.x; a5 = p2.y + p6.y; a6 = p3.x + p7.x; a7 = p3.y + p7.y;
    }
    else if (KIND == 6) {  // v_mul_f32 alone (VOP2, two vector operands)
      REP4(REP4(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m), "v"(c));))
    }
    else if (KIND == 7) {  // v_fma_f32 with a scalar operand (two vector operands)
      REP4(REP4(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "s"(m), "v"(c));))
    }
    else if (KIND == 8) {  // v_fmac_f32 (VOP2: d = a * b + d)
      REP4(REP4(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m), "v"(c));))
    }
    else if (KIND == 9) {  // v_med3_f32 alone (VOP3, one vector operand, two constants)
      REP4(REP4(asm volatile("v_med3_f32 %0, %0, 0, 1.0\n v_med3_f32 %1, %1, 0, 1.0\n v_med3_f32 %2, %2, 0, 1.0\n v_med3_f32 %3, %3, 0, 1.0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m), "v"(c));))
    }
    else if (KIND == 10) {  // v_sqrt_f32
      REP4(REP4(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(m), "v"(c));))
    }
  }
  if (KIND == 11) {  // v_mul_lo_u32: the sampler's hash rounds (dev_sampler.h laine_karras) are made of these
    uint32_t u0 = __float_as_uint(a0), u1 = __float_as_uint(a1), u2 = __float_as_uint(a2), u3 = __float_as_uint(a3), u4 = __float_as_uint(a4), u5 = __float_as_uint(a5), u6 = __float_as_uint(a6), u7 = __float_as_uint(a7);
    const uint32_t k = 0x6c50b47cu;
    for (int i = 0; i < iters; i++) {
      REP4(REP4(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7)
                             : "v"(k));))
    }
    a0 = __uint_as_float(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
  }
  if (sampler) { g_clock_samples[0] = __builtin_readcyclecounter() - c0; g_clock_samples[1] = real_time() - r0; }
  const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7;
  if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, int waves_per_simd, float* out, int cus) {
  const int iters = 20000;
  const dim3 grid(cus * waves_per_simd), blk(256);  // 256 threads = one wave per SIMD; `waves_per_simd` workgroups per CU
  hipLaunchKernelGGL((k_rate<KIND>), grid, blk, 0, 0, out, 100, 1.0f);
  hipEvent_t a, b;
  (void) hipEventCreate(&a); (void) hipEventCreate(&b);
  (void) hipEventRecord(a);
  hipLaunchKernelGGL((k_rate<KIND>), grid, blk, 0, 0, out, iters, 1.0f);
  (void) hipEventRecord(b);
  (void) hipEventSynchronize(b);
  float ms = 0;
  (void) hipEventElapsedTime(&ms, a, b);
  const double insts = (double) grid.x * 4.0 * iters * 64.0;  // wave-instructions
  const double per_simd_per_s = insts / (cus * 4.0) / (ms * 1e-3);
  unsigned long long samples[4] = {0, 0, 0, 0};
  (void) hipMemcpyFromSymbol(samples, HIP_SYMBOL(g_clock_samples), sizeof(samples));
  // s_memtime counts shader-clock cycles where the part exposes them; where it runs at the constant 100 MHz as well the ratio is 1 and says nothing
  const double ratio = samples[1] ? (double) samples[0] / (double) samples[1] : 0.0;
  const double clock_hz = ratio > 1.5 ? ratio * 100e6 : 0.0;
  printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"g_wave_insts_per_s\": %.1f, \"cycles_per_wave_inst_per_simd_at_2.4GHz\": %.2f, \"measured_clock_ghz\": %.3f, "
         "\"cycles_per_wave_inst_per_simd_at_measured_clock\": %.2f, \"memtime_ticks\": %llu, \"memrealtime_ticks\": %llu}\n", name, waves_per_simd, ms, insts / (ms * 1e-3) / 1e9,
         2.4e9 / per_simd_per_s, clock_hz / 1e9, clock_hz > 0.0 ? clock_hz / per_simd_per_s : 0.0, samples[0], samples[1]);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t prop;
  (void) hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  float* out;
  (void) hipMalloc(&out, (size_t) cus * 8 * 256 * 4);
  printf("# %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
  for (int w : {2, 3, 4}) {
    run<0>("v_fma_f32", w, out, cus);
    run<1>("v_pk_fma_f32", w, out, cus);
    run<2>("v_cmp_lt_f32+v_cndmask_b32", w, out, cus);
    run<3>("v_rcp_f32", w, out, cus);
    run<4>("v_mul_f32+v_med3_f32", w, out, cus);
    run<5>("v_pk_mul_f32", w, out, cus);
    run<6>("v_mul_f32", w, out, cus);
    run<7>("v_fma_f32 (one scalar operand)", w, out, cus);
    run<8>("v_fmac_f32", w, out, cus);
    run<9>("v_med3_f32 (one vector operand)", w, out, cus);
    run<10>("v_sqrt_f32", w, out, cus);
    run<11>("v_mul_lo_u32", w, out, cus);
  }
  (void) hipDeviceSynchronize();
  return 0;
}
