// What limits the L2 -> LDS fill of ONE workgroup per CU? In-kernel stamps of the GEMM tiles (profiles/r03f, r03k) read 27-33 B/clk per CU for a lone
// workgroup (4 or 8 waves, any ring depth) against 45-50 B/clk for two co-resident 4-wave workgroups. This probe streams 128-byte rows of an
// L2-resident matrix into an LDS ring by 16-byte LDS-DMA (1 KiB per wave-instruction, the GEMM's staging) and nothing else, and varies: waves per
// workgroup, workgroups per CU (forced through the LDS request), pieces per wave and step, ring depth (steps in flight across the wait) and whether
// the waves meet at a barrier every step.   Build: hipcc --offload-arch=gfx950 -O3 tools/micro/fill_rate2.hip -o tools/micro/fill_rate2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define GLDS16(gptr, ldsptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NW waves, PIECES 1-KiB pieces per wave and step, DEPTH steps kept in flight across the wait (ring of DEPTH + 1 slots), BAR: barrier per step
template <int NW, int PIECES, int DEPTH, int BAR>
__global__ __launch_bounds__(NW * 64) void fill_kernel(const char* src, long ld, int rows, int steps, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * 131) % (rows - PIECES * NW * 8);
  const char* p[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) p[i] = src + (long)(r0 + (wave * PIECES + i) * 8 + lane / 8) * ld + (lane % 8) * 16;
  constexpr int SLOT = PIECES * NW * 1024;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int s = 0; s < steps; ++s) {
    if (s % 20 == 0 && s) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) p[i] -= 20 * 128;
    }
    char* dst = smem + (s % (DEPTH + 1)) * SLOT + wave * PIECES * 1024;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) { GLDS16(p[i], dst + i * 1024); p[i] += 128; }
    wait_vm<DEPTH * PIECES>();             // all but the DEPTH youngest steps of this wave have landed
    if (BAR) asm volatile("s_barrier" ::: "memory");
  }
  wait_vm<0>();
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime() - c0;
  }
}

template <int NW, int PIECES, int DEPTH, int BAR>
static void run(const char* name, const char* src, long ld, int rows, int per_cu, unsigned long long* stamps) {
  const int steps = 20 * 16;
  const int ring = (DEPTH + 1) * PIECES * NW * 1024;
  const int lds = per_cu == 1 ? (ring > 96 * 1024 ? ring : 96 * 1024) : ring;        // one per CU: ask for more than half the LDS
  if (ring > 160 * 1024 || (per_cu == 2 && ring > 80 * 1024)) { printf("%-44s (ring %d KiB does not fit)\n", name, ring / 1024); return; }
  hipFuncSetAttribute((const void*)fill_kernel<NW, PIECES, DEPTH, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int blocks = 256 * per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<NW, PIECES, DEPTH, BAR>), dim3(blocks), dim3(NW * 64), lds, 0, src, ld, rows, steps, stamps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  static unsigned long long h[2 * 512];
  hipMemcpy(h, stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double rt = 0, ck = 0;
  for (int i = 0; i < blocks; ++i) { rt += (double)h[2 * i]; ck += (double)h[2 * i + 1]; }
  rt /= blocks; ck /= blocks;
  const double bytes_wg = (double)steps * PIECES * NW * 1024;
  printf("%-44s %d WG/CU x %d waves, %2d KiB/step/WG, %d steps in flight, barrier %d: %6.1f us; in-kernel %5.2f GHz, %6.1f cycles/step, %5.1f B/clk/CU, %5.1f GB/s/CU\n", name, per_cu, NW,
         PIECES * NW, DEPTH, BAR, ms * 1e3, ck / rt * 0.1, ck / steps, per_cu * bytes_wg / ck, per_cu * bytes_wg / (rt * 10.0));
}

// Mixed staging: PD pieces per wave and step by LDS-DMA, PR pieces through registers (global_load_dwordx4 this step, ds_write_b128 next step, behind
// the counted wait that also covers the DMA of the step before): does the second path add LDS write bandwidth to the DMA's ~64 B/clk?
template <int NW, int PD, int PR, int BAR>
__global__ __launch_bounds__(NW * 64) void mix_kernel(const char* __restrict__ src, long ld, int rows, int steps, unsigned long long* stamps) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int P = PD + PR;
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * 131) % (rows - P * NW * 8);
  long off[P];
#pragma unroll
  for (int i = 0; i < P; ++i) off[i] = (long)(r0 + (wave * P + i) * 8 + lane / 8) * ld + (lane % 8) * 16;
  constexpr int SLOT = P * NW * 1024;
  const __attribute__((address_space(1))) char* g = (const __attribute__((address_space(1))) char*)src;
  u4 v[PR > 0 ? PR : 1];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  int kk = 0;                                     // column block (128 B) within the K = 1280 row block
  auto issue = [&](int s) {
    char* dst = smem + (s & 1) * SLOT + wave * P * 1024;
#pragma unroll
    for (int i = 0; i < PD; ++i) GLDS16(g + off[i] + kk * 128, dst + i * 1024);
#pragma unroll
    for (int i = 0; i < PR; ++i) v[i] = *(const __attribute__((address_space(1))) u4*)(g + off[PD + i] + kk * 128);
    kk = kk == 19 ? 0 : kk + 1;
  };
  issue(0);
  for (int s = 1; s < steps; ++s) {
    if (PR > 0) {
      // the register pieces of step s-1 have landed: store them into that step's slot, then issue step s
      wait_vm<0>();
      char* prev = smem + ((s - 1) & 1) * SLOT + wave * P * 1024;
#pragma unroll
      for (int i = 0; i < PR; ++i) *(u4*)(prev + (PD + i) * 1024 + lane * 16) = v[i];
    } else {
      wait_vm<0>();
    }
    if (BAR) asm volatile("s_barrier" ::: "memory");
    issue(s);
  }
  wait_vm<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
    stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memtime() - c0;
  }
}

template <int NW, int PD, int PR, int BAR>
static void run_mix(const char* name, const char* src, long ld, int rows, int per_cu, unsigned long long* stamps) {
  const int steps = 20 * 16;
  constexpr int P = PD + PR;
  const int ring = 2 * P * NW * 1024;
  const int lds = per_cu == 1 ? (ring > 96 * 1024 ? ring : 96 * 1024) : ring;
  hipFuncSetAttribute((const void*)mix_kernel<NW, PD, PR, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int blocks = 256 * per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix_kernel<NW, PD, PR, BAR>), dim3(blocks), dim3(NW * 64), lds, 0, src, ld, rows, steps, stamps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  static unsigned long long h[2 * 512];
  hipMemcpy(h, stamps, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  double rt = 0, ck = 0;
  for (int i = 0; i < blocks; ++i) { rt += (double)h[2 * i]; ck += (double)h[2 * i + 1]; }
  rt /= blocks; ck /= blocks;
  const double bytes_wg = (double)steps * P * NW * 1024;
  printf("%-44s %d WG/CU x %d waves, %d DMA + %d register pieces per wave and step, barrier %d: %6.1f us; %6.1f cycles/step, %5.1f B/clk/CU\n", name, per_cu, NW, PD, PR, BAR, ms * 1e3,
         ck / steps, per_cu * bytes_wg / ck);
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 8192; const long ld = 2560;
  printf("source %d rows x 2560 B = %.1f MB (L2 / Infinity-Cache resident)\n", rows, rows * 2560 / 1e6);
  char* src; unsigned long long* stamps;
  hipMalloc(&src, rows * ld); hipMalloc(&stamps, 2 * 512 * 8);
  hipMemset(src, 1, rows * ld);
  run<4, 8, 1, 1>("128x128-like, two per CU", src, ld, rows, 2, stamps);
  run<4, 8, 1, 1>("128x128-like, ONE per CU", src, ld, rows, 1, stamps);
  run<4, 8, 1, 0>("  ... no barrier", src, ld, rows, 1, stamps);
  run<4, 8, 3, 1>("  ... three steps in flight", src, ld, rows, 1, stamps);
  run<4, 8, 3, 0>("  ... three steps in flight, no barrier", src, ld, rows, 1, stamps);
  run<4, 4, 2, 1>("64x64-like, ONE per CU", src, ld, rows, 1, stamps);
  run<4, 4, 2, 0>("  ... no barrier", src, ld, rows, 1, stamps);
  run<4, 4, 2, 1>("64x64-like, two per CU", src, ld, rows, 2, stamps);
  run<8, 6, 2, 1>("256x160-like (8 waves x 6), ONE per CU", src, ld, rows, 1, stamps);
  run<8, 6, 2, 0>("  ... no barrier", src, ld, rows, 1, stamps);
  run<8, 4, 2, 1>("8 waves x 4, ONE per CU", src, ld, rows, 1, stamps);
  run<8, 4, 1, 1>("8 waves x 4, two per CU", src, ld, rows, 2, stamps);
  run<16, 4, 1, 1>("16 waves x 4, ONE per CU", src, ld, rows, 1, stamps);
  run<16, 2, 2, 1>("16 waves x 2, ONE per CU", src, ld, rows, 1, stamps);
  run<2, 16, 1, 1>("2 waves x 16, ONE per CU", src, ld, rows, 1, stamps);
  run<4, 16, 1, 1>("4 waves x 16, ONE per CU", src, ld, rows, 1, stamps);
  run<4, 16, 1, 0>("  ... no barrier", src, ld, rows, 1, stamps);
  printf("-- mixed staging (LDS-DMA + register-staged ds_write_b128)\n");
  run_mix<4, 8, 0, 1>("DMA only", src, ld, rows, 2, stamps);
  run_mix<4, 0, 8, 1>("registers only", src, ld, rows, 2, stamps);
  run_mix<4, 6, 2, 1>("6 + 2", src, ld, rows, 2, stamps);
  run_mix<4, 4, 4, 1>("4 + 4", src, ld, rows, 2, stamps);
  run_mix<8, 8, 0, 1>("8 waves x 8 (64 KiB per step), DMA only", src, ld, rows, 1, stamps);
  run_mix<8, 6, 2, 1>("8 waves x 8, 6 + 2", src, ld, rows, 1, stamps);
  run_mix<8, 5, 3, 1>("8 waves x 8, 5 + 3", src, ld, rows, 1, stamps);
  run_mix<8, 4, 4, 1>("8 waves x 8, 4 + 4", src, ld, rows, 1, stamps);
  run_mix<8, 0, 8, 1>("8 waves x 8, registers only", src, ld, rows, 1, stamps);
  return 0;
}
