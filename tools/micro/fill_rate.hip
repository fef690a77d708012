// L2 -> LDS fill-rate ceiling on gfx950: a workgroup of 4 waves streams 128-byte rows of an L2-resident matrix into LDS with
// 16-byte LDS-DMA (global_load_lds, 1 KiB per wave-instruction), exactly as gemm_f16_kernel stages its operand tiles, and does
// nothing else. Compared with the same traffic loaded into registers. Build: hipcc --offload-arch=gfx950 -O3 fill_rate.hip -o fill_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define GLDS16(gptr, ldsptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

template <int PIECES, int MODE>   // MODE 0: LDS-DMA, 1: registers; PIECES = 1-KiB pieces per wave per step
__global__ __launch_bounds__(256, 2) void fill_kernel(const char* src, long ld, int rows, int steps, int wait_each, unsigned* sink) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (blockIdx.x * 131) % (rows - PIECES * 4 * 8);          // tile origin
  const char* p[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) p[i] = src + (long)(r0 + (wave * PIECES + i) * 8 + lane / 8) * ld + (lane % 8) * 16;
  unsigned acc = 0;
  for (int s = 0; s < steps; ++s) {
    if (s % 20 == 0 && s) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) p[i] -= 20 * 128;      // next pass over the same K = 1280 row block
    }
    char* dst = smem + (s & 1) * (PIECES * 4 * 1024) + wave * PIECES * 1024;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) { GLDS16(p[i], dst + i * 1024); p[i] += 128; }
    } else {
      uint4 v[PIECES];
#pragma unroll
      for (int i = 0; i < PIECES; ++i) { v[i] = *(const uint4*)p[i]; p[i] += 128; }
#pragma unroll
      for (int i = 0; i < PIECES; ++i) acc ^= v[i].x ^ v[i].w;
    }
    if (wait_each) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345) sink[0] = acc;
}

template <int PIECES, int MODE>
static void run(const char* name, const char* src, long ld, int rows, int blocks, int wait_each, unsigned* sink) {
  const int steps = 20 * 8;   // K = 1280 x 8 passes
  const int smem = MODE == 0 ? 2 * PIECES * 4 * 1024 : 1024;
  hipFuncSetAttribute((const void*)fill_kernel<PIECES, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const int lds = smem < 65536 ? smem : 65536;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<PIECES, MODE>), dim3(blocks), dim3(256), lds, 0, src, ld, rows, steps, wait_each, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)blocks * steps * PIECES * 4 * 1024;
  printf("%-28s blocks %4d pieces/wave %d wait_each %d : %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz, 256 CUs)\n", name, blocks, PIECES, wait_each, ms * 1e3,
         bytes / ms / 1e9, bytes / (ms * 1e-3) / 2.4e9 / 256);
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 8192; const long ld = 2560;        // rows x 1280 fp16 (8192 rows = 21 MB: L2 + Infinity Cache resident)
  printf("source %d rows x 2560 B = %.1f MB\n", rows, rows * 2560 / 1e6);
  char* src; unsigned* sink;
  hipMalloc(&src, rows * ld); hipMalloc(&sink, 64);
  hipMemset(src, 1, rows * ld);
  for (int blocks : {512, 1024}) {
    run<8, 0>("lds-dma (128x128 tile)", src, ld, rows, blocks, 1, sink);
    run<8, 0>("lds-dma, no per-step wait", src, ld, rows, blocks, 0, sink);
    run<4, 0>("lds-dma (64x64 tile)", src, ld, rows, blocks, 1, sink);
    run<8, 1>("registers", src, ld, rows, blocks, 1, sink);
    run<8, 1>("registers, no per-step wait", src, ld, rows, blocks, 0, sink);
  }
  return 0;
}
