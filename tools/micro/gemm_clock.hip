// In-kernel clock of the GEMM tiles under load (MI355X_MICROARCH.md "DVFS give-back" item 6): a DIAGNOSTIC build of the product's kernel template
// (gemm_kernel.h compiled with -DIA2P_CLOCK_STAMP: s_memtime / s_memrealtime stamped around the k-loop, written to a buffer of their own) is
// launched back to back on random fp16 data for >= 2 s; clock = median over workgroups of d(s_memtime) / d(s_memrealtime) x 100 MHz.
// The product library is built WITHOUT the stamps. Build (from the repo root):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIA2P_CLOCK_STAMP -mllvm -amdgpu-kernarg-preload-count=16 tools/micro/gemm_clock.hip -o tools/micro/gemm_clock
#include "../../instructany2pix_amd/csrc/gemm_kernel.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

bool ia2p_splitk_inkernel(int, int, int) { return false; }
int ia2p_sk_counter_capacity() { return 1 << 18; }
int* ia2p_sk_counters(hipStream_t, int) { return nullptr; }
void ia2p_sk_counters_invalidate() {}
const float* ia2p_phi_lut() { static float* d = nullptr; if (!d) { hipMalloc(&d, 2 * IA2P_PHI_LUT_N * sizeof(float)); hipMemset(d, 0, 2 * IA2P_PHI_LUT_N * sizeof(float)); } return d; }   // (timing only)

template <int BM, int BN, int ST, int WGM, int PP, int WGN = 2>
static void run(const char* name, int M, int N, int K, const half_t* A, const half_t* W, half_t* C, const half_t* zero, unsigned long long* stamps, int geglu = 0) {
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1; a.A = A; a.W = W; a.C = C; a.zero = zero; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = geglu ? N / 2 : N; a.rows_per_batch = 1;
  a.geglu = geglu; if (geglu) a.bias = W;        // (any N readable halves: timing only)
  a.partial = (float*)stamps; a.acc_scale = a.bias_scale = 1.f;
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  float ms = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) launch_cfg<BM, BN, ST, false, WGM, 64, PP, WGN>(a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    launches += 200;
  }
  std::vector<unsigned long long> h(8 * tiles);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> ghz, cyc, pro, epi, epi_issue, epi_at;
  unsigned long long first_entry = ~0ull, last_exit = 0;
  for (int i = 0; i < tiles; ++i) if (h[8 * i + 1]) {
    ghz.push_back((double)h[8 * i] / (double)h[8 * i + 1] * 0.1); cyc.push_back((double)h[8 * i]);
    pro.push_back((double)(h[8 * i + 3] - h[8 * i + 2]) * 0.01); epi.push_back((double)(h[8 * i + 5] - h[8 * i + 4]) * 0.01);     // us (100 MHz counter)
    epi_issue.push_back((double)(h[8 * i + 6] - h[8 * i + 4]) * 0.01);
    if (h[8 * i + 7]) epi_at.push_back((double)(h[8 * i + 7] - h[8 * i + 4]) * 0.01);
    first_entry = std::min(first_entry, h[8 * i + 2]); last_exit = std::max(last_exit, h[8 * i + 5]);
  }
  std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end()); std::sort(pro.begin(), pro.end()); std::sort(epi.begin(), epi.end()); std::sort(epi_issue.begin(), epi_issue.end());
  const double us = ms * 1e3 / 200, tf = 2.0 * M * N * K / us / 1e6, g = ghz[ghz.size() / 2], c = cyc[cyc.size() / 2];
  const double loop_us = c / (g * 1e3);
  printf("%-22s %5d x %5d x %5d: %7.2f us/launch = %6.0f TFLOP/s; in-kernel clock %.2f GHz (min %.2f max %.2f over %zu workgroups); k-loop %6.0f cycles = %5.2f us"
         " = %5.1f cycles per k-step; MFMA peak AT THAT CLOCK %.0f TFLOP/s\n", name, M, N, K, us, tf, g, ghz.front(), ghz.back(), ghz.size(), c, loop_us,
         c / (K / 64), 256 * 4 * 1024.0 * g / 1e3);
  std::sort(epi_at.begin(), epi_at.end());
  if (!epi_at.empty()) printf("%-22s   k-loop end -> the fp16 tile is in LDS (register epilogue; fp32 route with -DIA2P_STAMP_AT=1|2|3: chunk 0 in LDS | chunk 0 stored | chunk 1 in LDS) %.2f us (median)\n", "", epi_at[epi_at.size() / 2]);
  printf("%-22s   k-loop end -> last C store ISSUED by wave 0 %.2f us (median; max %.2f)\n", "", epi_issue[epi_issue.size() / 2], epi_issue.back());
  printf("%-22s   entry -> k-loop %.2f us (median; max %.2f), k-loop end -> stores left %.2f us (median; max %.2f), first entry -> last exit %.2f us, so %.2f us of the"
         " launch interval lie between kernels\n", "", pro[pro.size() / 2], pro.back(), epi[epi.size() / 2], epi.back(), (double)(last_exit - first_entry) * 0.01,
         us - (double)(last_exit - first_entry) * 0.01);
}

int main() {
  const int M = 4096, Nmax = 10240, Kmax = 5120;      // (M = 4096: the 4096^3 probe shape; the step's shapes use the first 2048 rows)
  const int M2 = 2048;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<half_t> hA((size_t)M * Kmax), hW((size_t)Nmax * Kmax);
  for (auto& v : hA) v = (half_t)nd(rng);
  for (auto& v : hW) v = (half_t)(nd(rng) * 0.02f);
  half_t *A, *W, *C, *zero; unsigned long long* stamps;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * Nmax * 2); hipMalloc(&zero, 4096); hipMalloc(&stamps, 1 << 20);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  hipMemset(zero, 0, 4096); hipMemset(stamps, 0, 1 << 20);
  if (getenv("IA2P_CLOCK_EPI")) {         // round 4: where the fixed cost of a launch goes (epilogue arithmetic vs the drain of its stores)
    run<256, 160, 3, 4, 1>("256x160x3 pp GEGLU", M2, 10240, 1280, A, W, C, zero, stamps, 1);
    run<256, 160, 3, 4, 1>("256x160x3 pp GEGLU K=64", M2, 10240, 64, A, W, C, zero, stamps, 1);
    run<256, 160, 3, 4, 1>("256x160x3 pp N=5120", M2, 5120, 1280, A, W, C, zero, stamps);
    run<256, 256, 2, 2, 2, 4>("256x256 8-phase QKV", 2048, 3840, 1280, A, W, C, zero, stamps);
    run<64, 64, 2, 2, 0>("64x64x2 (out-proj)", M2, 1280, 1280, A, W, C, zero, stamps);
    run<128, 128, 2, 2, 0>("128x128x2 (QKV)", M2, 3840, 1280, A, W, C, zero, stamps);
    return 0;
  }
  if (getenv("IA2P_CLOCK_8PHASE")) {      // round 4: the 8-phase 256 x 256 tile against the 256 x 128 ping-pong tile on the probe shape
    run<256, 256, 2, 2, 2, 4>("256x256 8-phase 4096^3", 4096, 4096, 4096, A, W, C, zero, stamps);
    run<256, 128, 3, 4, 1>("256x128x3 pp 4096^3", 4096, 4096, 4096, A, W, C, zero, stamps);
    run<256, 256, 2, 2, 2, 4>("256x256 8-phase K=1280", 4096, 4096, 1280, A, W, C, zero, stamps);
    run<256, 256, 2, 2, 2, 4>("256x256 8-phase QKV", 2048, 3840, 1280, A, W, C, zero, stamps);
    return 0;
  }
  run<128, 128, 2, 2, 0>("128x128x2 (QKV)", M2, 3840, 1280, A, W, C, zero, stamps);
  run<256, 128, 3, 4, 1>("256x128x3 pp (QKV)", M2, 3840, 1280, A, W, C, zero, stamps);
  run<128, 160, 2, 2, 0>("128x160x2 (FF-in*)", M2, 10240, 1280, A, W, C, zero, stamps);
  run<256, 128, 3, 4, 1>("256x128x3 pp K=5120", M2, 3840, 5120, A, W, C, zero, stamps);
  run<128, 128, 2, 2, 0>("128x128x2 K=5120", M2, 3840, 5120, A, W, C, zero, stamps);
  run<64, 64, 2, 2, 0>("64x64x2 (out-proj)", M2, 1280, 1280, A, W, C, zero, stamps);
  run<256, 160, 3, 4, 1>("256x160x3 pp (FF-in*)", M2, 10240, 1280, A, W, C, zero, stamps);
  run<256, 160, 3, 4, 1>("256x160x3 pp GEGLU", M2, 10240, 1280, A, W, C, zero, stamps, 1);
  run<128, 160, 2, 2, 0>("128x160x2 GEGLU", M2, 10240, 1280, A, W, C, zero, stamps, 1);
  run<256, 160, 3, 4, 1>("256x160x3 pp (QKV)", M2, 3840, 1280, A, W, C, zero, stamps);
  // batch-1 shapes (M = 256: 80-240 workgroups, at most one per CU): where does a latency-bound launch spend its time?
  run<64, 64, 3, 2, 0>("64x64x3 M=256 out-proj", 256, 1280, 1280, A, W, C, zero, stamps);
  run<64, 64, 4, 2, 0>("64x64x4 M=256 (swp)", 256, 1280, 1280, A, W, C, zero, stamps);
  run<64, 64, 3, 2, 0>("64x64x3 M=256 QKV", 256, 3840, 1280, A, W, C, zero, stamps);
  run<64, 64, 4, 2, 0>("64x64x4 M=256 QKV (swp)", 256, 3840, 1280, A, W, C, zero, stamps);
  run<64, 64, 3, 4, 0>("64x64x3 8 waves M=256", 256, 1280, 1280, A, W, C, zero, stamps);
  run<64, 64, 3, 4, 0>("64x64x3 8 waves QKV", 256, 3840, 1280, A, W, C, zero, stamps);
  run<64, 64, 3, 4, 0>("64x64x3 8 waves K=5120", 256, 1280, 5120, A, W, C, zero, stamps);
  run<128, 64, 3, 4, 0>("128x64x3 8 waves QKV", 256, 3840, 1280, A, W, C, zero, stamps);
  run<128, 64, 3, 2, 0>("128x64x3 4 waves QKV", 256, 3840, 1280, A, W, C, zero, stamps);
  run<32, 64, 3, 2, 0>("32x64x3 M=256", 256, 1280, 1280, A, W, C, zero, stamps);
  run<32, 64, 3, 2, 0>("32x64x3 QKV", 256, 3840, 1280, A, W, C, zero, stamps);
  run<32, 64, 3, 2, 0>("32x64x3 K=5120", 256, 1280, 5120, A, W, C, zero, stamps);
  run<64, 64, 6, 2, 0>("64x64x6 M=256 (swp)", 256, 1280, 1280, A, W, C, zero, stamps);
  run<64, 64, 6, 2, 0>("64x64x6 M=256 QKV (swp)", 256, 3840, 1280, A, W, C, zero, stamps);
  run<64, 64, 6, 2, 0>("64x64x6 M=256 K=5120", 256, 1280, 5120, A, W, C, zero, stamps);
  run<64, 64, 2, 2, 0>("64x64x2 M=256 K=5120", 256, 1280, 5120, A, W, C, zero, stamps);
  run<64, 64, 4, 2, 0>("64x64x4 M=256 K=5120", 256, 1280, 5120, A, W, C, zero, stamps);
  return 0;
}
