// Where a k-tile of the 256 x 320 GEGLU tile (gemm_geglu_kernel.h) goes: a DIAGNOSTIC build of the product's kernel (-DIA2P_CLOCK_STAMP -DIA2P_G320_PHASES: s_memtime around every
// interval of the k-loop -- cycles a wave WORKS up to its own counted wait, cycles it then WAITS in the workgroup barrier -- for one wave of each group) launched back to back on
// random fp16 data. The product library is built WITHOUT the stamps. Build and run (from the repo root, on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIA2P_CLOCK_STAMP -DIA2P_G320_PHASES -mllvm -amdgpu-kernarg-preload-count=16 tools/micro/geglu_clock.hip -o tools/micro/geglu_clock && tools/micro/geglu_clock
#include "../../instructany2pix_amd/csrc/gemm_geglu_kernel.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

const float* ia2p_phi_lut() { static float* d = nullptr; if (!d) { hipMalloc(&d, 2 * IA2P_PHI_LUT_N * sizeof(float)); hipMemset(d, 0, 2 * IA2P_PHI_LUT_N * sizeof(float)); } return d; }   // (timing only)

static void run(const char* name, int M, int N, int K, const half_t* A, const half_t* W, half_t* C, const half_t* zero, unsigned long long* stamps) {
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1; a.A = A; a.W = W; a.C = C; a.zero = zero; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = N / 2; a.rows_per_batch = 1;
  a.geglu = 1; a.bias = W; a.partial = (float*)stamps; a.acc_scale = a.bias_scale = 1.f;
  const int tiles = (M / 256) * (N / 320), nk = K / 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) if (launch_geglu320(a, 0) != hipSuccess) { printf("launch refused\n"); return; }
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) launch_geglu320(a, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)24 * tiles);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  std::vector<double> ghz, loop_cyc, pro, epi;
  for (int i = 0; i < tiles; ++i) if (h[8 * i + 1]) {
    ghz.push_back((double)h[8 * i] / (double)h[8 * i + 1] * 0.1); loop_cyc.push_back((double)h[8 * i]);
    pro.push_back((double)(h[8 * i + 3] - h[8 * i + 2]) * 0.01); epi.push_back((double)(h[8 * i + 5] - h[8 * i + 4]) * 0.01);
  }
  const double us = ms * 1e3 / 200, g = med(ghz), c = med(loop_cyc);
  printf("%-18s %5d x %5d x %5d: %7.2f us/launch = %6.0f TFLOP/s; clock %.2f GHz; k-loop %7.0f cycles = %5.2f us = %6.0f cycles per k-tile (MFMA issue: 2560); entry -> k-loop %.2f us, k-loop end -> stores left %.2f us\n",
         name, M, N, K, us, 2.0 * M * N * K / us / 1e6, g, c, c / (g * 1e3), c / nk, med(pro), med(epi));
  static const char* what[2][4] = {{"I(4t)   read kk0 + 5 W pieces", "I(4t+1) 40 MFMAs", "I(4t+2) read kk1 + 4 A pieces", "I(4t+3) 40 MFMAs"},
                                   {"I(4t)   40 MFMAs (t-1, kk1)", "I(4t+1) read kk0 + 5 W pieces", "I(4t+2) 40 MFMAs", "I(4t+3) read kk1 + 4 A pieces"}};
  for (int grp = 0; grp < 2; ++grp)
    for (int k = 0; k < 4; ++k) {
      std::vector<double> w, b;
      for (int i = 0; i < tiles; ++i) { w.push_back((double)h[8 * tiles + 16 * i + 8 * grp + k] / nk); b.push_back((double)h[8 * tiles + 16 * i + 8 * grp + 4 + k] / nk); }
      printf("    group %d  %-32s works %6.0f cycles, waits in the barrier %6.0f   (per k-tile, median over workgroups; s_memtime costs a few dozen cycles per stamp)\n", grp, what[grp][k], med(w), med(b));
    }
}

int main() {
  const int M = 8192, Nmax = 10240, Kmax = 2560;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<half_t> hA((size_t)M * Kmax), hW((size_t)Nmax * Kmax);
  for (auto& v : hA) v = (half_t)nd(rng);
  for (auto& v : hW) v = (half_t)(nd(rng) * 0.02f);
  half_t *A, *W, *C, *zero; unsigned long long* stamps;
  hipMalloc(&A, hA.size() * 2); hipMalloc(&W, hW.size() * 2); hipMalloc(&C, (size_t)M * Nmax); hipMalloc(&zero, 4096); hipMalloc(&stamps, 1 << 20);
  hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice);
  hipMemset(zero, 0, 4096); hipMemset(stamps, 0, 1 << 20);
  run("FF-in level 2", 2048, 10240, 1280, A, W, C, zero, stamps);
  run("FF-in level 1", 8192, 5120, 640, A, W, C, zero, stamps);
  run("K = 2560", 2048, 10240, 2560, A, W, C, zero, stamps);
  return 0;
}
