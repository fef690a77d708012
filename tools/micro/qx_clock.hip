// In-kernel stamps of the fused to_q + cross-attention launch (qxattn.hip) at the headline shape: 2048 x 1280 x 1280 projection, 8 images x 20 heads x 256
// queries against 77 text + 4 image-token keys. DIAGNOSTIC build of the product's sources (-DIA2P_CLOCK_STAMP); the product library carries no stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIA2P_CLOCK_STAMP -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-mfma-vgpr-form tools/micro/qx_clock.hip -o tools/micro/qx_clock
#include "../../instructany2pix_amd/csrc/qxattn.hip"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

bool ia2p_splitk_inkernel(int, int, int) { return false; }
int ia2p_sk_counter_capacity() { return 1 << 18; }
int* ia2p_sk_counters(hipStream_t, int) { return nullptr; }
bool ia2p_chain_words(hipStream_t, int**, int**, unsigned**) { return false; }
const float* ia2p_phi_lut() { return nullptr; }

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
  const int B = 8, heads = 20, Nq = 256, M = B * Nq, N = heads * 64, K = 1280, Lt = 77, Li = 4, kv_ld = 2 * N;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  auto dev = [&](size_t n, float sc) { std::vector<half_t> h(n); for (auto& v : h) v = (half_t)(nd(rng) * sc); half_t* d; hipMalloc(&d, n * 2); hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice); return d; };
  half_t *A = dev((size_t)M * K, 1.f), *W = dev((size_t)N * K, 0.03f), *kvt = dev((size_t)B * Lt * kv_ld, 1.f), *kvi = dev((size_t)B * Li * kv_ld, 1.f), *bias = dev(N, 0.1f);
  half_t *O, *zero; unsigned long long* stamps;
  hipMalloc(&O, (size_t)M * N * 2); hipMalloc(&zero, 4096); hipMemset(zero, 0, 4096); hipMalloc(&stamps, 1 << 20); hipMemset(stamps, 0, 1 << 20);
  GemmArgs a; memset(&a, 0, sizeof a);
  a.pad = 1; a.A = A; a.W = W; a.zero = zero; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.ldc = N; a.rows_per_batch = 1; a.bias = bias; a.acc_scale = a.bias_scale = 1.f;
  a.partial = (float*)stamps;
  AttnArgs x; memset(&x, 0, sizeof x);
  x.O = O; x.ldo = N; x.ldq = N; x.B = B; x.heads = heads; x.Nq = Nq; x.nseg = 2; x.scale_log2e = 0.125f * 1.4426950408889634f;
  x.seg[0] = AttnSeg{kvt, kvt + N, Lt, kv_ld, Lt, 1.0f};
  x.seg[1] = AttnSeg{kvi, kvi + N, Li, kv_ld, Li, 0.8f};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rows : {128}) {
    hipMemset(stamps, 0, 1 << 20);
    float ms = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.5) {
      hipEventRecord(e0);
      for (int i = 0; i < 200; ++i) if (ia2p_launch_qproj_xattn(a, x, 0) != hipSuccess) { printf("launch failed\n"); return 1; }
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const int tiles = (M / rows) * heads;
    std::vector<unsigned long long> h(8 * tiles);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> pro, loop, qbuild, core, store, ghz;
    unsigned long long first = ~0ull, last = 0;
    for (int i = 0; i < tiles; ++i) if (h[8 * i + 1]) {
      const unsigned long long* o = &h[8 * i];      // o[0] loop cycles, o[1] loop realtime, o[2] entry, o[3] loop start, o[4] loop end, o[6] Q + K/V ready, o[7] core done, o[5] stores left
      ghz.push_back((double)o[0] / (double)o[1] * 0.1);
      pro.push_back((o[3] - o[2]) * 0.01); loop.push_back((o[4] - o[3]) * 0.01); qbuild.push_back((o[6] - o[4]) * 0.01); core.push_back((o[7] - o[6]) * 0.01); store.push_back((o[5] - o[7]) * 0.01);
      first = std::min(first, o[2]); last = std::max(last, o[5]);
    }
    printf("to_q + cross-attention %d x %d x %d, %d query rows per workgroup, %d workgroups: %.2f us/launch (back to back, warm); in-kernel clock %.2f GHz\n", M, N, K, rows, tiles, ms * 1e3 / 200, med(ghz));
    printf("  medians per workgroup: entry -> k-loop %.2f us | k-loop %.2f us (%d k-steps) | tile through LDS, Q fragments, K/V images %.2f us | attention core %.2f us | O store + drain %.2f us\n",
           med(pro), med(loop), K / 64, med(qbuild), med(core), med(store));
    printf("  first entry -> last exit %.2f us\n", (double)(last - first) * 0.01);
  }
  return 0;
}
