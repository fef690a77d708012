// MFMA GEMM for gfx950: C[M,N] = epilogue(A[M,K] . W[N,K]^T), fp16 in, fp32 accumulate, fp16 out.
//
// One kernel serves every contraction on the denoise path (reference call sites: the nn.Linear calls of
// attention_processor.py:239,246-247,267,344,358-359,379-380,400 and, inside the diffusers UNet, the
// ResnetBlock2D / Downsample2D / Upsample2D 3x3 convolutions, 1x1 shortcuts, proj_in/out and the GEGLU
// feed-forward -- SURVEY.md §2b):
//   * LINEAR  : A rows read directly (optional affine row map, used to pick text / image-token rows of ctx)
//   * CONV3x3 : implicit GEMM over channels-last activations; K = (ky,kx,ci); stride 1/2; optional
//               nearest-x2 upsample folded into the gather; zero padding reads a zero page.
// Structure (cdna_hip_programming.md §5): 256 threads = 4 waves (2x2), BK = 64, both operand tiles staged
// global->LDS with 16-byte `global_load_lds` (per-lane SOURCE address makes the conv gather free), LDS
// double buffered, XOR-swizzled 16-B chunks (chunk ^ ((row>>1)&7): conflict-free ds_read_b128 fragments),
// v_mfma_f32_16x16x32_f16 with W as the first operand so each lane ends up holding 4 consecutive output
// columns of one row (8-byte epilogue stores/loads along N).
#include "gemm_kernel.h"
#include "conv_halo_kernel.h"
#include "gemm_geglu_kernel.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>


// variant id = index into IA2P_GEMM_TILES (common.h)
// Measured on MI355X (tools/gemm_bench.py): occupancy beats ring depth -- a third stage costs a resident block
// (96 KiB LDS at 128x128) and loses 20-30 %, so 2 stages is the default; tile = largest that still gives every
// CU >= 1.5-2 workgroups (the kernels run at ~13 TB/s of L2->LDS traffic, i.e. they are L2-bandwidth bound and
// more co-resident blocks hide the per-k-step load latency).
// generation of everything the executor's plan (and so its workspace allocation pattern) depends on: the measured-plan table AND the test hooks that override it.
// Hosts key their workspace on it (ia2p_plan_generation; HipUNet2DConditionModel.workspace_for) and re-query ia2p_workspace_bytes when it moves.
static std::mutex g_plan_mu;
static unsigned long long g_plan_gen = 0;
static void plan_gen_bump() { std::lock_guard<std::mutex> lk(g_plan_mu); ++g_plan_gen; }
static int g_force_variant = -1;   // test/tuning hook (ia2p_debug_set_gemm_tile)
extern "C" void ia2p_debug_set_gemm_tile(int v) { if (v != g_force_variant) plan_gen_bump(); g_force_variant = v; }
// the tile table, for tools and tests: out = {bm, bn, ring stages, schedule (0 plain, 1 ping-pong, 2 eight-phase, 3 ping-pong over halo-staged patches: 3x3 convolutions only, 4 ping-pong on 32-deep sub-steps: GEGLU launches only)}; returns 0, or -1 past the last variant
extern "C" int ia2p_debug_gemm_tile_info(int v, int* out) {
  if (v < 0 || v >= IA2P_GEMM_NVARIANT || !out) return -1;
  out[0] = IA2P_GEMM_TILES[v].bm; out[1] = IA2P_GEMM_TILES[v].bn; out[2] = IA2P_GEMM_TILES[v].stages; out[3] = IA2P_GEMM_TILES[v].halo ? 3 : IA2P_GEMM_TILES[v].pp;
  return 0;
}

// tuning hook: IA2P_GEMM_RULES="MxNxK=variant;..." overrides the choice for exact shapes (in-situ A/B runs of bench.py)
struct ShapeRule { int M, N, K, v; };
static const std::vector<ShapeRule>& shape_rules() {
  static std::vector<ShapeRule> rules = [] {
    std::vector<ShapeRule> r;
    if (const char* e = ia2p_exp_env("IA2P_GEMM_RULES")) {
      const char* p = e;
      while (*p) {
        ShapeRule x;
        int n = 0;
        if (sscanf(p, "%dx%dx%d=%d%n", &x.M, &x.N, &x.K, &x.v, &n) == 4) { if (x.v >= 0 && x.v < IA2P_GEMM_NVARIANT) r.push_back(x); p += n; }
        while (*p && *p != ';') ++p;
        if (*p == ';') ++p;
      }
    }
    return r;
  }();
  return rules;
}

// C[m,n] = epilogue(sum_s partial[s][m][n]), fixed summation order. One workgroup per output row (256 threads x 4 columns per
// pass), so the row-wise extras of the folded LayerNorm are reductions inside the workgroup: mean / rstd of the row for a consumer
// launch (ln_stats; every wave derives them itself), {sum, sum of squares} of the fp16 output row for a producer launch
// (stats_out, slot 0; wave partials combined through LDS in wave order). No GEGLU here.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* hpartial, half_t* hC, const half_t* hresidual, float* hstats_out, int hM, int hN, int hsplitk,
                                                            int hldc, int hldr, const GemmArgs p) {
  // leading scalars = what the streaming loop needs first (preloaded into SGPRs: build.py PRELOAD); the rest of the epilogue description follows
  // one WAVE per output row (4 rows per workgroup): every load of a row is independent, the row statistics need no workgroup barrier
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = hN >> 2;
  const __amdgpu_buffer_rsrc_t c_rsrc = wt_rsrc((void*)hC, (size_t)hM * hldc * 2);
  for (int m = blockIdx.x * 4 + wave; m < hM; m += gridDim.x * 4) {
    float mu = 0.f, rs = 1.f;
    if (p.ln_stats) {
      float s1 = 0.f, s2 = 0.f;
      for (int sl = lane; sl < p.ln_slots; sl += 64) { const float2 v = ((const float2*)p.ln_stats)[(size_t)sl * hM + m]; s1 += v.x; s2 += v.y; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      const float2 mr = ln_mean_rstd_f(s1, s2, p.K, p.ln_eps);
      mu = mr.x; rs = mr.y;
    }
    float st1 = 0.f, st2 = 0.f;
    for (int q = lane; q < nq; q += 64) {
      const int n = q * 4;
      f4 v = *(const f4*)(hpartial + (size_t)m * hN + n);
      for (int s = 1; s < hsplitk; ++s) {
        const f4 w = *(const f4*)(hpartial + ((size_t)s * hM + m) * hN + n);
        v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
      }
      if (p.ln_stats) {
        const f4 cs = *(const f4*)(p.ln_cs + n), lb = *(const f4*)(p.ln_bias + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ln_fold_f(v[r], mu, rs, cs[r], lb[r]);
      } else {
        const float e_as = p.acc_scale == 0.f ? 1.f : p.acc_scale, e_bs = p.bias_scale == 0.f ? 1.f : p.bias_scale;
        v[0] *= e_as; v[1] *= e_as; v[2] *= e_as; v[3] *= e_as;
        if (p.bias) { const h4 b = *(const h4*)(p.bias + n); v[0] = fmaf((float)b[0], e_bs, v[0]); v[1] = fmaf((float)b[1], e_bs, v[1]); v[2] = fmaf((float)b[2], e_bs, v[2]); v[3] = fmaf((float)b[3], e_bs, v[3]); }
      }
      if (p.act) { v[0] = act_f(v[0], p.act); v[1] = act_f(v[1], p.act); v[2] = act_f(v[2], p.act); v[3] = act_f(v[3], p.act); }
      if (p.rowvec || hresidual) { v[0] = (float)(half_t)v[0]; v[1] = (float)(half_t)v[1]; v[2] = (float)(half_t)v[2]; v[3] = (float)(half_t)v[3]; }      // (as the GEMM epilogues: the layer's output is an fp16 tensor before row / residual are added)
      if (p.rowvec) { const float e_bs = p.bias_scale == 0.f ? 1.f : p.bias_scale; const h4 b = *(const h4*)(p.rowvec + (size_t)(m / p.rows_per_batch) * p.rowvec_ld + n); v[0] = fmaf((float)b[0], e_bs, v[0]); v[1] = fmaf((float)b[1], e_bs, v[1]); v[2] = fmaf((float)b[2], e_bs, v[2]); v[3] = fmaf((float)b[3], e_bs, v[3]); }
      if (hresidual) { const h4 b = *(const h4*)(hresidual + (size_t)m * hldr + n); v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3]; }
      h4 o; o[0] = (half_t)v[0]; o[1] = (half_t)v[1]; o[2] = (half_t)v[2]; o[3] = (half_t)v[3];
      if (p.c_wt) store8_wt(c_rsrc, ((size_t)m * hldc + n) * 2, o);
      else *(h4*)(hC + (size_t)m * hldc + n) = o;
      if (hstats_out) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float f = (float)o[r]; st1 += f; st2 += f * f; }
      }
    }
    if (hstats_out) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { st1 += __shfl_xor(st1, o); st2 += __shfl_xor(st2, o); }
      if (lane == 0) ((float2*)hstats_out)[m] = make_float2(st1, st2);
    }
  }
}

// K-split launches combine their slabs inside the GEMM launch (last-arriving slice of a tile); IA2P_SPLITK_INKERNEL = byte threshold on
// splitk*M*N*4 up to which they do (0: never -- separate splitk_reduce_kernel launches, the A/B switch; default: always).
// Round 2 combined only slab sets <= 8 MiB: the last arriver read the slabs in a dependent loop (~16 serial round trips per thread) and 160 of them
// cost more than a whole-chip reduce launch (+0.6 ms / step at batch 8, profiles/r02e_splitk_inkernel_ab.txt). With the own partial sums kept on chip
// and every load of the other slabs in flight at once the in-launch route is level with the reduce launch at batch 8 (same box, same plans:
// 21.05-21.11 vs 21.10-21.16 ms / step; FF-out launches +4.3 us each against 78 reduce launches of 13.7 us, profiles/r03a_splitk_inkernel_ktable.txt)
// and ahead at batch 1, so every K split now finishes in its own launch: no reduce launches in a step.
static long long g_sk_limit = -1;     // bytes of splitk*M*N*4 up to which a K split combines inside the launch; < 0: IA2P_SPLITK_INKERNEL or the default
extern "C" void ia2p_debug_set_splitk_inkernel(long long bytes) { g_sk_limit = bytes; }
bool ia2p_splitk_inkernel(int M, int N, int splitk) {
  static const size_t env_limit = getenv("IA2P_SPLITK_INKERNEL") ? (size_t)atoll(getenv("IA2P_SPLITK_INKERNEL")) : ((size_t)1 << 46);
  const size_t limit = g_sk_limit >= 0 ? (size_t)g_sk_limit : env_limit;
  return splitk > 1 && (size_t)splitk * M * N * sizeof(float) <= limit;
}
static int g_force_gn = -1;        // test hook (ia2p_debug_set_gn_plan): -1 the plan's own bit, 1 every eligible site fuses its GroupNorm, 0 none does
extern "C" void ia2p_debug_set_gn_plan(int v) { if (v != g_force_gn) plan_gen_bump(); g_force_gn = v; }      // (changes every site's fuse decision: the workspace pattern moves with it)
static int g_force_splitk = -1;    // test/tuning hook
extern "C" void ia2p_debug_set_gemm_splitk(int s) { if (s != g_force_splitk) plan_gen_bump(); g_force_splitk = s; }

// Ticket counters of the in-launch K-split combine: one int per output tile, zero between launches (the last arriver of a tile resets its
// counter). One buffer per (device, stream): launches on one stream are ordered, so two K-split launches can only share counters when they cannot
// run at the same time -- contexts, executors and serving threads that work on different streams never see each other's tickets.
// A launch that dies mid-flight would leave tickets behind: the counters carry an EPOCH. ia2p_sk_counters_invalidate() (called when a context is
// created and whenever the library reports a HIP error) starts a new one, and the first K-split launch of a stream in a new epoch is preceded by a
// hipMemsetAsync of that stream's buffer ON that stream -- per epoch, not per launch (78 K-split launches per step would each pay a memset node).
static constexpr int POOL_INTS = 1 << 18, POOL_SK = POOL_INTS - 4096;
int ia2p_sk_counter_capacity() { return POOL_SK; }
static std::atomic<unsigned> g_sk_epoch{1};
void ia2p_sk_counters_invalidate() { g_sk_epoch.fetch_add(1, std::memory_order_relaxed); }
extern "C" void ia2p_debug_invalidate_splitk_counters(void) { ia2p_sk_counters_invalidate(); }
static int* pool_buffer(hipStream_t s) {
  struct Entry { int* p; unsigned epoch; };
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, Entry> pool;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  const unsigned epoch = g_sk_epoch.load(std::memory_order_relaxed);
  const size_t bytes = (size_t)POOL_INTS * sizeof(int);
  std::lock_guard<std::mutex> lk(mu);
  auto it = pool.find({dev, s});
  if (it != pool.end()) {
    if (it->second.p && it->second.epoch != epoch) {      // first K-split launch of this stream in a new epoch: re-zero, ordered in front of it
      if (hipMemsetAsync(it->second.p, 0, bytes, s) != hipSuccess) { (void)hipGetLastError(); return nullptr; }      // (this launch is finished by a reduce launch instead)
      it->second.epoch = epoch;
    }
    return it->second.p;
  }
  int* p = nullptr;
  if (hipMalloc((void**)&p, bytes) != hipSuccess || hipMemset(p, 0, bytes) != hipSuccess) {      // (hipMemset: synchronous with respect to the host, done before the first launch)
    (void)hipGetLastError();
    if (p) (void)hipFree(p);
    p = nullptr;                      // remembered: this stream's K-splits are finished by reduce launches
  }
  pool[{dev, s}] = Entry{p, epoch};
  return p;
}
int* ia2p_sk_counters(hipStream_t s, int tiles) { return tiles > POOL_SK ? nullptr : pool_buffer(s); }
// tests: every ticket of the stream's buffer := value, ordered on the stream (value != 0 poisons them the way a launch that died mid-flight would); 0 ok, -1 no buffer
extern "C" int ia2p_debug_fill_splitk_counters(void* stream, int value) {
  int* p = pool_buffer((hipStream_t)stream);
  if (!p) return -1;
  return hipMemsetD32Async((hipDeviceptr_t)p, value, POOL_SK, (hipStream_t)stream) == hipSuccess ? 0 : -1;
}
// normal-CDF table of the GEGLU gate activation (gelu_lut_f, common.h): IA2P_PHI_LUT_N entries {Phi(x_i), Phi(x_i+1) - Phi(x_i)}, x_i = (i - 256) / 32,
// computed in double precision (0.5 erfc(-x / sqrt 2)), one copy per device
const float* ia2p_phi_lut() {
  static std::mutex mu;
  static float* tab[64] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!tab[dev]) {
    std::vector<float> h(2 * IA2P_PHI_LUT_N);
    auto phi = [](double x) { return 0.5 * std::erfc(-x * 0.70710678118654752440); };
    for (int i = 0; i < IA2P_PHI_LUT_N; ++i) {
      const double a = phi((i - 256) / 32.0), b = phi((i - 255) / 32.0);
      h[2 * i] = (float)a; h[2 * i + 1] = (float)(b - a);
    }
    float* d = nullptr;
    if (hipMalloc((void**)&d, h.size() * sizeof(float)) != hipSuccess || hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipGetLastError();
      if (d) (void)hipFree(d);
      return nullptr;
    }
    tab[dev] = d;
  }
  return tab[dev];
}
// ---- tile / split-K choice: a small analytic cost model, calibrated on MI355X against the in-place timings of ~5400
// candidate launches that ia2p_autotune logged (IA2P_TUNE_LOG=1) for three workloads (batch 8 and 2 at 512^2, batch 4 at
// 768^2). Per CU, the tiles it owns (a blend of the mean and the worst CU) cost, per k-step, the largest of
//   * MFMA time at ~50 % of peak (a lone workgroup per CU -- one wave per SIMD -- reaches about half of that),
//   * the L2->LDS fill at ~80 GB/s per CU (the ~13-20 TB/s aggregate of DESIGN.md §7),
//   * the load latency of a k-tile over the tiles the LDS ring keeps in flight, once per batch of co-resident workgroups,
// plus a ramp per batch, and a K-split adds the slab round trip of the reduce kernel. The model only has to RANK the
// variants: summed over those workloads its picks cost ~2 % more time than the measured best picks, and the measured
// best is always within 1.55 x of the modelled best (the autotuner's candidate filter uses 1.7).
static double plan_cost_us(int M, int N, int K, bool conv, const GemmTile& t, int sk) {
  constexpr double LAT_US = 0.7, FILL_B_PER_US = 80000.0, MFMA_EFF = 0.5, LONE_EFF = 0.55, RAMP_US = 1.0, BASE_US = 3.0,
                   REDUCE_B_PER_US = 5.0e6, REDUCE_US = 5.0, CU_FLOPS_PER_US = 2.5e15 / 256.0 / 1e6;   // 2.5 PFLOP/s dense fp16 over 256 CUs
  const long tiles = (long)((M + t.bm - 1) / t.bm) * ((N + t.bn - 1) / t.bn) * sk;
  const int nk = std::max(1, (K / 64) / sk);
  const int lds = t.stages * (t.bm + t.bn) * 128;
  const int regs = t.bm * t.bn / 256 + (conv ? 100 : 86);   // accumulators + fragments / addressing (measured allocations)
  const int occ = std::max(1, std::min({160 * 1024 / lds, 512 / regs, 8}));
  const long worst = (tiles + 255) / 256;                   // tiles on the busiest CU
  const int conc = (int)std::min<long>(occ, worst);         // co-resident workgroups there
  const long batches = (worst + conc - 1) / conc;
  const double per_cu = std::max(1.0, 0.6 * tiles / 256.0 + 0.4 * worst);
  const bool pingpong = t.pp != 0;      // 8 waves in two half-step-shifted groups: one workgroup behaves like two co-resident ones
  const double t_mfma = per_cu * (2.0 * t.bm * t.bn * 64) / (MFMA_EFF * (conc == 1 && !pingpong ? LONE_EFF : 1.0) * CU_FLOPS_PER_US);
  const double t_fill = per_cu * (((t.halo ? 36 : t.bm) + t.bn) * 128.0) / FILL_B_PER_US;      // (halo-staged convolution: an 18 x 18 pixel image per nine k-tiles)
  const double t_lat = batches * LAT_US / (t.stages - 1) * (pingpong ? 0.5 : 1.0);
  double total = BASE_US + nk * std::max({t_mfma, t_fill, t_lat}) + RAMP_US * batches;
  if (sk > 1) total += REDUCE_US + 2.0 * sk * (double)M * N * 4 / REDUCE_B_PER_US;
  return total;
}

// ---- measured plans (ia2p_autotune): shape -> (variant, splitk), process-wide; consulted before the cost model
using PlanKey = std::tuple<int, int, int, int, int>;
extern "C" unsigned long long ia2p_plan_generation(void) { std::lock_guard<std::mutex> lk(g_plan_mu); return g_plan_gen; }
static std::map<PlanKey, GemmPlan>& tuned_plans() { static std::map<PlanKey, GemmPlan> m; return m; }

bool ia2p_plan_lookup(int M, int N, int K, bool conv, bool geglu, GemmPlan* out) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  auto it = tuned_plans().find(PlanKey{M, N, K, conv, geglu});
  if (it == tuned_plans().end()) return false;
  if (out) *out = it->second;
  return true;
}
void ia2p_plan_set(int M, int N, int K, bool conv, bool geglu, GemmPlan pl) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  tuned_plans()[PlanKey{M, N, K, conv, geglu}] = pl;
  ++g_plan_gen;
}
// does any measured plan of a 3x3 site with M output rows (one resolution level of a forward) run GroupNorm-fused? Producers of that level's tensors take their
// column sums only then (1 ... 2 us per launch that nobody would read otherwise)
bool ia2p_plan_any_gn(int M) {
  if (g_force_gn >= 0) return g_force_gn > 0;
  std::lock_guard<std::mutex> lk(g_plan_mu);
  for (const auto& kv : tuned_plans())
    if (kv.second.gn && std::get<3>(kv.first) && std::get<0>(kv.first) == M) return true;
  return false;
}
extern "C" void ia2p_plan_clear(void) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  tuned_plans().clear();
  ++g_plan_gen;
}
// text form "M,N,K,conv,geglu,variant,splitk;..." ; returns the length needed (excluding the terminator)
extern "C" size_t ia2p_plan_export(char* buf, size_t len) {
  std::lock_guard<std::mutex> lk(g_plan_mu);
  std::string s;
  char tmp[96];
  for (const auto& kv : tuned_plans()) {      // (an 8th field, "gn", only where it is set: tables without GroupNorm-fused sites keep round 4's seven-field form)
    if (kv.second.gn) snprintf(tmp, sizeof tmp, "%d,%d,%d,%d,%d,%d,%d,%d;", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first),
                               std::get<4>(kv.first), kv.second.variant, kv.second.splitk, kv.second.gn);
    else snprintf(tmp, sizeof tmp, "%d,%d,%d,%d,%d,%d,%d;", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first),
             std::get<4>(kv.first), kv.second.variant, kv.second.splitk);
    s += tmp;
  }
  if (buf && len) { strncpy(buf, s.c_str(), len - 1); buf[len - 1] = 0; }
  return s.size();
}
extern "C" int ia2p_plan_import(const char* text) {   // returns the number of entries read, -1 on a malformed / out-of-range entry
  if (!text) return -1;
  std::vector<std::pair<PlanKey, GemmPlan>> in;
  for (const char* p = text; *p;) {
    int M, N, K, cv, gg, v, sk, gn = 0, n = 0;
    if (sscanf(p, "%d,%d,%d,%d,%d,%d,%d,%d;%n", &M, &N, &K, &cv, &gg, &v, &sk, &gn, &n) != 8 || n == 0) {      // seven fields: gn = 0
      gn = 0; n = 0;
      if (sscanf(p, "%d,%d,%d,%d,%d,%d,%d;%n", &M, &N, &K, &cv, &gg, &v, &sk, &n) != 7 || n == 0) return -1;
    }
    if (M < 1 || N < 1 || K < 64 || v < 0 || v >= IA2P_GEMM_NVARIANT || sk < 1 || sk > K / 64 || (gg && (sk > 1 || IA2P_GEMM_TILES[v].bn % 32)) || (IA2P_GEMM_TILES[v].halo && !cv) || (ia2p_tile_geglu_only(IA2P_GEMM_TILES[v].pp) && !ia2p_geglu320_shape_ok(M, N, K, cv != 0, gg != 0)) ||
        gn < 0 || gn > 1 || (gn && !IA2P_GEMM_TILES[v].halo)) return -1;
    in.push_back({PlanKey{M, N, K, cv != 0, gg != 0}, GemmPlan{v, sk, gn}});
    p += n;
  }
  std::lock_guard<std::mutex> lk(g_plan_mu);
  for (const auto& e : in) tuned_plans()[e.first] = e.second;
  ++g_plan_gen;
  return (int)in.size();
}

// candidates worth measuring for a problem: every (variant, split) whose modelled cost is within `slack` x the best
// modelled cost and whose fp32 slabs fit max_slab_bytes; best-modelled first
void ia2p_gemm_candidates(int M, int N, int K, bool conv, bool geglu, size_t max_slab_bytes, double slack, std::vector<GemmPlan>* out) {
  static const int splits[] = {1, 2, 3, 4, 6, 8};
  const int nk = K / 64;
  std::vector<std::pair<double, GemmPlan>> all;
  static const unsigned excluded = [] {          // IA2P_TUNE_EXCLUDE="12,7": variants the tuner must not consider (A/B runs)
    unsigned m = 0;
    if (const char* e = getenv("IA2P_TUNE_EXCLUDE"))
      for (const char* q = e; *q; ++q)
        if (*q >= '0' && *q <= '9') { const int v = atoi(q); if (v >= 0 && v < 32) m |= 1u << v; while (q[1] >= '0' && q[1] <= '9') ++q; }
    return m;
  }();
  for (int v = 0; v < IA2P_GEMM_NVARIANT; ++v) {
    const GemmTile& t = IA2P_GEMM_TILES[v];
    if ((geglu && t.bn % 32) || (excluded >> v & 1) || (t.halo && !conv) || (ia2p_tile_geglu_only(t.pp) && !ia2p_geglu320_shape_ok(M, N, K, conv, geglu))) continue;
    for (int sk : splits) {
      if (sk > 1 && (geglu || nk / sk < 4 || (size_t)sk * M * N * 4 > max_slab_bytes)) break;
      all.push_back({plan_cost_us(M, N, K, conv, t, sk), GemmPlan{v, sk}});
    }
  }
  std::stable_sort(all.begin(), all.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
  out->clear();
  for (const auto& e : all)
    if (e.first <= slack * all.front().first || (IA2P_GEMM_TILES[e.second.variant].halo && e.second.splitk <= 4) || ia2p_tile_geglu_only(IA2P_GEMM_TILES[e.second.variant].pp))
      out->push_back(e.second);      // (the halo-staged tiles and the GEGLU tile: always measured -- the model was calibrated on the gathered kernels)
}

// Tile variant and split-K factor for a problem: a measured plan if ia2p_autotune recorded one, else the cost model.
// Pure function of the shape and the plan table (the executor sizes its workspace with it).
GemmPlan ia2p_gemm_plan(int M, int N, int K, bool conv, bool geglu) {
  GemmPlan pl{-1, 1};
  // (a forced variant that only takes GEGLU launches of linear layers in whole tiles leaves every other launch to the table / the model: tests force one variant for a whole network)
  const int force_variant = (g_force_variant >= 0 && g_force_variant < IA2P_GEMM_NVARIANT && ia2p_tile_geglu_only(IA2P_GEMM_TILES[g_force_variant].pp) && !ia2p_geglu320_shape_ok(M, N, K, conv, geglu)) ? -1 : g_force_variant;
  if (force_variant < 0 && ia2p_plan_lookup(M, N, K, conv, geglu, &pl)) {
    if (g_force_splitk >= 1 && !geglu) pl.splitk = std::min(g_force_splitk, K / 64);
    if (g_force_gn >= 0) pl.gn = g_force_gn && IA2P_GEMM_TILES[pl.variant].halo;
    return pl;
  }
  for (const ShapeRule& r : shape_rules())
    if (r.M == M && r.N == N && r.K == K) pl.variant = r.v;
  if (force_variant >= 0) pl.variant = force_variant;
  const int nk = K / 64;
  if (pl.variant < 0) {
    static const int splits[] = {1, 2, 3, 4, 6, 8};
    double best = 1e30;
    for (int v = 0; v < IA2P_GEMM_NVARIANT; ++v) {
      const GemmTile& t = IA2P_GEMM_TILES[v];
      if ((geglu && t.bn % 32) || t.halo || ia2p_tile_geglu_only(t.pp)) continue;              // a (value, gate) block of 32 packed columns must not straddle tiles; halo tiles and the GEGLU tile: by measurement only
      for (int sk : splits) {
        if (sk > 1 && (geglu || nk / sk < 4)) break;
        const double c = plan_cost_us(M, N, K, conv, t, sk);
        if (c < best) { best = c; pl.variant = v; pl.splitk = sk; }
      }
    }
  }
  if (g_force_splitk >= 1 && !geglu) pl.splitk = g_force_splitk;
  if (pl.splitk > nk) pl.splitk = nk;
  if (pl.splitk < 1) pl.splitk = 1;
  pl.gn = g_force_gn > 0 && conv && IA2P_GEMM_TILES[pl.variant].halo;      // (the cost model never fuses: by measurement only)
  return pl;
}

extern "C" void ia2p_debug_gemm_plan(int M, int N, int K, int conv, int geglu, int* variant, int* splitk) {
  const GemmPlan pl = ia2p_gemm_plan(M, N, K, conv != 0, geglu != 0);
  if (variant) *variant = pl.variant;
  if (splitk) *splitk = pl.splitk;
}

hipError_t ia2p_launch_splitk_reduce(const GemmArgs& a0, hipStream_t s) {
  GemmArgs a = a0;
  a.c_wt = ((ia2p_wt_mask() & 2) && (size_t)a.M * a.ldc * 2 < (size_t)0x7ffffff0) ? 1 : 0;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(std::min(16384, (a.M + 3) / 4)), dim3(256), 0, s, (const float*)a.partial, a.C, a.residual, a.stats_out, a.M, a.N, a.splitk,
                     a.ldc, a.ldr, a);
  return hipGetLastError();
}

// The ONE place that decides how a K-split launch is finished: counters attached -> the last-arriving K-slice of every tile combines inside the launch;
// no counters (policy, no buffer for this stream, too many tiles) -> the slices leave their slabs and splitk_reduce_kernel finishes, launched here
// (with_reduce) or by the caller, who learns which way it went through *combined.
template <bool CONV>
static hipError_t launch_any(const GemmArgs& a0, int v, hipStream_t s, bool with_reduce, int* combined, int* ran) {
  if (a0.splitk > 1 && !a0.partial) return hipErrorInvalidValue;
  hipError_t e;
  if (v < 0 || v >= IA2P_GEMM_NVARIANT) return hipErrorInvalidValue;
  if (a0.geglu && IA2P_GEMM_TILES[v].bn % 32) return hipErrorInvalidValue;   // a (value, gate) block of 32 packed columns must not straddle tiles
  GemmArgs a = a0;
  a.sk_counters = nullptr;
  if (a.splitk > 1 && ia2p_splitk_inkernel(a.M, a.N, a.splitk))
    a.sk_counters = ia2p_sk_counters(s, ((a.M + IA2P_GEMM_TILES[v].bm - 1) / IA2P_GEMM_TILES[v].bm) * ((a.N + IA2P_GEMM_TILES[v].bn - 1) / IA2P_GEMM_TILES[v].bn));
  if (combined) *combined = a.sk_counters != nullptr;
  // halo-staged variants run their gathered twin at a site they do not take (stride 2, ragged patches, more K slices than channel blocks): `ran` tells the caller which
  // kernel the launch really was (its profile class must match the rocprofv3 trace)
  const bool halo_site = CONV && ia2p_conv_halo_ok(a) && ia2p_conv_gn_ok(a) && a.splitk <= a.Cin / 64;
  if (ran) *ran = ia2p_gemm_variant_ran(a, CONV, v);
  if (a.gn.st0 && !(IA2P_GEMM_TILES[v].halo && halo_site)) return hipErrorInvalidValue;      // a GroupNorm-fused operand: the halo-staged kernel or nothing (callers ask ia2p_conv_gn_fusable first)
  switch (v) {
#define IA2P_TILE_CASE(ID, BM_, BN_, ST_)                                                                                    \
  case ID:                                                                                                                   \
    static_assert(IA2P_GEMM_TILES[ID].bm == BM_ && IA2P_GEMM_TILES[ID].bn == BN_ && IA2P_GEMM_TILES[ID].stages == ST_, "tile table"); \
    e = launch_cfg<BM_, BN_, ST_, CONV>(a, s);                                                                               \
    break;
    IA2P_TILE_CASE(0, 128, 128, 2)
    IA2P_TILE_CASE(1, 128, 128, 3)
    IA2P_TILE_CASE(2, 128, 64, 2)
    IA2P_TILE_CASE(3, 128, 64, 3)
    IA2P_TILE_CASE(4, 64, 64, 2)
    IA2P_TILE_CASE(5, 64, 64, 3)
    IA2P_TILE_CASE(6, 64, 160, 2)
    IA2P_TILE_CASE(7, 64, 160, 3)
    IA2P_TILE_CASE(8, 128, 160, 2)
    IA2P_TILE_CASE(9, 128, 160, 3)
    IA2P_TILE_CASE(10, 160, 128, 2)
    IA2P_TILE_CASE(11, 160, 160, 2)
    IA2P_TILE_CASE(13, 64, 64, 4)
    IA2P_TILE_CASE(14, 64, 64, 6)
    IA2P_TILE_CASE(15, 128, 64, 4)
    IA2P_TILE_CASE(20, 32, 64, 3)
    IA2P_TILE_CASE(21, 32, 128, 3)
    case 16:
      static_assert(IA2P_GEMM_TILES[16].bm == 128 && IA2P_GEMM_TILES[16].bn == 80 && IA2P_GEMM_TILES[16].stages == 2, "tile table");
      e = launch_cfg<128, 80, 2, CONV, 4, 64, 0, 1>(a, s);
      break;
    case 17:
      static_assert(IA2P_GEMM_TILES[17].bm == 128 && IA2P_GEMM_TILES[17].bn == 80 && IA2P_GEMM_TILES[17].stages == 4, "tile table");
      e = launch_cfg<128, 80, 4, CONV, 4, 64, 0, 1>(a, s);
      break;
    case 12:
      static_assert(IA2P_GEMM_TILES[12].bm == 256 && IA2P_GEMM_TILES[12].bn == 128 && IA2P_GEMM_TILES[12].stages == 3, "tile table");
      e = launch_cfg<256, 128, 3, CONV, 4, 64, 1>(a, s);
      break;
    case 18:
      static_assert(IA2P_GEMM_TILES[18].bm == 256 && IA2P_GEMM_TILES[18].bn == 160 && IA2P_GEMM_TILES[18].stages == 3, "tile table");
      e = launch_cfg<256, 160, 3, CONV, 4, 64, 1>(a, s);
      break;
    case 19:
      static_assert(IA2P_GEMM_TILES[19].bm == 128 && IA2P_GEMM_TILES[19].bn == 160 && IA2P_GEMM_TILES[19].stages == 3 && IA2P_GEMM_TILES[19].pp, "tile table");
      e = launch_cfg<128, 160, 3, CONV, 4, 64, 1>(a, s);
      break;
    case 22:
      static_assert(IA2P_GEMM_TILES[22].bm == 256 && IA2P_GEMM_TILES[22].bn == 256 && IA2P_GEMM_TILES[22].stages == 2 && IA2P_GEMM_TILES[22].pp == 2, "tile table");
      e = launch_cfg<256, 256, 2, CONV, 2, 64, 2, 4>(a, s);
      break;
    case 23:
      static_assert(IA2P_GEMM_TILES[23].bm == 256 && IA2P_GEMM_TILES[23].bn == 128 && IA2P_GEMM_TILES[23].stages == 2 && IA2P_GEMM_TILES[23].pp == 2, "tile table");
      e = launch_cfg<256, 128, 2, CONV, 2, 64, 2, 4>(a, s);
      break;
    case 27:      // GEGLU launches of linear layers only (gemm_geglu_kernel.h)
      static_assert(IA2P_GEMM_TILES[27].bm == 256 && IA2P_GEMM_TILES[27].bn == 320 && IA2P_GEMM_TILES[27].stages == 2 && IA2P_GEMM_TILES[27].pp == 4, "tile table");
      e = (!CONV && ia2p_geglu320_ok(a)) ? launch_geglu320(a, s) : launch_cfg<256, 160, 3, CONV, 4, 64, 1>(a, s);      // (a launch it does not take -- row map, odd strides -- runs the 256 x 160 ping-pong tile)
      break;
    case 24:      // halo-staged convolution; a launch it does not take (linear layer, stride 2, ragged patches, more K slices than blocks of 64 channels) runs the same tile shape with the gathered operand
      static_assert(IA2P_GEMM_TILES[24].bm == 256 && IA2P_GEMM_TILES[24].bn == 160 && IA2P_GEMM_TILES[24].halo, "tile table");
      e = halo_site ? launch_halo<160>(a, s) : launch_cfg<256, 160, 3, CONV, 4, 64, 1>(a, s);
      break;
    case 26:
      static_assert(IA2P_GEMM_TILES[26].bm == 256 && IA2P_GEMM_TILES[26].bn == 80 && IA2P_GEMM_TILES[26].halo, "tile table");
      e = halo_site ? launch_halo<80>(a, s) : launch_cfg<128, 80, 2, CONV, 4, 64, 0, 1>(a, s);
      break;
    case 25:
      static_assert(IA2P_GEMM_TILES[25].bm == 256 && IA2P_GEMM_TILES[25].bn == 128 && IA2P_GEMM_TILES[25].halo, "tile table");
      e = halo_site ? launch_halo<128>(a, s) : launch_cfg<256, 128, 3, CONV, 4, 64, 1>(a, s);
      break;
#undef IA2P_TILE_CASE
    // Measured and dropped in round 1 (tools/gemm_bench.py): 8-wave 256x128 (2- and 3-stage) and 256x320 at one
    // workgroup per CU; BK = 32 rings (64-byte rows halve the request efficiency). The template still takes WGM and BK.
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess || a.splitk <= 1 || !with_reduce || a.sk_counters) return e;
  return ia2p_launch_splitk_reduce(a, s);
}

// the variant whose kernel a launch of plan variant v really is (a.splitk as the caller set it)
int ia2p_gemm_variant_ran(const GemmArgs& a, bool conv, int v) {
  if (v >= 0 && v < IA2P_GEMM_NVARIANT && ia2p_tile_geglu_only(IA2P_GEMM_TILES[v].pp)) return (!conv && ia2p_geglu320_ok(a)) ? v : 18;
  if (v < 0 || v >= IA2P_GEMM_NVARIANT || !IA2P_GEMM_TILES[v].halo) return v;
  const bool halo_site = conv && ia2p_conv_halo_ok(a) && ia2p_conv_gn_ok(a) && a.splitk <= a.Cin / 64;
  return halo_site ? v : (v == 24 ? 18 : v == 25 ? 12 : 16);
}
// may this 3x3 site run GroupNorm-fused under plan (v, splitk)? (shape part of the answer: the statistics pointers may still be null -- dry pass)
bool ia2p_conv_gn_fusable(const GemmArgs& a, int v, int splitk) {
  return v >= 0 && v < IA2P_GEMM_NVARIANT && IA2P_GEMM_TILES[v].halo && ia2p_conv_halo_ok(a) && a.up == 0 && splitk <= a.Cin / 64;
}
// every (halo-staged tile, K split) a GroupNorm-fused launch of this site may run on, for the tuner
void ia2p_conv_gn_candidates(const GemmArgs& a, size_t max_slab_bytes, std::vector<GemmPlan>* out) {
  out->clear();
  static const int splits[] = {1, 2, 3, 4};
  for (int v = 0; v < IA2P_GEMM_NVARIANT; ++v)
    for (int sk : splits)
      if (ia2p_conv_gn_fusable(a, v, sk) && (sk == 1 || ((a.K / 64) / sk >= 4 && (size_t)sk * a.M * a.N * 4 <= max_slab_bytes))) out->push_back(GemmPlan{v, sk, 1});
}
// launch with an explicit tile variant (a.splitk / a.partial as the caller set them)
// with_reduce = false: a K-split launch leaves its slabs for a separate ia2p_launch_splitk_reduce (the executor times the two apart)
hipError_t ia2p_launch_gemm_variant(const GemmArgs& a, bool conv, int variant, hipStream_t s, bool with_reduce, int* combined, int* ran) {
  return conv ? launch_any<true>(a, variant, s, with_reduce, combined, ran) : launch_any<false>(a, variant, s, with_reduce, combined, ran);
}
// *picked (optional) receives the variant id. a.splitk / a.partial must follow ia2p_gemm_plan (the caller owns the slabs).
hipError_t ia2p_launch_gemm(const GemmArgs& a, bool conv, hipStream_t s, int* picked, int* combined) {
  const GemmPlan pl = ia2p_gemm_plan(a.M, a.N, a.K, conv, a.geglu != 0);
  if (picked) *picked = pl.variant;
  return ia2p_launch_gemm_variant(a, conv, pl.variant, s, true, combined, nullptr);
}
