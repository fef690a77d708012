// Shared device/host helpers for the gfx950 (CDNA4) kernels of the denoise hot path.
// Activations are fp16, channels-last: a feature map [B,C,H,W] of the reference lives in HBM as
// [B*H*W, C] row-major ("tokens x channels"), so the transformer path needs no permutes and every
// convolution is an implicit GEMM whose K axis (tap, channel) is contiguous in channels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));

#define IA2P_WAVE 64

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: these run once per activation element in HBM-bound kernels
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far inside the fp16 output grid): one v_rcp, one v_exp and five FMAs instead of
// the ~40-instruction libm erff -- the GEGLU epilogue evaluates it 10.5 M times per feed-forward launch
__device__ __forceinline__ float erf_as_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f)); }
// Exact-erf GELU of the GEGLU gate through a table of the normal CDF in LDS: Phi(x) on [-8, 8) in steps of 1/32 (513 float2 entries {Phi(x_i), Phi(x_i+1) - Phi(x_i)},
// built in double precision on the host, gemm.hip), linear interpolation -- |error of Phi| <= h^2/8 max|Phi''| = 3e-5, i.e. below a tenth of an fp16 ulp of
// the product it scales. 8 VALU slots and one LDS read per gate instead of ~20 slots (v_rcp, v_exp, a 5-term polynomial): the GEGLU epilogue is VALU-bound
// (20 480 gates per 256 x 160 tile; 5 of the 8 us between the k-loop's end and the last store, profiles/r03f_gemm_inkernel_clock.txt).
constexpr int IA2P_PHI_LUT_N = 513;
__device__ __forceinline__ float gelu_lut_f(float g, const float2* lut) {
  const float f = fmaf(__builtin_amdgcn_fmed3f(g, -8.0f, 7.99f), 32.0f, 256.0f);
  const float fl = floorf(f);
#ifdef IA2P_TIMING_NOLUT      // (timing experiments only, tools/micro/geglu_clock.hip: what the table gather costs; results are wrong)
  const float2 e = lut[threadIdx.x & 63];
#else
  const float2 e = lut[(int)fl];
#endif
  return g * fmaf(f - fl, e.y, e.x);
}
__device__ __forceinline__ float quick_gelu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float gelu_tanh_f(float x) { return 0.5f * x * (1.0f + tanhf(0.79788456080286535588f * (x + 0.044715f * x * x * x))); }   // "gelu_new" (GPT-2)
// Write-through (`sc1`) stores for tensors the NEXT kernel reads: the bytes leave the L2 while the kernel still runs, so the end-of-kernel
// write-back (which sits on the critical path between two dependent launches) has nothing left to flush. Offsets are 32-bit: callers use
// them only for tensors under 2 GiB.
#ifndef IA2P_WT_AUX
#define IA2P_WT_AUX 16      // cache-policy bits of the write-through stores: 16 = sc1 (build-time knob for A/B builds: IA2P_EXTRA_FLAGS=-DIA2P_WT_AUX=18 adds nt)
#endif
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_rsrc(void* base, size_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(bytes < (size_t)0x7ffffff0 ? bytes : (size_t)0x7ffffff0), 0x00020000);
}
__device__ __forceinline__ void store16_wt(__amdgpu_buffer_rsrc_t r, size_t byte_off, h8 v) {
  typedef unsigned u4v __attribute__((__vector_size__(4 * sizeof(unsigned))));
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), r, (int)byte_off, 0, IA2P_WT_AUX);
}
__device__ __forceinline__ void store8_wt(__amdgpu_buffer_rsrc_t r, size_t byte_off, h4 v) {
  typedef unsigned u2v __attribute__((__vector_size__(2 * sizeof(unsigned))));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, v), r, (int)byte_off, 0, IA2P_WT_AUX);
}
// Experiment knobs (A/B runs of a structure or a heuristic constant) are read from the environment only in builds made with -DIA2P_EXPERIMENTS
// (IA2P_EXTRA_FLAGS=-DIA2P_EXPERIMENTS python -m instructany2pix_amd.build --force); the product library answers with the built-in default.
// The eight runtime switches the product does read are listed in DESIGN.md §4 (IA2P_LN_FOLD, IA2P_XATTN_FUSE, IA2P_PREFETCH, IA2P_WT, IA2P_TILE_GROUP,
// IA2P_SPLITK_INKERNEL, IA2P_TUNE_LOG, IA2P_TUNE_EXCLUDE).
static inline const char* ia2p_exp_env(const char* name) {
#ifdef IA2P_EXPERIMENTS
  return getenv(name);
#else
  if (getenv(name)) {      // an A/B script is setting a knob this build does not read: say so (a few times), instead of silently comparing A with A
    static thread_local int warned = 0;
    if (warned < 8) { ++warned; fprintf(stderr, "[ia2p] %s is set but this build ignores it (experiment knob: rebuild with IA2P_EXTRA_FLAGS=-DIA2P_EXPERIMENTS)\n", name); }
  }
  return nullptr;
#endif
}
// IA2P_WT: bit mask of the kernels that store write-through (1 GEMM C, 2 K-split reduce, 4 GroupNorm, 8 attention, 16 concat); A/B switch.
// Same-box A/B at batch 8 (tools/ab_vals.sh): GEMM C -0.14 ms per step, GroupNorm -0.06, reduce / concat -0.02 each; attention +0.33 while O left
// as 8-byte per-query pieces (partial lines: write-through pays for every one of them), -0.06 once O goes through LDS and leaves as whole lines.
static inline int ia2p_wt_mask() {
  static const int m = getenv("IA2P_WT") ? atoi(getenv("IA2P_WT")) : 31;
  return m;
}

// the folded-LayerNorm epilogue of one element, rstd * (acc - mean * colsum) + fbias, as TWO explicit FMAs: every site (GEMM epilogues, the K-split
// reduce, the fused to_q tile) must round identically -- left to the compiler, the contraction of the plain expression differed between two instantiations
__device__ __forceinline__ float ln_fold_f(float acc, float mean, float rstd, float colsum, float fbias) { return fmaf(rstd, fmaf(-mean, colsum, acc), fbias); }
// mean and 1 / sqrt(var + eps) of a row from its {sum, sum of squares}: every rounding spelled out (two products, one FMA) -- left as the plain expression
// `s2 * inv - mean * mean`, two instantiations of the GEMM template contracted it differently and their LayerNorm-folded outputs differed in the last bit
__device__ __forceinline__ float2 ln_mean_rstd_f(float s1, float s2, int K, float eps) {
  const float inv = 1.f / (float)K;
  const float mean = __fmul_rn(s1, inv);
  const float var = fmaxf(fmaf(s2, inv, -__fmul_rn(mean, mean)), 0.f);
  return make_float2(mean, rsqrtf(__fadd_rn(var, eps)));
}
__device__ __forceinline__ float act_f(float x, int act) { return act == 1 ? gelu_erf_f(x) : act == 2 ? quick_gelu_f(x) : act == 3 ? gelu_tanh_f(x) : x; }

// ---- launch descriptors shared by kernels and the host executor --------------------------------------

struct GemmArgs {
  const half_t* A;       // activations (row source)
  const half_t* W;       // weights [N, K] row-major (torch Linear layout; conv pre-packed [Co][ky][kx][Ci])
  half_t* C;             // output [M, ldc]
  const half_t* zero;    // >=16 B of zeros (out-of-range rows / conv padding read this)
  int M, N, K;           // K % 64 == 0, N % 4 == 0
  int lda, ldc, ldw;     // row strides of A, C and W (ldw >= K)
  // LINEAR row map: src_row = (m / rpb) * bstride + (m % rpb) + roff   (rpb == 0: identity)
  int rpb, bstride, roff;
  // CONV3x3 gather (implicit GEMM): m = (b, oy, ox); k = (ky, kx, ci)
  int Hs, Ws;            // stored source height/width
  int Ho, Wo;            // output height/width
  int stride, up;        // conv stride (1|2); up=1 reads a nearest-x2 upsampled view of the source
  int pad;               // zero rows/cols before the image (1; the VAE's stride-2 downsample pads only after: 0)
  int Cin;               // channels per tap (Cin % 64 == 0)
  // CONV3x3 with an appended 1x1 block (K = 9 Cin + Cin2): the last Cin2 columns of the contraction read the pixel itself from a SECOND
  // tensor -- a ResnetBlock2D's conv2(h) + conv_shortcut(x) as ONE implicit GEMM (stride 1, no upsampling); A2 == nullptr: plain 3x3
  const half_t* A2; int lda2, Cin2;
  const half_t* A3; int lda3, Cin3;   // a second appended block (K = 9 Cin + Cin2 + Cin3): the skip half of an up-block input that is never concatenated
  // epilogue
  const half_t* bias;    // [N] (GEGLU: packed order) or null
  const half_t* rowvec;  // per-batch vector added to every row of that batch (time embedding) or null
  int rowvec_ld, rows_per_batch;
  const half_t* residual;  // [M, ldr] or null (may alias C)
  int ldr;
  int geglu;             // 1: W/bias rows interleaved in 16-row (a,g) pairs; out[m, n/2] = a * gelu(g)
  const float* phi_lut;  // set by the launcher for GEGLU launches: the normal-CDF table of gelu_lut_f (device memory, IA2P_PHI_LUT_N float2)
  int m_fastest;         // tile order: 1 = consecutive blocks walk M (weights panel shared), 0 = walk N
  int vec8;              // set by the launcher: strides / bases allow 16-byte epilogue accesses
  int c_wt;              // set by the launcher: C leaves through write-through (sc1) stores, so the end-of-kernel write-back has nothing left to do
  int group_w;           // > 0: grouped tile order in column panels of this many tiles (set by the launcher; overrides m_fastest)
  // weight prefetch: extra workgroups (launched after the tiles) stream the NEXT contraction's weights once,
  // sequentially, so they are in the Infinity Cache / L2 instead of HBM-cold when that kernel starts
  const void* pf;        // or null
  long pf_bytes;
  int pf_blocks;
  // split-K: `splitk` workgroups per tile each own a contiguous K range and write fp32 slabs partial[s][M][N];
  // splitk_reduce_kernel then sums them in slab order (deterministic) and applies the epilogue
  int splitk;            // 0/1 = off
  float* partial;
#ifdef IA2P_CLOCK_STAMP
  unsigned long long* stamp;      // diagnostic builds only: the stamp records of a launch WITH a K split (its `partial` holds the slabs); unsplit launches pass the buffer in `partial`
#endif
  int* sk_counters;      // set by the launcher: per-tile ticket counters of the in-launch combine (null: a separate splitk_reduce_kernel launch finishes)
  // range extension (VAE executor): out = acc * acc_scale + (bias + rowvec) * bias_scale + residual; 0 = 1.0. The VAE keeps its residual
  // stream multiplied by a power of two < 1 so that fp16 storage does not overflow where the reference upcasts to fp32 (vae_engine.hip)
  float acc_scale, bias_scale;
  int act;               // activation on (acc + bias) before rowvec / residual: 0 none, 1 GELU (erf), 2 quick-GELU x*sigmoid(1.702x) (CLIP MLPs)
  // LayerNorm folded into this contraction (consumer side). A is the un-normalised residual stream [M, K], W was pre-scaled by
  // the norm's gamma when the weights were finalized (fold_ln_kernel), and the epilogue finishes the normalisation:
  //   out[m][n] = rstd_m * (acc[m][n] - mean_m * ln_cs[n]) + ln_bias[n]        ln_cs[n] = sum_k W'[n][k],  ln_bias = b + W.beta
  // mean_m / rstd_m come from ln_stats: per row and slot a float2 {sum x, sum x^2} over K features, written by the producer
  // of A (stats_out below); ln_slots partial sums per row are added up in slot order (deterministic).
  const float* ln_stats;
  int ln_slots;
  const float* ln_cs;
  const float* ln_bias;
  float ln_eps;
  // producer side: row statistics of THIS launch's fp16 output, for the folded LayerNorm of the next contraction.
  // stats_out[(slot * M + m) * 2 + {0, 1}], slot = tile_n (or 0 for a K-split launch: the reduce kernel writes it)
  float* stats_out;
  // ---- GroupNorm fused into a 3x3 convolution (round 5; reference: diffusers ResnetBlock2D `conv1(nonlinearity(norm1(x)))` / `conv2(dropout(nonlinearity(norm2(h))))` behind
  //      instructany2pix/ddim/pnp_pipeline.py:253-260, in-tree twin llm/model/vae/modules/blocks.py:122-142).
  // producer side: gn_out[(tile_m * N + n)] = {sum, sum of squares} (fp64) of THIS launch's fp16 output column n over the rows of M-tile tile_m (row tile m0 / BM of a
  // linear tile, patch index of a halo-staged tile: both hold HW / BM slots per image when BM divides the image). Written by the register epilogue and by the last
  // K slice of an in-launch K-split combine; nullptr: off.
  double* gn_out;
  // consumer side (halo-staged convolution only): the 3x3 operand is the RAW input of the GroupNorm, one tensor or two that are never concatenated (A: channels
  // [0, gn.C0), A1b: channels [C0, Cin) -- the up path's [hidden | skip]); the kernel folds the producers' column sums into the 32 group statistics of its image and
  // normalises + SiLUs every halo image in LDS before the taps read it (border pixels stay zero: the reference pads the ACTIVATED tensor).
  const half_t* A1b; int lda1b;
  struct GnIn {
    const double* st0; int rows0;      // column sums of A's producer, rows per slot (HW % rows0 == 0)
    const double* st1; int rows1;      // ... of A1b's producer (st1 == nullptr: one source)
    int C0;                            // channels of the first source (== Cin when there is one)
    const half_t* gamma; const half_t* beta;   // [Cin], concatenated channel order
    int gs;                            // channels per group (Cin / groups)
    int groups;
    float eps; int silu;
  } gn;                                // gn.st0 == nullptr: plain convolution
};

// buffer-load staging (linear layers, the halo-staged convolution) addresses an operand with a 31-bit byte offset: what a launch of `rows` x `ld` fp16 elements needs
static inline bool ia2p_fits_buffer(size_t rows, size_t ld) { return rows * ld * 2 < (size_t)0x7ffffe00; }
// what the halo-staged 3x3 convolution (conv_halo_f16_kernel, gemm_kernel.h) takes: stride 1, one pixel of zero padding, the source itself or its nearest-x2 upsampled view, whole 16 x 16 patches, whole blocks of 64 channels in every source
// what the GroupNorm-fused form takes on top of ia2p_conv_halo_ok: no upsampled view, whole blocks of 64 channels in both sources, whole slots per image, at most
// IA2P_GN_MAX_SLOTS slots per image and source (every workgroup folds them itself), at most 64 groups
constexpr int IA2P_GN_MAX_SLOTS = 16;
static inline bool ia2p_conv_gn_ok(const GemmArgs& a) {
  if (!a.gn.st0) return true;
  const int HW = a.Ho * a.Wo, C1 = a.Cin - a.gn.C0;
  return a.up == 0 && a.gn.silu && a.gn.gamma && a.gn.beta && a.gn.groups > 0 && a.gn.groups <= 64 && a.gn.gs > 0 && a.gn.gs * a.gn.groups == a.Cin && a.gn.gs <= 96 && a.gn.C0 > 0 && a.gn.C0 % 64 == 0 && C1 >= 0 && C1 % 64 == 0 &&
         a.gn.rows0 > 0 && HW % a.gn.rows0 == 0 && HW / a.gn.rows0 <= IA2P_GN_MAX_SLOTS && ((C1 == 0 && !a.gn.st1 && !a.A1b) || (C1 > 0 && a.gn.st1 && a.A1b && a.lda1b >= C1 && a.gn.rows1 > 0 && HW % a.gn.rows1 == 0 &&
         HW / a.gn.rows1 <= IA2P_GN_MAX_SLOTS && ia2p_fits_buffer(a.M, a.lda1b))) && a.Cin <= 48 * 1024 / 16;      // (the per-channel sums of an image meet in one 48 KiB halo-image buffer)
}
static inline bool ia2p_conv_halo_ok(const GemmArgs& a) {
  const int c2 = a.A2 ? a.Cin2 : 0, c3 = a.A3 ? a.Cin3 : 0;
  return a.stride == 1 && (a.up == 0 || (a.up == 1 && !a.A2)) && a.pad == 1 && (a.Hs << a.up) == a.Ho && (a.Ws << a.up) == a.Wo && a.Ho > 0 && a.Ho % 16 == 0 && a.Wo % 16 == 0 && a.Cin >= 64 && a.Cin % 64 == 0 && c2 % 64 == 0 &&
         c3 % 64 == 0 && a.Cin2 == c2 && (a.A3 == nullptr || a.A2 != nullptr) && a.K == 9 * a.Cin + c2 + c3 && a.M % (a.Ho * a.Wo) == 0 && !a.rpb && !a.geglu &&
         ia2p_fits_buffer(a.M, a.lda) && ia2p_fits_buffer(a.N, a.ldw) && (!a.A2 || ia2p_fits_buffer(a.M, a.lda2)) && (!a.A3 || ia2p_fits_buffer(a.M, a.lda3));
}

struct GemmPlan { int variant; int splitk; int gn = 0; };      // gn = 1 (measured plans of 3x3 sites behind a GroupNorm only): the norm runs INSIDE the halo-staged convolution (conv_halo_kernel.h GN = 1)
// tile variants of gemm_f16_kernel (id = index): {BM, BN, LDS ring stages}; 4 waves (2 x 2), BK = 64
static inline bool ia2p_tile_geglu_only(int pp) { return pp == 4; }
// what the 256 x 320 GEGLU tile (variant 27, gemm_geglu_kernel.h) takes: GEGLU launches of linear layers in whole tiles
static inline bool ia2p_geglu320_shape_ok(int M, int N, int K, bool conv, bool geglu) { return geglu && !conv && M > 0 && M % 256 == 0 && N > 0 && N % 320 == 0 && K >= 64 && K % 64 == 0; }
struct GemmTile { int bm, bn, stages, pp, halo; };      // pp = 1: 8-wave ping-pong schedule (one workgroup per CU); pp = 2: 8-wave 8-phase schedule; pp = 4: ping-pong on 32-deep sub-steps, GEGLU launches of linear layers only (gemm_geglu_kernel.h); halo = 1: halo-staged 3x3 convolution only (conv_halo_f16_kernel)
constexpr int IA2P_GEMM_NVARIANT = 28;
constexpr GemmTile IA2P_GEMM_TILES[IA2P_GEMM_NVARIANT] = {{128, 128, 2, 0, 0}, {128, 128, 3, 0, 0}, {128, 64, 2, 0, 0}, {128, 64, 3, 0, 0}, {64, 64, 2, 0, 0}, {64, 64, 3, 0, 0},
                                                          {64, 160, 2, 0, 0}, {64, 160, 3, 0, 0}, {128, 160, 2, 0, 0}, {128, 160, 3, 0, 0}, {160, 128, 2, 0, 0}, {160, 160, 2, 0, 0},
                                                          {256, 128, 3, 1, 0},                               // 12: the ping-pong tile
                                                          {64, 64, 4, 0, 0}, {64, 64, 6, 0, 0}, {128, 64, 4, 0, 0},   // 13..15: deep rings for latency-bound launches (few workgroups, e.g. batch 1)
                                                          {128, 80, 2, 0, 0}, {128, 80, 4, 0, 0},               // 16..17: 4 x 1 waves; N = 1280 / 640 problems in exactly 256 / 512 tiles
                                                          {256, 160, 3, 1, 0},                               // 18: ping-pong, 160 wide (FF-in 2048 x 10240 in exactly 512 tiles = two rounds of one per CU; 0.72 x the LDS fill per flop of 128 x 160)
                                                          {128, 160, 3, 1, 0},                               // 19: ping-pong over 128 rows (two wave groups of 64 rows): N = 1280 / 640 problems at M = 2048 / 8192 in 128 / 256 tiles of one per CU
                                                          {32, 64, 3, 0, 0}, {32, 128, 3, 0, 0},                // 20..21: 32-row tiles (16-row wave tiles) for launches that leave CUs empty (batch 1: M = 256): a lone workgroup takes its
                                                                                                          // operands in at ~27 B/clk whatever its loop looks like (profiles/r03k_small_m_kloop.txt), so more, smaller workgroups win
                                                          {256, 256, 2, 2, 0}, {256, 128, 2, 2, 0},             // 22..23: 8-phase schedule (pp = 2): 2 x 4 waves of 128 x 64 (128 x 32), two k-tile buffers, one workgroup per CU, 128 flop per staged byte
                                                          {256, 160, 3, 1, 1}, {256, 128, 3, 1, 1},      // 24..25: halo-staged 3x3 convolution (16 x 16 pixel patches; an ineligible site runs variant 18 / 12 instead)
                                                          {256, 80, 3, 1, 1},                             // 26: the same, 80 wide (8 x 1 waves of 32 x 80; an ineligible site runs variant 16)
                                                          {256, 320, 2, 4, 0}};                           // 27 (round 6): the GEGLU projection in ONE round of 256 tiles (2048 x 10240): 4 x 2 waves of 64 x 160, two k-tile slots, ping-pong on 32-deep sub-steps; GEGLU launches only
GemmPlan ia2p_gemm_plan(int M, int N, int K, bool conv, bool geglu);
bool ia2p_plan_lookup(int M, int N, int K, bool conv, bool geglu, GemmPlan* out);     // measured plan table (ia2p_autotune)
void ia2p_plan_set(int M, int N, int K, bool conv, bool geglu, GemmPlan pl);

struct AttnSeg {
  const half_t* K;       // key rows:   K + ((b * rows_per_batch + r) * ld) + head*64
  const half_t* V;
  int nkeys, ld, rows_per_batch;
  float weight;          // out += weight * softmax(QK^T) V   (IP-Adapter: text 1.0, image tokens `scale`)
};

struct AttnArgs {
  const half_t* Q;       // Q + ((b * Nq + q) * ldq) + head*64
  half_t* O;             // same indexing with ldo
  int ldq, ldo, B, heads, Nq, nseg;
  float scale_log2e;     // (1/sqrt(64)) * log2(e)
  int xcd_map;           // 1: contiguous (batch, head, query block) range per XCD (set by the launcher)
  AttnSeg seg[2];
  const float* w1_b;     // optional, device: weight of segment 1 PER BATCH ELEMENT (overrides seg[1].weight) -- requests with different IP-Adapter
                         // scales (reference ip_adapter.py:211-214 `set_scale`, one value per call there) share one evaluation
};
