// GroupNorm from producer-side column sums: the pieces every consumer shares (the GroupNorm-fused halo-staged convolution, conv_halo_kernel.h; the stand-alone
// apply pass and the canonical statistics kernel, norm.hip), so that all of them derive the SAME bits from the same sums.
//
// Reference: torch.nn.GroupNorm(32, C, eps) (+ SiLU) in front of the 3x3 convolutions of diffusers' ResnetBlock2D (behind instructany2pix/ddim/pnp_pipeline.py:253-260;
// in-tree twin llm/model/vae/modules/blocks.py:122-142 `h = self.conv1(nonlinearity(self.norm1(x)))`).
//
// Statistics, canonical form (independent of the tile that produced them and of the thread count that folds them):
//   * a producer writes, per SLOT of `rows` consecutive output rows (one M-tile: HW / rows slots per image) and per channel, {sum x, sum x^2} in fp64, built from
//     fp32 sums over aligned runs of 16 pixels taken in pixel order (gn_seg16: the unit every tile shape -- 256-row linear tiles, 16 x 16 patches -- is made of);
//   * a consumer folds, per channel, the slots of its image in slot order, then per group the channels in channel order, all in fp64 (gn_channel_sums,
//     gn_group_stats), and turns {mean, rstd} into one fp32 scale / shift pair per channel (gn_scale_shift);
//   * an element is normalised as fp16(silu(fma(x, a, b))) (gn_apply_f): one rounding, the same instruction sequence everywhere.
#pragma once
#include "common.h"

// {sum, sum of squares} of 16 values in fp32, taken in order (the canonical unit of the producer-side statistics)
__device__ __forceinline__ void gn_seg16_add(float f, float& s, float& q) { s = __fadd_rn(s, f); q = fmaf(f, f, q); }

// per channel c of the (concatenated) input: sum of its slots in slot order -> chs[c] (LDS, double2 per channel). All NT threads; the caller puts a barrier behind it.
__device__ __forceinline__ void gn_channel_sums(const GemmArgs::GnIn& g, int Cin, int HW, int img, int tid, int nthreads, double2* chs) {
  const int C1 = Cin - g.C0;
  const int T0 = HW / g.rows0, T1 = C1 > 0 ? HW / g.rows1 : 0;
  for (int c = tid; c < Cin; c += nthreads) {
    const bool second = c >= g.C0;
    const int cl = second ? c - g.C0 : c, ld = second ? C1 : g.C0, T = second ? T1 : T0;
    const double2* src = (const double2*)(second ? g.st1 : g.st0) + (size_t)img * T * ld + cl;
    double a = 0.0, q = 0.0;
    for (int t0 = 0; t0 < T; t0 += 16) {      // (up to IA2P_GN_MAX_SLOTS slots: ONE round trip per channel -- a workgroup of the fused convolution does this fold in its prologue, alone on its CU)
      double2 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = src[(size_t)min(t0 + u, T - 1) * ld];      // all loads of a round in flight (clamped, never branched around)
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (t0 + u < T) { a += v[u].x; q += v[u].y; }
    }
    chs[c] = make_double2(a, q);
  }
}
// thread j < groups: its channels in channel order -> gstat[j] = {mean, rstd}. The caller puts a barrier in front (chs complete) and behind (gstat complete).
__device__ __forceinline__ void gn_group_stats(const GemmArgs::GnIn& g, int HW, int tid, const double2* chs, float2* gstat) {
  if (tid < g.groups) {
    double a = 0.0, q = 0.0;
    for (int c = tid * g.gs; c < (tid + 1) * g.gs; ++c) { const double2 v = chs[c]; a += v.x; q += v.y; }
    const double n = (double)HW * g.gs;
    const double mean = a / n;
    double var = fma(-mean, mean, q / n);      // (explicit: left to the compiler, two instantiations may contract `q / n - mean * mean` differently)
    if (var < 0.0) var = 0.0;
    gstat[tid] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)g.eps)));
  }
}
// scale / shift of one channel: y = x * a + b  with  a = rstd * gamma,  b = beta - mean * a   (every rounding spelled out)
__device__ __forceinline__ float2 gn_scale_shift(float2 mean_rstd, float gamma, float beta) {
  const float a = __fmul_rn(mean_rstd.y, gamma);
  return make_float2(a, fmaf(-mean_rstd.x, a, beta));
}
// the normalisation of one element, y = fma(x, a, b), and its SiLU y / (1 + exp(-y)) as FOUR primitive steps -- so that a kernel may spread them between other work
// (the fused convolution places them one by one between its MFMAs) and still produce the bits of gn_apply_f:
//   y = gn_step_y(x, a, b);  e = gn_step_e(y) = 2^(y * -log2 e);  r = gn_step_r(e) = 1 / (1 + e);  out = gn_step_o(y, r) = y * r
__device__ __forceinline__ float gn_step_y(float x, float a, float b) { return fmaf(x, a, b); }
__device__ __forceinline__ float gn_step_e(float y) { return __builtin_amdgcn_exp2f(__fmul_rn(y, -1.4426950408889634f)); }
__device__ __forceinline__ float gn_step_r(float e) { return __builtin_amdgcn_rcpf(__fadd_rn(1.0f, e)); }
__device__ __forceinline__ float gn_step_o(float y, float r) { return __fmul_rn(y, r); }
__device__ __forceinline__ float gn_apply_f(float x, float a, float b, bool silu) {
  const float y = gn_step_y(x, a, b);
  return silu ? gn_step_o(y, gn_step_r(gn_step_e(y))) : y;
}
