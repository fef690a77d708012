// Fused `to_q` projection + cross-attention for gfx950.
//
// Reference: attention_processor.py:344 (`query = attn.to_q(hidden_states)`), :358-359 / :379-380 (context K / V, projected once per step by the
// executor), :371 and :387 (the two scaled_dot_product_attention calls of IPAttnProcessor2_0) and :397 (`hidden_states + scale * ip_hidden_states`);
// AttnProcessor2_0 :239 / :259 when no adapter is installed.
//
// A 128 x 64 tile of the to_q GEMM is exactly 128 queries x ONE head, and a cross-attention head needs nothing else from Q: the tile goes from
// the MFMA accumulators through LDS into the Q^T fragments of the attention core (attention_core.h) and the workgroup walks the 77 (+4) context
// keys right there. One launch instead of two per layer (70 per UNet evaluation), and Q never travels to HBM and back. Same arithmetic in the
// same order as the two stand-alone kernels (projection epilogue in fp32, ONE rounding to fp16, the attention core unchanged): bit-identical
// to running `gemm_f16_kernel<128, 64, ...>` + `attention_f16_kernel` (tests/test_ops_gpu.py).
#include "gemm_kernel.h"

#ifndef IA2P_QX_STAGES
#define IA2P_QX_STAGES 2      // LDS ring depth of the projection loop (build-time knob for A/B builds)
#endif
template <int MODE>      // attention core mode: 0 one key segment; 1 text + <= 64 image-token keys; 2 generic two segments
// (the fused tiles take no row map and no K split: the four preloaded argument dwords those would use carry the folded LayerNorm's statistics pointer and slot count --
//  read from `p`, a cold scalar load of the argument block stood between the workgroup's entry and its statistics loads; gemm_geglu_kernel.h.
//  14 dwords are preloaded: four pointers, five ints, {slots, tile order} packed)
__global__ __launch_bounds__(256, 2) void qproj_xattn_kernel(const half_t* hA, const half_t* hW, const half_t* hzero, const float* h_ln_stats, int hM, int hN, int hK, int hlda, int hldw,
                                                             int h_slots_gw, const GemmArgs p, const AttnArgs xa) {
  gemm_tile_body<128, 64, IA2P_QX_STAGES, false, 2, 64, 0, 2, MODE + 1>(hA, hW, hzero, hM, hN, hK, hlda, hldw, 0, 0, 0, 0, IA2P_SKGW_GW(h_slots_gw), IA2P_SKGW_FLAGS(h_slots_gw), p, &xa, h_ln_stats,
                                                                        IA2P_SKGW_LO8(h_slots_gw));
}

// Q = epilogue(A . W^T) is [B * Nq, heads * 64]; x.Q / x.ldq are ignored (Q stays on chip). Requires Nq % 128 == 0 (a tile must not straddle
// two batch elements), N = heads * 64, no K-split, no GEGLU / residual / row vector / statistics output; the context must fit
// ATTN_PRE_TILES key tiles of 64 (77 text + 4 image tokens: 2 + 1).
bool ia2p_qproj_xattn_ok(const GemmArgs& a, const AttnArgs& x) {
  return x.Nq > 0 && x.Nq % 128 == 0 && a.M == x.B * x.Nq && a.N == x.heads * 64 && a.K >= 64 && a.K % 64 == 0 && a.splitk <= 1 && !a.rpb && !a.geglu && !a.residual && !a.rowvec &&
         !a.stats_out && !a.act && x.nseg >= 1 && x.nseg <= 2 && x.seg[0].nkeys > 0 && (x.nseg == 1 || x.seg[1].nkeys > 0) &&
         (!a.bias || ((((uintptr_t)a.bias) & 15) == 0 && a.N % 8 == 0)) && x.ldo % 8 == 0 && ((((uintptr_t)x.O) & 15) == 0) &&
         ia2p_fits_buffer(a.rpb ? ((size_t)a.M / a.rpb + 1) * (size_t)(a.bstride > 0 ? a.bstride : 0) + a.roff + a.rpb : (size_t)a.M, a.lda) && ia2p_fits_buffer(a.N, a.ldw) &&
         attn_kv_resident(x);      // short contexts only: their K / V ride in registers through the projection loop
}

hipError_t ia2p_launch_qproj_xattn(const GemmArgs& a, const AttnArgs& x, hipStream_t s) {
  if (!ia2p_qproj_xattn_ok(a, x)) return hipErrorInvalidValue;
  constexpr int BM = 128, BN = 64, SMEM = EpiCfg<128, 64, IA2P_QX_STAGES, 2, 64, 2>::SMEM > 65536 ? EpiCfg<128, 64, IA2P_QX_STAGES, 2, 64, 2>::SMEM : 65536;       // K / V images of the attention core: 2 x 32 KiB (the GEMM ring needs 48 KiB)
  static_assert(EpiCfg<BM, BN, IA2P_QX_STAGES, 2, 64, 2>::SMEM <= SMEM, "LDS budget");
  static bool attr_set[64] = {false};                  // per device (the attribute is)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)qproj_xattn_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)qproj_xattn_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)qproj_xattn_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int tiles_n = a.N / BN, tiles = (a.M / BM) * tiles_n;
  GemmArgs b = a;
  b.vec8 = 1; b.splitk = 0; b.sk_counters = nullptr;
#ifndef IA2P_CLOCK_STAMP
  b.partial = nullptr;
#endif
  b.group_w = ia2p_tile_group_w(tiles, tiles_n, SMEM, BM, BN);
  AttnArgs y = x;
  y.xcd_map = (((ia2p_wt_mask() & 8) && (size_t)x.B * x.Nq * x.ldo * 2 < (size_t)0x7ffffff0) ? 2 : 0) | (ia2p_attn_fold_enabled() ? 0 : 4);      // bit 1: write-through O; bit 2: image-token keys NOT folded into the last text tile
  const int extra = (a.pf && a.pf_bytes >= 4096) ? a.pf_blocks : 0;
  int slots_gw;
  if (tiles + extra >= (1 << 21) || !ia2p_pack_skgw(b.ln_stats ? b.ln_slots : 0, b.group_w, b.m_fastest, b.ln_stats != nullptr, &slots_gw)) return hipErrorInvalidValue;      // (udiv_small in the tile decode: counts below 2^21)
  const int mode = y.nseg == 1 ? 0 : (y.seg[1].nkeys <= 64 && y.seg[0].weight != 0.f) ? 1 : 2;     // as ia2p_launch_attention
#define IA2P_QX_LAUNCH(MODE)                                                                                                                   \
  hipLaunchKernelGGL(qproj_xattn_kernel<MODE>, dim3(tiles + extra), dim3(256), SMEM, s, b.A, b.W, b.zero, b.ln_stats, b.M, b.N, b.K, b.lda, b.ldw, slots_gw, b, y)
  if (mode == 0) IA2P_QX_LAUNCH(0);
  else if (mode == 1) IA2P_QX_LAUNCH(1);
  else IA2P_QX_LAUNCH(2);
#undef IA2P_QX_LAUNCH
  return hipGetLastError();
}


// ---- fused QKV projection + self-attention (round 4) -----------------------------------------------------------------------------------------------------
// Reference: AttnProcessor2_0 (attention_processor.py:239 to_q, :246-247 to_k / to_v, :259 scaled_dot_product_attention), with the LayerNorm in front folded
// into the projection. At the 16 x 16 level an image has 256 tokens: ONE 256 x 192 tile of the stacked projection = Q | K | V of all of an image's tokens for
// ONE head = everything that head's attention needs. The tile body (gemm_kernel.h, XA = 4) turns its accumulators into the K / V images and Q fragments of the
// attention core and writes only O: no QKV tensor (15.7 MB per layer at batch 8), no second launch. Same arithmetic in the same order as the stand-alone
// pair: bit-identical to `ia2p_gemm_ex` + `ia2p_attention` (tests/test_ops_gpu.py).
__global__ __launch_bounds__(512, 2) void qkv_sattn_kernel(const half_t* hA, const half_t* hW, const half_t* hzero, const float* h_ln_stats, int hM, int hN, int hK, int hlda, int hldw,
                                                           int h_slots_gw, const GemmArgs p, const AttnArgs xa) {
#ifndef IA2P_SATTN_PP
#define IA2P_SATTN_PP 3      // k-loop schedule of the fused QKV + self-attention tile: 3 = two-slot ping-pong, 0 = plain loop (one barrier per k-step); A/B builds
#endif
  gemm_tile_body<256, 192, 2, false, 4, 64, IA2P_SATTN_PP, 2, 4>(hA, hW, hzero, hM, hN, hK, hlda, hldw, 0, 0, 0, 0, IA2P_SKGW_GW(h_slots_gw), IA2P_SKGW_FLAGS(h_slots_gw), p, &xa, h_ln_stats, IA2P_SKGW_LO8(h_slots_gw));
}

// A [B * 256, K] (un-normalised rows with a.ln_*, or plain), W the stacked [3 * heads * 64, K] projection; x: O / ldo / B / heads / Nq = 256.
bool ia2p_qkv_sattn_ok(const GemmArgs& a, const AttnArgs& x) {
  return x.Nq == 256 && x.B > 0 && a.M == x.B * 256 && a.N == 3 * x.heads * 64 && a.K >= 64 && a.K % 64 == 0 && a.splitk <= 1 && !a.geglu && !a.residual && !a.rowvec &&
         !a.stats_out && !a.act && !a.rpb && x.nseg == 1 && x.ldo % 8 == 0 && ((((uintptr_t)x.O) & 15) == 0) && (!a.bias || ((((uintptr_t)a.bias) & 7) == 0)) &&
         (!a.ln_stats || (((((uintptr_t)a.ln_cs) | ((uintptr_t)a.ln_bias)) & 15) == 0)) && ia2p_fits_buffer(a.M, a.lda) && ia2p_fits_buffer(a.N, a.ldw);
}

hipError_t ia2p_launch_qkv_sattn(const GemmArgs& a, const AttnArgs& x, hipStream_t s) {
  if (!ia2p_qkv_sattn_ok(a, x)) return hipErrorInvalidValue;
  constexpr int SMEM = EpiCfg<256, 192, 2, 4, 64, 2, 0>::SMEM;      // (the epilogue configuration does not depend on the k-loop schedule)
  static_assert(SMEM >= 96 * 1024 + (2 * 256 + 2 * 192) * 4, "K / V / Q images + row and column constants");
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)qkv_sattn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int tiles_n = x.heads, tiles = x.B * tiles_n;
  GemmArgs b = a;
  b.vec8 = 1; b.splitk = 0; b.sk_counters = nullptr;
#ifndef IA2P_CLOCK_STAMP
  b.partial = nullptr;      // (diagnostic builds: the field carries the stamp buffer, tools/insitu_stamps.py)
#endif
  b.m_fastest = 0;
  b.group_w = ia2p_tile_group_w(tiles, tiles_n, SMEM, 256, 192);
  AttnArgs y = x;
  y.seg[0].nkeys = 256; y.seg[0].weight = 1.f; y.nseg = 1;
  y.xcd_map = ((ia2p_wt_mask() & 8) && (size_t)x.B * x.Nq * x.ldo * 2 < (size_t)0x7ffffff0) ? 2 : 0;      // bit 1: write-through O
  const int extra = (a.pf && a.pf_bytes >= 4096) ? a.pf_blocks : 0;
  int slots_gw;
  if (tiles + extra >= (1 << 21) || !ia2p_pack_skgw(b.ln_stats ? b.ln_slots : 0, b.group_w, b.m_fastest, b.ln_stats != nullptr, &slots_gw)) return hipErrorInvalidValue;      // (udiv_small in the tile decode: counts below 2^21)
  hipLaunchKernelGGL(qkv_sattn_kernel, dim3(tiles + extra), dim3(512), SMEM, s, b.A, b.W, b.zero, b.ln_stats, b.M, b.N, b.K, b.lda, b.ldw, slots_gw, b, y);
  return hipGetLastError();
}
