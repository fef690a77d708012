// Small kernels of the denoise step on gfx950: timestep / micro-conditioning embeddings, skinny (M<=16)
// linears for the embedding MLPs, the 4-channel latent convolutions at the NCHW boundary, channel concat,
// the fused CFG + DDIM update, and the one-time weight re-layouts.
#include "common.h"

#include <algorithm>

// ---- sinusoidal embeddings (diffusers Timesteps(flip_sin_to_cos=True, freq_shift=0)) ------------------
// tsin[b, :]   = [cos(t f_i), sin(t f_i)], i < Tp/2
// addin[b, :]  = [text_embeds[b, :P], sinusoid(time_ids[b,0]), ..., sinusoid(time_ids[b,5])]
// ts (optional, device [B]): one timestep per batch element (diffusers' UNet takes a [B] timestep tensor; requests at different steps of
// their schedules then share one evaluation); null: the scalar t for all
__global__ void embed_kernel(float t0, const float* ts, const half_t* text_embeds, const half_t* time_ids, half_t* tsin, half_t* addin,
                             int B, int Tp, int P, int Ad, int nids) {
  const int b = blockIdx.x;
  const float t = ts ? ts[b] : t0;
  const int Ain = P + nids * Ad;
  for (int i = threadIdx.x; i < Tp; i += blockDim.x) {
    const int half_ = Tp / 2, k = i % half_;
    const float f = expf(-9.210340371976184f * (float)k / (float)half_);
    const float a = t * f;
    tsin[(size_t)b * Tp + i] = (half_t)(i < half_ ? cosf(a) : sinf(a));
  }
  for (int i = threadIdx.x; i < Ain; i += blockDim.x) {
    half_t v;
    if (i < P) v = text_embeds[(size_t)b * P + i];
    else {
      const int j = (i - P) / Ad, r = (i - P) % Ad, half_ = Ad / 2, k = r % half_;
      const float id = (float)time_ids[(size_t)b * nids + j];
      const float a = id * expf(-9.210340371976184f * (float)k / (float)half_);
      v = (half_t)(r < half_ ? cosf(a) : sinf(a));
    }
    addin[(size_t)b * Ain + i] = v;
  }
}

// ---- out[M<=16, N] = act_out( f_in(X)[M,K] . W[N,K]^T + bias + addend ), one wave per 4 output columns ---------------
__global__ __launch_bounds__(64) void linear_small_kernel(const half_t* X, int ldx, const half_t* W, const half_t* bias,
                                                           const half_t* addend, int ldadd, half_t* out, int ldo,
                                                           int M, int N, int K, int silu_in, int silu_out) {
  const int lane = threadIdx.x;
  const int n0 = blockIdx.x * 4;            // one wave per workgroup: N / 4 workgroups spread over the CUs
  if (n0 >= N) return;
  float acc[4][16];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[c][m] = 0.f;
  const int nvec = K >> 3;
  for (int v = lane; v < nvec; v += 64) {
    float w[4][8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = min(n0 + c, N - 1);
      const h8 d = *(const h8*)(W + (size_t)n * K + v * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) w[c][e] = (float)d[e];
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (m < M) {
        const h8 d = *(const h8*)(X + (size_t)m * ldx + v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = (float)d[e];
          if (silu_in) x = silu_f(x);
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[c][m] += x * w[c][e];
        }
      }
    }
  }
  // 64 partial sums per lane -> lane l keeps the total of acc[l / 16][l % 16]: butterfly reduce-scatter, 63 shuffles instead of 64 x 6
  float v[64];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int m = 0; m < 16; ++m) v[c * 16 + m] = acc[c][m];
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    const bool up = (lane & s) != 0;
#pragma unroll
    for (int i = 0; i < s; ++i) {
      const float give = up ? v[i] : v[i + s], keep = up ? v[i + s] : v[i];
      v[i] = keep + __shfl_xor(give, s);
    }
  }
  {
    const int c = lane >> 4, m = lane & 15;
    if (m < M && n0 + c < N) {
      float r = v[0];
      if (bias) r += (float)bias[n0 + c];
      if (addend) r += (float)addend[(size_t)m * ldadd + n0 + c];
      if (silu_out) r = silu_f(r);
      out[(size_t)m * ldo + n0 + c] = (half_t)r;
    }
  }
}

// ---- conv_in: latent NCHW [B,Cin,H,W] -> channels-last [B*H*W, Co], 3x3 pad 1 (Cin*9 <= 64 taps) --------------------
// MFMA form: a wave owns 16 consecutive pixels. K = Cin*9 <= 64 (two 32-deep MFMA steps, zero padded); the weights are staged once per
// workgroup as a zero-padded [Co][64] LDS image (first operand: a lane ends up with 4 consecutive output channels of one pixel), the
// im2col fragment of the 16 pixels is gathered straight from the NCHW input into registers, and the 16 x Co output block goes through LDS
// so that every pixel row leaves as whole 16-byte pieces. HBM-bound on the output write (B*H*W*Co*2 bytes).
__global__ __launch_bounds__(256) void conv_in_kernel(const half_t* x, const half_t* w /*[Co][64]: k = ci*9 + tap, zero padded (ia2p_launch_pack_conv_in)*/, const half_t* bias,
                                                      half_t* y, int B, int Cin, int H, int W, int Co, int pix_per_block, float out_scale) {
  extern __shared__ __attribute__((aligned(16))) char cin_smem[];
  half_t* wl = (half_t*)cin_smem;                       // [Co][64], k = ci*9 + tap, zero beyond Cin*9
  half_t* ol = wl + (size_t)Co * 64;                    // [4 waves][16 pixels][Co] output staging
  const int KT = Cin * 9;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int npix = B * H * W;
  const int p0 = (blockIdx.x * 4 + wave) * 16;
  const int pl = lane & 15, kq = lane >> 4;
  // im2col fragment: lane holds X[k = 32*s + 8*kq + j][pixel pl]; gathered FIRST so that its loads fly while the weights are staged
  h8 xf[2];
  {
    const int pp = min(p0 + pl, npix - 1);
    const int b = pp / (H * W), rem = pp - b * H * W, oy = rem / W, ox = rem - oy * W;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = s2 * 32 + kq * 8 + j;
        const int ci = k / 9, tap = k - ci * 9, ky = tap / 3, kx = tap - ky * 3;
        const int iy = oy + ky - 1, ix = ox + kx - 1;
        const bool ok = k < KT && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        const half_t v = x[(((size_t)b * Cin + min(ci, Cin - 1)) * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)];
        xf[s2][j] = ok ? v : (half_t)0.f;
      }
  }
  {
    h8 wreg[8];                                 // weight image: loads of up to 8 pieces per thread in flight, then the LDS writes
    for (int i0 = tid; i0 < Co * 8; i0 += 256 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) wreg[u] = *(const h8*)(w + (size_t)min(i0 + u * 256, Co * 8 - 1) * 8);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * 256 < Co * 8) *(h8*)(wl + (size_t)(i0 + u * 256) * 8) = wreg[u];
    }
  }
  __syncthreads();
  half_t* ow = ol + (size_t)wave * 16 * Co;
  for (int n0 = 0; n0 < Co; n0 += 16) {
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const int nrow = min(n0 + pl, Co - 1);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const h8 wf = *(const h8*)(wl + (size_t)nrow * 64 + s2 * 32 + kq * 8);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[s2], acc, 0, 0, 0);      // D[row = channel n0 + 4*kq + i][col = pixel pl]
    }
    const int n = n0 + kq * 4;
    if (n < Co) {
      const h4 bv = *(const h4*)(bias + n);
      h4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (half_t)((acc[i] + (float)bv[i]) * out_scale);
      *(h4*)(ow + (size_t)pl * Co + n) = o;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): this wave's staging writes are visible to itself
  __builtin_amdgcn_wave_barrier();
  const int vec = Co >> 3;                    // 16-byte pieces per pixel row
  for (int i = lane; i < 16 * vec; i += 64) {
    const int r = i / vec, c8 = i - r * vec;
    if (p0 + r < npix) *(h8*)(y + (size_t)(p0 + r) * Co + c8 * 8) = *(const h8*)(ow + (size_t)r * Co + c8 * 8);
  }
}

// ---- conv_out: channels-last [B*H*W, C] -> NCHW [B,Co<=8,H,W], 3x3 pad 1 ---------------------------------------------------------
// MFMA form: a wave owns 16 consecutive pixels and accumulates a 16 x 16 output block of which Co <= 8 columns are real (lanes of the
// other columns feed zeros). Per tap and 32-channel chunk a lane loads 8 channels of its pixel's neighbour (16 B, zero outside the
// image) and the matching 8 weights from the LDS image of [Co][9][C]; the 3 x 3 neighbourhoods of adjacent pixels overlap, so the 9-fold
// re-read of the input is served by L1 / L2. Reads x once from HBM (B*H*W*C*2 bytes), writes 2*Co bytes per pixel.
template <int NC>        // 32-channel chunks per tap (C = 32 NC): all NC loads of a tap are in flight before its MFMAs; 0 = any C
__global__ __launch_bounds__(256) void conv_out_kernel(const half_t* x, int ldx, const half_t* w /*[Co][9][C]*/, const half_t* bias,
                                                       half_t* y, int B, int C, int H, int W, int Co) {
  extern __shared__ __attribute__((aligned(16))) char cout_smem[];
  half_t* wl = (half_t*)cout_smem;                      // [Co][9][C]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int npix = B * H * W;
  const int p0 = (blockIdx.x * 4 + wave) * 16;
  const int pl = lane & 15, kq = lane >> 4;
  const int pp = min(p0 + pl, npix - 1);
  const int b = pp / (H * W), rem = pp - b * H * W, oy = rem / W, ox = rem - oy * W;
  const int ncol = min(pl, Co - 1);
  const bool wreal = pl < Co;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const h8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  auto tap_ptr = [&](int tap, bool& ok) -> const half_t* {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = oy + ky - 1, ix = ox + kx - 1;
    ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
    return x + ((size_t)(b * H + min(max(iy, 0), H - 1)) * W + min(max(ix, 0), W - 1)) * ldx + kq * 8;
  };
  auto stage_weights = [&]() {
    constexpr int U = 4;                                // pieces per thread in flight
    const int n = (Co * 9 * C) >> 3;
    for (int i0 = tid; i0 < n; i0 += 256 * U) {
      h8 r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) r[u] = *(const h8*)(w + (size_t)min(i0 + u * 256, n - 1) * 8);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (i0 + u * 256 < n) *(h8*)(wl + (size_t)(i0 + u * 256) * 8) = r[u];
    }
    __syncthreads();
  };
  if constexpr (NC > 0) {
    // A kernel ROW of taps (3 x NC 16-byte loads per lane) is in flight at once, the first row already while the weights are staged: a wave's
    // chain is 3 load round trips instead of 9 (the kernel is latency-bound: 2 workgroups per CU, every wave waiting on its own loads).
    constexpr int TB = NC <= 10 ? 3 : 1;               // taps per batch (registers: TB * NC * 4)
    h8 a[TB][NC];
    bool ok[TB];
    auto load_batch = [&](int t0) {
#pragma unroll
      for (int tt = 0; tt < TB; ++tt) {
        const half_t* xp = tap_ptr(t0 + tt, ok[tt]);
#pragma unroll
        for (int u = 0; u < NC; ++u) a[tt][u] = *(const h8*)(xp + u * 32);
      }
    };
    load_batch(0);
    stage_weights();
#pragma unroll 1
    for (int t0 = 0; t0 < 9; t0 += TB) {
#pragma unroll
      for (int tt = 0; tt < TB; ++tt) {
        const half_t* wp = wl + ((size_t)ncol * 9 + t0 + tt) * C + kq * 8;
        h8 wv[NC];
#pragma unroll
        for (int u = 0; u < NC; ++u) wv[u] = *(const h8*)(wp + u * 32);
#pragma unroll
        for (int u = 0; u < NC; ++u) {
          h8 av = a[tt][u];
          if (!ok[tt]) av = zero8;
          if (!wreal) wv[u] = zero8;
          acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, wv[u], acc, 0, 0, 0);      // D[row = pixel 4*kq + i][col = output channel pl]
        }
      }
      if (t0 + TB < 9) load_batch(t0 + TB);
    }
  } else {
    stage_weights();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      bool ok;
      const half_t* xp = tap_ptr(tap, ok);
      const half_t* wp = wl + ((size_t)ncol * 9 + tap) * C + kq * 8;
#pragma unroll 4
      for (int c0 = 0; c0 < C; c0 += 32) {
        h8 a = *(const h8*)(xp + c0);
        h8 wv = *(const h8*)(wp + c0);
        if (!ok) a = zero8;
        if (!wreal) wv = zero8;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, wv, acc, 0, 0, 0);
      }
    }
  }
  if (p0 >= npix) return;
  if (wreal) {
    const float bs = (float)bias[pl];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = p0 + kq * 4 + i;
      if (q < npix) {
        const int qb = q / (H * W), qr = q - qb * H * W;
        y[((size_t)qb * Co + pl) * H * W + qr] = (half_t)(acc[i] + bs);
      }
    }
  }
}

// ---- channel concat of two channels-last tensors (torch.cat([h, skip], dim=1) of the up path) --------------------------
__global__ void concat_kernel(const half_t* a, int lda, int Ca, const half_t* b, int ldb, int Cb, half_t* y, long M, int wt) {
  const int va = Ca >> 3, vb = Cb >> 3, vt = va + vb;
  const long total = M * vt;
  const __amdgpu_buffer_rsrc_t y_rsrc = wt_rsrc((void*)y, (size_t)M * (Ca + Cb) * 2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long m = i / vt;
    const int v = (int)(i - m * vt);
    const uint4 d = v < va ? *(const uint4*)(a + m * lda + v * 8) : *(const uint4*)(b + m * ldb + (v - va) * 8);
    if (wt) store16_wt(y_rsrc, (size_t)(m * (long)(Ca + Cb) + v * 8) * 2, __builtin_bit_cast(h8, d));
    else *(uint4*)(y + m * (long)(Ca + Cb) + v * 8) = d;
  }
}

// ---- fused classifier-free guidance + DDIM update (fp32 math, one rounding) ------------------------------------------
// eps = eps_u + g (eps_c - eps_u)          reference ddim/sdxl_pipeline.py:842-844
// out = c_x * x + c_e * eps                 sampling: DDIMScheduler.step (eta 0); inversion: pnp_pipeline.py:73-85
// out2 (optional) receives a second copy (the cat([latents]*2) input of the next CFG evaluation, :826)
// coef (optional, device [B][3] = {g, c_x, c_e} per batch element of `per` elements each): requests with their own guidance scale and their own
// position in their own schedule share one launch; null: the scalars for all. Same arithmetic either way (fp32, one rounding).
__global__ void ddim_step_kernel(const half_t* x, const half_t* eps_u, const half_t* eps_c, float g, float c_x, float c_e,
                                 half_t* out, half_t* out2, long n, const float* coef, long per) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    if (coef) { const float* c = coef + 3 * (i / per); g = c[0]; c_x = c[1]; c_e = c[2]; }
    float e = (float)eps_u[i];
    if (eps_c) e = e + g * ((float)eps_c[i] - e);
    const half_t o = (half_t)(c_x * (float)x[i] + c_e * e);
    out[i] = o;
    if (out2) out2[i] = o;
  }
}

// Sampler update of the embedding prior (reference prior/model.py:208-240 get_eps, :627-637 guidance + DDPMScheduler.step), fp32:
//   eps_i = (s - sqrt_a * o_i) / sqrt_b ; eps = eps_u + g (eps_c - eps_u) ; x0 = (s - sqrt_b * eps) / sqrt_a
//   out = k0 * x0 + k1 * s + sigma * noise
// o_c / o_u: fp16 outputs of the sequence model for the conditioned / unconditioned half (o_c NULL = no guidance).
__global__ void prior_step_kernel(const float* s, const half_t* o_c, const half_t* o_u, const float* noise, float g, float sqrt_a, float sqrt_b,
                                  float k0, float k1, float sigma, float* out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float x = s[i];
    float e = (x - sqrt_a * (float)o_u[i]) / sqrt_b;
    if (o_c) { const float ec = (x - sqrt_a * (float)o_c[i]) / sqrt_b; e = e + g * (ec - e); }
    const float x0 = (x - sqrt_b * e) / sqrt_a;
    float o = k0 * x0 + k1 * x;
    if (noise) o += sigma * noise[i];
    out[i] = o;
  }
}

// out = (1 - m) * (c0 * init + c1 * noise) + m * x, m = mask[b, 0, y, x] shared by the C channels: the per-step latent blend of
// the inpainting loop (diffusers StableDiffusionXLInpaintPipeline, 4-channel UNet branch: known region re-noised to the next
// timestep with DDIM add_noise, c0 = sqrt(abar), c1 = sqrt(1 - abar); c0 = 1, c1 = 0 after the last step). out2 optional.
__global__ void mask_blend_kernel(const half_t* x, const half_t* init, const half_t* noise, const half_t* mask, float c0, float c1,
                                  half_t* out, half_t* out2, int C, long HW, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (C * HW), p = i % HW;
    const float m = (float)mask[b * HW + p];
    const float keep = c0 * (float)init[i] + c1 * (float)noise[i];
    const half_t o = (half_t)((1.f - m) * keep + m * (float)x[i]);
    out[i] = o;
    if (out2) out2[i] = o;
  }
}

// Fold a LayerNorm (gamma, beta over K features) into the linear layer that consumes it (weights finalize, once):
//   Wf[n][k] = fp16(W[n][k] * gamma[k]);  cs[n] = sum_k Wf[n][k] (of the ROUNDED values the MFMA will see);  lb[n] = bias[n] + sum_k W[n][k] * beta[k]
// so that  LN(x) . W^T + bias = rstd * (x . Wf^T - mean * cs) + lb  (gemm_f16_kernel epilogue). One workgroup per output row.
__global__ __launch_bounds__(256) void fold_ln_kernel(const half_t* W, const half_t* gamma, const half_t* beta, const half_t* bias,
                                                      half_t* Wf, float* cs, float* lb, int K) {
  __shared__ float red[2][4];
  const int n = blockIdx.x;
  float s = 0.f, t = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float w = (float)W[(size_t)n * K + k];
    const half_t wf = (half_t)(w * (float)gamma[k]);
    Wf[(size_t)n * K + k] = wf;
    s += (float)wf;
    t += w * (float)beta[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); t += __shfl_xor(t, o); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cs[n] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    lb[n] = (bias ? (float)bias[n] : 0.f) + ((red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

// ---- CLIP text encoders (conditioning side of the path; reference encode_prompt, ddim/sdxl_pipeline.py:202-395) -----------------
// x[row] = token_embedding[ids[row]] + position_embedding[row % T]; also the {sum, sum^2} of the fp16 row for the folded LayerNorm
// of the first layer (slot 0). One wave per token row.
__global__ __launch_bounds__(256) void clip_embed_kernel(const int* ids, const half_t* tok, const half_t* embeds, const half_t* pos, half_t* x, float* stats,
                                                         int rows, int T, int H, int vocab) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const half_t* a = embeds + (size_t)row * H;            // inputs_embeds form (GPT2Model(inputs_embeds=...))
  if (!embeds) {
    int id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    a = tok + (size_t)id * H;
  }
  const half_t* b = pos + (size_t)(row % T) * H;
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane * 8; k < H; k += 512) {
    const h8 va = *(const h8*)(a + k), vb = *(const h8*)(b + k);
    h8 o;
#pragma unroll
    for (int u = 0; u < 8; ++u) { o[u] = (half_t)((float)va[u] + (float)vb[u]); const float f = (float)o[u]; s1 += f; s2 += f * f; }
    *(h8*)(x + (size_t)row * H + k) = o;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  if (lane == 0) ((float2*)stats)[row] = make_float2(s1, s2);
}

// Causal self-attention over short sequences (T <= 128, head_dim 64): one workgroup per (batch, head); K and V of the head in LDS
// (rows padded to 66 halves: conflict-free column walks), one wave per query row at a time, fp32 softmax over keys 0..q.
// qkv: [B*T, 3*H] rows = [q | k | v]; out: [B*T, H]. Tiny problem (77 x 77 per head): VALU, no MFMA.
__global__ __launch_bounds__(256) void causal_attention_small_kernel(const half_t* qkv, half_t* out, int T, int heads) {
  constexpr int D = 64, LD = 66, TMAX = 128;
  __shared__ half_t sk[TMAX * LD], sv[TMAX * LD];
  __shared__ float sq[4][D], sp[4][TMAX];
  const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
  const int H = heads * D;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const half_t* base = qkv + (size_t)b * T * 3 * H + hd * D;
  for (int i = threadIdx.x; i < T * D; i += 256) {
    const int t = i / D, d = i % D;
    sk[t * LD + d] = base[(size_t)t * 3 * H + H + d];
    sv[t * LD + d] = base[(size_t)t * 3 * H + 2 * H + d];
  }
  __syncthreads();
  for (int q = wave; q < T; q += 4) {
    sq[wave][lane] = (float)base[(size_t)q * 3 * H + lane] * 0.125f;
    __builtin_amdgcn_wave_barrier();
    float s[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = lane + u * 64;
      float a = -INFINITY;
      if (k <= q) {
        a = 0.f;
#pragma unroll 16
        for (int d = 0; d < D; ++d) a += sq[wave][d] * (float)sk[k * LD + d];
      }
      s[u] = a;
    }
    float m = fmaxf(s[0], s[1]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float e0 = __expf(s[0] - m), e1 = lane + 64 <= q ? __expf(s[1] - m) : 0.f;
    float sum = e0 + e1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
    sp[wave][lane] = e0 * inv;
    if (lane + 64 < TMAX) sp[wave][lane + 64] = e1 * inv;
    __builtin_amdgcn_wave_barrier();
    float acc = 0.f;
    for (int k = 0; k <= q; ++k) acc += sp[wave][k] * (float)sv[k * LD + lane];
    out[((size_t)b * T + q) * H + hd * D + lane] = (half_t)acc;
    __builtin_amdgcn_wave_barrier();
  }
}

// pooled rows: for each sequence the EOS position (eos_id == 2: position of the largest id, the legacy CLIP rule; else the first
// occurrence of eos_id), then final LayerNorm of that row of the last hidden state. One workgroup (one wave) per sequence.
__global__ __launch_bounds__(64) void clip_pool_kernel(const int* ids, const half_t* x, const half_t* gamma, const half_t* beta, half_t* out,
                                                       int T, int H, int eos_id, float eps) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int best = -1, pos = 0x7fffffff;
  for (int t = lane; t < T; t += 64) {
    const int id = ids[b * T + t];
    const int key = eos_id == 2 ? id : (id == eos_id ? 1 : 0);
    if (key > best || (key == best && t < pos)) { best = key; pos = t; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int ob = __shfl_xor(best, o), op = __shfl_xor(pos, o);
    if (ob > best || (ob == best && op < pos)) { best = ob; pos = op; }
  }
  const half_t* row = x + ((size_t)b * T + pos) * H;
  float s1 = 0.f, s2 = 0.f;
  for (int k = lane; k < H; k += 64) { const float f = (float)row[k]; s1 += f; s2 += f * f; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
  const float mean = s1 / H;
  float var = 0.f;
  for (int k = lane; k < H; k += 64) { const float f = (float)row[k] - mean; var += f * f; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) var += __shfl_xor(var, o);
  const float rstd = rsqrtf(var / H + eps);
  (void)s2;
  for (int k = lane; k < H; k += 64) out[(size_t)b * H + k] = (half_t)(((float)row[k] - mean) * rstd * (float)gamma[k] + (float)beta[k]);
}

// reads a buffer once (16 B per lane) so that it sits in L2 / Infinity Cache again: the autotuner re-warms a site's ACTIVATIONS after
// flushing the caches, because in the real sequence they were written by the launch just before
__global__ void touch_kernel(const uint4* p, size_t n16, unsigned* sink) {
  unsigned acc = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = p[i]; acc ^= v.x ^ v.w; }
  if (acc == 0x9e3779b9u) *sink = acc;
}

// ---- one-time weight re-layouts -----------------------------------------------------------------------------------------
// conv [Co][Ci][3][3] -> [Co][ky][kx][Ci]: the K order of every 3x3 convolution kernel (the gathered tiles walk it tap-major, the halo-staged tiles block-major over the
// same rows: weight tile (block, tap) starts at column tap * Ci + block * 64; conv_out_kernel reads the same layout, any Ci)
__global__ void pack_conv_kernel(const half_t* src, half_t* dst, int Co, int Ci) {
  const long total = (long)Co * Ci * 9;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i / ((long)Ci * 9));
    const int k = (int)(i - (long)co * Ci * 9);
    const int tap = k / Ci, ci = k - tap * Ci;
    dst[i] = src[((long)co * Ci + ci) * 9 + tap];
  }
}
// GEGLU pairing: packed row p (block t = p/32): p%32 < 16 -> value row 16t + p%32 ; else gate row half + 16t + p%32 - 16
// conv_in weights [Co][Cin*9] -> [Co][64] (k = ci*9 + tap, zero padded): the first-operand image of conv_in_kernel's two MFMA steps
__global__ void pack_conv_in_kernel(const half_t* src, half_t* dst, int Co, int KT) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < (long)Co * 64; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i >> 6), k = (int)(i & 63);
    dst[i] = k < KT ? src[(size_t)co * KT + k] : (half_t)0.f;
  }
}
__global__ void pack_geglu_kernel(const half_t* src, half_t* dst, int rows, int rowlen) {
  const long total = (long)rows * rowlen;
  const int half_ = rows / 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i / rowlen), k = (int)(i - (long)p * rowlen);
    const int t = p >> 5, w = p & 31;
    const int s = w < 16 ? 16 * t + w : half_ + 16 * t + (w - 16);
    dst[i] = src[(long)s * rowlen + k];
  }
}

// ---- row softmax, in place: x[r, :] = softmax(scale * x[r, :]); one workgroup per row, row held in registers ----------------
// (VAE mid-block attention: single head, head_dim = channels; scores materialised by the GEMM kernel)
template <int NV>   // 16-byte vectors per thread: n <= NV * 256 * 8
__global__ __launch_bounds__(256) void softmax_rows_kernel(half_t* x, long ld, int n, float scale_log2e) {
  __shared__ float red[8];
  half_t* row = x + (long)blockIdx.x * ld;
  const int nvec = n >> 3, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float v[NV][8];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nvec) {
      const h8 d = *(const h8*)(row + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = (float)d[e]; mx = fmaxf(mx, v[i][e]); }
    }
  }
  mx = wave_max(mx);
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float mc = mx * scale_log2e;
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (threadIdx.x + i * 256 < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = __builtin_amdgcn_exp2f(fmaf(v[i][e], scale_log2e, -mc)); sum += v[i][e]; }
    }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nvec) {
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)(v[i][e] * inv);
      *(h8*)(row + c * 8) = o;
    }
  }
}

// ---- 1x1 convolution on a small-channel NCHW tensor (VAE quant_conv / post_quant_conv, <= 8 channels) -----------------------
__global__ void conv1x1_nchw_kernel(const half_t* x, const half_t* w /*[Co][Ci]*/, const half_t* bias, half_t* y, int B, int Ci, int Co, long HW) {
  const long total = (long)B * HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, p = i - b * HW;
    float in[8];
    for (int c = 0; c < Ci; ++c) in[c] = (float)x[(b * Ci + c) * HW + p];
    for (int o = 0; o < Co; ++o) {
      float a = (float)bias[o];
      for (int c = 0; c < Ci; ++c) a += in[c] * (float)w[o * Ci + c];
      y[(b * Co + o) * HW + p] = (half_t)a;
    }
  }
}


// ---- IPAttnProcessor2_0's `attn_map` side effect (reference attention_processor.py:390-391):
//      attn_map[b,h,q,t] = sum_d Q[b,q,h*64+d] * softmax_t(K_ip[b,t,h*64+d])  -- the softmax binds to ip_key^T (over the TOKEN axis, no 1/sqrt(d))
//      before the matmul. HBM-bound: Q once in (128 B per row and head), ntok halves out. One thread per (b, head, query).
__global__ __launch_bounds__(256) void ip_attn_map_kernel(const half_t* Q, int ldq, const half_t* Kip, int ldk, half_t* out, int heads, int Nq, int ntok) {
  __shared__ float S[64 * 16];           // [d][t]
  const int b = blockIdx.z, hd = blockIdx.y;
  if (threadIdx.x < 64) {
    const int d = threadIdx.x;
    float v[16] = {}, mx = -3.0e38f, sum = 0.f;
    for (int t = 0; t < ntok; ++t) { v[t] = (float)Kip[((size_t)b * ntok + t) * ldk + hd * 64 + d]; mx = fmaxf(mx, v[t]); }
    for (int t = 0; t < ntok; ++t) { v[t] = __expf(v[t] - mx); sum += v[t]; }
    const float inv = 1.f / sum;
    for (int t = 0; t < 16; ++t) S[d * 16 + t] = t < ntok ? v[t] * inv : 0.f;
  }
  __syncthreads();
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= Nq) return;
  const half_t* qp = Q + ((size_t)b * Nq + q) * ldq + hd * 64;
  float acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const h8 v = *(const h8*)(qp + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)v[j];
#pragma unroll
      for (int t = 0; t < 16; ++t) acc[t] = fmaf(x, S[(c * 8 + j) * 16 + t], acc[t]);     // columns t >= ntok of S are zero
    }
  }
  half_t* op = out + (((size_t)b * heads + hd) * Nq + q) * ntok;
  for (int t = 0; t < ntok; ++t) op[t] = (half_t)acc[t];
}

// ---- host launchers -------------------------------------------------------------------------------------------------------
static inline int grid_for(long n, int block) { long g = (n + block - 1) / block; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

hipError_t ia2p_launch_embed(float t, const float* ts, const half_t* text_embeds, const half_t* time_ids, half_t* tsin, half_t* addin,
                             int B, int Tp, int P, int Ad, int nids, hipStream_t s) {
  hipLaunchKernelGGL(embed_kernel, dim3(B), dim3(256), 0, s, t, ts, text_embeds, time_ids, tsin, addin, B, Tp, P, Ad, nids);
  return hipGetLastError();
}
hipError_t ia2p_launch_linear_small(const half_t* X, int ldx, const half_t* W, const half_t* bias, const half_t* addend, int ldadd,
                                    half_t* out, int ldo, int M, int N, int K, int silu_in, int silu_out, hipStream_t s) {
  if (M > 16 || K % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(linear_small_kernel, dim3((N + 3) / 4), dim3(64), 0, s, X, ldx, W, bias, addend, ldadd, out, ldo, M, N, K, silu_in, silu_out);
  return hipGetLastError();
}
hipError_t ia2p_launch_conv_in(const half_t* x, const half_t* w, const half_t* bias, half_t* y, int B, int Cin, int H, int W, int Co, hipStream_t s, float out_scale) {
  if (Co % 8 || Cin * 9 > 64) return hipErrorInvalidValue;
  const int ppb = 64;
  const size_t sm = (size_t)Co * 64 * sizeof(half_t) + (size_t)64 * Co * sizeof(half_t);      // padded weights + 4 x 16 pixel rows of output
  if (sm > 160 * 1024) return hipErrorInvalidValue;
  static bool attr[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (sm > 65536 && (dev < 0 || dev >= 64 || !attr[dev])) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_in_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr[dev] = true;
  }
  hipLaunchKernelGGL(conv_in_kernel, dim3((B * H * W + ppb - 1) / ppb), dim3(256), sm, s, x, w, bias, y, B, Cin, H, W, Co, ppb, out_scale);
  return hipGetLastError();
}
hipError_t ia2p_launch_conv_out(const half_t* x, int ldx, const half_t* w, const half_t* bias, half_t* y, int B, int C, int H, int W, int Co, hipStream_t s) {
  if (Co > 8 || C % 32) return hipErrorInvalidValue;
  const size_t sm = (size_t)Co * 9 * C * sizeof(half_t);
  if (sm > 160 * 1024) return hipErrorInvalidValue;
  static bool attr[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (sm > 65536 && (dev < 0 || dev >= 64 || !attr[dev])) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_out_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_out_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr[dev] = true;
  }
  const dim3 grid((B * H * W + 63) / 64);
  switch (C) {
    case 128: hipLaunchKernelGGL(conv_out_kernel<4>, grid, dim3(256), sm, s, x, ldx, w, bias, y, B, C, H, W, Co); break;
    case 320: hipLaunchKernelGGL(conv_out_kernel<10>, grid, dim3(256), sm, s, x, ldx, w, bias, y, B, C, H, W, Co); break;
    case 384: hipLaunchKernelGGL(conv_out_kernel<12>, grid, dim3(256), sm, s, x, ldx, w, bias, y, B, C, H, W, Co); break;
    case 512: hipLaunchKernelGGL(conv_out_kernel<16>, grid, dim3(256), sm, s, x, ldx, w, bias, y, B, C, H, W, Co); break;
    default: hipLaunchKernelGGL(conv_out_kernel<0>, grid, dim3(256), sm, s, x, ldx, w, bias, y, B, C, H, W, Co); break;
  }
  return hipGetLastError();
}
hipError_t ia2p_launch_concat(const half_t* a, int lda, int Ca, const half_t* b, int ldb, int Cb, half_t* y, long M, hipStream_t s) {
  if (Ca % 8 || Cb % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(concat_kernel, dim3(grid_for(M * ((Ca + Cb) / 8), 256)), dim3(256), 0, s, a, lda, Ca, b, ldb, Cb, y, M,
                     ((ia2p_wt_mask() & 16) && (size_t)M * (Ca + Cb) * 2 < (size_t)0x7ffffff0) ? 1 : 0);
  return hipGetLastError();
}
hipError_t ia2p_launch_ddim_step(const half_t* x, const half_t* eps_u, const half_t* eps_c, float g, float c_x, float c_e,
                                 half_t* out, half_t* out2, long n, hipStream_t s, const float* coef, long per) {
  hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, x, eps_u, eps_c, g, c_x, c_e, out, out2, n, coef, per > 0 ? per : 1);
  return hipGetLastError();
}
hipError_t ia2p_launch_mask_blend(const half_t* x, const half_t* init, const half_t* noise, const half_t* mask, float c0, float c1,
                                  half_t* out, half_t* out2, int B, int C, long HW, hipStream_t s) {
  const long n = (long)B * C * HW;
  hipLaunchKernelGGL(mask_blend_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, x, init, noise, mask, c0, c1, out, out2, C, HW, n);
  return hipGetLastError();
}
hipError_t ia2p_launch_prior_step(const float* smp, const half_t* o_c, const half_t* o_u, const float* noise, float g, float sqrt_a, float sqrt_b, float k0, float k1,
                                  float sigma, float* out, long n, hipStream_t s) {
  hipLaunchKernelGGL(prior_step_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, smp, o_c, o_u, noise, g, sqrt_a, sqrt_b, k0, k1, sigma, out, n);
  return hipGetLastError();
}
// dst[r] = [a[r] | b[r]] (rows of Ka and Kb elements, both multiples of 8), bias_dst = bias_a + bias_b (fp32 add, one rounding)
__global__ void cat_rows_kernel(const half_t* a, int Ka, const half_t* b, int Kb, const half_t* bias_a, const half_t* bias_b, half_t* dst, half_t* bias_dst, int rows) {
  const int va = Ka >> 3, vt = (Ka + Kb) >> 3;
  const long total = (long)rows * vt;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / vt;
    const int v = (int)(i - r * vt);
    *(uint4*)(dst + r * (long)(Ka + Kb) + v * 8) = v < va ? *(const uint4*)(a + r * (long)Ka + v * 8) : *(const uint4*)(b + r * (long)Kb + (v - va) * 8);
  }
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += gridDim.x * blockDim.x) bias_dst[r] = (half_t)((float)bias_a[r] + (float)bias_b[r]);
}
hipError_t ia2p_launch_cat_rows(const half_t* a, int Ka, const half_t* b, int Kb, const half_t* bias_a, const half_t* bias_b, half_t* dst, half_t* bias_dst, int rows, hipStream_t s) {
  if (Ka % 8 || Kb % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cat_rows_kernel, dim3(grid_for((long)rows * ((Ka + Kb) / 8), 256)), dim3(256), 0, s, a, Ka, b, Kb, bias_a, bias_b, dst, bias_dst, rows);
  return hipGetLastError();
}
hipError_t ia2p_launch_fold_ln(const half_t* W, const half_t* gamma, const half_t* beta, const half_t* bias, half_t* Wf, float* cs, float* lb,
                               int N, int K, hipStream_t s) {
  hipLaunchKernelGGL(fold_ln_kernel, dim3(N), dim3(256), 0, s, W, gamma, beta, bias, Wf, cs, lb, K);
  return hipGetLastError();
}
hipError_t ia2p_launch_clip_embed(const int* ids, const half_t* tok, const half_t* embeds, const half_t* pos, half_t* x, float* stats, int rows, int T, int H,
                                  int vocab, hipStream_t s) {
  if (H % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(clip_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, ids, tok, embeds, pos, x, stats, rows, T, H, vocab);
  return hipGetLastError();
}
hipError_t ia2p_launch_causal_attention_small(const half_t* qkv, half_t* out, int B, int T, int heads, hipStream_t s) {
  if (T < 1 || T > 128) return hipErrorInvalidValue;
  hipLaunchKernelGGL(causal_attention_small_kernel, dim3(B * heads), dim3(256), 0, s, qkv, out, T, heads);
  return hipGetLastError();
}
hipError_t ia2p_launch_clip_pool(const int* ids, const half_t* x, const half_t* gamma, const half_t* beta, half_t* out, int B, int T, int H, int eos_id,
                                 float eps, hipStream_t s) {
  hipLaunchKernelGGL(clip_pool_kernel, dim3(B), dim3(64), 0, s, ids, x, gamma, beta, out, T, H, eos_id, eps);
  return hipGetLastError();
}
hipError_t ia2p_launch_touch(const void* p, size_t bytes, unsigned* sink, hipStream_t s) {
  const size_t n16 = bytes / 16;
  if (!n16) return hipSuccess;
  hipLaunchKernelGGL(touch_kernel, dim3((unsigned)std::min<size_t>(2048, (n16 + 255) / 256)), dim3(256), 0, s, (const uint4*)p, n16, sink);
  return hipGetLastError();
}
hipError_t ia2p_launch_pack_conv(const half_t* src, half_t* dst, int Co, int Ci, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv_kernel, dim3(grid_for((long)Co * Ci * 9, 256)), dim3(256), 0, s, src, dst, Co, Ci);
  return hipGetLastError();
}
hipError_t ia2p_launch_pack_conv_in(const half_t* src, half_t* dst, int Co, int KT, hipStream_t s) {
  if (KT > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pack_conv_in_kernel, dim3(grid_for((long)Co * 64, 256)), dim3(256), 0, s, src, dst, Co, KT);
  return hipGetLastError();
}
hipError_t ia2p_launch_pack_geglu(const half_t* src, half_t* dst, int rows, int rowlen, hipStream_t s) {
  hipLaunchKernelGGL(pack_geglu_kernel, dim3(grid_for((long)rows * rowlen, 256)), dim3(256), 0, s, src, dst, rows, rowlen);
  return hipGetLastError();
}

hipError_t ia2p_launch_softmax_rows(half_t* x, long ld, int rows, int n, float scale, hipStream_t s) {
  if (n % 8 || n > 8 * 256 * 8) return hipErrorInvalidValue;
  const float sl = scale * 1.4426950408889634f;
  const int nv = (n / 8 + 255) / 256;
  if (nv <= 1) hipLaunchKernelGGL(softmax_rows_kernel<1>, dim3(rows), dim3(256), 0, s, x, ld, n, sl);
  else if (nv <= 2) hipLaunchKernelGGL(softmax_rows_kernel<2>, dim3(rows), dim3(256), 0, s, x, ld, n, sl);
  else if (nv <= 4) hipLaunchKernelGGL(softmax_rows_kernel<4>, dim3(rows), dim3(256), 0, s, x, ld, n, sl);
  else hipLaunchKernelGGL(softmax_rows_kernel<8>, dim3(rows), dim3(256), 0, s, x, ld, n, sl);
  return hipGetLastError();
}
hipError_t ia2p_launch_conv1x1_nchw(const half_t* x, const half_t* w, const half_t* bias, half_t* y, int B, int Ci, int Co, long HW, hipStream_t s) {
  if (Ci > 8 || Co > 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(conv1x1_nchw_kernel, dim3(grid_for((long)B * HW, 256)), dim3(256), 0, s, x, w, bias, y, B, Ci, Co, HW);
  return hipGetLastError();
}
hipError_t ia2p_launch_ip_attn_map(const half_t* Q, int ldq, const half_t* Kip, int ldk, half_t* out, int B, int heads, int Nq, int ntok, hipStream_t s) {
  if (ntok < 1 || ntok > 16 || ldq % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(ip_attn_map_kernel, dim3((Nq + 255) / 256, heads, B), dim3(256), 0, s, Q, ldq, Kip, ldk, out, heads, Nq, ntok);
  return hipGetLastError();
}
