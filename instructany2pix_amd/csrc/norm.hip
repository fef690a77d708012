// GroupNorm(+SiLU) and LayerNorm for channels-last fp16 activations on gfx950. HBM-bound kernels:
// 16-byte coalesced loads/stores, fp32 statistics, wavefront/LDS reductions.
//
// Replaces, inside the diffusers UNet the reference drives:
//   ResnetBlock2D.norm1/norm2 + SiLU (eps 1e-5), Transformer2DModel.norm (eps 1e-6, no SiLU),
//   conv_norm_out + SiLU, and BasicTransformerBlock.norm1/2/3 (LayerNorm eps 1e-5).
//
// GroupNorm over x[B, HW, C] (C contiguous), G groups of Cg = C/G channels:
//   pass 1 (gn_stats): grid (chunks, B); a block owns `rows` pixels x all channels, every thread keeps a
//          FIXED set of 8 (or 16) channels so per-channel sum / sum-of-squares live in registers; they are
//          folded per group through LDS and written as one (sum, sumsq) pair per (b, chunk, group).
//   pass 2 (gn_apply): same thread->channel map; each block folds the chunk partials of its batch in fp64,
//          derives per-channel scale/shift once, then streams its rows: y = silu(x*scale + shift).
// Algorithmic traffic: read x twice (second read normally served by L2 / Infinity Cache), write y once.
#include "common.h"
#include "gn_fold.h"

#include <cstdlib>

struct GNArgs {
  const half_t* x; half_t* y;
  const half_t* gamma; const half_t* beta;
  float* partial;         // [B][chunks][G][2]
  int B, HW, C, G, chunks, rows;   // rows = pixels per stats chunk (last chunk may be short)
  int arows;                       // pixels per apply block (finer than the stats chunking)
  int ldx, ldy;
  float eps; int silu;
  const half_t* x2; int ldx2, Ca;   // optional second source: channels [Ca, C) come from x2 (the skip half of an up-block input that is never concatenated)
};

// thread map: TX = (C/8)/V threads across channels (V vectors of 8 channels each), TY = blockDim/TX rows in flight
template <int V>
__global__ __launch_bounds__(256) void gn_stats_kernel(const half_t* ax, half_t* ay, const half_t* agamma, const half_t* abeta, float* apartial, int aB, int aHW, int aC, int aG,
                                                           int achunks, int arows_, int aarows, int aldx, int aldy, float aeps, int asilu, const half_t* ax2, int aldx2, int aCa) {
  // scalar arguments (the first 14 dwords are preloaded into SGPRs: build.py PRELOAD), re-assembled into the descriptor the body uses
  GNArgs p;
  p.x = ax; p.y = ay; p.gamma = agamma; p.beta = abeta; p.partial = apartial; p.B = aB; p.HW = aHW; p.C = aC; p.G = aG; p.chunks = achunks;
  p.rows = arows_; p.arows = aarows; p.ldx = aldx; p.ldy = aldy; p.eps = aeps; p.silu = asilu; p.x2 = ax2; p.ldx2 = aldx2; p.Ca = aCa;
  extern __shared__ float red[];   // [2][TY][C]
  const int nvec = p.C >> 3, TX = nvec / V, TY = blockDim.x / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int b = blockIdx.y, r0 = blockIdx.x * p.rows, r1 = min(r0 + p.rows, p.HW);
  float s[V][8], ss[V][8];
#pragma unroll
  for (int v = 0; v < V; ++v)
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[v][e] = 0.f; ss[v][e] = 0.f; }
  if (ty < TY) {
    const half_t* src[V]; int sld[V];                 // a thread's channels are fixed: so is the tensor they come from
#pragma unroll
    for (int v = 0; v < V; ++v) {
      const int c0 = (tx + v * TX) * 8;
      const bool second = p.x2 && c0 >= p.Ca;
      src[v] = second ? p.x2 + (size_t)b * p.HW * p.ldx2 + (c0 - p.Ca) : p.x + (size_t)b * p.HW * p.ldx + c0;
      sld[v] = second ? p.ldx2 : p.ldx;
    }
    for (int r = r0 + ty; r < r1; r += 4 * TY) {       // 4 rows (4*V 16-byte loads) in flight per thread
      h8 d[4][V];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < V; ++v) {
          const int rr = min(r + u * TY, r1 - 1);
          d[u][v] = *(const h8*)(src[v] + (size_t)rr * sld[v]);
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u * TY < r1) {
#pragma unroll
          for (int v = 0; v < V; ++v)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)d[u][v][e]; s[v][e] += f; ss[v][e] += f * f; }
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = (tx + v * TX) * 8 + e;
        red[ty * p.C + c] = s[v][e];
        red[(TY + ty) * p.C + c] = ss[v][e];
      }
  }
  __syncthreads();
  // fold the TY row-slices per channel (column c is touched by one thread only), then per group
  for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
    float a = 0.f, q = 0.f;
    for (int t = 0; t < TY; ++t) { a += red[t * p.C + c]; q += red[(TY + t) * p.C + c]; }
    red[c] = a; red[TY * p.C + c] = q;
  }
  __syncthreads();
  const int Cg = p.C / p.G;
  if ((int)threadIdx.x < p.G) {
    const int g = threadIdx.x;
    float a = 0.f, q = 0.f;
    for (int c = g * Cg; c < (g + 1) * Cg; ++c) { a += red[c]; q += red[TY * p.C + c]; }
    float* out = p.partial + (((size_t)b * p.chunks + blockIdx.x) * p.G + g) * 2;
    out[0] = a; out[1] = q;
  }
}

template <int V>
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* ax, half_t* ay, const half_t* agamma, const half_t* abeta, float* apartial, int aB, int aHW, int aC, int aG,
                                                           int achunks, int arows_, int aarows, int aldx, int aldy, float aeps, int asilu, const half_t* ax2, int aldx2, int aCa) {
  // scalar arguments (the first 14 dwords are preloaded into SGPRs: build.py PRELOAD), re-assembled into the descriptor the body uses
  GNArgs p;
  p.x = ax; p.y = ay; p.gamma = agamma; p.beta = abeta; p.partial = apartial; p.B = aB; p.HW = aHW; p.C = aC; p.G = aG; p.chunks = achunks;
  p.rows = arows_; p.arows = aarows; p.ldx = aldx; p.ldy = aldy; p.eps = aeps; p.silu = asilu & 1; p.x2 = ax2; p.ldx2 = aldx2; p.Ca = aCa;
  const bool wt = (asilu & 2) != 0;          // write-through output (bit 1 of the flag word)
  extern __shared__ float stat[];   // [G][2] mean, rstd
  const int nvec = p.C >> 3, TX = nvec / V, TY = blockDim.x / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int b = blockIdx.y, r0 = blockIdx.x * p.arows, r1 = min(r0 + p.arows, p.HW);
  const int Cg = p.C / p.G;
  {   // fold the chunk partials of this batch: all threads, (group, chunk-slice) each, then across slices in fp64
    double* fold = (double*)(stat + 2 * p.G);          // [nsl][G][2]
    const int nsl = blockDim.x / p.G, g = threadIdx.x % p.G, sl = threadIdx.x / p.G;
    if (sl < nsl) {
      float a = 0.f, q = 0.f;
      for (int c = sl; c < p.chunks; c += nsl) {
        const float2 v = *(const float2*)(p.partial + (((size_t)b * p.chunks + c) * p.G + g) * 2);
        a += v.x; q += v.y;
      }
      fold[(sl * p.G + g) * 2] = (double)a; fold[(sl * p.G + g) * 2 + 1] = (double)q;
    }
    __syncthreads();
    if ((int)threadIdx.x < p.G) {
      double a = 0.0, q = 0.0;
      for (int t = 0; t < nsl; ++t) { a += fold[(t * p.G + g) * 2]; q += fold[(t * p.G + g) * 2 + 1]; }
      const double n = (double)p.HW * Cg;
      const double mean = a / n;
      double var = q / n - mean * mean;
      if (var < 0.0) var = 0.0;
      stat[g * 2] = (float)mean;
      stat[g * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
  }
  __syncthreads();
  if (ty >= TY) return;
  float sc[V][8], sh[V][8];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int c0 = (tx + v * TX) * 8;
    const h8 ga = *(const h8*)(p.gamma + c0), be = *(const h8*)(p.beta + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int g = (c0 + e) / Cg;
      const float mean = stat[g * 2], rstd = stat[g * 2 + 1];
      sc[v][e] = rstd * (float)ga[e];
      sh[v][e] = (float)be[e] - mean * sc[v][e];
    }
  }
  const half_t* src[V]; int sld[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int c0 = (tx + v * TX) * 8;
    const bool second = p.x2 && c0 >= p.Ca;
    src[v] = second ? p.x2 + (size_t)b * p.HW * p.ldx2 + (c0 - p.Ca) : p.x + (size_t)b * p.HW * p.ldx + c0;
    sld[v] = second ? p.ldx2 : p.ldx;
  }
  half_t* obase = p.y + (size_t)b * p.HW * p.ldy;
  const __amdgpu_buffer_rsrc_t y_rsrc = wt_rsrc((void*)obase, (size_t)p.HW * p.ldy * 2);
  for (int r = r0 + ty; r < r1; r += 4 * TY) {         // 4 rows in flight per thread
    h8 d[4][V];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < V; ++v) {
        const int rr = min(r + u * TY, r1 - 1);
        d[u][v] = *(const h8*)(src[v] + (size_t)rr * sld[v]);
      }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + u * TY < r1) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          h8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = (float)d[u][v][e] * sc[v][e] + sh[v][e];
            if (p.silu) f = silu_f(f);
            o[e] = (half_t)f;
          }
          if (wt) store16_wt(y_rsrc, ((size_t)(r + u * TY) * p.ldy + (tx + v * TX) * 8) * 2, o);
          else *(h8*)(obase + (size_t)(r + u * TY) * p.ldy + (tx + v * TX) * 8) = o;
        }
      }
  }
}

// ---- single-pass GroupNorm (round 4): ONE launch, the tensor read ONCE ------------------------------------------------------------------------------
// A workgroup owns, for one image, a SPAN of `gs` whole groups = `NCH` 16-byte chunks per pixel (the smallest span that is both whole groups and whole
// chunks: Cg = 40 or 80 channels -> one group = 5 / 10 chunks; Cg = 10 / 20 / 30 / 60 -> 4 / 2 / 4 / 2 groups = 5 / 5 / 15 / 15 chunks). Its slice of the image,
// HW pixels x NCH chunks, stays IN REGISTERS between the statistics and the normalisation: wave w reads chunk column cc = w / WPC of pixels
// (w % WPC) * 64 + lane, + P, + 2 P, ... (P = 64 WPC pixel lanes; IT of them per thread, compile time), so every lane of a wave has the same channels --
// hence the same group membership, gamma and beta -- and the statistics are a fixed butterfly over the wave (a chunk spans at most two groups: two
// {sum, sum of squares} pairs per lane), wave partials through LDS, folded per group in fp64 in wave order: deterministic, independent of the batch.
// Replaces gn_stats_kernel + gn_apply_kernel (two launches, the tensor read twice) wherever HW x NCH fits the registers of one 15-wave workgroup: every
// GroupNorm of the SDXL UNet at 512^2 except the 960-channel one at 64^2. Work: 2 x B x HW x C x 2 bytes (read once, write once).
template <int NCH, int IT>
__global__ __launch_bounds__(NCH == 10 ? 640 : 960) void gn_fused_kernel(const half_t* x, half_t* y, const half_t* gamma, const half_t* beta, int HW, int C, int Cg, int gs,
                                                                                  int ldx, int ldy, float eps, int flags, const half_t* x2, int ldx2, int Ca) {
  constexpr int WPC = NCH == 5 ? 3 : 1;            // waves per chunk column
  constexpr int P = 64 * WPC, NW = NCH * WPC;      // pixel lanes, waves
  __shared__ float wsum[NW][4];                    // per wave: {sum, sum of squares} of its first / second group
  __shared__ float stat[16][2];                    // per group of the span: mean, rstd
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cc = wave / WPC, pl = (wave % WPC) * 64 + lane;
  const int b = blockIdx.y, span = blockIdx.x;
  const int c0 = span * NCH * 8 + cc * 8;          // first channel of this thread's chunk
  const int g0 = span * gs;                        // first group of the span
  const int ga = c0 / Cg - g0;                     // (wave-uniform) group of the chunk's first element, span-local
  const int esplit = min(8, (g0 + ga + 1) * Cg - c0);      // elements [0, esplit) belong to group ga, [esplit, 8) to ga + 1
  const bool second = x2 && c0 >= Ca;
  const half_t* src = second ? x2 + (size_t)b * HW * ldx2 + (c0 - Ca) : x + (size_t)b * HW * ldx + c0;
  const int sld = second ? ldx2 : ldx;
  h8 d[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) d[k] = *(const h8*)(src + (size_t)min(pl + k * P, HW - 1) * sld);      // all loads in flight (clamped, never branched around)
  float sa = 0.f, qa = 0.f, sb = 0.f, qb = 0.f;
#pragma unroll
  for (int k = 0; k < IT; ++k)
    if (pl + k * P < HW) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)d[k][e];
        if (e < esplit) { sa += f; qa = fmaf(f, f, qa); } else { sb += f; qb = fmaf(f, f, qb); }
      }
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o, 64); qa += __shfl_xor(qa, o, 64); sb += __shfl_xor(sb, o, 64); qb += __shfl_xor(qb, o, 64); }
  if (lane == 0) { wsum[wave][0] = sa; wsum[wave][1] = qa; wsum[wave][2] = sb; wsum[wave][3] = qb; }
  __syncthreads();
  if ((int)threadIdx.x < gs) {                     // group j of the span: its waves in wave order, fp64
    const int j = threadIdx.x;
    double a = 0.0, q = 0.0;
    for (int w = 0; w < NW; ++w) {
      const int wc0 = span * NCH * 8 + (w / WPC) * 8, wga = wc0 / Cg - g0;
      const bool two = (g0 + wga + 1) * Cg - wc0 < 8;
      if (wga == j) { a += (double)wsum[w][0]; q += (double)wsum[w][1]; }
      if (two && wga + 1 == j) { a += (double)wsum[w][2]; q += (double)wsum[w][3]; }
    }
    const double n = (double)HW * Cg;
    const double mean = a / n;
    double var = q / n - mean * mean;
    if (var < 0.0) var = 0.0;
    stat[j][0] = (float)mean;
    stat[j][1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  const h8 gam = *(const h8*)(gamma + c0), bet = *(const h8*)(beta + c0);
  float sc[8], sh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int g = e < esplit ? ga : ga + 1;
    const float mean = stat[g][0], rstd = stat[g][1];
    sc[e] = rstd * (float)gam[e];
    sh[e] = (float)bet[e] - mean * sc[e];
  }
  const bool silu = (flags & 1) != 0, wt = (flags & 2) != 0;
  half_t* obase = y + (size_t)b * HW * ldy;
  const __amdgpu_buffer_rsrc_t y_rsrc = wt_rsrc((void*)obase, (size_t)HW * ldy * 2);
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int r = pl + k * P;
    if (r < HW) {
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = (float)d[k][e] * sc[e] + sh[e];
        if (silu) f = silu_f(f);
        o[e] = (half_t)f;
      }
      const size_t off = (size_t)r * ldy + c0;
      if (wt) store16_wt(y_rsrc, off * 2, o);
      else *(h8*)(obase + off) = o;
    }
  }
}

// single-pass launch if the shape fits (returns false otherwise: the caller runs the two-pass kernels)
static bool gn_fused_launch(const half_t* x, int ldx, half_t* y, int ldy, const half_t* gamma, const half_t* beta, int B, int HW, int C, int G, float eps, int flags,
                            hipStream_t s, const half_t* x2, int ldx2, int Ca) {
#ifdef IA2P_GN_TWOPASS      // A/B builds: the two-pass kernels everywhere
  return false;
#endif
  if (C % G || C % 8 || ldx % 8 || ldy % 8 || (x2 && (ldx2 % 8 || Ca % 8))) return false;
  const int Cg = C / G;
  if (Cg < 8) return false;                        // (a chunk would span more than two groups)
  int gs = 1;
  while ((gs * Cg) % 8) ++gs;                      // smallest span of whole groups that is whole 16-byte chunks
  const int nch = gs * Cg / 8;
  if (G % gs || (nch != 5 && nch != 10 && nch != 15) || gs > 16) return false;
  const int P = nch == 5 ? 192 : 64, it = (HW + P - 1) / P;
  // one workgroup per CU takes its slice in at the rate of ONE CU (~25 GB/s from beyond L2): only launches that put a workgroup on (nearly) every CU pay --
  // with 64 ... 128 workgroups (the 64^2 / 32^2 levels at 8 requests) the single pass measured +0.7 ms per step against the two-pass kernels' 2 000 workgroups
  // ... or whose slices are small enough that a workgroup's chain (load, reduce, normalise, store) is shorter than a second launch: the 16 x 16 level at any batch
  if ((long)(G / gs) * B < 200 && (long)HW * nch * 16 > 48 * 1024) return false;
  const dim3 grid(G / gs, B), block(nch == 10 ? 640 : 960);
#define IA2P_GNF(NCH_, IT_) hipLaunchKernelGGL((gn_fused_kernel<NCH_, IT_>), grid, block, 0, s, x, y, gamma, beta, HW, C, Cg, gs, ldx, ldy, eps, flags, x2, ldx2, Ca)
  if (nch == 5) {
    if (it <= 2) IA2P_GNF(5, 2); else if (it <= 6) IA2P_GNF(5, 6); else if (it <= 22) IA2P_GNF(5, 22); else return false;
  } else if (nch == 10) {
    if (it <= 4) IA2P_GNF(10, 4); else if (it <= 16) IA2P_GNF(10, 16); else return false;
  } else {
    if (it <= 4) IA2P_GNF(15, 4); else if (it <= 16) IA2P_GNF(15, 16); else return false;
  }
#undef IA2P_GNF
  return true;
}

// partial must hold B*chunks*G*2 floats. Returns the chunk count it used via *chunks_out when partial == null.
int ia2p_gn_chunks(int B, int HW) {
  static const int cap = ia2p_exp_env("IA2P_GN_STATS_WGS") ? atoi(ia2p_exp_env("IA2P_GN_STATS_WGS")) : 512;      // tuning hook (tools/gn_bench.py)
  int chunks = 1;
  while (chunks < 64 && B * chunks < cap && (HW / (chunks * 2)) >= 8) chunks *= 2;
  return chunks;
}

hipError_t ia2p_launch_groupnorm(const half_t* x, int ldx, half_t* y, int ldy, const half_t* gamma, const half_t* beta,
                                 float* partial, int B, int HW, int C, int G, float eps, int silu, hipStream_t s, const half_t* x2, int ldx2, int Ca) {
  if (x2 && (Ca % 8 || Ca <= 0 || Ca >= C)) return hipErrorInvalidValue;
  GNArgs a;
  a.x2 = x2; a.ldx2 = ldx2; a.Ca = Ca;
  a.x = x; a.y = y; a.gamma = gamma; a.beta = beta; a.partial = partial;
  a.B = B; a.HW = HW; a.C = C; a.G = G; a.ldx = ldx; a.ldy = ldy; a.eps = eps; a.silu = silu;
  a.chunks = ia2p_gn_chunks(B, HW);
  a.rows = (HW + a.chunks - 1) / a.chunks;
  a.chunks = (HW + a.rows - 1) / a.rows;
  const int nvec = C / 8;
  int V = 1;
  while (nvec / V > 256 || nvec % V) ++V;
  if (V > 2 || C % 8 || C % G) return hipErrorInvalidValue;
  const int TX = nvec / V, TY = 256 / TX;
  // apply pass: ~2k workgroups, each thread streaming >= 4 rows
  constexpr int acap = 2048;
  int ablocks = 1;
  while (B * ablocks < acap && HW / (ablocks * 2) >= 4 * TY) ablocks *= 2;
  a.arows = (HW + ablocks - 1) / ablocks;
  ablocks = (HW + a.arows - 1) / a.arows;
  dim3 grid(a.chunks, B), agrid(ablocks, B), block(256);
  const size_t sm1 = (size_t)2 * TY * C * sizeof(float), sm2 = (size_t)G * 2 * sizeof(float) + (size_t)(256 / G) * G * 2 * sizeof(double);
  if (sm1 > 65536) return hipErrorInvalidValue;
  const int wtf = ((ia2p_wt_mask() & 4) && (size_t)HW * ldy * 2 < (size_t)0x7ffffff0) ? 2 : 0;      // write-through y (per batch element: 32-bit offsets)
  if (gn_fused_launch(x, ldx, y, ldy, gamma, beta, B, HW, C, G, eps, silu | wtf, s, x2, ldx2, Ca)) return hipGetLastError();      // one launch, one read of the tensor
  if (V == 1) {
    hipLaunchKernelGGL(gn_stats_kernel<1>, grid, block, sm1, s, a.x, a.y, a.gamma, a.beta, a.partial, a.B, a.HW, a.C, a.G, a.chunks, a.rows, a.arows, a.ldx, a.ldy, a.eps, a.silu, a.x2, a.ldx2, a.Ca);
    hipLaunchKernelGGL(gn_apply_kernel<1>, agrid, block, sm2, s, a.x, a.y, a.gamma, a.beta, a.partial, a.B, a.HW, a.C, a.G, a.chunks, a.rows, a.arows, a.ldx, a.ldy, a.eps, a.silu | wtf, a.x2, a.ldx2, a.Ca);
  } else {
    hipLaunchKernelGGL(gn_stats_kernel<2>, grid, block, sm1, s, a.x, a.y, a.gamma, a.beta, a.partial, a.B, a.HW, a.C, a.G, a.chunks, a.rows, a.arows, a.ldx, a.ldy, a.eps, a.silu, a.x2, a.ldx2, a.Ca);
    hipLaunchKernelGGL(gn_apply_kernel<2>, agrid, block, sm2, s, a.x, a.y, a.gamma, a.beta, a.partial, a.B, a.HW, a.C, a.G, a.chunks, a.rows, a.arows, a.ldx, a.ldy, a.eps, a.silu | wtf, a.x2, a.ldx2, a.Ca);
  }
  return hipGetLastError();
}

// ---- GroupNorm from producer-side column sums (round 5, gn_fold.h) -----------------------------------------------------------------------------------------
// gn_colstats_kernel: the canonical statistics of a tensor whose producer did not emit them (conv_in's output; a launch finished by splitk_reduce_kernel; tiles of
// fewer than 16 rows): out[(slot * C + c)] = {sum, sum of squares} (fp64) of channel c over the `rows` rows of slot `slot` -- fp32 over aligned runs of 16 rows in row
// order, fp64 from there on: the very numbers a GEMM / conv epilogue writes for the same tensor (tests/test_ops_gpu.py compares them to the bit).
// Block = (slot, span of 64 channels): thread (chunk tx < 8, slice ty < 32) takes the runs ty, ty + 32, ...; the slices meet in LDS.
__global__ __launch_bounds__(256) void gn_colstats_kernel(const half_t* x, int ldx, int M, int C, int rows, double* out) {
  __shared__ double2 red[32][64];
  const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;
  const int slot = blockIdx.x, c0 = min(blockIdx.y * 64 + tx * 8, C - 8);      // (C % 8 == 0; the last span of a width that is not a multiple of 64 re-reads the final chunk: not written)
  const int r0 = slot * rows, nseg = rows / 16;
  double ds[8], dq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { ds[e] = 0.0; dq[e] = 0.0; }
  for (int sg = ty; sg < nseg; sg += 32) {
    h8 d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = *(const h8*)(x + (size_t)min(r0 + sg * 16 + i, M - 1) * ldx + c0);      // (clamped, never branched around)
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.f; b[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (r0 + sg * 16 + i < M) {
#pragma unroll
        for (int e = 0; e < 8; ++e) gn_seg16_add((float)d[i][e], a[e], b[e]);
      }
#pragma unroll
    for (int e = 0; e < 8; ++e) { ds[e] += (double)a[e]; dq[e] += (double)b[e]; }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[ty][tx * 8 + e] = make_double2(ds[e], dq[e]);
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.y * 64 + threadIdx.x < C) {
    double a = 0.0, q = 0.0;
    const int nsl = nseg < 32 ? nseg : 32;
    for (int t = 0; t < nsl; ++t) { const double2 v = red[t][threadIdx.x]; a += v.x; q += v.y; }
    ((double2*)out)[(size_t)slot * C + blockIdx.y * 64 + threadIdx.x] = make_double2(a, q);
  }
}
hipError_t ia2p_launch_gn_colstats(const half_t* x, int ldx, int M, int C, int rows, double* out, hipStream_t s) {
  if (C < 8 || C % 8 || ldx % 8 || rows < 16 || rows % 16 || M < 1 || M % rows) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gn_colstats_kernel, dim3(M / rows, (C + 63) / 64), dim3(256), 0, s, x, ldx, M, C, rows, out);
  return hipGetLastError();
}

// gn_apply_stats_kernel: y = [silu](GroupNorm(x)) with the statistics folded from producer-side column sums -- the stand-alone form of what the GroupNorm-fused
// convolution does to its halo images (same fold, same scale / shift, same element formula: gn_fold.h), for consumers that are not halo-staged convolutions and as
// the unfused twin the fused kernel is tested against (bit-identical). x: one tensor or two that are never concatenated ([x0: C0 channels | x1: C - C0]).
// Block = (row block, image): every block folds the image's sums itself (C x slots loads out of L2), then streams its rows with a fixed thread -> channel map.
template <int V>
__global__ __launch_bounds__(256) void gn_apply_stats_kernel(const half_t* x0, int ld0, const half_t* x1, int ld1, half_t* y, int ldy, int HW, int C, int arows, int wt,
                                                                 const GemmArgs::GnIn g) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  double2* chs = (double2*)gsm;                        // [C]
  float2* gstat = (float2*)(gsm + (size_t)C * 16);     // [groups]
  const int b = blockIdx.y;
  gn_channel_sums(g, C, HW, b, threadIdx.x, 256, chs);
  __syncthreads();
  gn_group_stats(g, HW, threadIdx.x, chs, gstat);
  __syncthreads();
  const int nvec = C >> 3, TX = nvec / V, TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  if (ty >= TY) return;
  float sc[V][8], sh[V][8];
  const half_t* src[V]; int sld[V];
#pragma unroll
  for (int v = 0; v < V; ++v) {
    const int c0 = (tx + v * TX) * 8;
    const h8 ga = *(const h8*)(g.gamma + c0), be = *(const h8*)(g.beta + c0);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float2 ab = gn_scale_shift(gstat[(c0 + e) / g.gs], (float)ga[e], (float)be[e]);
      sc[v][e] = ab.x; sh[v][e] = ab.y;
    }
    const bool second = x1 && c0 >= g.C0;
    src[v] = second ? x1 + (size_t)b * HW * ld1 + (c0 - g.C0) : x0 + (size_t)b * HW * ld0 + c0;
    sld[v] = second ? ld1 : ld0;
  }
  const int r0 = blockIdx.x * arows, r1 = min(r0 + arows, HW);
  half_t* obase = y + (size_t)b * HW * ldy;
  const __amdgpu_buffer_rsrc_t y_rsrc = wt_rsrc((void*)obase, (size_t)HW * ldy * 2);
  for (int r = r0 + ty; r < r1; r += 4 * TY) {         // 4 rows in flight per thread
    h8 d[4][V];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int v = 0; v < V; ++v) d[u][v] = *(const h8*)(src[v] + (size_t)min(r + u * TY, r1 - 1) * sld[v]);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (r + u * TY < r1) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          h8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (half_t)gn_apply_f((float)d[u][v][e], sc[v][e], sh[v][e], g.silu != 0);
          if (wt) store16_wt(y_rsrc, ((size_t)(r + u * TY) * ldy + (tx + v * TX) * 8) * 2, o);
          else *(h8*)(obase + (size_t)(r + u * TY) * ldy + (tx + v * TX) * 8) = o;
        }
      }
  }
}
// g: st0 / rows0 (/ st1 / rows1 / C0), gamma, beta, gs, groups, eps, silu as in GemmArgs::GnIn; x1 == nullptr: one source of C channels
hipError_t ia2p_launch_gn_apply_stats(const half_t* x0, int ld0, const half_t* x1, int ld1, half_t* y, int ldy, int B, int HW, int C, const GemmArgs::GnIn& g, hipStream_t s) {
  const int C1 = C - g.C0;
  if (!g.st0 || !g.gamma || !g.beta || C % 8 || g.groups < 1 || g.groups > 64 || g.gs * g.groups != C || g.C0 < 8 || g.C0 % 8 || C1 < 0 || (C1 > 0) != (x1 != nullptr) || (C1 > 0 && (!g.st1 || g.rows1 < 1 || HW % g.rows1)) ||
      g.rows0 < 1 || HW % g.rows0 || ld0 % 8 || ldy % 8 || (x1 && ld1 % 8))
    return hipErrorInvalidValue;
  const int nvec = C / 8;
  int V = 1;
  while (nvec / V > 256 || nvec % V) ++V;
  if (V > 2) return hipErrorInvalidValue;
  const int TX = nvec / V, TY = 256 / TX;
  int ablocks = 1;
  while (B * ablocks < 2048 && HW / (ablocks * 2) >= 4 * TY) ablocks *= 2;
  const int arows = (HW + ablocks - 1) / ablocks;
  ablocks = (HW + arows - 1) / arows;
  const size_t sm = (size_t)C * 16 + 64 * sizeof(float2);
  const int wt = ((ia2p_wt_mask() & 4) && (size_t)HW * ldy * 2 < (size_t)0x7ffffff0) ? 1 : 0;
  if (V == 1) {
    if (sm > 48 * 1024 && hipFuncSetAttribute((const void*)gn_apply_stats_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return hipGetLastError();
    hipLaunchKernelGGL(gn_apply_stats_kernel<1>, dim3(ablocks, B), dim3(256), sm, s, x0, ld0, x1, ld1, y, ldy, HW, C, arows, wt, g);
  } else {
    if (sm > 48 * 1024 && hipFuncSetAttribute((const void*)gn_apply_stats_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm) != hipSuccess) return hipGetLastError();
    hipLaunchKernelGGL(gn_apply_stats_kernel<2>, dim3(ablocks, B), dim3(256), sm, s, x0, ld0, x1, ld1, y, ldy, HW, C, arows, wt, g);
  }
  return hipGetLastError();
}

// ---- LayerNorm over the last dim of x[M, C]: one wave per row, row held in registers (true two-pass stats)
template <int NV>  // NV = ceil((C/8)/64) vectors of 8 per lane
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* x, int ldx, half_t* y, int ldy, const half_t* gamma,
                                                        const half_t* beta, int M, int C, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = blockIdx.x * 4 + wave;
  if (m >= M) return;
  const int nvec = C >> 3;
  const half_t* row = x + (size_t)m * ldx;
  float v[NV][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
      const h8 d = *(const h8*)(row + c * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[i][e] = (float)d[e]; sum += v[i][e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i)
    if (lane + i * 64 < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[i][e] - mean; sq += d * d; }
    }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  half_t* orow = y + (size_t)m * ldy;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < nvec) {
      const h8 ga = *(const h8*)(gamma + c * 8), be = *(const h8*)(beta + c * 8);
      h8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)((v[i][e] - mean) * rstd * (float)ga[e] + (float)be[e]);
      *(h8*)(orow + c * 8) = o;
    }
  }
}

hipError_t ia2p_launch_layernorm(const half_t* x, int ldx, half_t* y, int ldy, const half_t* gamma, const half_t* beta,
                                 int M, int C, float eps, hipStream_t s) {
  if (C % 8) return hipErrorInvalidValue;
  const int nv = (C / 8 + 63) / 64;
  dim3 grid((M + 3) / 4), block(256);
  switch (nv) {
    case 1: hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, ldx, y, ldy, gamma, beta, M, C, eps); break;
    case 2: hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, ldx, y, ldy, gamma, beta, M, C, eps); break;
    case 3: hipLaunchKernelGGL(layernorm_kernel<3>, grid, block, 0, s, x, ldx, y, ldy, gamma, beta, M, C, eps); break;
    case 4: hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, ldx, y, ldy, gamma, beta, M, C, eps); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}
