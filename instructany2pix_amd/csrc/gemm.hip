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
#include "common.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define GLDS16(gptr, ldsptr)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                   (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
  // counted wait for this wave's LDS-DMA pieces + workgroup barrier, as ONE opaque statement: the "memory" clobber
  // keeps the compiler from moving LDS reads / DMA issues across it
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// BM x BN tile; WGM x 2 waves, each owning a (BM/WGM) x (BN/2) sub-tile
// LDS image of a k-tile: rows of ROWB = 2*BK bytes, 16-byte chunks XOR-swizzled so that the ds_read_b128 fragment reads of
// v_mfma_f32_16x16x32_f16 (16 rows x one chunk per 16-lane group) are bank-conflict free:
//   BK = 64 (8 chunks/row):  chunk ^ ((row >> 1) & 7)        BK = 32 (4 chunks/row):  chunk ^ ((-(row >> 2)) & 3)
template <int BK> __device__ __forceinline__ int lds_swz(int row) { return BK == 64 ? (row >> 1) & 7 : (-(row >> 2)) & 3; }

template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int BK = 64>
__global__ __launch_bounds__(WGM * 128, 2) void gemm_f16_kernel(const GemmArgs p) {   // >= 2 waves/SIMD: big tiles must fit 256 registers
  constexpr int NWAVE = WGM * 2;
  constexpr int WM = BM / WGM, WN = BN / 2;      // wave tile (waves arranged WGM x 2)
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int ROWB = 2 * BK, CPR = ROWB / 16, RPP = 1024 / ROWB;   // row bytes, chunks per row, rows per 1-KiB staging piece
  constexpr int A_PW = BM / RPP / NWAVE, B_PW = BN / RPP / NWAVE;    // staging pieces per wave
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int KSUB = BK / 32;                                       // 32-deep MFMA sub-steps per k-tile
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- tile of this workgroup; blocks b, b+8, ... share an XCD (its L2): give each XCD a contiguous tile range
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
  int bid = blockIdx.x;
  const int nsplit = p.splitk > 1 ? p.splitk : 1;
  const int split = bid / (tiles_m * tiles_n);          // >= nsplit: prefetch workgroup
  if (split < nsplit) bid -= split * tiles_m * tiles_n;
  if (split >= nsplit) {   // prefetch workgroup: touch its slice of the next kernel's weights and leave
    bid -= nsplit * tiles_m * tiles_n;
    const long per = ((p.pf_bytes + p.pf_blocks - 1) / p.pf_blocks + 4095) & ~4095L;
    const long lo = (long)bid * per, hi = min(lo + per, p.pf_bytes & ~15L);
    const char* src = (const char*)p.pf;
    unsigned acc = 0;
    constexpr long SW = NWAVE * 64 * 16;   // bytes swept by the workgroup per pass
    for (long o = lo + tid * 16; o < hi; o += 4 * SW) {
      unsigned v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const unsigned*)(src + min(o + u * SW, hi - 16));   // one dword per 16-B slot pulls the whole line
#pragma unroll
      for (int u = 0; u < 4; ++u) acc ^= v[u];
    }
    asm volatile("" ::"v"(acc));    // keep the loads alive
    return;
  }
  {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  int tm, tn;
  if (p.m_fastest) { tn = bid / tiles_m; tm = bid - tn * tiles_m; }
  else             { tm = bid / tiles_n; tn = bid - tm * tiles_n; }
  const int bm0 = tm * BM, bn0 = tn * BN;

  // ---- staging addresses. Piece `pi` covers tile rows pi*8 .. pi*8+7; lane -> (row pi*8 + lane/8, LDS chunk lane%8),
  //      which must hold global chunk (lane%8) ^ swz(row), swz(row) = (row>>1)&7.
  const int srow = lane / CPR, cpos = lane % CPR;
  const half_t* a_ptr[A_PW];
  int a_inc[A_PW];
  int a_y[A_PW], a_x[A_PW], a_pix[A_PW], a_ch[A_PW];
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int pi = wave * A_PW + i;
    const int m = bm0 + pi * RPP + srow;
    const int gch = cpos ^ lds_swz<BK>(pi * RPP + srow);
    if (!CONV) {
      if (m < p.M) {
        int src = m;
        if (p.rpb) { const int b = m / p.rpb; src = b * p.bstride + (m - b * p.rpb) + p.roff; }
        a_ptr[i] = p.A + (size_t)src * p.lda + gch * 8;
        a_inc[i] = BK;
      } else { a_ptr[i] = p.zero; a_inc[i] = 0; }
    } else {
      a_ch[i] = gch * 8;
      if (m < p.M) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_y[i] = oy * p.stride - p.pad; a_x[i] = ox * p.stride - p.pad; a_pix[i] = b * p.Hs * p.Ws;
      } else { a_y[i] = -(1 << 20); a_x[i] = 0; a_pix[i] = 0; }
    }
  }
  const half_t* w_ptr[B_PW];
  int w_inc[B_PW];
#pragma unroll
  for (int i = 0; i < B_PW; ++i) {
    const int pi = wave * B_PW + i;
    const int n = bn0 + pi * RPP + srow;
    const int gch = cpos ^ lds_swz<BK>(pi * RPP + srow);
    if (n < p.N) { w_ptr[i] = p.W + (size_t)n * p.ldw + gch * 8; w_inc[i] = BK; }
    else         { w_ptr[i] = p.zero; w_inc[i] = 0; }
  }

  const int Hv = p.Hs << p.up, Wv = p.Ws << p.up;
  const int nk_all = p.K / BK;
  const int kt0 = (int)((long)split * nk_all / nsplit), kt1 = (int)((long)(split + 1) * nk_all / nsplit);   // this workgroup's k-tiles
  int tap = (kt0 * BK) / (CONV ? p.Cin : BK), ci0 = CONV ? (kt0 * BK) % p.Cin : 0;  // conv: position of the k-tile being staged
  bool tap_fresh = true;
  if (kt0) {             // split-K: this workgroup starts at k-tile kt0
    if (!CONV) {
#pragma unroll
      for (int i = 0; i < A_PW; ++i) a_ptr[i] += (size_t)kt0 * a_inc[i];
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) w_ptr[i] += (size_t)kt0 * w_inc[i];
  }

  auto stage = [&](int kt, int buf) {
    char* sA = smem + buf * STAGE + wave * (A_PW * 1024);
    char* sB = smem + buf * STAGE + BM * ROWB + wave * (B_PW * 1024);
    if (CONV) {
      if (tap_fresh) {        // (wave-uniform) new filter tap: re-derive the gathered pixel of each row once per Cin/64 k-steps
        const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
        for (int i = 0; i < A_PW; ++i) {
          const int iy = a_y[i] + ky, ix = a_x[i] + kx;
          const bool ok = (unsigned)iy < (unsigned)Hv && (unsigned)ix < (unsigned)Wv;
          a_ptr[i] = ok ? p.A + (size_t)(a_pix[i] + (iy >> p.up) * p.Ws + (ix >> p.up)) * p.lda + ci0 + a_ch[i] : p.zero;
          a_inc[i] = ok ? BK : 0;
        }
        tap_fresh = false;
      }
#pragma unroll
      for (int i = 0; i < A_PW; ++i) { GLDS16(a_ptr[i], sA + i * 1024); a_ptr[i] += a_inc[i]; }
      ci0 += BK;
      if (ci0 >= p.Cin) { ci0 = 0; ++tap; tap_fresh = true; }
    } else {
#pragma unroll
      for (int i = 0; i < A_PW; ++i) { GLDS16(a_ptr[i], sA + i * 1024); a_ptr[i] += a_inc[i]; }   // running pointers: no per-step multiply
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) { GLDS16(w_ptr[i], sB + i * 1024); w_ptr[i] += w_inc[i]; }
  };

  // ---- fragment read offsets (wave tile origin is a multiple of 16, so swz(row) = (lane>>1)&7)
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;   // wave / 2 in [0, WGM)
  const int frow = lane & 15, fq = lane >> 4;
  const int fswz = lds_swz<BK>(frow);
  const int a_off = (wm0 + frow) * ROWB, w_off = BM * ROWB + (wn0 + frow) * ROWB;

  f4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  const int nk = kt1 - kt0;
  constexpr int LPS = A_PW + B_PW;   // LDS-DMA pieces this wave issues per k-tile
  // NSTAGE-deep LDS ring: tiles kt+1 .. kt+NSTAGE-2 stay in flight across the barrier of step kt (counted vmcnt,
  // raw s_barrier -- cdna_hip_programming.md §5 "Pipelining across barriers"); ONE barrier per k-step.
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s, s);
  int cur = 0, nxt = NSTAGE - 1;      // ring slots: `cur` is consumed this step, `nxt` is refilled
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;    // tiles issued after tile kt that may remain in flight
    if (NSTAGE >= 3 && ahead >= 1) wait_vm_barrier<LPS>();
    else wait_vm_barrier<0>();
    // every wave has passed the barrier => tile kt has landed for all, and slot `nxt` (read in step kt-1) is free
    if (kt + NSTAGE - 1 < nk) stage(kt + NSTAGE - 1, nxt);
    const char* base = smem + cur * STAGE;
    if constexpr (MR * NR <= 16) {
      // all fragment reads of the k-step are issued before the first MFMA
      h8 af[KSUB][MR], wf[KSUB][NR];
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk) {
        const int coff = ((kk * 4 + fq) ^ fswz) << 4;
#pragma unroll
        for (int i = 0; i < MR; ++i) af[kk][i] = *(const h8*)(base + a_off + i * 16 * ROWB + coff);
#pragma unroll
        for (int j = 0; j < NR; ++j) wf[kk][j] = *(const h8*)(base + w_off + j * 16 * ROWB + coff);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep hipcc from sinking the reads back between the MFMAs
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
    } else {
      // big wave tiles: registers go to accumulators, fragments are read per 32-deep half
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk) {
        const int coff = ((kk * 4 + fq) ^ fswz) << 4;
        h8 af[MR], wf[NR];
#pragma unroll
        for (int i = 0; i < MR; ++i) af[i] = *(const h8*)(base + a_off + i * 16 * ROWB + coff);
#pragma unroll
        for (int j = 0; j < NR; ++j) wf[j] = *(const h8*)(base + w_off + j * 16 * ROWB + coff);
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
    cur = cur + 1 == NSTAGE ? 0 : cur + 1;
    nxt = nxt + 1 == NSTAGE ? 0 : nxt + 1;
  }

  // ---- epilogue: acc[i][j][r] = C[m = bm0+wm0+16i+(lane&15)][n = bn0+wn0+16j+4*(lane>>4)+r]
  if (nsplit > 1) {      // raw fp32 slab of this K range; splitk_reduce_kernel finishes
    float* slab = p.partial + (size_t)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int m = bm0 + wm0 + i * 16 + frow;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = bn0 + wn0 + j * 16 + fq * 4;
        if (n < p.N) *(f4*)(slab + (size_t)m * p.N + n) = acc[i][j];
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const int m = bm0 + wm0 + i * 16 + frow;
    if (m >= p.M) continue;
    const half_t* rv = nullptr;
    if (p.rowvec) rv = p.rowvec + (size_t)(m / p.rows_per_batch) * p.rowvec_ld;
    if (!p.geglu) {
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int n = bn0 + wn0 + j * 16 + fq * 4;
        if (n >= p.N) continue;
        f4 v = acc[i][j];
        if (p.bias) { const h4 b = *(const h4*)(p.bias + n); v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3]; }
        if (rv) { const h4 b = *(const h4*)(rv + n); v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3]; }
        if (p.residual) { const h4 b = *(const h4*)(p.residual + (size_t)m * p.ldr + n); v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3]; }
        h4 o; o[0] = (half_t)v[0]; o[1] = (half_t)v[1]; o[2] = (half_t)v[2]; o[3] = (half_t)v[3];
        *(h4*)(p.C + (size_t)m * p.ldc + n) = o;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NR; j += 2) {
        const int n = bn0 + wn0 + j * 16 + fq * 4;     // packed row of the `a` half; gate rows sit 16 further
        if (n >= p.N) continue;
        const h4 ba = *(const h4*)(p.bias + n), bg = *(const h4*)(p.bias + n + 16);
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = (half_t)((acc[i][j][r] + (float)ba[r]) * gelu_erf_f(acc[i][j + 1][r] + (float)bg[r]));
        const int nout = ((bn0 + wn0 + j * 16) >> 1) + fq * 4;
        *(h4*)(p.C + (size_t)m * p.ldc + nout) = o;
      }
    }
  }
}

template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int BK = 64>
static hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
  constexpr int smem = NSTAGE * (BM + BN) * 2 * BK;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_f16_kernel<BM, BN, NSTAGE, CONV, WGM, BK>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const int extra = (a.pf && a.pf_bytes >= 4096) ? a.pf_blocks : 0;
  hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, NSTAGE, CONV, WGM, BK>), dim3(tiles * (a.splitk > 1 ? a.splitk : 1) + extra), dim3(WGM * 128), smem, s, a);
  return hipGetLastError();
}

// variant id = tile * 2 + (stages - 2); tile 0: 128x128, 1: 128x64, 2: 64x64; stages 2..3
// Measured on MI355X (tools/gemm_bench.py): occupancy beats ring depth -- a third stage costs a resident block
// (96 KiB LDS at 128x128) and loses 20-30 %, so 2 stages is the default; tile = largest that still gives every
// CU >= 1.5-2 workgroups (the kernels run at ~13 TB/s of L2->LDS traffic, i.e. they are L2-bandwidth bound and
// more co-resident blocks hide the per-k-step load latency).
static int g_force_variant = -1;   // test/tuning hook (ia2p_debug_set_gemm_tile)
extern "C" void ia2p_debug_set_gemm_tile(int v) { g_force_variant = v; }

// tuning hook: IA2P_GEMM_RULES="MxNxK=variant;..." overrides the choice for exact shapes (in-situ A/B runs of bench.py)
struct ShapeRule { int M, N, K, v; };
static const std::vector<ShapeRule>& shape_rules() {
  static std::vector<ShapeRule> rules = [] {
    std::vector<ShapeRule> r;
    if (const char* e = getenv("IA2P_GEMM_RULES")) {
      const char* p = e;
      while (*p) {
        ShapeRule x;
        int n = 0;
        if (sscanf(p, "%dx%dx%d=%d%n", &x.M, &x.N, &x.K, &x.v, &n) == 4) { r.push_back(x); p += n; }
        while (*p && *p != ';') ++p;
        if (*p == ';') ++p;
      }
    }
    return r;
  }();
  return rules;
}

// C[m,n] = sum_s partial[s][m][n] + bias + rowvec + residual, fixed summation order
__global__ void splitk_reduce_kernel(const GemmArgs p) {
  const long total = (long)p.M * (p.N >> 2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / (p.N >> 2)), n = (int)(i - (long)m * (p.N >> 2)) * 4;
    f4 v = *(const f4*)(p.partial + (size_t)m * p.N + n);
    for (int s = 1; s < p.splitk; ++s) {
      const f4 w = *(const f4*)(p.partial + ((size_t)s * p.M + m) * p.N + n);
      v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
    }
    if (p.bias) { const h4 b = *(const h4*)(p.bias + n); v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3]; }
    if (p.rowvec) { const h4 b = *(const h4*)(p.rowvec + (size_t)(m / p.rows_per_batch) * p.rowvec_ld + n); v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3]; }
    if (p.residual) { const h4 b = *(const h4*)(p.residual + (size_t)m * p.ldr + n); v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3]; }
    h4 o; o[0] = (half_t)v[0]; o[1] = (half_t)v[1]; o[2] = (half_t)v[2]; o[3] = (half_t)v[3];
    *(h4*)(p.C + (size_t)m * p.ldc + n) = o;
  }
}

static int g_force_splitk = -1;    // test/tuning hook
extern "C" void ia2p_debug_set_gemm_splitk(int s) { g_force_splitk = s; }

// Tile variant and split-K factor for a problem. Pure function of the shape (the executor sizes its workspace with it).
GemmPlan ia2p_gemm_plan(int M, int N, int K, bool conv, bool geglu) {
  auto tiles = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
  GemmPlan pl{-1, 1};
  for (const ShapeRule& r : shape_rules())
    if (r.M == M && r.N == N && r.K == K) pl.variant = r.v;
  if (g_force_variant >= 0) pl.variant = g_force_variant;
  const int nk = K / 64;
  if (pl.variant < 0) {
    int tile;
    if (M <= 64) tile = 2;
    else if (tiles(128, 128) >= 384) tile = (N % 128 != 0 && N % 64 == 0) ? 1 : 0;   // N = 320: exact 5 x 64 columns
    else if (tiles(128, 64) >= 512) tile = 1;
    else tile = 2;
    pl.variant = tile * 2;
    // long-K problems with too few tiles to fill 256 CUs: split K over workgroups (deterministic slab reduce).
    //  * 3x3 convs at 16x16 (2048 x 1280 x 11520..23040): 160 tiles of 128x128, 3 K-slices each
    //  * small-batch projections (M <= 512): 64x64 tiles, K sliced so that >= ~256 workgroups exist
    if (!geglu) {
      if (conv && tiles(128, 128) >= 96 && tiles(128, 128) < 256 && nk >= 96) { pl.variant = 0; pl.splitk = 3; }
      else if (tiles(64, 64) < 160 && nk >= 16) {
        int s = (int)std::min<long>(8, (320 + tiles(64, 64) - 1) / tiles(64, 64));
        while (s > 1 && nk / s < 4) --s;
        pl.variant = 4; pl.splitk = s;
      }
    }
  }
  if (g_force_splitk >= 1 && !geglu) pl.splitk = g_force_splitk;
  if (pl.splitk > nk) pl.splitk = nk;
  if (pl.splitk < 1) pl.splitk = 1;
  return pl;
}

template <bool CONV>
static hipError_t launch_any(const GemmArgs& a, hipStream_t s, int* picked) {
  const GemmPlan pl = ia2p_gemm_plan(a.M, a.N, a.K, CONV, a.geglu != 0);
  const int v = pl.variant;
  if (a.splitk > 1 && !a.partial) return hipErrorInvalidValue;
  if (picked) *picked = v;
  hipError_t e;
  switch (v) {
    case 0: e = launch_cfg<128, 128, 2, CONV>(a, s); break;
    case 1: e = launch_cfg<128, 128, 3, CONV>(a, s); break;
    case 2: e = launch_cfg<128, 64, 2, CONV>(a, s); break;
    case 3: e = launch_cfg<128, 64, 3, CONV>(a, s); break;
    case 4: e = launch_cfg<64, 64, 2, CONV>(a, s); break;
    case 5: e = launch_cfg<64, 64, 3, CONV>(a, s); break;
    // Experimental tiles measured and dropped in round 1 (tools/gemm_bench.py): 8-wave 256x128 (2- and 3-stage) and
    // 256x320 at one workgroup per CU; BK = 32 rings (256x128 / 128x128, more co-resident blocks) -- 64-byte rows halve the
    // request efficiency. The kernel template still takes WGM and BK for later rounds.
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess || a.splitk <= 1) return e;
  const long total = (long)a.M * (a.N >> 2);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((int)std::min<long>(2048, (total + 255) / 256)), dim3(256), 0, s, a);
  return hipGetLastError();
}

// *picked (optional) receives the variant id. a.splitk / a.partial must follow ia2p_gemm_plan (the caller owns the slabs).
hipError_t ia2p_launch_gemm(const GemmArgs& a, bool conv, hipStream_t s, int* picked) {
  return conv ? launch_any<true>(a, s, picked) : launch_any<false>(a, s, picked);
}
