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
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

bool ia2p_splitk_inkernel(int M, int N, int splitk);

#define GLDS16(gptr, ldsptr)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                   (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
  // counted wait for this wave's LDS-DMA pieces + workgroup barrier, as ONE opaque statement: the "memory" clobber
  // keeps the compiler from moving LDS reads / DMA issues across it
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// wait until all but `tiles` (0 .. MAXT, wave-uniform) k-tiles of LPS pieces each have landed, then the workgroup barrier
template <int MAXT, int LPS> __device__ __forceinline__ void wait_ring(int tiles) {
  static_assert(MAXT * LPS <= 63, "vmcnt immediate");
  if constexpr (MAXT <= 0) wait_vm_barrier<0>();
  else {
    if (tiles >= MAXT) wait_vm_barrier<MAXT * LPS>();
    else wait_ring<MAXT - 1, LPS>(tiles);
  }
}

// BM x BN tile; WGM x 2 waves, each owning a (BM/WGM) x (BN/2) sub-tile
// LDS image of a k-tile: rows of ROWB = 2*BK bytes, 16-byte chunks XOR-swizzled so that the ds_read_b128 fragment reads of
// v_mfma_f32_16x16x32_f16 (16 rows x one chunk per 16-lane group) are bank-conflict free:
//   BK = 64 (8 chunks/row):  chunk ^ ((row >> 1) & 7)        BK = 32 (4 chunks/row):  chunk ^ ((-(row >> 2)) & 3)
template <int BK> __device__ __forceinline__ int lds_swz(int row) { return BK == 64 ? (row >> 1) & 7 : (-(row >> 2)) & 3; }


// LDS budget of the staged epilogue: the fp32 tile is read out in NCHUNK row chunks so that chunk + row constants (+ statistics partials) stay
// within what two co-resident workgroups can hold (<= 80 KiB each), or within the stage buffers when those are larger
template <int BM, int BN, int NSTAGE, int WGM, int BK, int WGN = 2>
struct EpiCfg {
  static constexpr int BNL = ((BN / (1024 / (2 * BK)) + WGM * WGN - 1) / (WGM * WGN)) * (WGM * WGN) * (1024 / (2 * BK));   // weight rows staged (>= BN)
  static constexpr int STAGE_BYTES = NSTAGE * (BM + BNL) * 2 * BK;
  static constexpr int PITCH = ((BN / 4 + 7) & ~7) * 4;        // floats per fp32 tile row: whole groups of 8 chunks (the XOR swizzle stays inside a group)
  static constexpr bool POW2 = ((BN / 8) & (BN / 8 - 1)) == 0;
  static constexpr int extra(int cr) { return (2 * BM + 2 * BN + 4) * 4 + (POW2 ? 0 : cr * (BN / 8) * 8); }
  static constexpr int LIMIT = STAGE_BYTES > 80 * 1024 ? STAGE_BYTES : 80 * 1024;
  static constexpr int NCHUNK = (BM * PITCH * 4 + extra(BM) <= LIMIT) ? 1 : 2;
  static_assert(WGM % NCHUNK == 0, "a chunk holds whole wave rows");
  static constexpr int CR = BM / NCHUNK;
  static constexpr int TILE_BYTES = CR * PITCH * 4;
  static_assert(TILE_BYTES + extra(CR) <= LIMIT, "epilogue staging does not fit");
  static constexpr int SMEM = STAGE_BYTES > TILE_BYTES + extra(CR) ? STAGE_BYTES : TILE_BYTES + extra(CR);
};

// PP = 1 ("ping-pong", 8 waves = WGM 4, 3-stage ring, ONE workgroup per CU): waves 0-3 own the upper half of the tile rows, waves 4-7 the
// lower half, and the two groups run half a k-step apart -- while one group reads its fragments from LDS the other issues its MFMAs, with a
// workgroup barrier between the half-steps. Eight waves behind one barrier per k-step would all read, then all multiply (the LDS and the
// MFMA phases add up); two independent workgroups per CU de-phase by themselves but need twice the LDS fill per flop
// (profiles/r01g_gemm_loop_ablation.txt: the fill is the largest term of the 128x128 kernel).
template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int BK = 64, int PP = 0, int WGN = 2>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void gemm_f16_kernel(const half_t* hA, const half_t* hW, const half_t* hzero, int hM, int hN, int hK, int hlda, int hldw, int hrpb, int hbstride,
                                                                         int hroff, int hsplitk, int hgroup_w, const GemmArgs p) {
  // The leading 16 dwords of the argument list are what the prologue needs; built with -amdgpu-kernarg-preload-count=16 the command processor
  // hands them over in SGPRs, so the first tile loads go out without waiting for a cold read of the argument block (which costs every launch
  // ~1 us: tools/micro/launch_floor2.hip). The rest of GemmArgs (epilogue, conv geometry) arrives while those loads fly.   // >= 2 waves/SIMD: big tiles must fit 256 registers
  static_assert(!PP || (WGM == 4 && NSTAGE == 3), "ping-pong schedule: 8 waves, 3-stage ring");
  constexpr int NWAVE = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;    // wave tile (waves arranged WGM x WGN; WGN = 1: narrow tiles, one wave per 128-byte column block)
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int ROWB = 2 * BK, CPR = ROWB / 16, RPP = 1024 / ROWB;   // row bytes, chunks per row, rows per 1-KiB staging piece
  constexpr int A_PW = BM / RPP / NWAVE, B_PW = (BN / RPP + NWAVE - 1) / NWAVE;    // staging pieces per wave (B rounded up: the surplus rows read the zero page)
  constexpr int BNL = B_PW * NWAVE * RPP;                            // weight rows held in LDS (>= BN)
  static_assert(BM % (RPP * NWAVE) == 0 && BN % 16 == 0 && WN % 16 == 0 && WM % 16 == 0, "tile / wave layout");
  constexpr int STAGE = (BM + BNL) * ROWB;
  constexpr int KSUB = BK / 32;                                       // 32-deep MFMA sub-steps per k-tile
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- tile of this workgroup; blocks b, b+8, ... share an XCD (its L2): give each XCD a contiguous tile range
  const int tiles_m = (hM + BM - 1) / BM, tiles_n = (hN + BN - 1) / BN;
  int bid = blockIdx.x;
  const int nsplit = hsplitk > 1 ? hsplitk : 1;
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
  if (hgroup_w > 0) {
    // grouped order: column panels of group_w tiles, row-major inside a panel, so that the contiguous range an XCD works on (and the
    // workgroups co-resident on it) cover a compact rows x cols block: the operand panels its L2 has to fetch shrink with the perimeter
    const int per = tiles_m * hgroup_w;
    const int panel = bid / per, r = bid - panel * per;
    const int w = min(hgroup_w, tiles_n - panel * hgroup_w);
    tm = r / w; tn = panel * hgroup_w + (r - tm * w);
  } else if (p.m_fastest) { tn = bid / tiles_m; tm = bid - tn * tiles_m; }
  else                    { tm = bid / tiles_n; tn = bid - tm * tiles_n; }
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
      if (m < hM) {
        int src = m;
        if (hrpb) { const int b = m / hrpb; src = b * hbstride + (m - b * hrpb) + hroff; }
        a_ptr[i] = hA + (size_t)src * hlda + gch * 8;
        a_inc[i] = BK;
      } else { a_ptr[i] = hzero; a_inc[i] = 0; }
    } else {
      a_ch[i] = gch * 8;
      if (m < hM) {
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
    if (n < hN && pi * RPP + srow < BN) { w_ptr[i] = hW + (size_t)n * hldw + gch * 8; w_inc[i] = BK; }
    else         { w_ptr[i] = hzero; w_inc[i] = 0; }
  }

  const int Hv = p.Hs << p.up, Wv = p.Ws << p.up;
  const int nk_all = hK / BK;
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
          a_ptr[i] = ok ? hA + (size_t)(a_pix[i] + (iy >> p.up) * p.Ws + (ix >> p.up)) * hlda + ci0 + a_ch[i] : hzero;
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
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int frow = lane & 15, fq = lane >> 4;
  const int fswz = lds_swz<BK>(frow);
  const int a_off = (wm0 + frow) * ROWB, w_off = BM * ROWB + (wn0 + frow) * ROWB;

  f4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};

  // ---- folded LayerNorm (consumer): thread r < BM collects the {sum, sum of squares} partials of tile row r. Issued behind the
  //      first tile loads, all slots in flight at once (slot order kept in the sums); turned into mean / rstd after the k-loop.
  //      Ping-pong tile: loaded AHEAD of the first tiles and folded at once (48 registers carried through the loop would spill, and a spill
  //      reload in the loop waits for vmcnt, i.e. drains the DMA queue); the prologue DMA stays in flight behind them.
  float ln_s1 = 0.f, ln_s2 = 0.f;
  auto load_ln = [&]() {
  if (p.ln_stats && tid < BM && bm0 + tid < hM) {
    const float2* st = (const float2*)p.ln_stats + (bm0 + tid);
    constexpr int MAXS = 24;
    if (p.ln_slots <= MAXS) {
      float2 v[MAXS];
#pragma unroll
      for (int u = 0; u < MAXS; ++u) v[u] = st[(size_t)min(u, p.ln_slots - 1) * hM];
#pragma unroll
      for (int u = 0; u < MAXS; ++u)
        if (u < p.ln_slots) { ln_s1 += v[u].x; ln_s2 += v[u].y; }
    } else {
      for (int sl = 0; sl < p.ln_slots; ++sl) { const float2 v = st[(size_t)sl * hM]; ln_s1 += v.x; ln_s2 += v.y; }
    }
  }
  };
  if (PP) load_ln();
  const int nk = kt1 - kt0;
  constexpr int LPS = A_PW + B_PW;   // LDS-DMA pieces this wave issues per k-tile
  // NSTAGE-deep LDS ring: tiles kt+1 .. kt+NSTAGE-2 stay in flight across the barrier of step kt (counted vmcnt,
  // raw s_barrier -- cdna_hip_programming.md §5 "Pipelining across barriers"); ONE barrier per k-step.
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < nk) stage(s, s);
  if (!PP) load_ln();
  else asm volatile("" : "+v"(ln_s1), "+v"(ln_s2));     // folds now; the counted wait leaves the prologue DMA in flight
  if constexpr (PP) {
    // Barrier sequence b0, b1, ...; interval I_n lies between b_n and b_n+1. Group 0 reads tile t in I_2t and multiplies it in I_2t+1; group 1
    // reads it in I_2t+1 and multiplies it in I_2t+2. Every wave waits for its DMA pieces of tile t before b_2t; the slot of tile t-1 is free
    // after b_2t (group 1 finished reading it in I_2t-1), so tile t+2 is issued into it in I_2t: two tiles stay in flight.
    static_assert(MR * NR <= 16, "ping-pong keeps the fragments of a whole k-tile in registers across a barrier");
    const int grp = wave >> 2;
    h8 af[KSUB][MR], wf[KSUB][NR];
    auto rd = [&](int slot) {
      const char* base = smem + slot * STAGE;
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk) {
        const int coff = ((kk * 4 + fq) ^ fswz) << 4;
#pragma unroll
        for (int i = 0; i < MR; ++i) af[kk][i] = *(const h8*)(base + a_off + i * 16 * ROWB + coff);
#pragma unroll
        for (int j = 0; j < NR; ++j) wf[kk][j] = *(const h8*)(base + w_off + j * 16 * ROWB + coff);
      }
    };
    auto mm = [&]() {
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
    };
    // One loop per group (straight-line bodies: a shared loop with per-group arms makes the compiler shuffle the 128 fragment / accumulator
    // registers between the arms every iteration). Both loops pass exactly two barriers per k-tile.
    auto top = [&](int t) {      // b_2t: tile t has landed for every wave (tile t+1 may still be in flight)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) wait_vm_barrier<LPS>();
      else wait_vm_barrier<0>();
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mid = [&]() {           // b_2t+1
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    int slot_r = 0, slot_s = NSTAGE - 1;
    auto adv = [&]() { slot_r = slot_r + 1 == NSTAGE ? 0 : slot_r + 1; slot_s = slot_s + 1 == NSTAGE ? 0 : slot_s + 1; };
    // a group issues its DMA pieces of tile t+2 behind the fragment reads of its READ half-step
    if (grp == 0) {
      for (int t = 0; t < nk; ++t) {
        top(t);
        rd(slot_r);
        __builtin_amdgcn_sched_barrier(0);
        if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1, slot_s);
        mid();
        mm();
        adv();
      }
    } else {
      for (int t = 0; t < nk; ++t) {
        top(t);
        if (t > 0) mm();
        mid();
        rd(slot_r);
        __builtin_amdgcn_sched_barrier(0);
        if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1, slot_s);
        adv();
      }
      mm();
    }
  } else {
  int cur = 0, nxt = NSTAGE - 1;      // ring slots: `cur` is consumed this step, `nxt` is refilled
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;    // tiles issued after tile kt that may remain in flight
    // tiles kt+1 .. kt+NSTAGE-2 were issued before this wait and may stay in flight (fewer at the tail): vmcnt counts this wave's pieces
    wait_ring<NSTAGE - 2, LPS>(ahead < NSTAGE - 2 ? ahead : NSTAGE - 2);
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

  }
  // Ping-pong tile: no separate prefetch workgroups (a workgroup holds a whole CU's LDS, so they would queue up behind the tiles): every
  // tile workgroup touches its slice of the next contraction's weights here; the loads fly during the epilogue.
  unsigned pfacc = 0;
  if (PP && p.pf) {
    const long nwg = (long)tiles_m * tiles_n * nsplit;
    const long per = ((p.pf_bytes + nwg - 1) / nwg + 255) & ~255L;
    const long lo = (long)blockIdx.x * per, hi = min(lo + per, p.pf_bytes & ~15L);
    const char* src = (const char*)p.pf;
    constexpr long SW = NWAVE * 64 * 16;
    if (lo < hi)
      for (long o = lo + tid * 16; o < hi; o += 8 * SW) {      // 8 independent loads in flight per thread (clamped, never branched around)
        unsigned v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const unsigned*)(src + min(o + u * SW, hi - 16));
#pragma unroll
        for (int u = 0; u < 8; ++u) pfacc ^= v[u];
      }
  }
  // ---- epilogue, staged through LDS. The MFMA layout gives a lane 4 consecutive columns of ONE row (acc[i][j][r] = C[bm0+wm0+16i+(lane&15)]
  //      [bn0+wn0+16j+4(lane>>4)+r]): stored from there, a wave-instruction touches 16 rows x 32 B -- quarter cache lines, and so does every
  //      residual read (profiles/r01g_gemm_loop_ablation.txt: 15 us of a 34 us launch at K -> 0). Instead the fp32 tile goes through the (now
  //      free) stage buffers once: written in the MFMA layout (16-B chunks XOR-swizzled by row & 7: conflict-free ds_write_b128), read back
  //      row-major, 8 columns per thread, so that bias / time-embedding row / folded-LayerNorm constants / residual are 16-B loads and C is
  //      written in whole 128-B lines; everything is still applied to the fp32 accumulator and rounded once.
  using EC = EpiCfg<BM, BN, NSTAGE, WGM, BK, WGN>;
  constexpr int PITCH = EC::PITCH;
  constexpr int NT = NWAVE * 64, CR = EC::CR;                           // threads, tile rows per chunk
  float* tile = (float*)smem;
  float* ln_rows = (float*)(smem + EC::TILE_BYTES);                     // [0, BM): mean, [BM, 2 BM): rstd
  float* ln_cs = ln_rows + 2 * BM;                                      // BN column sums and BN folded biases of this tile
  float* ln_lb = ln_cs + BN;
  int* sk_flag = (int*)(ln_lb + BN);                                    // K-split: the ticket this workgroup drew, broadcast to its waves
  float2* part = (float2*)(ln_lb + BN + 4);                             // row-statistics partials (tile widths whose 8-column groups per row are not a power of two)
  __syncthreads();                    // every wave has finished reading the stage buffers
  if (p.ln_stats) {
    if (tid < BM) {
      const float inv = 1.f / (float)hK;
      const float mean = ln_s1 * inv;
      const float var = fmaxf(ln_s2 * inv - mean * mean, 0.f);
      ln_rows[tid] = mean;
      ln_rows[BM + tid] = rsqrtf(var + p.ln_eps);
    }
    if (tid < BN / 4 && bn0 + tid * 4 < hN) {
      *(f4*)(ln_cs + tid * 4) = *(const f4*)(p.ln_cs + bn0 + tid * 4);
      *(f4*)(ln_lb + tid * 4) = *(const f4*)(p.ln_bias + bn0 + tid * 4);
    }
  }
  const float e_as = p.acc_scale == 0.f ? 1.f : p.acc_scale, e_bs = p.bias_scale == 0.f ? 1.f : p.bias_scale;
  const bool fast = p.vec8 != 0 && (hN & 7) == 0;     // 16-byte accesses everywhere (every shape of the executors); else 8-byte pieces
  auto tl = [&](int r, int c) -> f4 { return *(const f4*)(tile + (size_t)r * PITCH + ((c ^ (r & 7)) << 2)); };
  auto acc_to_tile = [&](int ch) {
    if (wm0 / CR == ch) {
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const int r = wm0 - ch * CR + i * 16 + frow;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const int c = (wn0 + j * 16) / 4 + fq;
          *(f4*)(tile + (size_t)r * PITCH + ((c ^ (r & 7)) << 2)) = acc[i][j];
        }
      }
    }
  };
  bool from_slabs = false;
  if (nsplit > 1) {
    // ---- K-split: this workgroup holds the partial sums of ONE K range. Every K-slice writes its raw fp32 slab (write-through `sc1` stores:
    //      the bytes are in memory-side coherence when the wave's vmcnt drains, no release fence -- cdna_hip_programming.md §5 "In-launch split-K
    //      reduction"); the slice that arrives LAST at the tile's ticket counter adds the slabs up in slab order (deterministic whoever is last)
    //      and runs the epilogue: no reduce launch, no spin (nobody waits for anybody).
    constexpr int GPR = BN / 4;
    const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc((void*)(p.partial + (size_t)split * hM * hN), 0, (int)min((size_t)hM * hN * 4, (size_t)0x7ffffff0), 0x00020000);
#pragma unroll 1
    for (int ch = 0; ch < EC::NCHUNK; ++ch) {
      if (ch) __syncthreads();
      acc_to_tile(ch);
      __syncthreads();
      const int row0 = bm0 + ch * CR;
      for (int idx = tid; idx < CR * GPR; idx += NT) {
        const int r = idx / GPR, g = idx - r * GPR;
        const int m = row0 + r, n = bn0 + g * 4;
        if (m < hM && n < hN) {
          const f4 v = tl(r, g);
          typedef unsigned u4v __attribute__((__vector_size__(4 * sizeof(unsigned))));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), slab, (int)(((size_t)m * hN + n) * 4), 0, 16);      // aux 16 = sc1 (write-through)
        }
      }
    }
    if (!p.sk_counters) { if (PP) asm volatile("" ::"v"(pfacc)); return; }      // finished by a separate splitk_reduce_kernel launch (A/B switch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                       // ... before ONE lane signals for the workgroup
    if (tid == 0) *sk_flag = __hip_atomic_fetch_add(p.sk_counters + (tm * tiles_n + tn), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*sk_flag != nsplit - 1) { if (PP) asm volatile("" ::"v"(pfacc)); return; }
    if (tid == 0) {
      __hip_atomic_store(p.sk_counters + (tm * tiles_n + tn), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (launches are stream-ordered)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's stale lines before the plain loads below
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    from_slabs = true;
  }
#pragma unroll 1
  for (int ch = 0; ch < EC::NCHUNK; ++ch) {
    if (ch || from_slabs) __syncthreads();          // the previous chunk has been read out
    const int row0 = bm0 + ch * CR;
    if (!from_slabs) acc_to_tile(ch);
    else {                            // tile chunk = sum of the K-slice slabs, slab 0 first
      constexpr int GPR = BN / 4;
      for (int idx = tid; idx < CR * GPR; idx += NT) {
        const int r = idx / GPR, g = idx - r * GPR;
        const int m = min(row0 + r, hM - 1), n = min(bn0 + g * 4, hN - 4);
        const float* src = p.partial + (size_t)m * hN + n;
        f4 v = *(const f4*)src;
        for (int sl = 1; sl < nsplit; ++sl) { const f4 w = *(const f4*)(src + (size_t)sl * hM * hN); v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3]; }
        *(f4*)(tile + (size_t)r * PITCH + ((g ^ (r & 7)) << 2)) = v;
      }
    }
    __syncthreads();
    if (p.geglu) {                    // packed columns: 32-wide blocks [16 values | 16 gates]; out[m][n/2] = a * gelu(g)
      constexpr int GPR = BN / 16;    // groups of 8 OUTPUT columns per row
      constexpr int TOTAL = CR * GPR, U = EC::NCHUNK == 1 ? 2 : 1, ITER = (TOTAL + NT * U - 1) / (NT * U);
#pragma unroll 1
      for (int k = 0; k < ITER; ++k) {
        f4 a0[U], a1[U], g0[U], g1[U];
        h8 ba[U], bg[U];
        int rr[U], gg[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {       // all loads of U groups in flight before any arithmetic
          const int idx = tid + (k * U + u) * NT;
          const int r = min(idx / GPR, CR - 1), g = idx - (idx / GPR) * GPR;
          const int ca = (g >> 1) * 8 + (g & 1) * 2;                   // first 16-B chunk of the 8 value columns; the gates sit 4 chunks further
          rr[u] = r; gg[u] = g;
          live[u] = idx < TOTAL && row0 + r < hM && bn0 + ca * 4 < hN;
          a0[u] = tl(r, ca); a1[u] = tl(r, ca + 1); g0[u] = tl(r, ca + 4); g1[u] = tl(r, ca + 5);
          if (!p.ln_stats) {
            const int n = min(bn0 + ca * 4, hN - 24);      // values n .. n+7, gates n+16 .. n+23
            ba[u] = *(const h8*)(p.bias + n); bg[u] = *(const h8*)(p.bias + n + 16);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = rr[u], g = gg[u];
          const int cl = ((g >> 1) * 8 + (g & 1) * 2) * 4;              // tile-local packed column of the first value
          float va[8] = {a0[u][0], a0[u][1], a0[u][2], a0[u][3], a1[u][0], a1[u][1], a1[u][2], a1[u][3]};
          float vg[8] = {g0[u][0], g0[u][1], g0[u][2], g0[u][3], g1[u][0], g1[u][1], g1[u][2], g1[u][3]};
          if (p.ln_stats) {
            const float mu = ln_rows[ch * CR + r], rs = ln_rows[BM + ch * CR + r];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              va[e] = rs * (va[e] - mu * ln_cs[cl + e]) + ln_lb[cl + e];
              vg[e] = rs * (vg[e] - mu * ln_cs[cl + 16 + e]) + ln_lb[cl + 16 + e];
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { va[e] += (float)ba[u][e]; vg[e] += (float)bg[u][e]; }
          }
          h8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = (half_t)(va[e] * gelu_erf_f(vg[e]));
          if (live[u]) *(h8*)(p.C + (size_t)(row0 + r) * p.ldc + (bn0 >> 1) + (g >> 1) * 16 + (g & 1) * 8) = o;
        }
      }
    } else {
      constexpr int GPR = BN / 8;     // groups of 8 columns per row
      constexpr bool POW2 = (GPR & (GPR - 1)) == 0;
      constexpr int TOTAL = CR * GPR, U = EC::NCHUNK == 1 ? 4 : 1, ITER = (TOTAL + NT * U - 1) / (NT * U);   // (two chunks: the second chunk's accumulators are still live)
      static_assert(!POW2 || NT % GPR == 0, "row groups must not straddle waves");
#pragma unroll 1
      for (int k = 0; k < ITER; ++k) {
        f4 x0[U], x1[U];
        h8 hb[U], hv[U], hr[U];
        int rr[U], gg[U];
        bool live[U];
        float st1[U], st2[U];
        if (fast) {
#pragma unroll
          for (int u = 0; u < U; ++u) {     // all loads of U groups in flight before any arithmetic (clamped addresses, never branched around)
            const int idx = tid + (k * U + u) * NT;
            const int r = min(idx / GPR, CR - 1), g = idx - (idx / GPR) * GPR;
            rr[u] = r; gg[u] = g;
            const int m = row0 + r, n = bn0 + g * 8;
            live[u] = idx < TOTAL && m < hM && n < hN;
            const int mc = min(m, hM - 1), nc = min(n, hN - 8);
            x0[u] = tl(r, 2 * g); x1[u] = tl(r, 2 * g + 1);
            if (p.bias && !p.ln_stats) hb[u] = *(const h8*)(p.bias + nc);
            if (p.rowvec) hv[u] = *(const h8*)(p.rowvec + (size_t)(mc / p.rows_per_batch) * p.rowvec_ld + nc);
            if (p.residual) hr[u] = *(const h8*)(p.residual + (size_t)mc * p.ldr + nc);
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int r = rr[u], cl = gg[u] * 8;
            float v[8] = {x0[u][0], x0[u][1], x0[u][2], x0[u][3], x1[u][0], x1[u][1], x1[u][2], x1[u][3]};
            if (p.ln_stats) {
              const float mu = ln_rows[ch * CR + r], rs = ln_rows[BM + ch * CR + r];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = rs * (v[e] - mu * ln_cs[cl + e]) + ln_lb[cl + e];
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= e_as;
              if (p.bias) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf((float)hb[u][e], e_bs, v[e]);
              }
            }
            if (p.act) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = act_f(v[e], p.act);
            }
            if (p.rowvec) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaf((float)hv[u][e], e_bs, v[e]);
            }
            if (p.residual) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += (float)hr[u][e];
            }
            h8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
            if (live[u]) *(h8*)(p.C + (size_t)(row0 + r) * p.ldc + bn0 + cl) = o;
            st1[u] = st2[u] = 0.f;
            if (p.stats_out && live[u]) {
#pragma unroll
              for (int e = 0; e < 8; ++e) { const float f = (float)o[e]; st1[u] += f; st2[u] += f * f; }
            }
          }
        } else {
          // strides / widths that only allow 8-byte accesses (N % 8 == 4, odd leading dimensions): two 4-column halves per group
#pragma unroll 1
          for (int u = 0; u < U; ++u) {
            const int idx = tid + (k * U + u) * NT;
            const int r = min(idx / GPR, CR - 1), g = idx - (idx / GPR) * GPR;
            rr[u] = r; gg[u] = g;
            const int m = row0 + r;
            live[u] = idx < TOTAL && m < hM && bn0 + g * 8 < hN;
            st1[u] = st2[u] = 0.f;
            if (!live[u]) continue;
            for (int hf = 0; hf < 2; ++hf) {
              const int n = bn0 + g * 8 + hf * 4, cl = g * 8 + hf * 4;
              if (n >= hN) break;
              f4 v = tl(r, 2 * g + hf);
              if (p.ln_stats) {
                const float mu = ln_rows[ch * CR + r], rs = ln_rows[BM + ch * CR + r];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rs * (v[e] - mu * ln_cs[cl + e]) + ln_lb[cl + e];
              } else {
                v[0] *= e_as; v[1] *= e_as; v[2] *= e_as; v[3] *= e_as;
                if (p.bias) { const h4 b = *(const h4*)(p.bias + n); v[0] = fmaf((float)b[0], e_bs, v[0]); v[1] = fmaf((float)b[1], e_bs, v[1]); v[2] = fmaf((float)b[2], e_bs, v[2]); v[3] = fmaf((float)b[3], e_bs, v[3]); }
              }
              if (p.act) { v[0] = act_f(v[0], p.act); v[1] = act_f(v[1], p.act); v[2] = act_f(v[2], p.act); v[3] = act_f(v[3], p.act); }
              if (p.rowvec) { const h4 b = *(const h4*)(p.rowvec + (size_t)(m / p.rows_per_batch) * p.rowvec_ld + n); v[0] = fmaf((float)b[0], e_bs, v[0]); v[1] = fmaf((float)b[1], e_bs, v[1]); v[2] = fmaf((float)b[2], e_bs, v[2]); v[3] = fmaf((float)b[3], e_bs, v[3]); }
              if (p.residual) { const h4 b = *(const h4*)(p.residual + (size_t)m * p.ldr + n); v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3]; }
              h4 o; o[0] = (half_t)v[0]; o[1] = (half_t)v[1]; o[2] = (half_t)v[2]; o[3] = (half_t)v[3];
              *(h4*)(p.C + (size_t)m * p.ldc + n) = o;
              if (p.stats_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float f = (float)o[e]; st1[u] += f; st2[u] += f * f; }
              }
            }
          }
        }
        if (p.stats_out) {             // {sum, sum of squares} of the fp16 output row over this tile's columns: ONE partial per row and tile (slot = tile_n)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int idx = tid + (k * U + u) * NT;
            if constexpr (POW2) {
              float a = st1[u], b = st2[u];
#pragma unroll
              for (int o = 1; o < GPR; o <<= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }      // fixed butterfly: deterministic
              if (gg[u] == 0 && idx < TOTAL && row0 + rr[u] < hM) ((float2*)p.stats_out)[(size_t)tn * hM + row0 + rr[u]] = make_float2(a, b);
            } else if (idx < TOTAL) part[idx] = make_float2(st1[u], st2[u]);
          }
        }
      }
      if constexpr (!POW2) {
        if (p.stats_out) {
          __syncthreads();
          if (tid < CR && row0 + tid < hM) {
            float s1 = 0.f, s2 = 0.f;
            for (int g = 0; g < GPR; ++g) { const float2 v = part[tid * GPR + g]; s1 += v.x; s2 += v.y; }      // group order: deterministic
            ((float2*)p.stats_out)[(size_t)tn * hM + row0 + tid] = make_float2(s1, s2);
          }
        }
      }
    }
  }
  if (PP) asm volatile("" ::"v"(pfacc));
}

template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int BK = 64, int PP = 0, int WGN = 2>
static hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
  constexpr int smem = EpiCfg<BM, BN, NSTAGE, WGM, BK, WGN>::SMEM;
  // the attribute is per DEVICE: one flag per device id (several contexts on several GPUs in one process)
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_f16_kernel<BM, BN, NSTAGE, CONV, WGM, BK, PP, WGN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  GemmArgs b = a;
  // 16-byte epilogue accesses need 8-element row strides and 16-byte-aligned bases; otherwise the epilogue falls back to 8-byte pieces
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  b.vec8 = (a.ldc % 8 == 0 && al16(a.C) && (!a.bias || al16(a.bias)) && (!a.residual || (a.ldr % 8 == 0 && al16(a.residual))) &&
            (!a.rowvec || (a.rowvec_ld % 8 == 0 && al16(a.rowvec)))) ? 1 : 0;
  if (a.geglu && !b.vec8) return hipErrorInvalidValue;
  b.sk_counters = nullptr;
  if (ia2p_splitk_inkernel(a.M, a.N, a.splitk)) {
    // ticket counters of the in-launch K-split combine: one int per output tile, zero between launches (the last arriver resets its tile's)
    constexpr int NCNT = 1 << 20;
    static int* cnt[64] = {nullptr};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && tiles <= NCNT) {
      if (!cnt[dev]) {
        if (hipMalloc((void**)&cnt[dev], NCNT * sizeof(int)) != hipSuccess || hipMemset(cnt[dev], 0, NCNT * sizeof(int)) != hipSuccess) { (void)hipGetLastError(); cnt[dev] = nullptr; }
      }
      b.sk_counters = cnt[dev];
    }
  }
  static const int group_mode = getenv("IA2P_TILE_GROUP") ? atoi(getenv("IA2P_TILE_GROUP")) : 1;
  if (group_mode) {
    const int tiles_n = (a.N + BN - 1) / BN;
    const int smem_per_cu = 160 * 1024 / smem;                                   // co-resident workgroups per CU by LDS
    const double resident = std::min<double>(tiles / 8.0, 32.0 * std::max(1, std::min(smem_per_cu, 2)));   // tiles an XCD holds at once
    static const double gscale = getenv("IA2P_TILE_GROUP_SCALE") ? atof(getenv("IA2P_TILE_GROUP_SCALE")) : 1.0;
    int w = (int)(gscale * std::sqrt(resident * BM / BN) + 0.5);
    b.group_w = std::max(1, std::min(w, tiles_n));
  }
  const int extra = (!PP && a.pf && a.pf_bytes >= 4096) ? a.pf_blocks : 0;
  hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, NSTAGE, CONV, WGM, BK, PP, WGN>), dim3(tiles * (a.splitk > 1 ? a.splitk : 1) + extra), dim3(WGM * WGN * 64), smem, s,
                     b.A, b.W, b.zero, b.M, b.N, b.K, b.lda, b.ldw, b.rpb, b.bstride, b.roff, b.splitk, b.group_w, b);
  return hipGetLastError();
}

// variant id = index into IA2P_GEMM_TILES (common.h)
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
  for (int m = blockIdx.x * 4 + wave; m < hM; m += gridDim.x * 4) {
    float mu = 0.f, rs = 1.f;
    if (p.ln_stats) {
      float s1 = 0.f, s2 = 0.f;
      for (int sl = lane; sl < p.ln_slots; sl += 64) { const float2 v = ((const float2*)p.ln_stats)[(size_t)sl * hM + m]; s1 += v.x; s2 += v.y; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
      const float inv = 1.f / (float)p.K;
      mu = s1 * inv;
      rs = rsqrtf(fmaxf(s2 * inv - mu * mu, 0.f) + p.ln_eps);
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
        for (int r = 0; r < 4; ++r) v[r] = rs * (v[r] - mu * cs[r]) + lb[r];
      } else {
        const float e_as = p.acc_scale == 0.f ? 1.f : p.acc_scale, e_bs = p.bias_scale == 0.f ? 1.f : p.bias_scale;
        v[0] *= e_as; v[1] *= e_as; v[2] *= e_as; v[3] *= e_as;
        if (p.bias) { const h4 b = *(const h4*)(p.bias + n); v[0] = fmaf((float)b[0], e_bs, v[0]); v[1] = fmaf((float)b[1], e_bs, v[1]); v[2] = fmaf((float)b[2], e_bs, v[2]); v[3] = fmaf((float)b[3], e_bs, v[3]); }
      }
      if (p.act) { v[0] = act_f(v[0], p.act); v[1] = act_f(v[1], p.act); v[2] = act_f(v[2], p.act); v[3] = act_f(v[3], p.act); }
      if (p.rowvec) { const float e_bs = p.bias_scale == 0.f ? 1.f : p.bias_scale; const h4 b = *(const h4*)(p.rowvec + (size_t)(m / p.rows_per_batch) * p.rowvec_ld + n); v[0] = fmaf((float)b[0], e_bs, v[0]); v[1] = fmaf((float)b[1], e_bs, v[1]); v[2] = fmaf((float)b[2], e_bs, v[2]); v[3] = fmaf((float)b[3], e_bs, v[3]); }
      if (hresidual) { const h4 b = *(const h4*)(hresidual + (size_t)m * hldr + n); v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3]; }
      h4 o; o[0] = (half_t)v[0]; o[1] = (half_t)v[1]; o[2] = (half_t)v[2]; o[3] = (half_t)v[3];
      *(h4*)(hC + (size_t)m * hldc + n) = o;
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

// K-split launches combine their slabs inside the GEMM launch (last-arriving slice of a tile); IA2P_SPLITK_INKERNEL=0: separate
// splitk_reduce_kernel launches instead (A/B switch, read once)
// Which K-split launches combine inside the launch: those whose slabs are small (latency-bound launches: batch 1, tiny contractions), where the
// reduce launch costs more than the last arriver's serial slab read. Measured on one box (profiles/r02e_splitk_inkernel_ab.txt): every K-split
// in-launch: batch 8 +0.6 ms / step, batch 1 -0.16 ms; large slabs are summed faster by a whole-chip reduce launch than by 160 lone workgroups.
// IA2P_SPLITK_INKERNEL = byte threshold on splitk*M*N*4 (0: never, default 8 MiB).
bool ia2p_splitk_inkernel(int M, int N, int splitk) {
  static const size_t limit = getenv("IA2P_SPLITK_INKERNEL") ? (size_t)atoll(getenv("IA2P_SPLITK_INKERNEL")) : ((size_t)8 << 20);
  return splitk > 1 && (size_t)splitk * M * N * sizeof(float) <= limit;
}
static int g_force_splitk = -1;    // test/tuning hook
extern "C" void ia2p_debug_set_gemm_splitk(int s) { g_force_splitk = s; }

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
  const bool pingpong = t.bm == 256;    // 8 waves in two half-step-shifted groups: one workgroup behaves like two co-resident ones
  const double t_mfma = per_cu * (2.0 * t.bm * t.bn * 64) / (MFMA_EFF * (conc == 1 && !pingpong ? LONE_EFF : 1.0) * CU_FLOPS_PER_US);
  const double t_fill = per_cu * ((t.bm + t.bn) * 128.0) / FILL_B_PER_US;
  const double t_lat = batches * LAT_US / (t.stages - 1) * (pingpong ? 0.5 : 1.0);
  double total = BASE_US + nk * std::max({t_mfma, t_fill, t_lat}) + RAMP_US * batches;
  if (sk > 1) total += REDUCE_US + 2.0 * sk * (double)M * N * 4 / REDUCE_B_PER_US;
  return total;
}

// ---- measured plans (ia2p_autotune): shape -> (variant, splitk), process-wide; consulted before the cost model
using PlanKey = std::tuple<int, int, int, int, int>;
static std::mutex g_plan_mu;
static unsigned long long g_plan_gen = 0;      // bumped by every change of the table (hosts re-size their workspace when it moves)
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
  for (const auto& kv : tuned_plans()) {
    snprintf(tmp, sizeof tmp, "%d,%d,%d,%d,%d,%d,%d;", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first),
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
    int M, N, K, cv, gg, v, sk, n = 0;
    if (sscanf(p, "%d,%d,%d,%d,%d,%d,%d;%n", &M, &N, &K, &cv, &gg, &v, &sk, &n) != 7 || n == 0) return -1;
    if (M < 1 || N < 1 || K < 64 || v < 0 || v >= IA2P_GEMM_NVARIANT || sk < 1 || sk > K / 64 || (gg && (sk > 1 || IA2P_GEMM_TILES[v].bn % 32))) return -1;
    in.push_back({PlanKey{M, N, K, cv != 0, gg != 0}, GemmPlan{v, sk}});
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
    if ((geglu && t.bn % 32) || (excluded >> v & 1)) continue;
    for (int sk : splits) {
      if (sk > 1 && (geglu || nk / sk < 4 || (size_t)sk * M * N * 4 > max_slab_bytes)) break;
      all.push_back({plan_cost_us(M, N, K, conv, t, sk), GemmPlan{v, sk}});
    }
  }
  std::stable_sort(all.begin(), all.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
  out->clear();
  for (const auto& e : all)
    if (e.first <= slack * all.front().first) out->push_back(e.second);
}

// Tile variant and split-K factor for a problem: a measured plan if ia2p_autotune recorded one, else the cost model.
// Pure function of the shape and the plan table (the executor sizes its workspace with it).
GemmPlan ia2p_gemm_plan(int M, int N, int K, bool conv, bool geglu) {
  GemmPlan pl{-1, 1};
  if (g_force_variant < 0 && ia2p_plan_lookup(M, N, K, conv, geglu, &pl)) {
    if (g_force_splitk >= 1 && !geglu) pl.splitk = std::min(g_force_splitk, K / 64);
    return pl;
  }
  for (const ShapeRule& r : shape_rules())
    if (r.M == M && r.N == N && r.K == K) pl.variant = r.v;
  if (g_force_variant >= 0) pl.variant = g_force_variant;
  const int nk = K / 64;
  if (pl.variant < 0) {
    static const int splits[] = {1, 2, 3, 4, 6, 8};
    double best = 1e30;
    for (int v = 0; v < IA2P_GEMM_NVARIANT; ++v) {
      const GemmTile& t = IA2P_GEMM_TILES[v];
      if (geglu && t.bn % 32) continue;              // a (value, gate) block of 32 packed columns must not straddle tiles
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
  return pl;
}

extern "C" void ia2p_debug_gemm_plan(int M, int N, int K, int conv, int geglu, int* variant, int* splitk) {
  const GemmPlan pl = ia2p_gemm_plan(M, N, K, conv != 0, geglu != 0);
  if (variant) *variant = pl.variant;
  if (splitk) *splitk = pl.splitk;
}

hipError_t ia2p_launch_splitk_reduce(const GemmArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(std::min(16384, (a.M + 3) / 4)), dim3(256), 0, s, (const float*)a.partial, a.C, a.residual, a.stats_out, a.M, a.N, a.splitk,
                     a.ldc, a.ldr, a);
  return hipGetLastError();
}

template <bool CONV>
static hipError_t launch_any(const GemmArgs& a, int v, hipStream_t s, bool with_reduce) {
  if (a.splitk > 1 && !a.partial) return hipErrorInvalidValue;
  hipError_t e;
  if (v < 0 || v >= IA2P_GEMM_NVARIANT) return hipErrorInvalidValue;
  if (a.geglu && IA2P_GEMM_TILES[v].bn % 32) return hipErrorInvalidValue;   // a (value, gate) block of 32 packed columns must not straddle tiles
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
#undef IA2P_TILE_CASE
    // Measured and dropped in round 1 (tools/gemm_bench.py): 8-wave 256x128 (2- and 3-stage) and 256x320 at one
    // workgroup per CU; BK = 32 rings (64-byte rows halve the request efficiency). The template still takes WGM and BK.
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess || a.splitk <= 1 || !with_reduce || ia2p_splitk_inkernel(a.M, a.N, a.splitk)) return e;
  return ia2p_launch_splitk_reduce(a, s);
}

// launch with an explicit tile variant (a.splitk / a.partial as the caller set them)
// with_reduce = false: a K-split launch leaves its slabs for a separate ia2p_launch_splitk_reduce (the executor times the two apart)
hipError_t ia2p_launch_gemm_variant(const GemmArgs& a, bool conv, int variant, hipStream_t s, bool with_reduce) {
  return conv ? launch_any<true>(a, variant, s, with_reduce) : launch_any<false>(a, variant, s, with_reduce);
}
// *picked (optional) receives the variant id. a.splitk / a.partial must follow ia2p_gemm_plan (the caller owns the slabs).
hipError_t ia2p_launch_gemm(const GemmArgs& a, bool conv, hipStream_t s, int* picked) {
  const GemmPlan pl = ia2p_gemm_plan(a.M, a.N, a.K, conv, a.geglu != 0);
  if (picked) *picked = pl.variant;
  return ia2p_launch_gemm_variant(a, conv, pl.variant, s, true);
}
