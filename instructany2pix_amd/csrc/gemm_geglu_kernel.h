// The GEGLU projection on a 256 x 320 tile (round 6; tile variant 27).
//
// Reference op: diffusers FeedForward / GEGLU of a BasicTransformerBlock behind instructany2pix/ddim/pnp_pipeline.py:253-260 (in-tree twin
// llm/model/vae/modules/attention.py:37-44): h = proj(norm3(x)); a, g = h.chunk(2); out = a * gelu(g) -- `ff.net.0`, the dominant layer role of a denoise step
// (60 launches of 2048 x 10240 x 1280 + 10 of 8192 x 5120 x 640 at batch 8). On the 256 x 160 ping-pong tile the 2048 x 10240 problem is 512 tiles = TWO rounds of one
// workgroup per CU with a prologue, an epilogue and a workgroup hand-over each, and the k-loop runs at 77 % of MFMA issue because a wave's read half-step
// (18 fragment reads + 6.5 LDS-DMA pieces) is as long as the other group's 40 MFMAs. 256 x 320 is ONE round of 256 tiles at 142 flop per staged byte (256 x 160: 98), and its
// read half-step carries 14 fragment reads + 4.5 pieces against the same 40 MFMAs.
//
// Structure: 8 waves = 4 x 2 wave tiles of 64 x 160 (40 accumulator fragments = 160 registers per lane), two wave groups by tile rows (waves 0-3: rows 0-127, waves 4-7:
// rows 128-255) on TWO k-tile slots of 72 KiB (three would need 216 KiB). With 160 accumulator registers a wave cannot hold the fragments of a whole 64-deep k-tile
// (112 registers): the ping-pong runs on 32-deep SUB-steps -- 14 fragment reads, then 40 MFMAs -- so a k-tile is four intervals between workgroup barriers, in each of
// which exactly one group multiplies while the other reads and stages:
//
//   interval      group 0 (waves 0-3)                               group 1 (waves 4-7)
//   I(4t)         read (t, kk 0); issue W pieces of tile t+1        MFMA (t-1, kk 1)
//   I(4t+1)       MFMA (t, kk 0)                                    read (t, kk 0); issue W pieces of tile t+1
//   I(4t+2)       read (t, kk 1); issue OWN A rows of tile t+1      MFMA (t, kk 0)
//   I(4t+3)       MFMA (t, kk 1)                                    read (t, kk 1); issue OWN A rows of tile t+1
//
// Hazards. Tile t+1 goes into the slot of tile t-1. Its weight region was last read by group 1 in I(4t-1) (retired before barrier b(4t)): group 0 re-stages it from I(4t),
// group 1 from I(4t+1). A group stages only the activation rows it reads itself (a wave's A pieces are rows 32 w .. 32 w + 31): rows 0-127 were last read by group 0 in
// I(4t-2) and are re-staged in I(4t+2); rows 128-255 by group 1 in I(4t-1), re-staged in I(4t+3). Read-after-write: group 0 reads tile t+1 from I(4t+4): its own pieces
// are waited for (vmcnt(0)) in front of b(4t+4), and group 1 passes b(4t+4) only with its W pieces landed (counted vmcnt(4): its four younger A pieces may still fly -- only
// group 1 reads them, from I(4t+5), behind its vmcnt(0) in front of b(4t+5)). Every piece has at least two intervals (>= 1 280 cycles of MFMA issue) to land.
//
// Accumulation order is the family's (32-deep MFMA chunks in k order), the epilogue formulas are gemm_epilogue.h's register route: bit-identical to every other tile
// (tests/test_ops_gpu.py). GEGLU launches only (no K split, no convolution): ia2p_launch_gemm_variant refuses anything else.
#pragma once
#include "gemm_tile.h"

struct Geglu320 {
  static constexpr int BM = 256, BN = 320, BK = 64, NWAVE = 8, WGN = 2, WM = 64, WN = 160, MR = 4, NR = 10;
  static constexpr int ROWB = 128, RPP = 8, A_PW = 4, B_PW = 5;               // row bytes of a k-tile, rows per 1-KiB staging piece, pieces per wave and k-tile
  static constexpr int STAGE = (BM + BN) * ROWB, RING = 2 * STAGE;           // 73 728 B per k-tile slot
  // epilogue: the OUTPUT tile (BM x 160 fp16: the GEGLU happens in registers), rows padded by 16 B; column constants; the normal-CDF table of the gate
  static constexpr int HN = BN / 2, P16 = HN * 2 + 16, T16_BYTES = BM * P16;
  // behind the ring, written in the PROLOGUE (their global loads fly beside the first k-tile's DMA; fetched at the epilogue's start they cost every workgroup a cold
  // round trip between its last MFMA and its first conversion): row mean / rstd, the column constants, the table
  static constexpr int CONST_BYTES = 2 * BN * 4, LUT_BYTES = (IA2P_PHI_LUT_N * 8 + 15) & ~15, LNROW_BYTES = 2 * BM * 4;
  static_assert(T16_BYTES <= RING, "the output tile lives inside the ring");
  static constexpr int SMEM = RING + LNROW_BYTES + CONST_BYTES + LUT_BYTES;
  static_assert(SMEM <= 160 * 1024 && A_PW * NWAVE * RPP == BM && B_PW * NWAVE * RPP == BN, "LDS budget / staging split");
};

// The leading dwords of the argument list arrive in SGPRs (kernarg preload, build.py); everything in `p` is a scalar load from the argument block, cold at every launch.
// What a workgroup needs BEFORE its first DMA piece and its statistics loads is therefore spelled out in front of `p`: operands, shape, strides, tile order, and the folded
// LayerNorm's statistics / column-sum pointers (with the pointers in `p`, the statistics loads left a cold scalar round trip after the DMA and came back that much later).
// (FOURTEEN dwords are preloaded -- 16 user SGPRs less the argument block's address, gemm_kernel.h: four pointers, five ints, {slots, tile order} packed in one.)
__global__ __launch_bounds__(512, 2) void gemm_geglu_f16_kernel(const half_t* hA, const half_t* hW, const float* h_ln_stats, const float* h_ln_cs, int hM, int hN, int hK, int hlda, int hldw,
                                                               int h_slots_gw, const GemmArgs p) {
  const int h_ln_slots = IA2P_SKGW_LO8(h_slots_gw), hgroup_w = IA2P_SKGW_GW(h_slots_gw), h_m_fastest = IA2P_SKGW_FLAGS(h_slots_gw) & 1;
  using G = Geglu320;
  constexpr int BM = G::BM, BN = G::BN, MR = G::MR, NR = G::NR, ROWB = G::ROWB, RPP = G::RPP, A_PW = G::A_PW, B_PW = G::B_PW, STAGE = G::STAGE, NT = G::NWAVE * 64;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)hA, 0, 0x7ffffe00, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)hW, 0, 0x7ffffe00, 0x00020000);
  IA2P_STAMP(const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();)

  const int tiles_m = (hM + BM - 1) / BM, tiles_n = (hN + BN - 1) / BN;
  int tm, tn;
  tile_order(blockIdx.x, tiles_m, tiles_n, hgroup_w, h_m_fastest, tm, tn);
  const int bm0 = tm * BM, bn0 = tn * BN;

  // ---- staging: piece pi covers tile rows 8 pi .. 8 pi + 7; lane -> (row 8 pi + lane / 8, LDS chunk lane % 8), which holds global chunk (lane % 8) ^ swz(row),
  //      swz(row) = (row >> 1) & 7 = 4 (pi & 1) + (lane >> 4). Buffer loads to LDS: descriptor + this lane's byte offset + a SCALAR byte offset. A wave's pieces are
  //      consecutive, so piece i sits 8 i rows below piece 0 -- a wave-uniform distance that rides in the scalar offset -- and only the parity of pi changes the
  //      lane's chunk: TWO offset registers per operand (even / odd piece) instead of one per piece. (The k-loop runs at the register limit: with nine offset
  //      registers the allocator spilled them, and a scratch reload in the loop waits on vmcnt, i.e. drains the DMA queue.) Whole tiles only (the launcher checks
  //      M % 256 == 0, N % 320 == 0, no row map): no row is ever out of range.
  const int srow = lane >> 3, cpos = lane & 7;
  int a_voff[2], w_voff[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {           // [par]: this wave's pieces i with i & 1 == par (first piece of the wave: 4 wave -- even -- for A, 5 wave for W)
    a_voff[par] = ((bm0 + wave * A_PW * RPP + srow) * hlda + (cpos ^ ((srow >> 1) + 4 * par)) * 8) * 2;
    w_voff[par] = ((bn0 + wave * B_PW * RPP + srow) * hldw + (cpos ^ ((srow >> 1) + 4 * (par ^ (wave & 1)))) * 8) * 2;
  }
  const int a_pstep = RPP * hlda * 2, w_pstep = RPP * hldw * 2;                  // bytes from a piece to the next
  auto issue_a = [&](int slot, int soff) {
#pragma unroll
    for (int i = 0; i < A_PW; ++i) BLDS16(rs_a, smem + slot * STAGE + (wave * A_PW + i) * 1024, a_voff[i & 1], soff + i * a_pstep);
  };
  auto issue_w = [&](int slot, int soff) {
#pragma unroll
    for (int i = 0; i < B_PW; ++i) BLDS16(rs_w, smem + slot * STAGE + BM * ROWB + (wave * B_PW + i) * 1024, w_voff[i & 1], soff + i * w_pstep);
  };

  // ---- fragments: wave tile origin is a multiple of 16, so swz(row) = (lane >> 1) & 7 for every fragment row
  const int wm0 = (wave >> 1) * G::WM, wn0 = (wave & 1) * G::WN;
  const int frow = lane & 15, fq = lane >> 4;
  const int fswz = lds_swz<64>(frow);
  const int a_off = (wm0 + frow) * ROWB, w_off = BM * ROWB + (wn0 + frow) * ROWB;

  const int nk = hK / 64;
  // folded LayerNorm (consumer): thread r < BM collects the {sum, sum of squares} partials of tile row r. The loads go out AHEAD of the prologue DMA and are folded behind
  // it (vmcnt retires in order: behind the DMA their wait would also be a wait for the whole first k-tile, and then for their own round trip on top; gemm_kernel.h)
  constexpr int MAXS = 24;
  float2 ln_v[MAXS];
  const bool ln_wide = h_ln_stats && tid < BM && h_ln_slots <= MAXS;
  if (ln_wide) {
    const float2* st = (const float2*)h_ln_stats + (bm0 + tid);
#pragma unroll
    for (int u = 0; u < MAXS; ++u) ln_v[u] = st[(size_t)min(u, h_ln_slots - 1) * hM];
  }
  __builtin_amdgcn_sched_barrier(0);
  issue_a(0, 0);            // k-tile 0, whole, into slot 0
  issue_w(0, 0);
  __builtin_amdgcn_sched_barrier(0);
  // column constants and the gate table: loaded here, behind the DMA issue (their pointers come out of the argument block: ahead of the DMA issue they would hold it up for that cold read), parked in the LDS behind the ring before the first barrier
  f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
  const bool ccol = h_ln_stats && tid < BN / 4;             // (whole tiles: every column is in range)
  if (ccol) { c0 = *(const f4*)(h_ln_cs + bn0 + tid * 4); c1 = *(const f4*)(p.ln_bias + bn0 + tid * 4); }
  const float2 t0 = ((const float2*)p.phi_lut)[tid], t1 = ((const float2*)p.phi_lut)[min(tid + NT, IA2P_PHI_LUT_N - 1)];
  static_assert(IA2P_PHI_LUT_N > NT && IA2P_PHI_LUT_N <= 2 * NT, "table copy: two entries per thread");
  // the row's mean / rstd are parked in the 2 KiB of LDS behind the ring (nothing is carried through the k-loop in registers)
  float* ln_rows = (float*)(smem + G::RING);                 // [0, BM): mean, [BM, 2 BM): rstd
  float* ln_cs = ln_rows + 2 * BM;                           // BN column sums, BN folded biases
  float* ln_lb = ln_cs + BN;
  float2* phi = (float2*)(ln_lb + BN);                       // the normal-CDF table of the gate activation (gelu_lut_f)
  if (ccol) { *(f4*)(ln_cs + tid * 4) = c0; *(f4*)(ln_lb + tid * 4) = c1; }
  phi[tid] = t0;
  if (tid + NT < IA2P_PHI_LUT_N) phi[tid + NT] = t1;
  if (h_ln_stats && tid < BM) {
    float ln_s1 = 0.f, ln_s2 = 0.f;
    const float2* st = (const float2*)h_ln_stats + (bm0 + tid);
    if (ln_wide) {
#pragma unroll
      for (int u = 0; u < MAXS; u += 12)      // (opaque to the optimizer: the first addition must not be hoisted to right behind the loads, ahead of the DMA issue)
        asm volatile("" : "+v"(ln_v[u].x), "+v"(ln_v[u].y), "+v"(ln_v[u + 1].x), "+v"(ln_v[u + 1].y), "+v"(ln_v[u + 2].x), "+v"(ln_v[u + 2].y), "+v"(ln_v[u + 3].x), "+v"(ln_v[u + 3].y),
                          "+v"(ln_v[u + 4].x), "+v"(ln_v[u + 4].y), "+v"(ln_v[u + 5].x), "+v"(ln_v[u + 5].y), "+v"(ln_v[u + 6].x), "+v"(ln_v[u + 6].y), "+v"(ln_v[u + 7].x), "+v"(ln_v[u + 7].y),
                          "+v"(ln_v[u + 8].x), "+v"(ln_v[u + 8].y), "+v"(ln_v[u + 9].x), "+v"(ln_v[u + 9].y), "+v"(ln_v[u + 10].x), "+v"(ln_v[u + 10].y), "+v"(ln_v[u + 11].x), "+v"(ln_v[u + 11].y));
#pragma unroll
      for (int u = 0; u < MAXS; ++u)
        if (u < h_ln_slots) { ln_s1 += ln_v[u].x; ln_s2 += ln_v[u].y; }
    } else {
      for (int sl = 0; sl < h_ln_slots; ++sl) { const float2 v = st[(size_t)sl * hM]; ln_s1 += v.x; ln_s2 += v.y; }
    }
    const float2 mr = ln_mean_rstd_f(ln_s1, ln_s2, hK, p.ln_eps);
    ln_rows[tid] = mr.x;
    ln_rows[BM + tid] = mr.y;
  }
  f4 acc[MR][NR];
  // No zero-initialisation: the FIRST sub-step's MFMAs take the literal 0 as their addend and define the accumulators (mm(First)). Left to the compiler, the zeroing went into
  // the first sub-step anyway, an accumulator at a time right in front of its first MFMA and into registers that had just been a fragment of the MFMA before --
  // `v_mfma a[0:3], v[30:33], ..` / `v_mov v[30:33], 0` / `v_mfma v[30:33], v[10:13], ..` / `v_mov v[10:13], 0` / `v_mfma v[10:13], ..` -- and since the MFMAs are inline asm
  // (below) nobody keeps the wait states such a sequence needs: the accumulators concerned, (3, 8) / (3, 9) of the second wave group, came out a last fp32 bit off in rare
  // elements (1 - 6 of 5 M outputs one fp16 ulp off every other tile; tools/geglu_tile_diag.py).
  auto pin_acc = [&]() {
    asm volatile("" : "+v"(acc[0][8]), "+v"(acc[0][9]), "+v"(acc[1][8]), "+v"(acc[1][9]), "+v"(acc[2][8]), "+v"(acc[2][9]), "+v"(acc[3][8]), "+v"(acc[3][9]));
#pragma unroll
    for (int i = 0; i < MR; ++i)
      asm volatile("" : "+a"(acc[i][0]), "+a"(acc[i][1]), "+a"(acc[i][2]), "+a"(acc[i][3]), "+a"(acc[i][4]), "+a"(acc[i][5]), "+a"(acc[i][6]), "+a"(acc[i][7]));
  };

  h8 af[MR], wf[NR];
  auto rd = [&](int slot, auto kk_tag) {
    constexpr int KK = decltype(kk_tag)::value;
    const char* base = smem + slot * STAGE;
    const int coff = ((KK * 4 + fq) ^ fswz) << 4;
#pragma unroll
    for (int i = 0; i < MR; ++i) af[i] = *(const h8*)(base + a_off + i * 16 * ROWB + coff);
#pragma unroll
    for (int j = 0; j < NR; ++j) wf[j] = *(const h8*)(base + w_off + j * 16 * ROWB + coff);
  };
  auto mm = [&](auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value;      // the very first sub-step: addend 0, the accumulators are DEFINED here
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        // accumulators pinned by hand: fragment columns 0-7 (128 registers) to the ACC half of the register file, columns 8-9 (32) to the arch half. At two waves per
        // SIMD the allocator splits a wave's 256 registers 128 / 128 once ACC registers are in use; left to itself (everything in arch VGPRs, or 160 ACC registers) it
        // spilled the staging offsets / the accumulators -- and a scratch reload in the loop waits on vmcnt, i.e. drains the DMA queue.
        // (W is the first operand: a lane ends up with 4 consecutive output columns of one row.)
        if constexpr (FIRST) {
          if (j < 8) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&a"(acc[i][j]) : "v"(wf[j]), "v"(af[i]));      // (early clobber: a destination must not land on a fragment that dies here)
          else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc[i][j]) : "v"(wf[j]), "v"(af[i]));
        } else {
          if (j < 8) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(wf[j]), "v"(af[i]));
          else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(wf[j]), "v"(af[i]));
        }
      }
  };
  using First = std::true_type;
  using Next = std::false_type;
  // The MFMAs are inline asm: the compiler's hazard recognizer does not know that whatever it places behind them (register copies where the two groups' paths join,
  // spills, the epilogue's reads) reads the result of a 16-pass matrix instruction, which needs 19 wait states and is not interlocked. In the loop nothing reads an
  // accumulator but the next MFMA on it, 40 instructions later; at the END of each group's path the wait states are spelled out -- as the last statement of the path, so
  // that the copies the allocator inserts at the join come behind them -- and every accumulator passes through an asm statement behind the wait, so that no read of it
  // can be placed earlier. (The ISA of a build without it showed v_accvgpr_read / v_accvgpr_mov of the accumulators straight behind the last MFMAs of group 1.)
  auto settle = [&]() {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    pin_acc();
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  // barriers as opaque statements ("memory": no LDS access or DMA issue moves across), pinned against the scheduler on both sides
#ifndef IA2P_G320_PHASES
#define IA2P_G320_BAR(text)                          \
  do {                                               \
    __builtin_amdgcn_sched_barrier(0);               \
    asm volatile(text "\n\ts_barrier" ::: "memory"); \
    __builtin_amdgcn_sched_barrier(0);               \
  } while (0)
#define IA2P_G320_PH(k)
#else      // diagnostic build (tools/micro/geglu_clock.hip): per interval of a k-tile, the cycles a wave WORKS (up to its own counted wait) and the cycles it then WAITS in the barrier
  unsigned long long ph_prev = 0, ph_arrive = 0, ph_work[4] = {0, 0, 0, 0}, ph_wait[4] = {0, 0, 0, 0};
#define IA2P_G320_BAR(text)                            \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    asm volatile(text ::: "memory");                   \
    ph_arrive = __builtin_amdgcn_s_memtime();          \
    asm volatile("s_barrier" ::: "memory");            \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)
#define IA2P_G320_PH(k) do { const unsigned long long now = __builtin_amdgcn_s_memtime(); ph_work[k] += ph_arrive - ph_prev; ph_wait[k] += now - ph_arrive; ph_prev = now; } while (0)
#endif
  IA2P_STAMP(const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();)
  IA2P_G320_BAR("s_waitcnt vmcnt(0)");               // b(0): k-tile 0 has landed for every wave
#ifdef IA2P_G320_PHASES
  ph_prev = __builtin_amdgcn_s_memtime();
#endif
  if (wave < 4) {
    auto step = [&](int t, int slot, auto first_tag) {
      const bool more = t + 1 < nk;
      rd(slot, K0{});
      __builtin_amdgcn_sched_barrier(0);
      if (more) issue_w(slot ^ 1, (t + 1) * ROWB);
      IA2P_G320_BAR("s_waitcnt lgkmcnt(0)");         // b(4t+1)
      IA2P_G320_PH(0);
      mm(first_tag);
      IA2P_G320_BAR("");                             // b(4t+2)
      IA2P_G320_PH(1);
      rd(slot, K1{});
      __builtin_amdgcn_sched_barrier(0);
      if (more) issue_a(slot ^ 1, (t + 1) * ROWB);
      IA2P_G320_BAR("s_waitcnt lgkmcnt(0)");         // b(4t+3)
      IA2P_G320_PH(2);
      mm(Next{});
      if (more) IA2P_G320_BAR("s_waitcnt vmcnt(0)"); // b(4t+4): this wave's pieces of tile t+1 have landed
      else IA2P_G320_BAR("s_nop 7\n\ts_nop 7\n\ts_nop 7");      // ... behind the last sub-step: the wait states of `settle`, in front of whatever the allocator places on the loop's exit edge
      IA2P_G320_PH(3);
    };
    step(0, 0, First{});
    for (int t = 1, slot = 1; t < nk; ++t, slot ^= 1) step(t, slot, Next{});
    settle();
  } else {
    auto step = [&](int t, int slot, auto first_tag) {
      const bool more = t + 1 < nk;
      IA2P_G320_BAR("s_waitcnt vmcnt(0)");           // b(4t+1): this wave's A rows of tile t (issued in I(4t-1)) have landed
      IA2P_G320_PH(0);                               // (interval I(4t): group 1 multiplied (t-1, kk 1), or idled in the first k-tile)
      rd(slot, K0{});
      __builtin_amdgcn_sched_barrier(0);
      if (more) issue_w(slot ^ 1, (t + 1) * ROWB);
      IA2P_G320_BAR("s_waitcnt lgkmcnt(0)");         // b(4t+2)
      IA2P_G320_PH(1);
      mm(first_tag);
      IA2P_G320_BAR("");                             // b(4t+3)
      IA2P_G320_PH(2);
      rd(slot, K1{});
      __builtin_amdgcn_sched_barrier(0);
      if (more) issue_a(slot ^ 1, (t + 1) * ROWB);
      // b(4t+4): the reads of (t, kk 1) are retired (group 0 re-stages this slot's weights next), the W pieces of tile t+1 have landed (the 4 younger A pieces may fly)
      if (more) IA2P_G320_BAR("s_waitcnt vmcnt(4) lgkmcnt(0)");
      else IA2P_G320_BAR("s_waitcnt vmcnt(0) lgkmcnt(0)");
      IA2P_G320_PH(3);
    };
    step(0, 0, First{});
    for (int t = 1, slot = 1; t < nk; ++t, slot ^= 1) {
      mm(Next{});                                    // (t-1, kk 1)
      step(t, slot, Next{});
    }
    mm(Next{});                                      // (nk-1, kk 1)
    settle();
  }
#undef IA2P_G320_BAR
#ifdef IA2P_G320_PHASES
  if ((tid == 0 || tid == 256) && p.partial) {      // wave 0 (group 0) / wave 4 (group 1): behind the workgroups' 8-slot records
    unsigned long long* o = (unsigned long long*)p.partial + 8 * gridDim.x + 16 * blockIdx.x + (tid ? 8 : 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) { o[k] = ph_work[k]; o[4 + k] = ph_wait[k]; }
  }
#endif
#undef IA2P_G320_PH
  IA2P_STAMP(
    if (tid == 0 && p.partial) {      // (the stamps go to a buffer nothing else reads: GEGLU launches have no slabs)
      unsigned long long* o = (unsigned long long*)p.partial + 8 * blockIdx.x;
      const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
      o[0] = __builtin_amdgcn_s_memtime() - stamp_c0; o[1] = r1 - stamp_r0; o[2] = stamp_entry; o[3] = stamp_r0; o[4] = r1;
    }
  )

  // ---- epilogue. The formulas are gemm_epilogue.h's register route -- bias / folded LayerNorm on the accumulators, ONE rounding to fp16 (the projection's output is an
  //      fp16 tensor), out = fp16(value x GELU(gate)) through the normal-CDF table -- but the GEGLU happens IN REGISTERS: in the packed column order a 32-wide block is
  //      [16 values | 16 gates], i.e. fragment column 2 b holds the values and 2 b + 1 the gates of block b, and the MFMA layout gives a lane the same 4 columns of both:
  //      value and gate of an output sit in the same lane, acc[i][2 b][e] and acc[i][2 b + 1][e]. All eight waves work at once, only the OUTPUT tile (256 x 160 fp16: half
  //      the bytes) crosses the LDS, once, to leave as whole 128-byte lines. (The first version sent the projected tile through the LDS in two 160-column halves, four waves
  //      converting while four waited: 11.0 us from the k-loop's end to the last store drained, tools/micro/geglu_clock.hip.)
  char* t16 = smem;                                          // output tile [BM][160] fp16, rows padded by 16 B
  __syncthreads();                    // every wave has finished reading the stage buffers
  // this workgroup's slice of the NEXT contraction's weights (a workgroup owns its CU's LDS: no separate prefetch workgroups), issued here and
  // consumed at the kernel's end
  unsigned pfacc = 0, pfv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (p.pf) {
    const long nwg = (long)tiles_m * tiles_n;
    const long per = ((p.pf_bytes + nwg - 1) / nwg + 255) & ~255L;
    const long lo = (long)blockIdx.x * per, hi = min(lo + per, p.pf_bytes & ~15L);
    const char* src = (const char*)p.pf;
    constexpr long SW = NT * 16;
    if (lo < hi)
      for (long o = lo + tid * 16; o < hi; o += 8 * SW) {
#pragma unroll
        for (int u = 0; u < 8; ++u) pfacc ^= pfv[u];
#pragma unroll
        for (int u = 0; u < 8; ++u) pfv[u] = *(const unsigned*)(src + min(o + u * SW, hi - 16));
      }
  }
  const float e_as = p.acc_scale == 0.f ? 1.f : p.acc_scale, e_bs = p.bias_scale == 0.f ? 1.f : p.bias_scale;
  const __amdgpu_buffer_rsrc_t c_rsrc = wt_rsrc((void*)p.C, (size_t)hM * p.ldc * 2);
  // (row / column constants and the table have been in LDS since the prologue)
  auto geglu_regs = [&](auto ln_tag) {
    constexpr bool LN = decltype(ln_tag)::value;
    float mu[MR], rs[MR];
#pragma unroll
    for (int i = 0; i < MR; ++i) { mu[i] = LN ? ln_rows[wm0 + i * 16 + frow] : 0.f; rs[i] = LN ? ln_rows[BM + wm0 + i * 16 + frow] : 1.f; }
#pragma unroll
    for (int b = 0; b < NR / 2; ++b) {                                // 32-wide packed block b of this wave's 160 columns: fragment column 2 b = values, 2 b + 1 = gates
      f4 ca0 = {0.f, 0.f, 0.f, 0.f}, ca1 = {0.f, 0.f, 0.f, 0.f}, cg0 = {0.f, 0.f, 0.f, 0.f}, cg1 = {0.f, 0.f, 0.f, 0.f};      // folded LayerNorm: column sums, folded biases; else: bias, -
      const int cl = wn0 + b * 32 + fq * 4;                           // tile column of this lane's 4 values (their gates: + 16)
      if constexpr (LN) { ca0 = *(const f4*)(ln_cs + cl); ca1 = *(const f4*)(ln_lb + cl); cg0 = *(const f4*)(ln_cs + cl + 16); cg1 = *(const f4*)(ln_lb + cl + 16); }
      else {
        const h4 ha = *(const h4*)(p.bias + min(bn0 + cl, hN - 4)), hg = *(const h4*)(p.bias + min(bn0 + cl + 16, hN - 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) { ca0[e] = (float)ha[e]; cg0[e] = (float)hg[e]; }
      }
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float va = acc[i][2 * b][e], vg = acc[i][2 * b + 1][e];
          if constexpr (LN) { va = ln_fold_f(va, mu[i], rs[i], ca0[e], ca1[e]); vg = ln_fold_f(vg, mu[i], rs[i], cg0[e], cg1[e]); }
          else { va = fmaf(ca0[e], e_bs, va * e_as); vg = fmaf(cg0[e], e_bs, vg * e_as); }
          va = (float)(half_t)va; vg = (float)(half_t)vg;             // (the projection's output is an fp16 tensor: the rounding every other tile makes on its way through the LDS)
          o[e] = (half_t)(va * gelu_lut_f(vg, phi));
        }
        *(h4*)(t16 + (wm0 + i * 16 + frow) * G::P16 + ((wn0 >> 1) + b * 16 + fq * 4) * 2) = o;
      }
      __builtin_amdgcn_sched_barrier(0);      // one block at a time (constant loads of all five blocks hoisted to the top would cost 80 registers)
    }
  };
  if (h_ln_stats) geglu_regs(std::true_type{});
  else geglu_regs(std::false_type{});
  __syncthreads();
  IA2P_STAMP(if (tid == 0 && p.partial) ((unsigned long long*)p.partial)[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();)      // the output tile is in LDS
  {
    // read out row-major, 16 B = 8 outputs per thread: 20 groups per row, 5 120 in all = 10 per thread; whole-line write-through stores
    constexpr int GPR = G::HN / 8, TOTAL = BM * GPR, U = 5, ITER = TOTAL / (NT * U);
    static_assert(TOTAL % (NT * U) == 0, "read-out: whole rounds");
#pragma unroll 1
    for (int k = 0; k < ITER; ++k) {
      h8 hv[U];
      int rr[U], gg[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = tid + (k * U + u) * NT;
        const int r = idx / GPR, g = idx - r * GPR;
        rr[u] = r; gg[u] = g;
        hv[u] = *(const h8*)(t16 + r * G::P16 + g * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = bm0 + rr[u] < hM && bn0 + gg[u] * 16 < hN;
        if (live) {
          const size_t elem = (size_t)(bm0 + rr[u]) * p.ldc + (bn0 >> 1) + gg[u] * 8;
          if (p.c_wt) store16_wt(c_rsrc, elem * 2, hv[u]);
          else *(h8*)(p.C + elem) = hv[u];
        }
      }
    }
  }
  asm volatile("" ::"v"(pfacc), "v"(pfv[0]), "v"(pfv[1]), "v"(pfv[2]), "v"(pfv[3]), "v"(pfv[4]), "v"(pfv[5]), "v"(pfv[6]), "v"(pfv[7]));      // keep the prefetch loads alive up to here
  IA2P_STAMP(
    if (tid == 0 && p.partial) ((unsigned long long*)p.partial)[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();      // this wave has ISSUED its last C store
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && p.partial) ((unsigned long long*)p.partial)[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();
  )
}

// GEGLU launches of a linear layer only, whole tiles (M % 256 == 0, N % 320 == 0: every GEGLU projection of the BASELINE shapes), no row map, 16-byte epilogue accesses, no K split
static inline bool ia2p_geglu320_ok(const GemmArgs& a) {
  return a.geglu && a.splitk <= 1 && !a.act && !a.residual && !a.rowvec && !a.stats_out && !a.gn_out && !a.gn.st0 && a.K >= 64 && a.K % 64 == 0 && a.M > 0 && a.M % Geglu320::BM == 0 &&
         a.N > 0 && a.N % Geglu320::BN == 0 && !a.rpb && (a.bias || a.ln_stats);
}
static hipError_t launch_geglu320(const GemmArgs& a, hipStream_t s) {
  using G = Geglu320;
  if (!ia2p_geglu320_ok(a)) return hipErrorInvalidValue;
  static bool attr_set[64] = {false};      // per device (the attribute is)
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_geglu_f16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  GemmArgs b = a;
  ia2p_gemm_prepare(b, G::SMEM, G::BM, G::BN, false);
  if (!b.vec8) return hipErrorInvalidValue;
  if (!b.phi_lut) return hipErrorOutOfMemory;
  const size_t a_rows = a.rpb ? ((size_t)a.M / a.rpb + 1) * (size_t)std::max(a.bstride, 0) + a.roff + a.rpb : (size_t)a.M;
  if (!ia2p_fits_buffer(a_rows, a.lda) || !ia2p_fits_buffer(a.N, a.ldw)) return hipErrorInvalidValue;
  if ((a.ln_stats && ((((uintptr_t)a.ln_cs) | ((uintptr_t)a.ln_bias)) & 15)) || (!a.ln_stats && (((uintptr_t)a.bias) & 7))) return hipErrorInvalidValue;
  b.splitk = 0; b.sk_counters = nullptr;
#ifndef IA2P_CLOCK_STAMP
  b.partial = nullptr;
#endif
  const int tiles = ((a.M + G::BM - 1) / G::BM) * ((a.N + G::BN - 1) / G::BN);
  if (tiles >= (1 << 21)) return hipErrorInvalidValue;      // (udiv_small in the tile decode)
  int slots_gw;
  if (!ia2p_pack_skgw(b.ln_stats ? b.ln_slots : 0, b.group_w, b.m_fastest, b.ln_stats != nullptr, &slots_gw)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gemm_geglu_f16_kernel, dim3(tiles), dim3(512), G::SMEM, s, b.A, b.W, b.ln_stats, b.ln_cs, b.M, b.N, b.K, b.lda, b.ldw, slots_gw, b);
  return hipGetLastError();
}
