// The GEMM / implicit-GEMM kernel template and its launcher (see gemm.hip for the structure). A header so that the fused
// to_q + cross-attention tile (qxattn.hip) can instantiate it with the attention core as its epilogue, in a translation unit of its own
// (the attention code wants -amdgpu-mfma-vgpr-form, the plain GEMM tiles do not).
// Shared pieces: gemm_tile.h (staging macros, counted waits, LDS budget, tile order), gemm_epilogue.h (the epilogue every tile runs); the halo-staged
// 3x3 convolution has a k-loop of its own in conv_halo_kernel.h.
#pragma once
#include "gemm_epilogue.h"

// PP = 1 ("ping-pong", 8 waves = WGM 4, 3-stage ring, ONE workgroup per CU): waves 0-3 own the upper half of the tile rows, waves 4-7 the
// lower half, and the two groups run half a k-step apart -- while one group reads its fragments from LDS the other issues its MFMAs, with a
// workgroup barrier between the half-steps. Eight waves behind one barrier per k-step would all read, then all multiply (the LDS and the
// MFMA phases add up); two independent workgroups per CU de-phase by themselves but need twice the LDS fill per flop
// (profiles/r01g_gemm_loop_ablation.txt: the fill is the largest term of the 128x128 kernel).
// XA != 0 (qxattn.hip; 128 x 64 tiles only): the tile is the to_q projection of 128 queries x ONE head and never leaves the CU -- the epilogue turns it
// into the Q fragments of the attention core (attention_core.h, MODE = XA - 1) and writes the cross-attention output instead.
template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int BK = 64, int PP = 0, int WGN = 2, int XA = 0>
__device__ __forceinline__ void gemm_tile_body(const half_t* hA, const half_t* hW, const half_t* hzero, int hM, int hN, int hK, int hlda, int hldw, int hrpb, int hbstride,
                                               int hroff, int hsplitk, int hgroup_w, int hflags, const GemmArgs& p, const AttnArgs* xa, const float* pre_ln_stats = nullptr, int pre_ln_slots = 0) {
  // hflags (preloaded with the other leading arguments): bit 0 = p.m_fastest, bit 1 = p.ln_stats != nullptr. Read from `p` they are scalar loads from the cold argument block
  // whose results stand between the workgroup's entry and its first DMA piece: the tile decode does not even use m_fastest under a grouped order, but the load was in flight
  // into registers the decode reuses (a wait for it, in every launch), and every launch asked p.ln_stats whether it had statistics to fetch ahead of its DMA.
  // The leading arguments of the kernels are what the prologue needs; built with -amdgpu-kernarg-preload-count=16 the command processor
  // hands the first 14 dwords over in SGPRs (the kernels pack theirs to fit: gemm_f16_kernel below), so the first tile loads go out without waiting for a cold read of the argument block (which costs every launch
  // ~1 us: tools/micro/launch_floor2.hip). The rest of GemmArgs (epilogue, conv geometry) arrives while those loads fly.   // >= 2 waves/SIMD: big tiles must fit 256 registers
  static_assert(PP != 1 || (WGM * WGN == 8 && (WGM == 4 || WGM == 8) && NSTAGE == 3), "ping-pong schedule: 8 waves (two groups of 4 by tile rows), 3-stage ring");
  static_assert(PP != 3 || (WGM == 4 && NSTAGE == 2 && !CONV), "two-slot ping-pong schedule: 8 waves, 2 k-tile slots");
  static_assert(PP != 2 || (WGM == 2 && WGN == 4 && NSTAGE == 2 && BK == 64 && BM == 256 && (BN == 256 || BN == 128) && XA == 0), "8-phase schedule: 256-row tiles, 2 x 4 waves, two k-tile buffers");
  constexpr int NWAVE = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;    // wave tile (waves arranged WGM x WGN; WGN = 1: narrow tiles, one wave per 128-byte column block)
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int ROWB = 2 * BK, CPR = ROWB / 16, RPP = 1024 / ROWB;   // row bytes, chunks per row, rows per 1-KiB staging piece
  using BS = BStage<BN, BK, NWAVE, PP>;
  constexpr int A_PW = BM / RPP / NWAVE, B_PW = BS::HI;             // staging pieces per wave (B rounded up: the surplus rows read the zero page -- or,
  constexpr bool B_UNEVEN = BS::UNEVEN;                              //  ping-pong tile, the second wave group takes one piece less per wave: BStage)
  constexpr int BNL = BS::BNL;                                       // weight rows held in LDS (>= BN)
  static_assert(BM % (RPP * NWAVE) == 0 && BN % 16 == 0 && WN % 16 == 0 && WM % 16 == 0, "tile / wave layout");
  constexpr int STAGE = (BM + BNL) * ROWB;                             // bytes of a ring slot
  constexpr int KSUB = BK / 32;                                       // 32-deep MFMA sub-steps per k-tile
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // linear layers: both operands through buffer loads -- a piece is (descriptor, this lane's byte offset of its row and chunk at k = 0, scalar byte offset of the
  // k-tile); rows past M / N carry an offset past the descriptor's range and read zeros. Operands are addressed with 31 bits: the launcher refuses larger ones.
  constexpr bool LINBUF = !CONV && IA2P_LIN_BUF != 0;
  constexpr int OOB = 0x7fffff00;
  const __amdgpu_buffer_rsrc_t lin_rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)hA, 0, 0x7ffffe00, 0x00020000);
  const __amdgpu_buffer_rsrc_t lin_rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)hW, 0, 0x7ffffe00, 0x00020000);
  IA2P_STAMP(const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();)

  // ---- tile of this workgroup; blocks b, b+8, ... share an XCD (its L2): give each XCD a contiguous tile range
  const int tiles_m = (hM + BM - 1) / BM, tiles_n = (hN + BN - 1) / BN;
  int bid = blockIdx.x;
  const int nsplit = hsplitk > 1 ? hsplitk : 1;
  const int split = bid < tiles_m * tiles_n ? 0 : udiv_small(bid, tiles_m * tiles_n);          // >= nsplit: prefetch workgroup (the common case, a tile of an unsplit launch: one comparison)
  if (split < nsplit) bid -= split * tiles_m * tiles_n;
  if (__builtin_expect(split >= nsplit, 0)) {   // prefetch workgroup: touch its slice of the next kernel's weights and leave (out of line: a tile workgroup's start should FALL THROUGH -- every
                                                // taken branch over a cold block is an instruction fetch the sequential prefetcher has not made, a few hundred ns at a launch's start)
    bid -= nsplit * tiles_m * tiles_n;
    const long per = ((p.pf_bytes + p.pf_blocks - 1) / p.pf_blocks + 4095) & ~4095L;
    const long lo = (long)bid * per, hi = min(lo + per, p.pf_bytes & ~15L);
    const char* src = (const char*)p.pf;
    unsigned acc = 0;
    constexpr long SW = NWAVE * 64 * 16;   // bytes swept by the workgroup per pass
#ifndef IA2P_PF_UNROLL
#define IA2P_PF_UNROLL 4      // loads in flight per thread (build-time knob for A/B builds)
#endif
    constexpr int PU = IA2P_PF_UNROLL;
    for (long o = lo + tid * 16; o < hi; o += PU * SW) {
      unsigned v[PU];
#pragma unroll
      for (int u = 0; u < PU; ++u) v[u] = *(const unsigned*)(src + min(o + u * SW, hi - 16));   // one dword per 16-B slot pulls the whole line
#pragma unroll
      for (int u = 0; u < PU; ++u) acc ^= v[u];
    }
    asm volatile("" ::"v"(acc));    // keep the loads alive
    return;
  }
  int tm, tn;
  tile_order(bid, tiles_m, tiles_n, hgroup_w, hflags & 1, tm, tn);
  const int bm0 = tm * BM, bn0 = tn * BN;

  // ---- staging addresses. Piece `pi` covers tile rows pi*8 .. pi*8+7; lane -> (row pi*8 + lane/8, LDS chunk lane%8),
  //      which must hold global chunk (lane%8) ^ swz(row), swz(row) = (row>>1)&7.
  const int srow = lane / CPR, cpos = lane % CPR;
  // staging piece i of this wave -> piece index inside the operand tile. 8-phase tile: an operand tile is staged as two half-tiles (rows [0, B/2) and
  // [B/2, B)) in different phases of the k-loop, and EVERY wave carries an equal share of each half (one counted vmcnt per wave fits all)
  constexpr int A_HP = PP == 2 ? A_PW / 2 : 1, B_HP = PP == 2 ? B_PW / 2 : 1;      // pieces per wave and half-tile
  auto a_piece = [&](int i) { return PP == 2 ? (i / A_HP) * (BM / 2 / RPP) + wave * A_HP + i % A_HP : wave * A_PW + i; };
  const half_t* a_ptr[A_PW];
  int a_inc[A_PW];
  int a_voff[A_PW];
  // conv gather, per piece (= one tile row per lane): a_base = address of filter tap (0, 0)'s pixel for this lane's 16-byte chunk (may lie outside the image: only
  // dereferenced under the mask); a_mask = bits 0-8: tap (ky, kx) falls inside the (virtual) image, bits 9 / 10: parity of the tap-0 row / column in the
  // nearest-x2 upsampled view (source step of tap k = (k + parity) >> up). A new tap's pointer is a bit test, a wave-uniform offset and a select.
  const half_t* a_base[A_PW];
  int a_mask[A_PW];
#pragma unroll
  for (int i = 0; i < A_PW; ++i) {
    const int pi = a_piece(i);
    const int m = bm0 + pi * RPP + srow;
    const int gch = cpos ^ lds_swz<BK>(pi * RPP + srow);
    if (!CONV) {
      if (m < hM) {
        int src = m;
        if (__builtin_expect(hrpb != 0, 0)) { const int b = m / hrpb; src = b * hbstride + (m - b * hrpb) + hroff; }
        a_ptr[i] = hA + (size_t)src * hlda + gch * 8;
        a_inc[i] = BK;
        a_voff[i] = (src * hlda + gch * 8) * 2;
      } else { a_ptr[i] = hzero; a_inc[i] = 0; a_voff[i] = OOB; }
    } else {
      a_mask[i] = 0; a_base[i] = hzero;
      if (m < hM) {
        const int hw = p.Ho * p.Wo;
        const int b = m / hw, rem = m - b * hw;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int y0 = oy * p.stride - p.pad, x0 = ox * p.stride - p.pad;      // tap (0, 0) in the (virtual, upsampled) image
        const int Hv_ = p.Hs << p.up, Wv_ = p.Ws << p.up;
        int mk = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) mk |= ((unsigned)(y0 + t / 3) < (unsigned)Hv_ && (unsigned)(x0 + t % 3) < (unsigned)Wv_) ? 1 << t : 0;
        a_mask[i] = mk | ((y0 & 1) << 9) | ((x0 & 1) << 10);
        a_base[i] = hA + ((long)b * p.Hs * p.Ws + (long)(y0 >> p.up) * p.Ws + (x0 >> p.up)) * hlda + gch * 8;      // (arithmetic shifts: row / column -1 stays -1)
      }
    }
  }
  const half_t* w_ptr[B_PW];
  int w_inc[B_PW];
  int w_voff[B_PW];
  // this wave's weight pieces: [b_pi0, b_pi0 + b_npw)
  const int b_npw = B_UNEVEN && wave >= NWAVE / 2 ? B_PW - 1 : B_PW;
  const int b_pi0 = B_UNEVEN ? (wave < NWAVE / 2 ? wave * B_PW : (NWAVE / 2) * B_PW + (wave - NWAVE / 2) * (B_PW - 1)) : wave * B_PW;
  auto b_piece = [&](int i) { return PP == 2 ? (i / B_HP) * (BN / 2 / RPP) + wave * B_HP + i % B_HP : b_pi0 + i; };
#pragma unroll
  for (int i = 0; i < B_PW; ++i) {
    const int pi = min(b_piece(i), BNL / RPP - 1);
    int n = bn0 + pi * RPP + srow;
    if constexpr (XA == 4) {        // fused QKV + self-attention: the tile's 192 columns are rows h*64 .. h*64+63 of the Q, the K and the V block of the stacked [3C, C] weight
      const int jl = pi * RPP + srow;
      n = (jl >> 6) * (hN / 3) + tn * 64 + (jl & 63);
    }
    const int gch = cpos ^ lds_swz<BK>(pi * RPP + srow);
    if (n < hN && pi * RPP + srow < BN) { w_ptr[i] = hW + (size_t)n * hldw + gch * 8; w_inc[i] = BK; }
    else         { w_ptr[i] = hzero; w_inc[i] = 0; }
    if constexpr (LINBUF) w_voff[i] = w_inc[i] ? (n * hldw + gch * 8) * 2 : OOB;      // (buffer loads: byte offset of the piece's chunk in the weight matrix, column 0)
  }

  const int nk_all = hK / BK;
  // this workgroup's k-tiles (no K split: all of them -- the 64-bit divisions this was written with ran in every launch: ~200 instructions of a workgroup's start)
  const int kt0 = nsplit == 1 ? 0 : udiv_small(split * nk_all, nsplit), kt1 = nsplit == 1 ? nk_all : udiv_small((split + 1) * nk_all, nsplit);
  // conv: K = (tap, channel) is walked TAP-major -- all Cin channels of a filter tap (64 per k-tile, running pointers: the gather of a row is derived once per
  // tap), then the next tap; after the 9 Cin columns of the 3x3 part the appended 1x1 blocks. Weights are packed in that order ([Co][tap][Cin], misc.hip).
  // (The gathered operand walked channel-block-major -- the nine taps of a block of 64 channels, then the next block -- halves the fabric traffic of the large convolutions
  // and was 10 ... 20 % slower, docs/LOG.md r04u: the tap then changes every k-tile and its per-row pointer select sits in the read half-step. The block-major walk
  // ships in the halo-staged kernel, HALO = 1, where a tap is a constant offset.)
  int cin_main = 0, cin_extra = 0;                         // (two named scalars: a select between two argument FIELDS became a 2-entry table in scratch)
  if (CONV) { cin_main = p.Cin; cin_extra = p.Cin2; asm volatile("" : "+s"(cin_main), "+s"(cin_extra)); }
  const int nk_main = CONV ? 9 * (cin_main / BK) : 0;      // k-tiles of the 3x3 part
  int tap = 0, ci0 = 0;
  if (CONV) {
    if (kt0 < nk_main) { tap = (kt0 * BK) / cin_main; ci0 = (kt0 * BK) % cin_main; }
    else {                                                 // (inside the appended 1x1 blocks)
      ci0 = kt0 * BK - 9 * cin_main; tap = 9;
      if (ci0 >= cin_extra) { ci0 -= cin_extra; tap = 10; }
    }
  }
  bool tap_fresh = true;
  // the walk, one k-tile on: (tap, ci0) -> next
  auto k_next = [&](int& t, int& c) {
    if (t < 9) {
      c += BK;
      if (c >= cin_main) { c = 0; ++t; }                                    // next tap (tap = 9: appended block / end of K)
    } else {
      c += BK;
      if (t < 10 && c >= cin_extra) { c = 0; ++t; }                           // (the last block runs to the end of K)
    }
  };
  int k_soff = kt0 * (2 * BK);      // buffer-load form: byte offset of the k-tile being staged
  if (kt0 && !LINBUF) {             // split-K: this workgroup starts at k-tile kt0
    if (!CONV) {
#pragma unroll
      for (int i = 0; i < A_PW; ++i) a_ptr[i] += (size_t)kt0 * a_inc[i];
    }
#pragma unroll
    for (int i = 0; i < B_PW; ++i) w_ptr[i] += (size_t)kt0 * w_inc[i];
  }

  // new filter tap (wave-uniform): re-derive the gathered pixel of each row (once per Cin / 64 k-tiles)
  auto conv_tap_setup = [&]() {
    if (tap_fresh) {
      if (tap < 9) {
        const int ky = tap / 3, kx = tap - ky * 3;
        if (!p.up) {
          const long toff = ((long)ky * p.Ws + kx) * hlda + ci0;      // wave-uniform: elements from tap (0, 0)'s pixel to this tap's, plus the channel block
#pragma unroll
          for (int i = 0; i < A_PW; ++i) {
            const bool ok = (a_mask[i] >> tap) & 1;
            a_ptr[i] = ok ? a_base[i] + toff : hzero;
            a_inc[i] = ok ? BK : 0;
          }
        } else {                  // nearest x2 upsample folded into the gather: the source step of a tap depends on the parity of the row / column
#pragma unroll
          for (int i = 0; i < A_PW; ++i) {
            const bool ok = (a_mask[i] >> tap) & 1;
            const int dy = (ky + ((a_mask[i] >> 9) & 1)) >> 1, dx = (kx + ((a_mask[i] >> 10) & 1)) >> 1;
            a_ptr[i] = ok ? a_base[i] + ((long)dy * p.Ws + dx) * hlda + ci0 : hzero;
            a_inc[i] = ok ? BK : 0;
          }
        }
      } else {              // appended 1x1 block: the output pixel itself, from the second tensor (stride 1: pixel index = output row)
        // (two copies of the loop, not a select between p.A2 and p.A3: a select between FIELDS of the by-value argument struct is compiled
        //  as an indexed access and pushes the whole struct to scratch)
        if (tap == 9) {
#pragma unroll
          for (int i = 0; i < A_PW; ++i) {
            const int m = bm0 + a_piece(i) * RPP + srow;      // (re-derived: happens once or twice per launch)
            const bool ok = m < hM;
            a_ptr[i] = ok ? p.A2 + (size_t)m * p.lda2 + ci0 + (cpos ^ lds_swz<BK>(a_piece(i) * RPP + srow)) * 8 : hzero;
            a_inc[i] = ok ? BK : 0;
          }
        } else {
#pragma unroll
          for (int i = 0; i < A_PW; ++i) {
            const int m = bm0 + a_piece(i) * RPP + srow;
            const bool ok = m < hM;
            a_ptr[i] = ok ? p.A3 + (size_t)m * p.lda3 + ci0 + (cpos ^ lds_swz<BK>(a_piece(i) * RPP + srow)) * 8 : hzero;
            a_inc[i] = ok ? BK : 0;
          }
        }
      }
      tap_fresh = false;
    }
  };
  auto conv_tap_advance = [&]() {
    const int t_old = tap;
    k_next(tap, ci0);
    if (tap != t_old) tap_fresh = true;                   // (inside a tap / an appended block the running pointers just move on)
  };
  // LDS-DMA of this wave's activation pieces [i0, i1) / weight pieces [i0, i1) of the next k-tile into ring slot `buf`; running pointers: no per-step multiply
  auto issue_a = [&](int buf, auto i0_tag, auto i1_tag) {
#pragma unroll
    for (int i = decltype(i0_tag)::value; i < decltype(i1_tag)::value; ++i) {
      if constexpr (LINBUF) BLDS16(lin_rs_a, smem + buf * STAGE + a_piece(i) * 1024, a_voff[i], k_soff);
      else { GLDS16(a_ptr[i], smem + buf * STAGE + a_piece(i) * 1024); a_ptr[i] += a_inc[i]; }
    }
  };
  auto issue_b = [&](int buf, auto i0_tag, auto i1_tag) {
#pragma unroll
    for (int i = decltype(i0_tag)::value; i < decltype(i1_tag)::value; ++i)
      if (!B_UNEVEN || i < b_npw) {      // (wave-uniform)
        if constexpr (LINBUF) BLDS16(lin_rs_w, smem + buf * STAGE + BM * ROWB + b_piece(i) * 1024, w_voff[i], k_soff);
        else { GLDS16(w_ptr[i], smem + buf * STAGE + BM * ROWB + b_piece(i) * 1024); w_ptr[i] += w_inc[i]; }
      }
  };
  // The pointer arithmetic of the NEXT k-tile's gather (a few VALU instructions per piece, every k-tile) belongs beside the MFMAs of the current one, where its
  // issue slots are free -- not in front of the DMA issue, between the barrier and the fragment reads: the loops below call this right ahead of their MFMA
  // blocks (a no-op when the pointers are current; `stage` still derives them itself when nobody did).
  auto conv_prepare = [&]() { if (CONV) conv_tap_setup(); };
  using I0 = std::integral_constant<int, 0>;
  auto stage = [&](int kt, int buf) {
    if (CONV) conv_tap_setup();
    issue_a(buf, I0{}, std::integral_constant<int, A_PW>{});
    if (CONV) conv_tap_advance();
    issue_b(buf, I0{}, std::integral_constant<int, B_PW>{});
    if constexpr (LINBUF) k_soff += 2 * BK;
  };
  // 8-phase tile: one HALF of an operand tile per call, in the order B0, A0, B1, A1 of a k-tile (A0 opens the k-tile for the conv gather, A1 closes it)
  auto stage_part = [&](int buf, auto which_tag) {
    constexpr int WHICH = decltype(which_tag)::value;
    using AH = std::integral_constant<int, A_HP>;
    using BH = std::integral_constant<int, B_HP>;
    if constexpr (WHICH == 0) issue_b(buf, I0{}, BH{});
    else if constexpr (WHICH == 1) { if (CONV) conv_tap_setup(); issue_a(buf, I0{}, AH{}); }
    else if constexpr (WHICH == 2) issue_b(buf, BH{}, std::integral_constant<int, B_PW>{});
    else { issue_a(buf, AH{}, std::integral_constant<int, A_PW>{}); if (CONV) conv_tap_advance(); if constexpr (LINBUF) k_soff += 2 * BK; }
  };

  // ---- fragment read offsets (wave tile origin is a multiple of 16, so swz(row) = (lane>>1)&7)
  // wave tile: MR x NR fragments of 16 x 16. Plain / ping-pong tiles: one contiguous WM x WN block. 8-phase tile: a 2 x 2 arrangement of quadrants, one
  // in each half-tile of A and of B (rows wm0 + [0, WM/2) and BM/2 + wm0 + [0, WM/2), columns likewise), so that a whole half-tile is free for the
  // next k-tile's DMA as soon as every wave has read ITS quadrant rows out of it
  constexpr int MRH = MR / 2, NRH = NR / 2;
  auto frag_row = [](int i) constexpr { return PP == 2 ? (i / (MR / 2)) * (BM / 2) + (i % (MR / 2)) * 16 : i * 16; };
  auto frag_col = [](int j) constexpr { return PP == 2 ? (j / (NR / 2)) * (BN / 2) + (j % (NR / 2)) * 16 : j * 16; };
  const int wm0 = (wave / WGN) * (PP == 2 ? WM / 2 : WM), wn0 = (wave % WGN) * (PP == 2 ? WN / 2 : WN);
  const int frow = lane & 15, fq = lane >> 4;
  const int fswz = lds_swz<BK>(frow);
  const int a_off = (wm0 + frow) * ROWB, w_off = BM * ROWB + (wn0 + frow) * ROWB;

  f4 acc[MR][NR];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  };
  if constexpr (PP != 2) zero_acc();      // (8-phase tile: zeroed right before its loop -- 128 registers of zeros live across the prologue were spilled to scratch and reloaded)

  // ---- folded LayerNorm (consumer): thread r < BM collects the {sum, sum of squares} partials of tile row r. Issued behind the
  //      first tile loads, all slots in flight at once (slot order kept in the sums); turned into mean / rstd after the k-loop.
  //      Ping-pong tile: loaded AHEAD of the first tiles and folded at once (48 registers carried through the loop would spill, and a spill
  //      reload in the loop waits for vmcnt, i.e. drains the DMA queue); the prologue DMA stays in flight behind them.
  float ln_s1 = 0.f, ln_s2 = 0.f;
  constexpr int LN_MAXS = 24;
  float2 ln_v[LN_MAXS];
  // (fused attention tiles: pointer and slot count from the preloaded arguments of their kernels, qxattn.hip -- the same values as in `p`, without the argument block's round trip)
  // `p` behind a pointer the optimizer cannot see through before the point of the call (the leading arguments fill 14 dwords: `p` sits at byte 56 of the argument block):
  // scalar loads from the argument block are hoisted and speculated freely, and every one that lands in front of the first DMA piece is a cold round trip there
  auto p_opaque = [&]() -> const GemmArgs* {
    int off = 56;
    asm volatile("" : "+s"(off));
    return (const GemmArgs*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + off);
  };
  const float* ln_stats_p = nullptr;
  int ln_nslots = 0;
  if constexpr (XA != 0) { ln_stats_p = pre_ln_stats; ln_nslots = pre_ln_slots; }
  else if (hflags & 2) { const GemmArgs* pe = p_opaque(); ln_stats_p = pe->ln_stats; ln_nslots = pe->ln_slots; }      // (consumers only: the others never ask)
#ifdef IA2P_TIMING_NOSTATS      // (timing experiments only: what the statistics' loads cost a workgroup's start; results are wrong)
  const bool ln_row = false;
#else
  const bool ln_row = ln_stats_p && tid < BM && bm0 + tid < hM;
#endif
  const bool ln_wide = ln_row && ln_nslots <= LN_MAXS;
  // Round 6: the loads are ISSUED ahead of the prologue DMA and FOLDED behind it. vmcnt retires in order: issued behind the DMA (rounds 3-5), their wait was also a
  // wait for every piece of the prologue and then for their own round trip on top -- in the step 1.4 us (to_q + cross-attention) to 1.9 us (QKV + self-attention) per
  // workgroup start (timing build without the statistics: profiles/r06ad_nostats_timing.txt); issued first and waited for at once (round 2) the DMA started a round trip
  // late. Now both fly together and the counted wait of the fold leaves the younger DMA pieces in flight.
  if (XA == 0 ? __builtin_expect(ln_wide, 0) : ln_wide) {      // (the shared kernels' launches are mostly NOT consumers: the 24 loads sit out of line there)
    const float2* st = (const float2*)ln_stats_p + (bm0 + tid);
#pragma unroll
    for (int u = 0; u < LN_MAXS; ++u) ln_v[u] = st[(size_t)min(u, ln_nslots - 1) * hM];
  }
  __builtin_amdgcn_sched_barrier(0);
  auto fold_ln = [&]() {
    if (XA == 0 ? __builtin_expect(ln_wide, 0) : ln_wide) {
      // (opaque to the optimizer: without it the first addition -- and with it a full round trip's wait -- is hoisted to right behind the loads, ahead of the DMA issue)
#pragma unroll
      for (int u = 0; u < LN_MAXS; u += 12)
        asm volatile("" : "+v"(ln_v[u].x), "+v"(ln_v[u].y), "+v"(ln_v[u + 1].x), "+v"(ln_v[u + 1].y), "+v"(ln_v[u + 2].x), "+v"(ln_v[u + 2].y), "+v"(ln_v[u + 3].x), "+v"(ln_v[u + 3].y),
                          "+v"(ln_v[u + 4].x), "+v"(ln_v[u + 4].y), "+v"(ln_v[u + 5].x), "+v"(ln_v[u + 5].y), "+v"(ln_v[u + 6].x), "+v"(ln_v[u + 6].y), "+v"(ln_v[u + 7].x), "+v"(ln_v[u + 7].y),
                          "+v"(ln_v[u + 8].x), "+v"(ln_v[u + 8].y), "+v"(ln_v[u + 9].x), "+v"(ln_v[u + 9].y), "+v"(ln_v[u + 10].x), "+v"(ln_v[u + 10].y), "+v"(ln_v[u + 11].x), "+v"(ln_v[u + 11].y));
#pragma unroll
      for (int u = 0; u < LN_MAXS; ++u)
        if (u < ln_nslots) { ln_s1 += ln_v[u].x; ln_s2 += ln_v[u].y; }
    } else if (ln_row) {
      const float2* st = (const float2*)ln_stats_p + (bm0 + tid);
      for (int sl = 0; sl < ln_nslots; ++sl) { const float2 v = st[(size_t)sl * hM]; ln_s1 += v.x; ln_s2 += v.y; }
    }
  };
  const int nk = kt1 - kt0;
  constexpr int LPS = A_PW + B_PW;   // LDS-DMA pieces this wave issues per k-tile
  // NSTAGE-deep LDS ring: tiles kt+1 .. kt+NSTAGE-2 stay in flight across the barrier of step kt (counted vmcnt,
  // raw s_barrier -- cdna_hip_programming.md §5 "Pipelining across barriers"); ONE barrier per k-step.
  if constexpr (PP == 2) {      // 8-phase tile: k-tile 0 whole, k-tile 1 up to its third half (the fourth is issued in the first phase of the loop)
    stage_part(0, std::integral_constant<int, 0>{}); stage_part(0, std::integral_constant<int, 1>{});
    stage_part(0, std::integral_constant<int, 2>{}); stage_part(0, std::integral_constant<int, 3>{});
    if (nk > 1) { stage_part(1, std::integral_constant<int, 0>{}); stage_part(1, std::integral_constant<int, 1>{}); stage_part(1, std::integral_constant<int, 2>{}); }
  } else {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nk) stage(s, s);
  }
  __builtin_amdgcn_sched_barrier(0);
  fold_ln();               // (folded here, before the loop: 48 registers carried through it would spill)
  if (PP) asm volatile("" : "+v"(ln_s1), "+v"(ln_s2));
  // fused cross-attention: the context K / V of this tile's (batch element, head) travel to registers while the projection runs
  AttnKvRegs kvr;       // loaded inside the k-loop, behind the first tile     // folds now; the counted wait leaves the prologue DMA in flight
  IA2P_STAMP(const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();)      // (tools/micro/gemm_clock.hip: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the k-loop)
  if constexpr (PP == 1) {
    // Barrier sequence b0, b1, ...; interval I_n lies between b_n and b_n+1. Group 0 reads tile t in I_2t and multiplies it in I_2t+1; group 1
    // reads it in I_2t+1 and multiplies it in I_2t+2. Every wave waits for its DMA pieces of tile t before b_2t; the slot of tile t-1 is free
    // after b_2t (group 1 finished reading it in I_2t-1), so tile t+2 is issued into it in I_2t: two tiles stay in flight.
    static_assert(MR * NR <= 20, "ping-pong keeps the fragments of a whole k-tile in registers across a barrier");
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
    auto top = [&](int t, auto lps_tag) {      // b_2t: tile t has landed for every wave (tile t+1 may still be in flight: this wave's lps_tag pieces of it)
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) wait_vm_barrier<decltype(lps_tag)::value>();
      else wait_vm_barrier<0>();
      __builtin_amdgcn_sched_barrier(0);
    };
    constexpr int LPS0 = LPS, LPS1 = B_UNEVEN ? LPS - 1 : LPS;       // pieces per k-tile of a wave of group 0 / group 1
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
        top(t, std::integral_constant<int, LPS0>{});
        rd(slot_r);
        __builtin_amdgcn_sched_barrier(0);
        if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1, slot_s);
        mid();
        conv_prepare();
        mm();
        adv();
      }
    } else {
      for (int t = 0; t < nk; ++t) {
        top(t, std::integral_constant<int, LPS1>{});
        conv_prepare();
        if (t > 0) mm();
        mid();
        rd(slot_r);
        __builtin_amdgcn_sched_barrier(0);
        if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1, slot_s);
        adv();
      }
      mm();
    }
  } else if constexpr (PP == 3) {
    // ---- ping-pong on TWO k-tile slots (the fused QKV + self-attention tile: 256 x 192 leaves LDS for two stages of 56 KiB, not three). Barriers b0, b1, ...;
    //      group 0 (waves 0-3, rows 0-127) reads tile t in I_2t and multiplies it in I_2t+1; group 1 reads it in I_2t+1 and multiplies it in I_2t+2. EVERY wave
    //      issues its DMA pieces of tile t+1 in I_2t -- group 0 behind its fragment reads, group 1 ahead of its MFMAs -- into the slot of tile t-1, whose last
    //      reader (group 1, in I_2t-1) has retired its reads before b_2t; the pieces have two intervals to land and are waited for in front of b_2t+2.
    static_assert(MR * NR <= 24, "two-slot ping-pong keeps the fragments of a whole k-tile in registers across a barrier");
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
    auto top = [&]() {           // b_2t: tile t has landed for every wave (nothing else is in flight); this wave's fragment reads of the interval before are retired
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    auto mid = [&]() {           // b_2t+1
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    if (grp == 0) {
      for (int t = 0; t < nk; ++t) {
        top();
        rd(t & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 1 < nk) stage(t + 1, (t + 1) & 1);
        mid();
        mm();
      }
    } else {
      for (int t = 0; t < nk; ++t) {
        top();
        if (t + 1 < nk) stage(t + 1, (t + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (t > 0) mm();
        mid();
        rd(t & 1);
      }
      mm();
    }
  } else if constexpr (PP == 2) {
    // ---- 8-phase schedule (cdna_hip_programming.md §5 "The 256^2 8-phase template"): a k-tile is multiplied in FOUR phases, one 64-row x (BN/8)-column
    //      quadrant of the wave tile x K = 64 each; every phase = { fragment reads of the quadrant's new operand half, LDS-DMA of ONE half-tile of a later
    //      k-tile } -> barrier -> MFMA cluster -> barrier. Waves 4-7 run one barrier behind waves 0-3, so on every SIMD one wave multiplies while its
    //      partner reads and stages. Phase p of k-tile t (slot b = t & 1):
    //        P1  read b0 (B half 0), a0 (A half 0)   stage A1(t+1) -> slot b^1    MFMA a0 x b0
    //        P2  read b1 (B half 1)                  stage B0(t+2) -> slot b      MFMA a0 x b1
    //        P3  read a1 (A half 1, into a0's regs)  stage A0(t+2) -> slot b      MFMA a1 x b1
    //        P4  --                                  stage B1(t+2) -> slot b      MFMA a1 x b0      + counted vmcnt: k-tile t+1 has landed
    //      Hazards (two groups one barrier apart). Write-after-read: a half-tile is re-staged >= 2 phases after its last fragment read (A0: P1 -> P3, B1: P2 -> P4,
    //      A1: P3 -> next P1), or 1 phase after when the reads were retired BEFORE the reading phase's first barrier (B0: issued first in P1, lgkmcnt(8) before
    //      the barrier, re-staged in P2). Read-after-write: the counted vmcnt of P4 leaves the three half-tiles issued in P2..P4 in flight and retires all of
    //      k-tile t+1 (its last half, A1, was issued in P1); the first read of k-tile t+1 is one phase later, behind barriers both groups have passed.
    static_assert(A_PW % 2 == 0 && B_PW % 2 == 0 && MR % 2 == 0 && NR % 2 == 0, "8-phase: even pieces / fragments per half");
    constexpr int HLPS = A_HP + 2 * B_HP;        // this wave's LDS-DMA pieces of the three half-tiles B0, A0, B1 that stay in flight across P4
    h8 af[KSUB][MRH], wf0[KSUB][NRH], wf1[KSUB][NRH];
    auto rd_a = [&](int slot, auto half_tag) {
      constexpr int H = decltype(half_tag)::value;
      const char* base = smem + slot * STAGE;
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk) {
        const int coff = ((kk * 4 + fq) ^ fswz) << 4;
#pragma unroll
        for (int i = 0; i < MRH; ++i) af[kk][i] = *(const h8*)(base + a_off + frag_row(H * MRH + i) * ROWB + coff);
      }
    };
    auto rd_b = [&](int slot, auto half_tag, h8 (&wf)[KSUB][NRH]) {
      constexpr int H = decltype(half_tag)::value;
      const char* base = smem + slot * STAGE;
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk) {
        const int coff = ((kk * 4 + fq) ^ fswz) << 4;
#pragma unroll
        for (int j = 0; j < NRH; ++j) wf[kk][j] = *(const h8*)(base + w_off + frag_col(H * NRH + j) * ROWB + coff);
      }
    };
    auto mmq = [&](auto ah_tag, auto bh_tag, const h8 (&wf)[KSUB][NRH]) {
      constexpr int AHf = decltype(ah_tag)::value, BHf = decltype(bh_tag)::value;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
        for (int i = 0; i < MRH; ++i)
#pragma unroll
          for (int j = 0; j < NRH; ++j)
            acc[AHf * MRH + i][BHf * NRH + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][j], af[kk][i], acc[AHf * MRH + i][BHf * NRH + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    using S_B0 = std::integral_constant<int, 0>;
    using S_A0 = std::integral_constant<int, 1>;
    using S_B1 = std::integral_constant<int, 2>;
    using S_A1 = std::integral_constant<int, 3>;
    auto bar1 = [&]() {            // first barrier of a phase, then this wave's fragment reads have to be in
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    auto bar2 = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    // prologue (issued above): k-tile 0 whole (slot 0), k-tile 1 without its last half (slot 1)
    zero_acc();
    if (nk > 1) wait_vm_barrier<HLPS>(); else wait_vm_barrier<0>();
    if (wave >= NWAVE / 2) bar2();                // the second wave group runs one barrier behind the first
    auto ktile = [&](int t, auto slot_tag) {
      constexpr int SL = decltype(slot_tag)::value;
      // P1
      rd_b(SL, H0{}, wf0);
      __builtin_amdgcn_sched_barrier(0);          // issue order pinned: the B reads first (retired by the counted wait below)
      rd_a(SL, H0{});
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) stage_part(SL ^ 1, S_A1{});
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(KSUB * MRH) : "memory");      // all but the A reads: half-tile B0 of this slot is free for P2's DMA
      bar1();
      mmq(H0{}, H0{}, wf0);
      bar2();
      // P2
      rd_b(SL, H1{}, wf1);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < nk) stage_part(SL, S_B0{});
      bar1();
      conv_prepare();
      mmq(H0{}, H1{}, wf1);
      bar2();
      // P3
      rd_a(SL, H1{});
      __builtin_amdgcn_sched_barrier(0);
      if (t + 2 < nk) stage_part(SL, S_A0{});
      bar1();
      mmq(H1{}, H1{}, wf1);
      bar2();
      // P4
      if (t + 2 < nk) {
        stage_part(SL, S_B1{});
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(HLPS) : "memory");
      } else {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      bar1();
      mmq(H1{}, H0{}, wf0);
      bar2();
    };
    for (int t = 0; t < nk; t += 2) {
      ktile(t, std::integral_constant<int, 0>{});
      if (t + 1 < nk) ktile(t + 1, std::integral_constant<int, 1>{});
    }
    if (wave < NWAVE / 2) bar2();
  } else {
  int cur = 0, nxt = NSTAGE - 1;      // ring slots: `cur` is consumed this step, `nxt` is refilled
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;    // tiles issued after tile kt that may remain in flight
    // tiles kt+1 .. kt+NSTAGE-2 were issued before this wait and may stay in flight (fewer at the tail): vmcnt counts this wave's pieces
    wait_ring<NSTAGE - 2, LPS>(ahead < NSTAGE - 2 ? ahead : NSTAGE - 2);
    IA2P_STAMP(if (IA2P_STAMP_AT == 4 && kt == 0) stamp_put(p, nsplit, 7);)      // the first k-tile has landed for every wave (tools/insitu_stamps.py: what a launch's cold start costs)
    // every wave has passed the barrier => tile kt has landed for all, and slot `nxt` (read in step kt-1) is free
    if (kt + NSTAGE - 1 < nk) stage(kt + NSTAGE - 1, nxt);
    if constexpr (XA != 0 && XA != 4) {
      if (kt == 0) attn_kv_load(*xa, bm0 / xa->Nq, tn, tid, kvr);
    }
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
      conv_prepare();
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
        if (kk == 0) conv_prepare();
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
  // ---- epilogue (gemm_epilogue.h)
  TileCtx tc;
  tc.hM = hM; tc.hN = hN; tc.hK = hK; tc.tm = tm; tc.tn = tn; tc.bm0 = bm0; tc.bn0 = bn0; tc.tiles_m = tiles_m; tc.tiles_n = tiles_n; tc.split = split; tc.nsplit = nsplit;
  tc.h_m0 = 0; tc.ln_s1 = ln_s1; tc.ln_s2 = ln_s2;
  IA2P_STAMP(
    if (tid == 0 && stamp_base(p, nsplit)) {      // (the stamps go to a buffer nothing else reads)
      unsigned long long* o = stamp_base(p, nsplit) + 8 * blockIdx.x;
      const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
      o[0] = __builtin_amdgcn_s_memtime() - stamp_c0; o[1] = r1 - stamp_r0; o[2] = stamp_entry; o[3] = stamp_r0; o[4] = r1;
    }
  )
  tile_epilogue<BM, BN, NSTAGE, WGM, BK, PP, WGN, XA, 0>(acc, tc, p, xa, kvr);
}

template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int BK = 64, int PP = 0, int WGN = 2>
// Leading arguments = what a workgroup needs before its first DMA piece, in FOURTEEN dwords: that is how many the command processor preloads into SGPRs (16 user SGPRs, two
// of them the argument block's address; `.amdhsa_user_sgpr_kernarg_preload_length 14`). Rounds 2-5 spelled the row map and the K split / tile order out as five ints, 16 dwords
// in all: the last two -- hsplitk and hgroup_w, which the tile decode needs first -- were NOT preloaded, and every launch began with a cold scalar read of its argument block.
// Packed (ia2p_pack_rowmap / ia2p_pack_skgw, with the two flags of the prologue: the launcher refuses values that do not fit).
__global__ __launch_bounds__(WGM * WGN * 64, 2) void gemm_f16_kernel(const half_t* hA, const half_t* hW, const half_t* hzero, int hM, int hN, int hK, int hlda, int hldw, int hrowmap,
                                                                         int hbstride, int hsk_gw, const GemmArgs p) {
  gemm_tile_body<BM, BN, NSTAGE, CONV, WGM, BK, PP, WGN, 0>(hA, hW, hzero, hM, hN, hK, hlda, hldw, hrowmap & 0xffff, hbstride, (int)((unsigned)hrowmap >> 16), IA2P_SKGW_LO8(hsk_gw),
                                                            IA2P_SKGW_GW(hsk_gw), IA2P_SKGW_FLAGS(hsk_gw), p, nullptr);
}
template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int BK = 64, int PP = 0, int WGN = 2>
static hipError_t launch_cfg(const GemmArgs& a, hipStream_t s) {
  constexpr int smem = EpiCfg<BM, BN, NSTAGE, WGM, BK, WGN, PP>::SMEM;
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
  ia2p_gemm_prepare(b, smem, BM, BN, CONV);
  if (a.geglu && !b.vec8) return hipErrorInvalidValue;
  if (a.gn.st0) return hipErrorInvalidValue;                               // a GroupNorm-fused operand: the halo-staged convolution only (conv_halo_kernel.h)
  if (a.gn_out && (!b.vec8 || a.geglu)) return hipErrorInvalidValue;      // (the column sums of the output are taken on the 16-byte epilogue routes)
  if (a.geglu && !b.phi_lut) return hipErrorOutOfMemory;
  if (!CONV && IA2P_LIN_BUF) {      // buffer-load staging addresses an operand with a 31-bit byte offset
    const size_t a_rows = a.rpb ? ((size_t)a.M / a.rpb + 1) * (size_t)std::max(a.bstride, 0) + a.roff + a.rpb : (size_t)a.M;
    if (!ia2p_fits_buffer(a_rows, a.lda) || !ia2p_fits_buffer(a.N, a.ldw)) return hipErrorInvalidValue;
  }
  // b.sk_counters: as the caller (launch_any, gemm.hip) attached them -- null: the K-slices only write their slabs and a splitk_reduce_kernel launch finishes
  if (b.sk_counters && tiles > ia2p_sk_counter_capacity()) return hipErrorInvalidValue;
  const int extra = (!PP && a.pf && a.pf_bytes >= 4096) ? a.pf_blocks : 0;
  if ((long)tiles * (a.splitk > 1 ? a.splitk : 1) + extra >= (1L << 21)) return hipErrorInvalidValue;      // (udiv_small in the tile decode: block and tile counts below 2^21)
  int rowmap, sk_gw;
  if (!ia2p_pack_rowmap(b.rpb, b.roff, &rowmap) || !ia2p_pack_skgw(b.splitk, b.group_w, b.m_fastest, b.ln_stats != nullptr, &sk_gw)) return hipErrorInvalidValue;
  hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, NSTAGE, CONV, WGM, BK, PP, WGN>), dim3(tiles * (a.splitk > 1 ? a.splitk : 1) + extra), dim3(WGM * WGN * 64), smem, s,
                     b.A, b.W, b.zero, b.M, b.N, b.K, b.lda, b.ldw, rowmap, b.bstride, sk_gw, b);
  return hipGetLastError();
}
