// The halo-staged 3x3 convolution (conv_halo_f16_kernel): stride 1, one pixel of zero padding, the source itself or its nearest-x2 upsampled view, optionally with
// appended 1x1 blocks (a ResnetBlock2D's conv2 + conv_shortcut as ONE implicit GEMM). Replaces the 3x3 Conv2d modules of diffusers' ResnetBlock2D / Upsample2D behind the
// reference's UNet call (instructany2pix/ddim/pnp_pipeline.py:253-260; in-tree twin: llm/model/vae/modules/blocks.py:122-142).
//
// 256 x BN tiles on the ping-pong schedule (8 waves: waves 0-3 own the upper half of the tile rows, waves 4-7 the lower half, the two groups half a k-step apart, one
// workgroup per CU, 3-slot weight ring). The tile is a 16 x 16 pixel PATCH of one image, and the activation operand is not staged k-tile by k-tile (nine taps = nine fetches
// of nearly the same pixels through the fabric) but once per block of 64 channels, as the patch plus its one-pixel border: 18 x 18 pixels x 128 B, 144-B pixel pitch
// (8 data chunks + 1 pad chunk: fragment reads of 16 consecutive pixels hit 16 different 16-byte bank groups), two such images (the next block's lands while this one is
// multiplied). A filter tap is then a CONSTANT byte offset on the fragment reads -- the k-loop has no gather arithmetic at all -- and the L2 -> LDS traffic of a k-tile drops
// from (256 + BN) x 128 B to (36 + BN) x 128 B. K is walked block-major (nine taps of a block, then the next block; appended 1x1 blocks: the centre tap of their own image)
// over the SAME packed weights: the weight tile of (block, tap) starts at column tap * Cin + block * 64 of the [Co][tap][Cin] row.
// The epilogue is the shared one (gemm_epilogue.h, HALO = 1: tile row r = pixel (r >> 4, r & 15) of the patch).
//
// GN = 1 (round 5): GroupNorm + SiLU FUSED into the convolution that consumes it (ResnetBlock2D: conv1(silu(norm1(x))), conv2(silu(norm2(h)))): the operand is the RAW
// tensor -- one, or the up path's [hidden | skip] pair that is never concatenated -- and every halo image is normalised IN LDS before the taps read it:
//   * statistics come from the PRODUCER's epilogue (per M-tile and channel {sum, sum of squares}, gn_fold.h); every workgroup folds the slots of ITS image into the 32
//     group {mean, rstd} pairs while its first image and weight tiles are in flight (the sums meet in the halo-image buffer that is still free);
//   * per block of 64 channels one wave turns {mean, rstd, gamma, beta} into a 512-byte scale / shift table (written at tap 0 of the block before, gamma / beta loaded
//     one tap earlier, behind that interval's counted wait);
//   * every wave normalises the 1-KiB pieces IT fetched: piece k of the next block's image is issued at tap k, has landed for its own wave two taps later (in-order
//     vmcnt, no barrier needed for a wave's own data) and is rewritten in place -- LDS -> fma -> SiLU -> fp16 -> LDS -- beside the MFMAs of tap k + 2; chunks outside
//     the image and the pad chunks are left alone: they stay exactly zero, the reference pads the ACTIVATED tensor.
//   Bit-identical to gn_apply_stats_kernel (norm.hip) followed by the plain kernel: same fold, same scale / shift, same element formula, rounded to fp16 at the same point.
#pragma once
#include "gemm_epilogue.h"

// wave arrangement: 4 x 2 waves of 64 x (BN / 2); the 80-wide tile 8 x 1 waves of 32 x 80 (N = 640 / 1280 problems at M = 8192 / 2048 in 256 / 128 tiles of one per CU:
// the 32^2 maps fill the chip without a K split, the 16^2 maps with two slices instead of three)
template <int BN> struct HaloWaves { static constexpr int WGM = BN == 80 ? 8 : 4, WGN = BN == 80 ? 1 : 2; };

// LDS of a launch: weight ring + two halo images (+ GN: group statistics and the scale / shift table of a block) or the shared epilogue's needs, whichever is larger
template <int BN, int GN> struct HaloSmem {
  using BS = BStage<BN, 64, 8, 1>;
  static constexpr int H_BASE = 3 * BS::BNL * 128, H_BYTES = 48 * 1024;
  static constexpr int GN_GSTAT = H_BASE + 2 * H_BYTES, GN_TBL = GN_GSTAT + 512, GN_ZERO = GN_TBL + 512, LOOP = GN ? GN_ZERO + 512 : H_BASE + 2 * H_BYTES;
  static constexpr int EPI = EpiCfg<256, BN, 3, HaloWaves<BN>::WGM, 64, HaloWaves<BN>::WGN, 1>::SMEM;
  static constexpr int SMEM = LOOP > EPI ? LOOP : EPI;
  static_assert(SMEM <= 160 * 1024, "halo-staged tile exceeds the CU's LDS");
};

// The normalisation of an image piece (8 elements per lane: fma, exp, 1 / (1 + e), product -- gn_fold.h's four steps) goes BESIDE the MFMAs of a tap, one step per MFMA in
// program order: GN_OPS[k] = {step, element} placed behind MFMA LEAD + k, software-pipelined over the elements (a step's operand was produced three MFMAs earlier). Left
// to the scheduler the whole piece sits in front of the first MFMA (the matrix pipe idles through ~500 cycles of VALU issue, six taps out of nine); a
// sched_group_barrier pipeline interleaves it but re-orders the MFMAs into back-to-back pairs on ONE accumulator (measured: +3 us per block of 64 channels).
struct GnOp { int step, e; };
constexpr int GN_NOPS = 32;
constexpr GnOp GN_OPS[GN_NOPS] = {{0, 0}, {0, 1}, {0, 2}, {1, 0}, {0, 3}, {1, 1}, {2, 0}, {0, 4}, {1, 2}, {2, 1}, {3, 0}, {0, 5}, {1, 3}, {2, 2}, {3, 1}, {0, 6},
                                  {1, 4}, {2, 3}, {3, 2}, {0, 7}, {1, 5}, {2, 4}, {3, 3}, {1, 6}, {2, 5}, {3, 4}, {1, 7}, {2, 6}, {3, 5}, {2, 7}, {3, 6}, {3, 7}};

template <int BN, int GN>
__device__ __forceinline__ void conv_halo_tile_body(const half_t* hA, const half_t* hW, int hM, int hN, int hK, int hlda, int hldw, int hsplitk, int hgroup_w, int hmfast, int gHo, int gWo, int gHs, int gWs,
                                                    int gUp, int gCin, int gCin2, const GemmArgs& p) {
  constexpr int BM = 256, BK = 64, NSTAGE = 3, PP = 1, WGM = HaloWaves<BN>::WGM, WGN = HaloWaves<BN>::WGN;
  constexpr int NWAVE = WGM * WGN;
  static_assert(NWAVE == 8, "ping-pong schedule: 8 waves (two groups of 4 by tile rows)");
  constexpr int WM = BM / WGM, WN = BN / WGN;    // wave tile
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int ROWB = 2 * BK, CPR = ROWB / 16, RPP = 1024 / ROWB;   // row bytes, chunks per row, rows per 1-KiB staging piece
  using BS = BStage<BN, BK, NWAVE, PP>;
  constexpr int A_PW = BM / RPP / NWAVE, B_PW = BS::HI;             // staging pieces per wave: activation rows of an appended 1x1 tile, weight rows (BStage: the second wave group may take one piece less)
  constexpr bool B_UNEVEN = BS::UNEVEN;
  constexpr int BNL = BS::BNL;                                       // weight rows held in LDS (>= BN)
  static_assert(BN % 16 == 0 && WN % 16 == 0 && WM % 16 == 0, "tile / wave layout");
  constexpr int STAGE = BNL * ROWB;                                    // bytes of a ring slot: the weight tile only
  constexpr int H_PITCH = 144, H_ROW = 18 * H_PITCH;                   // halo image: bytes per pixel (8 chunks + 1 pad), per row of 18 pixels
  constexpr int H_SLOTS = ((18 * 18 * 9 + 63) / 64 + NWAVE - 1) / NWAVE, H_PIECES = H_SLOTS * NWAVE, H_BYTES = H_PIECES * 1024;      // 1-KiB DMA pieces of an image per wave (6), per image (48: the last two are padding, so that every wave issues the same count)
  constexpr int H_BASE = NSTAGE * STAGE;                               // the two halo images sit behind the weight ring
  static_assert(H_BASE == HaloSmem<BN, GN>::H_BASE && H_BYTES == HaloSmem<BN, GN>::H_BYTES, "LDS layout");
  constexpr int KSUB = BK / 32;                                       // 32-deep MFMA sub-steps per k-tile
  extern __shared__ __attribute__((aligned(1024))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int OOB = 0x7fffff00;      // a byte offset past any operand: the buffer load's range check writes zeros (no zero page, no select)
  IA2P_STAMP(const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();)

  // ---- tile of this workgroup: the 16 x 16 patch (h_y0, h_x0) of image h_img; tile row r = pixel (r >> 4, r & 15) of it
  const int tiles_m = hM / BM, tiles_n = (hN + BN - 1) / BN;
  int bid = blockIdx.x;
  const int nsplit = hsplitk > 1 ? hsplitk : 1;
  const int split = bid < tiles_m * tiles_n ? 0 : udiv_small(bid, tiles_m * tiles_n);
  bid -= split * tiles_m * tiles_n;
  int tm, tn;
  tile_order(bid, tiles_m, tiles_n, hgroup_w, hmfast, tm, tn);
  const int bm0 = tm * BM, bn0 = tn * BN;
  const int tpr = gWo >> 4, tpi = (gHo >> 4) * tpr;
  const int h_img = tm / tpi;
  const int h_rem = tm - h_img * tpi, h_ty = h_rem / tpr;
  const int h_y0 = h_ty * 16, h_x0 = (h_rem - h_ty * tpr) * 16;
  const int h_m0 = (h_img * gHo + h_y0) * gWo + h_x0;
  auto row_m = [&](int r) { return h_m0 + (r >> 4) * gWo + (r & 15); };

  // ---- staging addresses. Halo image: DMA piece (slot * NWAVE + wave), lane -> 16-byte chunk j = piece * 64 + lane = (halo pixel j / 9, chunk j % 9);
  //      h_voff = byte offset of that chunk inside the source tensor (channel block 0), or OOB: pad chunk / outside the image / past the image's last pixel
  const int srow = lane / CPR, cpos = lane % CPR;
  auto a_piece = [&](int i) { return wave * A_PW + i; };
  int h_voff[H_SLOTS];
  int h_voff2[GN ? H_SLOTS : 1];
#pragma unroll
  for (int sl = 0; sl < H_SLOTS; ++sl) {
    const int j = (sl * NWAVE + wave) * 64 + lane;
    const int hp = j / 9, c = j - hp * 9;
    const int hy = hp / 18, hx = hp - hy * 18;
    const int y = h_y0 - 1 + hy, x = h_x0 - 1 + hx;
    const bool ok = c < 8 && hp < 18 * 18 && (unsigned)y < (unsigned)gHo && (unsigned)x < (unsigned)gWo;
    // (nearest-x2 upsampled view: the image in LDS IS the upsampled patch -- pixel (y, x) of it comes from source pixel (y / 2, x / 2), fetched up to four times out of L2)
    h_voff[sl] = ok ? (((h_img * gHs + (y >> gUp)) * gWs + (x >> gUp)) * hlda + c * 8) * 2 : OOB;
    if constexpr (GN != 0) h_voff2[sl] = ok ? (((h_img * gHs + y) * gWs + x) * p.lda1b + c * 8) * 2 : OOB;      // (second source of a GroupNorm-fused launch: its own row stride; no upsampled view there)
  }
  // this wave's weight pieces: [b_pi0, b_pi0 + b_npw); piece `pi` covers tile columns pi*8 .. pi*8+7: lane -> (row pi*8 + lane/8, LDS chunk lane%8 holding global chunk (lane%8) ^ swz(row))
  int w_voff[B_PW];
  const int b_npw = B_UNEVEN && wave >= NWAVE / 2 ? B_PW - 1 : B_PW;
  const int b_pi0 = B_UNEVEN ? (wave < NWAVE / 2 ? wave * B_PW : (NWAVE / 2) * B_PW + (wave - NWAVE / 2) * (B_PW - 1)) : wave * B_PW;
  auto b_piece = [&](int i) { return b_pi0 + i; };
#pragma unroll
  for (int i = 0; i < B_PW; ++i) {
    const int pi = min(b_piece(i), BNL / RPP - 1);
    const int n = bn0 + pi * RPP + srow;
    const int gch = cpos ^ lds_swz<BK>(pi * RPP + srow);
    w_voff[i] = (n < hN && pi * RPP + srow < BN) ? (n * hldw + gch * 8) * 2 : OOB;      // byte offset of the piece's chunk in the weight matrix, column 0
  }

  const int nk_all = hK / BK;
  // this workgroup's k-tiles (no K split: all of them -- the 64-bit divisions this was written with ran in every launch: ~200 instructions of a workgroup's start)
  const int kt0 = nsplit == 1 ? 0 : udiv_small(split * nk_all, nsplit), kt1 = nsplit == 1 ? nk_all : udiv_small((split + 1) * nk_all, nsplit);
  int cin_main = gCin, cin_extra = gCin2;                 // (two named scalars: a select between two argument FIELDS became a 2-entry table in scratch)
  asm volatile("" : "+s"(cin_main), "+s"(cin_extra));

  // ---- fragment read offsets (wave tile origin is a multiple of 16, so swz(row) = (lane>>1)&7)
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int frow = lane & 15, fq = lane >> 4;
  const int fswz = lds_swz<BK>(frow);
  const int a_off = (wm0 + frow) * ROWB, w_off = (wn0 + frow) * ROWB;

  f4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = (f4){0.f, 0.f, 0.f, 0.f};
  IA2P_STAMP(const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();)
  // ---- halo-staged convolution on the ping-pong schedule (barriers and groups as in the plain ping-pong loop below). Everything a k-tile does is known at compile
  //      time: the nine taps of a block are nine straight-line bodies (tap offset of the fragment reads, ring slots t mod 3 = tap mod 3, the image piece this wave
  //      fetches -- piece `tap` of the next block's image for taps 0..5 --, the counted vmcnt), operands come through BUFFER loads to LDS (descriptor + one 32-bit
  //      VGPR offset per piece + a scalar offset for the channel block / weight column: no pointer arithmetic, rows and pixels outside the operand are range-
  //      checked to zero by the load), and launches past the end of the K range still issue their loads against an EMPTY descriptor, so that every k-tile of
  //      every wave has the same number of pieces in flight. What bounds a ping-pong k-tile is the issue of its DMA pieces in the read half-step
  //      (100 ... 185 cycles each beside 18 ds_read_b128): 6.5 per wave in the gathered 256 x 160 tile, 3.5 here.
  static_assert(MR * NR <= 20 && NSTAGE == 3, "ping-pong keeps the fragments of a whole k-tile in registers across a barrier; ring slot = tap mod 3");
  const int grp = wave >> 2;
  const half_t* src2 = p.A2;
  const half_t* src3 = p.A3;
  int ld2 = p.lda2, ld3 = p.lda3;
  asm volatile("" : "+s"(src2), "+s"(src3), "+s"(ld2), "+s"(ld3));      // (named scalars: a select between FIELDS of the by-value argument struct goes through scratch)
  const int nb_main = cin_main / BK, nk_main = 9 * nb_main, cin2 = cin_main * 2;
  // this workgroup's k-tiles [k0, k1); inside the 3x3 part a K split starts and ends on whole blocks (the same rounding on both sides of a boundary)
  int k0 = kt0, k1 = kt1;
  if (k0 < nk_main) k0 -= k0 % 9;
  if (k1 < nk_main) k1 -= k1 % 9;
  const int nkt = k1 - k0;
  const int blk0 = k0 < nk_main ? k0 / 9 : nb_main, blk1 = min(k1, nk_main) / 9 > blk0 ? min(k1, nk_main) / 9 : blk0;      // its blocks of the 3x3 part
  const int n2 = nkt - 9 * (blk1 - blk0);                                                                               // its tiles of the appended 1x1 blocks
  auto mk_rsrc = [](const void* q, size_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)q, 0, (int)min(bytes, (size_t)0x7ffffe00), 0x00020000); };
  const __amdgpu_buffer_rsrc_t rs_w = mk_rsrc(hW, (size_t)hN * hldw * 2), rs_a = mk_rsrc(hA, (size_t)(hM / (gHo * gWo)) * gHs * gWs * hlda * 2), rs_none = mk_rsrc(hW, 0);
  // GroupNorm-fused launch: channel blocks [0, nb0) come from A, [nb0, nb_main) from A1b (one source: nb0 = nb_main)
  const int nb0 = GN ? p.gn.C0 / BK : nb_main;
  const __amdgpu_buffer_rsrc_t rs_a1b = GN ? mk_rsrc(p.A1b, p.A1b ? (size_t)(hM / (p.Ho * p.Wo)) * p.Hs * p.Ws * p.lda1b * 2 : 0) : rs_none;
  auto issue_w = [&](__amdgpu_buffer_rsrc_t rs, int slot, int soff) {
#pragma unroll
    for (int i = 0; i < B_PW; ++i)
      if (!B_UNEVEN || i < b_npw) BLDS16(rs, smem + slot * STAGE + b_piece(i) * 1024, w_voff[i], soff);
  };
  h8 af[KSUB][MR], wf[KSUB][NR];
  auto mm = [&]() {
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
  };
  auto mid = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int LPS0 = B_PW, LPS1 = B_UNEVEN ? B_PW - 1 : B_PW;       // weight pieces per k-tile of a wave of group 0 / group 1
  const int a_rd0 = H_BASE + ((wm0 >> 4) * 18 + frow) * H_PITCH + fq * 16;      // this lane's pixel (tap (0, 0) of it) and 16-byte chunk inside a halo image
  int a_rd = 0;                                                                  // + the image of the current block
  int t = 0;                                                                     // k-tile of this workgroup at the top of the current block
  // fragment reads of tap TAP of the current block, then -- behind them -- piece TAP of the next block's image and the weights of the tile two ahead
  auto rd_tap = [&](auto tap_tag) {
    constexpr int TAP = decltype(tap_tag)::value;
    const char* hb = smem + a_rd + ((TAP / 3) * 18 + TAP % 3) * H_PITCH;
    const char* bb = smem + (TAP % 3) * STAGE + w_off;
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) {
      const int coff = ((kk * 4 + fq) ^ fswz) << 4;
#pragma unroll
      for (int i = 0; i < MR; ++i) af[kk][i] = *(const h8*)(hb + i * H_ROW + kk * 64);
#pragma unroll
      for (int j = 0; j < NR; ++j) wf[kk][j] = *(const h8*)(bb + j * 16 * ROWB + coff);
    }
  };
  auto issue_tap = [&](auto tap_tag, int blk) {
    constexpr int TAP = decltype(tap_tag)::value;
    if constexpr (TAP < H_SLOTS) {
      if constexpr (GN != 0) {
        // (two copies of the load under a wave-uniform branch, not a select between h_voff and h_voff2: a select between two register ARRAYS sends both to scratch)
        if (blk + 1 >= nb0) BLDS16(blk + 1 < blk1 ? rs_a1b : rs_none, smem + H_BASE + ((blk + 1) & 1) * H_BYTES + (TAP * NWAVE + wave) * 1024, h_voff2[TAP], (blk + 1 - nb0) * (2 * BK));
        else BLDS16(blk + 1 < blk1 ? rs_a : rs_none, smem + H_BASE + ((blk + 1) & 1) * H_BYTES + (TAP * NWAVE + wave) * 1024, h_voff[TAP], (blk + 1) * (2 * BK));
      } else BLDS16(blk + 1 < blk1 ? rs_a : rs_none, smem + H_BASE + ((blk + 1) & 1) * H_BYTES + (TAP * NWAVE + wave) * 1024, h_voff[TAP], (blk + 1) * (2 * BK));
    }
    constexpr int S = TAP + 2;                                   // tap of the tile two ahead (9, 10: taps 0, 1 of the next block, or the first appended tiles)
    int soff;
    if constexpr (S <= 8) soff = blk * (2 * BK) + S * cin2;
    else soff = blk + 1 < nb_main ? (blk + 1) * (2 * BK) + (S - 9) * cin2 : 9 * cin2 + (S - 9) * (2 * BK);
    issue_w(t + S < nkt ? rs_w : rs_none, S % 3, soff);
  };
  auto top = [&](auto n_tag) {
    __builtin_amdgcn_sched_barrier(0);
    wait_vm_barrier<decltype(n_tag)::value>();
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- GroupNorm-fused launch: the pieces (see the head of this file)
  float2* gn_gstat = (float2*)(smem + HaloSmem<BN, GN>::GN_GSTAT);      // [groups] {mean, rstd} of this workgroup's image
  char* gn_tbl = smem + HaloSmem<BN, GN>::GN_TBL;                       // scale / shift of the 64 channels of a block: planes {a[0..3], a[4..7], b[0..3], b[4..7]} x 8 chunks x 16 B
  // LDS address of the scale / shift entries of this lane's chunk in DMA slot sl (chunk (j mod 9) of 9 per halo pixel, j = (sl * 8 + wave) * 64 + lane) -- or, for chunks outside
  // the image and pad chunks, of a 512-byte block of zeros: scale = shift = 0 turns their zeros into silu(0) = 0, no branch, no select on the data
  int gn_toff[GN ? H_SLOTS : 1];
  if constexpr (GN != 0) {
#pragma unroll
    for (int sl = 0; sl < H_SLOTS; ++sl) {
      const int c = ((sl * NWAVE + wave) * 64 + lane) % 9;
      gn_toff[sl] = h_voff[sl] != OOB ? HaloSmem<BN, GN>::GN_TBL + c * 16 : HaloSmem<BN, GN>::GN_ZERO;
    }
  }
  const int gn_qoff = H_BASE + wave * 1024 + lane * 16;                   // this lane's 16 bytes inside piece (sl, wave) of image buffer 0
  float gn_gam = 0.f, gn_bet = 0.f;                                      // wave 0: gamma / beta of channel 64 blk + lane of the block whose table is built next
  auto gn_load_gb = [&](int blk) {                                       // (two 2-byte loads, issued BEHIND a counted wait and ahead of the interval's DMA pieces: the next counted wait covers them)
    if constexpr (GN != 0) {
      if (wave == 0 && blk < blk1) { gn_gam = (float)p.gn.gamma[blk * BK + lane]; gn_bet = (float)p.gn.beta[blk * BK + lane]; }
    }
  };
  auto gn_build_tbl = [&](int blk) {                                     // wave 0, lane = channel 64 blk + lane of the concatenated input
    if constexpr (GN != 0) {
      if (wave == 0 && blk < blk1) {
        const float2 ab = gn_scale_shift(gn_gstat[(blk * BK + lane) / p.gn.gs], gn_gam, gn_bet);
        char* q = gn_tbl + ((lane & 7) >> 2) * 128 + (lane >> 3) * 16 + (lane & 3) * 4;
        *(float*)q = ab.x;
        *(float*)(q + 256) = ab.y;
      }
    }
  };
  // piece SL of the image in buffer `buf` (this wave's own DMA piece: landed for THIS wave once its vmcnt has retired it), normalised in place
  auto gn_norm_piece = [&](auto sl_tag, int buf) {
    if constexpr (GN != 0) {
      constexpr int SL = decltype(sl_tag)::value;
      char* q = smem + gn_qoff + buf * H_BYTES + SL * NWAVE * 1024;
      const h8 v = *(const h8*)q;
      const char* tb = smem + gn_toff[SL];
      const f4 a0 = *(const f4*)tb, a1 = *(const f4*)(tb + 128), b0 = *(const f4*)(tb + 256), b1 = *(const f4*)(tb + 384);
      h8 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o[e] = (half_t)gn_apply_f((float)v[e], a0[e], b0[e], true); o[4 + e] = (half_t)gn_apply_f((float)v[4 + e], a1[e], b1[e], true); }
      *(h8*)q = o;                                                       // (chunks outside the image / pad chunks: 0 * 0 + 0 -> silu(0) = 0: they stay zero, the reference pads the activated tensor)
    }
  };
  // piece SL of image buffer `buf` and its scale / shift entries, LDS -> registers: issued in the wave's READ half-step behind the fragment reads, where the barrier's
  // own lgkmcnt(0) retires them -- read in the multiply half-step they would stall the in-order issue of the MFMAs behind them for a whole LDS round trip under load
  h8 gn_v = {0, 0, 0, 0, 0, 0, 0, 0};
  f4 gn_a0 = {0.f, 0.f, 0.f, 0.f}, gn_a1 = gn_a0, gn_b0 = gn_a0, gn_b1 = gn_a0;
  auto gn_fetch = [&](auto sl_tag, int buf) {
    if constexpr (GN != 0) {
      constexpr int SL = decltype(sl_tag)::value;
#if !(defined(IA2P_GN_ABL) && (IA2P_GN_ABL & 1))      // (ablation builds, tools/conv_gn_probe.py: timing only, wrong results)
      gn_v = *(const h8*)(smem + gn_qoff + buf * H_BYTES + SL * NWAVE * 1024);
      const char* tb = smem + gn_toff[SL];
      gn_a0 = *(const f4*)tb; gn_a1 = *(const f4*)(tb + 128); gn_b0 = *(const f4*)(tb + 256); gn_b1 = *(const f4*)(tb + 384);
#endif
    }
  };
  // the MFMAs of a tap with the normalisation of the fetched piece (SL of image buffer `buf`) between them (same MFMA order as mm())
  auto mm_gn = [&](auto sl_tag, int buf) {
    if constexpr (GN != 0) {
      constexpr int SL = decltype(sl_tag)::value;
#ifndef IA2P_GN_LEAD
#define IA2P_GN_LEAD 4        // MFMAs ahead of the first step; IA2P_GN_EVERY: a step behind every N-th MFMA (build-time knobs for A/B builds: tools/gn_ablation.sh)
#endif
#ifndef IA2P_GN_EVERY
#define IA2P_GN_EVERY 1
#endif
      constexpr int NM = KSUB * MR * NR, LEAD = NM >= GN_NOPS + 4 ? (IA2P_GN_LEAD < NM ? IA2P_GN_LEAD : NM) : 2, STRIDE = NM - LEAD >= GN_NOPS ? 1 : 2;      // (tiles with fewer MFMAs than steps take two steps per MFMA)
      char* q = smem + gn_qoff + buf * H_BYTES + SL * NWAVE * 1024;
      const h8 v = gn_v;
      const f4 a0 = gn_a0, a1 = gn_a1, b0 = gn_b0, b1 = gn_b1;
      float gy[8], gt[8];
      h8 o;
      auto op = [&](int k) {
        if (k < 0 || k >= GN_NOPS) return;
#if defined(IA2P_GN_ABL) && (IA2P_GN_ABL & 2)
        if (GN_OPS[k].step == 3) o[GN_OPS[k].e] = v[GN_OPS[k].e];
        return;
#endif
        // (every step's result passes through an empty volatile asm: such statements keep their program order among themselves and the MFMA statements, which pins the
        //  step between ITS two MFMAs -- the instruction selector's own list scheduler would otherwise sink all of them to their use, the LDS write behind the last MFMA)
        const int st = GN_OPS[k].step, e = GN_OPS[k].e;
        if (st == 0) { gy[e] = gn_step_y((float)v[e], e < 4 ? a0[e & 3] : a1[e & 3], e < 4 ? b0[e & 3] : b1[e & 3]); asm volatile("" : "+v"(gy[e])); }
        else if (st == 1) { gt[e] = gn_step_e(gy[e]); asm volatile("" : "+v"(gt[e])); }
        else if (st == 2) { gt[e] = gn_step_r(gt[e]); asm volatile("" : "+v"(gt[e])); }
        else { float pr = gn_step_o(gy[e], gt[e]); asm volatile("" : "+v"(pr)); o[e] = (half_t)pr; }
      };
      int m = 0;
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
        for (int i = 0; i < MR; ++i)
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            // (the MFMA as a volatile asm statement: statements of that kind keep their program order and nothing is scheduled across them, so the steps written between
            //  two MFMAs ARE issued between them -- the instruction itself is the one mm() emits)
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i][j]) : "v"(wf[kk][j]), "v"(af[kk][i]));
            if (m >= LEAD && (m - LEAD) % IA2P_GN_EVERY == 0) {
#pragma unroll
              for (int u = 0; u < STRIDE; ++u) op((m - LEAD) / IA2P_GN_EVERY * STRIDE + u);
            }
            ++m;
          }
#pragma unroll
      for (int k = (NM - LEAD + IA2P_GN_EVERY - 1) / IA2P_GN_EVERY * STRIDE; k < GN_NOPS; ++k) op(k);      // (whatever did not fit beside the MFMAs)
#if defined(IA2P_GN_ABL) && (IA2P_GN_ABL & 4)
      asm volatile("" :: "v"(o));
      (void)q;
#else
      *(h8*)q = o;                                                       // (chunks outside the image / pad chunks: 0 * 0 + 0 -> silu(0) = 0: they stay zero, the reference pads the activated tensor)
#endif
    }
  };
  if (blk1 > blk0) {
    // prologue: the first block's image, the weights of its first two tiles
    if (GN && blk0 >= nb0) {      // (a K slice that starts inside the second source)
#pragma unroll
      for (int sl = 0; sl < H_SLOTS; ++sl) BLDS16(rs_a1b, smem + H_BASE + (blk0 & 1) * H_BYTES + (sl * NWAVE + wave) * 1024, h_voff2[GN ? sl : 0], (blk0 - nb0) * (2 * BK));
    } else {
#pragma unroll
      for (int sl = 0; sl < H_SLOTS; ++sl) BLDS16(rs_a, smem + H_BASE + (blk0 & 1) * H_BYTES + (sl * NWAVE + wave) * 1024, h_voff[sl], blk0 * (2 * BK));
    }
    issue_w(rs_w, 0, blk0 * (2 * BK));
    issue_w(rs_w, 1, blk0 * (2 * BK) + cin2);
    if constexpr (GN != 0) {
      // group statistics of this workgroup's image while the first tiles fly: per channel the producers' slots in slot order (sums meet in the image buffer that is
      // still free), per group the channels in channel order -- gn_fold.h, the same fold as gn_apply_stats_kernel
      double2* chs = (double2*)(smem + H_BASE + ((blk0 + 1) & 1) * H_BYTES);
      gn_channel_sums(p.gn, cin_main, p.Ho * p.Wo, h_img, tid, NWAVE * 64, chs);
      if (tid < 128) *(float*)(smem + HaloSmem<BN, GN>::GN_ZERO + tid * 4) = 0.f;
      gn_load_gb(blk0);
      __syncthreads();
      gn_group_stats(p.gn, p.Ho * p.Wo, tid, chs, gn_gstat);
      __syncthreads();
      gn_build_tbl(blk0);
      gn_load_gb(blk0 + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the first image have landed (and gamma / beta of the next block)
      __syncthreads();
      static_assert(H_SLOTS == 6, "six image pieces per wave");
      gn_norm_piece(std::integral_constant<int, 0>{}, blk0 & 1); gn_norm_piece(std::integral_constant<int, 1>{}, blk0 & 1); gn_norm_piece(std::integral_constant<int, 2>{}, blk0 & 1);
      gn_norm_piece(std::integral_constant<int, 3>{}, blk0 & 1); gn_norm_piece(std::integral_constant<int, 4>{}, blk0 & 1); gn_norm_piece(std::integral_constant<int, 5>{}, blk0 & 1);
      __syncthreads();                                       // the first image is normalised for every wave; the table may be rebuilt
    }
    if (grp == 0) {
      auto body = [&](auto tap_tag, int blk) {
        constexpr int TAP = decltype(tap_tag)::value;
        top(std::integral_constant<int, LPS0 + (TAP >= 1 && TAP <= H_SLOTS ? 1 : 0)>{});      // in flight: what the interval before issued (weights; + an image piece after taps 0..5)
        if constexpr (GN != 0 && TAP == 0) gn_build_tbl(blk + 1);      // table of the NEXT block (its gamma / beta were loaded at tap 8 of the block before: landed by this wait); written before mid() retires it
        if constexpr (GN != 0 && TAP == 8) gn_load_gb(blk + 2);        // (behind the counted wait, ahead of this interval's DMA pieces)
        rd_tap(tap_tag);
        if constexpr (GN != 0 && TAP >= 2 && TAP <= 7) gn_fetch(std::integral_constant<int, (TAP >= 2 && TAP <= 7) ? TAP - 2 : 0>{}, (blk + 1) & 1);      // (this wave's own piece TAP - 2: landed by this tap's counted wait)
        __builtin_amdgcn_sched_barrier(0);
        issue_tap(tap_tag, blk);
        mid();
        if constexpr (GN != 0 && TAP >= 2 && TAP <= 7) mm_gn(std::integral_constant<int, (TAP >= 2 && TAP <= 7) ? TAP - 2 : 0>{}, (blk + 1) & 1);      // piece TAP - 2 of the next image, beside this tap's MFMAs
        else mm();
      };
      for (int blk = blk0; blk < blk1; ++blk) {
        a_rd = a_rd0 + (blk & 1) * H_BYTES;
        body(std::integral_constant<int, 0>{}, blk); body(std::integral_constant<int, 1>{}, blk); body(std::integral_constant<int, 2>{}, blk);
        body(std::integral_constant<int, 3>{}, blk); body(std::integral_constant<int, 4>{}, blk); body(std::integral_constant<int, 5>{}, blk);
        body(std::integral_constant<int, 6>{}, blk); body(std::integral_constant<int, 7>{}, blk); body(std::integral_constant<int, 8>{}, blk);
        t += 9;
      }
    } else {
      bool first = true;
      auto body = [&](auto tap_tag, int blk) {
        constexpr int TAP = decltype(tap_tag)::value;
        top(std::integral_constant<int, LPS1 + (TAP >= 1 && TAP <= H_SLOTS ? 1 : 0)>{});
        if constexpr (GN != 0 && TAP >= 3 && TAP <= 8) mm_gn(std::integral_constant<int, (TAP >= 3 && TAP <= 8) ? TAP - 3 : 0>{}, (blk + 1) & 1);      // the piece fetched in the read half-step before, beside the MFMAs of the tap before
        else if (TAP != 0 || !first) mm();
        mid();
        rd_tap(tap_tag);
        if constexpr (GN != 0 && TAP >= 2 && TAP <= 7) gn_fetch(std::integral_constant<int, (TAP >= 2 && TAP <= 7) ? TAP - 2 : 0>{}, (blk + 1) & 1);      // (own piece TAP - 2: issued in tap TAP - 2's read half-step, landed by this tap's counted wait)
        __builtin_amdgcn_sched_barrier(0);
        issue_tap(tap_tag, blk);
      };
      for (int blk = blk0; blk < blk1; ++blk) {
        a_rd = a_rd0 + (blk & 1) * H_BYTES;
        body(std::integral_constant<int, 0>{}, blk); first = false;
        body(std::integral_constant<int, 1>{}, blk); body(std::integral_constant<int, 2>{}, blk);
        body(std::integral_constant<int, 3>{}, blk); body(std::integral_constant<int, 4>{}, blk); body(std::integral_constant<int, 5>{}, blk);
        body(std::integral_constant<int, 6>{}, blk); body(std::integral_constant<int, 7>{}, blk); body(std::integral_constant<int, 8>{}, blk);
        t += 9;
      }
      mm();
    }
  }
  if (n2 > 0) {
    // ---- the appended 1x1 blocks: one k-tile per block of 64 channels, nothing to share between tiles -- the plain ping-pong ring, its activation slots
    //      (256 rows x 128 B, XOR-swizzled, rows = the patch's pixels) in the place of the two halo images. The pipeline is drained once in between.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    int a2v[A_PW], a3v[A_PW];
#pragma unroll
    for (int i = 0; i < A_PW; ++i) {
      const int r = a_piece(i) * RPP + srow, m = row_m(r), gch = cpos ^ lds_swz<BK>(r);
      a2v[i] = (m * ld2 + gch * 8) * 2;
      a3v[i] = (m * ld3 + gch * 8) * 2;
    }
    const __amdgpu_buffer_rsrc_t rs2 = mk_rsrc(src2, (size_t)hM * ld2 * 2), rs3 = mk_rsrc(src3, src3 ? (size_t)hM * ld3 * 2 : 0);
    const int e0 = k0 > nk_main ? k0 - nk_main : 0;                      // first appended tile of this workgroup
    const int staged = blk1 > blk0 ? 2 : 0;                              // tiles whose weights the 3x3 part has already put into the ring
    auto stage2 = [&](int j, int slot, bool with_w) {                   // tile j of this part: weights (unless staged), then the activation rows
      const int ch = (e0 + j) * BK;
      if (with_w) issue_w(rs_w, slot, 9 * cin2 + (e0 + j) * (2 * BK));
      char* dst = smem + H_BASE + slot * (BM * ROWB);
      if (ch < cin_extra) {
#pragma unroll
        for (int i = 0; i < A_PW; ++i) BLDS16(rs2, dst + a_piece(i) * 1024, a2v[i], ch * 2);
      } else {
#pragma unroll
        for (int i = 0; i < A_PW; ++i) BLDS16(rs3, dst + a_piece(i) * 1024, a3v[i], (ch - cin_extra) * 2);
      }
    };
    auto rd2 = [&](int slot) {
      const char* ab = smem + H_BASE + slot * (BM * ROWB) + a_off;
      const char* bb = smem + slot * STAGE + w_off;
#pragma unroll
      for (int kk = 0; kk < KSUB; ++kk) {
        const int coff = ((kk * 4 + fq) ^ fswz) << 4;
#pragma unroll
        for (int i = 0; i < MR; ++i) af[kk][i] = *(const h8*)(ab + i * 16 * ROWB + coff);
#pragma unroll
        for (int j = 0; j < NR; ++j) wf[kk][j] = *(const h8*)(bb + j * 16 * ROWB + coff);
      }
    };
    stage2(0, 0, staged < 1);
    if (n2 > 1) stage2(1, 1, staged < 2);
    const int allow0 = n2 > 1 ? A_PW + (staged < 2 ? b_npw : 0) : 0;      // pieces of tile 1 that may still fly when tile 0 is read
    int slot_r = 0, slot_s = NSTAGE - 1;
    auto adv = [&]() { slot_r = slot_r + 1 == NSTAGE ? 0 : slot_r + 1; slot_s = slot_s + 1 == NSTAGE ? 0 : slot_s + 1; };
    auto top2 = [&](int j, auto lps_tag) {
      __builtin_amdgcn_sched_barrier(0);
      if (j == 0) wait_ring<A_PW + B_PW, 1>(allow0);
      else if (j + 1 < n2) wait_vm_barrier<decltype(lps_tag)::value>();
      else wait_vm_barrier<0>();
      __builtin_amdgcn_sched_barrier(0);
    };
    if (grp == 0) {
      for (int j = 0; j < n2; ++j) {
        top2(j, std::integral_constant<int, A_PW + LPS0>{});
        rd2(slot_r);
        __builtin_amdgcn_sched_barrier(0);
        if (j + NSTAGE - 1 < n2) stage2(j + NSTAGE - 1, slot_s, true);
        mid();
        mm();
        adv();
      }
    } else {
      for (int j = 0; j < n2; ++j) {
        top2(j, std::integral_constant<int, A_PW + LPS1>{});
        if (j > 0) mm();
        mid();
        rd2(slot_r);
        __builtin_amdgcn_sched_barrier(0);
        if (j + NSTAGE - 1 < n2) stage2(j + NSTAGE - 1, slot_s, true);
        adv();
      }
      mm();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the loads against the empty descriptor write zeros into the ring: they have to be in before the epilogue takes the LDS)

  // ---- epilogue (gemm_epilogue.h)
  TileCtx tc;
  tc.hM = hM; tc.hN = hN; tc.hK = hK; tc.tm = tm; tc.tn = tn; tc.bm0 = bm0; tc.bn0 = bn0; tc.tiles_m = tiles_m; tc.tiles_n = tiles_n; tc.split = split; tc.nsplit = nsplit;
  tc.h_m0 = h_m0; tc.ln_s1 = 0.f; tc.ln_s2 = 0.f;
  IA2P_STAMP(
    if (tid == 0 && p.partial && nsplit == 1) {      // (the stamps go to a buffer nothing else reads)
      unsigned long long* o = (unsigned long long*)p.partial + 8 * blockIdx.x;
      const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
      o[0] = __builtin_amdgcn_s_memtime() - stamp_c0; o[1] = r1 - stamp_r0; o[2] = stamp_entry; o[3] = stamp_r0; o[4] = r1;
    }
  )
  AttnKvRegs kvr;
  tile_epilogue<BM, BN, NSTAGE, WGM, BK, PP, WGN, 0, 1>(acc, tc, p, nullptr, kvr);
}

template <int BN, int GN = 0>
// (leading arguments: 14 preloaded dwords, gemm_kernel.h -- operands, shape, strides, {K split, tile order} and the image geometry the first DMA piece is addressed with:
//  {Ho, Wo} and {Hs, Ws} in 16 bits each, {upsampled view, Cin}, Cin2)
__global__ __launch_bounds__(512, 2) void conv_halo_f16_kernel(const half_t* hA, const half_t* hW, int hM, int hN, int hK, int hlda, int hldw, int hsk_gw, int hHoWo, int hHsWs, int hUpCin,
                                                                 int hCin2, const GemmArgs p) {
  conv_halo_tile_body<BN, GN>(hA, hW, hM, hN, hK, hlda, hldw, IA2P_SKGW_LO8(hsk_gw), IA2P_SKGW_GW(hsk_gw), IA2P_SKGW_FLAGS(hsk_gw) & 1, hHoWo & 0xffff, (int)((unsigned)hHoWo >> 16), hHsWs & 0xffff,
                              (int)((unsigned)hHsWs >> 16), hUpCin & 3, (int)((unsigned)hUpCin >> 2), hCin2, p);
}

template <int BN, int GN>
static hipError_t launch_halo_gn(const GemmArgs& a, hipStream_t s) {
  constexpr int smem = HaloSmem<BN, GN>::SMEM;
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_halo_f16_kernel<BN, GN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int tiles = (a.M / 256) * ((a.N + BN - 1) / BN);
  GemmArgs b = a;
  ia2p_gemm_prepare(b, smem, 256, BN, true);
  if (b.gn_out && !b.vec8) return hipErrorInvalidValue;      // (the column sums of the output are taken on the 16-byte epilogue routes)
  if (b.sk_counters && tiles > ia2p_sk_counter_capacity()) return hipErrorInvalidValue;
  if ((long)tiles * (a.splitk > 1 ? a.splitk : 1) >= (1L << 21)) return hipErrorInvalidValue;      // (udiv_small in the tile decode: block and tile counts below 2^21)
  int sk_gw, howo, hsws;
  if (!ia2p_pack_skgw(b.splitk, b.group_w, b.m_fastest, false, &sk_gw) || !ia2p_pack_rowmap(b.Ho, b.Wo, &howo) || !ia2p_pack_rowmap(b.Hs, b.Ws, &hsws) || b.up < 0 || b.up > 3 || b.Cin < 0 || b.Cin >= (1 << 29))
    return hipErrorInvalidValue;
  hipLaunchKernelGGL((conv_halo_f16_kernel<BN, GN>), dim3(tiles * (a.splitk > 1 ? a.splitk : 1)), dim3(512), smem, s,
                     b.A, b.W, b.M, b.N, b.K, b.lda, b.ldw, sk_gw, howo, hsws, (int)((unsigned)b.up | ((unsigned)b.Cin << 2)), b.Cin2, b);
  return hipGetLastError();
}
// a.gn.st0 != nullptr: the GroupNorm-fused form (the operand is the raw input of the norm; ia2p_conv_gn_ok)
template <int BN>
static hipError_t launch_halo(const GemmArgs& a, hipStream_t s) {
  if (!ia2p_conv_halo_ok(a) || !ia2p_conv_gn_ok(a)) return hipErrorInvalidValue;
  return a.gn.st0 ? launch_halo_gn<BN, 1>(a, s) : launch_halo_gn<BN, 0>(a, s);
}
