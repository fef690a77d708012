// The GEMM / implicit-GEMM kernel template and its launcher (see gemm.hip for the structure). A header so that the fused
// to_q + cross-attention tile (qxattn.hip) can instantiate it with the attention core as its epilogue, in a translation unit of its own
// (the attention code wants -amdgpu-mfma-vgpr-form, the plain GEMM tiles do not).
#pragma once
#include "common.h"
#include "attention_core.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

bool ia2p_splitk_inkernel(int M, int N, int splitk);
int ia2p_sk_counter_capacity();
int* ia2p_sk_counters(hipStream_t s, int tiles);
void ia2p_sk_counters_invalidate();      // new epoch: every stream's ticket buffer is re-zeroed in front of its next K-split launch
const float* ia2p_phi_lut();

#ifndef IA2P_LIN_BUF
#define IA2P_LIN_BUF 1      // linear layers stage their operands with BUFFER loads to LDS (descriptor + one 32-bit offset register per piece + a scalar k offset) instead of
#endif                      // per-piece 64-bit running pointers; 0: the pointer form (A/B builds)
#define BLDS16(rsrc, ldsptr, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(ldsptr), 16, voff, soff, 0, 0)
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
// weight-tile staging pieces (1 KiB = RPP rows) per wave. Even split where the pieces divide by the waves (surplus rows would read the zero page);
// the ping-pong tile may split UNEVENLY -- its first wave group takes one piece more per wave than its second -- so that no LDS goes to padding rows
// (256 x 160: 20 pieces = 4 x 3 + 4 x 2; three stages of (256 + 160) rows are 156 KiB, with padding to 192 rows they would not fit the CU's 160 KiB)
template <int BN, int BK, int NWAVE, int PP>
struct BStage {
  static constexpr int RPP = 1024 / (2 * BK), P = BN / RPP, HI = (P + NWAVE - 1) / NWAVE;
  static constexpr bool UNEVEN = PP != 0 && P % NWAVE != 0 && P == (NWAVE / 2) * (2 * HI - 1);
  static constexpr int BNL = UNEVEN ? BN : HI * NWAVE * RPP;     // weight rows held in LDS
};

template <int BM, int BN, int NSTAGE, int WGM, int BK, int WGN = 2, int PP = 0>
struct EpiCfg {
  static constexpr int BNL = BStage<BN, BK, WGM * WGN, PP>::BNL;   // weight rows staged (>= BN)
  static constexpr int STAGE_BYTES = NSTAGE * (BM + BNL) * 2 * BK;
  static constexpr int PITCH = ((BN / 4 + 7) & ~7) * 4;        // floats per fp32 tile row: whole groups of 8 chunks (the XOR swizzle stays inside a group)
  static constexpr bool POW2 = ((BN / 8) & (BN / 8 - 1)) == 0;
  static constexpr int LUT_BYTES = BN % 32 == 0 ? ((IA2P_PHI_LUT_N * 8 + 15) & ~15) : 0;          // GEGLU-capable widths: the normal-CDF table of the gate activation
  static constexpr int extra_nolut(int cr) { return (2 * BM + 2 * BN + 4) * 4 + (POW2 ? 0 : cr * (BN / 8) * 8); }
  static constexpr int extra(int cr) { return extra_nolut(cr) + LUT_BYTES; }
  static constexpr int LIMIT = (PP == 2 || (BM == 256 && BN == 192)) ? 160 * 1024 : STAGE_BYTES > 80 * 1024 ? STAGE_BYTES : 80 * 1024;   // (the 8-phase tile and the fused QKV + self-attention tile own their CU: the whole LDS)
  static constexpr int NCHUNK = (PP != 2 && BM * PITCH * 4 + extra(BM) <= LIMIT) ? 1 : 2;     // (8-phase tile: always one chunk per row half, the way its waves hold the rows)
  static_assert(WGM % NCHUNK == 0, "a chunk holds whole wave rows");
  static constexpr int CR = BM / NCHUNK;
  static constexpr int TILE_BYTES = CR * PITCH * 4;
  static_assert(TILE_BYTES + extra(CR) <= LIMIT, "epilogue staging does not fit");
  static constexpr int SMEM_F32 = STAGE_BYTES > TILE_BYTES + extra(CR) ? STAGE_BYTES : TILE_BYTES + extra(CR);
  // register epilogue (launches without a K split, 16-byte-aligned outputs): the accumulators get bias / folded LayerNorm / activation in the MFMA layout, are
  // rounded to fp16 and cross the LDS ONCE as a [BM][BN] fp16 tile (rows padded by 16 B: the 8-byte fragment writes of 16 rows land on 16 different bank groups)
  static constexpr int P16 = BN * 2 + 16, T16_BYTES = BM * P16;
  static constexpr int EXTRA16 = (2 * BM + 2 * BN + 4) * 4 + (POW2 ? 0 : BM * (BN / 8) * 8) + LUT_BYTES;
#ifndef IA2P_REG_EPI_MIN
#define IA2P_REG_EPI_MIN 0        // tiles of fewer elements keep the fp32 route (build-time knob for A/B builds)
#endif
  static constexpr bool REG_EPI = T16_BYTES + EXTRA16 <= LIMIT && BM * BN >= IA2P_REG_EPI_MIN;      // (else the fp32 chunked route only: 160 x 160)
  static constexpr int SMEM = REG_EPI && T16_BYTES + EXTRA16 > SMEM_F32 ? T16_BYTES + EXTRA16 : SMEM_F32;
};

// PP = 1 ("ping-pong", 8 waves = WGM 4, 3-stage ring, ONE workgroup per CU): waves 0-3 own the upper half of the tile rows, waves 4-7 the
// lower half, and the two groups run half a k-step apart -- while one group reads its fragments from LDS the other issues its MFMAs, with a
// workgroup barrier between the half-steps. Eight waves behind one barrier per k-step would all read, then all multiply (the LDS and the
// MFMA phases add up); two independent workgroups per CU de-phase by themselves but need twice the LDS fill per flop
// (profiles/r01g_gemm_loop_ablation.txt: the fill is the largest term of the 128x128 kernel).
// XA != 0 (qxattn.hip; 128 x 64 tiles only): the tile is the to_q projection of 128 queries x ONE head and never leaves the CU -- the epilogue turns it
// into the Q fragments of the attention core (attention_core.h, MODE = XA - 1) and writes the cross-attention output instead.
// HALO = 1 (conv_halo_f16_kernel; 3x3, stride 1, ping-pong schedule, 256-row tiles): the tile is a 16 x 16 pixel PATCH of one image, and the activation operand is
// not staged k-tile by k-tile (nine taps = nine fetches of nearly the same pixels through the fabric) but once per block of 64 channels, as the patch plus its
// one-pixel border: 18 x 18 pixels x 128 B, 144-B pixel pitch (8 data chunks + 1 pad chunk: fragment reads of 16 consecutive pixels hit 16 different 16-byte bank
// groups), two such images (the next block's lands while this one is multiplied). A filter tap is then a CONSTANT byte offset on the fragment reads -- the k-loop has no
// gather arithmetic at all -- and the L2 -> LDS traffic of a k-tile drops from (256 + BN) x 128 B to (36 + BN) x 128 B. K is walked block-major (nine taps of a block,
// then the next block; appended 1x1 blocks: the centre tap of their own image) over the SAME packed weights: the weight tile of (block, tap) starts at column
// tap * Cin + block * 64 of the [Co][tap][Cin] row.
template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int BK = 64, int PP = 0, int WGN = 2, int XA = 0, int HALO = 0>
__device__ __forceinline__ void gemm_tile_body(const half_t* hA, const half_t* hW, const half_t* hzero, int hM, int hN, int hK, int hlda, int hldw, int hrpb, int hbstride,
                                               int hroff, int hsplitk, int hgroup_w, const GemmArgs& p, const AttnArgs* xa) {
  // The leading 16 dwords of the argument list are what the prologue needs; built with -amdgpu-kernarg-preload-count=16 the command processor
  // hands them over in SGPRs, so the first tile loads go out without waiting for a cold read of the argument block (which costs every launch
  // ~1 us: tools/micro/launch_floor2.hip). The rest of GemmArgs (epilogue, conv geometry) arrives while those loads fly.   // >= 2 waves/SIMD: big tiles must fit 256 registers
  static_assert(PP != 1 || (WGM * WGN == 8 && (WGM == 4 || WGM == 8) && NSTAGE == 3), "ping-pong schedule: 8 waves (two groups of 4 by tile rows), 3-stage ring");
  static_assert(PP != 3 || (WGM == 4 && NSTAGE == 2 && !CONV), "two-slot ping-pong schedule: 8 waves, 2 k-tile slots");
  static_assert(PP != 2 || (WGM == 2 && WGN == 4 && NSTAGE == 2 && BK == 64 && BM == 256 && (BN == 256 || BN == 128) && XA == 0), "8-phase schedule: 256-row tiles, 2 x 4 waves, two k-tile buffers");
  static_assert(HALO == 0 || (CONV && PP == 1 && BM == 256 && BK == 64 && XA == 0), "halo-staged convolution: ping-pong schedule over a 16 x 16 patch");
  constexpr int NWAVE = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;    // wave tile (waves arranged WGM x WGN; WGN = 1: narrow tiles, one wave per 128-byte column block)
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int ROWB = 2 * BK, CPR = ROWB / 16, RPP = 1024 / ROWB;   // row bytes, chunks per row, rows per 1-KiB staging piece
  using BS = BStage<BN, BK, NWAVE, PP>;
  constexpr int A_PW = BM / RPP / NWAVE, B_PW = BS::HI;             // staging pieces per wave (B rounded up: the surplus rows read the zero page -- or,
  constexpr bool B_UNEVEN = BS::UNEVEN;                              //  ping-pong tile, the second wave group takes one piece less per wave: BStage)
  constexpr int BNL = BS::BNL;                                       // weight rows held in LDS (>= BN)
  static_assert(BM % (RPP * NWAVE) == 0 && BN % 16 == 0 && WN % 16 == 0 && WM % 16 == 0, "tile / wave layout");
  constexpr int STAGE = ((HALO ? 0 : BM) + BNL) * ROWB;                // bytes of a ring slot (halo-staged convolution: the weight tile only)
  constexpr int H_PITCH = 144, H_ROW = 18 * H_PITCH;                   // halo image: bytes per pixel (8 chunks + 1 pad), per row of 18 pixels
  constexpr int H_SLOTS = ((18 * 18 * 9 + 63) / 64 + NWAVE - 1) / NWAVE, H_PIECES = H_SLOTS * NWAVE, H_BYTES = H_PIECES * 1024;      // 1-KiB DMA pieces of an image per wave (6), per image (48: the last two are padding, so that every wave issues the same count)
  constexpr int H_BASE = NSTAGE * STAGE;                               // the two halo images sit behind the weight ring
  static_assert(!HALO || H_BASE + 2 * H_BYTES <= EpiCfg<BM, BN, NSTAGE, WGM, BK, WGN, PP>::SMEM, "halo images + weight ring exceed the tile's LDS");
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
#ifdef IA2P_CLOCK_STAMP
  const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();
#endif

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
  // tile row r -> output row (pixel index). Halo-staged convolution: the tile is the 16 x 16 patch (h_y0, h_x0) of image h_img, row r = pixel (r >> 4, r & 15) of it
  int h_img = 0, h_y0 = 0, h_x0 = 0, h_m0 = 0;
  if constexpr (HALO != 0) {
    const int tpr = p.Wo >> 4, tpi = (p.Ho >> 4) * tpr;
    h_img = tm / tpi;
    const int rem = tm - h_img * tpi, ty = rem / tpr;
    h_y0 = ty * 16; h_x0 = (rem - ty * tpr) * 16;
    h_m0 = (h_img * p.Ho + h_y0) * p.Wo + h_x0;
  }
  auto row_m = [&](int r) { return HALO ? h_m0 + (r >> 4) * p.Wo + (r & 15) : bm0 + r; };

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
  // halo-staged convolution: DMA piece (slot * NWAVE + wave) of a halo image, lane -> 16-byte chunk j = piece * 64 + lane = (halo pixel j / 9, chunk j % 9);
  // h_voff = byte offset of that chunk inside the source tensor (channel block 0), or an offset past any tensor: pad chunk / outside the image / past the image's
  // last pixel -- the buffer load's range check then writes zeros (no zero page, no select)
  int h_voff[H_SLOTS];
  if constexpr (HALO != 0) {
#pragma unroll
    for (int sl = 0; sl < H_SLOTS; ++sl) {
      const int j = (sl * NWAVE + wave) * 64 + lane;
      const int hp = j / 9, c = j - hp * 9;
      const int hy = hp / 18, hx = hp - hy * 18;
      const int y = h_y0 - 1 + hy, x = h_x0 - 1 + hx;
      const bool ok = c < 8 && hp < 18 * 18 && (unsigned)y < (unsigned)p.Ho && (unsigned)x < (unsigned)p.Wo;
      // (nearest-x2 upsampled view: the image in LDS IS the upsampled patch -- pixel (y, x) of it comes from source pixel (y / 2, x / 2), fetched up to four times out of L2)
      h_voff[sl] = ok ? (((h_img * p.Hs + (y >> p.up)) * p.Ws + (x >> p.up)) * hlda + c * 8) * 2 : OOB;
    }
  }
#pragma unroll
  for (int i = 0; i < (HALO ? 0 : A_PW); ++i) {
    const int pi = a_piece(i);
    const int m = bm0 + pi * RPP + srow;
    const int gch = cpos ^ lds_swz<BK>(pi * RPP + srow);
    if (!CONV) {
      if (m < hM) {
        int src = m;
        if (hrpb) { const int b = m / hrpb; src = b * hbstride + (m - b * hrpb) + hroff; }
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
    if constexpr (HALO != 0 || LINBUF) w_voff[i] = w_inc[i] ? (n * hldw + gch * 8) * 2 : OOB;      // (buffer loads: byte offset of the piece's chunk in the weight matrix, column 0)
  }

  const int nk_all = hK / BK;
  const int kt0 = (int)((long)split * nk_all / nsplit), kt1 = (int)((long)(split + 1) * nk_all / nsplit);   // this workgroup's k-tiles
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
    if constexpr (HALO == 0) {
#pragma unroll
      for (int i = 0; i < B_PW; ++i) w_ptr[i] += (size_t)kt0 * w_inc[i];
    }
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
  const int a_off = (wm0 + frow) * ROWB, w_off = (HALO ? 0 : BM * ROWB) + (wn0 + frow) * ROWB;

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
  const int nk = kt1 - kt0;
  constexpr int LPS = A_PW + B_PW;   // LDS-DMA pieces this wave issues per k-tile
  // NSTAGE-deep LDS ring: tiles kt+1 .. kt+NSTAGE-2 stay in flight across the barrier of step kt (counted vmcnt,
  // raw s_barrier -- cdna_hip_programming.md §5 "Pipelining across barriers"); ONE barrier per k-step.
  if constexpr (PP == 2) {      // 8-phase tile: k-tile 0 whole, k-tile 1 up to its third half (the fourth is issued in the first phase of the loop)
    stage_part(0, std::integral_constant<int, 0>{}); stage_part(0, std::integral_constant<int, 1>{});
    stage_part(0, std::integral_constant<int, 2>{}); stage_part(0, std::integral_constant<int, 3>{});
    if (nk > 1) { stage_part(1, std::integral_constant<int, 0>{}); stage_part(1, std::integral_constant<int, 1>{}); stage_part(1, std::integral_constant<int, 2>{}); }
  } else if constexpr (HALO == 0) {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
      if (s < nk) stage(s, s);
  }
  load_ln();               // behind the prologue DMA: the statistics' round trip overlaps the first tiles' (the ping-pong tile used to load them AHEAD of
  if (PP) asm volatile("" : "+v"(ln_s1), "+v"(ln_s2));      //  its prologue -- a serial 1-2 us at every workgroup start; folded here, before the loop, they still cost it no registers)
  // fused cross-attention: the context K / V of this tile's (batch element, head) travel to registers while the projection runs
  AttnKvRegs kvr;       // loaded inside the k-loop, behind the first tile     // folds now; the counted wait leaves the prologue DMA in flight
#ifdef IA2P_CLOCK_STAMP     // diagnostic build only (tools/micro/gemm_clock.hip): in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz around the k-loop
  const unsigned long long stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  if constexpr (PP == 1 && HALO != 0) {
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
    const __amdgpu_buffer_rsrc_t rs_w = mk_rsrc(hW, (size_t)hN * hldw * 2), rs_a = mk_rsrc(hA, (size_t)(hM / (p.Ho * p.Wo)) * p.Hs * p.Ws * hlda * 2), rs_none = mk_rsrc(hW, 0);
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
      if constexpr (TAP < H_SLOTS)
        BLDS16(blk + 1 < blk1 ? rs_a : rs_none, smem + H_BASE + ((blk + 1) & 1) * H_BYTES + (TAP * NWAVE + wave) * 1024, h_voff[TAP], (blk + 1) * (2 * BK));
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
    if (blk1 > blk0) {
      // prologue: the first block's image, the weights of its first two tiles
#pragma unroll
      for (int sl = 0; sl < H_SLOTS; ++sl) BLDS16(rs_a, smem + H_BASE + (blk0 & 1) * H_BYTES + (sl * NWAVE + wave) * 1024, h_voff[sl], blk0 * (2 * BK));
      issue_w(rs_w, 0, blk0 * (2 * BK));
      issue_w(rs_w, 1, blk0 * (2 * BK) + cin2);
      if (grp == 0) {
        auto body = [&](auto tap_tag, int blk) {
          constexpr int TAP = decltype(tap_tag)::value;
          top(std::integral_constant<int, LPS0 + (TAP >= 1 && TAP <= H_SLOTS ? 1 : 0)>{});      // in flight: what the interval before issued (weights; + an image piece after taps 0..5)
          rd_tap(tap_tag);
          __builtin_amdgcn_sched_barrier(0);
          issue_tap(tap_tag, blk);
          mid();
          mm();
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
          if (TAP != 0 || !first) mm();
          mid();
          rd_tap(tap_tag);
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
  } else if constexpr (PP == 1) {
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
#ifdef IA2P_CLOCK_STAMP
  if (tid == 0 && p.partial && nsplit == 1) {      // (the stamps go to a buffer nothing else reads)
    unsigned long long* o = (unsigned long long*)p.partial + 8 * blockIdx.x;
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    o[0] = __builtin_amdgcn_s_memtime() - stamp_c0; o[1] = r1 - stamp_r0; o[2] = stamp_entry; o[3] = stamp_r0; o[4] = r1;
  }
#endif
  // ---- epilogue, staged through LDS. The MFMA layout gives a lane 4 consecutive columns of ONE row (acc[i][j][r] = C[bm0+wm0+16i+(lane&15)]
  //      [bn0+wn0+16j+4(lane>>4)+r]): stored from there, a wave-instruction touches 16 rows x 32 B -- quarter cache lines, and so does every
  //      residual read (profiles/r01g_gemm_loop_ablation.txt: 15 us of a 34 us launch at K -> 0). Instead the fp32 tile goes through the (now
  //      free) stage buffers once: written in the MFMA layout (16-B chunks XOR-swizzled by row & 7: conflict-free ds_write_b128), read back
  //      row-major, 8 columns per thread, so that bias / time-embedding row / folded-LayerNorm constants / residual are 16-B loads and C is
  //      written in whole 128-B lines; everything is still applied to the fp32 accumulator and rounded once.
  using EC = EpiCfg<BM, BN, NSTAGE, WGM, BK, WGN, PP>;
  constexpr int PITCH = EC::PITCH;
  constexpr int NT = NWAVE * 64, CR = EC::CR;                           // threads, tile rows per chunk
  float* tile = (float*)smem;
  // two routes: the register epilogue (no K split, 16-byte accesses everywhere -- every shape of the executors; below) and the chunked fp32 route (K splits: the slabs are
  // fp32; odd strides). Both compute  h = fp16(acc * as + bias * bs | folded LayerNorm, activation)  and  out = fp16(h + rowvec * bs + residual): the rounding of the
  // reference's own fp16 modules (a Linear / Conv2d output is an fp16 tensor before the time-embedding row or the residual is added to it).
  const bool reg_epi = XA == 0 && EC::REG_EPI && nsplit == 1 && p.vec8 != 0 && (hN & 7) == 0 && !p.act;      // (activations other than GEGLU: the CLIP / prior MLPs, on the fp32 route)
  char* cbase = smem + (reg_epi ? EC::T16_BYTES : EC::TILE_BYTES);
  float* ln_rows = (float*)cbase;                                       // [0, BM): mean, [BM, 2 BM): rstd
  float* ln_cs = ln_rows + 2 * BM;                                      // BN column sums and BN folded biases of this tile
  float* ln_lb = ln_cs + BN;
  int* sk_flag = (int*)(ln_lb + BN);                                    // K-split: the ticket this workgroup drew, broadcast to its waves
  float2* part = (float2*)(ln_lb + BN + 4);                             // row-statistics partials (tile widths whose 8-column groups per row are not a power of two)
  const float2* phi = (const float2*)(cbase + (reg_epi ? EC::EXTRA16 - EC::LUT_BYTES : EC::extra_nolut(CR)));      // GEGLU: normal-CDF table (gelu_lut_f), copied in below
  __syncthreads();                    // every wave has finished reading the stage buffers
  if (p.ln_stats) {
    if (tid < BM) {
      const float2 mr = ln_mean_rstd_f(ln_s1, ln_s2, hK, p.ln_eps);
      ln_rows[tid] = mr.x;
      ln_rows[BM + tid] = mr.y;
    }
    if constexpr (XA == 4) {
      if (tid < BN / 4) {           // columns 4 tid .. of the tile = the same offsets inside the Q / K / V block of head tn
        const int n = ((tid * 4) >> 6) * (hN / 3) + tn * 64 + ((tid * 4) & 63);
        *(f4*)(ln_cs + tid * 4) = *(const f4*)(p.ln_cs + n);
        *(f4*)(ln_lb + tid * 4) = *(const f4*)(p.ln_bias + n);
      }
    } else if (tid < BN / 4 && bn0 + tid * 4 < hN) {
      *(f4*)(ln_cs + tid * 4) = *(const f4*)(p.ln_cs + bn0 + tid * 4);
      *(f4*)(ln_lb + tid * 4) = *(const f4*)(p.ln_bias + bn0 + tid * 4);
    }
  }
  // Ping-pong tile: no separate prefetch workgroups (a workgroup holds a whole CU's LDS, so they would queue up behind the tiles): every
  // tile workgroup touches its slice of the next contraction's weights. The loads are issued HERE -- behind the epilogue's own constant loads,
  // whose wait would otherwise (vmcnt retires in order) also wait for these HBM-cold lines -- and nothing consumes them before the kernel's
  // end, so they fly during the whole epilogue (round 2 XOR-ed them together right away: a 2-3 us stall of every workgroup ahead of its epilogue).
  unsigned pfacc = 0, pfv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if ((PP == 1 || PP == 2) && p.pf) {
    const long nwg = (long)tiles_m * tiles_n * nsplit;
    const long per = ((p.pf_bytes + nwg - 1) / nwg + 255) & ~255L;
    const long lo = (long)blockIdx.x * per, hi = min(lo + per, p.pf_bytes & ~15L);
    const char* src = (const char*)p.pf;
    constexpr long SW = NWAVE * 64 * 16;
    if (lo < hi)
      for (long o = lo + tid * 16; o < hi; o += 8 * SW) {      // 8 independent loads in flight per thread (clamped, never branched around)
#pragma unroll
        for (int u = 0; u < 8; ++u) pfacc ^= pfv[u];            // the PREVIOUS round's lines (zeros the first time): nothing waits for the loads just issued
#pragma unroll
        for (int u = 0; u < 8; ++u) pfv[u] = *(const unsigned*)(src + min(o + u * SW, hi - 16));
#ifdef IA2P_PP_EAGER_PF      // A/B builds: consume at once, as round 2 did
#pragma unroll
        for (int u = 0; u < 8; ++u) asm volatile("" : "+v"(pfv[u]));
#endif
      }
  }
  auto pf_sink = [&]() {              // keep the loads alive up to here
    if (PP == 1 || PP == 2) asm volatile("" ::"v"(pfacc), "v"(pfv[0]), "v"(pfv[1]), "v"(pfv[2]), "v"(pfv[3]), "v"(pfv[4]), "v"(pfv[5]), "v"(pfv[6]), "v"(pfv[7]));
  };
  const float e_as = p.acc_scale == 0.f ? 1.f : p.acc_scale, e_bs = p.bias_scale == 0.f ? 1.f : p.bias_scale;
  const bool fast = p.vec8 != 0 && (hN & 7) == 0;     // 16-byte accesses everywhere (every shape of the executors); else 8-byte pieces
  // C stores: plain, or write-through (`sc1`) when the launcher asks for it
  const __amdgpu_buffer_rsrc_t c_rsrc = wt_rsrc((void*)p.C, (size_t)hM * p.ldc * 2);
  auto store_c8 = [&](size_t elem, h8 o) {
    if (p.c_wt) store16_wt(c_rsrc, elem * 2, o);
    else *(h8*)(p.C + elem) = o;
  };
  auto tl = [&](int r, int c) -> f4 { return *(const f4*)(tile + (size_t)r * PITCH + ((c ^ (r & 7)) << 2)); };
  auto acc_to_tile = [&](int ch) {
    if constexpr (PP == 2) {        // 8-phase tile: chunk ch = half-tile ch of the rows; every wave holds MR / 2 fragment rows of it
      static_assert(PP != 2 || (EC::NCHUNK == 2 && CR == BM / 2), "8-phase tile: one epilogue chunk per row half");
      // (two explicit arms with compile-time fragment indices: written as `if (i / MRH == ch)` inside one unrolled loop, the compiler re-rolled it into
      //  acc[ch * MRH + i] -- a runtime index, i.e. the accumulators in scratch)
      auto half = [&](auto h_tag) {
        constexpr int H = decltype(h_tag)::value;
#pragma unroll
        for (int i = 0; i < MRH; ++i) {
          const int r = wm0 + i * 16 + frow;
#pragma unroll
          for (int j = 0; j < NR; ++j) {
            const int c = (wn0 + frag_col(j)) / 4 + fq;
            *(f4*)(tile + (size_t)r * PITCH + ((c ^ (r & 7)) << 2)) = acc[H * MRH + i][j];
          }
        }
      };
      if (ch == 0) half(std::integral_constant<int, 0>{});
      else half(std::integral_constant<int, 1>{});
    } else if (wm0 / CR == ch) {
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
  if constexpr (XA == 4) {
    // ---- fused QKV projection + self-attention (reference attention_processor.py:239 `attn.to_q`, :246-247 `to_k` / `to_v`, :259 scaled_dot_product_attention of
    //      AttnProcessor2_0; the LayerNorm in front of them folded into the projection): this tile is Q | K | V of ALL 256 tokens of one image x ONE head --
    //      everything that head's attention needs. The accumulators get the projection epilogue (fp32, ONE rounding to fp16, exactly what the stand-alone GEMM
    //      stores), go straight into the LDS images of the attention core (K and V images as stage_kv lays them out, Q as swizzled rows), and only O is written:
    //      Q, K and V (15.7 MB per layer at batch 8) never travel to memory and back, one launch instead of two.
    static_assert(XA != 4 || (BM == 256 && BN == 192 && WGM == 4 && WGN == 2 && !CONV && (PP == 0 || PP == 3)), "fused QKV + self-attention: 256 x 192 tiles, 8 waves");
    char* sK = smem;
    char* sV = smem + 32768;
    char* sQ = smem + 65536;              // [256][128 B], 16-byte chunks XOR-swizzled by row & 7; later: the waves' O staging tiles (4 KiB each, a wave's own query rows)
    __syncthreads();                      // row / column constants are in LDS (and every wave is through with the stage buffers: barrier above)
#pragma unroll
    for (int i = 0; i < MR; ++i) {
      const int r = wm0 + i * 16 + frow;
      const float mu = p.ln_stats ? ln_rows[r] : 0.f, rs = p.ln_stats ? ln_rows[BM + r] : 1.f;
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const int cl = wn0 + j * 16 + fq * 4;                     // tile column of acc[i][j][0]: 4 consecutive columns inside ONE of Q / K / V
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (p.ln_stats) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ln_fold_f(v[e], mu, rs, ln_cs[cl + e], ln_lb[cl + e]);
        } else if (p.bias) {
          const h4 hb = *(const h4*)(p.bias + (cl >> 6) * (hN / 3) + tn * 64 + (cl & 63));
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)hb[e];
        }
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
        const int which = cl >> 6, d = cl & 63, chunk = d >> 3, sub = (d & 7) * 2;
        const int sw = which == 0 ? (r & 7) : which == 1 ? ((r >> 1) & 7) : (((r >> 1) & 1) << 2);
        char* img = which == 0 ? sQ : which == 1 ? sK : sV;
        *(h4*)(img + r * 128 + ((chunk ^ sw) << 4) + sub) = o;
      }
    }
    __syncthreads();
    const int r31 = lane & 31, hh = lane >> 5;
    h8 qf[4];
    {
      const int r = wave * 32 + r31;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[s] = *(const h8*)(sQ + r * 128 + (((2 * s + hh) ^ (r & 7)) << 4));
    }
    const AttnArgs& ap = *xa;
    f16v otot[2];
    attn_core<0, true>(ap, tm, tn, qf, sK, sV, tid, otot);       // this wave's 32 queries over the image's 256 keys (no workgroup barrier inside: resident images)
    attn_store_o(otot, ap.O, (size_t)ap.B * ap.Nq * ap.ldo, tm, wave * 32, tn, ap.Nq, ap.ldo, sQ + wave * 4096, lane, (ap.xcd_map & 2) != 0);
    return;
  } else if constexpr (XA != 0) {
    // ---- fused to_q + cross-attention (reference attention_processor.py:344 `attn.to_q`, :371 / :387 the two SDPA calls, :397 `text + scale * ip`):
    //      this tile is Q of 128 queries x one head. It goes through the fp32 LDS tile once (projection epilogue applied there, rounded to fp16
    //      exactly as the stand-alone GEMM would store it), comes back as the Q^T fragments of the attention core, and only O is written.
    static_assert(BM == 128 && BN == 64 && WGM == 2 && WGN == 2 && !CONV && !PP && EC::NCHUNK == 1, "fused cross-attention: 128 x 64 tiles, 4 waves");
#ifdef IA2P_CLOCK_STAMP
    unsigned long long* xo = p.partial ? (unsigned long long*)p.partial + 8 * blockIdx.x : nullptr;      // (diagnostic build, tools/micro/qx_clock.hip: the fused launch has no slabs, the field carries the stamp buffer)
#endif
    acc_to_tile(0);
    __syncthreads();
    const int r31 = lane & 31, hh = lane >> 5;
    h8 qf[4];
    {
      const int r = wave * 32 + r31;
      const float mu = p.ln_stats ? ln_rows[r] : 0.f, rs = p.ln_stats ? ln_rows[BM + r] : 1.f;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int cl = 16 * s + 8 * hh;
        const f4 x0 = tl(r, cl >> 2), x1 = tl(r, (cl >> 2) + 1);
        float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
        if (p.ln_stats) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ln_fold_f(v[e], mu, rs, ln_cs[cl + e], ln_lb[cl + e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] *= e_as;
          if (p.bias) {
            const h8 hb = *(const h8*)(p.bias + bn0 + cl);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf((float)hb[e], e_bs, v[e]);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (half_t)v[e];
      }
    }
    __syncthreads();                  // tile and row constants consumed: the LDS now belongs to K / V
    const AttnArgs& ap = *xa;
    const int b = bm0 / ap.Nq, q0 = bm0 - b * ap.Nq + wave * 32, hd = tn;
    f16v otot[2];
    attn_kv_store(ap, tid, kvr, smem, smem + 32768);     // only short contexts are fused (the launcher checks): their K / V are in registers by now
    __syncthreads();
#ifdef IA2P_CLOCK_STAMP
    if (tid == 0 && xo) xo[6] = __builtin_amdgcn_s_memrealtime();      // Q fragments built, K / V images in LDS
#endif
    attn_core<XA - 1, true>(ap, b, hd, qf, smem, smem + 32768, tid, otot);
    __syncthreads();                  // every wave is through with the K / V images
#ifdef IA2P_CLOCK_STAMP
    if (tid == 0 && xo) xo[7] = __builtin_amdgcn_s_memrealtime();      // attention core done
#endif
    attn_store_o(otot, ap.O, (size_t)ap.B * ap.Nq * ap.ldo, b, q0, hd, ap.Nq, ap.ldo, smem + wave * 4096, lane, (ap.xcd_map & 2) != 0);
#ifdef IA2P_CLOCK_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && xo) xo[5] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }
  if constexpr (XA == 0 && EC::REG_EPI) {
  if (reg_epi) {
    // ---- register epilogue. In the MFMA layout a lane holds 4 consecutive columns of MR x NR (row, column-quad) positions: the row constants of the folded
    //      LayerNorm are MR values per lane, the column constants (or the bias) one 16-byte read per fragment column -- no index arithmetic, no per-group constant
    //      reads. The fp16 tile then crosses the LDS once, half the bytes of the fp32 route and in ONE piece (no chunking, two barriers), and is read out
    //      row-major, 16 bytes = 8 outputs per thread: plain launches copy it to C, the others add the time-embedding row / residual (fp32, one more rounding) or
    //      multiply values by GELU(gates).
    constexpr int P16 = EC::P16;
    char* t16 = smem;
    if (p.geglu && EC::LUT_BYTES) {
      for (int i = tid; i < IA2P_PHI_LUT_N; i += NT) ((float2*)phi)[i] = ((const float2*)p.phi_lut)[i];
    }
    __syncthreads();                  // row / column constants are in LDS (and every wave is through with the stage buffers: barrier above)
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      const int cl = wn0 + frag_col(j) + fq * 4;                      // tile column of acc[.][j][0]
      f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};       // folded LayerNorm: column sums, folded biases; else: bias * bs, -
      if (p.ln_stats) { c0 = *(const f4*)(ln_cs + cl); c1 = *(const f4*)(ln_lb + cl); }
      else if (p.bias) {
        const h4 hb = *(const h4*)(p.bias + min(bn0 + cl, hN - 4));
#pragma unroll
        for (int e = 0; e < 4; ++e) c0[e] = (float)hb[e];
      }
#pragma unroll
      for (int i = 0; i < MR; ++i) {
        const int r = wm0 + frag_row(i) + frow;
        float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        if (p.ln_stats) {
          const float mu = ln_rows[r], rs = ln_rows[BM + r];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = ln_fold_f(v[e], mu, rs, c0[e], c1[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= e_as;
          if (p.bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(c0[e], e_bs, v[e]);
          }
        }
        h4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
        *(h4*)(t16 + r * P16 + cl * 2) = o;
      }
    }
    __syncthreads();
#ifdef IA2P_CLOCK_STAMP
    if (tid == 0 && p.partial && nsplit == 1) ((unsigned long long*)p.partial)[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();      // the fp16 tile is in LDS
#endif
    auto rowm = [&](int r) { return row_m(r); };
    if (p.geglu) {                    // packed columns: 32-wide blocks [16 values | 16 gates]; out[m][n/2] = a * gelu(g)
      constexpr int GPR = BN / 16, TOTAL = BM * GPR, U = 2, ITER = (TOTAL + NT * U - 1) / (NT * U);      // groups of 8 OUTPUT columns per row
#pragma unroll 1
      for (int k = 0; k < ITER; ++k) {
        h8 ha[U], hg[U];
        int rr[U], gg[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = min(tid + (k * U + u) * NT, TOTAL - 1);
          const int r = idx / GPR, g = idx - r * GPR;
          rr[u] = r; gg[u] = g;
          const char* q = t16 + r * P16 + ((g >> 1) * 32 + (g & 1) * 8) * 2;      // 8 values; their gates 16 columns on
          ha[u] = *(const h8*)q; hg[u] = *(const h8*)(q + 32);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int r = rr[u], g = gg[u];
          h8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
#ifdef IA2P_GEGLU_ERF
            o[e] = (half_t)((float)ha[u][e] * gelu_erf_f((float)hg[u][e]));
#else
            o[e] = (half_t)((float)ha[u][e] * gelu_lut_f((float)hg[u][e], phi));
#endif
          }
          const bool live = tid + (k * U + u) * NT < TOTAL && rowm(r) < hM && bn0 + (g >> 1) * 32 < hN;
          if (live) store_c8((size_t)(rowm(r)) * p.ldc + (bn0 >> 1) + (g >> 1) * 16 + (g & 1) * 8, o);
        }
      }
    } else {
      constexpr int GPR = BN / 8, TOTAL = BM * GPR, U = 4, ITER = (TOTAL + NT * U - 1) / (NT * U);
      constexpr bool POW2 = (GPR & (GPR - 1)) == 0;
      static_assert(!POW2 || NT % GPR == 0, "row groups must not straddle waves");
#pragma unroll 1
      for (int k = 0; k < ITER; ++k) {
        h8 hh[U], hv[U], hr[U];
        int rr[U], gg[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {     // all loads of U groups in flight before any arithmetic (clamped addresses, never branched around)
          const int idx = tid + (k * U + u) * NT;
          const int r = min(idx / GPR, BM - 1), g = idx - (idx / GPR) * GPR;
          rr[u] = r; gg[u] = g;
          const int m = rowm(r), n = bn0 + g * 8;
          live[u] = idx < TOTAL && m < hM && n < hN;
          const int mc = min(m, hM - 1), nc = min(n, hN - 8);
          hh[u] = *(const h8*)(t16 + r * P16 + g * 16);
          if (p.rowvec) hv[u] = *(const h8*)(p.rowvec + (size_t)(mc / p.rows_per_batch) * p.rowvec_ld + nc);
          if (p.residual) hr[u] = *(const h8*)(p.residual + (size_t)mc * p.ldr + nc);
        }
        float st1[U], st2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          h8 o = hh[u];
          if (p.rowvec || p.residual) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (float)hh[u][e];
            if (p.rowvec) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaf((float)hv[u][e], e_bs, v[e]);
            }
            if (p.residual) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += (float)hr[u][e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
          }
          if (live[u]) store_c8((size_t)(rowm(rr[u])) * p.ldc + bn0 + gg[u] * 8, o);
          st1[u] = st2[u] = 0.f;
          if (p.stats_out && live[u]) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = (float)o[e]; st1[u] += f; st2[u] += f * f; }
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
              if (gg[u] == 0 && idx < TOTAL && rowm(rr[u]) < hM) ((float2*)p.stats_out)[(size_t)tn * hM + rowm(rr[u])] = make_float2(a, b);
            } else if (idx < TOTAL) part[idx] = make_float2(st1[u], st2[u]);
          }
        }
      }
      if constexpr (!POW2) {
        if (p.stats_out) {
          __syncthreads();
          if (tid < BM && rowm(tid) < hM) {
            float s1 = 0.f, s2 = 0.f;
            for (int g = 0; g < GPR; ++g) { const float2 v = part[tid * GPR + g]; s1 += v.x; s2 += v.y; }      // group order: deterministic
            ((float2*)p.stats_out)[(size_t)tn * hM + rowm(tid)] = make_float2(s1, s2);
          }
        }
      }
    }
    pf_sink();
#ifdef IA2P_CLOCK_STAMP
    if (tid == 0 && p.partial && nsplit == 1) ((unsigned long long*)p.partial)[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && p.partial && nsplit == 1) ((unsigned long long*)p.partial)[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }
  }
  bool from_slabs = false;
  if (nsplit > 1) {
    // ---- K-split: this workgroup holds the partial sums of ONE K range. Every K-slice writes its raw fp32 slab (write-through `sc1` stores:
    //      the bytes are in memory-side coherence when the wave's vmcnt drains, no release fence -- cdna_hip_programming.md §5 "In-launch split-K
    //      reduction"); the slice that arrives LAST at the tile's ticket counter adds the slabs up in slab order (deterministic whoever is last)
    //      and runs the epilogue: no reduce launch, no spin (nobody waits for anybody).
    constexpr int GPR = BN / 4;
    const __amdgpu_buffer_rsrc_t slab = __builtin_amdgcn_make_buffer_rsrc((void*)(p.partial + (size_t)split * hM * hN), 0, (int)min((size_t)hM * hN * 4, (size_t)0x7ffffff0), 0x00020000);
#pragma unroll(PP == 2 ? 2 : 1)      // (8-phase tile: both chunks spelled out, so that the accumulator fragments of a chunk are compile-time register names)
    for (int ch = 0; ch < EC::NCHUNK; ++ch) {
      if (ch) __syncthreads();
      acc_to_tile(ch);
      __syncthreads();
      auto rowm = [&](int r) { return row_m(ch * CR + r); };      // tile row of this chunk -> output row
      for (int idx = tid; idx < CR * GPR; idx += NT) {
        const int r = idx / GPR, g = idx - r * GPR;
        const int m = rowm(r), n = bn0 + g * 4;
        if (m < hM && n < hN) {
          const f4 v = tl(r, g);
          typedef unsigned u4v __attribute__((__vector_size__(4 * sizeof(unsigned))));
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4v, v), slab, (int)(((size_t)m * hN + n) * 4), 0, 16);      // aux 16 = sc1 (write-through)
        }
      }
    }
    if (!p.sk_counters) { pf_sink(); return; }      // finished by a separate splitk_reduce_kernel launch (A/B switch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // EVERY storing wave drains its write-through stores ...
    __syncthreads();                                       // ... before ONE lane signals for the workgroup
    if (tid == 0) *sk_flag = __hip_atomic_fetch_add(p.sk_counters + (tm * tiles_n + tn), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*sk_flag != nsplit - 1) { pf_sink(); return; }
    if (tid == 0) {
      __hip_atomic_store(p.sk_counters + (tm * tiles_n + tn), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (launches are stream-ordered)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's stale lines before the plain loads below
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    from_slabs = true;
  }
#pragma unroll(PP == 2 ? 2 : 1)
  for (int ch = 0; ch < EC::NCHUNK; ++ch) {
#if defined(IA2P_CLOCK_STAMP) && defined(IA2P_STAMP_AT)
    if (tid == 0 && p.partial && nsplit == 1 && IA2P_STAMP_AT == 2 && ch == 1) ((unsigned long long*)p.partial)[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();      // chunk 0's stores issued
#endif
    if (ch || from_slabs) __syncthreads();          // the previous chunk has been read out
    auto rowm = [&](int r) { return row_m(ch * CR + r); };      // tile row of this chunk -> output row
    acc_to_tile(ch);
    if (from_slabs) {
      // tile chunk = sum of the K-slice slabs in slab order (slab 0 first), whoever arrived last. This slice's own partial sums are still in its
      // accumulators (now in the LDS tile: the very fp32 values its slab holds), so only the OTHER slabs are read back -- with every load of a
      // thread in flight at once: one workgroup alone reads at the latency of its round trips, not at a bandwidth (a dependent loop over the slabs
      // took ~16 serial trips per thread and cost more than the whole-chip reduce launch it replaced: profiles/r02e_splitk_inkernel_ab.txt).
      __syncthreads();
      constexpr int GPR = BN / 4;
      constexpr int PER = (CR * GPR + NT - 1) / NT;                     // float4 positions per thread
      const size_t slab_elems = (size_t)hM * hN;
      auto combine = [&](auto no_tag) {
        constexpr int NO = decltype(no_tag)::value;                     // slabs of OTHER slices (nsplit - 1)
        constexpr int BUDGET = EC::NCHUNK == 1 ? 32 : 8;                // 16-B loads in flight per lane (two chunks: the second chunk's accumulators are live -- more spills)
        constexpr int UC = PER < BUDGET / NO ? PER : BUDGET / NO;
#pragma unroll 1
        for (int i0 = 0; i0 < PER; i0 += UC) {
          f4 oth[NO][UC];
          int off[UC];
#pragma unroll
          for (int u = 0; u < UC; ++u) {
            const int idx = min(tid + (i0 + u) * NT, CR * GPR - 1);
            const int r = idx / GPR, g = idx - r * GPR;
            off[u] = r * PITCH + ((g ^ (r & 7)) << 2);
            const float* src = p.partial + (size_t)min(rowm(r), hM - 1) * hN + min(bn0 + g * 4, hN - 4);
#pragma unroll
            for (int j = 0; j < NO; ++j) oth[j][u] = *(const f4*)(src + (size_t)(j < split ? j : j + 1) * slab_elems);     // (clamped addresses, never branched around)
          }
#pragma unroll
          for (int u = 0; u < UC; ++u) {
            if (i0 + u < PER && tid + (i0 + u) * NT < CR * GPR) {
              const f4 own = *(const f4*)(tile + off[u]);
              f4 v = split == 0 ? own : oth[0][u];
#pragma unroll
              for (int k = 1; k <= NO; ++k) {                            // slab k of the ordered sum: own partial, or the (k - (k > split))-th other slab
                f4 w = own;
                if (k != split) w = k > split ? oth[k - 1][u] : oth[k < NO ? k : NO - 1][u];
                v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
              }
              *(f4*)(tile + off[u]) = v;
            }
          }
        }
      };
      if (nsplit == 2) combine(std::integral_constant<int, 1>{});
      else if (nsplit == 3) combine(std::integral_constant<int, 2>{});
      else if (nsplit == 4) combine(std::integral_constant<int, 3>{});
      else {                          // wider splits (rare): plain ordered loop
        for (int idx = tid; idx < CR * GPR; idx += NT) {
          const int r = idx / GPR, g = idx - r * GPR;
          const int o = r * PITCH + ((g ^ (r & 7)) << 2);
          const float* src = p.partial + (size_t)min(rowm(r), hM - 1) * hN + min(bn0 + g * 4, hN - 4);
          const f4 own = *(const f4*)(tile + o);
          f4 v = split == 0 ? own : *(const f4*)src;
          for (int sl = 1; sl < nsplit; ++sl) {
            f4 w = own;
            if (sl != split) w = *(const f4*)(src + (size_t)sl * slab_elems);
            v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
          }
          *(f4*)(tile + o) = v;
        }
      }
    }
    if (p.geglu && ch == 0 && EC::LUT_BYTES) {
      for (int i = tid; i < IA2P_PHI_LUT_N; i += NT) ((float2*)phi)[i] = ((const float2*)p.phi_lut)[i];
    }
    __syncthreads();
#if defined(IA2P_CLOCK_STAMP) && defined(IA2P_STAMP_AT)      // (diagnostic build: where the epilogue's time goes; slot 7 of the stamp buffer)
    if (tid == 0 && p.partial && nsplit == 1 && ((IA2P_STAMP_AT == 1 && ch == 0) || (IA2P_STAMP_AT == 3 && ch == 1))) ((unsigned long long*)p.partial)[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();
#endif
    if (p.geglu) {                    // packed columns: 32-wide blocks [16 values | 16 gates]; out[m][n/2] = a * gelu(g)
      constexpr int GPR = BN / 16;    // groups of 8 OUTPUT columns per row
#ifndef IA2P_GEGLU_U
#define IA2P_GEGLU_U 1        // groups in flight per thread in the two-chunk tiles (build-time knob for A/B builds)
#endif
      constexpr int TOTAL = CR * GPR, U = EC::NCHUNK == 1 ? 2 : IA2P_GEGLU_U, ITER = (TOTAL + NT * U - 1) / (NT * U);
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
          live[u] = idx < TOTAL && rowm(r) < hM && bn0 + ca * 4 < hN;
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
              va[e] = ln_fold_f(va[e], mu, rs, ln_cs[cl + e], ln_lb[cl + e]);
              vg[e] = ln_fold_f(vg[e], mu, rs, ln_cs[cl + 16 + e], ln_lb[cl + 16 + e]);
            }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { va[e] += (float)ba[u][e]; vg[e] += (float)bg[u][e]; }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) { va[e] = (float)(half_t)va[e]; vg[e] = (float)(half_t)vg[e]; }      // (the projection's output is an fp16 tensor: same rounding as the register epilogue)
          // (packed fp32 -- v_pk_fma_f32 on element pairs, the same operations -- was built and measured: +0.1 ms per step, same box, A/B builds;
          //  the compiler's own mix of scalar and packed instructions is the faster one. Round 3)
          h8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
#ifdef IA2P_GEGLU_ERF        // A/B builds: the arithmetic form (Abramowitz & Stegun 7.1.26)
            o[e] = (half_t)(va[e] * gelu_erf_f(vg[e]));
#else
            o[e] = (half_t)(va[e] * gelu_lut_f(vg[e], phi));
#endif
          }
          if (live[u]) store_c8((size_t)(rowm(r)) * p.ldc + (bn0 >> 1) + (g >> 1) * 16 + (g & 1) * 8, o);
        }
      }
    } else {
      constexpr int GPR = BN / 8;     // groups of 8 columns per row
      constexpr bool POW2 = (GPR & (GPR - 1)) == 0;
      #ifndef IA2P_EPI_U2
#define IA2P_EPI_U2 1         // groups in flight per thread in the two-chunk tiles (build-time knob for A/B builds)
#endif
      constexpr int TOTAL = CR * GPR, U = EC::NCHUNK == 1 ? 4 : IA2P_EPI_U2, ITER = (TOTAL + NT * U - 1) / (NT * U);   // (two chunks: the second chunk's accumulators are still live)
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
            const int m = rowm(r), n = bn0 + g * 8;
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
              for (int e = 0; e < 8; ++e) v[e] = ln_fold_f(v[e], mu, rs, ln_cs[cl + e], ln_lb[cl + e]);
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
            if (p.rowvec || p.residual) {      // (the layer's own output is an fp16 tensor before the time-embedding row / the residual is added: same rounding as the register epilogue)
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (float)(half_t)v[e];
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
            if (live[u]) store_c8((size_t)(rowm(r)) * p.ldc + bn0 + cl, o);
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
            const int m = rowm(r);
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
                for (int e = 0; e < 4; ++e) v[e] = ln_fold_f(v[e], mu, rs, ln_cs[cl + e], ln_lb[cl + e]);
              } else {
                v[0] *= e_as; v[1] *= e_as; v[2] *= e_as; v[3] *= e_as;
                if (p.bias) { const h4 b = *(const h4*)(p.bias + n); v[0] = fmaf((float)b[0], e_bs, v[0]); v[1] = fmaf((float)b[1], e_bs, v[1]); v[2] = fmaf((float)b[2], e_bs, v[2]); v[3] = fmaf((float)b[3], e_bs, v[3]); }
              }
              if (p.act) { v[0] = act_f(v[0], p.act); v[1] = act_f(v[1], p.act); v[2] = act_f(v[2], p.act); v[3] = act_f(v[3], p.act); }
              if (p.rowvec || p.residual) { v[0] = (float)(half_t)v[0]; v[1] = (float)(half_t)v[1]; v[2] = (float)(half_t)v[2]; v[3] = (float)(half_t)v[3]; }
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
              if (gg[u] == 0 && idx < TOTAL && rowm(rr[u]) < hM) ((float2*)p.stats_out)[(size_t)tn * hM + rowm(rr[u])] = make_float2(a, b);
            } else if (idx < TOTAL) part[idx] = make_float2(st1[u], st2[u]);
          }
        }
      }
      if constexpr (!POW2) {
        if (p.stats_out) {
          __syncthreads();
          if (tid < CR && rowm(tid) < hM) {
            float s1 = 0.f, s2 = 0.f;
            for (int g = 0; g < GPR; ++g) { const float2 v = part[tid * GPR + g]; s1 += v.x; s2 += v.y; }      // group order: deterministic
            ((float2*)p.stats_out)[(size_t)tn * hM + rowm(tid)] = make_float2(s1, s2);
          }
        }
      }
    }
  }
  pf_sink();
#ifdef IA2P_CLOCK_STAMP
  if (tid == 0 && p.partial && nsplit == 1) ((unsigned long long*)p.partial)[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();      // this wave has ISSUED its last C store
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the C stores of this wave have left
  __syncthreads();
  if (tid == 0 && p.partial && nsplit == 1) ((unsigned long long*)p.partial)[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int BM, int BN, int NSTAGE, bool CONV, int WGM = 2, int BK = 64, int PP = 0, int WGN = 2>
__global__ __launch_bounds__(WGM * WGN * 64, 2) void gemm_f16_kernel(const half_t* hA, const half_t* hW, const half_t* hzero, int hM, int hN, int hK, int hlda, int hldw, int hrpb, int hbstride,
                                                                         int hroff, int hsplitk, int hgroup_w, const GemmArgs p) {
  gemm_tile_body<BM, BN, NSTAGE, CONV, WGM, BK, PP, WGN, 0>(hA, hW, hzero, hM, hN, hK, hlda, hldw, hrpb, hbstride, hroff, hsplitk, hgroup_w, p, nullptr);
}

// the halo-staged 3x3 convolution (gemm_tile_body, HALO = 1): 256 x BN tiles of 16 x 16 pixel patches, ping-pong schedule
// wave arrangement: 4 x 2 waves of 64 x (BN / 2); the 80-wide tile 8 x 1 waves of 32 x 80 (N = 640 / 1280 problems at M = 8192 / 2048 in 256 / 128 tiles of one per CU:
// the 32^2 maps fill the chip without a K split, the 16^2 maps with two slices instead of three)
template <int BN> struct HaloWaves { static constexpr int WGM = BN == 80 ? 8 : 4, WGN = BN == 80 ? 1 : 2; };
template <int BN>
__global__ __launch_bounds__(512, 2) void conv_halo_f16_kernel(const half_t* hA, const half_t* hW, const half_t* hzero, int hM, int hN, int hK, int hlda, int hldw, int hrpb, int hbstride,
                                                                 int hroff, int hsplitk, int hgroup_w, const GemmArgs p) {
  gemm_tile_body<256, BN, 3, true, HaloWaves<BN>::WGM, 64, 1, HaloWaves<BN>::WGN, 0, 1>(hA, hW, hzero, hM, hN, hK, hlda, hldw, hrpb, hbstride, hroff, hsplitk, hgroup_w, p, nullptr);
}
// grouped tile order: panel width (in tiles) such that the contiguous tile range an XCD works on is a compact block; 0 = plain order.
// An XCD that holds r x c tiles fetches r activation row panels and c weight column panels: r a + c w bytes with r c fixed is least at c = sqrt(resident a / w).
// a_over_w = bytes of one activation row panel over bytes of one weight column panel: BM / BN for a linear layer (both K deep), but a 3x3 convolution's
// row panel holds only Cin channels of (BM + halo) pixels while its weight panel is 9 Cin deep -- ~ BM / (6 BN): narrow, tall blocks. Round 3 used BM / BN for
// both and the M = 2048, K = 11520 ... 23040 convolutions fetched their 29 ... 59 MB of weights into nearly every XCD (PMC traffic 5.5 x algorithmic).
static inline int ia2p_tile_group_w(int tiles, int tiles_n, int smem, double a_over_w) {
  static const int group_mode = getenv("IA2P_TILE_GROUP") ? atoi(getenv("IA2P_TILE_GROUP")) : 2;      // 0 plain order, 1 round 3's BM / BN rule for everything, 2 byte-aware
  if (!group_mode) return 0;
  const int smem_per_cu = 160 * 1024 / smem;                                   // co-resident workgroups per CU by LDS
  const double resident = std::min<double>(tiles / 8.0, 32.0 * std::max(1, std::min(smem_per_cu, 2)));   // tiles an XCD holds at once
  static const double gscale = ia2p_exp_env("IA2P_TILE_GROUP_SCALE") ? atof(ia2p_exp_env("IA2P_TILE_GROUP_SCALE")) : 1.0;
  const int w = (int)(gscale * std::sqrt(resident * a_over_w) + 0.5);
  return std::max(1, std::min(w, tiles_n));
}
static inline int ia2p_tile_group_w(int tiles, int tiles_n, int smem, int BM, int BN) { return ia2p_tile_group_w(tiles, tiles_n, smem, (double)BM / BN); }

// launcher-side fields of a launch description: epilogue access width, write-through C, grouped tile order
static inline void ia2p_gemm_prepare(GemmArgs& b, int smem, int BM, int BN, bool conv = false) {
  // 16-byte epilogue accesses need 8-element row strides and 16-byte-aligned bases; otherwise the epilogue falls back to 8-byte pieces
  auto al16 = [](const void* q) { return (((uintptr_t)q) & 15) == 0; };
  b.vec8 = (b.ldc % 8 == 0 && al16(b.C) && (!b.bias || al16(b.bias)) && (!b.residual || (b.ldr % 8 == 0 && al16(b.residual))) &&
            (!b.rowvec || (b.rowvec_ld % 8 == 0 && al16(b.rowvec)))) ? 1 : 0;
  b.c_wt = ((ia2p_wt_mask() & 1) && (size_t)b.M * b.ldc * 2 < (size_t)0x7ffffff0) ? 1 : 0;      // same box: -0.14 ms per step at batch 8
  b.phi_lut = b.geglu ? ia2p_phi_lut() : nullptr;
  const int tiles_n = (b.N + BN - 1) / BN;
  double a_over_w = (double)BM / BN;
  static const int group_mode = getenv("IA2P_TILE_GROUP") ? atoi(getenv("IA2P_TILE_GROUP")) : 2;
  if (conv && group_mode >= 2 && b.K > 0)      // distinct activation bytes of a row panel: Cin channels of the tile's pixels plus their halo (~ 1.5 x), plus the appended 1x1 blocks
    a_over_w = BM * (1.5 * b.Cin + (b.A2 ? b.Cin2 : 0) + (b.A3 ? b.Cin3 : 0)) / ((double)BN * b.K);
  b.group_w = ia2p_tile_group_w(((b.M + BM - 1) / BM) * tiles_n, tiles_n, smem, a_over_w);
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
  if (a.geglu && !b.phi_lut) return hipErrorOutOfMemory;
  if (!CONV && IA2P_LIN_BUF) {      // buffer-load staging addresses an operand with a 31-bit byte offset
    const size_t a_rows = a.rpb ? ((size_t)a.M / a.rpb + 1) * (size_t)std::max(a.bstride, 0) + a.roff + a.rpb : (size_t)a.M;
    if (!ia2p_fits_buffer(a_rows, a.lda) || !ia2p_fits_buffer(a.N, a.ldw)) return hipErrorInvalidValue;
  }
  // b.sk_counters: as the caller (launch_any, gemm.hip) attached them -- null: the K-slices only write their slabs and a splitk_reduce_kernel launch finishes
  if (b.sk_counters && tiles > ia2p_sk_counter_capacity()) return hipErrorInvalidValue;
  const int extra = (!PP && a.pf && a.pf_bytes >= 4096) ? a.pf_blocks : 0;
  hipLaunchKernelGGL((gemm_f16_kernel<BM, BN, NSTAGE, CONV, WGM, BK, PP, WGN>), dim3(tiles * (a.splitk > 1 ? a.splitk : 1) + extra), dim3(WGM * WGN * 64), smem, s,
                     b.A, b.W, b.zero, b.M, b.N, b.K, b.lda, b.ldw, b.rpb, b.bstride, b.roff, b.splitk, b.group_w, b);
  return hipGetLastError();
}

template <int BN>
static hipError_t launch_halo(const GemmArgs& a, hipStream_t s) {
  if (!ia2p_conv_halo_ok(a)) return hipErrorInvalidValue;
  constexpr int smem = EpiCfg<256, BN, 3, HaloWaves<BN>::WGM, 64, HaloWaves<BN>::WGN, 1>::SMEM;
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)conv_halo_f16_kernel<BN>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int tiles = (a.M / 256) * ((a.N + BN - 1) / BN);
  GemmArgs b = a;
  ia2p_gemm_prepare(b, smem, 256, BN, true);
  if (b.sk_counters && tiles > ia2p_sk_counter_capacity()) return hipErrorInvalidValue;
  hipLaunchKernelGGL((conv_halo_f16_kernel<BN>), dim3(tiles * (a.splitk > 1 ? a.splitk : 1)), dim3(512), smem, s,
                     b.A, b.W, b.zero, b.M, b.N, b.K, b.lda, b.ldw, b.rpb, b.bstride, b.roff, b.splitk, b.group_w, b);
  return hipGetLastError();
}
