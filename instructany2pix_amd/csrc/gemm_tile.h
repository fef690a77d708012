// What every MFMA tile kernel of the library shares (gemm_kernel.h: linear layers and the gathered 3x3 convolution; conv_halo_kernel.h: the halo-staged 3x3
// convolution; qxattn.hip: the fused projection + attention tiles): staging macros, counted waits, the LDS swizzle, the LDS budget of a tile and its epilogue
// (EpiCfg), the XCD-aware tile order, and the per-tile context the shared epilogue (gemm_epilogue.h) takes over from a k-loop.
#pragma once
#include "common.h"
#include "attention_core.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

bool ia2p_splitk_inkernel(int M, int N, int splitk);
int ia2p_gemm_variant_ran(const GemmArgs& a, bool conv, int v);      // gemm.hip: the variant whose kernel a launch of plan variant v really is
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
#define IA2P_REG_EPI_MIN 4097     // tiles of fewer elements (64 x 64 and below: the M = 256 layers of a batch-1 step, the batch-8 out-projections) keep the fp32 route: same box -0.065 ms at batch 1,
                                  // -0.04 ms at batch 8 against 0 (profiles/r05p_regepi.txt); the two routes give the same bits (build-time knob for A/B builds)
#endif
  static constexpr bool REG_EPI = T16_BYTES + EXTRA16 <= LIMIT && BM * BN >= IA2P_REG_EPI_MIN;      // (else the fp32 chunked route only: 160 x 160)
  static constexpr int SMEM = REG_EPI && T16_BYTES + EXTRA16 > SMEM_F32 ? T16_BYTES + EXTRA16 : SMEM_F32;
};

// ---- diagnostic stamps (tools/micro/gemm_clock.hip, qx_clock.hip: builds with -DIA2P_CLOCK_STAMP; the product library is built WITHOUT them). One macro pair:
//      IA2P_STAMP(statements) compiles its statements in a diagnostic build only; stamp_put writes s_memrealtime into slot `slot` of the workgroup's 8-slot record
//      (p.partial carries the stamp buffer: launches without a K split, which nothing else reads).
#ifdef IA2P_CLOCK_STAMP
#define IA2P_STAMP(...) __VA_ARGS__
#ifndef IA2P_STAMP_AT
#define IA2P_STAMP_AT 0
#endif
__device__ __forceinline__ unsigned long long* stamp_base(const GemmArgs& p, int nsplit) {      // (a launch with a K split carries its records in p.stamp: p.partial holds the slabs)
  return p.stamp ? p.stamp : (nsplit == 1 ? (unsigned long long*)p.partial : nullptr);
}
__device__ __forceinline__ void stamp_put(const GemmArgs& p, int nsplit, int slot) {
  unsigned long long* b = stamp_base(p, nsplit);
  if (threadIdx.x == 0 && b) b[8 * blockIdx.x + slot] = __builtin_amdgcn_s_memrealtime();
}
#else
#define IA2P_STAMP(...)
#endif

// ---- tile of a workgroup. Blocks b, b + 8, ... share an XCD (its L2): every XCD gets a CONTIGUOUS range of the tile order; inside it either plain row / column
//      major (m_fastest) or, group_w > 0, column panels of group_w tiles walked row-major, so that the range an XCD works on (and the workgroups co-resident on it)
//      covers a compact rows x cols block: the operand panels its L2 has to fetch shrink with the perimeter.
// a / b for the tile decode: 0 <= a < 2^21, 0 < b < 2^21 (block and tile counts). One reciprocal, one correction either way: |a * rcp(b) - a / b| < 1 there (fp32 holds
// both operands exactly, v_rcp_f32 is good to 1 ulp). The compiler's own expansion of a 32-bit signed division is ~60 instructions, and a workgroup's start is bound by
// instruction issue (one or two waves per SIMD running ~600 dependent instructions before the first DMA piece: ~2 ns apiece); the decode has three of them.
__device__ __forceinline__ int udiv_small(int a, int b) {
  int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
  const int r = a - q * b;
  q += (r >= b ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}
__device__ __forceinline__ void tile_order(int bid, int tiles_m, int tiles_n, int group_w, int m_fastest, int& tm, int& tn) {
  {
    const int nwg = tiles_m * tiles_n;
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7, i = bid >> 3;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
  }
  if (group_w > 0) {
    const int per = tiles_m * group_w;
    const int panel = udiv_small(bid, per), r = bid - panel * per;
    const int w = min(group_w, tiles_n - panel * group_w);
    tm = udiv_small(r, w); tn = panel * group_w + (r - tm * w);
  } else if (m_fastest) { tn = udiv_small(bid, tiles_m); tm = bid - tn * tiles_m; }
  else                  { tm = udiv_small(bid, tiles_n); tn = bid - tm * tiles_n; }
}

// what the epilogue needs to know about the tile whose accumulators it takes over
struct TileCtx {
  int hM, hN, hK;                       // problem
  int tm, tn, bm0, bn0, tiles_m, tiles_n;
  int split, nsplit;                    // K slice of this workgroup
  int h_m0;                             // halo-staged convolution: output row of the patch's first pixel (tile row r = pixel (r >> 4, r & 15) of the patch)
  float ln_s1, ln_s2;                   // folded LayerNorm (consumer): {sum, sum of squares} of tile row threadIdx.x, collected beside the k-loop
#ifdef IA2P_CLOCK_STAMP
  unsigned long long stamp_c0, stamp_r0, stamp_entry;
#endif
};

// ---- launcher side ------------------------------------------------------------------------------------------------------------------------------
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

// packed leading kernel arguments (gemm_kernel.h: 14 preloaded dwords): row map {rows per batch, row offset} in 16 bits each; {K split, grouped tile order} in 8 + 24 bits;
// fused LayerNorm consumers of their own kernels: {statistics slots, tile order} likewise. false: the value does not fit (the launcher returns hipErrorInvalidValue)
static inline bool ia2p_pack_rowmap(int rpb, int roff, int* out) {
  if (rpb < 0 || rpb > 0xffff || roff < 0 || roff > 0xffff) return false;
  *out = (int)((unsigned)rpb | ((unsigned)roff << 16));
  return true;
}
static inline bool ia2p_pack_skgw(int lo8, int group_w, int m_fastest, bool has_ln, int* out) {      // [7:0] K split (or slots), [8] m_fastest, [9] folded-LayerNorm consumer, [31:10] tile order
  if (lo8 < 0 || lo8 > 0xff || group_w < 0 || group_w > 0x3fffff) return false;
  *out = (int)((unsigned)lo8 | (m_fastest ? 0x100u : 0u) | (has_ln ? 0x200u : 0u) | ((unsigned)group_w << 10));
  return true;
}
#define IA2P_SKGW_LO8(w) ((w) & 0xff)
#define IA2P_SKGW_FLAGS(w) (((w) >> 8) & 3)      // bit 0: m_fastest, bit 1: the launch consumes folded-LayerNorm statistics
#define IA2P_SKGW_GW(w) ((int)((unsigned)(w) >> 10))

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

