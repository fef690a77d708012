// Attention core shared by attention_f16_kernel (attention.hip) and the fused to_q + cross-attention tile (qxattn.hip):
// K/V staging, the key-tile loop with its softmax(es) and the P.V accumulation for ONE wave's 32 queries of one (batch, head).
// See attention.hip for the structure and the reference call sites.
#pragma once
#include "common.h"

#define MASKED (-1.0e30f)
#ifndef IA2P_ATTN_DMA
#define IA2P_ATTN_DMA 1     // long single-segment contexts: K / V staged by double-buffered LDS-DMA (build-time knob for A/B builds)
#endif
#ifndef IA2P_ATTN_PK
#define IA2P_ATTN_PK 1      // packed fp32 softmax arithmetic (build-time knob for A/B builds)
#endif

__device__ __forceinline__ fp16x4 lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4f16((fp16x4 __attribute__((address_space(3)))*)p);
}

// Stage NT key tiles (64 keys each) of K and V into their swizzled LDS images. All 4*NT loads of a thread are
// issued before the first LDS write (clamped row index: never a per-lane branch around a load -- hipcc would
// wait vmcnt(0) per element); padding rows are zeroed by select.
template <int NT>
__device__ __forceinline__ void stage_kv(const half_t* Kb, const half_t* Vb, int ld, int s0, int nsk, int tid, char* sK, char* sV) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 kreg[2 * NT], vreg[2 * NT];
#pragma unroll
  for (int u = 0; u < 2 * NT; ++u) {
    const int c = tid + 256 * u, row = c >> 3, pos = c & 7;
    const size_t off = (size_t)(s0 + min(row, nsk - 1)) * ld + pos * 8;
    kreg[u] = *(const u4*)(Kb + off);
    vreg[u] = *(const u4*)(Vb + off);
  }
#pragma unroll
  for (int u = 0; u < 2 * NT; ++u) {
    const int c = tid + 256 * u, row = c >> 3, pos = c & 7;
    const unsigned keep = row < nsk ? 0xFFFFFFFFu : 0u;     // bit mask, not a pointer select (that went through scratch)
    *(u4*)(sK + row * 128 + ((pos ^ ((row >> 1) & 7)) << 4)) = kreg[u] & keep;
    *(u4*)(sV + row * 128 + ((pos ^ (((row >> 1) & 1) << 2)) << 4)) = vreg[u] & keep;
  }
}

// FOLD (round 6): the image-token keys of an IP-Adapter call ride in the FREE slots of the last text-key tile -- 77 text + 4 image tokens are 64 + (13 + 4) keys = two
// tiles, the second a single live half -- instead of a tile of their own (reference attention_processor.py:371, :387, :397: two scaled_dot_product_attention calls over
// the same queries, `text + scale * ip`; the two softmaxes stay two: a key-index mask selects which running (max, sum) pair a score feeds). Taken when the last text tile
// is partial and has room: (n0 % 64) + n1 <= 64, n1 <= 32 (one staging chunk).
__device__ __host__ __forceinline__ bool attn_fold_ok(int n0, int n1) { return (n0 & 63) != 0 && n1 > 0 && n1 <= 32 && (n0 & 63) + n1 <= 64; }
// launcher-side switch (attention.hip): IA2P_ATTN_FOLD=0 / ia2p_debug_set_attn_fold(0) keep the image-token keys in a tile of their own (A/B runs, the test's reference);
// it travels in AttnArgs.xcd_map bit 2 (4 = do not fold)
bool ia2p_attn_fold_enabled();

// Both key segments of a short-context cross-attention (e.g. 77 text + 4 image-token keys = 2 + 1 tiles) staged in ONE
// load phase: tiles [0, nt0) hold segment 0, tiles [nt0, NT) segment 1 -- or, `fold`, segment 1's rows follow segment 0's last row inside tile nt0 - 1.
template <int NT>
__device__ __forceinline__ void stage_kv2(const half_t* K0, const half_t* V0, int ld0, int n0, const half_t* K1, const half_t* V1,
                                          int ld1, int n1, int nt0, int tid, char* sK, char* sV, bool fold = false) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 kreg[2 * NT], vreg[2 * NT];
#pragma unroll
  for (int u = 0; u < 2 * NT; ++u) {
    const bool s1 = (u >> 1) >= nt0;                       // wave-uniform
    const int c = tid + 256 * u, row = (c >> 3) - (s1 ? nt0 * 64 : 0), pos = c & 7;
    const size_t off = (size_t)min(row, (s1 ? n1 : n0) - 1) * (s1 ? ld1 : ld0) + pos * 8;
    kreg[u] = *(const u4*)((s1 ? K1 : K0) + off);
    vreg[u] = *(const u4*)((s1 ? V1 : V0) + off);
  }
#pragma unroll
  for (int u = 0; u < 2 * NT; ++u) {
    const bool s1 = (u >> 1) >= nt0;
    const int c = tid + 256 * u, pos = c & 7;
    int lrow = c >> 3;
    const int row = lrow - (s1 ? nt0 * 64 : 0);
    const unsigned keep = row < (s1 ? n1 : n0) ? 0xFFFFFFFFu : 0u;
    bool write = true;
    if (fold) {                      // (wave-uniform) image-token rows land behind the last text row; the text chunks leave those rows alone (two threads must not write one row)
      if (s1) { write = u == 2 * nt0 && row < n1; lrow = n0 + row; }
      else write = lrow < n0 || lrow >= n0 + n1;
    }
    if (write) {
      *(u4*)(sK + lrow * 128 + ((pos ^ ((lrow >> 1) & 7)) << 4)) = kreg[u] & keep;
      *(u4*)(sV + lrow * 128 + ((pos ^ (((lrow >> 1) & 1) << 2)) << 4)) = vreg[u] & keep;
    }
  }
}

// Two-phase form of the resident staging (the fused to_q + cross-attention tile issues the loads ahead of its GEMM loop, so the context K / V
// are in registers when the projection is through): all ATTN_PRE_TILES tile slots are loaded (rows clamped into the segment: slots past the last tile re-read
// its lines), only the live ones are written. Same LDS image as stage_kv2.
typedef unsigned int attn_u4 __attribute__((ext_vector_type(4)));
constexpr int ATTN_PRE_TILES = 3;       // 77 text + 4 image-token keys = 2 + 1 tiles
struct AttnKvRegs { attn_u4 k[2 * ATTN_PRE_TILES], v[2 * ATTN_PRE_TILES]; };
__device__ __host__ __forceinline__ bool attn_kv_resident(const AttnArgs& p) {
  return ((p.seg[0].nkeys + 63) >> 6) + (p.nseg == 2 ? (p.seg[1].nkeys + 63) >> 6 : 0) <= ATTN_PRE_TILES;
}
__device__ __forceinline__ void attn_kv_load(const AttnArgs& p, int b, int hd, int tid, AttnKvRegs& rg) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const int nt0 = (p.seg[0].nkeys + 63) >> 6;
  const bool two = p.nseg == 2;                            // one segment: the unused slots re-read segment 0
  const AttnSeg sg1 = two ? p.seg[1] : p.seg[0];
  const half_t* K0 = p.seg[0].K + (size_t)b * p.seg[0].rows_per_batch * p.seg[0].ld + hd * 64;
  const half_t* V0 = p.seg[0].V + (size_t)b * p.seg[0].rows_per_batch * p.seg[0].ld + hd * 64;
  const half_t* K1 = sg1.K + (size_t)b * sg1.rows_per_batch * sg1.ld + hd * 64;
  const half_t* V1 = sg1.V + (size_t)b * sg1.rows_per_batch * sg1.ld + hd * 64;
#pragma unroll
  for (int u = 0; u < 2 * ATTN_PRE_TILES; ++u) {
    const bool s1 = (u >> 1) >= nt0;                       // wave-uniform
    const int c = tid + 256 * u, row = (c >> 3) - (s1 ? nt0 * 64 : 0), pos = c & 7;
    const size_t off = (size_t)min(row, (s1 ? sg1.nkeys : p.seg[0].nkeys) - 1) * (s1 ? sg1.ld : p.seg[0].ld) + pos * 8;
    rg.k[u] = *(const u4*)((s1 ? K1 : K0) + off);
    rg.v[u] = *(const u4*)((s1 ? V1 : V0) + off);
  }
}
template <bool FOLD = false>
__device__ __forceinline__ void attn_kv_store(const AttnArgs& p, int tid, const AttnKvRegs& rg, char* sK, char* sV) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const int n0 = p.seg[0].nkeys, n1 = p.nseg == 2 ? p.seg[1].nkeys : 0;
  const int nt0 = (n0 + 63) >> 6, nt = nt0 + ((n1 + 63) >> 6);
  const bool fold = FOLD && !(p.xcd_map & 4) && n1 > 0 && attn_fold_ok(n0, n1);      // (wave-uniform; FOLD: the caller runs attn_core<1>, the only mode that folds)
#pragma unroll
  for (int u = 0; u < 2 * ATTN_PRE_TILES; ++u) {
    if ((u >> 1) >= nt || (fold && u > 2 * nt0)) break;     // wave-uniform
    const bool s1 = (u >> 1) >= nt0;
    const int c = tid + 256 * u, pos = c & 7;
    int lrow = c >> 3;
    const int row = lrow - (s1 ? nt0 * 64 : 0);
    const unsigned keep = row < (s1 ? n1 : n0) ? 0xFFFFFFFFu : 0u;
    bool write = true;
    if (fold) {
      if (s1) { write = row < n1; lrow = n0 + row; }
      else write = lrow < n0 || lrow >= n0 + n1;
    }
    if (write) {
      *(u4*)(sK + lrow * 128 + ((pos ^ ((lrow >> 1) & 7)) << 4)) = rg.k[u] & keep;
      *(u4*)(sV + lrow * 128 + ((pos ^ (((lrow >> 1) & 1) << 2)) << 4)) = rg.v[u] & keep;
    }
  }
}

// qf: Q^T as the B operand of S^T = K . Q^T -- lane holds Q[q0 + lane%32][16*s + 8*(lane/32) + 0..7] (fp16, already rounded).
// Result: otot[d][r] <-> channel 32d + (r&3) + 8(r>>2) + 4*(lane/32) of query q0 + lane%32. All 256 threads of the workgroup must call
// (the staging and its barriers are workgroup-wide); sK / sV: 32 KiB each.
// PRE: the caller has already put a resident context into sK / sV (attn_kv_load / attn_kv_store + barrier).
template <int MODE, bool PRE = false>      // 0: one key segment; 1: two segments, the second <= 64 keys (IP-Adapter image tokens), merged accumulator; 2: generic two segments
__device__ __forceinline__ void attn_core(const AttnArgs& p, int b, int hd, const h8 (&qf)[4], char* sK, char* sV, int tid, f16v (&otot)[2]) {
  const int lane = tid & 63;
  const int r31 = lane & 31, hh = lane >> 5;
  (void)r31;
  // Accumulators. MODE 0 / 1 keep ONE output accumulator `o` for the whole launch (MODE 1 scales the image-token probabilities so that both
  // segments share the final 1 / l_text); MODE 2 (generic second segment) sums weight_s * o_s / l_s per segment into `otot`.
  f16v o[2];
  float mrun = MASKED, lrun = 0.f;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[d][r] = 0.f; if (MODE == 2) otot[d][r] = 0.f; }

  // short two-segment contexts: everything resident after one load phase and one barrier
  const int nt0 = (p.seg[0].nkeys + 63) >> 6, nt1 = p.nseg == 2 ? (p.seg[1].nkeys + 63) >> 6 : 0;
  const bool resident = PRE || (MODE != 0 && nt0 + nt1 <= 4);
  // MODE 1, resident context: the image-token keys folded into the free slots of the last text tile (attn_fold_ok; the staging put them there)
  const int n0 = p.seg[0].nkeys, n1 = p.nseg == 2 ? p.seg[1].nkeys : 0;
  const bool fold = MODE == 1 && resident && !(p.xcd_map & 4) && attn_fold_ok(n0, n1);
  if (resident && !PRE) {
    const half_t* K0 = p.seg[0].K + (size_t)b * p.seg[0].rows_per_batch * p.seg[0].ld + hd * 64;
    const half_t* V0 = p.seg[0].V + (size_t)b * p.seg[0].rows_per_batch * p.seg[0].ld + hd * 64;
    const half_t* K1 = p.seg[1].K + (size_t)b * p.seg[1].rows_per_batch * p.seg[1].ld + hd * 64;
    const half_t* V1 = p.seg[1].V + (size_t)b * p.seg[1].rows_per_batch * p.seg[1].ld + hd * 64;
    switch (nt0 + nt1) {
      case 2: stage_kv2<2>(K0, V0, p.seg[0].ld, p.seg[0].nkeys, K1, V1, p.seg[1].ld, p.seg[1].nkeys, nt0, tid, sK, sV, fold); break;
      case 3: stage_kv2<3>(K0, V0, p.seg[0].ld, p.seg[0].nkeys, K1, V1, p.seg[1].ld, p.seg[1].nkeys, nt0, tid, sK, sV, fold); break;
      default: stage_kv2<4>(K0, V0, p.seg[0].ld, p.seg[0].nkeys, K1, V1, p.seg[1].ld, p.seg[1].nkeys, nt0, tid, sK, sV, fold); break;
    }
    __syncthreads();
  }
  float lt = 1.f;                          // MODE 1: softmax denominator of the text segment, final once segment 0 is through
  constexpr int NSEG = MODE == 0 ? 1 : 2;
#pragma unroll 1
  for (int sg = 0; sg < NSEG; ++sg) {
    if (MODE == 1 && fold && sg == 1) break;             // the image-token keys went through the last text tile
    AttnSeg seg = sg == 0 ? p.seg[0] : p.seg[1];         // (a runtime index into the by-value argument would push it to scratch)
    if (sg == 1 && p.w1_b) seg.weight = p.w1_b[b];       // per-request IP-Adapter scale
    const half_t* Kb = seg.K + (size_t)b * seg.rows_per_batch * seg.ld + hd * 64;
    const half_t* Vb = seg.V + (size_t)b * seg.rows_per_batch * seg.ld + hd * 64;
    const bool ip_tile = MODE == 1 && sg == 1;           // one tile of <= 64 keys with its own softmax, merged into `o`
    if (MODE == 2 || ip_tile) { mrun = MASKED; lrun = 0.f; }
    if (MODE == 2) {
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    }

    // Self-attention (one long segment): stages of 128 keys, double buffered, filled by LDS-DMA (16 bytes per lane, the swizzles of the two
    // images applied through the SOURCE address) -- the stage after this one flies while this one is consumed, no staging registers, one
    // barrier per stage. Rows past the end re-read the last key (finite data; their scores are masked, so their probabilities are 0).
    constexpr bool DMA = IA2P_ATTN_DMA && MODE == 0 && !PRE;
    constexpr int SSZ = DMA ? 128 : 256;
    auto stage_dma = [&](int s0n, int buf) {
      const int wv = tid >> 6, ln = tid & 63;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int pi = wv * 8 + u;                        // 32 pieces of 1 KiB: 0-15 K rows, 16-31 V rows (waves 0-1 / 2-3)
        const bool isV = pi >= 16;
        const int row = (pi & 15) * 8 + (ln >> 3), cp = ln & 7;
        const int g = cp ^ (isV ? (((row >> 1) & 1) << 2) : ((row >> 1) & 7));
        const half_t* src = (isV ? Vb : Kb) + (size_t)min(s0n + row, seg.nkeys - 1) * seg.ld + g * 8;
        char* dst = (isV ? sV : sK) + buf * 16384 + (pi & 15) * 1024;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    };
    if (DMA) stage_dma(0, 0);
    char* sKs = sK;
    char* sVs = sV;
    for (int s0 = 0; s0 < seg.nkeys; s0 += SSZ) {
      const int nsk = min(SSZ, seg.nkeys - s0);          // keys in this super-tile
      const int ntile = (nsk + 63) >> 6;
      if (DMA) {
        const int stg = s0 / SSZ;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // this stage has landed for every wave; the other buffer is free
        if (s0 + SSZ < seg.nkeys) stage_dma(s0 + SSZ, (stg + 1) & 1);
        sKs = sK + (stg & 1) * 16384; sVs = sV + (stg & 1) * 16384;
      } else if (!resident) {
        __syncthreads();  // previous super-tile fully consumed
        switch (ntile) {   // wave-uniform; each arm is fully unrolled so the staging registers never go to scratch
          case 1: stage_kv<1>(Kb, Vb, seg.ld, s0, nsk, tid, sK, sV); break;
          case 2: stage_kv<2>(Kb, Vb, seg.ld, s0, nsk, tid, sK, sV); break;
          case 3: stage_kv<3>(Kb, Vb, seg.ld, s0, nsk, tid, sK, sV); break;
          default: stage_kv<4>(Kb, Vb, seg.ld, s0, nsk, tid, sK, sV); break;
        }
        __syncthreads();
      }
      const int tbase = resident && sg == 1 ? nt0 : 0;
      // Software pipeline over the tiles of the super-tile (cdna_hip_programming.md T15): S^T of tile t+1 is ISSUED ahead of the softmax
      // arithmetic of tile t, so the matrix pipe works through it (and through the P.V of tile t-1 queued before it) while the VALU runs the
      // exponentials -- a wave's in-order stream otherwise leaves the MFMA pipe idle for the whole softmax (PMC r01i: 13.5 VALU per MFMA).
      // Two named score sets (A / B), swapped by a two-step body: a runtime-indexed set would live in scratch.
      auto qk = [&](int tl, f16v (&st)[2]) {       // S^T[key][q] for the two 32-key halves of tile tl; K fragments read ahead of the MFMAs
        const char* sKt = sKs + (tbase + tl) * 8192;
        // short contexts end in short tiles (77 text keys = 64 + 13, 4 image-token keys): a tile with <= 32 live keys is ONE half
        const int nkh = ((MODE != 0 || PRE) && (fold ? n0 + n1 : seg.nkeys) - (s0 + tl * 64) <= 32) ? 1 : 2;       // wave-uniform
        h8 kf[2][4];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          if (kh >= nkh) break;
          const int row = kh * 32 + r31;
          const char* kp = sKt + row * 128;
          const int sw = (row >> 1) & 7;
#pragma unroll
          for (int s = 0; s < 4; ++s) kf[kh][s] = *(const h8*)(kp + (((2 * s + hh) ^ sw) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          if (kh >= nkh) break;
#pragma unroll
          for (int r = 0; r < 16; ++r) st[kh][r] = 0.f;
#pragma unroll
          for (int s = 0; s < 4; ++s) st[kh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kh][s], qf[s], st[kh], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto soft_pv = [&](int tl, f16v (&st)[2]) {
        const int k0 = s0 + tl * 64;
        const int live_end = (MODE == 1 && fold) ? n0 + n1 : seg.nkeys;                            // keys of this segment's tiles that exist (folded: text keys, then the image-token keys)
        const bool merged = MODE == 1 && fold && k0 + 64 > n0;                                     // the last text tile carries the image-token keys too (wave-uniform)
        const int nkh = ((MODE != 0 || PRE) && live_end - k0 <= 32) ? 1 : 2;                     // as in qk
        const char* sVt = sVs + (tbase + tl) * 8192;
        // V^T fragments of the whole tile: issued now, their latency hides under the softmax arithmetic below
        const int gi = (lane >> 4) & 1, li = lane & 15;
        h8 vf[2][2][2];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          if (kh >= nkh) break;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int kb0 = kh * 32 + s2 * 16 + 4 * hh + (li >> 2);   // row this lane addresses for j<4
#pragma unroll
            for (int d = 0; d < 2; ++d) {
              const int col = d * 32 + gi * 16 + 4 * (li & 3);          // first of 4 contiguous d this lane addresses
              const int pos = col >> 3, sub = (col & 7) * 2;
              const int r0 = kb0, r1 = kb0 + 8;
              const fp16x4 lo = lds_tr16(sVt + r0 * 128 + ((pos ^ (((r0 >> 1) & 1) << 2)) << 4) + sub);
              const fp16x4 hi = lds_tr16(sVt + r1 * 128 + ((pos ^ (((r1 >> 1) & 1) << 2)) << 4) + sub);
              vf[kh][s2][d][0] = lo[0]; vf[kh][s2][d][1] = lo[1]; vf[kh][s2][d][2] = lo[2]; vf[kh][s2][d][3] = lo[3];
              vf[kh][s2][d][4] = hi[0]; vf[kh][s2][d][5] = hi[1]; vf[kh][s2][d][6] = hi[2]; vf[kh][s2][d][7] = hi[3];
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        // st[kh][r] <-> key = k0 + 32kh + (r&3) + 8(r>>2) + 4*hh, query = q0 + lane%32
        // softmax on RAW scores: max commutes with the positive scale, and exp2(c*s - c*m) is one FMA + one v_exp_f32
        // merged tile: the image-token scores leave `st` for `si` (their own softmax below); what stays in `st` is the text softmax's, exactly as in a tile of its own
        f16v si[2];
        if (k0 + 64 > seg.nkeys) {        // only the last, partial tile of a segment needs masking (wave-uniform)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            if (kh >= nkh) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int key = k0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
              if (merged) si[kh][r] = (key >= n0 && key < live_end) ? st[kh][r] : MASKED;
              if (key >= seg.nkeys) st[kh][r] = MASKED;
            }
          }
        }
        float mx = st[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[0][r]);
        if (nkh == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[1][r]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun, mx);
        const float mc = mnew * p.scale_log2e;
        float psum = 0.f;
        h8 pf[2][2];
        typedef float f2v __attribute__((ext_vector_type(2)));
        if (!ip_tile) {
          const f2v sc2 = {p.scale_log2e, p.scale_log2e}, mc2 = {mc, mc};
          f2v psum2 = {0.f, 0.f};
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            if (kh >= nkh) break;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              // pairs: ONE v_pk_fma_f32 for the two scaled differences, one v_pk_add_f32 into a two-lane running sum (half the fma / add
              // issue slots of the scalar form, and no serial chain through the sum), one v_cvt_pk_f16_f32
#if IA2P_ATTN_PK
              const f2v t = (f2v){st[kh][r], st[kh][r + 1]} * sc2 - mc2;
              const f2v e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
              psum2 += e;
#else
              const f2v e = {__builtin_amdgcn_exp2f(fmaf(st[kh][r], p.scale_log2e, -mc)), __builtin_amdgcn_exp2f(fmaf(st[kh][r + 1], p.scale_log2e, -mc))};
              psum2[0] += e[0]; psum2[0] += e[1];
#endif
              const h2 pr = __builtin_convertvector(e, h2);
              pf[kh][r >> 3][r & 7] = pr[0]; pf[kh][r >> 3][(r & 7) + 1] = pr[1];
            }
          }
          psum = psum2[0] + psum2[1];
          if (__any(mnew != mrun)) {        // running max moved for some query of this wave: rescale (rare after the first tiles)
            const float alpha = __builtin_amdgcn_exp2f((mrun - mnew) * p.scale_log2e);
            lrun *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
              for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            mrun = mnew;
          }
          lrun += psum;
          if (merged) {
            // the image-token keys of this tile: their own softmax (reference attention_processor.py:387), scaled as the stand-alone image-token tile scales it --
            // (w_ip / w_text) * l_text / l_ip, with l_text final now (this IS the last text tile) -- and dropped into the probability slots the text softmax left at zero
            lt = lrun + __shfl_xor(lrun, 32, 64);
            float mxi = si[0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mxi = fmaxf(mxi, si[0][r]);
            if (nkh == 2) {
#pragma unroll
              for (int r = 0; r < 16; ++r) mxi = fmaxf(mxi, si[1][r]);
            }
            mxi = fmaxf(mxi, __shfl_xor(mxi, 32, 64));
            const float mci = mxi * p.scale_log2e;
            float psi = 0.f;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
              if (kh >= nkh) break;
#pragma unroll
              for (int r = 0; r < 16; ++r) { si[kh][r] = __builtin_amdgcn_exp2f(fmaf(si[kh][r], p.scale_log2e, -mci)); psi += si[kh][r]; }
            }
            const float li = psi + __shfl_xor(psi, 32, 64);
            const float w_ip = p.w1_b ? p.w1_b[b] : p.seg[1].weight;
            const float ci = w_ip / p.seg[0].weight * lt / li;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
              if (kh >= nkh) break;
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                const h2 pr = __builtin_convertvector((f2v){si[kh][r] * ci, si[kh][r + 1] * ci}, h2);
                const int key = k0 + kh * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;      // (r even: r and r + 1 are keys key, key + 1)
                if (key >= n0) pf[kh][r >> 3][r & 7] = pr[0];
                if (key + 1 >= n0) pf[kh][r >> 3][(r & 7) + 1] = pr[1];
              }
            }
          }
        } else {
          // image-token tile (MODE 1): its own softmax over its <= 64 keys, all of them in this tile. `o` already holds sum_t e^(s_t - m_t) v_t
          // of the text keys, to be divided by l_t at the end: scale these probabilities by (w_ip / w_text) * l_t / l_ip so that the common
          // division leaves  w_text * text / l_t + w_ip * ip / l_ip  (reference attention_processor.py:371,387,397: two softmaxes, text + scale * ip)
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            if (kh >= nkh) break;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kh][r] = __builtin_amdgcn_exp2f(fmaf(st[kh][r], p.scale_log2e, -mc)); psum += st[kh][r]; }
          }
          const float li = psum + __shfl_xor(psum, 32, 64);
          const float c = seg.weight / p.seg[0].weight * lt / li;
#pragma unroll
          for (int kh = 0; kh < 2; ++kh) {
            if (kh >= nkh) break;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
              const h2 pr = __builtin_convertvector((f2v){st[kh][r] * c, st[kh][r + 1] * c}, h2);
              pf[kh][r >> 3][r & 7] = pr[0]; pf[kh][r >> 3][(r & 7) + 1] = pr[1];
            }
          }
        }

        // ---- O^T[d][q] += V^T[d][key] . P^T[key][q]; k-step (kh,s2): slot (hh, j) <-> key 32kh + 16s2 + 8(j>>2) + 4hh + (j&3)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          if (kh >= nkh) break;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < 2; ++d)
              o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kh][s2][d], pf[kh][s2], o[d], 0, 0, 0);
        }
      };
      f16v sA[2];
      if constexpr (MODE != 0) {           // two-segment launches (short contexts, both segments staged at once): no room for a second score set
        for (int tl = 0; tl < ntile; ++tl) { qk(tl, sA); soft_pv(tl, sA); }
      } else {
        f16v sB[2];
        qk(0, sA);
        for (int tl = 0; tl < ntile; tl += 2) {
          if (tl + 1 < ntile) qk(tl + 1, sB);
          soft_pv(tl, sA);
          if (tl + 1 < ntile) {
            if (tl + 2 < ntile) qk(tl + 2, sA);
            soft_pv(tl + 1, sB);
          }
        }
      }
    }
    if (MODE == 2) {
      const float l = lrun + __shfl_xor(lrun, 32, 64);
      const float w = seg.weight / l;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) otot[d][r] += o[d][r] * w;
    } else if (sg == 0) lt = lrun + __shfl_xor(lrun, 32, 64);
  }
  if (MODE != 2) {
    const float w = p.seg[0].weight / lt;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) otot[d][r] = o[d][r] * w;
  }
}

// O of one wave's 32 queries (otot[d][r] <-> channel 32d + (r&3) + 8(r>>2) + 4*(lane/32) of query q0 + lane%32) through a 4 KiB LDS tile of the
// wave's own -- a query's 64 channels of one head are ONE 128-byte line: written from the accumulator layout they leave as sixteen 8-byte pieces
// per line, read back row-major they leave as whole lines (16 bytes per lane), optionally write-through (wt). The caller has made sure that no
// wave still reads the K / V images under `tile` (a workgroup barrier after the core).
__device__ __forceinline__ void attn_store_o(const f16v (&otot)[2], half_t* O, size_t o_elems, int b, int q0, int hd, int Nq, int ldo, char* tile, int lane, bool wt) {
  const int r31 = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      h4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = (half_t)otot[d][g * 4 + r];
      *(h4*)(tile + r31 * 128 + (((d * 4 + g) ^ (r31 & 7)) << 4) + hh * 8) = v;       // 16-byte chunk d*4+g, XOR-swizzled by the row
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);         // lgkmcnt(0): the wave's own writes are visible to itself
  __builtin_amdgcn_wave_barrier();
  const __amdgpu_buffer_rsrc_t o_rsrc = wt_rsrc((void*)O, o_elems * 2);
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int row = ps * 8 + (lane >> 3), ch = lane & 7;
    const h8 v = *(const h8*)(tile + row * 128 + ((ch ^ (row & 7)) << 4));
    const int q = q0 + row;
    if (q < Nq) {
      const size_t e = ((size_t)b * Nq + q) * ldo + hd * 64 + ch * 8;
      if (wt) store16_wt(o_rsrc, e * 2, v);
      else *(h8*)(O + e) = v;
    }
  }
}
