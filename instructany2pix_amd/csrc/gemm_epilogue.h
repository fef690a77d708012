// The epilogue every MFMA tile shares: takes the fp32 accumulators of a BM x BN tile in the MFMA layout (acc[i][j][r] = C[wm0 + 16 i + (lane & 15)][wn0 + 16 j + 4 (lane >> 4) + r])
// and finishes the launch -- bias / folded LayerNorm / activation / time-embedding row / residual / GEGLU, the K-split combine, the row statistics of the next folded
// LayerNorm, the fused attention tiles (XA) -- exactly as the stand-alone layers of the reference round: a Linear / Conv2d output is an fp16 tensor before anything is added.
#pragma once
#include "gemm_tile.h"
#include "gn_fold.h"

template <int BM, int BN, int NSTAGE, int WGM, int BK, int PP, int WGN, int XA, int HALO>
__device__ __forceinline__ void tile_epilogue(f4 (&acc)[BM / WGM / 16][BN / WGN / 16], const TileCtx& tc, const GemmArgs& p, const AttnArgs* xa, AttnKvRegs& kvr) {
  constexpr int NWAVE = WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int MRH = MR / 2;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hM = tc.hM, hN = tc.hN, hK = tc.hK, tm = tc.tm, tn = tc.tn, bm0 = tc.bm0, bn0 = tc.bn0, tiles_m = tc.tiles_m, tiles_n = tc.tiles_n, split = tc.split, nsplit = tc.nsplit;
  const float ln_s1 = tc.ln_s1, ln_s2 = tc.ln_s2;
  (void)tiles_m; (void)hK;
  auto row_m = [&](int r) { return HALO ? tc.h_m0 + (r >> 4) * p.Wo + (r & 15) : bm0 + r; };      // tile row -> output row
  auto frag_row = [](int i) constexpr { return PP == 2 ? (i / (MR / 2)) * (BM / 2) + (i % (MR / 2)) * 16 : i * 16; };
  auto frag_col = [](int j) constexpr { return PP == 2 ? (j / (NR / 2)) * (BN / 2) + (j % (NR / 2)) * 16 : j * 16; };
  const int wm0 = (wave / WGN) * (PP == 2 ? WM / 2 : WM), wn0 = (wave % WGN) * (PP == 2 ? WN / 2 : WN);
  const int frow = lane & 15, fq = lane >> 4;
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
    static_assert(XA != 4 || (BM == 256 && BN == 192 && WGM == 4 && WGN == 2 && (PP == 0 || PP == 3)), "fused QKV + self-attention: 256 x 192 tiles, 8 waves");
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
    IA2P_STAMP(if (tid == 0 && p.partial) ((unsigned long long*)p.partial)[8 * blockIdx.x + 7] = __builtin_amdgcn_s_memrealtime();)      // Q / K / V images in LDS
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
    IA2P_STAMP(if (tid == 0 && p.partial) ((unsigned long long*)p.partial)[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memrealtime();)      // wave 0's attention core done
    attn_store_o(otot, ap.O, (size_t)ap.B * ap.Nq * ap.ldo, tm, wave * 32, tn, ap.Nq, ap.ldo, sQ + wave * 4096, lane, (ap.xcd_map & 2) != 0);
    IA2P_STAMP(
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0 && p.partial) ((unsigned long long*)p.partial)[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memrealtime();
    )
    return;
  } else if constexpr (XA != 0) {
    // ---- fused to_q + cross-attention (reference attention_processor.py:344 `attn.to_q`, :371 / :387 the two SDPA calls, :397 `text + scale * ip`):
    //      this tile is Q of 128 queries x one head. It goes through the fp32 LDS tile once (projection epilogue applied there, rounded to fp16
    //      exactly as the stand-alone GEMM would store it), comes back as the Q^T fragments of the attention core, and only O is written.
    static_assert(BM == 128 && BN == 64 && WGM == 2 && WGN == 2 && !PP && EC::NCHUNK == 1, "fused cross-attention: 128 x 64 tiles, 4 waves");
    IA2P_STAMP(unsigned long long* xo = p.partial ? (unsigned long long*)p.partial + 8 * blockIdx.x : nullptr;)      // (diagnostic build, tools/micro/qx_clock.hip: the fused launch has no slabs, the field carries the stamp buffer)
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
    attn_kv_store<XA - 1 == 1>(ap, tid, kvr, smem, smem + 32768);     // only short contexts are fused (the launcher checks): their K / V are in registers by now
    __syncthreads();
    IA2P_STAMP(if (tid == 0 && xo) xo[6] = __builtin_amdgcn_s_memrealtime();)      // Q fragments built, K / V images in LDS
    attn_core<XA - 1, true>(ap, b, hd, qf, smem, smem + 32768, tid, otot);
    __syncthreads();                  // every wave is through with the K / V images
    IA2P_STAMP(if (tid == 0 && xo) xo[7] = __builtin_amdgcn_s_memrealtime();)      // attention core done
    attn_store_o(otot, ap.O, (size_t)ap.B * ap.Nq * ap.ldo, b, q0, hd, ap.Nq, ap.ldo, smem + wave * 4096, lane, (ap.xcd_map & 2) != 0);
  IA2P_STAMP(
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && xo) xo[5] = __builtin_amdgcn_s_memrealtime();
  )
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
    IA2P_STAMP(if (IA2P_STAMP_AT != 4) stamp_put(p, nsplit, 7);)      // the fp16 tile is in LDS (IA2P_STAMP_AT == 4: slot 7 is the k-loop's "first k-tile has landed")
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
          if (p.gn_out && tid + (k * U + u) * NT < TOTAL) {      // GroupNorm statistics of the output: the value as stored goes back into the slot it came from (only this thread touches it; dead rows / columns: zeros)
            const h8 zz = {0, 0, 0, 0, 0, 0, 0, 0};
            *(h8*)(t16 + rr[u] * P16 + gg[u] * 16) = live[u] ? o : zz;
          }
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
      if (p.gn_out) {
        // ---- GroupNorm statistics of this tile's output (producer side, gn_fold.h): per column {sum, sum of squares} over the tile rows -- fp32 over aligned runs of 16 rows
        //      (= 16 pixels of one image row, in pixel order, whatever the tile shape), fp64 from there on. Thread (column pair cp, slice sl) takes runs sl, sl + NSL, ...;
        //      the slices meet in LDS (the tile memory, free by then) and thread c < BN adds them up in slice order: deterministic.
        constexpr int CP = BN / 2, NSEG = BM / 16, NSLQ = NT / CP, NSL = NSLQ < NSEG ? NSLQ : NSEG;
        static_assert(NT >= CP && BM % 16 == 0, "column pass: a thread per column pair");
        __syncthreads();                 // the rounded tile is back in LDS
        const int cp = tid % CP, sl = tid / CP;
        double s0 = 0.0, s1 = 0.0, q0 = 0.0, q1 = 0.0;
        if (sl < NSL) {
          for (int sg = sl; sg < NSEG; sg += NSL) {
            float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
            const char* col = t16 + (sg * 16) * P16 + cp * 4;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const h2 v = *(const h2*)(col + i * P16);
              gn_seg16_add((float)v[0], a0, b0); gn_seg16_add((float)v[1], a1, b1);
            }
            s0 += (double)a0; q0 += (double)b0; s1 += (double)a1; q1 += (double)b1;
          }
        }
        __syncthreads();                 // every column pass is through: the tile memory is free
        double2* red = (double2*)smem;   // [NSL][BN]
        if (sl < NSL) { red[sl * BN + 2 * cp] = make_double2(s0, q0); red[sl * BN + 2 * cp + 1] = make_double2(s1, q1); }
        __syncthreads();
        if (tid < BN && bn0 + tid < hN) {
          double a = 0.0, q = 0.0;
#pragma unroll
          for (int x = 0; x < NSL; ++x) { const double2 v = red[x * BN + tid]; a += v.x; q += v.y; }
          ((double2*)p.gn_out)[(size_t)tm * hN + bn0 + tid] = make_double2(a, q);
        }
      }
    }
    pf_sink();
  IA2P_STAMP(
    stamp_put(p, nsplit, 6);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp_put(p, nsplit, 5);
  )
    return;
  }
  }
  bool from_slabs = false;
  double gcol_a = 0.0, gcol_q = 0.0;       // GroupNorm statistics of the output (p.gn_out): thread c < BN: {sum, sum of squares} of tile column c over the chunks done so far
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
    IA2P_STAMP(stamp_put(p, nsplit, 0);)                   // K-split launches: slot 0 = this slice's slab has drained (slots 5 / 6 stay 0 for a slice that is not the last arriver)
    if (tid == 0) *sk_flag = __hip_atomic_fetch_add(p.sk_counters + (tm * tiles_n + tn), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*sk_flag != nsplit - 1) { pf_sink(); return; }
    if (tid == 0) {
      __hip_atomic_store(p.sk_counters + (tm * tiles_n + tn), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (launches are stream-ordered)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // drop this CU's stale lines before the plain loads below
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    IA2P_STAMP(if (IA2P_STAMP_AT == 5) stamp_put(p, nsplit, 7);)      // the last arriver is through its ticket and the acquire fence
    from_slabs = true;
  }
#pragma unroll(PP == 2 ? 2 : 1)
  for (int ch = 0; ch < EC::NCHUNK; ++ch) {
    IA2P_STAMP(if (IA2P_STAMP_AT == 2 && ch == 1) stamp_put(p, nsplit, 7);)      // chunk 0's stores issued
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
    IA2P_STAMP(if (((IA2P_STAMP_AT == 1 && ch == 0) || (IA2P_STAMP_AT == 3 && ch == 1))) stamp_put(p, nsplit, 7);)
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
            if (p.gn_out && tid + (k * U + u) * NT < TOTAL) {      // the value as stored (fp16) goes back into the tile slots it came from (only this thread touches them; dead rows / columns: zeros)
              const f4 w0 = live[u] ? (f4){(float)o[0], (float)o[1], (float)o[2], (float)o[3]} : (f4){0.f, 0.f, 0.f, 0.f};
              const f4 w1 = live[u] ? (f4){(float)o[4], (float)o[5], (float)o[6], (float)o[7]} : (f4){0.f, 0.f, 0.f, 0.f};
              *(f4*)(tile + (size_t)r * PITCH + (((2 * gg[u]) ^ (r & 7)) << 2)) = w0;
              *(f4*)(tile + (size_t)r * PITCH + (((2 * gg[u] + 1) ^ (r & 7)) << 2)) = w1;
            }
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
      if (p.gn_out && fast) {
        // GroupNorm statistics of the output, chunk by chunk (the register epilogue's column pass on the fp32 tile): thread (column quad c4, slice sl) takes the runs of
        // 16 rows sl, sl + GNSL, ... of this chunk; the slices meet in LDS and thread c < BN adds them to its running fp64 column sums
        constexpr int GQ = BN / 4, NSEG = CR / 16, GNSLQ = NT / GQ, GNSL = GNSLQ < NSEG ? GNSLQ : NSEG;
        static_assert(NT >= GQ && CR % 16 == 0, "column pass: a thread per column quad");
        __syncthreads();
        const int c4 = tid % GQ, sl = tid / GQ;
        double ds[4] = {0.0, 0.0, 0.0, 0.0}, dq[4] = {0.0, 0.0, 0.0, 0.0};
        if (sl < GNSL) {
          for (int sg = sl; sg < NSEG; sg += GNSL) {
            float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
              const f4 v = tl(sg * 16 + i, c4);
#pragma unroll
              for (int e = 0; e < 4; ++e) gn_seg16_add(v[e], a[e], b[e]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { ds[e] += (double)a[e]; dq[e] += (double)b[e]; }
          }
        }
        __syncthreads();                 // every column pass is through: the tile memory is free
        double2* red = (double2*)smem;   // [GNSL][BN]
        if (sl < GNSL) {
#pragma unroll
          for (int e = 0; e < 4; ++e) red[sl * BN + c4 * 4 + e] = make_double2(ds[e], dq[e]);
        }
        __syncthreads();
        if (tid < BN) {
#pragma unroll
          for (int x = 0; x < GNSL; ++x) { const double2 v = red[x * BN + tid]; gcol_a += v.x; gcol_q += v.y; }
        }
      }
    }
  }
  if (p.gn_out && fast && tid < BN && bn0 + tid < hN) ((double2*)p.gn_out)[(size_t)tm * hN + bn0 + tid] = make_double2(gcol_a, gcol_q);
  pf_sink();
  IA2P_STAMP(
  stamp_put(p, nsplit, 6);      // this wave has ISSUED its last C store
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the C stores of this wave have left
  __syncthreads();
  stamp_put(p, nsplit, 5);
  )
}
