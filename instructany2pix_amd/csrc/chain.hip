// Two dependent GEMMs in ONE launch: the GEGLU feed-forward of a BasicTransformerBlock, ff.net.0 (GEGLU in, reference diffusers FeedForward behind
// attention_processor.py's blocks) -> ff.net.2 (out projection + residual), without the kernel boundary between them.
//
// Inside a transformer block everything except self-attention is row-local: output rows m of FF-out depend only on rows m of FF-in's output. So a
// tile (tm, .) of FF-out does not need "all of FF-in": it needs the 128-row panel tm of it. The launch holds FF-in's tiles first (block ids
// [0, nA)), then FF-out's; FF-in tiles publish their panel (write-through C stores, drained; one agent-scope add on the panel's arrival counter),
// FF-out tiles wait for their panel's count, take ONE agent-scope acquire and run -- while other panels' FF-in tiles are still in flight. No grid-wide
// barrier, no spin on anything but the own panel (cdna_hip_programming.md §5.6 "chains of small dependent ops", Guideline 16 hand-off recipe).
// hipExtAnyOrderLaunch would give the same overlap across SEPARATE launches; this runtime ignores it on gfx950 (profiles/r03a_anyorder_launch_probe.txt),
// so the chain has to be one kernel: both tile bodies (gemm_kernel.h, unchanged arithmetic) behind a branch on the block id.
//
// Liveness: the hardware dispatches the workgroups of a launch in block-id order, so when a consumer tile is resident every producer tile has been
// dispatched (it is resident or finished) and producers never wait -- no deadlock. HIP does not PROMISE that order, so every spin is bounded and gives
// up into a flag the host checks (results are then wrong, the GPU is not hung); IA2P_CHAIN=0 runs the two launches.
// Results are bit-identical to the two launches (same tiles, same K order, same epilogues): tests/test_ops_gpu.py::test_ffn_chain_*.
#include "gemm_kernel.h"

struct ChainArgs {
  int nA;            // blocks of the first GEMM (tiles + its prefetch workgroups)
  int* cnt;          // arrival counters, one per 128 output rows
  int* done;         // consumers through with a panel (the last one re-arms it)
  int target;        // first-GEMM tiles per panel
  int consumers;     // second-GEMM workgroups per panel (tiles x K-slices)
  unsigned* err;
  int diag;          // IA2P_CHAIN_DIAG (timing diagnostics, results may be wrong)
};

template <int ABN, int AST, int BBN, int BST>       // both tiles are 128 rows high (one row panel), 4 waves
__global__ __launch_bounds__(256, 2) void gemm_chain2_kernel(const half_t* hA, const half_t* hW, const half_t* hzero, int hM, int hN, int hK, int hlda, int hldw, int hrpb, int hbstride,
                                                             int hroff, int hsplitk, int hgroup_w, const GemmArgs pa, const GemmArgs pb, const ChainArgs ch) {
  if ((int)blockIdx.x < ch.nA) {
    const TileCtl ctl{(int)blockIdx.x, nullptr, 0, 1, nullptr, 0, ch.cnt, 128, ch.err, ch.diag};
    gemm_tile_body<128, ABN, AST, false, 2, 64, 0, 2, 0>(hA, hW, hzero, hM, hN, hK, hlda, hldw, hrpb, hbstride, hroff, hsplitk, hgroup_w, pa, nullptr, ctl);
  } else {
    const TileCtl ctl{(int)blockIdx.x - ch.nA, ch.cnt, ch.target, 128, ch.done, ch.consumers, nullptr, 1, ch.err, ch.diag};
    gemm_tile_body<128, BBN, BST, false, 2, 64, 0, 2, 0>(pb.A, pb.W, pb.zero, pb.M, pb.N, pb.K, pb.lda, pb.ldw, pb.rpb, pb.bstride, pb.roff, pb.splitk, pb.group_w, pb, nullptr, ctl);
  }
}

// which (first, second) tile variants have a chained kernel: FF-in on 128x160 or 128x128, FF-out on 128x128 (the plans the tuner picks at M = 2048)
static bool pair_ok(int va, int vb) { return (va == 8 || va == 0) && vb == 0; }

bool ia2p_chain2_ok(const GemmArgs& a, int va, const GemmArgs& b, int vb) {
  return pair_ok(va, vb) && a.M == b.M && a.M % 128 == 0 && a.M / 128 <= 1024 && a.splitk <= 1 && !a.stats_out &&
         b.A == a.C && !b.rpb && !b.ln_stats && (size_t)a.M * a.ldc * 2 < (size_t)0x7ffffff0 && (size_t)b.M * b.ldc * 2 < (size_t)0x7ffffff0 &&
         (b.splitk <= 1 || b.partial);
}

template <int ABN, int AST, int BBN, int BST>
static hipError_t launch_pair(const GemmArgs& a0, const GemmArgs& b0, hipStream_t s) {
  using EA = EpiCfg<128, ABN, AST, 2, 64, 2>;
  using EB = EpiCfg<128, BBN, BST, 2, 64, 2>;
  constexpr int smem = EA::SMEM > EB::SMEM ? EA::SMEM : EB::SMEM;
  static_assert(smem <= 80 * 1024, "two workgroups per CU");
  static bool attr_set[64] = {false};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)gemm_chain2_kernel<ABN, AST, BBN, BST>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  GemmArgs a = a0, b = b0;
  ChainArgs ch;
  if (!ia2p_chain_words(s, &ch.cnt, &ch.done, &ch.err)) return hipErrorOutOfMemory;
  const int tiles_na = (a.N + ABN - 1) / ABN, tiles_a = (a.M / 128) * tiles_na;
  const int tiles_nb = (b.N + BBN - 1) / BBN, tiles_b = (b.M / 128) * tiles_nb;
  const int nsb = b.splitk > 1 ? b.splitk : 1;
  ia2p_gemm_prepare(a, smem, 128, ABN);
  ia2p_gemm_prepare(b, smem, 128, BBN);
  if (!a.vec8 || !b.vec8) return hipErrorInvalidValue;
  a.c_wt = 1;                                     // the hand-off: the consumer reads these rows behind an acquire, so they must be in memory, not dirty in an L2
  a.sk_counters = nullptr;
  b.sk_counters = nsb > 1 ? ia2p_sk_counters(s, tiles_b) : nullptr;
  if (nsb > 1 && !b.sk_counters) return hipErrorOutOfMemory;      // a chained K split must finish inside the launch
  const int pfa = (a.pf && a.pf_bytes >= 4096) ? a.pf_blocks : 0, pfb = (b.pf && b.pf_bytes >= 4096) ? b.pf_blocks : 0;
  ch.nA = tiles_a + pfa;
  ch.target = tiles_na;
  ch.consumers = tiles_nb * nsb;
  static const int diag = getenv("IA2P_CHAIN_DIAG") ? atoi(getenv("IA2P_CHAIN_DIAG")) : 0;
  ch.diag = diag;
  hipLaunchKernelGGL((gemm_chain2_kernel<ABN, AST, BBN, BST>), dim3(ch.nA + tiles_b * nsb + pfb), dim3(256), smem, s, a.A, a.W, a.zero, a.M, a.N, a.K, a.lda, a.ldw, a.rpb, a.bstride,
                     a.roff, a.splitk, a.group_w, a, b, ch);
  return hipGetLastError();
}

hipError_t ia2p_launch_gemm_chain2(const GemmArgs& a, int va, const GemmArgs& b, int vb, hipStream_t s) {
  if (!ia2p_chain2_ok(a, va, b, vb)) return hipErrorInvalidValue;
  if (va == 8) return launch_pair<160, 2, 128, 2>(a, b, s);
  return launch_pair<128, 2, 128, 2>(a, b, s);
}

// give-up flag of the stream's chained launches: 0 = every wait was satisfied (read after a synchronisation; resets the flag)
extern "C" int ia2p_chain_errors(void* stream) {
  int *cnt, *done;
  unsigned* err;
  if (!ia2p_chain_words((hipStream_t)stream, &cnt, &done, &err)) return 0;
  unsigned h = 0;
  if (hipMemcpy(&h, err, sizeof h, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (h) (void)hipMemset(err, 0, sizeof h);
  return (int)h;
}
