// Runtime shared by the three executors (conditional UNet: engine.hip, VAE: vae_engine.hip, CLIP text towers: clip_engine.hip):
// weight arena + parameter table, workspace allocator, weight-prefetch plan, per-kernel event timing, and the operator wrappers that
// plan and launch the HIP kernels. Definitions live in engine.hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ia2p_debug.h"      // (the product ABI, ia2p.h, + the test hooks / profile interface this library also exports)
#include "common.h"

// ---- kernel launchers (gemm.hip, attention.hip, norm.hip, misc.hip) ------------------------------------------
hipError_t ia2p_launch_gemm(const GemmArgs& a, bool conv, hipStream_t s, int* picked, int* combined = nullptr);
// *combined (optional): 1 when a K-split launch finishes inside the launch (ticket counters attached), 0 when its slabs wait for ia2p_launch_splitk_reduce
// *ran (optional): the variant whose kernel the launch really was (a halo-staged variant runs its gathered twin at a site it does not take)
hipError_t ia2p_launch_gemm_variant(const GemmArgs& a, bool conv, int variant, hipStream_t s, bool with_reduce = true, int* combined = nullptr, int* ran = nullptr);
int ia2p_gemm_variant_ran(const GemmArgs& a, bool conv, int v);
bool ia2p_conv_gn_fusable(const GemmArgs& a, int v, int splitk);
bool ia2p_plan_any_gn(int M);
void ia2p_conv_gn_candidates(const GemmArgs& a, size_t max_slab_bytes, std::vector<GemmPlan>* out);
hipError_t ia2p_launch_gn_colstats(const half_t* x, int ldx, int M, int C, int rows, double* out, hipStream_t s);
hipError_t ia2p_launch_gn_apply_stats(const half_t* x0, int ld0, const half_t* x1, int ld1, half_t* y, int ldy, int B, int HW, int C, const GemmArgs::GnIn& g, hipStream_t s);
hipError_t ia2p_launch_splitk_reduce(const GemmArgs& a, hipStream_t s);
bool ia2p_splitk_inkernel(int M, int N, int splitk);
void ia2p_gemm_candidates(int M, int N, int K, bool conv, bool geglu, size_t max_slab_bytes, double slack, std::vector<GemmPlan>* out);
hipError_t ia2p_launch_attention(const AttnArgs& a, hipStream_t s);
bool ia2p_qproj_xattn_ok(const GemmArgs& a, const AttnArgs& x);
hipError_t ia2p_launch_qproj_xattn(const GemmArgs& a, const AttnArgs& x, hipStream_t s);   // qxattn.hip: to_q tile -> attention core, one launch
bool ia2p_qkv_sattn_ok(const GemmArgs& a, const AttnArgs& x);
hipError_t ia2p_launch_qkv_sattn(const GemmArgs& a, const AttnArgs& x, hipStream_t s);      // qxattn.hip: QKV tile of one image x one head -> self-attention, one launch
int ia2p_gn_chunks(int B, int HW);
hipError_t ia2p_launch_groupnorm(const half_t* x, int ldx, half_t* y, int ldy, const half_t* gamma, const half_t* beta,
                                 float* partial, int B, int HW, int C, int G, float eps, int silu, hipStream_t s, const half_t* x2 = nullptr, int ldx2 = 0, int Ca = 0);
hipError_t ia2p_launch_layernorm(const half_t* x, int ldx, half_t* y, int ldy, const half_t* gamma, const half_t* beta,
                                 int M, int C, float eps, hipStream_t s);
hipError_t ia2p_launch_embed(float t, const float* ts, const half_t* text_embeds, const half_t* time_ids, half_t* tsin, half_t* addin,
                             int B, int Tp, int P, int Ad, int nids, hipStream_t s);
hipError_t ia2p_launch_linear_small(const half_t* X, int ldx, const half_t* W, const half_t* bias, const half_t* addend, int ldadd,
                                    half_t* out, int ldo, int M, int N, int K, int silu_in, int silu_out, hipStream_t s);
hipError_t ia2p_launch_conv_in(const half_t* x, const half_t* w, const half_t* bias, half_t* y, int B, int Cin, int H, int W, int Co, hipStream_t s, float out_scale = 1.0f);
hipError_t ia2p_launch_conv_out(const half_t* x, int ldx, const half_t* w, const half_t* bias, half_t* y, int B, int C, int H, int W, int Co, hipStream_t s);
hipError_t ia2p_launch_concat(const half_t* a, int lda, int Ca, const half_t* b, int ldb, int Cb, half_t* y, long M, hipStream_t s);
hipError_t ia2p_launch_ddim_step(const half_t* x, const half_t* eps_u, const half_t* eps_c, float g, float c_x, float c_e,
                                 half_t* out, half_t* out2, long n, hipStream_t s, const float* coef = nullptr, long per = 1);
hipError_t ia2p_launch_mask_blend(const half_t* x, const half_t* init, const half_t* noise, const half_t* mask, float c0, float c1,
                                  half_t* out, half_t* out2, int B, int C, long HW, hipStream_t s);
hipError_t ia2p_launch_cat_rows(const half_t* a, int Ka, const half_t* b, int Kb, const half_t* bias_a, const half_t* bias_b, half_t* dst, half_t* bias_dst, int rows, hipStream_t s);
hipError_t ia2p_launch_fold_ln(const half_t* W, const half_t* gamma, const half_t* beta, const half_t* bias, half_t* Wf, float* cs, float* lb,
                               int N, int K, hipStream_t s);
hipError_t ia2p_launch_clip_embed(const int* ids, const half_t* tok, const half_t* embeds, const half_t* pos, half_t* x, float* stats, int rows, int T, int H,
                                  int vocab, hipStream_t s);
hipError_t ia2p_launch_prior_step(const float* smp, const half_t* o_c, const half_t* o_u, const float* noise, float g, float sqrt_a, float sqrt_b, float k0, float k1,
                                  float sigma, float* out, long n, hipStream_t s);
hipError_t ia2p_launch_causal_attention_small(const half_t* qkv, half_t* out, int B, int T, int heads, hipStream_t s);
hipError_t ia2p_launch_clip_pool(const int* ids, const half_t* x, const half_t* gamma, const half_t* beta, half_t* out, int B, int T, int H, int eos_id,
                                 float eps, hipStream_t s);
hipError_t ia2p_launch_ip_attn_map(const half_t* Q, int ldq, const half_t* Kip, int ldk, half_t* out, int B, int heads, int Nq, int ntok, hipStream_t s);
hipError_t ia2p_launch_touch(const void* p, size_t bytes, unsigned* sink, hipStream_t s);
hipError_t ia2p_launch_pack_conv(const half_t* src, half_t* dst, int Co, int Ci, hipStream_t s);      // [Co][Ci][3][3] -> [Co][tap][Ci]
hipError_t ia2p_launch_pack_conv_in(const half_t* src, half_t* dst, int Co, int KT, hipStream_t s);
hipError_t ia2p_launch_pack_geglu(const half_t* src, half_t* dst, int rows, int rowlen, hipStream_t s);
hipError_t ia2p_launch_softmax_rows(half_t* x, long ld, int rows, int n, float scale, hipStream_t s);
hipError_t ia2p_launch_conv1x1_nchw(const half_t* x, const half_t* w, const half_t* bias, half_t* y, int B, int Ci, int Co, long HW, hipStream_t s);

extern thread_local std::string g_err;   // error of a failed ia2p_*_create / ctx-less entry point
const half_t* zero_page();

enum PKind { PK_COPY = 0, PK_CONV = 1, PK_GEGLU_W = 2, PK_GEGLU_B = 3, PK_PAD_CONV_IN = 4, PK_CONV_TAP = 5 };   // PK_CONV / PK_CONV_TAP: 3x3 weights [Co][tap][Ci] (one layout since round 4; two kinds kept for the callers that name their consumer: implicit GEMM / conv_out_kernel);   // PK_PAD_CONV_IN: d0 = Co, d1 = Cin*9; arena holds [Co][64]

struct Param {
  size_t off;       // element offset in the arena
  size_t elems;     // elements of the tensor as the checkpoint holds it (PK_PAD_CONV_IN occupies d0 * 64 in the arena)
  int kind;
  int d0, d1;       // conv: Co, Ci ; geglu: rows, rowlen
  bool loaded;
  bool optional;    // IP-Adapter tensors
};

struct Block { size_t off, size; bool free_; };

struct Arena {            // deterministic first-fit allocator over [0, cap)
  std::vector<Block> blocks;
  size_t cap, high;
  void reset(size_t c) { cap = c; high = 0; blocks.clear(); blocks.push_back({0, c, true}); }
  size_t alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    for (size_t i = 0; i < blocks.size(); ++i)
      if (blocks[i].free_ && blocks[i].size >= bytes) {
        const size_t off = blocks[i].off;
        if (blocks[i].size > bytes) {
          Block rest{off + bytes, blocks[i].size - bytes, true};
          blocks[i].size = bytes;
          blocks.insert(blocks.begin() + i + 1, rest);
        }
        blocks[i].free_ = false;
        if (off + bytes > high) high = off + bytes;
        return off;
      }
    return (size_t)-1;
  }
  void release(size_t off) {
    for (size_t i = 0; i < blocks.size(); ++i)
      if (blocks[i].off == off && !blocks[i].free_) {
        blocks[i].free_ = true;
        if (i + 1 < blocks.size() && blocks[i + 1].free_) { blocks[i].size += blocks[i + 1].size; blocks.erase(blocks.begin() + i + 1); }
        if (i > 0 && blocks[i - 1].free_) { blocks[i - 1].size += blocks[i].size; blocks.erase(blocks.begin() + i); }
        return;
      }
  }
};

struct ProfRec { hipEvent_t e0, e1; int k; double flops, bytes; int region; double pf; int role; };
// profile regions of a UNet evaluation: what part of the network a launch belongs to (bench.py: conv-block roofline)
enum { PR_OTHER = 0, PR_CONV_BLOCK = 1, PR_TRANSFORMER = 2, PR_NREGION };
// profile ROLES: which layer of the network a launch implements, whatever tile / fusion the plan table picked for it (bench.py keys its roofline by role: the dominant
// kernel INSTANTIATION flips with the tuner's picks, the dominant role does not). Reference ops: attention_processor.py:239-267 / :344-400 (attention projections),
// diffusers' BasicTransformerBlock.ff / ResnetBlock2D / Transformer2DModel.proj_in/out behind pnp_pipeline.py:253-260
enum { ROLE_OTHER = 0, ROLE_FF_IN, ROLE_FF_OUT, ROLE_QKV_SATTN, ROLE_ATTN_OUT, ROLE_Q_XATTN, ROLE_CONV3X3, ROLE_GROUPNORM, ROLE_PROJ_IO, ROLE_CTX_KV, ROLE_EMBED, ROLE_CONV_IO, ROLE_NROLE };
const char* role_name(int r);
// profile classes = device kernel names as rocprofv3 prints them (template arguments included)
enum { PK_GEMM0 = 0, PK_CONV0 = IA2P_GEMM_NVARIANT, PK_ATTN = 2 * IA2P_GEMM_NVARIANT, PK_GN, PK_LN, PK_EMBED, PK_CONV_IN, PK_CONV_OUT, PK_CONCAT, PK_REDUCE, PK_QXATTN, PK_QKVATTN, PK_HALO_GN0 /* + 0 .. 2: the GroupNorm-fused halo-staged tiles 24 .. 26 */, PK_NCLASS = PK_HALO_GN0 + 3 };
const char* prof_name(int k);

// state shared by the executors (conditional UNet, VAE): weights, workspace, prefetch plan, per-kernel timing
void ia2p_sk_counters_invalidate();      // gemm.hip: new epoch of the K-split ticket buffers
int ia2p_default_xattn_min_tiles();      // < 0: the built-in threshold (engine.hip; test hook ia2p_debug_set_xattn_min_tiles)
struct RunCtx {
  std::string err;
  std::unordered_map<std::string, Param> params;
  size_t arena_elems = 0;
  half_t* arena = nullptr;
  bool finalized = false;
  int groups = 32;          // GroupNorm groups
  // run state
  Arena ws;
  char* ws_base = nullptr;
  bool dry = false;
  hipStream_t stream = nullptr;
  bool failed = false;
  // weight prefetch plan: weights of every GEMM/conv launch of a pass, in launch order
  std::vector<std::pair<const half_t*, size_t>> wseq;
  size_t widx = 0;
  bool record = false;
  int wseq_key = -1;
  bool prefetch = true;
  const half_t* tail_pf = nullptr;   // what the LAST contraction of a pass prefetches: the first weights of the next pass (embedding MLPs)
  size_t tail_pf_bytes = 0;
  int xattn_min_tiles = 128; // ... and only when the fused launch has at least this many 128-query x head tiles (40-tile launches lose 4 us each, 160-tile ones gain 1)
  bool cat_free = true;      // up path: torch.cat([hidden, skip]) never materialised (needs sc_fuse; IA2P_CAT_FREE=0: concat_kernel, for A/B runs)
  bool sc_fuse = true;       // ResnetBlock2D: conv2 + conv_shortcut as one implicit GEMM (IA2P_SC_FUSE=0: separate 1x1 launch + residual, for A/B runs)
  bool xattn_fuse = true;    // to_q + cross-attention as one launch where the shape allows (IA2P_XATTN_FUSE=0: two launches, for A/B runs)
  int gn_dry_mode = -1;      // (ia2p_workspace_bytes: the dry passes walk every gn_fuse mode; -1: the context's own)
  int gn_fuse = 1;           // 1: GroupNorm + SiLU of a ResnetBlock2D applied inside the halo-staged convolution that consumes it, statistics from the producers' epilogues;
                             // 0 (IA2P_GN_FUSE=0): GroupNorm launches; 2: the fused path's UNFUSED TWIN -- the same statistics, gn_apply_stats_kernel + the plain convolution (tests: same bits as 1)
  bool sattn_fuse = true;    // QKV projection + self-attention as one launch at 256 tokens per image (follows IA2P_XATTN_FUSE=0; IA2P_SATTN_FUSE in experiment builds)
  bool ln_fold = true;       // LayerNorms folded into their consumer GEMMs (IA2P_LN_FOLD=0: separate layernorm_kernel launches, for A/B runs)
  bool prof = false;
#ifdef IA2P_CLOCK_STAMP
  // diagnostic builds (tools/insitu_stamps.py): the launches of ONE layer role get a stamp buffer (GemmArgs.partial carries it, see gemm_tile.h IA2P_STAMP): 8 x u64 per workgroup,
  // STAMP_WG workgroups per launch slot, launches of the role in executor order
  static constexpr int STAMP_WG = 2048;
  unsigned long long* stamp_buf = nullptr;
  int stamp_role = -1, stamp_cap = 0, stamp_n = 0;
  struct StampMeta { int M, N, K, variant, tiles; };
  std::vector<StampMeta> stamp_meta;
#endif
  // autotune pass (ia2p_autotune): every GEMM / conv site of an unmeasured shape times its candidate plans in place
  bool tuning = false;
  int tune_reps = 5, tune_sites = 0;
  double tune_gn_ms = 0.0;                       // autotune: time of the GroupNorm launch in front of the convolution being tuned next ...
  const GemmArgs* tune_fused = nullptr;          // ... and that convolution's GroupNorm-FUSED form (raw operand + producer statistics): its candidates are timed against GroupNorm launch + plain plan
  char* tune_scratch = nullptr;                 // [slab region | flush region]
  size_t tune_slab_bytes = 0, tune_flush_bytes = 0;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> evpool;
  double p_ms[PK_NCLASS], p_fl[PK_NCLASS], p_by[PK_NCLASS], p_pf[PK_NCLASS];     // p_pf: bytes of the NEXT contraction's weights the class's launches prefetched
  int64_t p_n[PK_NCLASS];
  int region = PR_OTHER;     // region the executor is in (tags the profile records)
  int role = ROLE_OTHER;     // layer role the executor is issuing launches for (tags the profile records)
  double o_ms[ROLE_NROLE], o_fl[ROLE_NROLE], o_by[ROLE_NROLE];
  int64_t o_n[ROLE_NROLE];
  double oc_ms[ROLE_NROLE][PK_NCLASS];      // ... split by kernel class (which instantiations carried the role on this plan table)
  int64_t oc_n[ROLE_NROLE][PK_NCLASS];
  double r_ms[PR_NREGION], r_fl[PR_NREGION], r_by[PR_NREGION];
  int64_t r_n[PR_NREGION];
  float ep_acc_scale = 1.f, ep_bias_scale = 1.f;   // epilogue scales of the NEXT op_gemm / op_conv3 call (reset by it): range extension, vae_engine.hip
  bool fold_dirty = false;   // a LayerNorm-fold source tensor was (re)loaded after the last fold: re-fold before the next forward
  RunCtx() {
    if (const char* e = getenv("IA2P_PREFETCH")) prefetch = atoi(e) != 0;
    if (const char* e = getenv("IA2P_LN_FOLD")) ln_fold = atoi(e) != 0;
    if (const char* e = getenv("IA2P_XATTN_FUSE")) xattn_fuse = sattn_fuse = atoi(e) != 0;
    if (const char* e = getenv("IA2P_GN_FUSE")) gn_fuse = atoi(e);
    if (const char* e = ia2p_exp_env("IA2P_SATTN_FUSE")) sattn_fuse = atoi(e) != 0;
    if (const char* e = ia2p_exp_env("IA2P_SC_FUSE")) sc_fuse = atoi(e) != 0;
    if (const char* e = ia2p_exp_env("IA2P_CAT_FREE")) cat_free = atoi(e) != 0;
    if (ia2p_default_xattn_min_tiles() >= 0) xattn_min_tiles = ia2p_default_xattn_min_tiles();      // (test hook: ia2p_debug_set_xattn_min_tiles)
    for (int k = 0; k < PK_NCLASS; ++k) { p_ms[k] = p_fl[k] = p_by[k] = p_pf[k] = 0; p_n[k] = 0; }
    for (int k = 0; k < PR_NREGION; ++k) { r_ms[k] = r_fl[k] = r_by[k] = 0; r_n[k] = 0; }
    for (int k = 0; k < ROLE_NROLE; ++k) { o_ms[k] = o_fl[k] = o_by[k] = 0; o_n[k] = 0; for (int q = 0; q < PK_NCLASS; ++q) { oc_ms[k][q] = 0; oc_n[k][q] = 0; } }
  }
  ~RunCtx() {
    for (auto& r : recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    for (auto e : evpool) (void)hipEventDestroy(e);
  }
};

ia2p_status fail(RunCtx* c, ia2p_status st, const char* fmt, ...);
// a HIP runtime error seen OUTSIDE CHECK_LAUNCH / RET_HIP (copies, synchronisations, the sampler-update launchers): unless it is a launcher's own refusal of its
// arguments (hipErrorInvalidValue: nothing was launched) a launch may have died mid-flight and left K-split tickets behind -> new ticket epoch, then fail()
ia2p_status fail_hip(RunCtx* c, hipError_t e, const char* what);

struct T2 { size_t off; half_t* p; };   // workspace tensor
T2 wsalloc(RunCtx* c, size_t elems);
void wsfree(RunCtx* c, T2 t);
hipEvent_t get_event(RunCtx* c);

struct ProfScope {
  RunCtx* c; int k; double fl, by; hipEvent_t e0, e1; bool on; double pf = 0;
  ProfScope(RunCtx* c_, int k_, double fl_, double by_) : c(c_), k(k_), fl(fl_), by(by_), on(c_ && c_->prof && !c_->dry) {
    if (on) { e0 = get_event(c); e1 = get_event(c); (void)hipEventRecord(e0, c->stream); }
  }
  ~ProfScope() { if (on) { (void)hipEventRecord(e1, c->stream); c->recs.push_back(ProfRec{e0, e1, k, fl, by, c->region, pf, c->role}); } }
  void set_class(int kk) { k = kk; }
};
struct RoleScope { RunCtx* c; int prev; RoleScope(RunCtx* c_, int r) : c(c_), prev(c_->role) { c->role = r; } ~RoleScope() { c->role = prev; } };
// (a launcher's own refusal of its arguments -- hipErrorInvalidValue, nothing was launched -- leaves the K-split tickets alone; any other error may come from a launch that
//  died mid-flight and starts a new ticket epoch)
#define CHECK_LAUNCH(c, expr, what)                                                             \
  do { if (!(c)->dry && !(c)->failed) { hipError_t e_ = (expr); if (e_ != hipSuccess) { if (e_ != hipErrorInvalidValue) ia2p_sk_counters_invalidate(); fail((c), IA2P_ERR_HIP, "%s: %s", what, hipGetErrorString(e_)); } } } while (0)

inline const half_t* W_(RunCtx* c, size_t off) { return c->arena + off; }

#define RET_HIP(e, what) do { if ((e) == hipSuccess) return IA2P_OK; if ((e) != hipErrorInvalidValue) ia2p_sk_counters_invalidate(); return fail(nullptr, IA2P_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); } while (0)

// GroupNorm statistics of an activation tensor, as its producer's epilogue (or gn_colstats_kernel) left them: [M / rows slots][C] {sum, sum of squares} fp64
struct GnStats { T2 buf{(size_t)-1, nullptr}; int rows = 0; bool ok() const { return rows > 0; } };
// what a producer launch is asked for: statistics of its output, an image being HW rows; filled in by run_gemm / op_gemm / op_conv3
struct GnWant { int HW; GnStats out; };
// GroupNorm fused into a 3x3 convolution (op_conv3): the operand is the raw tensor X (C0 channels) [| X1b (Cin - C0)] with the statistics of its producer(s)
// (tune: autotune pass -- the launch itself is the plain one, on the normalised tensor; Xraw [| X1b] + statistics describe the site's FUSED form for tune_site to time beside it)
struct ConvGn { bool fused = false; const half_t* X1b = nullptr; int C0 = 0; GnStats s0, s1; const half_t* gamma = nullptr; const half_t* beta = nullptr; float eps = 1e-5f; int groups = 32;
                bool tune = false; const half_t* Xraw = nullptr; };
// LayerNorm folded into a GEMM: where the consumer finds the row statistics and the folded constants
struct LnIn { const float* stats; int slots; const float* cs; const float* lb; float eps; };

// ---- operator wrappers: plan (tile / K-split / autotune), slabs, weight prefetch, profiling class, launch
void op_gemm(RunCtx* c, const half_t* A, int lda, const half_t* W, const half_t* bias, const half_t* residual, int ldr,
             half_t* C, int ldc, int M, int N, int K, int geglu = 0, int rpb = 0, int bstride = 0, int roff = 0, int ldw = 0,
             const LnIn* ln = nullptr, float* stats_out = nullptr, int* stat_slots = nullptr, int act = 0, GnWant* gw = nullptr);
void op_conv3(RunCtx* c, const half_t* X, int B, int Hs, int Ws, int Cin, const half_t* W, const half_t* bias, int Co,
              int stride, int up, const half_t* rowvec, int rowvec_ld, const half_t* residual, half_t* Y, int pad_lo = 1, const half_t* X2 = nullptr, int Cin2 = 0, const half_t* X3 = nullptr, int Cin3 = 0,
              const ConvGn* gn = nullptr, GnWant* gw = nullptr);
void op_gn(RunCtx* c, const half_t* x, half_t* y, size_t g, size_t b, int B, int HW, int C, float eps, int silu, float* partial, const half_t* x2 = nullptr, int Ca = 0);
void op_ln(RunCtx* c, const half_t* x, half_t* y, size_t g, size_t b, int M, int C);

// ---- weight arena plumbing shared by the three contexts
ia2p_status rc_bind_arena(RunCtx* c, void* dev, size_t bytes);
ia2p_status rc_load_tensor(RunCtx* c, const char* key, const void* src, const int64_t* shape, int ndim, void* stream);
ia2p_status rc_finalize(RunCtx* c, const char* what);
ia2p_status rc_adopt(RunCtx* c, bool with_optional = true);
