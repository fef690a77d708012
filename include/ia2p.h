/* ia2p.h -- C ABI of libia2p_hip.so: the MI355X (gfx950) denoise hot path of InstructAny2Pix.
 *
 * The reference (pure Python, no FFI of its own) reaches this path through three seams; each entry point
 * below names the reference interface it stands in for (paths relative to the reference checkout):
 *
 *   UNet callable   unet(sample, t, encoder_hidden_states=, added_cond_kwargs={text_embeds,time_ids})[0]
 *                   instructany2pix/ddim/pnp_pipeline.py:253-260, instructany2pix/ddim/sdxl_pipeline.py:832-839
 *                   -> ia2p_unet_forward
 *   operator plugin attn.processor(attn, hidden_states, encoder_hidden_states)
 *                   instructany2pix/diffusion/ip_adapter/attention_processor.py:205-279 (AttnProcessor2_0),
 *                   :310-412 (IPAttnProcessor2_0); installed by ip_adapter.py:120-142, scale set by :211-214
 *                   -> ia2p_set_ip_adapter, ia2p_attention / ia2p_qproj_attention (+ ia2p_gemm for the projections)
 *   sampler update  _backward_ddim pnp_pipeline.py:73-85; CFG combine sdxl_pipeline.py:842-844;
 *                   DDIMScheduler.step (diffusers 0.26.3) called at sdxl_pipeline.py:851
 *                   -> ia2p_ddim_step
 *
 * Conventions: plain pointers and sizes only (no torch types); every `const void*` / `void*` tensor argument
 * is a DEVICE pointer to fp16 data unless stated otherwise; kernels are enqueued on the caller's HIP stream
 * (`stream` is a hipStream_t passed as void*, NULL = default stream) with no hidden synchronisation;
 * functions return IA2P_OK or an error code and never abort; ia2p_last_error() gives the message.
 * A context is bound to the device that was current at ia2p_create and is not re-entrant.
 * Activations inside the library are channels-last ([B*H*W, C]); the latent boundary is NCHW like the reference.
 */
#ifndef IA2P_H
#define IA2P_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ia2p_ctx ia2p_ctx;

typedef enum {
  IA2P_OK = 0,
  IA2P_ERR_INVALID = 1,   /* bad argument (null pointer, negative size, unknown enum) */
  IA2P_ERR_SHAPE = 2,     /* shape / divisibility constraint violated */
  IA2P_ERR_KEY = 3,       /* unknown or duplicate parameter key, or parameters missing at finalize */
  IA2P_ERR_STATE = 4,     /* call order (weights not finalized, arena not bound, ...) */
  IA2P_ERR_NOMEM = 5,     /* arena / workspace too small */
  IA2P_ERR_HIP = 6,       /* HIP runtime error */
  IA2P_ERR_ARCH = 7       /* current device is not gfx950 */
} ia2p_status;

#define IA2P_MAX_BLOCKS 4

/* Mirrors the fields of diffusers' unet/config.json the reference reads (pnp_pipeline.py:44-47, ip_adapter.py:114,124-132). */
typedef struct {
  int in_channels, out_channels;
  int n_blocks;
  int block_out_channels[IA2P_MAX_BLOCKS];
  int transformer_layers_per_block[IA2P_MAX_BLOCKS];   /* 0 = no attention in that block */
  int num_heads[IA2P_MAX_BLOCKS];                      /* diffusers "attention_head_dim" (head counts for SDXL) */
  int layers_per_block;
  int cross_attention_dim;
  int norm_num_groups;
  float norm_eps;
  int addition_time_embed_dim;
  int projection_class_embeddings_input_dim;
  int time_embed_dim;
  int time_proj_dim;
  int mid_transformer_layers;   /* transformer layers of the mid block; < 0 = transformer_layers_per_block[n_blocks-1] (SDXL base). The
                                 * SDXL refiner (pipeline.py:128-131) ends in a plain DownBlock2D but has 4 layers in its mid block. */
  int num_time_ids;             /* micro-conditioning ids per sample: 6 (base: sizes + crop), 5 (refiner: + aesthetic score); <= 0 = 6 */
} ia2p_unet_config;

/* ---- lifetime ---------------------------------------------------------------------------------------------- */
ia2p_status ia2p_create(const ia2p_unet_config* cfg, ia2p_ctx** out);
void ia2p_destroy(ia2p_ctx* ctx);
const char* ia2p_last_error(ia2p_ctx* ctx);            /* ctx may be NULL: error of the failed ia2p_create */
int ia2p_device_is_gfx950(void);

/* ---- weights: one flat, position-deterministic arena (so ranks can RCCL-broadcast it as one buffer) ---------- */
size_t ia2p_arena_bytes(ia2p_ctx* ctx);
ia2p_status ia2p_bind_arena(ia2p_ctx* ctx, void* dev_arena, size_t bytes);   /* caller owns the memory */
/* Load one parameter by its diffusers state-dict key (e.g. "down_blocks.1.attentions.0.transformer_blocks.0.attn1.to_q.weight")
 * or IP-Adapter key ("ip_adapter.<idx>.to_k_ip.weight", idx = position in unet.attn_processors; ip_adapter.py:168-169).
 * `dev_src` is fp16 in the checkpoint's own layout; it is re-laid-out into the arena on `stream`. */
ia2p_status ia2p_load_tensor(ia2p_ctx* ctx, const char* key, const void* dev_src, const int64_t* shape, int ndim, void* stream);
ia2p_status ia2p_finalize_weights(ia2p_ctx* ctx);      /* verifies every UNet parameter was loaded; derives the LayerNorm-folded copies */
/* A tensor (re)loaded after finalize marks the derived data stale; the next forward re-derives it on its stream before running.
 * The arena is [head | tail]: head = the parameters as loaded (ia2p_arena_raw_bytes; what a data-parallel rank must RECEIVE), tail = data
 * derived from them at finalize (LayerNorm-folded weight copies, +43 %). ia2p_adopt_arena: the head was filled elsewhere (RCCL broadcast
 * from the rank that read the checkpoint) -- marks the UNet parameters (and, with_ip_adapter != 0, the IP-Adapter
 * tensors) present and derives the tail locally. */
size_t ia2p_arena_raw_bytes(ia2p_ctx* ctx);
ia2p_status ia2p_adopt_arena(ia2p_ctx* ctx, int with_ip_adapter);
/* the same with the fold kernels ordered on `stream` -- the stream the broadcast that filled the head was enqueued on -- and synchronised there */
ia2p_status ia2p_adopt_arena_on(ia2p_ctx* ctx, int with_ip_adapter, void* stream);
/* The weight distribution of the batch-data-parallel path as ONE call (BASELINE north_star: "RCCL broadcast of UNet weights over xGMI"; the reference itself is
 * single-GPU: pipeline.py:124,131 place one model on one device): ncclBroadcast of the arena head [0, ia2p_arena_raw_bytes) from rank `root` on the caller's
 * communicator and stream, in <= 1 GiB messages, then -- on every rank but `root` -- ia2p_adopt_arena_on (the LayerNorm-folded tail is derived per rank, 2.5 GB stay
 * off xGMI). `rccl_comm` is an ncclComm_t passed as void*, created by the host on the CURRENT device (one process per GPU); the rank is read from it. The stream is
 * synchronised before the call returns. RCCL is bound at run time from the instance already in the process (a PyTorch host: torch's bundled librccl.so.1), else
 * librccl.so.1 is loaded: the library has no link-time dependency on it (ia2p_rccl_available: 1 when the symbols resolved). Every rank of the communicator must call. */
ia2p_status ia2p_bcast_arena(ia2p_ctx* ctx, void* rccl_comm, int root, int with_ip_adapter, void* stream);
int ia2p_rccl_available(void);
/* GroupNorm + SiLU of the ResnetBlock2Ds (reference: diffusers ResnetBlock2D.norm1 / norm2 behind pnp_pipeline.py:253-260): 1 (default; IA2P_GN_FUSE) = applied inside the
 * halo-staged 3x3 convolution that consumes it, statistics from the producers' epilogues; 0 = GroupNorm launches of their own (round 4's path); 2 = the fused path's
 * unfused twin (the same statistics, ia2p_gn_apply_stats-style passes + plain convolutions: bit-identical to 1, for tests). Workspace sizes are valid for every mode. */
ia2p_status ia2p_set_gn_fuse(ia2p_ctx* ctx, int mode);
/* IP-Adapter plugin state: set_ip_adapter (ip_adapter.py:120-142) / set_scale (:211-214) / disable (:153-154). */
ia2p_status ia2p_set_ip_adapter(ia2p_ctx* ctx, int enabled, int num_tokens, float scale);

/* ---- the UNet callable ------------------------------------------------------------------------------------------- */
size_t ia2p_workspace_bytes(ia2p_ctx* ctx, int B, int h, int w, int L);
/* sample, out: [B, in/out_channels, h, w] NCHW; context: [B, L, cross_attention_dim]; text_embeds: [B, pooled];
 * time_ids: [B, num_time_ids]. With the IP-Adapter enabled the last num_tokens rows of each context are the image tokens. */
ia2p_status ia2p_unet_forward(ia2p_ctx* ctx, void* stream, const void* sample, float timestep, const void* context, int L,
                              const void* text_embeds, const void* time_ids, void* out, int B, int h, int w,
                              void* workspace, size_t workspace_bytes);
/* Per-request knobs inside ONE evaluation (batched serving of independent edit requests). timesteps: device, float [B], one
 * timestep per batch element (diffusers' UNet2DConditionModel.forward accepts a [B] timestep tensor; the reference always passes a scalar,
 * pnp_pipeline.py:253-260 / sdxl_pipeline.py:832-839, because it serves one request at a time). ip_scales: device, float [B], or NULL --
 * the IP-Adapter scale of each batch element (reference: one `set_scale` value per call, ip_adapter.py:211-214; used at
 * attention_processor.py:397). Exactly one of context / kv (ia2p_project_context) is non-NULL. Batch element b gets the bits it gets in a
 * uniform batch of the same size evaluated at (timesteps[b], ip_scales[b]). */
ia2p_status ia2p_unet_forward_v(ia2p_ctx* ctx, void* stream, const void* sample, const float* timesteps, const float* ip_scales,
                                const void* context, const void* kv, int L, const void* text_embeds, const void* time_ids, void* out,
                                int B, int h, int w, void* workspace, size_t workspace_bytes);

/* ---- context K/V hoisted out of the step (optional) ------------------------------------------------------------------
 * The K/V projections of every cross-attention layer (reference attention_processor.py:358-359,379-380) depend only on the context and the
 * weights; over the 25-50 denoise steps of a request the context does not change. ia2p_project_context computes them once into a caller-owned
 * buffer (ia2p_context_kv_bytes), ia2p_unet_forward_kv is ia2p_unet_forward reading them from there: same kernels, same bits, one 420-GFLOP GEMM
 * and 1.36 GB of weight streaming less per step. Re-project after changing the context, the weights, or ia2p_set_ip_adapter(enabled, tokens). */
size_t ia2p_context_kv_bytes(ia2p_ctx* ctx, int B, int L);
ia2p_status ia2p_project_context(ia2p_ctx* ctx, void* stream, const void* context, int L, int B, void* kv, size_t kv_bytes,
                                 void* workspace, size_t workspace_bytes);
ia2p_status ia2p_unet_forward_kv(ia2p_ctx* ctx, void* stream, const void* sample, float timestep, const void* kv, int L,
                                 const void* text_embeds, const void* time_ids, void* out, int B, int h, int w,
                                 void* workspace, size_t workspace_bytes);

/* ---- measured kernel plans (optional) ---------------------------------------------------------------------------- */
/* Same arguments as ia2p_unet_forward, plus reps (timed launches per candidate, <1 = 5). Runs one forward in which every
 * GEMM / conv site of a shape without a measured plan times its candidate (tile, K-split) plans in place and records the
 * fastest in a process-wide table that all later launches of that shape use; *sites (optional) = shapes measured.
 * `out` is scratch. Tile choice never changes results; a K-split choice changes fp32 summation order (deterministic per
 * choice), so export the table from one rank and import it on the others to keep ranks bit-identical. Without this call
 * the library uses its built-in cost model. Call ia2p_workspace_bytes again afterwards. New: the reference has no
 * counterpart (cuDNN/cuBLAS pick their kernels internally). */
ia2p_status ia2p_autotune(ia2p_ctx* ctx, void* stream, const void* sample, float timestep, const void* context, int L,
                          const void* text_embeds, const void* time_ids, void* out, int B, int h, int w,
                          void* workspace, size_t workspace_bytes, int reps, int* sites);
size_t ia2p_plan_export(char* buf, size_t len);   /* "M,N,K,conv,geglu,variant,splitk;..." -> buf; returns the length needed */
int ia2p_plan_import(const char* text);           /* entries read, -1 if malformed */
void ia2p_plan_clear(void);
unsigned long long ia2p_plan_generation(void);   /* changes whenever the table does: re-query ia2p_workspace_bytes then */

/* ---- sampler update --------------------------------------------------------------------------------------------- */
/* out = c_x * x + c_e * (eps_u + g * (eps_c - eps_u)); eps_c may be NULL (no guidance); out2 may be NULL. */
ia2p_status ia2p_ddim_step(void* stream, const void* x, const void* eps_u, const void* eps_c, float g, float c_x, float c_e,
                           void* out, void* out2, int64_t n);
/* the same update with per-request coefficients: coef = device float [B][3] = {g, c_x, c_e} of batch element b (`per` elements each), so that
 * requests with their own guidance scale (reference pipeline.py:303 `cfg`) at their own step of their own schedule share one launch */
ia2p_status ia2p_ddim_step_v(void* stream, const void* x, const void* eps_u, const void* eps_c, const float* coef, void* out, void* out2,
                             int B, int64_t per);
/* out = (1 - m) * (c0 * init + c1 * noise) + m * x with m = mask[b, 0, :, :] ([B,1,h,w], shared by the C channels): the per-step
 * blend of the inpainting loop behind `pipe_inpainting` (reference pipeline.py:132-139, gdino/lib.py:89-102; diffusers
 * StableDiffusionXLInpaintPipeline, 4-channel UNet branch). c0 = sqrt(abar_next), c1 = sqrt(1 - abar_next) re-noise the known
 * region to the next timestep (DDIM add_noise); c0 = 1, c1 = 0 after the last step. out2 may be NULL. HW = h * w. */
ia2p_status ia2p_mask_blend(void* stream, const void* x, const void* init, const void* noise, const void* mask, float c0, float c1,
                            void* out, void* out2, int B, int C, int64_t HW);

/* ---- per-operator entry points (unit tests, and hosts that keep their own module tree) ----------------------------- */
/* Operand size: the linear-layer tiles (buffer-load staging), the halo-staged 3x3 convolution and the fused attention tiles address an operand (activations,
 * weights) with a 31-bit byte offset and REFUSE a matrix of 2 GiB or more with IA2P_ERR_HIP / invalid value; the gathered 3x3 convolution kernels use 64-bit
 * pointers and carry no such limit (a halo-ineligible site runs there). The largest operand of the reference's path, the stacked context K/V weights, is 0.68 GB. */
ia2p_status ia2p_groupnorm_silu(void* stream, const void* x, void* y, const void* gamma, const void* beta, int B, int HW, int C,
                                int groups, float eps, int silu, float* partial_ws /* >= B*64*groups*2 floats */);
ia2p_status ia2p_layernorm(void* stream, const void* x, void* y, const void* gamma, const void* beta, int M, int C, float eps);
/* C[M,N] = A[M,K] . W[N,K]^T + bias + residual ; geglu: W/bias packed by ia2p_pack_geglu, C is [M, N/2] */
ia2p_status ia2p_gemm(void* stream, const void* A, const void* W, const void* bias, const void* residual, void* C,
                      int M, int N, int K, int geglu);
/* LayerNorm folded into the contraction that consumes it (what the executor does for norm1/2/3 of every BasicTransformerBlock;
 * reference call sites: diffusers BasicTransformerBlock `attn1(norm1(x))`, `attn2(norm2(x), ctx)`, `ff(norm3(x))` behind
 * pnp_pipeline.py:253-260). Two pieces:
 *   ia2p_fold_layernorm: W [N,K], gamma/beta [K], bias [N] or NULL  ->  Wf = fp16(W * gamma), colsum[n] = sum_k Wf[n][k], fbias = bias + W . beta
 *   ia2p_gemm_ex:        C = epilogue(A . W^T) like ia2p_gemm, plus
 *       ln != NULL   : A holds the UN-normalised rows, W = Wf; out = rstd_m * (acc - mean_m * colsum[n]) + fbias[n] (then GEGLU if geglu);
 *                      mean/rstd from ln->stats = {sum, sum of squares} partials per row, `slots` of them ([slot][M] float2), eps = ln->eps
 *       stats_out    : this launch also writes the {sum, sum^2} partials of ITS fp16 output rows (for the next folded LayerNorm);
 *                      *stats_slots = number of slots written; stats_out must hold (N/64 + 1) * M float2
 *       splitk > 1   : K split as in ia2p_gemm_splitk (partial: splitk*M*N floats); 0/1 = the library's unsplit choice */
typedef struct { const float* stats; int slots; const float* colsum; const float* fbias; float eps; } ia2p_ln_fold;
/* GEGLU feed-forward of a BasicTransformerBlock (diffusers FeedForward: ff.net.0 GEGLU + ff.net.2): H = geglu(X W1p^T + b1p) [M, 4C]
 * (W1p / b1p packed by ia2p_pack_geglu), out = H W2^T + b2 + R, as two launches under the library's plans. splitk > 1: K split of the second GEMM
 * (partial: splitk*M*C floats). */
ia2p_status ia2p_ffn(void* stream, const void* X, const void* W1p, const void* b1p, const void* W2, const void* b2, const void* R, void* H, void* out,
                     int M, int C, int splitk, float* partial);
ia2p_status ia2p_fold_layernorm(void* stream, const void* W, const void* gamma, const void* beta, const void* bias, void* Wf,
                                float* colsum, float* fbias, int N, int K);
ia2p_status ia2p_gemm_ex(void* stream, const void* A, const void* W, const void* bias, const void* residual, void* C, int M, int N, int K,
                         int geglu, const ia2p_ln_fold* ln, float* stats_out, int* stats_slots, int splitk, float* partial);
/* same with K split over `splitk` workgroups per tile (1 .. min(K / 64, 255): the split rides in 8 bits of a preloaded kernel argument); partial holds splitk*M*N floats
 * (deterministic slab reduce) */
ia2p_status ia2p_gemm_splitk(void* stream, const void* A, const void* W, const void* bias, const void* residual, void* C,
                             int M, int N, int K, int splitk, float* partial);
/* 3x3 conv, pad 1, over channels-last x[B,Hs,Ws,Cin] with W packed by ia2p_pack_conv3x3 ([Co][3][3][Cin]);
 * stride 1|2; up=1 convolves the nearest-x2 upsampled x; rowvec [B,Co] (time embedding) and residual optional. */
ia2p_status ia2p_conv3x3(void* stream, const void* x, const void* Wp, const void* bias, const void* rowvec, const void* residual,
                         void* y, int B, int Hs, int Ws, int Cin, int Co, int stride, int up);
/* ---- GroupNorm + SiLU fused into the 3x3 convolution that consumes it (round 5). Reference: diffusers ResnetBlock2D `conv1(nonlinearity(norm1(x)))` /
 * `conv2(dropout(nonlinearity(norm2(h))))` behind instructany2pix/ddim/pnp_pipeline.py:253-260 (in-tree twin: llm/model/vae/modules/blocks.py:122-142).
 * The norm's statistics come from the PRODUCER of its input: per slot of `rows` consecutive rows (one M-tile; HW / rows slots per image) and per channel, fp64
 * {sum, sum of squares}. A producer launch leaves them for its own output (ia2p_conv3x3_gn's gn_out, ia2p_gemm_gnstats); ia2p_gn_colstats computes the same numbers
 * for any tensor. The consumer folds them per image (slot order, then channel order, fp64) and applies y = fp16(silu(fma(x, rstd*gamma, beta - mean*rstd*gamma))):
 * inside the convolution's LDS images (ia2p_conv3x3_gn) or as a pass of its own (ia2p_gn_apply_stats) -- the two agree to the bit. */
ia2p_status ia2p_gn_colstats(void* stream, const void* x, int M, int C, int rows, double* out /* [M / rows][C][2] */);
ia2p_status ia2p_gn_apply_stats(void* stream, const void* x0, int C0, const double* st0, int rows0, const void* x1 /* or NULL */, int C1, const double* st1, int rows1,
                                const void* gamma, const void* beta, void* y /* [B*HW, C0 + C1] */, int B, int HW, int groups, float eps, int silu);
typedef struct {
  const void* x0; int C0; const double* st0; int rows0;     /* operand [B*H*W, C0] and its producer's column sums; st0 == NULL: plain convolution of x0 (no norm) */
  const void* x1; int C1; const double* st1; int rows1;     /* optional second source, channels [C0, C0 + C1): the up path's [hidden | skip] pair, never concatenated */
  const void* gamma; const void* beta; int groups; float eps;   /* the GroupNorm's affine parameters over the C0 + C1 channels */
  const void* Wp; const void* bias; const void* rowvec; const void* residual; void* y;   /* as ia2p_conv3x3 (Wp: ia2p_pack_conv3x3 over C0 + C1 [+ appended Ca columns]) */
  int B, H, W, Co;
  const void* xa; int Ca;                                   /* appended 1x1 block (conv2 + conv_shortcut as one implicit GEMM) or NULL / 0 */
  int splitk; float* partial;                               /* K split (<= 1: none); partial: splitk * B*H*W * Co floats */
  double* gn_out;                                           /* or NULL: column sums of y, [B*H*W / rows][Co][2] with rows = *gn_out_rows */
} ia2p_conv_gn;
ia2p_status ia2p_conv3x3_gn(void* stream, const ia2p_conv_gn* d, int* gn_out_rows);
ia2p_status ia2p_gemm_gnstats(void* stream, const void* A, const void* W, const void* bias, const void* residual, void* C, int M, int N, int K, int splitk, float* partial,
                              int HW, double* gn_out, int* rows);
/* stride 1, no upsampling, with K split over `splitk` workgroups per tile (what the executor launches on the 16 x 16 feature maps); partial holds
 * splitk * B*Hs*Ws * Co floats. Same reference call sites as ia2p_conv3x3 (diffusers ResnetBlock2D conv1 / conv2 behind pnp_pipeline.py:253-260). */
ia2p_status ia2p_conv3x3_splitk(void* stream, const void* x, const void* Wp, const void* bias, const void* rowvec, const void* residual,
                                void* y, int B, int Hs, int Ws, int Cin, int Co, int splitk, float* partial);
/* The tail of a ResnetBlock2D with a channel change as ONE implicit GEMM (what the executor does; diffusers ResnetBlock2D `conv2(h) + conv_shortcut(x)`
 * behind pnp_pipeline.py:253-260):  y = conv3x3(x, W2) + conv1x1(x2, Wsc) + bias, K = 9 Cin + Cin2, stride 1. Wcat [Co][9 Cin + Cin2] holds, per output
 * channel, the ia2p_pack_conv3x3 row of W2 followed by the row of Wsc; bias = b2 + bsc; x [B,Hs,Ws,Cin], x2 [B,Hs,Ws,Cin2] channels-last. */
ia2p_status ia2p_conv3x3_cat(void* stream, const void* x, const void* x2, const void* Wcat, const void* bias, void* y,
                             int B, int Hs, int Ws, int Cin, int Cin2, int Co);
ia2p_status ia2p_pack_conv3x3(void* stream, const void* w_oihw, void* w_packed, int Co, int Cin);   /* [Co][Cin][3][3] -> the implicit-GEMM K order ia2p_conv3x3 / _cat walk ([Co][3][3][Cin] in the shipped library); Cin % 64 == 0. Opaque to callers: pack with this, pass to those */
ia2p_status ia2p_pack_conv_out(void* stream, const void* w_oihw, void* w_packed, int Co, int C);       /* [Co][C][3][3] -> [Co][3][3][C]: the layout ia2p_conv_out reads (any C % 32 == 0) */
ia2p_status ia2p_pack_geglu(void* stream, const void* src, void* dst, int rows, int rowlen);
/* The UNet's latent-boundary 3x3 convolutions (diffusers UNet2DConditionModel.conv_in / .conv_out behind pnp_pipeline.py:253-260; the VAE's too), pad 1:
 *   ia2p_conv_in : x NCHW [B,Cin,H,W] (Cin*9 <= 64), w OIHW [Co,Cin,3,3] (Co % 8 == 0), bias [Co] -> y channels-last [B*H*W, Co];
 *                  w_scratch: Co*64 fp16 elements for the zero-padded [Co][64] weight image the kernel reads (the executors keep it in their arena)
 *   ia2p_conv_out: x channels-last [B*H*W, C] (C % 32 == 0), w packed [Co][3][3][C] (ia2p_pack_conv_out), Co <= 8, bias [Co] -> y NCHW [B,Co,H,W] */
ia2p_status ia2p_conv_in(void* stream, const void* x_nchw, const void* w_oihw, const void* bias, void* y_nhwc, void* w_scratch,
                         int B, int Cin, int H, int W, int Co);
ia2p_status ia2p_conv_out(void* stream, const void* x_nhwc, const void* w_packed, const void* bias, void* y_nchw, int B, int C, int H, int W, int Co);
/* O[b,q,h*64:] = sum_s weight_s * softmax(Q K_s^T / 8) V_s over nseg <= 2 key segments (head_dim 64).
 * Q rows have stride ldq, K_s/V_s rows stride ld_s; segment s has nkeys_s keys per batch. Strides are multiples of 8 elements, O is 16-byte aligned
 * (a query's 64 channels of one head leave as one 128-byte line). */
ia2p_status ia2p_attention(void* stream, const void* Q, int ldq, void* O, int ldo, int B, int heads, int Nq, int nseg,
                           const void* K0, const void* V0, int ld0, int nkeys0, float w0,
                           const void* K1, const void* V1, int ld1, int nkeys1, float w1);
/* `to_q` fused with the cross-attention that consumes it (reference attention_processor.py:344 + :371 / :387 / :397; AttnProcessor2_0 :239 + :259):
 *   O = ia2p_attention(Q = epilogue(X . Wq^T), ...)  in ONE launch, Q never written -- bit-identical to ia2p_gemm_ex on a 128 x 64 tile followed
 *   by ia2p_attention. X [B*Nq, K], Wq [heads*64, K] (gamma-folded when ln != NULL, as ia2p_gemm_ex), bias [heads*64] or NULL (ignored with ln).
 *   Nq must be a multiple of 128 (one tile = 128 queries of one batch element x one head), K of 64. */
ia2p_status ia2p_qproj_attention(void* stream, const void* X, const void* Wq, const void* bias, const ia2p_ln_fold* ln, void* O, int ldo,
                                 int B, int heads, int Nq, int K, int nseg,
                                 const void* K0, const void* V0, int ld0, int nkeys0, float w0,
                                 const void* K1, const void* V1, int ld1, int nkeys1, float w1);
/* Self-attention of AttnProcessor2_0 at 256 tokens per image (the 16 x 16 level) with its three projections, ONE launch (reference attention_processor.py:239 to_q,
 * :246-247 to_k / to_v, :259 scaled_dot_product_attention): O[b, q, h*64:] = softmax(Q K^T / 8) V with [Q | K | V] = epilogue(X . Wqkv^T), the projected
 * tensors never written -- bit-identical to ia2p_gemm_ex (N = 3 * heads * 64) followed by ia2p_attention. X [B*256, K]; Wqkv the stacked [3*heads*64, K]
 * weight (rows: all of to_q, then to_k, then to_v; gamma-folded when ln != NULL, as ia2p_gemm_ex); bias [3*heads*64] or NULL (ignored with ln). */
ia2p_status ia2p_qkv_self_attention(void* stream, const void* X, const void* Wqkv, const void* bias, const ia2p_ln_fold* ln, void* O, int ldo,
                                    int B, int heads, int K);
/* The `attn_map` side effect of IPAttnProcessor2_0 (reference attention_processor.py:390-391; stored on the processor, read only by the
 * attention-map hooks of diffusion/ip_adapter/utils.py:15-20):  out[b,h,q,t] = sum_d Q[b,q,h*64+d] * softmax_t(Kip[b,t,h*64+d]) -- the
 * softmax binds to ip_key^T, i.e. runs over the TOKEN axis, unscaled, before the matmul. Q rows stride ldq, Kip [B*ntok, ldk], out fp16
 * [B, heads, Nq, ntok], ntok <= 16. */
ia2p_status ia2p_ip_attn_map(void* stream, const void* Q, int ldq, const void* Kip, int ldk, void* out, int B, int heads, int Nq, int ntok);
ia2p_status ia2p_linear_small(void* stream, const void* X, const void* W, const void* bias, void* out, int M, int N, int K,
                              int silu_in, int silu_out);

/* Test / tuning hooks (ia2p_debug_*) and the per-kernel timing interface of bench.py's roofline leg (ia2p_profile_*) are declared in ia2p_debug.h: they are
 * exported by the same library but are not part of the product boundary. */

/* ---- VAE (diffusers AutoencoderKL): pipe.vae.encode / pipe.vae.decode ----------------------------
 * reference call sites: ddim/pnp_pipeline.py:190-204 (prepare_latents of the img2img base class), ddim/sdxl_pipeline.py:859-871.
 * NCHW fp16 at both ends; h, w are LATENT sizes in both directions (image side = latent side * 2^(n_blocks-1)).
 * encode returns the posterior moments [B, 2*latent_channels, h, w] (mean | logvar); sampling and the 0.13025 scaling
 * stay on the host. The reference upcasts this model to fp32 (its activations overflow fp16); this build keeps fp16 storage with fp32
 * accumulation and extends the range of the residual stream by a power-of-two storage scale (ia2p_vae_config.stream_scale). */
typedef struct ia2p_vae ia2p_vae;
typedef struct {
  int in_channels, out_channels, latent_channels;
  int n_blocks;
  int block_out_channels[IA2P_MAX_BLOCKS];
  int layers_per_block;
  int norm_num_groups;
  float norm_eps;
  float stream_scale;   /* range extension in place of the reference's fp32 upcast (sdxl_pipeline.py:860-865): the residual stream is stored
                         * multiplied by this power of two in [2^-16, 1] (<= 0 means 1 = plain fp16 storage); 2^-7 covers +-8.4e6 */
} ia2p_vae_config;
ia2p_status ia2p_vae_create(const ia2p_vae_config* cfg, ia2p_vae** out);
void ia2p_vae_destroy(ia2p_vae* vae);
const char* ia2p_vae_last_error(ia2p_vae* vae);
size_t ia2p_vae_arena_bytes(ia2p_vae* vae);
ia2p_status ia2p_vae_bind_arena(ia2p_vae* vae, void* dev_arena, size_t bytes);
ia2p_status ia2p_vae_load_tensor(ia2p_vae* vae, const char* key, const void* dev_src, const int64_t* shape, int ndim, void* stream);
ia2p_status ia2p_vae_finalize_weights(ia2p_vae* vae);
size_t ia2p_vae_workspace_bytes(ia2p_vae* vae, int B, int h, int w, int decode);
ia2p_status ia2p_vae_decode(ia2p_vae* vae, void* stream, const void* latents, void* image, int B, int h, int w, void* workspace, size_t workspace_bytes);
ia2p_status ia2p_vae_encode(ia2p_vae* vae, void* stream, const void* image, void* moments, int B, int h, int w, void* workspace, size_t workspace_bytes);

/* ---- CLIP text encoders (conditioning side): the two encoders behind `encode_prompt` ------------------
 * (reference ddim/sdxl_pipeline.py:202-395: `text_encoder(ids, output_hidden_states=True)`, `.hidden_states[-2]` of both encoders
 * concatenated, pooled `[0]` of the second). transformers `CLIPTextModel` / `CLIPTextModelWithProjection` semantics: token + position
 * embeddings, pre-LayerNorm blocks with causal self-attention (head_dim 64) and a GELU / quick-GELU MLP, final LayerNorm, pooled row
 * = final-normed hidden state at the EOS position (eos_token_id 2: position of the largest id), optional bias-free projection.
 * Parameter keys are the transformers state-dict keys ("text_model.encoder.layers.0.self_attn.q_proj.weight", ...). */
typedef struct ia2p_clip ia2p_clip;
typedef struct {
  int vocab_size, hidden_size, num_layers, num_heads, intermediate_size, max_positions;
  int projection_dim;     /* 0: no text_projection (CLIPTextModel) */
  int hidden_act;         /* 1 = gelu, 2 = quick_gelu, 3 = gelu_new (tanh form; GPT-2) */
  int eos_token_id;
  float layer_norm_eps;
} ia2p_clip_config;
ia2p_status ia2p_clip_create(const ia2p_clip_config* cfg, ia2p_clip** out);
void ia2p_clip_destroy(ia2p_clip* clip);
const char* ia2p_clip_last_error(ia2p_clip* clip);
size_t ia2p_clip_arena_bytes(ia2p_clip* clip);
ia2p_status ia2p_clip_bind_arena(ia2p_clip* clip, void* dev_arena, size_t bytes);
ia2p_status ia2p_clip_load_tensor(ia2p_clip* clip, const char* key, const void* dev_src, const int64_t* shape, int ndim, void* stream);
ia2p_status ia2p_clip_finalize_weights(ia2p_clip* clip);
size_t ia2p_clip_workspace_bytes(ia2p_clip* clip, int B, int T);
/* input_ids: int32 [B, T] on the device (T <= max_positions, <= 128). Outputs (each may be NULL): hidden_penultimate [B, T, hidden]
 * (= hidden_states[-2], the input of the last layer), last_hidden [B, T, hidden] (= last_hidden_state: final LayerNorm of the last
 * layer's output), pooled [B, projection_dim or hidden] (text_embeds / pooler_output). The last layer is skipped when only
 * hidden_penultimate is requested. */
ia2p_status ia2p_clip_encode(ia2p_clip* clip, void* stream, const int32_t* input_ids, int B, int T, void* hidden_penultimate,
                             void* last_hidden, void* pooled, void* workspace, size_t workspace_bytes);
/* The same pre-LayerNorm causal transformer driven with `inputs_embeds` [B, T, hidden] fp16 instead of token ids (position embeddings
 * are added inside): how the reference runs the GPT-2 sequence model of its embedding prior, `self.model(inputs_embeds=...,
 * attention_mask=ones)["last_hidden_state"]` (instructany2pix/prior/model.py:493-495, :611-613; transformers GPT2Model with
 * hidden_act 3, created with vocab_size 0 since the token table is never read). All-ones attention mask only. */
ia2p_status ia2p_clip_encode_embeds(ia2p_clip* clip, void* stream, const void* inputs_embeds, int B, int T, void* hidden_penultimate,
                                    void* last_hidden, void* workspace, size_t workspace_bytes);
/* Sampler update of the embedding prior in fp32 (prior/model.py:208-240 `get_eps`, :627-637 guidance + diffusers DDPMScheduler.step):
 *   eps_i = (sample - sqrt_a * o_i) / sqrt_b;  eps = eps_u + g * (eps_c - eps_u);  x0 = (sample - sqrt_b * eps) / sqrt_a;
 *   out = k0 * x0 + k1 * sample + sigma * noise
 * sample / noise / out: fp32 [n] device; out_cond / out_uncond: fp16 [n] outputs of the sequence model (out_cond NULL = no guidance;
 * noise NULL = none). sqrt_a = sqrt(abar_t), sqrt_b = sqrt(1 - abar_t); k0, k1, sigma from the DDPM posterior (host: scheduler.py). */
ia2p_status ia2p_prior_step(void* stream, const float* sample, const void* out_cond, const void* out_uncond, const float* noise, float g,
                            float sqrt_a, float sqrt_b, float k0, float k1, float sigma, float* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* IA2P_H */
