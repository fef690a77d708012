// Host-side executor of one conditional-UNet evaluation + the C ABI (include/ia2p.h).
//
// The module wiring follows diffusers' UNet2DConditionModel with the SDXL-base config, i.e. the object the
// reference calls at instructany2pix/ddim/pnp_pipeline.py:253-260 and ddim/sdxl_pipeline.py:832-839
// (SURVEY.md §3.4, Appendix A). All launches of a forward are issued from here on the caller's stream: ~1.2k
// launches per evaluation would be host-bound if driven op-by-op from Python.
//
// Memory: weights live in ONE flat fp16 arena whose layout depends only on the config (so data-parallel ranks
// can receive it with a single RCCL broadcast); activations come from a caller-provided workspace managed by a
// deterministic first-fit allocator (sized by a dry run of the same code path).
#include "engine_rt.h"

#include <dlfcn.h>
#include <rccl/rccl.h>      // types and prototypes only: the symbols are bound at run time (ia2p_bcast_arena), the library is not linked

thread_local std::string g_err;

const half_t* zero_page() {
  static thread_local void* z[16] = {nullptr};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  if (!z[dev]) {
    if (hipMalloc(&z[dev], 256) != hipSuccess) return nullptr;
    (void)hipMemset(z[dev], 0, 256);
  }
  return (const half_t*)z[dev];
}

struct Resnet {
  int cin, cout, temb_off;
  bool shortcut;
  size_t n1g, n1b, w1, b1, n2g, n2b, w2, b2, wsc, bsc;
  size_t wcat, bcat;      // derived at finalize when the block has a shortcut: [cout][9 cout + cin] = conv2 rows with the shortcut rows appended; b2 + bsc
};
struct TBlock {
  size_t ln1g, ln1b, wqkv, wo1, bo1, ln2g, ln2b, wq2, wkv2, wkvip, wo2, bo2, ln3g, ln3b, wff1, bff1, wff2, bff2;
  // LayerNorms folded into their consumers at finalize (fold_all): gamma-scaled weight copies + fp32 column sums / biases.
  // The raw tensors above stay as loaded, so finalize can be repeated and single tensors reloaded.
  size_t fqkv, fq2, fff1, cs1, lb1, cs2, lb2, cs3, lb3;
  int kv_col;   // column of this layer's [K | V] block in the batched context projection
};
struct Transformer {
  int c, heads;
  size_t ng, nb, win, bin, wout, bout;
  std::vector<TBlock> blocks;
};
struct Stage {            // one down/up block
  std::vector<Resnet> res;
  std::vector<Transformer> att;   // empty or same length as res
  bool resample;
  size_t rw, rb;
  int rc;
};

static int g_xattn_min_tiles = -1;      // test hook: fusion threshold of contexts created from now on (< 0: the built-in 128)
int ia2p_default_xattn_min_tiles() { return g_xattn_min_tiles; }
extern "C" void ia2p_debug_set_xattn_min_tiles(int tiles) { g_xattn_min_tiles = tiles; }

const char* prof_name(int k) {
  static char buf[PK_NCLASS][64];
  static const char* const other[] = {"attention_f16_kernel", "gn_stats_kernel+gn_apply_kernel", "layernorm_kernel",
                                      "embed_kernel+linear_small_kernel", "conv_in_kernel", "conv_out_kernel", "concat_kernel", "splitk_reduce_kernel", "qproj_xattn_kernel", "qkv_sattn_kernel"};
  if (k >= PK_HALO_GN0) { snprintf(buf[k], sizeof buf[k], "conv_halo_f16_kernel<%d, 1>", IA2P_GEMM_TILES[24 + k - PK_HALO_GN0].bn); return buf[k]; }
  if (k >= PK_ATTN) return other[k - PK_ATTN];
  const GemmTile t = IA2P_GEMM_TILES[(k % PK_CONV0) % IA2P_GEMM_NVARIANT];
  if (t.halo) snprintf(buf[k], sizeof buf[k], "conv_halo_f16_kernel<%d, 0>", t.bn);
  else if (t.pp == 4) snprintf(buf[k], sizeof buf[k], "gemm_geglu_f16_kernel");
  else if (t.pp == 2) snprintf(buf[k], sizeof buf[k], "gemm_f16_kernel<%d, %d, %d, %s, 2, 64, 2, 4>", t.bm, t.bn, t.stages, k >= PK_CONV0 ? "true" : "false");
  else if (t.pp) snprintf(buf[k], sizeof buf[k], "gemm_f16_kernel<%d, %d, %d, %s, 4, 64, 1, 2>", t.bm, t.bn, t.stages, k >= PK_CONV0 ? "true" : "false");
  else if (t.bn == 80) snprintf(buf[k], sizeof buf[k], "gemm_f16_kernel<%d, %d, %d, %s, 4, 64, 0, 1>", t.bm, t.bn, t.stages, k >= PK_CONV0 ? "true" : "false");
  else snprintf(buf[k], sizeof buf[k], "gemm_f16_kernel<%d, %d, %d, %s, 2, 64, 0, 2>", t.bm, t.bn, t.stages, k >= PK_CONV0 ? "true" : "false");
  return buf[k];
}

const char* role_name(int r) {
  static const char* const names[ROLE_NROLE] = {"other", "ff_in (GEGLU projection, norm3 folded)", "ff_out", "qkv + self-attention (norm1 folded)", "attention out-projections (attn1 / attn2 to_out)",
                                                "to_q + cross-attention (norm2 folded)", "conv3x3 (ResnetBlock2D convs incl. fused shortcut, resample convs)", "groupnorm (+SiLU)",
                                                "proj_in / proj_out", "context K/V projection", "time / add embeddings", "conv_in / conv_out"};
  return r >= 0 && r < ROLE_NROLE ? names[r] : "?";
}

struct ia2p_ctx : RunCtx {
  ia2p_unet_config cfg;
  int ip_enabled = 0, ip_tokens = 4;
  float ip_scale = 1.0f;
  // plan
  size_t conv_in_w, conv_in_b, te1w, te1b, te2w, te2b, ae1w, ae1b, ae2w, ae2b, tw_all, tb_all, ngo, nbo, conv_out_w, conv_out_b;
  int temb_total = 0;
  size_t embed_lo = 0, embed_hi = 0;
  // context K/V projection weights of ALL cross-attention layers, stacked [kv_rows, ctx] (text) / (image tokens):
  // the context is the same for every layer, so one GEMM per step projects it for all of them
  size_t kv_text_base = 0, kv_ip_base = 0;
  int kv_rows = 0;
  std::vector<Stage> down, up;
  Resnet mid_r0, mid_r1;
  Transformer mid_t;
  int n_attn2 = 0;
  size_t arena_raw_elems = 0;   // head of the arena: everything a checkpoint provides; [arena_raw_elems, arena_elems) is derived at finalize
};

ia2p_status fail(RunCtx* c, ia2p_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) { c->err = buf; c->failed = true; }
  g_err = buf;
  return st;      // (K-split tickets: invalidated where a launch / sync error is seen -- CHECK_LAUNCH, RET_HIP, fail_hip -- not for argument refusals)
}
ia2p_status fail_hip(RunCtx* c, hipError_t e, const char* what) {
  if (e != hipErrorInvalidValue) ia2p_sk_counters_invalidate();
  return fail(c, IA2P_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

// ---------------------------------------------------------------------------------------------------------------------
// plan: enumerate parameters in the diffusers key layout (same walk as instructany2pix_amd/weights.py)
// ---------------------------------------------------------------------------------------------------------------------
struct Planner {
  ia2p_ctx* c;
  size_t cur = 0;
  int kv_cursor = 0;
  // derived data (LayerNorm-folded weight copies, fp32 column sums / biases) goes to a TAIL region [fold_base, fold_base + fold_cur): the
  // head [0, fold_base) then holds exactly what a checkpoint provides, and a rank that receives only the head re-derives the tail itself
  size_t fold_base = 0, fold_cur = 0;
  size_t take(size_t elems) { size_t o = cur; cur += (elems + 127) & ~(size_t)127; return o; }
  size_t take_fold(size_t elems) { size_t o = fold_base + fold_cur; fold_cur += (elems + 127) & ~(size_t)127; return o; }
  void reg(const std::string& key, size_t off, size_t elems, int kind = PK_COPY, int d0 = 0, int d1 = 0, bool optional = false) {
    c->params[key] = Param{off, elems, kind, d0, d1, false, optional};
  }
  size_t vec(const std::string& key, int n) { size_t o = take(n); reg(key, o, n); return o; }
  size_t mat(const std::string& key, int rows, int cols) { size_t o = take((size_t)rows * cols); reg(key, o, (size_t)rows * cols); return o; }
  size_t conv3(const std::string& key, int co, int ci, PKind kind = PK_CONV) { size_t o = take((size_t)co * ci * 9); reg(key, o, (size_t)co * ci * 9, kind, co, ci); return o; }

  Resnet resnet(const std::string& p, int cin, int cout, int temb, size_t tw_all, size_t tb_all) {
    Resnet r;
    r.cin = cin; r.cout = cout; r.shortcut = cin != cout;
    r.n1g = vec(p + ".norm1.weight", cin); r.n1b = vec(p + ".norm1.bias", cin);
    r.w1 = conv3(p + ".conv1.weight", cout, cin); r.b1 = vec(p + ".conv1.bias", cout);
    r.temb_off = c->temb_total;
    reg(p + ".time_emb_proj.weight", tw_all + (size_t)r.temb_off * temb, (size_t)cout * temb);
    reg(p + ".time_emb_proj.bias", tb_all + r.temb_off, cout);
    c->temb_total += cout;
    r.n2g = vec(p + ".norm2.weight", cout); r.n2b = vec(p + ".norm2.bias", cout);
    r.w2 = conv3(p + ".conv2.weight", cout, cout); r.b2 = vec(p + ".conv2.bias", cout);
    r.wsc = r.bsc = 0;
    r.wcat = r.bcat = 0;
    if (r.shortcut) {
      r.wsc = mat(p + ".conv_shortcut.weight", cout, cin); r.bsc = vec(p + ".conv_shortcut.bias", cout);
      r.wcat = take_fold((size_t)cout * (9 * cout + cin)); r.bcat = take_fold(cout);
    }
    return r;
  }
  Transformer transformer(const std::string& p, int ch, int heads, int depth, int ctx, std::vector<std::pair<std::string, size_t>>& ipslots) {
    Transformer t;
    t.c = ch; t.heads = heads;
    t.ng = vec(p + ".norm.weight", ch); t.nb = vec(p + ".norm.bias", ch);
    t.win = mat(p + ".proj_in.weight", ch, ch); t.bin = vec(p + ".proj_in.bias", ch);
    for (int k = 0; k < depth; ++k) {
      const std::string q = p + ".transformer_blocks." + std::to_string(k);
      TBlock b;
      b.ln1g = vec(q + ".norm1.weight", ch); b.ln1b = vec(q + ".norm1.bias", ch);
      b.wqkv = take((size_t)3 * ch * ch);
      reg(q + ".attn1.to_q.weight", b.wqkv, (size_t)ch * ch);
      reg(q + ".attn1.to_k.weight", b.wqkv + (size_t)ch * ch, (size_t)ch * ch);
      reg(q + ".attn1.to_v.weight", b.wqkv + (size_t)2 * ch * ch, (size_t)ch * ch);
      b.wo1 = mat(q + ".attn1.to_out.0.weight", ch, ch); b.bo1 = vec(q + ".attn1.to_out.0.bias", ch);
      b.ln2g = vec(q + ".norm2.weight", ch); b.ln2b = vec(q + ".norm2.bias", ch);
      b.wq2 = mat(q + ".attn2.to_q.weight", ch, ch);
      b.kv_col = kv_cursor;
      b.wkv2 = c->kv_text_base + (size_t)kv_cursor * ctx;
      reg(q + ".attn2.to_k.weight", b.wkv2, (size_t)ch * ctx);
      reg(q + ".attn2.to_v.weight", b.wkv2 + (size_t)ch * ctx, (size_t)ch * ctx);
      b.wkvip = c->kv_ip_base + (size_t)kv_cursor * ctx;
      ipslots.push_back({q, b.wkvip});
      kv_cursor += 2 * ch;
      b.wo2 = mat(q + ".attn2.to_out.0.weight", ch, ch); b.bo2 = vec(q + ".attn2.to_out.0.bias", ch);
      b.ln3g = vec(q + ".norm3.weight", ch); b.ln3b = vec(q + ".norm3.bias", ch);
      b.wff1 = take((size_t)8 * ch * ch); reg(q + ".ff.net.0.proj.weight", b.wff1, (size_t)8 * ch * ch, PK_GEGLU_W, 8 * ch, ch);
      b.bff1 = take((size_t)8 * ch); reg(q + ".ff.net.0.proj.bias", b.bff1, (size_t)8 * ch, PK_GEGLU_B, 8 * ch, 1);
      b.wff2 = mat(q + ".ff.net.2.weight", ch, 4 * ch); b.bff2 = vec(q + ".ff.net.2.bias", ch);
      b.fqkv = take_fold((size_t)3 * ch * ch); b.fq2 = take_fold((size_t)ch * ch); b.fff1 = take_fold((size_t)8 * ch * ch);
      b.cs1 = take_fold((size_t)2 * 3 * ch); b.lb1 = take_fold((size_t)2 * 3 * ch);         // fp32 arrays: 2 half-slots per value
      b.cs2 = take_fold((size_t)2 * ch); b.lb2 = take_fold((size_t)2 * ch);
      b.cs3 = take_fold((size_t)2 * 8 * ch); b.lb3 = take_fold((size_t)2 * 8 * ch);
      t.blocks.push_back(b);
    }
    t.wout = mat(p + ".proj_out.weight", ch, ch); t.bout = vec(p + ".proj_out.bias", ch);
    return t;
  }
};

static ia2p_status plan_pass(ia2p_ctx* c, size_t fold_base, size_t* raw_elems, size_t* fold_elems) {
  c->params.clear(); c->down.clear(); c->up.clear();
  const ia2p_unet_config& g = c->cfg;
  const int n = g.n_blocks;
  if (n < 1 || n > IA2P_MAX_BLOCKS) return fail(c, IA2P_ERR_INVALID, "n_blocks %d out of range", n);
  for (int i = 0; i < n; ++i) {
    const int ch = g.block_out_channels[i];
    if (ch % 64 || ch % g.norm_num_groups) return fail(c, IA2P_ERR_SHAPE, "block_out_channels[%d]=%d must be a multiple of 64 and of norm_num_groups", i, ch);
    if (g.transformer_layers_per_block[i] > 0 && g.num_heads[i] * 64 != ch)
      return fail(c, IA2P_ERR_SHAPE, "block %d: heads*64 must equal channels (head_dim is fixed to 64)", i);
  }
  if (g.cross_attention_dim % 64 || g.time_embed_dim % 8 || g.projection_class_embeddings_input_dim % 8 || g.time_proj_dim % 8 || g.time_proj_dim % 2 || g.addition_time_embed_dim % 2)
    return fail(c, IA2P_ERR_SHAPE, "embedding / context dims violate the 8/64 divisibility rules");
  if (g.in_channels * 9 > 64 || g.out_channels > 8) return fail(c, IA2P_ERR_SHAPE, "latent channels too large for the boundary convolutions");
  if (c->cfg.num_time_ids <= 0) c->cfg.num_time_ids = 6;
  if (c->cfg.mid_transformer_layers < 0) c->cfg.mid_transformer_layers = g.transformer_layers_per_block[n - 1];
  if (g.num_time_ids > 8) return fail(c, IA2P_ERR_SHAPE, "num_time_ids %d out of range", g.num_time_ids);
  if (g.mid_transformer_layers < 1 || g.num_heads[n - 1] * 64 != g.block_out_channels[n - 1])
    return fail(c, IA2P_ERR_SHAPE, "mid block: needs >= 1 transformer layer and heads*64 == channels");
  const int pooled = g.projection_class_embeddings_input_dim - g.num_time_ids * g.addition_time_embed_dim;
  if (pooled <= 0) return fail(c, IA2P_ERR_SHAPE, "projection_class_embeddings_input_dim smaller than the time ids");

  Planner P{c};
  P.fold_base = fold_base;
  const int T = g.time_embed_dim, ctx = g.cross_attention_dim;
  const int* ch = g.block_out_channels;
  c->conv_in_w = P.take((size_t)ch[0] * 64); P.reg("conv_in.weight", c->conv_in_w, (size_t)ch[0] * g.in_channels * 9, PK_PAD_CONV_IN, ch[0], g.in_channels * 9);
  c->conv_in_b = P.vec("conv_in.bias", ch[0]);
  c->te1w = P.mat("time_embedding.linear_1.weight", T, g.time_proj_dim); c->te1b = P.vec("time_embedding.linear_1.bias", T);
  c->te2w = P.mat("time_embedding.linear_2.weight", T, T); c->te2b = P.vec("time_embedding.linear_2.bias", T);
  c->ae1w = P.mat("add_embedding.linear_1.weight", T, g.projection_class_embeddings_input_dim); c->ae1b = P.vec("add_embedding.linear_1.bias", T);
  c->ae2w = P.mat("add_embedding.linear_2.weight", T, T); c->ae2b = P.vec("add_embedding.linear_2.bias", T);
  // all time_emb_proj matrices stacked into one [sum Cout, T] projection
  int nres = 0, tot = 0;
  {
    int skipn = 0;
    for (int i = 0; i < n; ++i) { nres += g.layers_per_block; tot += g.layers_per_block * ch[i]; }
    tot += 2 * ch[n - 1];
    for (int i = 0; i < n; ++i) tot += (g.layers_per_block + 1) * ch[n - 1 - i];
    (void)skipn; (void)nres;
  }
  c->tw_all = P.take((size_t)tot * T);
  c->tb_all = P.take(tot);
  c->embed_lo = c->te1w; c->embed_hi = P.cur;      // [time/add embedding MLPs | stacked time_emb_proj]: the first weights a pass reads
  c->temb_total = 0;
  {
    int rows = 0;
    for (int i = 0; i < n; ++i) rows += g.layers_per_block * g.transformer_layers_per_block[i] * 2 * ch[i];          // down
    rows += g.mid_transformer_layers * 2 * ch[n - 1];                                                               // mid
    for (int i = 0; i < n; ++i) rows += (g.layers_per_block + 1) * g.transformer_layers_per_block[n - 1 - i] * 2 * ch[n - 1 - i];   // up
    c->kv_rows = rows;
    c->kv_text_base = P.take((size_t)rows * ctx);
    c->kv_ip_base = P.take((size_t)rows * ctx);
  }

  std::vector<std::pair<std::string, size_t>> ipslots_down, ipslots_up, ipslots_mid;
  std::vector<int> skip_ch;
  skip_ch.push_back(ch[0]);
  int cprev = ch[0];
  for (int i = 0; i < n; ++i) {
    Stage st;
    const std::string bp = "down_blocks." + std::to_string(i);
    for (int j = 0; j < g.layers_per_block; ++j) {
      st.res.push_back(P.resnet(bp + ".resnets." + std::to_string(j), j == 0 ? cprev : ch[i], ch[i], T, c->tw_all, c->tb_all));
      if (g.transformer_layers_per_block[i] > 0)
        st.att.push_back(P.transformer(bp + ".attentions." + std::to_string(j), ch[i], g.num_heads[i], g.transformer_layers_per_block[i], ctx, ipslots_down));
      skip_ch.push_back(ch[i]);
    }
    cprev = ch[i];
    st.resample = i != n - 1;
    st.rc = ch[i];
    if (st.resample) {
      st.rw = P.conv3(bp + ".downsamplers.0.conv.weight", ch[i], ch[i]); st.rb = P.vec(bp + ".downsamplers.0.conv.bias", ch[i]);
      skip_ch.push_back(ch[i]);
    }
    c->down.push_back(st);
  }
  const int cm = ch[n - 1];
  c->mid_r0 = P.resnet("mid_block.resnets.0", cm, cm, T, c->tw_all, c->tb_all);
  c->mid_t = P.transformer("mid_block.attentions.0", cm, g.num_heads[n - 1], g.mid_transformer_layers, ctx, ipslots_mid);
  c->mid_r1 = P.resnet("mid_block.resnets.1", cm, cm, T, c->tw_all, c->tb_all);
  cprev = cm;
  for (int i = 0; i < n; ++i) {
    Stage st;
    const int co = ch[n - 1 - i];
    const std::string bp = "up_blocks." + std::to_string(i);
    for (int j = 0; j < g.layers_per_block + 1; ++j) {
      const int cs = skip_ch.back(); skip_ch.pop_back();
      st.res.push_back(P.resnet(bp + ".resnets." + std::to_string(j), (j == 0 ? cprev : co) + cs, co, T, c->tw_all, c->tb_all));
      if (g.transformer_layers_per_block[n - 1 - i] > 0)
        st.att.push_back(P.transformer(bp + ".attentions." + std::to_string(j), co, g.num_heads[n - 1 - i], g.transformer_layers_per_block[n - 1 - i], ctx, ipslots_up));
    }
    cprev = co;
    st.resample = i != n - 1;
    st.rc = co;
    if (st.resample) { st.rw = P.conv3(bp + ".upsamplers.0.conv.weight", co, co); st.rb = P.vec(bp + ".upsamplers.0.conv.bias", co); }
    c->up.push_back(st);
  }
  if (c->temb_total != tot) return fail(c, IA2P_ERR_STATE, "internal: time_emb_proj stacking mismatch %d != %d", c->temb_total, tot);
  if (P.kv_cursor != c->kv_rows) return fail(c, IA2P_ERR_STATE, "internal: context K/V stacking mismatch %d != %d", P.kv_cursor, c->kv_rows);
  c->ngo = P.vec("conv_norm_out.weight", ch[0]); c->nbo = P.vec("conv_norm_out.bias", ch[0]);
  c->conv_out_w = P.conv3("conv_out.weight", g.out_channels, ch[0], PK_CONV_TAP); c->conv_out_b = P.vec("conv_out.bias", g.out_channels);

  // IP-Adapter keys: index = position in unet.attn_processors = down, up, mid (attn1 even, attn2 odd)
  int idx = 0;
  auto add_ip = [&](std::vector<std::pair<std::string, size_t>>& v) {
    for (auto& s : v) {
      const std::string& q = s.first;
      // channel count from the attn2.to_q registration
      const size_t cc = (size_t)std::llround(std::sqrt((double)c->params[q + ".attn2.to_q.weight"].elems));
      const std::string k = "ip_adapter." + std::to_string(2 * idx + 1);
      P.reg(k + ".to_k_ip.weight", s.second, cc * ctx, PK_COPY, 0, 0, true);
      P.reg(k + ".to_v_ip.weight", s.second + cc * ctx, cc * ctx, PK_COPY, 0, 0, true);
      ++idx;
    }
  };
  add_ip(ipslots_down); add_ip(ipslots_up); add_ip(ipslots_mid);
  c->n_attn2 = idx;
  *raw_elems = P.cur; *fold_elems = P.fold_cur;
  return IA2P_OK;
}
// two passes of the same deterministic walk: the first measures the head (checkpoint data), the second places the derived tail behind it
static ia2p_status build_plan(ia2p_ctx* c) {
  size_t raw = 0, fold = 0;
  ia2p_status st = plan_pass(c, 0, &raw, &fold);
  if (st != IA2P_OK) return st;
  st = plan_pass(c, raw, &raw, &fold);
  if (st != IA2P_OK) return st;
  c->arena_raw_elems = raw;
  c->arena_elems = raw + fold;
  return IA2P_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// run helpers
// ---------------------------------------------------------------------------------------------------------------------
T2 wsalloc(RunCtx* c, size_t elems) {
  const size_t off = c->ws.alloc(elems * sizeof(half_t));
  if (off == (size_t)-1) { if (!c->failed) fail(c, IA2P_ERR_NOMEM, "workspace too small"); return T2{off, nullptr}; }
  return T2{off, c->dry ? nullptr : (half_t*)(c->ws_base + off)};
}
void wsfree(RunCtx* c, T2 t) {
  if (t.off == (size_t)-1) return;
  c->ws.release(t.off);
}

hipEvent_t get_event(RunCtx* c) {
  if (!c->evpool.empty()) { hipEvent_t e = c->evpool.back(); c->evpool.pop_back(); return e; }
  hipEvent_t e; (void)hipEventCreate(&e); return e;
}
static void set_prefetch(RunCtx* c, GemmArgs& a, const half_t* W, size_t bytes) {
  if (c->dry) { if (c->record) c->wseq.push_back({W, bytes}); return; }
  if (!c->prefetch || c->widx + 1 > c->wseq.size()) { ++c->widx; return; }
  const bool last = c->widx + 1 == c->wseq.size();
  if (last && !c->tail_pf) { ++c->widx; return; }
  const std::pair<const half_t*, size_t> nx = last ? std::make_pair(c->tail_pf, c->tail_pf_bytes) : c->wseq[c->widx + 1];
  ++c->widx;
  if (nx.second > ((size_t)96 << 20)) return;           // larger than the Infinity Cache can usefully hold
  a.pf = nx.first; a.pf_bytes = (long)nx.second;
  static const size_t pf_cap = ia2p_exp_env("IA2P_PF_BLOCKS") ? (size_t)atoi(ia2p_exp_env("IA2P_PF_BLOCKS")) : 128;            // tuning hooks: most prefetch workgroups per launch,
  static const size_t pf_per = ia2p_exp_env("IA2P_PF_BLOCK_BYTES") ? (size_t)atol(ia2p_exp_env("IA2P_PF_BLOCK_BYTES")) : 131072;   // bytes per workgroup below that
  a.pf_blocks = (int)std::max<size_t>(1, std::min<size_t>(pf_cap, (nx.second + pf_per - 1) / pf_per));
}

// In-place measurement of the candidate plans of one GEMM / conv site (autotune pass). Every candidate runs once untimed and
// tune_reps times bracketed by events (round-robin over the candidates), with the L2s flushed (a memset over the flush region)
// before every launch and the site's activations read back in: in the real sequence the activations were just written and the weights sit in the Infinity Cache (prefetched by
// the previous launch), not in L2. The prefetch workgroups of the site are part of every candidate launch. The fastest
// goes into the plan table. Re-running a site is harmless: outputs are rewritten (in-place residuals only drift).
static void tune_site(RunCtx* c, const GemmArgs& a, bool conv) {
  // (what the caller left for THIS site -- taken and cleared before anything else: the pointer refers to the caller's frame)
  const GemmArgs* fa = c->tune_fused;
  const float gn_ms = (float)c->tune_gn_ms;
  c->tune_fused = nullptr; c->tune_gn_ms = 0.0;
  if (ia2p_plan_lookup(a.M, a.N, a.K, conv, a.geglu != 0, nullptr)) return;
  std::vector<GemmPlan> cands;
  ia2p_gemm_candidates(a.M, a.N, a.K, conv, a.geglu != 0, c->tune_slab_bytes, ia2p_exp_env("IA2P_TUNE_SLACK") ? atof(ia2p_exp_env("IA2P_TUNE_SLACK")) : 2.5, &cands);      // (2.5 x the modelled best: -0.04 ms per step against 1.7 on the same box, profiles/r05t_slack_ab.txt; 4.0 measures 4 x the candidates for the same picks)
  static const bool tune_log = getenv("IA2P_TUNE_LOG") != nullptr;      // every candidate's time, for calibrating the cost model
  hipEvent_t e0 = get_event(c), e1 = get_event(c);
  // Rounds over all candidates (round -1 untimed), so that clock / cache drift during the measurement hits every candidate alike;
  // a candidate's score is its FASTEST round (launch-time noise only ever adds).
  std::vector<float> best_ms(cands.size(), 1e30f);
  std::vector<char> ok(cands.size(), 1);
  for (size_t i = 0; i < cands.size(); ++i)
    if (IA2P_GEMM_TILES[cands[i].variant].halo && !(conv && ia2p_conv_halo_ok(a) && cands[i].splitk <= a.Cin / 64)) ok[i] = 0;      // (this site is not one for the halo-staged kernel: its twin tile is in the list anyway)
  for (int r = -1; r < c->tune_reps; ++r)
    for (size_t i = 0; i < cands.size(); ++i) {
      if (!ok[i]) continue;
      GemmArgs b = a;
      b.splitk = cands[i].splitk > 1 ? cands[i].splitk : 0;
      b.partial = cands[i].splitk > 1 ? (float*)c->tune_scratch : nullptr;
      bool good = hipMemsetAsync(c->tune_scratch + c->tune_slab_bytes, r & 1, c->tune_flush_bytes, c->stream) == hipSuccess;
      // ... but the site's activations (and residual) were written by the launch just before it in the real sequence: warm them again
      const size_t a_bytes = conv ? (size_t)(a.M / std::max(1, a.Ho * a.Wo)) * a.Hs * a.Ws * a.Cin * sizeof(half_t)
                                  : (a.rpb ? 0 : (size_t)a.M * a.lda * sizeof(half_t));
      good = good && ia2p_launch_touch(a.A, std::min<size_t>(a_bytes, (size_t)64 << 20), (unsigned*)c->tune_scratch, c->stream) == hipSuccess;
      if (a.residual) good = good && ia2p_launch_touch(a.residual, std::min<size_t>((size_t)a.M * a.ldr * sizeof(half_t), (size_t)64 << 20), (unsigned*)c->tune_scratch, c->stream) == hipSuccess;
      good = good && hipEventRecord(e0, c->stream) == hipSuccess;
      good = good && ia2p_launch_gemm_variant(b, conv, cands[i].variant, c->stream) == hipSuccess;
      good = good && hipEventRecord(e1, c->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
      float ms = 0.f;
      good = good && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
      if (!good) { (void)hipGetLastError(); ok[i] = 0; continue; }
      if (r >= 0 && ms < best_ms[i]) best_ms[i] = ms;
    }
  c->evpool.push_back(e0); c->evpool.push_back(e1);
  float fastest = 1e30f;
  for (size_t i = 0; i < cands.size(); ++i) {
    if (!ok[i]) continue;
    if (tune_log) fprintf(stderr, "[ia2p tune] %d %d %d conv=%d geglu=%d variant=%d splitk=%d us=%.2f\n", a.M, a.N, a.K, (int)conv, a.geglu, cands[i].variant, cands[i].splitk, 1e3 * best_ms[i]);
    fastest = std::min(fastest, best_ms[i]);
  }
  // candidates come best-modelled first: among those within 2 % of the fastest measurement the model's favourite wins (stable picks)
  GemmPlan best{-1, 1};
  for (size_t i = 0; i < cands.size() && best.variant < 0; ++i)
    if (ok[i] && best_ms[i] <= 1.02f * fastest) best = cands[i];
  // A site behind a GroupNorm (the caller left c->tune_fused: the site's GroupNorm-FUSED form -- raw operand, producer statistics -- and c->tune_gn_ms: what the
  // GroupNorm launch in front of it just took): the fused launch on every halo-staged tile / K split it may run on is timed the same way, and taken when it beats
  // GroupNorm launch + best plain plan. The fused kernel normalises every halo image in LDS beside its MFMAs (+15 ... 25 % per launch): it pays where the norm's
  // launch is expensive against the convolution (few input channels on the large maps), not everywhere.
  if (fa && conv && best.variant >= 0 && gn_ms > 0.f) {
    std::vector<GemmPlan> fc;
    ia2p_conv_gn_candidates(*fa, c->tune_slab_bytes, &fc);
    std::vector<float> fms(fc.size(), 1e30f);
    hipEvent_t f0 = get_event(c), f1 = get_event(c);
    for (int r = -1; r < c->tune_reps; ++r)
      for (size_t i = 0; i < fc.size(); ++i) {
        GemmArgs b = *fa;
        b.splitk = fc[i].splitk > 1 ? fc[i].splitk : 0;
        b.partial = fc[i].splitk > 1 ? (float*)c->tune_scratch : nullptr;
        bool good = hipMemsetAsync(c->tune_scratch + c->tune_slab_bytes, r & 1, c->tune_flush_bytes, c->stream) == hipSuccess;
        const size_t a_bytes = (size_t)(fa->M / std::max(1, fa->Ho * fa->Wo)) * fa->Hs * fa->Ws * fa->gn.C0 * sizeof(half_t);
        good = good && ia2p_launch_touch(fa->A, std::min<size_t>(a_bytes, (size_t)64 << 20), (unsigned*)c->tune_scratch, c->stream) == hipSuccess;
        if (fa->A1b) good = good && ia2p_launch_touch(fa->A1b, std::min<size_t>((size_t)fa->M * fa->lda1b * sizeof(half_t), (size_t)64 << 20), (unsigned*)c->tune_scratch, c->stream) == hipSuccess;
        if (fa->residual) good = good && ia2p_launch_touch(fa->residual, std::min<size_t>((size_t)fa->M * fa->ldr * sizeof(half_t), (size_t)64 << 20), (unsigned*)c->tune_scratch, c->stream) == hipSuccess;
        good = good && hipEventRecord(f0, c->stream) == hipSuccess;
        good = good && ia2p_launch_gemm_variant(b, true, fc[i].variant, c->stream) == hipSuccess;
        good = good && hipEventRecord(f1, c->stream) == hipSuccess && hipEventSynchronize(f1) == hipSuccess;
        float ms = 0.f;
        good = good && hipEventElapsedTime(&ms, f0, f1) == hipSuccess;
        if (!good) { (void)hipGetLastError(); fms[i] = -1.f; continue; }
        if (r >= 0 && fms[i] >= 0.f && ms < fms[i]) fms[i] = ms;
      }
    c->evpool.push_back(f0); c->evpool.push_back(f1);
    float unfused = 1e30f;
    for (size_t i = 0; i < cands.size(); ++i)
      if (ok[i]) unfused = std::min(unfused, best_ms[i]);
    unfused += gn_ms;
    int bi = -1;
    for (size_t i = 0; i < fc.size(); ++i) {
      if (tune_log && fms[i] >= 0.f) fprintf(stderr, "[ia2p tune] %d %d %d conv=1 FUSED groupnorm variant=%d splitk=%d us=%.2f (groupnorm launch %.2f us + best plain plan = %.2f us)\n", fa->M, fa->N, fa->K, fc[i].variant, fc[i].splitk,
                                             1e3 * fms[i], 1e3 * gn_ms, 1e3 * unfused);
      if (fms[i] >= 0.f && fms[i] < 1e29f && (bi < 0 || fms[i] < fms[bi])) bi = (int)i;
    }
    if (bi >= 0 && fms[bi] < 0.97f * unfused) best = fc[bi];      // (3 % margin: the fused form also pays for the statistics in its producers' epilogues)
  }
  if (best.variant < 0) { fail(c, IA2P_ERR_HIP, "autotune: no candidate plan ran for %d x %d x %d", a.M, a.N, a.K); return; }
  ia2p_plan_set(a.M, a.N, a.K, conv, a.geglu != 0, best);
  ++c->tune_sites;
}

// plan, K-split slabs, profiling class and launch of one GEMM / implicit-GEMM conv
// rows per slot of the stand-alone statistics pass over an image of HW rows: the largest multiple of 16 that divides HW and is <= 1024 (0: none)
static int gn_fallback_rows(int HW) {
  for (int k = 1; k <= HW / 16; ++k)
    if (HW % k == 0 && (HW / k) % 16 == 0 && HW / k <= 1024) return HW / k;
  return 0;
}
// rows per slot of the column sums a launch's own epilogue leaves (0: it cannot): whole 16-row runs, whole tiles per image, 16-byte epilogue routes, at most
// IA2P_GN_MAX_SLOTS slots per image; a K split only when it combines inside the launch
static int gn_epilogue_rows(const GemmArgs& a, bool conv, int variant, int splitk, bool combined, int HW) {
  const int bm = IA2P_GEMM_TILES[ia2p_gemm_variant_ran(a, conv, variant)].bm;
  return (!a.geglu && !a.act && a.ldc % 8 == 0 && a.N % 8 == 0 && bm % 16 == 0 && HW % bm == 0 && a.M % HW == 0 && HW / bm <= IA2P_GN_MAX_SLOTS && (splitk <= 1 || combined)) ? bm : 0;
}
// gw != nullptr: the launch also leaves the GroupNorm statistics of its output (gn_fold.h) -- from its own epilogue when the tile allows, else from a gn_colstats_kernel pass
static void run_gemm(RunCtx* c, GemmArgs& a, bool conv, const char* what, double flops, double bytes, int* stat_slots = nullptr, GnWant* gw = nullptr) {
  if (c->tuning && !c->dry && !c->failed) tune_site(c, a, conv);
  c->tune_fused = nullptr; c->tune_gn_ms = 0.0;
  const GemmPlan pl = ia2p_gemm_plan(a.M, a.N, a.K, conv, a.geglu != 0);
  if (pl.variant < 0 || pl.variant >= IA2P_GEMM_NVARIANT) { fail(c, IA2P_ERR_INVALID, "%s: tile variant %d out of range", what, pl.variant); return; }
  if (c->tuning && !c->dry && pl.splitk > 1 && (size_t)pl.splitk * a.M * a.N * sizeof(float) > c->tune_slab_bytes) {
    fail(c, IA2P_ERR_NOMEM, "%s: plan (variant %d, K split %d) needs %zu bytes of slabs, the autotune scratch holds %zu", what, pl.variant, pl.splitk,
         (size_t)pl.splitk * a.M * a.N * sizeof(float), c->tune_slab_bytes);
    return;
  }
  T2 slab{(size_t)-1, nullptr};
  if (pl.splitk > 1) {
    a.splitk = pl.splitk;
    if (c->tuning && !c->dry) a.partial = (float*)c->tune_scratch;      // plans change during the pass: slabs live outside the workspace
    else { slab = wsalloc(c, (size_t)pl.splitk * a.M * a.N * 2); a.partial = (float*)slab.p; }
  }
  struct Rel { RunCtx* c; T2 t; ~Rel() { wsfree(c, t); } } rel{c, slab};
  int combined = pl.splitk > 1 && ia2p_splitk_inkernel(a.M, a.N, pl.splitk);     // (dry pass: the policy's answer; the launcher reports what it really did)
  int gn_rows_epi = 0;
  if (gw) {      // (a launch whose tile cannot take the sums leaves none: the consumer that wants them runs the canonical pass itself, gn_ensure_stats)
    gw->out = GnStats{};
    gn_rows_epi = gn_epilogue_rows(a, conv, pl.variant, pl.splitk, combined != 0, gw->HW);
    if (gn_rows_epi) {
      gw->out.buf = wsalloc(c, (size_t)(a.M / gn_rows_epi) * a.N * 8);      // double2 per slot and column
      a.gn_out = (double*)gw->out.buf.p;
    }
  }
#ifdef IA2P_CLOCK_STAMP
  if (c->stamp_buf && c->role == c->stamp_role && (pl.splitk <= 1 || combined) && !c->dry && !c->tuning && c->stamp_n < c->stamp_cap) {
    const GemmTile& t = IA2P_GEMM_TILES[pl.variant];
    const int tiles = ((a.M + t.bm - 1) / t.bm) * ((a.N + t.bn - 1) / t.bn) * (pl.splitk > 1 ? pl.splitk : 1);      // (workgroups: a K split launches one per tile and slice)
    if (tiles <= RunCtx::STAMP_WG && !t.halo) {
      unsigned long long* rec = c->stamp_buf + (size_t)c->stamp_n * RunCtx::STAMP_WG * 8;
      if (pl.splitk > 1) a.stamp = rec;
      else a.partial = (float*)rec;
      c->stamp_meta.push_back({a.M, a.N, a.K, pl.splitk > 1 ? -100 * pl.splitk - pl.variant : pl.variant, tiles});
      ++c->stamp_n;
    }
  }
#endif
  {
    ProfScope ps(c, (conv ? PK_CONV0 : PK_GEMM0) + pl.variant, flops, bytes);
    ps.pf = a.pf ? (double)a.pf_bytes : 0.0;
    int ran = pl.variant;
    CHECK_LAUNCH(c, ia2p_launch_gemm_variant(a, conv, pl.variant, c->stream, false, &combined, &ran), what);
    ps.set_class(conv && a.gn.st0 && ran >= 24 && ran <= 26 ? PK_HALO_GN0 + ran - 24 : (conv ? PK_CONV0 : PK_GEMM0) + ran);      // (a halo-staged plan runs its gathered twin at a site it does not take: booked under the kernel that ran)
  }
  if (pl.splitk > 1 && !combined) {
    ProfScope ps(c, PK_REDUCE, 0, (double)pl.splitk * a.M * a.N * 4 + 2.0 * a.M * a.N);
    CHECK_LAUNCH(c, ia2p_launch_splitk_reduce(a, c->stream), what);
  }
  if (gw && gw->out.buf.off != (size_t)-1) {
    if (gn_rows_epi && (pl.splitk <= 1 || combined)) gw->out.rows = gn_rows_epi;
    else { wsfree(c, gw->out.buf); gw->out = GnStats{}; }      // (the launcher finished the K split with a reduce launch after all)
  }
  // row-statistics slots of this launch's output: one per tile column, or ONE when a reduce launch wrote it
  if (stat_slots) *stat_slots = (pl.splitk > 1 && !combined) ? 1 : (a.N + IA2P_GEMM_TILES[pl.variant].bn - 1) / IA2P_GEMM_TILES[pl.variant].bn;
}

static GemmArgs gemm_args(RunCtx* c, const half_t* A, int lda, const half_t* W, const half_t* bias, const half_t* residual, int ldr,
                          half_t* C, int ldc, int M, int N, int K, int geglu, int rpb, int bstride, int roff, int ldw,
                          const LnIn* ln, float* stats_out, int act) {
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  if (ln) { a.ln_stats = ln->stats; a.ln_slots = ln->slots; a.ln_cs = ln->cs; a.ln_bias = ln->lb; a.ln_eps = ln->eps; }
  a.stats_out = stats_out;
  a.act = act;
  a.A = A; a.W = W; a.C = C; a.zero = zero_page(); a.M = M; a.N = N; a.K = K; a.ldw = ldw ? ldw : K; a.lda = lda; a.ldc = ldc;
  a.rpb = rpb; a.bstride = bstride; a.roff = roff; a.bias = bias; a.residual = residual; a.ldr = ldr; a.geglu = geglu;
  a.rows_per_batch = 1;
  a.m_fastest = (long)M * K <= (long)N * K ? 1 : 0;
  a.acc_scale = c->ep_acc_scale; a.bias_scale = c->ep_bias_scale; c->ep_acc_scale = c->ep_bias_scale = 1.f;
  return a;
}
static double gemm_bytes(int M, int N, int K, int geglu, bool residual) { return 2.0 * ((double)M * K + (double)N * K + (double)M * (geglu ? N / 2 : N) + (residual ? (double)M * N : 0)); }
void op_gemm(RunCtx* c, const half_t* A, int lda, const half_t* W, const half_t* bias, const half_t* residual, int ldr,
             half_t* C, int ldc, int M, int N, int K, int geglu, int rpb, int bstride, int roff, int ldw,
             const LnIn* ln, float* stats_out, int* stat_slots, int act, GnWant* gw) {
  GemmArgs a = gemm_args(c, A, lda, W, bias, residual, ldr, C, ldc, M, N, K, geglu, rpb, bstride, roff, ldw, ln, stats_out, act);
  set_prefetch(c, a, W, (size_t)N * K * sizeof(half_t));
  run_gemm(c, a, false, "gemm", 2.0 * M * N * K, gemm_bytes(M, N, K, geglu, residual != nullptr), stat_slots, gw);
}
// GEGLU feed-forward: ff.net.0 (a: K = C, N = 8 C packed, GEGLU epilogue -> H [M, 4 C]) then ff.net.2 (b: reads H, + bias + residual): two launches.
// (One launch with a per-row-panel hand-off between the two was built and measured in round 3: +0.45 ... +0.8 ms per step, docs/LOG.md; removed in round 4.)
static void run_ffn(RunCtx* c, GemmArgs& a, GemmArgs& b, int* stat_slots_b) {
  set_prefetch(c, a, a.W, (size_t)a.N * a.K * sizeof(half_t));
  set_prefetch(c, b, b.W, (size_t)b.N * b.K * sizeof(half_t));
  { RoleScope role(c, ROLE_FF_IN); run_gemm(c, a, false, "ff.net.0", 2.0 * a.M * (double)a.N * a.K, gemm_bytes(a.M, a.N, a.K, 1, false)); }
  RoleScope role(c, ROLE_FF_OUT);
  run_gemm(c, b, false, "ff.net.2", 2.0 * b.M * (double)b.N * b.K, gemm_bytes(b.M, b.N, b.K, 0, true), stat_slots_b);
}
void op_conv3(RunCtx* c, const half_t* X, int B, int Hs, int Ws, int Cin, const half_t* W, const half_t* bias, int Co,
              int stride, int up, const half_t* rowvec, int rowvec_ld, const half_t* residual, half_t* Y, int pad_lo, const half_t* X2, int Cin2,
              const half_t* X3, int Cin3, const ConvGn* gn, GnWant* gw) {
  GemmArgs a;
  memset(&a, 0, sizeof a);
  // (appended blocks are described by their channel counts: in a dry pass the pointers are null, the shapes -- hence plans and slabs -- must not change)
  if ((Cin2 > 0 && (stride != 1 || up || pad_lo != 1 || Cin2 % 64)) || (Cin3 > 0 && (Cin2 <= 0 || Cin3 % 64)) || Cin2 < 0 || Cin3 < 0 ||
      (!c->dry && ((Cin2 > 0) != (X2 != nullptr) || (Cin3 > 0) != (X3 != nullptr)))) { fail(c, IA2P_ERR_SHAPE, "conv3x3 with appended 1x1 blocks: stride 1, no upsampling, Cin2 / Cin3 %% 64 == 0 (stride %d up %d pad %d Cin %d Cin2 %d Cin3 %d, X2 %s, X3 %s)", stride, up, pad_lo, Cin, Cin2, Cin3, X2 ? "set" : "null", X3 ? "set" : "null"); return; }
  a.pad = pad_lo;           // zero rows/cols before the image; one row/col of zeros after it in every mode
  const int Hv = Hs << up, Wv = Ws << up;
  a.Ho = (Hv + pad_lo + 1 - 3) / stride + 1; a.Wo = (Wv + pad_lo + 1 - 3) / stride + 1;
  a.A = X; a.W = W; a.C = Y; a.zero = zero_page(); a.M = B * a.Ho * a.Wo; a.N = Co; a.K = 9 * Cin + Cin2 + Cin3; a.ldw = a.K; a.lda = Cin; a.ldc = Co;
  if (gn && gn->fused) {      // GroupNorm + SiLU applied inside the convolution: X [| X1b] is the RAW input of the norm (the caller asked ia2p_conv_gn_fusable)
    a.lda = gn->C0; a.A1b = gn->X1b; a.lda1b = Cin - gn->C0;
    a.gn.st0 = (const double*)gn->s0.buf.p; a.gn.rows0 = gn->s0.rows; a.gn.st1 = (const double*)gn->s1.buf.p; a.gn.rows1 = gn->s1.rows; a.gn.C0 = gn->C0;
    a.gn.gamma = gn->gamma; a.gn.beta = gn->beta; a.gn.groups = gn->groups; a.gn.gs = Cin / gn->groups; a.gn.eps = gn->eps; a.gn.silu = 1;
  }
  a.A2 = X2; a.lda2 = Cin2; a.Cin2 = Cin2;
  a.A3 = X3; a.lda3 = Cin3; a.Cin3 = Cin3;
  a.Hs = Hs; a.Ws = Ws; a.stride = stride; a.up = up; a.Cin = Cin;
  a.bias = bias; a.rowvec = rowvec; a.rowvec_ld = rowvec_ld; a.rows_per_batch = a.Ho * a.Wo; a.residual = residual; a.ldr = Co;
  a.m_fastest = 0;
  a.acc_scale = c->ep_acc_scale; a.bias_scale = c->ep_bias_scale; c->ep_acc_scale = c->ep_bias_scale = 1.f;
  set_prefetch(c, a, W, (size_t)Co * a.K * sizeof(half_t));
  GemmArgs fa;      // autotune pass: the site's GroupNorm-fused form, for tune_site to time against GroupNorm launch + plain plan
  if (gn && gn->tune && !gn->fused && c->tuning && !c->dry && gn->s0.ok() && gn->Xraw) {
    fa = a;
    fa.A = gn->Xraw; fa.lda = gn->C0; fa.A1b = gn->X1b; fa.lda1b = Cin - gn->C0;
    fa.gn.st0 = (const double*)gn->s0.buf.p; fa.gn.rows0 = gn->s0.rows; fa.gn.st1 = gn->X1b ? (const double*)gn->s1.buf.p : nullptr; fa.gn.rows1 = gn->s1.rows; fa.gn.C0 = gn->C0;
    fa.gn.gamma = gn->gamma; fa.gn.beta = gn->beta; fa.gn.groups = gn->groups; fa.gn.gs = Cin / gn->groups; fa.gn.eps = gn->eps; fa.gn.silu = 1;
    if (ia2p_conv_gn_ok(fa)) c->tune_fused = &fa;
  }
  RoleScope role(c, ROLE_CONV3X3);
  run_gemm(c, a, true, "conv3x3", 2.0 * a.M * (double)Co * a.K, 2.0 * ((double)B * Hs * Ws * Cin + (double)Co * a.K + (double)a.M * Co + (residual ? (double)a.M * Co : 0) + (double)a.M * (Cin2 + Cin3)), nullptr, gw);
}
void op_gn(RunCtx* c, const half_t* x, half_t* y, size_t g, size_t b, int B, int HW, int C, float eps, int silu, float* partial, const half_t* x2, int Ca) {
  RoleScope role(c, ROLE_GROUPNORM);
  ProfScope ps(c, PK_GN, 8.0 * B * HW * C, 4.0 * B * HW * C);
  if (x2) CHECK_LAUNCH(c, ia2p_launch_groupnorm(x, Ca, y, C, W_(c, g), W_(c, b), partial, B, HW, C, c->groups, eps, silu, c->stream, x2, C - Ca, Ca), "groupnorm");
  else CHECK_LAUNCH(c, ia2p_launch_groupnorm(x, C, y, C, W_(c, g), W_(c, b), partial, B, HW, C, c->groups, eps, silu, c->stream), "groupnorm");
}
void op_ln(RunCtx* c, const half_t* x, half_t* y, size_t g, size_t b, int M, int C) {
  ProfScope ps(c, PK_LN, 8.0 * M * C, 4.0 * M * C);
  CHECK_LAUNCH(c, ia2p_launch_layernorm(x, C, y, C, W_(c, g), W_(c, b), M, C, 1e-5f, c->stream), "layernorm");
}

struct Fwd {
  ia2p_ctx* c;
  int B, h, w, L;
  const half_t* ctxp;
  T2 temb_all;
  float* gn_partial;
  T2 kv_text, kv_ip;   // [B*Lt, kv_rows], [B*Li, kv_rows]
  const float* ip_scales = nullptr;   // device [B] or null: per-request IP-Adapter scale (else the context's one value)
  // GroupNorm statistics of live activation tensors (keyed by workspace offset): left by the producer's epilogue, read by the GroupNorm-fused convolution that consumes
  // the tensor -- possibly much later (the skips of the down path) --, released with the tensor
  std::unordered_map<size_t, GnStats> gst;
  bool gn_on = false;
  bool tune_like = false;      // the autotune pass (or the dry pass that sizes the workspace for it): GroupNorm launches, plus what tune_site needs to time the fused forms beside them
};
static void gst_put(Fwd& f, T2 t, const GnStats& s) { if (s.ok() && t.off != (size_t)-1) f.gst[t.off] = s; else if (s.buf.off != (size_t)-1) wsfree(f.c, s.buf); }
static GnStats gst_get(Fwd& f, T2 t) { auto it = f.gst.find(t.off); return it == f.gst.end() ? GnStats{} : it->second; }
static void act_free(Fwd& f, T2 t) {      // release an activation tensor and its statistics
  auto it = f.gst.find(t.off);
  if (it != f.gst.end()) { wsfree(f.c, it->second.buf); f.gst.erase(it); }
  wsfree(f.c, t);
}

// autotune pass: times a GroupNorm launch in place and leaves the figure for the tune_site call of the convolution behind it
struct TuneGnTimer {
  RunCtx* c; hipEvent_t e0{}, e1{}; bool on;
  explicit TuneGnTimer(RunCtx* c_) : c(c_), on(c_->tuning && !c_->dry && !c_->failed && c_->gn_fuse != 0) {
    if (on) { e0 = get_event(c); e1 = get_event(c); (void)hipEventRecord(e0, c->stream); }
  }
  void stop() {
    if (!on) return;
    float ms = 0.f;
    if (hipEventRecord(e1, c->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) c->tune_gn_ms = ms;
    c->evpool.push_back(e0); c->evpool.push_back(e1);
    on = false;
  }
};
struct RegionScope { RunCtx* c; int prev; RegionScope(RunCtx* c_, int r) : c(c_), prev(c_->region) { c->region = r; } ~RegionScope() { c->region = prev; } };

// x2 != null: the block input is [x (cx channels) | x2 (cin - cx channels)], never concatenated (up path: hidden state | skip) -- GroupNorm reads the two
// tensors, and the shortcut rides in conv2 as two appended K-blocks. Only with the fused shortcut (c->sc_fuse); the caller concatenates otherwise.
// may the GroupNorm in front of a stride-1 3x3 convolution of `cin` -> `cout` channels (+ cin2 + cin3 appended 1x1 channels) run inside that convolution? The shape part:
// the plan table gives the site a halo-staged tile; the statistics part: every source has its producer's column sums, at most IA2P_GN_MAX_SLOTS slots per image
static bool gn_conv_fusable(Fwd& f, int H, int Wd, int cin, int cout, int cin2, int cin3, const GnStats& s0, const GnStats* s1, int c0) {
  if (!f.gn_on || !s0.ok() || (s1 && !s1->ok())) return false;
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1; a.Hs = a.Ho = H; a.Ws = a.Wo = Wd; a.stride = 1; a.Cin = cin; a.M = f.B * H * Wd; a.N = cout; a.K = 9 * cin + cin2 + cin3; a.lda = c0; a.ldw = a.K; a.ldc = cout;
  a.Cin2 = cin2; a.Cin3 = cin3; a.lda2 = cin2; a.lda3 = cin3;
  if (cin2) a.A2 = (const half_t*)16;      // (shape query: any non-null pointer)
  if (cin3) a.A3 = (const half_t*)16;
  const GemmPlan pl = ia2p_gemm_plan(a.M, a.N, a.K, true, false);
  if (!pl.gn || !ia2p_conv_gn_fusable(a, pl.variant, pl.splitk)) return false;      // (the measured plan of the site says whether fusing pays there: tune_site)
  const int HW = H * Wd, groups = f.c->groups;
  if (cin % groups || cin / groups > 96 || c0 % 64 || (cin - c0) % 64 || cin > 3072) return false;
  return HW % s0.rows == 0 && HW / s0.rows <= IA2P_GN_MAX_SLOTS && (!s1 || (HW % s1->rows == 0 && HW / s1->rows <= IA2P_GN_MAX_SLOTS));
}

// statistics of tensor t ([B*HW, C]) for a consumer that is going to fuse: what its producer's epilogue left, or -- when the producer's tile could not (or left more than
// IA2P_GN_MAX_SLOTS slots per image) -- the canonical pass over the tensor, run once and kept with it (conv_in's output feeds two such consumers)
static GnStats gn_ensure_stats(Fwd& f, T2 t, int C, int HW) {
  GnStats s = gst_get(f, t);
  if (s.ok() && HW % s.rows == 0 && HW / s.rows <= IA2P_GN_MAX_SLOTS) return s;
  RunCtx* c = f.c;
  const int rows = gn_fallback_rows(HW);
  if (!rows || C % 8) return GnStats{};
  auto it = f.gst.find(t.off);
  if (it != f.gst.end()) { wsfree(c, it->second.buf); f.gst.erase(it); }
  s = GnStats{};
  s.buf = wsalloc(c, (size_t)(f.B * HW / rows) * C * 8);
  s.rows = rows;
  {
    RoleScope role(c, ROLE_GROUPNORM);
    ProfScope ps(c, PK_GN, 4.0 * f.B * HW * C, 2.0 * f.B * HW * C);
    CHECK_LAUNCH(c, ia2p_launch_gn_colstats(t.p, C, f.B * HW, C, rows, (double*)s.buf.p, c->stream), "groupnorm statistics");
  }
  f.gst[t.off] = s;
  return s;
}

// out_stats: the block's output feeds another GroupNorm-fusable convolution (the next ResnetBlock2D, or -- as a skip -- one of the up path): its statistics are taken
static T2 run_resnet(Fwd& f, const Resnet& r, T2 x, int H, int Wd, const T2* x2t = nullptr, int cx = 0, bool out_stats = false) {
  const half_t* x2 = x2t ? x2t->p : nullptr;
  const bool two = x2t != nullptr;
  ia2p_ctx* c = f.c;
  RegionScope rs(c, PR_CONV_BLOCK);
  const int HW = H * Wd, M = f.B * HW;
  // norm1 + SiLU + conv1: inside the convolution when the site is a halo-staged one and the producers left their statistics (conv_halo_kernel.h, GN = 1)
  // (first by shape alone -- does the plan give conv1 a halo-staged tile? -- then with the statistics, fetched or computed only when the site can use them)
  GnStats probe; probe.rows = gn_fallback_rows(HW);
  GnStats sx, sx2;
  bool fuse1 = probe.ok() && gn_conv_fusable(f, H, Wd, r.cin, r.cout, 0, 0, probe, two ? &probe : nullptr, two ? cx : r.cin);
  if (fuse1) {
    sx = gn_ensure_stats(f, x, two ? cx : r.cin, HW);
    if (two) sx2 = gn_ensure_stats(f, *x2t, r.cin - cx, HW);
    fuse1 = gn_conv_fusable(f, H, Wd, r.cin, r.cout, 0, 0, sx, two ? &sx2 : nullptr, two ? cx : r.cin);
  }
  GnWant w1{HW};
  const bool cat = r.shortcut && c->sc_fuse;      // conv2(h) + conv_shortcut(x) as ONE implicit GEMM (K = 9 cout + cin): no shortcut launch, no xs round trip
  // (conv1's statistics are wanted when conv2 can take them: its plan is a halo-staged one)
  const bool want1 = f.gn_on && probe.ok() && gn_conv_fusable(f, H, Wd, r.cout, r.cout, cat ? (two ? cx : r.cin) : 0, cat && two ? r.cin - cx : 0, probe, nullptr, r.cout);
  T2 hh = wsalloc(c, (size_t)M * r.cout);
  const bool twin = (c->gn_dry_mode >= 0 ? c->gn_dry_mode : c->gn_fuse) == 2;      // the fused path's unfused twin: the SAME statistics, normalised by a pass of its own, then the plain convolution
  auto apply_stats = [&](const half_t* a0, int c0, const half_t* a1, const GnStats& s0, const GnStats& s1, size_t gam, size_t bet, int C, half_t* y) {
    GemmArgs::GnIn g;
    memset(&g, 0, sizeof g);
    g.st0 = (const double*)s0.buf.p; g.rows0 = s0.rows; g.st1 = a1 ? (const double*)s1.buf.p : nullptr; g.rows1 = s1.rows; g.C0 = c0; g.gamma = W_(c, gam); g.beta = W_(c, bet);
    g.groups = c->groups; g.gs = C / c->groups; g.eps = c->cfg.norm_eps; g.silu = 1;
    RoleScope role(c, ROLE_GROUPNORM);
    ProfScope ps(c, PK_GN, 8.0 * M * C, 4.0 * M * C);
    CHECK_LAUNCH(c, ia2p_launch_gn_apply_stats(a0, c0, a1, C - c0, y, C, f.B, HW, C, g, c->stream), "groupnorm (apply from producer statistics)");
  };
  if (fuse1 && twin) {
    T2 n1 = wsalloc(c, (size_t)M * r.cin);
    apply_stats(x.p, two ? cx : r.cin, x2, sx, sx2, r.n1g, r.n1b, r.cin, n1.p);
    op_conv3(c, n1.p, f.B, H, Wd, r.cin, W_(c, r.w1), W_(c, r.b1), r.cout, 1, 0, c->dry ? nullptr : f.temb_all.p + r.temb_off, c->temb_total, nullptr, hh.p, 1, nullptr, 0, nullptr, 0, nullptr, want1 ? &w1 : nullptr);
    wsfree(c, n1);
  } else if (fuse1) {
    ConvGn g;
    g.fused = true; g.X1b = x2; g.C0 = two ? cx : r.cin; g.s0 = sx; g.s1 = sx2; g.gamma = W_(c, r.n1g); g.beta = W_(c, r.n1b); g.eps = c->cfg.norm_eps; g.groups = c->groups;
    op_conv3(c, x.p, f.B, H, Wd, r.cin, W_(c, r.w1), W_(c, r.b1), r.cout, 1, 0, c->dry ? nullptr : f.temb_all.p + r.temb_off, c->temb_total, nullptr, hh.p, 1, nullptr, 0, nullptr, 0, &g, want1 ? &w1 : nullptr);
  } else {
    T2 n1 = wsalloc(c, (size_t)M * r.cin);
    // autotune pass: the site is measured both ways -- this GroupNorm launch + the best plain plan against the fused launch on the raw tensor(s) with their statistics
    const bool tune = f.tune_like && probe.ok() && H % 16 == 0 && Wd % 16 == 0;
    ConvGn tg1;
    if (tune) {
      tg1.tune = true; tg1.Xraw = x.p; tg1.X1b = x2; tg1.C0 = two ? cx : r.cin; tg1.gamma = W_(c, r.n1g); tg1.beta = W_(c, r.n1b); tg1.eps = c->cfg.norm_eps; tg1.groups = c->groups;
      tg1.s0 = gn_ensure_stats(f, x, tg1.C0, HW);
      if (two) tg1.s1 = gn_ensure_stats(f, *x2t, r.cin - cx, HW);
    }
    TuneGnTimer tg(c);
    op_gn(c, x.p, n1.p, r.n1g, r.n1b, f.B, HW, r.cin, c->cfg.norm_eps, 1, f.gn_partial, x2, cx);
    tg.stop();
    op_conv3(c, n1.p, f.B, H, Wd, r.cin, W_(c, r.w1), W_(c, r.b1), r.cout, 1, 0, c->dry ? nullptr : f.temb_all.p + r.temb_off, c->temb_total, nullptr, hh.p, 1, nullptr, 0, nullptr, 0, tune ? &tg1 : nullptr,
             want1 ? &w1 : nullptr);
    wsfree(c, n1);
  }
  const GnStats sh = w1.out;
  const bool fuse2 = want1 && gn_conv_fusable(f, H, Wd, r.cout, r.cout, cat ? (two ? cx : r.cin) : 0, cat && two ? r.cin - cx : 0, sh, nullptr, r.cout);
  ConvGn g2;
  g2.fused = fuse2 && !twin; g2.C0 = r.cout; g2.s0 = sh; g2.gamma = W_(c, r.n2g); g2.beta = W_(c, r.n2b); g2.eps = c->cfg.norm_eps; g2.groups = c->groups;
  T2 n2{(size_t)-1, nullptr};
  const bool tune2 = f.tune_like && probe.ok() && H % 16 == 0 && Wd % 16 == 0;      // (autotune pass: conv2 measured both ways, as conv1 above; hh stays alive for it)
  if (tune2) {
    g2.tune = true; g2.Xraw = hh.p;
    g2.s0 = gn_ensure_stats(f, hh, r.cout, HW);
  }
  if (!fuse2 || twin) {
    n2 = wsalloc(c, (size_t)M * r.cout);
    if (fuse2) apply_stats(hh.p, r.cout, nullptr, sh, GnStats{}, r.n2g, r.n2b, r.cout, n2.p);
    else {
      TuneGnTimer tg(c);
      op_gn(c, hh.p, n2.p, r.n2g, r.n2b, f.B, HW, r.cout, c->cfg.norm_eps, 1, f.gn_partial);
      tg.stop();
    }
    if (!tune2) wsfree(c, hh);
  }
  const half_t* in2 = g2.fused ? hh.p : n2.p;
  GnWant w2{HW};
  GnWant* gw2 = f.gn_on && out_stats && ia2p_plan_any_gn(M) ? &w2 : nullptr;      // (only when some 3x3 site of this resolution level fuses its GroupNorm under the measured plans)
  T2 xs{(size_t)-1, nullptr};
  const half_t* resid = x.p;
  if (r.shortcut && !cat) {
    xs = wsalloc(c, (size_t)M * r.cout);
    { RoleScope role(c, ROLE_CONV3X3); op_gemm(c, x.p, r.cin, W_(c, r.wsc), W_(c, r.bsc), nullptr, 0, xs.p, r.cout, M, r.cout, r.cin); }
    resid = xs.p;
  }
  T2 out = wsalloc(c, (size_t)M * r.cout);
  if (cat && two) op_conv3(c, in2, f.B, H, Wd, r.cout, W_(c, r.wcat), W_(c, r.bcat), r.cout, 1, 0, nullptr, 0, nullptr, out.p, 1, x.p, cx, x2, r.cin - cx, &g2, gw2);
  else if (cat) op_conv3(c, in2, f.B, H, Wd, r.cout, W_(c, r.wcat), W_(c, r.bcat), r.cout, 1, 0, nullptr, 0, nullptr, out.p, 1, x.p, r.cin, nullptr, 0, &g2, gw2);
  else op_conv3(c, in2, f.B, H, Wd, r.cout, W_(c, r.w2), W_(c, r.b2), r.cout, 1, 0, nullptr, 0, c->dry ? nullptr : resid, out.p, 1, nullptr, 0, nullptr, 0, &g2, gw2);
  if (g2.fused) wsfree(c, hh); else { wsfree(c, n2); if (tune2) act_free(f, hh); }
  if (sh.buf.off != (size_t)-1) wsfree(c, sh.buf);
  if (r.shortcut && !cat) wsfree(c, xs);
  if (gw2) gst_put(f, out, w2.out);
  return out;
}

// QKV projection (LayerNorm folded) + the self-attention that consumes it in ONE launch (qxattn.hip): Q, K, V never leave the CU. x: O / ldo / B / heads / Nq = 256
static void op_qkv_sattn(RunCtx* c, const half_t* A, int lda, const half_t* W, const LnIn* ln, int M, int C, const AttnArgs& x) {
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  if (ln) { a.ln_stats = ln->stats; a.ln_slots = ln->slots; a.ln_cs = ln->cs; a.ln_bias = ln->lb; a.ln_eps = ln->eps; }
  a.A = A; a.W = W; a.zero = zero_page(); a.M = M; a.N = 3 * C; a.K = C; a.ldw = C; a.lda = lda; a.ldc = 3 * C;
  a.rows_per_batch = 1;
  set_prefetch(c, a, W, (size_t)3 * C * C * sizeof(half_t));
  RoleScope role(c, ROLE_QKV_SATTN);
#ifdef IA2P_CLOCK_STAMP
  if (c->stamp_buf && c->role == c->stamp_role && !c->dry && !c->tuning && c->stamp_n < c->stamp_cap && x.B * x.heads <= RunCtx::STAMP_WG) {
    a.partial = (float*)(c->stamp_buf + (size_t)c->stamp_n * RunCtx::STAMP_WG * 8);
    c->stamp_meta.push_back({a.M, a.N, a.K, -4, x.B * x.heads});      // (variant -4: the fused QKV + self-attention tile)
    ++c->stamp_n;
  }
#endif
  ProfScope ps(c, PK_QKVATTN, 2.0 * M * 3.0 * C * C + 4.0 * x.B * x.heads * (double)x.Nq * x.Nq * 64, 2.0 * ((double)M * C + 3.0 * C * C + (double)M * C));
  ps.pf = a.pf ? (double)a.pf_bytes : 0.0;
  CHECK_LAUNCH(c, ia2p_launch_qkv_sattn(a, x, c->stream), "qkv projection + self-attention");
}

static void op_attn(RunCtx* c, const AttnArgs& a) {
  double keys = 0;
  for (int s = 0; s < a.nseg; ++s) keys += a.seg[s].nkeys;
  ProfScope ps(c, PK_ATTN, 4.0 * a.B * a.heads * (double)a.Nq * keys * 64, 2.0 * ((double)a.B * a.Nq * a.heads * 64 * 2 + 2.0 * a.B * keys * a.heads * 64));
  CHECK_LAUNCH(c, ia2p_launch_attention(a, c->stream), "attention");
}

// to_q projection + the cross-attention that consumes it in ONE launch (qxattn.hip); Q never leaves the CU
static void op_qxattn(RunCtx* c, const half_t* A, int lda, const half_t* W, const LnIn* ln, int M, int N, int K, const AttnArgs& x) {
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  if (ln) { a.ln_stats = ln->stats; a.ln_slots = ln->slots; a.ln_cs = ln->cs; a.ln_bias = ln->lb; a.ln_eps = ln->eps; }
  a.A = A; a.W = W; a.zero = zero_page(); a.M = M; a.N = N; a.K = K; a.ldw = K; a.lda = lda; a.ldc = N;
  a.rows_per_batch = 1;
  a.m_fastest = M <= N ? 1 : 0;
  a.acc_scale = a.bias_scale = 1.f;
  set_prefetch(c, a, W, (size_t)N * K * sizeof(half_t));
  double keys = 0;
  for (int s = 0; s < x.nseg; ++s) keys += x.seg[s].nkeys;
  RoleScope role(c, ROLE_Q_XATTN);
#ifdef IA2P_CLOCK_STAMP
  if (c->stamp_buf && c->role == c->stamp_role && !c->dry && !c->tuning && c->stamp_n < c->stamp_cap && (M / 128) * (N / 64) <= RunCtx::STAMP_WG) {
    a.partial = (float*)(c->stamp_buf + (size_t)c->stamp_n * RunCtx::STAMP_WG * 8);
    c->stamp_meta.push_back({a.M, a.N, a.K, -5, (M / 128) * (N / 64)});      // (variant -5: the fused to_q + cross-attention tile, 128 x 64)
    ++c->stamp_n;
  }
#endif
  ProfScope ps(c, PK_QXATTN, 2.0 * M * N * K + 4.0 * x.B * x.heads * (double)x.Nq * keys * 64,
               2.0 * ((double)M * K + (double)N * K + (double)M * N + 2.0 * x.B * keys * x.heads * 64));
  CHECK_LAUNCH(c, ia2p_launch_qproj_xattn(a, x, c->stream), "to_q + cross-attention");
}

static T2 run_transformer(Fwd& f, const Transformer& t, T2 x, int H, int Wd, bool out_stats = false) {
  ia2p_ctx* c = f.c;
  RegionScope rs(c, PR_TRANSFORMER);
  const int HW = H * Wd, M = f.B * HW, C = t.c;
  const int ctxd = c->cfg.cross_attention_dim;
  const int Lt = c->ip_enabled ? f.L - c->ip_tokens : f.L;
  const int Li = c->ip_enabled ? c->ip_tokens : 0;
  const float sl2e = 0.125f * 1.4426950408889634f;
  T2 n = wsalloc(c, (size_t)M * C);
  op_gn(c, x.p, n.p, t.ng, t.nb, f.B, HW, C, 1e-6f, 0, f.gn_partial);
  // The three LayerNorms of a block never run as kernels: every GEMM that writes the token stream `tk` also emits per-row
  // {sum, sum of squares} partials of its fp16 output (`st`), and the GEMM that consumes LN(tk) reads raw `tk` against the
  // gamma-folded weights and finishes the normalisation in its epilogue (LnIn; GemmArgs.ln_* in common.h).
  T2 tk = wsalloc(c, (size_t)M * C);
  T2 stt = wsalloc(c, (size_t)M * ((C + 63) / 64) * 2 * 2);                // float2 per row and slot (one slot per tile column, tiles >= 64 wide)
  float* st = (float*)stt.p;
  int slots = 0;
  const float eps = 1e-5f;
  auto F_ = [&](size_t off) { return (const float*)(c->arena + off); };
  const bool fold = c->ln_fold;
  if (!fold) st = nullptr;
  { RoleScope role(c, ROLE_PROJ_IO); op_gemm(c, n.p, C, W_(c, t.win), W_(c, t.bin), nullptr, 0, tk.p, C, M, C, C, 0, 0, 0, 0, 0, nullptr, st, &slots); }
  wsfree(c, n);
  T2 lnb = fold ? T2{(size_t)-1, nullptr} : wsalloc(c, (size_t)M * C);
  T2 qkv = wsalloc(c, (size_t)M * 3 * C), att = wsalloc(c, (size_t)M * C);
  T2 ff = wsalloc(c, (size_t)M * 4 * C);
  const int ldkv = c->kv_rows;
  for (const TBlock& b : t.blocks) {
    // self-attention (AttnProcessor2_0, reference attention_processor.py:205-279)
    // 256 tokens per image (the 16 x 16 level): the QKV tile of one image x one head holds everything that head's attention needs -- projection and attention
    // as ONE launch when there are enough (image, head) pairs to fill the chip (same threshold and switch as the fused cross-attention)
    {
    RoleScope role_sa(c, ROLE_QKV_SATTN);
#ifdef IA2P_NO_SATTN_FUSE      // A/B builds: projection and self-attention as two launches everywhere
    bool fuse_sa = false;
#else
    bool fuse_sa = fold && c->sattn_fuse && HW == 256 && C == t.heads * 64 && (long)f.B * t.heads >= c->xattn_min_tiles;
#endif
    AttnArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.O = att.p; sa.ldo = C; sa.B = f.B; sa.heads = t.heads; sa.Nq = HW; sa.nseg = 1; sa.scale_log2e = sl2e;
    sa.seg[0].nkeys = HW; sa.seg[0].weight = 1.f;
    if (fuse_sa) {      // a site the fused tile does not take (alignment of O / the folded constants, the 31-bit operand limit) runs projection + attention as two launches
      GemmArgs g;       // (workspace and arena offsets are 256-byte aligned: the dry pass, with null pointers, decides the same way)
      memset(&g, 0, sizeof g);
      g.A = tk.p; g.W = W_(c, b.fqkv); g.M = M; g.N = 3 * C; g.K = C; g.lda = C; g.ldw = C; g.ldc = 3 * C;
      g.ln_stats = st; g.ln_slots = slots; g.ln_cs = F_(b.cs1); g.ln_bias = F_(b.lb1);
      fuse_sa = ia2p_qkv_sattn_ok(g, sa);
    }
    if (fuse_sa) {
      const LnIn ln{st, slots, F_(b.cs1), F_(b.lb1), eps};
      op_qkv_sattn(c, tk.p, C, W_(c, b.fqkv), &ln, M, C, sa);
    } else if (fold) {
      const LnIn ln{st, slots, F_(b.cs1), F_(b.lb1), eps};
      op_gemm(c, tk.p, C, W_(c, b.fqkv), nullptr, nullptr, 0, qkv.p, 3 * C, M, 3 * C, C, 0, 0, 0, 0, 0, &ln);
    } else {
      op_ln(c, tk.p, lnb.p, b.ln1g, b.ln1b, M, C);
      op_gemm(c, lnb.p, C, W_(c, b.wqkv), nullptr, nullptr, 0, qkv.p, 3 * C, M, 3 * C, C);
    }
    if (!fuse_sa) {
      AttnArgs a;
      memset(&a, 0, sizeof a);
      a.Q = qkv.p; a.ldq = 3 * C; a.O = att.p; a.ldo = C; a.B = f.B; a.heads = t.heads; a.Nq = HW; a.nseg = 1; a.scale_log2e = sl2e;
      a.seg[0].K = c->dry ? nullptr : qkv.p + C; a.seg[0].V = c->dry ? nullptr : qkv.p + 2 * C;
      a.seg[0].nkeys = HW; a.seg[0].ld = 3 * C; a.seg[0].rows_per_batch = HW; a.seg[0].weight = 1.f;
      op_attn(c, a);
    }
    }
    { RoleScope role(c, ROLE_ATTN_OUT); op_gemm(c, att.p, C, W_(c, b.wo1), W_(c, b.bo1), tk.p, C, tk.p, C, M, C, C, 0, 0, 0, 0, 0, nullptr, st, &slots); }
    // cross-attention (IPAttnProcessor2_0 :310-412 when the adapter is installed, else AttnProcessor2_0)
    {
      RoleScope role(c, ROLE_Q_XATTN);
      AttnArgs a;
      memset(&a, 0, sizeof a);
      a.Q = qkv.p; a.ldq = C; a.O = att.p; a.ldo = C; a.B = f.B; a.heads = t.heads; a.Nq = HW; a.nseg = Li ? 2 : 1; a.scale_log2e = sl2e;
      const half_t* kt = c->dry ? nullptr : f.kv_text.p + b.kv_col;
      const half_t* ki = (c->dry || !Li) ? nullptr : f.kv_ip.p + b.kv_col;
      a.seg[0].K = kt; a.seg[0].V = c->dry ? nullptr : kt + C; a.seg[0].nkeys = Lt; a.seg[0].ld = ldkv; a.seg[0].rows_per_batch = Lt; a.seg[0].weight = 1.f;
      a.seg[1].K = ki; a.seg[1].V = ki ? ki + C : nullptr; a.seg[1].nkeys = Li; a.seg[1].ld = ldkv; a.seg[1].rows_per_batch = Li; a.seg[1].weight = c->ip_scale;
      a.w1_b = Li ? f.ip_scales : nullptr;
      // a 128 x 64 tile of to_q is 128 queries x one head: projection and attention run as one launch when tiles do not straddle batch elements
      // (and the context fits 3 key tiles: its K / V ride in registers through the projection loop); small problems keep the finer 64 x 64 split
      const bool fuse = c->xattn_fuse && HW % 128 == 0 && C == t.heads * 64 && (Lt + 63) / 64 + (Li + 63) / 64 <= 3 && (long)(M / 128) * t.heads >= c->xattn_min_tiles;
      if (fold) {
        const LnIn ln{st, slots, F_(b.cs2), F_(b.lb2), eps};
        if (fuse) op_qxattn(c, tk.p, C, W_(c, b.fq2), &ln, M, C, C, a);
        else op_gemm(c, tk.p, C, W_(c, b.fq2), nullptr, nullptr, 0, qkv.p, C, M, C, C, 0, 0, 0, 0, 0, &ln);
      } else {
        op_ln(c, tk.p, lnb.p, b.ln2g, b.ln2b, M, C);
        if (fuse) op_qxattn(c, lnb.p, C, W_(c, b.wq2), nullptr, M, C, C, a);
        else op_gemm(c, lnb.p, C, W_(c, b.wq2), nullptr, nullptr, 0, qkv.p, C, M, C, C);
      }
      if (!fuse) op_attn(c, a);
    }
    { RoleScope role(c, ROLE_ATTN_OUT); op_gemm(c, att.p, C, W_(c, b.wo2), W_(c, b.bo2), tk.p, C, tk.p, C, M, C, C, 0, 0, 0, 0, 0, nullptr, st, &slots); }
    // GEGLU feed-forward
    if (!fold) op_ln(c, tk.p, lnb.p, b.ln3g, b.ln3b, M, C);
    {
      const LnIn ln{st, slots, F_(b.cs3), F_(b.lb3), eps};
      GemmArgs g1 = fold ? gemm_args(c, tk.p, C, W_(c, b.fff1), W_(c, b.bff1), nullptr, 0, ff.p, 4 * C, M, 8 * C, C, 1, 0, 0, 0, 0, &ln, nullptr, 0)
                         : gemm_args(c, lnb.p, C, W_(c, b.wff1), W_(c, b.bff1), nullptr, 0, ff.p, 4 * C, M, 8 * C, C, 1, 0, 0, 0, 0, nullptr, nullptr, 0);
      GemmArgs g2 = gemm_args(c, ff.p, 4 * C, W_(c, b.wff2), W_(c, b.bff2), tk.p, C, tk.p, C, M, C, 4 * C, 0, 0, 0, 0, 0, nullptr, st, 0);
      run_ffn(c, g1, g2, &slots);
    }
  }
  wsfree(c, stt); wsfree(c, lnb); wsfree(c, qkv); wsfree(c, att); wsfree(c, ff);
  (void)ctxd;
  T2 out = wsalloc(c, (size_t)M * C);
  GnWant gw{HW};
  const bool want = f.gn_on && out_stats && ia2p_plan_any_gn(M);
  { RoleScope role(c, ROLE_PROJ_IO); op_gemm(c, tk.p, C, W_(c, t.wout), W_(c, t.bout), x.p, C, out.p, C, M, C, C, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, 0, want ? &gw : nullptr); }
  wsfree(c, tk);
  if (want) gst_put(f, out, gw.out);
  return out;
}

// context K/V of every cross-attention layer in one GEMM each (text rows / image-token rows of ctx); per layer: reference
// attention_processor.py:358-359 (to_k/to_v) and :379-380 (to_k_ip/to_v_ip). kv_text: [B*Lt, kv_rows], kv_ip: [B*Li, kv_rows].
static void project_context(ia2p_ctx* c, const half_t* context, int L, int B, half_t* kv_text, half_t* kv_ip) {
  RegionScope rs(c, PR_TRANSFORMER);
  RoleScope role(c, ROLE_CTX_KV);
  const int ctxd = c->cfg.cross_attention_dim;
  const int Lt = c->ip_enabled ? L - c->ip_tokens : L, Li = c->ip_enabled ? c->ip_tokens : 0;
  op_gemm(c, context, ctxd, W_(c, c->kv_text_base), nullptr, nullptr, 0, kv_text, c->kv_rows, B * Lt, c->kv_rows, ctxd, 0, Lt, L, 0);
  if (Li) op_gemm(c, context, ctxd, W_(c, c->kv_ip_base), nullptr, nullptr, 0, kv_ip, c->kv_rows, B * Li, c->kv_rows, ctxd, 0, Li, L, Lt);
}

// kv_cached != nullptr: the context projections were computed before (ia2p_project_context) and are read from there
static ia2p_status run_forward(ia2p_ctx* c, const half_t* sample, float timestep, const half_t* context, int L,
                               const half_t* text_embeds, const half_t* time_ids, half_t* out, int B, int h, int w,
                               const half_t* kv_cached = nullptr, const float* timesteps = nullptr, const float* ip_scales = nullptr) {
  const ia2p_unet_config& g = c->cfg;
  const int n = g.n_blocks;
  const int T = g.time_embed_dim, Tp = g.time_proj_dim, Ain = g.projection_class_embeddings_input_dim, Ad = g.addition_time_embed_dim;
  const int pooled = Ain - g.num_time_ids * Ad;
  Fwd f{c, B, h, w, L, context, T2{(size_t)-1, nullptr}, nullptr, T2{(size_t)-1, nullptr}, T2{(size_t)-1, nullptr}};
  f.ip_scales = ip_scales;
  const int gn_mode = c->gn_dry_mode >= 0 ? c->gn_dry_mode : c->gn_fuse;
  f.tune_like = c->gn_fuse != 0 && (c->dry ? c->gn_dry_mode == 3 || (c->gn_dry_mode < 0 && c->tuning) : c->tuning);
  f.gn_on = gn_mode != 0 && gn_mode != 3 && !c->tuning;      // (the autotune pass measures the plain kernels: GroupNorm launches there)

  // GroupNorm partial sums (fp32) live at the front of the workspace
  T2 gnp = wsalloc(c, (size_t)B * 64 * g.norm_num_groups * 2 * 2);
  f.gn_partial = (float*)gnp.p;
  // ---- embeddings
  T2 tsin = wsalloc(c, (size_t)B * Tp), addin = wsalloc(c, (size_t)B * Ain), e1 = wsalloc(c, (size_t)B * T), emb0 = wsalloc(c, (size_t)B * T);
  T2 a1 = wsalloc(c, (size_t)B * T), emb = wsalloc(c, (size_t)B * T);
  f.temb_all = wsalloc(c, (size_t)B * c->temb_total);
  {
    RoleScope role(c, ROLE_EMBED);
    ProfScope ps(c, PK_EMBED, 0, 0);
    CHECK_LAUNCH(c, ia2p_launch_embed(timestep, timesteps, text_embeds, time_ids, tsin.p, addin.p, B, Tp, pooled, Ad, g.num_time_ids, c->stream), "embed");
    // skinny linears hold <= 16 rows per launch: larger batches go in row chunks
    auto lin = [&](const half_t* X, int ldx, size_t w, size_t b, const half_t* add, int ldadd, half_t* o, int ldo, int N, int K, int si, int so, const char* what) {
      for (int r0 = 0; r0 < B; r0 += 16) {
        const int rows = std::min(16, B - r0);
        CHECK_LAUNCH(c, ia2p_launch_linear_small(c->dry ? nullptr : X + (size_t)r0 * ldx, ldx, W_(c, w), W_(c, b), (c->dry || !add) ? nullptr : add + (size_t)r0 * ldadd, ldadd,
                                                 c->dry ? nullptr : o + (size_t)r0 * ldo, ldo, rows, N, K, si, so, c->stream), what);
      }
    };
    lin(tsin.p, Tp, c->te1w, c->te1b, nullptr, 0, e1.p, T, T, Tp, 0, 1, "time_embedding.linear_1");
    lin(e1.p, T, c->te2w, c->te2b, nullptr, 0, emb0.p, T, T, T, 0, 0, "time_embedding.linear_2");
    lin(addin.p, Ain, c->ae1w, c->ae1b, nullptr, 0, a1.p, T, T, Ain, 0, 1, "add_embedding.linear_1");
    // `emb` is only ever consumed through SiLU (every ResnetBlock2D: time_emb_proj(nonlinearity(temb))), so the last embedding linear stores
    // SiLU(emb) -- applied to the fp32 sum, rounded once -- and the stacked projection reads it as is: the activation used to be recomputed inside that
    // kernel by every one of its 3 440 output-column waves (64 SiLUs per wave-iteration against 256 FMAs)
    lin(a1.p, T, c->ae2w, c->ae2b, emb0.p, T, emb.p, T, T, T, 0, 1, "add_embedding.linear_2 (+ SiLU)");
    lin(emb.p, T, c->tw_all, c->tb_all, nullptr, 0, f.temb_all.p, c->temb_total, c->temb_total, T, 0, 0, "time_emb_proj (stacked)");
  }
  wsfree(c, tsin); wsfree(c, addin); wsfree(c, e1); wsfree(c, emb0); wsfree(c, a1); wsfree(c, emb);

  // ---- context K/V for every cross-attention layer in one GEMM each (text rows / image-token rows of ctx);
  //      per layer: reference attention_processor.py:358-359 (to_k/to_v) and :379-380 (to_k_ip/to_v_ip)
  if (c->kv_rows > 0) {
    const int Lt = c->ip_enabled ? L - c->ip_tokens : L, Li = c->ip_enabled ? c->ip_tokens : 0;
    if (kv_cached) {
      f.kv_text.p = const_cast<half_t*>(kv_cached);
      if (Li) f.kv_ip.p = const_cast<half_t*>(kv_cached) + (size_t)B * Lt * c->kv_rows;
    } else {
      f.kv_text = wsalloc(c, (size_t)B * Lt * c->kv_rows);
      if (Li) f.kv_ip = wsalloc(c, (size_t)B * Li * c->kv_rows);
      // (On a low-priority side stream beside the start of the step -- round 3, docs/LOG.md -- the projection cost +0.7 ms per step: the work is conserved, the
      //  interleaving costs. The step belongs on ONE queue; the pipelines hoist the projection out of the loop anyway.)
      project_context(c, context, L, B, f.kv_text.p, f.kv_ip.p);
    }
  }

  // ---- down path
  RegionScope rs_conv(c, PR_CONV_BLOCK);      // from here on everything outside run_transformer belongs to the conv blocks
  int H = h, Wd = w;
  std::vector<T2> skips;
  std::vector<int> skip_c;
  T2 x = wsalloc(c, (size_t)B * H * Wd * g.block_out_channels[0]);
  {
    RoleScope role(c, ROLE_CONV_IO);
    ProfScope ps(c, PK_CONV_IN, 2.0 * B * H * Wd * 9.0 * g.in_channels * g.block_out_channels[0],
                 2.0 * ((double)B * H * Wd * (g.in_channels + g.block_out_channels[0]) + 64.0 * g.block_out_channels[0]));
    CHECK_LAUNCH(c, ia2p_launch_conv_in(sample, W_(c, c->conv_in_w), W_(c, c->conv_in_b), x.p, B, g.in_channels, H, Wd, g.block_out_channels[0], c->stream), "conv_in");
  }
  // Every tensor below feeds a GroupNorm in front of a 3x3 convolution -- the next ResnetBlock2D's norm1, or (the skips) the norm1 of an up-path block much later --
  // and carries its producer's column sums with it (f.gst); conv_in is a direct kernel: the consumer that fuses runs the canonical statistics pass over its output
  skips.push_back(x); skip_c.push_back(g.block_out_channels[0]);
  for (int i = 0; i < n; ++i) {
    const Stage& st = c->down[i];
    for (size_t j = 0; j < st.res.size(); ++j) {
      const bool att = !st.att.empty();
      T2 r = run_resnet(f, st.res[j], x, H, Wd, nullptr, 0, !att);      // (with a transformer behind it the block's output only feeds that transformer's own GroupNorm)
      if (att) { T2 t = run_transformer(f, st.att[j], r, H, Wd, true); act_free(f, r); r = t; }
      x = r;
      skips.push_back(x); skip_c.push_back(st.res[j].cout);
    }
    if (st.resample) {
      const int Ho = (H - 1) / 2 + 1, Wo = (Wd - 1) / 2 + 1;
      T2 d = wsalloc(c, (size_t)B * Ho * Wo * st.rc);
      GnWant gw{Ho * Wo};
      const bool want = f.gn_on && ia2p_plan_any_gn(B * Ho * Wo);
      op_conv3(c, x.p, B, H, Wd, st.rc, W_(c, st.rw), W_(c, st.rb), st.rc, 2, 0, nullptr, 0, nullptr, d.p, 1, nullptr, 0, nullptr, 0, nullptr, want ? &gw : nullptr);
      if (want) gst_put(f, d, gw.out);
      H = Ho; Wd = Wo; x = d;
      skips.push_back(x); skip_c.push_back(st.rc);
    }
  }
  // ---- mid
  {
    T2 r0 = run_resnet(f, c->mid_r0, x, H, Wd);          // x stays alive: it is the top skip
    T2 t = run_transformer(f, c->mid_t, r0, H, Wd, true); act_free(f, r0);
    T2 r1 = run_resnet(f, c->mid_r1, t, H, Wd, nullptr, 0, true); act_free(f, t);
    x = r1;
  }
  // ---- up path
  for (int i = 0; i < n; ++i) {
    const Stage& st = c->up[i];
    for (size_t j = 0; j < st.res.size(); ++j) {
      T2 sk = skips.back(); skips.pop_back();
      const int cs = skip_c.back(); skip_c.pop_back();
      const int cx = st.res[j].cin - cs;
      const long M = (long)B * H * Wd;
      T2 r;
      const bool att = !st.att.empty();
      const bool last = i == n - 1 && j + 1 == st.res.size();      // (the last block's output feeds conv_norm_out: a GroupNorm launch of its own)
      if (c->sc_fuse && c->cat_free && st.res[j].shortcut && cx % 64 == 0 && cs % 64 == 0) {
        // torch.cat([hidden, skip]) never materialised: GroupNorm and the appended shortcut blocks of conv2 read the two tensors
        r = run_resnet(f, st.res[j], x, H, Wd, &sk, cx, !att && !last);
        act_free(f, x); act_free(f, sk);
      } else {
        T2 cat = wsalloc(c, (size_t)M * st.res[j].cin);
        {
          ProfScope ps(c, PK_CONCAT, 0, 4.0 * M * st.res[j].cin);
          CHECK_LAUNCH(c, ia2p_launch_concat(x.p, cx, cx, sk.p, cs, cs, cat.p, M, c->stream), "concat");
        }
        act_free(f, x); act_free(f, sk);
        r = run_resnet(f, st.res[j], cat, H, Wd, nullptr, 0, !att && !last);
        act_free(f, cat);      // (with whatever statistics a fusing consumer computed for it)
      }
      if (att) { T2 t = run_transformer(f, st.att[j], r, H, Wd, true); act_free(f, r); r = t; }
      x = r;
    }
    if (st.resample) {
      T2 u = wsalloc(c, (size_t)B * (2 * H) * (2 * Wd) * st.rc);
      GnWant gw{4 * H * Wd};
      const bool want = f.gn_on && ia2p_plan_any_gn(4 * B * H * Wd);
      op_conv3(c, x.p, B, H, Wd, st.rc, W_(c, st.rw), W_(c, st.rb), st.rc, 1, 1, nullptr, 0, nullptr, u.p, 1, nullptr, 0, nullptr, 0, nullptr, want ? &gw : nullptr);
      if (want) gst_put(f, u, gw.out);
      act_free(f, x);
      H *= 2; Wd *= 2; x = u;
    }
  }
  if (H != h || Wd != w) return fail(c, IA2P_ERR_SHAPE, "latent %dx%d does not survive the down/up path (needs divisibility by 2^%d)", h, w, n - 1);
  // ---- out
  const int c0 = g.block_out_channels[0];
  T2 no = wsalloc(c, (size_t)B * H * Wd * c0);
  op_gn(c, x.p, no.p, c->ngo, c->nbo, B, H * Wd, c0, g.norm_eps, 1, f.gn_partial);
  act_free(f, x);
  {
    RoleScope role(c, ROLE_CONV_IO);
    ProfScope ps(c, PK_CONV_OUT, 2.0 * B * H * Wd * 9.0 * c0 * g.out_channels, 2.0 * ((double)B * H * Wd * (c0 + g.out_channels) + 9.0 * c0 * g.out_channels));
    CHECK_LAUNCH(c, ia2p_launch_conv_out(no.p, c0, W_(c, c->conv_out_w), W_(c, c->conv_out_b), out, B, c0, H, Wd, g.out_channels, c->stream), "conv_out");
  }
  for (auto& kv : f.gst) wsfree(c, kv.second.buf);      // (none left on a complete pass)
  f.gst.clear();
  wsfree(c, no); wsfree(c, f.temb_all); wsfree(c, gnp); wsfree(c, f.kv_text); wsfree(c, f.kv_ip);
  return c->failed ? IA2P_ERR_HIP : IA2P_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------------------------
// ---- weight arena plumbing shared by the UNet, VAE and CLIP contexts
ia2p_status rc_bind_arena(RunCtx* c, void* dev, size_t bytes) {
  if (!c || !dev) return fail(c, IA2P_ERR_INVALID, "bind_arena: null argument");
  if (bytes < c->arena_elems * sizeof(half_t)) return fail(c, IA2P_ERR_NOMEM, "arena needs %zu bytes, got %zu", c->arena_elems * sizeof(half_t), bytes);
  if (((uintptr_t)dev) & 255) return fail(c, IA2P_ERR_INVALID, "arena must be 256-byte aligned");
  c->arena = (half_t*)dev;
  c->finalized = false;
  c->wseq_key = -1;
  return IA2P_OK;
}
ia2p_status rc_load_tensor(RunCtx* c, const char* key, const void* src, const int64_t* shape, int ndim, void* stream) {
  if (!c || !key || !src || !shape) return fail(c, IA2P_ERR_INVALID, "load_tensor: null argument");
  if (!c->arena) return fail(c, IA2P_ERR_STATE, "load_tensor before bind_arena");
  auto it = c->params.find(key);
  if (it == c->params.end()) return fail(c, IA2P_ERR_KEY, "unknown parameter key '%s'", key);
  Param& p = it->second;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  if (n != p.elems) return fail(c, IA2P_ERR_SHAPE, "parameter '%s': expected %zu elements, got %zu", key, p.elems, n);
  hipStream_t s = (hipStream_t)stream;
  half_t* dst = c->arena + p.off;
  hipError_t e = hipSuccess;
  switch (p.kind) {
    case PK_COPY: e = hipMemcpyAsync(dst, src, n * sizeof(half_t), hipMemcpyDeviceToDevice, s); break;
    case PK_CONV: e = ia2p_launch_pack_conv((const half_t*)src, dst, p.d0, p.d1, s); break;
    case PK_CONV_TAP: e = ia2p_launch_pack_conv((const half_t*)src, dst, p.d0, p.d1, s); break;
    case PK_GEGLU_W: case PK_GEGLU_B: e = ia2p_launch_pack_geglu((const half_t*)src, dst, p.d0, p.d1, s); break;
    case PK_PAD_CONV_IN: e = ia2p_launch_pack_conv_in((const half_t*)src, dst, p.d0, p.d1, s); break;
  }
  if (e != hipSuccess) return fail_hip(c, e, (std::string("load '") + key + "'").c_str());
  p.loaded = true;
  c->fold_dirty = true;     // data derived from the parameters at finalize (LayerNorm folds) is stale until the owner re-derives it
  return IA2P_OK;
}
ia2p_status rc_finalize(RunCtx* c, const char* what) {
  if (!c) return IA2P_ERR_INVALID;
  if (!c->arena) return fail(c, IA2P_ERR_STATE, "finalize before bind_arena");
  int missing = 0;
  std::string first;
  for (auto& kv : c->params)
    if (!kv.second.loaded && !kv.second.optional) { if (!missing) first = kv.first; ++missing; }
  if (missing) return fail(c, IA2P_ERR_KEY, "%d %s parameters not loaded (e.g. '%s')", missing, what, first.c_str());
  c->finalized = true;
  return IA2P_OK;
}
ia2p_status rc_adopt(RunCtx* c, bool with_optional) {
  if (!c || !c->arena) return fail(c, IA2P_ERR_STATE, "adopt_arena before bind_arena");
  for (auto& kv : c->params)
    if (with_optional || !kv.second.optional) kv.second.loaded = true;
  c->finalized = true;
  return IA2P_OK;
}

extern "C" {

int ia2p_device_is_gfx950(void) {
  int dev = 0;
  hipDeviceProp_t p;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
  return strncmp(p.gcnArchName, "gfx950", 6) == 0;
}

ia2p_status ia2p_create(const ia2p_unet_config* cfg, ia2p_ctx** out) {
  if (!cfg || !out) return fail(nullptr, IA2P_ERR_INVALID, "ia2p_create: null argument");
  ia2p_ctx* c = new ia2p_ctx();
  c->cfg = *cfg;
  ia2p_status st = build_plan(c);
  if (st != IA2P_OK) { g_err = c->err; delete c; *out = nullptr; return st; }
  c->failed = false;
  c->groups = cfg->norm_num_groups;
  ia2p_sk_counters_invalidate();      // a fresh context never trusts tickets an earlier (possibly failed) one left behind
  *out = c;
  return IA2P_OK;
}

void ia2p_destroy(ia2p_ctx* c) { delete c; }

const char* ia2p_last_error(ia2p_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

size_t ia2p_arena_bytes(ia2p_ctx* c) { return c ? c->arena_elems * sizeof(half_t) : 0; }

ia2p_status ia2p_bind_arena(ia2p_ctx* c, void* dev, size_t bytes) { return rc_bind_arena(c, dev, bytes); }
ia2p_status ia2p_load_tensor(ia2p_ctx* c, const char* key, const void* src, const int64_t* shape, int ndim, void* stream) {
  return rc_load_tensor(c, key, src, shape, ndim, stream);
}
// LayerNorm folding pass over every transformer block (idempotent: reads the raw tensors, writes the folded copies)
static ia2p_status fold_all(ia2p_ctx* c, hipStream_t stream, bool sync) {
  std::vector<const Transformer*> ts;
  for (const Stage& s : c->down) for (const Transformer& t : s.att) ts.push_back(&t);
  ts.push_back(&c->mid_t);
  for (const Stage& s : c->up) for (const Transformer& t : s.att) ts.push_back(&t);
  hipError_t e = hipSuccess;
  auto H = [&](size_t off) { return c->arena + off; };
  auto F = [&](size_t off) { return (float*)(c->arena + off); };
  for (const Transformer* t : ts)
    for (const TBlock& b : t->blocks) {
      const int C = t->c;
      if (e == hipSuccess) e = ia2p_launch_fold_ln(H(b.wqkv), H(b.ln1g), H(b.ln1b), nullptr, H(b.fqkv), F(b.cs1), F(b.lb1), 3 * C, C, stream);
      if (e == hipSuccess) e = ia2p_launch_fold_ln(H(b.wq2), H(b.ln2g), H(b.ln2b), nullptr, H(b.fq2), F(b.cs2), F(b.lb2), C, C, stream);
      if (e == hipSuccess) e = ia2p_launch_fold_ln(H(b.wff1), H(b.ln3g), H(b.ln3b), H(b.bff1), H(b.fff1), F(b.cs3), F(b.lb3), 8 * C, C, stream);
    }
  // conv2 + conv_shortcut of a ResnetBlock2D as ONE implicit GEMM: weight rows concatenated along K, biases added
  std::vector<const Resnet*> rs;
  for (const Stage& s : c->down) for (const Resnet& r : s.res) rs.push_back(&r);
  rs.push_back(&c->mid_r0); rs.push_back(&c->mid_r1);
  for (const Stage& s : c->up) for (const Resnet& r : s.res) rs.push_back(&r);
  for (const Resnet* r : rs)
    if (r->shortcut && e == hipSuccess)
      e = ia2p_launch_cat_rows(H(r->w2), 9 * r->cout, H(r->wsc), r->cin, H(r->b2), H(r->bsc), H(r->wcat), H(r->bcat), r->cout, stream);
  if (e == hipSuccess && sync) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return fail_hip(c, e, "weight folding");
  c->fold_dirty = false;
  return IA2P_OK;
}
ia2p_status ia2p_finalize_weights(ia2p_ctx* c) {
  const ia2p_status st = rc_finalize(c, "UNet");
  return st == IA2P_OK ? fold_all(c, nullptr, true) : st;
}
size_t ia2p_arena_raw_bytes(ia2p_ctx* c) { return c ? c->arena_raw_elems * sizeof(half_t) : 0; }
// The head of the arena ([0, ia2p_arena_raw_bytes): parameters as loaded) was filled elsewhere -- an RCCL broadcast from the rank that read the
// checkpoint; the derived tail (LayerNorm folds) is recomputed here from it, so 2.5 GB of it never cross xGMI.
// The fold kernels must run AFTER the broadcast that filled the head: they are enqueued on the caller's stream (the one the collective was
// ordered on) and that stream is synchronised before returning, so forwards on any other stream afterwards see finished folds.
ia2p_status ia2p_adopt_arena_on(ia2p_ctx* c, int with_ip_adapter, void* stream) {
  const ia2p_status st = rc_adopt(c, with_ip_adapter != 0);
  return st == IA2P_OK ? fold_all(c, (hipStream_t)stream, true) : st;
}
ia2p_status ia2p_adopt_arena(ia2p_ctx* c, int with_ip_adapter) { return ia2p_adopt_arena_on(c, with_ip_adapter, nullptr); }

// ---- the one collective of the batch-data-parallel path, through the C ABI: RCCL broadcast of the arena head + local adoption -------------------------
// RCCL is bound late (dlopen / dlsym): the library has no link-time dependency on it, and a host that already carries an RCCL instance (PyTorch bundles its
// own librccl.so with SONAME librccl.so.1) must be served by THAT instance -- the communicator handed in was created by it. RTLD_NOLOAD first: whatever is
// already in the process; only a host without any RCCL gets the system one loaded for it.
namespace {
struct Rccl {
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};
const Rccl& rccl() {
  static const Rccl r = [] {
    Rccl x;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) return x;
    x.Broadcast = (decltype(x.Broadcast))dlsym(h, "ncclBroadcast");
    x.CommUserRank = (decltype(x.CommUserRank))dlsym(h, "ncclCommUserRank");
    x.CommCount = (decltype(x.CommCount))dlsym(h, "ncclCommCount");
    x.GetErrorString = (decltype(x.GetErrorString))dlsym(h, "ncclGetErrorString");
    x.ok = x.Broadcast && x.CommUserRank && x.CommCount && x.GetErrorString;
    return x;
  }();
  return r;
}
}  // namespace
int ia2p_rccl_available(void) { return rccl().ok ? 1 : 0; }
ia2p_status ia2p_bcast_arena(ia2p_ctx* c, void* rccl_comm, int root, int with_ip_adapter, void* stream) {
  if (!c || !rccl_comm) return fail(c, IA2P_ERR_INVALID, "bcast_arena: null argument");
  if (!c->arena) return fail(c, IA2P_ERR_STATE, "bcast_arena before bind_arena");
  const Rccl& r = rccl();
  if (!r.ok) return fail(c, IA2P_ERR_STATE, "bcast_arena: no RCCL in this process and none could be loaded (librccl.so.1)");
  ncclComm_t comm = (ncclComm_t)rccl_comm;
  int rank = -1, n = 0;
  ncclResult_t e = r.CommUserRank(comm, &rank);
  if (e == ncclSuccess) e = r.CommCount(comm, &n);
  if (e != ncclSuccess) return fail(c, IA2P_ERR_INVALID, "bcast_arena: not a usable communicator (%s)", r.GetErrorString(e));
  if (root < 0 || root >= n) return fail(c, IA2P_ERR_INVALID, "bcast_arena: root %d outside a communicator of %d ranks", root, n);
  // a few large messages (ring broadcast over point-to-point xGMI is per-link bound: message COUNT is what to keep small), each below 2^31 bytes
  char* p = (char*)c->arena;
  const size_t total = c->arena_raw_elems * sizeof(half_t), piece = (size_t)1 << 30;
  for (size_t lo = 0; lo < total && e == ncclSuccess; lo += piece)
    e = r.Broadcast(p + lo, p + lo, std::min(piece, total - lo), ncclInt8, root, comm, (hipStream_t)stream);
  if (e != ncclSuccess) { ia2p_sk_counters_invalidate(); return fail(c, IA2P_ERR_HIP, "bcast_arena: ncclBroadcast: %s", r.GetErrorString(e)); }
  if (rank == root) {                  // the root keeps its own tail; its stream is synchronised like the receivers' (the call returns with the head sent)
    const hipError_t he = hipStreamSynchronize((hipStream_t)stream);
    return he == hipSuccess ? IA2P_OK : fail_hip(c, he, "bcast_arena");
  }
  return ia2p_adopt_arena_on(c, with_ip_adapter, stream);      // fold kernels ordered behind the broadcast on the same stream, which is synchronised before returning
}

ia2p_status ia2p_set_gn_fuse(ia2p_ctx* c, int mode) {
  if (!c || mode < 0 || mode > 2) return fail(c, IA2P_ERR_INVALID, "set_gn_fuse: mode %d (0 GroupNorm launches, 1 fused into the convolutions, 2 the fused path's unfused twin)", mode);
  c->gn_fuse = mode;
  c->wseq_key = -1;
  return IA2P_OK;
}
ia2p_status ia2p_set_ip_adapter(ia2p_ctx* c, int enabled, int num_tokens, float scale) {
  if (!c) return IA2P_ERR_INVALID;
  if (enabled) {
    if (num_tokens < 1 || num_tokens > 64) return fail(c, IA2P_ERR_INVALID, "num_tokens %d out of range", num_tokens);
    for (auto& kv : c->params)
      if (kv.second.optional && !kv.second.loaded) return fail(c, IA2P_ERR_KEY, "IP-Adapter enabled but '%s' was never loaded", kv.first.c_str());
  }
  c->ip_enabled = enabled ? 1 : 0; c->ip_tokens = num_tokens; c->ip_scale = scale;
  return IA2P_OK;
}

static ia2p_status check_fwd_shape(ia2p_ctx* c, int B, int h, int w, int L) {
  if (B < 1 || B > 1024) return fail(c, IA2P_ERR_SHAPE, "batch %d outside 1..1024", B);
  const int div = 1 << (c->cfg.n_blocks - 1);
  if (h < div || w < div || h % div || w % div) return fail(c, IA2P_ERR_SHAPE, "latent %dx%d must be divisible by %d", h, w, div);
  if (L < 1) return fail(c, IA2P_ERR_SHAPE, "context length %d", L);
  if (c->ip_enabled && L <= c->ip_tokens) return fail(c, IA2P_ERR_SHAPE, "context length %d must exceed the %d image tokens", L, c->ip_tokens);
  return IA2P_OK;
}

size_t ia2p_workspace_bytes(ia2p_ctx* c, int B, int h, int w, int L) {
  if (!c || check_fwd_shape(c, B, h, w, L) != IA2P_OK) return 0;
  // four dry passes: the GroupNorms as launches of their own, inside their convolutions (the product path), the fused path's unfused twin, and the autotune pass
  // (GroupNorm launches + the statistics and live tensors its fused candidates need) -- a workspace sized here serves every ia2p_set_gn_fuse mode and ia2p_autotune
  size_t high = 0;
  for (int pass = 0; pass < 4; ++pass) {
    c->dry = true; c->failed = false;
    c->gn_dry_mode = pass;
    c->ws.reset((size_t)1 << 46);
    c->ws_base = nullptr;
    (void)run_forward(c, nullptr, 0.f, nullptr, L, nullptr, nullptr, nullptr, B, h, w);
    c->dry = false; c->gn_dry_mode = -1;
    if (c->failed) return 0;
    high = std::max(high, c->ws.high);
  }
  return high + 256;
}

static ia2p_status unet_forward_impl(ia2p_ctx* c, void* stream, const void* sample, float timestep, const void* context, const void* kv, int L,
                                     const void* text_embeds, const void* time_ids, void* out, int B, int h, int w, void* ws, size_t ws_bytes,
                                     const float* timesteps = nullptr, const float* ip_scales = nullptr) {
  if (!c || !sample || (!context && !kv) || !text_embeds || !time_ids || !out || !ws) return fail(c, IA2P_ERR_INVALID, "unet_forward: null argument");
  if (!c->finalized) return fail(c, IA2P_ERR_STATE, "unet_forward before weights were finalized");
  ia2p_status st = check_fwd_shape(c, B, h, w, L);
  if (st != IA2P_OK) return st;
  if (!zero_page()) return fail(c, IA2P_ERR_HIP, "cannot allocate zero page");
  if (c->fold_dirty) {            // a tensor was reloaded after finalize (hot swap, strict=False load): re-derive the folds on this stream and wait
    st = fold_all(c, (hipStream_t)stream, true);       // (rare; the wait keeps a following forward on ANOTHER stream from reading half-written folds)
    if (st != IA2P_OK) return st;
  }
  const uintptr_t base = ((uintptr_t)ws + 255) & ~(uintptr_t)255;
  const size_t usable = ws_bytes - (base - (uintptr_t)ws);
  const int key = (c->ip_enabled ? 1 + c->ip_tokens : 0) + (kv ? 1000 : 0);
  if (c->wseq_key != key) {          // (re)build the weight launch sequence with a dry pass of the same code path
    c->wseq.clear();
    c->dry = true; c->record = true; c->failed = false;
    c->ws.reset((size_t)1 << 46); c->ws_base = nullptr;
    (void)run_forward(c, nullptr, 0.f, nullptr, L, nullptr, nullptr, nullptr, B, h, w, kv ? (const half_t*)1 : nullptr);
    c->dry = false; c->record = false;
    c->wseq_key = key;
  }
  c->widx = 0;
  c->dry = false; c->failed = false; c->stream = (hipStream_t)stream;
  c->ws.reset(usable);
  c->ws_base = (char*)base;
  c->tail_pf = c->arena + c->embed_lo; c->tail_pf_bytes = (c->embed_hi - c->embed_lo) * sizeof(half_t);   // the next step starts with these
  st = run_forward(c, (const half_t*)sample, timestep, (const half_t*)context, L, (const half_t*)text_embeds, (const half_t*)time_ids, (half_t*)out, B, h, w,
                   (const half_t*)kv, timesteps, ip_scales);
  if (c->failed && st == IA2P_OK) st = IA2P_ERR_HIP;
  if (c->failed && c->err == "workspace too small") st = IA2P_ERR_NOMEM;
  return st;
}
ia2p_status ia2p_unet_forward(ia2p_ctx* c, void* stream, const void* sample, float timestep, const void* context, int L,
                              const void* text_embeds, const void* time_ids, void* out, int B, int h, int w, void* ws, size_t ws_bytes) {
  if (!context) return fail(c, IA2P_ERR_INVALID, "unet_forward: null argument");
  return unet_forward_impl(c, stream, sample, timestep, context, nullptr, L, text_embeds, time_ids, out, B, h, w, ws, ws_bytes);
}

// Per-request knobs inside ONE evaluation: `timesteps` (device, float [B]) gives every batch element its own timestep -- diffusers' UNet accepts
// a [B] timestep tensor; the reference always passes one scalar (pnp_pipeline.py:253-260, sdxl_pipeline.py:832-839) because it serves one
// request at a time -- and `ip_scales` (device, float [B], or NULL) its own IP-Adapter scale (reference: one `set_scale` value per call,
// ip_adapter.py:211-214, attention_processor.py:397). Exactly one of `context` / `kv` (ia2p_project_context) is non-NULL. Batch element b gets
// the same bits as in a uniform batch of the same size evaluated at (timesteps[b], ip_scales[b]) (tests/test_batch_gpu.py).
ia2p_status ia2p_unet_forward_v(ia2p_ctx* c, void* stream, const void* sample, const float* timesteps, const float* ip_scales, const void* context, const void* kv,
                                int L, const void* text_embeds, const void* time_ids, void* out, int B, int h, int w, void* ws, size_t ws_bytes) {
  if (!timesteps || (!context) == (!kv)) return fail(c, IA2P_ERR_INVALID, "unet_forward_v: timesteps and exactly one of context / kv are required");
  if (ip_scales && c && !c->ip_enabled) return fail(c, IA2P_ERR_STATE, "unet_forward_v: per-request IP-Adapter scales without an installed adapter");
  return unet_forward_impl(c, stream, sample, 0.f, context, kv, L, text_embeds, time_ids, out, B, h, w, ws, ws_bytes, timesteps, ip_scales);
}

// ---- context K/V hoisted out of the step: the projections depend on (context, weights) only, constant over a request's steps
size_t ia2p_context_kv_bytes(ia2p_ctx* c, int B, int L) {
  if (!c || B < 1 || L < 1 || (c->ip_enabled && L <= c->ip_tokens)) return 0;
  return (size_t)B * L * c->kv_rows * sizeof(half_t);
}
ia2p_status ia2p_project_context(ia2p_ctx* c, void* stream, const void* context, int L, int B, void* kv, size_t kv_bytes, void* ws, size_t ws_bytes) {
  if (!c || !context || !kv || !ws) return fail(c, IA2P_ERR_INVALID, "project_context: null argument");
  if (!c->finalized) return fail(c, IA2P_ERR_STATE, "project_context before weights were finalized");
  const size_t need = ia2p_context_kv_bytes(c, B, L);
  if (!need) return fail(c, IA2P_ERR_SHAPE, "project_context: B=%d L=%d", B, L);
  if (kv_bytes < need) return fail(c, IA2P_ERR_NOMEM, "project_context: kv buffer holds %zu bytes, needs %zu", kv_bytes, need);
  if (!zero_page()) return fail(c, IA2P_ERR_HIP, "cannot allocate zero page");
  if (c->fold_dirty) {
    const ia2p_status fs = fold_all(c, (hipStream_t)stream, true);
    if (fs != IA2P_OK) return fs;
  }
  const uintptr_t base = ((uintptr_t)ws + 255) & ~(uintptr_t)255;
  const bool pf = c->prefetch;
  c->prefetch = false;               // a stand-alone call: no "next launch" to stream weights for
  c->widx = 0; c->dry = false; c->failed = false; c->stream = (hipStream_t)stream;
  c->ws.reset(ws_bytes - (base - (uintptr_t)ws));
  c->ws_base = (char*)base;
  const int Lt = c->ip_enabled ? L - c->ip_tokens : L;
  project_context(c, (const half_t*)context, L, B, (half_t*)kv, (half_t*)kv + (size_t)B * Lt * c->kv_rows);
  c->prefetch = pf;
  c->wseq_key = -1;                  // the launch sequence of the next forward is rebuilt
  if (c->failed) return c->err == "workspace too small" ? IA2P_ERR_NOMEM : IA2P_ERR_HIP;
  return IA2P_OK;
}
ia2p_status ia2p_unet_forward_kv(ia2p_ctx* c, void* stream, const void* sample, float timestep, const void* kv, int L, const void* text_embeds,
                                 const void* time_ids, void* out, int B, int h, int w, void* ws, size_t ws_bytes) {
  if (!kv) return fail(c, IA2P_ERR_INVALID, "unet_forward_kv: null argument");
  return unet_forward_impl(c, stream, sample, timestep, nullptr, kv, L, text_embeds, time_ids, out, B, h, w, ws, ws_bytes);
}

// Measure-and-pick pass: one forward in which every GEMM / conv site whose shape has no measured plan yet times its
// candidate tile / K-split plans in place (tune_site) and records the fastest in the process-wide plan table.
// Re-query ia2p_workspace_bytes afterwards: K-split choices change the slab sizes.
static ia2p_status tune_begin(RunCtx* c, int reps) {
  c->tune_slab_bytes = (size_t)256 << 20; c->tune_flush_bytes = (size_t)48 << 20;
  if (hipMalloc((void**)&c->tune_scratch, c->tune_slab_bytes + c->tune_flush_bytes) != hipSuccess) {
    (void)hipGetLastError();
    c->tune_scratch = nullptr;
    return fail(c, IA2P_ERR_NOMEM, "autotune: cannot allocate %zu MiB of scratch", (c->tune_slab_bytes + c->tune_flush_bytes) >> 20);
  }
  c->tuning = true; c->tune_reps = reps < 1 ? 5 : reps; c->tune_sites = 0;
  return IA2P_OK;
}
static void tune_end(RunCtx* c, hipStream_t s) {
  c->tuning = false;
  (void)hipStreamSynchronize(s);
  (void)hipFree(c->tune_scratch);
  c->tune_scratch = nullptr;
}
ia2p_status ia2p_autotune(ia2p_ctx* c, void* stream, const void* sample, float timestep, const void* context, int L, const void* text_embeds,
                          const void* time_ids, void* out, int B, int h, int w, void* ws, size_t ws_bytes, int reps, int* sites) {
  if (!c) return fail(c, IA2P_ERR_INVALID, "autotune: null context");
  const bool prof = c->prof;
  c->prof = false;
  ia2p_status st = tune_begin(c, reps);
  if (st != IA2P_OK) return st;
  st = ia2p_unet_forward(c, stream, sample, timestep, context, L, text_embeds, time_ids, out, B, h, w, ws, ws_bytes);
  tune_end(c, (hipStream_t)stream);
  c->prof = prof;
  if (sites) *sites = c->tune_sites;
  return st;
}

ia2p_status ia2p_ddim_step(void* stream, const void* x, const void* eu, const void* ec, float g, float c_x, float c_e, void* out, void* out2, int64_t n) {
  if (!x || !eu || !out || n < 0) return fail(nullptr, IA2P_ERR_INVALID, "ddim_step: null argument");
  hipError_t e = ia2p_launch_ddim_step((const half_t*)x, (const half_t*)eu, (const half_t*)ec, g, c_x, c_e, (half_t*)out, (half_t*)out2, (long)n, (hipStream_t)stream);
  return e == hipSuccess ? IA2P_OK : fail_hip(nullptr, e, "ddim_step");
}

// The same update with per-request coefficients: coef (device, float [B][3]) = {guidance g, c_x, c_e} of batch element b, `per` elements each --
// requests with their own guidance scale (reference pipeline.py:303 `cfg`) at their own step of their own schedule share one launch.
ia2p_status ia2p_ddim_step_v(void* stream, const void* x, const void* eu, const void* ec, const float* coef, void* out, void* out2, int B, int64_t per) {
  if (!x || !eu || !out || !coef || B < 0 || per < 1) return fail(nullptr, IA2P_ERR_INVALID, "ddim_step_v: bad argument");
  hipError_t e = ia2p_launch_ddim_step((const half_t*)x, (const half_t*)eu, (const half_t*)ec, 0.f, 0.f, 0.f, (half_t*)out, (half_t*)out2, (long)B * per, (hipStream_t)stream, coef, (long)per);
  return e == hipSuccess ? IA2P_OK : fail_hip(nullptr, e, "ddim_step_v");
}

ia2p_status ia2p_mask_blend(void* stream, const void* x, const void* init, const void* noise, const void* mask, float c0, float c1,
                            void* out, void* out2, int B, int C, int64_t HW) {
  if (!x || !init || !noise || !mask || !out || B < 0 || C < 1 || HW < 1) return fail(nullptr, IA2P_ERR_INVALID, "mask_blend: bad argument");
  hipError_t e = ia2p_launch_mask_blend((const half_t*)x, (const half_t*)init, (const half_t*)noise, (const half_t*)mask, c0, c1, (half_t*)out, (half_t*)out2,
                                        B, C, (long)HW, (hipStream_t)stream);
  return e == hipSuccess ? IA2P_OK : fail_hip(nullptr, e, "mask_blend");
}

ia2p_status ia2p_prior_step(void* stream, const float* sample, const void* out_cond, const void* out_uncond, const float* noise, float g, float sqrt_a,
                            float sqrt_b, float k0, float k1, float sigma, float* out, int64_t n) {
  if (!sample || !out_uncond || !out || n < 0 || !(sqrt_a > 0.f) || !(sqrt_b > 0.f)) return fail(nullptr, IA2P_ERR_INVALID, "prior_step: bad argument");
  hipError_t e = ia2p_launch_prior_step(sample, (const half_t*)out_cond, (const half_t*)out_uncond, noise, g, sqrt_a, sqrt_b, k0, k1, sigma, out, (long)n, (hipStream_t)stream);
  return e == hipSuccess ? IA2P_OK : fail_hip(nullptr, e, "prior_step");
}

// ---- per-operator entry points ---------------------------------------------------------------------------------------

ia2p_status ia2p_groupnorm_silu(void* stream, const void* x, void* y, const void* gamma, const void* beta, int B, int HW, int C, int groups, float eps, int silu, float* partial) {
  if (!x || !y || !gamma || !beta || !partial) return fail(nullptr, IA2P_ERR_INVALID, "groupnorm: null argument");
  if (C % 8 || C % groups || groups > 256) return fail(nullptr, IA2P_ERR_SHAPE, "groupnorm: C=%d groups=%d", C, groups);
  hipError_t e = ia2p_launch_groupnorm((const half_t*)x, C, (half_t*)y, C, (const half_t*)gamma, (const half_t*)beta, partial, B, HW, C, groups, eps, silu, (hipStream_t)stream);
  RET_HIP(e, "groupnorm");
}
ia2p_status ia2p_layernorm(void* stream, const void* x, void* y, const void* gamma, const void* beta, int M, int C, float eps) {
  if (!x || !y || !gamma || !beta) return fail(nullptr, IA2P_ERR_INVALID, "layernorm: null argument");
  if (C % 8 || C > 2048) return fail(nullptr, IA2P_ERR_SHAPE, "layernorm: C=%d must be a multiple of 8 and <= 2048", C);
  hipError_t e = ia2p_launch_layernorm((const half_t*)x, C, (half_t*)y, C, (const half_t*)gamma, (const half_t*)beta, M, C, eps, (hipStream_t)stream);
  RET_HIP(e, "layernorm");
}
#ifdef IA2P_CLOCK_STAMP
// Diagnostic builds only (IA2P_EXTRA_FLAGS=-DIA2P_CLOCK_STAMP; not declared in the headers, not part of the product library): in-kernel s_memrealtime stamps of the
// launches of one layer role INSIDE a step. buf: device memory, `launch_slots` x STAMP_WG x 8 u64, zeroed by the caller; role: ROLE_* index (ia2p_profile_read_role).
int ia2p_debug_stamp_begin(ia2p_ctx* c, int role, void* buf, int launch_slots) {
  if (!c) return -1;
  c->stamp_buf = (unsigned long long*)buf; c->stamp_role = role; c->stamp_cap = buf ? launch_slots : 0; c->stamp_n = 0; c->stamp_meta.clear();
  return RunCtx::STAMP_WG;
}
// launches stamped since begin; meta (optional): 5 ints per launch {M, N, K, variant, tiles}
int ia2p_debug_stamp_read(ia2p_ctx* c, int* meta, int max_launches) {
  if (!c) return -1;
  for (int i = 0; meta && i < (int)c->stamp_meta.size() && i < max_launches; ++i) {
    const auto& m = c->stamp_meta[i];
    meta[5 * i] = m.M; meta[5 * i + 1] = m.N; meta[5 * i + 2] = m.K; meta[5 * i + 3] = m.variant; meta[5 * i + 4] = m.tiles;
  }
  return (int)c->stamp_meta.size();
}
#endif
ia2p_status ia2p_gemm(void* stream, const void* A, const void* W, const void* bias, const void* residual, void* C, int M, int N, int K, int geglu) {
  if (!A || !W || !C) return fail(nullptr, IA2P_ERR_INVALID, "gemm: null argument");
  if (K % 64 || N % 4 || (geglu && (N % 32 || !bias))) return fail(nullptr, IA2P_ERR_SHAPE, "gemm: K=%d must be a multiple of 64, N=%d of 4 (GEGLU: 32, with bias)", K, N);
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  const int No = geglu ? N / 2 : N;
  a.A = (const half_t*)A; a.W = (const half_t*)W; a.C = (half_t*)C; a.zero = zero_page(); a.M = M; a.N = N; a.K = K; a.ldw = a.K; a.lda = K; a.ldc = No;
  a.bias = (const half_t*)bias; a.residual = (const half_t*)residual; a.ldr = No; a.geglu = geglu; a.rows_per_batch = 1;
  a.m_fastest = M <= N;
  hipError_t e = ia2p_launch_gemm(a, false, (hipStream_t)stream, nullptr);
  RET_HIP(e, "gemm");
}
// GEGLU feed-forward of a BasicTransformerBlock as an operator: H = geglu(X . W1p^T + b1p) [M, 4 C] (packed weights: ia2p_pack_geglu), out = H . W2^T + b2 + R,
// as the executor runs it (two launches, the library's plans). splitk / partial: K split of the second GEMM (partial: splitk * M * C floats) or 0.
ia2p_status ia2p_ffn(void* stream, const void* X, const void* W1p, const void* b1p, const void* W2, const void* b2, const void* R, void* H, void* out,
                     int M, int C, int splitk, float* partial) {
  if (!X || !W1p || !b1p || !W2 || !H || !out) return fail(nullptr, IA2P_ERR_INVALID, "ffn: null argument");
  if (C % 64 || (splitk > 1 && (!partial || splitk > 4 * C / 64 || splitk > 255))) return fail(nullptr, IA2P_ERR_SHAPE, "ffn: C=%d must be a multiple of 64, splitk=%d needs slabs", C, splitk);
  RunCtx rc;
  GemmArgs a = gemm_args(&rc, (const half_t*)X, C, (const half_t*)W1p, (const half_t*)b1p, nullptr, 0, (half_t*)H, 4 * C, M, 8 * C, C, 1, 0, 0, 0, 0, nullptr, nullptr, 0);
  GemmArgs b = gemm_args(&rc, (const half_t*)H, 4 * C, (const half_t*)W2, (const half_t*)b2, (const half_t*)R, C, (half_t*)out, C, M, C, 4 * C, 0, 0, 0, 0, 0, nullptr, nullptr, 0);
  a.m_fastest = a.M <= a.N; b.m_fastest = b.M <= b.N;
  if (splitk > 1) { b.splitk = splitk; b.partial = partial; }
  const GemmPlan pa = ia2p_gemm_plan(a.M, a.N, a.K, false, true), pb = ia2p_gemm_plan(b.M, b.N, b.K, false, false);
  hipError_t e = ia2p_launch_gemm_variant(a, false, pa.variant, (hipStream_t)stream);
  if (e == hipSuccess) e = ia2p_launch_gemm_variant(b, false, pb.variant, (hipStream_t)stream);
  RET_HIP(e, "ffn");
}
ia2p_status ia2p_fold_layernorm(void* stream, const void* W, const void* gamma, const void* beta, const void* bias, void* Wf, float* colsum,
                                float* fbias, int N, int K) {
  if (!W || !gamma || !beta || !Wf || !colsum || !fbias || N < 1 || K < 1) return fail(nullptr, IA2P_ERR_INVALID, "fold_layernorm: bad argument");
  hipError_t e = ia2p_launch_fold_ln((const half_t*)W, (const half_t*)gamma, (const half_t*)beta, (const half_t*)bias, (half_t*)Wf, colsum, fbias, N, K, (hipStream_t)stream);
  RET_HIP(e, "fold_layernorm");
}
ia2p_status ia2p_gemm_ex(void* stream, const void* A, const void* W, const void* bias, const void* residual, void* C, int M, int N, int K, int geglu,
                         const ia2p_ln_fold* ln, float* stats_out, int* stats_slots, int splitk, float* partial) {
  if (!A || !W || !C) return fail(nullptr, IA2P_ERR_INVALID, "gemm_ex: null argument");
  if (K % 64 || N % 4 || (geglu && (N % 32 || (!bias && !ln)))) return fail(nullptr, IA2P_ERR_SHAPE, "gemm_ex: K=%d must be a multiple of 64, N=%d of 4 (GEGLU: 32, with bias)", K, N);
  if (splitk > 1 && (!partial || geglu || splitk > K / 64 || splitk > 255)) return fail(nullptr, IA2P_ERR_SHAPE, "gemm_ex: splitk=%d needs a slab, no GEGLU, and <= min(K/64, 255)", splitk);
  if (ln && (!ln->stats || !ln->colsum || !ln->fbias || ln->slots < 1)) return fail(nullptr, IA2P_ERR_INVALID, "gemm_ex: incomplete ia2p_ln_fold");
  if (stats_out && geglu) return fail(nullptr, IA2P_ERR_INVALID, "gemm_ex: row statistics of a GEGLU output are not provided");
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  const int No = geglu ? N / 2 : N;
  a.A = (const half_t*)A; a.W = (const half_t*)W; a.C = (half_t*)C; a.zero = zero_page(); a.M = M; a.N = N; a.K = K; a.ldw = a.K; a.lda = K; a.ldc = No;
  a.bias = (const half_t*)bias; a.residual = (const half_t*)residual; a.ldr = No; a.geglu = geglu; a.rows_per_batch = 1;
  a.m_fastest = M <= N;
  if (ln) { a.ln_stats = ln->stats; a.ln_slots = ln->slots; a.ln_cs = ln->colsum; a.ln_bias = ln->fbias; a.ln_eps = ln->eps; }
  a.stats_out = stats_out;
  if (splitk > 1) { a.splitk = splitk; a.partial = partial; }
#ifdef IA2P_CLOCK_STAMP
  else if (partial) a.partial = partial;      // (diagnostic builds: the in-kernel stamps of an unsplit launch go to the caller's buffer, tools/insitu_stamps.py)
#endif
  int pick = 0, combined = 0;
  hipError_t e = ia2p_launch_gemm(a, false, (hipStream_t)stream, &pick, &combined);
  if (stats_slots && pick >= 0 && pick < IA2P_GEMM_NVARIANT) *stats_slots = (splitk > 1 && !combined) ? 1 : (N + IA2P_GEMM_TILES[pick].bn - 1) / IA2P_GEMM_TILES[pick].bn;
  RET_HIP(e, "gemm_ex");
}
ia2p_status ia2p_gemm_splitk(void* stream, const void* A, const void* W, const void* bias, const void* residual, void* C, int M, int N, int K,
                             int splitk, float* partial) {
  if (!A || !W || !C || !partial) return fail(nullptr, IA2P_ERR_INVALID, "gemm_splitk: null argument");
  if (K % 64 || N % 4 || splitk < 1 || splitk > K / 64 || splitk > 255) return fail(nullptr, IA2P_ERR_SHAPE, "gemm_splitk: K=%d N=%d splitk=%d (1 .. min(K / 64, 255))", K, N, splitk);
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  a.A = (const half_t*)A; a.W = (const half_t*)W; a.C = (half_t*)C; a.zero = zero_page(); a.M = M; a.N = N; a.K = K; a.ldw = a.K; a.lda = K; a.ldc = N;
  a.bias = (const half_t*)bias; a.residual = (const half_t*)residual; a.ldr = N; a.rows_per_batch = 1; a.m_fastest = M <= N;
  a.splitk = splitk; a.partial = partial;
  hipError_t e = ia2p_launch_gemm(a, false, (hipStream_t)stream, nullptr);
  RET_HIP(e, "gemm_splitk");
}
ia2p_status ia2p_conv3x3(void* stream, const void* x, const void* Wp, const void* bias, const void* rowvec, const void* residual, void* y,
                         int B, int Hs, int Ws, int Cin, int Co, int stride, int up) {
  if (!x || !Wp || !y) return fail(nullptr, IA2P_ERR_INVALID, "conv3x3: null argument");
  if (Cin % 64 || Co % 4 || (stride != 1 && stride != 2) || (up != 0 && up != 1)) return fail(nullptr, IA2P_ERR_SHAPE, "conv3x3: Cin=%d (mult of 64) Co=%d (mult of 4) stride=%d up=%d", Cin, Co, stride, up);
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  const int Hv = Hs << up, Wv = Ws << up;
  a.Ho = (Hv - 1) / stride + 1; a.Wo = (Wv - 1) / stride + 1;
  a.A = (const half_t*)x; a.W = (const half_t*)Wp; a.C = (half_t*)y; a.zero = zero_page(); a.M = B * a.Ho * a.Wo; a.N = Co; a.K = 9 * Cin; a.ldw = a.K; a.lda = Cin; a.ldc = Co;
  a.Hs = Hs; a.Ws = Ws; a.stride = stride; a.up = up; a.Cin = Cin; a.bias = (const half_t*)bias;
  a.rowvec = (const half_t*)rowvec; a.rowvec_ld = Co; a.rows_per_batch = a.Ho * a.Wo; a.residual = (const half_t*)residual; a.ldr = Co;
  hipError_t e = ia2p_launch_gemm(a, true, (hipStream_t)stream, nullptr);
  RET_HIP(e, "conv3x3");
}
// the stride-1 form with K split over `splitk` workgroups per tile (what the executor launches for the 16 x 16 feature maps); partial: splitk * B*Hs*Ws * Co floats
ia2p_status ia2p_conv3x3_splitk(void* stream, const void* x, const void* Wp, const void* bias, const void* rowvec, const void* residual, void* y,
                                int B, int Hs, int Ws, int Cin, int Co, int splitk, float* partial) {
  if (!x || !Wp || !y || (splitk > 1 && !partial)) return fail(nullptr, IA2P_ERR_INVALID, "conv3x3_splitk: null argument (partial is needed for splitk > 1 only)");
  if (Cin % 64 || Co % 4 || splitk < 1 || splitk > 9 * Cin / 64 || splitk > 255) return fail(nullptr, IA2P_ERR_SHAPE, "conv3x3_splitk: Cin=%d (mult of 64) Co=%d (mult of 4) splitk=%d (1 .. 9 Cin / 64)", Cin, Co, splitk);
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  a.Ho = Hs; a.Wo = Ws;
  a.A = (const half_t*)x; a.W = (const half_t*)Wp; a.C = (half_t*)y; a.zero = zero_page(); a.M = B * Hs * Ws; a.N = Co; a.K = 9 * Cin; a.ldw = a.K; a.lda = Cin; a.ldc = Co;
  a.Hs = Hs; a.Ws = Ws; a.stride = 1; a.up = 0; a.Cin = Cin; a.bias = (const half_t*)bias;
  a.rowvec = (const half_t*)rowvec; a.rowvec_ld = Co; a.rows_per_batch = Hs * Ws; a.residual = (const half_t*)residual; a.ldr = Co;
  if (splitk > 1) { a.splitk = splitk; a.partial = partial; }
  hipError_t e = ia2p_launch_gemm(a, true, (hipStream_t)stream, nullptr);
  RET_HIP(e, "conv3x3_splitk");
}
// ---- GroupNorm from producer-side column sums (round 5; csrc/gn_fold.h): the operators of the fused path, one by one -----------------------------------------------
// canonical statistics of a tensor x [M, C]: out[(slot * C + c) * 2 + {0, 1}] = {sum, sum of squares} (fp64) over the `rows` rows of slot `slot` (what a GEMM / conv epilogue
// leaves for its own output when asked: ia2p_gemm_gnstats, ia2p_conv3x3_gn)
ia2p_status ia2p_gn_colstats(void* stream, const void* x, int M, int C, int rows, double* out) {
  if (!x || !out) return fail(nullptr, IA2P_ERR_INVALID, "gn_colstats: null argument");
  if (C < 8 || C % 8 || rows < 16 || rows % 16 || M < 1 || M % rows) return fail(nullptr, IA2P_ERR_SHAPE, "gn_colstats: C=%d (multiple of 8), rows=%d (multiple of 16 dividing M=%d)", C, rows, M);
  hipError_t e = ia2p_launch_gn_colstats((const half_t*)x, C, M, C, rows, out, (hipStream_t)stream);
  RET_HIP(e, "gn_colstats");
}
static ia2p_status gn_in_from_abi(const char* what, GemmArgs::GnIn* g, int C0, const double* st0, int rows0, int C1, const double* st1, int rows1, const void* gamma, const void* beta, int groups, float eps, int silu, int HW) {
  const int C = C0 + C1;
  if (!st0 || !gamma || !beta || (C1 > 0 && !st1)) return fail(nullptr, IA2P_ERR_INVALID, "%s: null argument", what);
  if (groups < 1 || groups > 64 || C % groups || C0 < 8 || C0 % 8 || C1 < 0 || C1 % 8 || rows0 < 1 || HW % rows0 || (C1 > 0 && (rows1 < 1 || HW % rows1)))
    return fail(nullptr, IA2P_ERR_SHAPE, "%s: C0=%d C1=%d groups=%d rows0=%d rows1=%d HW=%d", what, C0, C1, groups, rows0, rows1, HW);
  memset(g, 0, sizeof *g);
  g->st0 = st0; g->rows0 = rows0; g->st1 = C1 > 0 ? st1 : nullptr; g->rows1 = rows1; g->C0 = C0; g->gamma = (const half_t*)gamma; g->beta = (const half_t*)beta;
  g->groups = groups; g->gs = C / groups; g->eps = eps; g->silu = silu;
  return IA2P_OK;
}
// y = [silu](GroupNorm(groups)([x0 | x1])) with the statistics folded from the column sums of the sources' producers: the stand-alone twin of what ia2p_conv3x3_gn
// does to its operand inside the convolution (same fold, same scale / shift, same element formula: the two agree to the bit)
ia2p_status ia2p_gn_apply_stats(void* stream, const void* x0, int C0, const double* st0, int rows0, const void* x1, int C1, const double* st1, int rows1,
                                const void* gamma, const void* beta, void* y, int B, int HW, int groups, float eps, int silu) {
  if (!x0 || !y || (C1 > 0 && !x1)) return fail(nullptr, IA2P_ERR_INVALID, "gn_apply_stats: null argument");
  GemmArgs::GnIn g;
  const ia2p_status st = gn_in_from_abi("gn_apply_stats", &g, C0, st0, rows0, C1, st1, rows1, gamma, beta, groups, eps, silu, HW);
  if (st != IA2P_OK) return st;
  hipError_t e = ia2p_launch_gn_apply_stats((const half_t*)x0, C0, C1 > 0 ? (const half_t*)x1 : nullptr, C1, (half_t*)y, C0 + C1, B, HW, C0 + C1, g, (hipStream_t)stream);
  RET_HIP(e, "gn_apply_stats");
}
// 3x3 convolution (stride 1) of silu(GroupNorm([x0 | x1])) with the norm applied INSIDE the convolution (d->st0 != NULL; conv_halo_kernel.h GN = 1), or of x0 itself
// (d->st0 == NULL), + optional appended 1x1 block, time-embedding row, residual, K split; d->gn_out != NULL: also the column sums of y (*gn_out_rows: rows per slot,
// 0 when this launch could not take them). Fused form: the site must have a halo-staged plan (IA2P_ERR_SHAPE otherwise; tests force one with ia2p_debug_set_gemm_tile).
ia2p_status ia2p_conv3x3_gn(void* stream, const ia2p_conv_gn* d, int* gn_out_rows) {
  if (gn_out_rows) *gn_out_rows = 0;
  if (!d || !d->x0 || !d->Wp || !d->y || (d->splitk > 1 && !d->partial) || (d->Ca > 0 && !d->xa)) return fail(nullptr, IA2P_ERR_INVALID, "conv3x3_gn: null argument");
  const int Cin = d->C0 + (d->st0 ? d->C1 : 0), HW = d->H * d->W;
  if (Cin % 64 || d->C0 % 64 || d->Co % 8 || d->Ca % 64 || d->B < 1 || HW < 1 || d->splitk < 0 || d->splitk > (9 * Cin + d->Ca) / 64 || d->splitk > 255) return fail(nullptr, IA2P_ERR_SHAPE, "conv3x3_gn: C0=%d C1=%d Co=%d Ca=%d splitk=%d", d->C0, d->C1, d->Co, d->Ca, d->splitk);
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1; a.Ho = a.Hs = d->H; a.Wo = a.Ws = d->W; a.stride = 1;
  a.A = (const half_t*)d->x0; a.W = (const half_t*)d->Wp; a.C = (half_t*)d->y; a.zero = zero_page(); a.M = d->B * HW; a.N = d->Co; a.K = 9 * Cin + d->Ca; a.ldw = a.K; a.lda = d->C0; a.ldc = d->Co;
  a.Cin = Cin; a.bias = (const half_t*)d->bias; a.rowvec = (const half_t*)d->rowvec; a.rowvec_ld = d->Co; a.rows_per_batch = HW; a.residual = (const half_t*)d->residual; a.ldr = d->Co;
  if (d->Ca > 0) { a.A2 = (const half_t*)d->xa; a.lda2 = d->Ca; a.Cin2 = d->Ca; }
  if (d->splitk > 1) { a.splitk = d->splitk; a.partial = d->partial; }
  const GemmPlan pl = ia2p_gemm_plan(a.M, a.N, a.K, true, false);
  if (d->st0) {
    const ia2p_status st = gn_in_from_abi("conv3x3_gn", &a.gn, d->C0, d->st0, d->rows0, d->C1, d->st1, d->rows1, d->gamma, d->beta, d->groups, d->eps, 1, HW);
    if (st != IA2P_OK) return st;
    a.A1b = d->C1 > 0 ? (const half_t*)d->x1 : nullptr; a.lda1b = d->C1;
    if (d->C1 > 0 && !d->x1) return fail(nullptr, IA2P_ERR_INVALID, "conv3x3_gn: null second source");
    if (!ia2p_conv_gn_fusable(a, pl.variant, a.splitk) || !ia2p_conv_gn_ok(a)) return fail(nullptr, IA2P_ERR_SHAPE, "conv3x3_gn: this site has no halo-staged plan (variant %d) or its statistics do not fit the fused kernel", pl.variant);
  }
  const bool combined = a.splitk > 1 && ia2p_splitk_inkernel(a.M, a.N, a.splitk);
  int rows = 0;
  if (d->gn_out) { rows = gn_epilogue_rows(a, true, pl.variant, a.splitk, combined, HW); if (rows) a.gn_out = d->gn_out; }
  int comb = 0;
  hipError_t e = ia2p_launch_gemm_variant(a, true, pl.variant, (hipStream_t)stream, true, &comb);
  if (e == hipSuccess && gn_out_rows) *gn_out_rows = (rows && (a.splitk <= 1 || comb)) ? rows : 0;
  RET_HIP(e, "conv3x3_gn");
}
// C = A . W^T + bias + residual as ia2p_gemm_splitk (splitk <= 1: no split), also leaving the GroupNorm column sums of C for images of HW rows (a Transformer2DModel's
// proj_out in front of the next ResnetBlock2D); *rows: rows per slot, 0 when the tile the plan picked cannot take them (the caller runs ia2p_gn_colstats)
ia2p_status ia2p_gemm_gnstats(void* stream, const void* A, const void* W, const void* bias, const void* residual, void* C, int M, int N, int K, int splitk, float* partial,
                              int HW, double* gn_out, int* rows) {
  if (rows) *rows = 0;
  if (!A || !W || !C || !gn_out || !rows || (splitk > 1 && !partial)) return fail(nullptr, IA2P_ERR_INVALID, "gemm_gnstats: null argument");
  if (K % 64 || N % 8 || HW < 16 || M % HW || splitk < 0 || splitk > K / 64 || splitk > 255) return fail(nullptr, IA2P_ERR_SHAPE, "gemm_gnstats: K=%d N=%d HW=%d splitk=%d", K, N, HW, splitk);
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  a.A = (const half_t*)A; a.W = (const half_t*)W; a.C = (half_t*)C; a.zero = zero_page(); a.M = M; a.N = N; a.K = K; a.ldw = a.K; a.lda = K; a.ldc = N;
  a.bias = (const half_t*)bias; a.residual = (const half_t*)residual; a.ldr = N; a.rows_per_batch = 1; a.m_fastest = M <= N;
  if (splitk > 1) { a.splitk = splitk; a.partial = partial; }
  const GemmPlan pl = ia2p_gemm_plan(M, N, K, false, false);
  const bool combined = splitk > 1 && ia2p_splitk_inkernel(M, N, splitk);
  const int r = gn_epilogue_rows(a, false, pl.variant, a.splitk, combined, HW);
  if (r) a.gn_out = gn_out;
  int comb = 0;
  hipError_t e = ia2p_launch_gemm_variant(a, false, pl.variant, (hipStream_t)stream, true, &comb);
  if (e == hipSuccess) *rows = (r && (a.splitk <= 1 || comb)) ? r : 0;
  RET_HIP(e, "gemm_gnstats");
}
// ResnetBlock2D tail as one implicit GEMM: y = conv3x3(x, W2) + conv1x1(x2, Wsc) + bias (+ rowvec), K = 9 Cin + Cin2; Wcat rows = [packed W2 row | Wsc row]
ia2p_status ia2p_conv3x3_cat(void* stream, const void* x, const void* x2, const void* Wcat, const void* bias, void* y, int B, int Hs, int Ws, int Cin, int Cin2, int Co) {
  if (!x || !x2 || !Wcat || !y) return fail(nullptr, IA2P_ERR_INVALID, "conv3x3_cat: null argument");
  if (Cin % 64 || Cin2 % 64 || Cin2 < 64 || Co % 4) return fail(nullptr, IA2P_ERR_SHAPE, "conv3x3_cat: Cin=%d, Cin2=%d (multiples of 64) Co=%d (mult of 4)", Cin, Cin2, Co);
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  a.Ho = Hs; a.Wo = Ws;
  a.A = (const half_t*)x; a.W = (const half_t*)Wcat; a.C = (half_t*)y; a.zero = zero_page(); a.M = B * Hs * Ws; a.N = Co; a.K = 9 * Cin + Cin2; a.ldw = a.K; a.lda = Cin; a.ldc = Co;
  a.Hs = Hs; a.Ws = Ws; a.stride = 1; a.up = 0; a.Cin = Cin; a.bias = (const half_t*)bias; a.rows_per_batch = Hs * Ws;
  a.A2 = (const half_t*)x2; a.lda2 = Cin2; a.Cin2 = Cin2;
  hipError_t e = ia2p_launch_gemm(a, true, (hipStream_t)stream, nullptr);
  RET_HIP(e, "conv3x3_cat");
}
ia2p_status ia2p_pack_conv3x3(void* stream, const void* src, void* dst, int Co, int Cin) {
  if (!src || !dst) return fail(nullptr, IA2P_ERR_INVALID, "pack_conv3x3: null argument");
  if (Cin % 64) return fail(nullptr, IA2P_ERR_SHAPE, "pack_conv3x3: Cin=%d must be a multiple of 64 (the layout of ia2p_conv3x3; ia2p_pack_conv_out packs for ia2p_conv_out)", Cin);
  hipError_t e = ia2p_launch_pack_conv((const half_t*)src, (half_t*)dst, Co, Cin, (hipStream_t)stream);
  RET_HIP(e, "pack_conv3x3");
}
ia2p_status ia2p_pack_conv_out(void* stream, const void* src, void* dst, int Co, int C) {
  if (!src || !dst) return fail(nullptr, IA2P_ERR_INVALID, "pack_conv_out: null argument");
  hipError_t e = ia2p_launch_pack_conv((const half_t*)src, (half_t*)dst, Co, C, (hipStream_t)stream);
  RET_HIP(e, "pack_conv_out");
}
// latent-boundary convolutions as operators (the executors call the launchers directly): conv_in reads NCHW and writes channels-last,
// conv_out reads channels-last and writes NCHW; reference call sites: the diffusers UNet's conv_in / conv_out behind pnp_pipeline.py:253-260
ia2p_status ia2p_conv_in(void* stream, const void* x_nchw, const void* w_oihw, const void* bias, void* y_nhwc, void* w_scratch, int B, int Cin, int H, int W, int Co) {
  if (!x_nchw || !w_oihw || !bias || !y_nhwc || !w_scratch) return fail(nullptr, IA2P_ERR_INVALID, "conv_in: null argument");
  if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cin * 9 > 64 || Co < 8 || Co % 8) return fail(nullptr, IA2P_ERR_SHAPE, "conv_in: Cin*9=%d must be <= 64, Co=%d a multiple of 8", Cin * 9, Co);
  hipError_t e = ia2p_launch_pack_conv_in((const half_t*)w_oihw, (half_t*)w_scratch, Co, Cin * 9, (hipStream_t)stream);
  if (e == hipSuccess) e = ia2p_launch_conv_in((const half_t*)x_nchw, (const half_t*)w_scratch, (const half_t*)bias, (half_t*)y_nhwc, B, Cin, H, W, Co, (hipStream_t)stream);
  RET_HIP(e, "conv_in");
}
ia2p_status ia2p_conv_out(void* stream, const void* x_nhwc, const void* w_packed, const void* bias, void* y_nchw, int B, int C, int H, int W, int Co) {
  if (!x_nhwc || !w_packed || !bias || !y_nchw) return fail(nullptr, IA2P_ERR_INVALID, "conv_out: null argument");
  if (B < 1 || H < 1 || W < 1 || C < 32 || C % 32 || Co < 1 || Co > 8) return fail(nullptr, IA2P_ERR_SHAPE, "conv_out: C=%d must be a multiple of 32, Co=%d <= 8", C, Co);
  hipError_t e = ia2p_launch_conv_out((const half_t*)x_nhwc, C, (const half_t*)w_packed, (const half_t*)bias, (half_t*)y_nchw, B, C, H, W, Co, (hipStream_t)stream);
  RET_HIP(e, "conv_out");
}
ia2p_status ia2p_pack_geglu(void* stream, const void* src, void* dst, int rows, int rowlen) {
  if (!src || !dst || rows % 32) return fail(nullptr, IA2P_ERR_SHAPE, "pack_geglu: rows must be a multiple of 32");
  hipError_t e = ia2p_launch_pack_geglu((const half_t*)src, (half_t*)dst, rows, rowlen, (hipStream_t)stream);
  RET_HIP(e, "pack_geglu");
}
ia2p_status ia2p_attention(void* stream, const void* Q, int ldq, void* O, int ldo, int B, int heads, int Nq, int nseg,
                           const void* K0, const void* V0, int ld0, int nkeys0, float w0, const void* K1, const void* V1, int ld1, int nkeys1, float w1) {
  if (!Q || !O || !K0 || !V0 || nseg < 1 || nseg > 2 || (nseg == 2 && (!K1 || !V1))) return fail(nullptr, IA2P_ERR_INVALID, "attention: bad argument");
  if (nkeys0 < 1 || (nseg == 2 && nkeys1 < 1) || ldq % 8 || ldo % 8 || (((uintptr_t)O) & 15) || ld0 % 8 || (nseg == 2 && ld1 % 8)) return fail(nullptr, IA2P_ERR_SHAPE, "attention: key counts must be >= 1, strides multiples of 8, O 16-byte aligned");
  AttnArgs a;
  memset(&a, 0, sizeof a);
  a.Q = (const half_t*)Q; a.ldq = ldq; a.O = (half_t*)O; a.ldo = ldo; a.B = B; a.heads = heads; a.Nq = Nq; a.nseg = nseg;
  a.scale_log2e = 0.125f * 1.4426950408889634f;
  a.seg[0] = AttnSeg{(const half_t*)K0, (const half_t*)V0, nkeys0, ld0, nkeys0, w0};
  a.seg[1] = AttnSeg{(const half_t*)K1, (const half_t*)V1, nkeys1, ld1, nkeys1, w1};
  hipError_t e = ia2p_launch_attention(a, (hipStream_t)stream);
  RET_HIP(e, "attention");
}
ia2p_status ia2p_qproj_attention(void* stream, const void* X, const void* Wq, const void* bias, const ia2p_ln_fold* ln, void* O, int ldo, int B, int heads,
                                 int Nq, int K, int nseg, const void* K0, const void* V0, int ld0, int nkeys0, float w0,
                                 const void* K1, const void* V1, int ld1, int nkeys1, float w1) {
  if (!X || !Wq || !O || !K0 || !V0 || nseg < 1 || nseg > 2 || (nseg == 2 && (!K1 || !V1))) return fail(nullptr, IA2P_ERR_INVALID, "qproj_attention: bad argument");
  if (ln && (!ln->stats || !ln->colsum || !ln->fbias || ln->slots < 1)) return fail(nullptr, IA2P_ERR_INVALID, "qproj_attention: incomplete ia2p_ln_fold");
  if (B < 1 || heads < 1 || Nq < 128 || Nq % 128 || K < 64 || K % 64 || nkeys0 < 1 || (nseg == 2 && nkeys1 < 1) || ldo % 8 || (((uintptr_t)O) & 15) || ld0 % 8 || (nseg == 2 && ld1 % 8))
    return fail(nullptr, IA2P_ERR_SHAPE, "qproj_attention: Nq=%d must be a multiple of 128, K=%d of 64, key counts >= 1, strides multiples of 8, O 16-byte aligned", Nq, K);
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  a.A = (const half_t*)X; a.W = (const half_t*)Wq; a.zero = zero_page(); a.M = B * Nq; a.N = heads * 64; a.K = K; a.ldw = K; a.lda = K; a.ldc = a.N;
  a.bias = (const half_t*)bias; a.rows_per_batch = 1; a.m_fastest = a.M <= a.N;
  if (ln) { a.ln_stats = ln->stats; a.ln_slots = ln->slots; a.ln_cs = ln->colsum; a.ln_bias = ln->fbias; a.ln_eps = ln->eps; }
  AttnArgs x;
  memset(&x, 0, sizeof x);
  x.O = (half_t*)O; x.ldo = ldo; x.B = B; x.heads = heads; x.Nq = Nq; x.nseg = nseg;
  x.scale_log2e = 0.125f * 1.4426950408889634f;
  x.seg[0] = AttnSeg{(const half_t*)K0, (const half_t*)V0, nkeys0, ld0, nkeys0, w0};
  x.seg[1] = AttnSeg{(const half_t*)K1, (const half_t*)V1, nkeys1, ld1, nkeys1, w1};
  if (!ia2p_qproj_xattn_ok(a, x)) return fail(nullptr, IA2P_ERR_SHAPE, "qproj_attention: shape not supported by the fused tile (bias must be 16-byte aligned)");
  hipError_t e = ia2p_launch_qproj_xattn(a, x, (hipStream_t)stream);
  RET_HIP(e, "qproj_attention");
}
ia2p_status ia2p_qkv_self_attention(void* stream, const void* X, const void* Wqkv, const void* bias, const ia2p_ln_fold* ln, void* O, int ldo, int B, int heads, int K) {
  if (!X || !Wqkv || !O) return fail(nullptr, IA2P_ERR_INVALID, "qkv_self_attention: null argument");
  if (ln && (!ln->stats || !ln->colsum || !ln->fbias || ln->slots < 1)) return fail(nullptr, IA2P_ERR_INVALID, "qkv_self_attention: incomplete ia2p_ln_fold");
  if (B < 1 || heads < 1 || K < 64 || K % 64 || ldo % 8 || (((uintptr_t)O) & 15)) return fail(nullptr, IA2P_ERR_SHAPE, "qkv_self_attention: K=%d (multiple of 64), ldo=%d (multiple of 8), O 16-byte aligned", K, ldo);
  GemmArgs a;
  memset(&a, 0, sizeof a);
  a.pad = 1;
  a.A = (const half_t*)X; a.W = (const half_t*)Wqkv; a.zero = zero_page(); a.M = B * 256; a.N = 3 * heads * 64; a.K = K; a.ldw = K; a.lda = K; a.ldc = a.N;
  a.bias = (const half_t*)bias; a.rows_per_batch = 1;
  if (ln) { a.ln_stats = ln->stats; a.ln_slots = ln->slots; a.ln_cs = ln->colsum; a.ln_bias = ln->fbias; a.ln_eps = ln->eps; }
  AttnArgs x;
  memset(&x, 0, sizeof x);
  x.O = (half_t*)O; x.ldo = ldo; x.B = B; x.heads = heads; x.Nq = 256; x.nseg = 1;
  x.scale_log2e = 0.125f * 1.4426950408889634f;
  x.seg[0].nkeys = 256; x.seg[0].weight = 1.f;
  if (!ia2p_qkv_sattn_ok(a, x)) return fail(nullptr, IA2P_ERR_SHAPE, "qkv_self_attention: shape / alignment not supported by the fused tile");
  hipError_t e = ia2p_launch_qkv_sattn(a, x, (hipStream_t)stream);
  RET_HIP(e, "qkv_self_attention");
}
ia2p_status ia2p_ip_attn_map(void* stream, const void* Q, int ldq, const void* Kip, int ldk, void* out, int B, int heads, int Nq, int ntok) {
  if (!Q || !Kip || !out || B < 1 || heads < 1 || Nq < 1) return fail(nullptr, IA2P_ERR_INVALID, "ip_attn_map: bad argument");
  if (ntok < 1 || ntok > 16 || ldq % 8 || ldq < heads * 64 || ldk < heads * 64) return fail(nullptr, IA2P_ERR_SHAPE, "ip_attn_map: ntok=%d (1..16), ldq=%d (mult of 8), ldk=%d", ntok, ldq, ldk);
  hipError_t e = ia2p_launch_ip_attn_map((const half_t*)Q, ldq, (const half_t*)Kip, ldk, (half_t*)out, B, heads, Nq, ntok, (hipStream_t)stream);
  RET_HIP(e, "ip_attn_map");
}
ia2p_status ia2p_linear_small(void* stream, const void* X, const void* W, const void* bias, void* out, int M, int N, int K, int silu_in, int silu_out) {
  if (!X || !W || !out) return fail(nullptr, IA2P_ERR_INVALID, "linear_small: null argument");
  if (M > 16 || K % 8) return fail(nullptr, IA2P_ERR_SHAPE, "linear_small: M=%d (<=16) K=%d (mult of 8)", M, K);
  hipError_t e = ia2p_launch_linear_small((const half_t*)X, K, (const half_t*)W, (const half_t*)bias, nullptr, 0, (half_t*)out, N, M, N, K, silu_in, silu_out, (hipStream_t)stream);
  RET_HIP(e, "linear_small");
}

ia2p_status ia2p_profile_enable(ia2p_ctx* c, int on) {
  if (!c) return IA2P_ERR_INVALID;
  for (auto& r : c->recs) { c->evpool.push_back(r.e0); c->evpool.push_back(r.e1); }
  c->recs.clear();
  for (int k = 0; k < PK_NCLASS; ++k) { c->p_ms[k] = c->p_fl[k] = c->p_by[k] = c->p_pf[k] = 0; c->p_n[k] = 0; }
  for (int k = 0; k < PR_NREGION; ++k) { c->r_ms[k] = c->r_fl[k] = c->r_by[k] = 0; c->r_n[k] = 0; }
  for (int k = 0; k < ROLE_NROLE; ++k) { c->o_ms[k] = c->o_fl[k] = c->o_by[k] = 0; c->o_n[k] = 0; for (int q = 0; q < PK_NCLASS; ++q) { c->oc_ms[k][q] = 0; c->oc_n[k][q] = 0; } }
  c->prof = on != 0;
  return IA2P_OK;
}
int ia2p_profile_classes(void) { return PK_NCLASS; }
static void prof_fold(ia2p_ctx* c) {
  for (auto& r : c->recs) {     // fold finished records (synchronises on their stop events)
    float t = 0.f;
    (void)hipEventSynchronize(r.e1);
    (void)hipEventElapsedTime(&t, r.e0, r.e1);
    c->p_ms[r.k] += t; c->p_fl[r.k] += r.flops; c->p_by[r.k] += r.bytes; c->p_pf[r.k] += r.pf; c->p_n[r.k] += 1;
    const int g = r.region >= 0 && r.region < PR_NREGION ? r.region : PR_OTHER;
    c->r_ms[g] += t; c->r_fl[g] += r.flops; c->r_by[g] += r.bytes; c->r_n[g] += 1;
    const int o = r.role >= 0 && r.role < ROLE_NROLE ? r.role : ROLE_OTHER;
    c->o_ms[o] += t; c->o_fl[o] += r.flops; c->o_by[o] += r.bytes; c->o_n[o] += 1;
    c->oc_ms[o][r.k] += t; c->oc_n[o][r.k] += 1;
    c->evpool.push_back(r.e0); c->evpool.push_back(r.e1);
  }
  c->recs.clear();
}
ia2p_status ia2p_profile_read_region(ia2p_ctx* c, int region, int64_t* launches, double* ms, double* flops, double* bytes) {
  if (!c || region < 0 || region >= PR_NREGION) return IA2P_ERR_INVALID;
  prof_fold(c);
  if (launches) *launches = c->r_n[region];
  if (ms) *ms = c->r_ms[region];
  if (flops) *flops = c->r_fl[region];
  if (bytes) *bytes = c->r_by[region];
  return IA2P_OK;
}
// per-ROLE sums (engine_rt.h ROLE_*): the layer a launch implements, whatever kernel instantiation the plan table picked for it
int ia2p_profile_roles(void) { return ROLE_NROLE; }
ia2p_status ia2p_profile_read_role(ia2p_ctx* c, int role, char* name, int name_len, int64_t* launches, double* ms, double* flops, double* bytes) {
  if (!c || role < 0 || role >= ROLE_NROLE) return IA2P_ERR_INVALID;
  prof_fold(c);
  if (name && name_len > 0) { strncpy(name, role_name(role), name_len - 1); name[name_len - 1] = 0; }
  if (launches) *launches = c->o_n[role];
  if (ms) *ms = c->o_ms[role];
  if (flops) *flops = c->o_fl[role];
  if (bytes) *bytes = c->o_by[role];
  return IA2P_OK;
}
// ... and the share of kernel class k in it (which instantiations carried the role on this plan table)
ia2p_status ia2p_profile_read_role_class(ia2p_ctx* c, int role, int k, int64_t* launches, double* ms) {
  if (!c || role < 0 || role >= ROLE_NROLE || k < 0 || k >= PK_NCLASS) return IA2P_ERR_INVALID;
  prof_fold(c);
  if (launches) *launches = c->oc_n[role][k];
  if (ms) *ms = c->oc_ms[role][k];
  return IA2P_OK;
}
// bytes of next-contraction weights that the launches of class k streamed with their trailing prefetch workgroups (part of the class's HBM-side
// traffic that is NOT its own operands: bench.py separates the two when it prices the PMC figure)
ia2p_status ia2p_profile_read_prefetch(ia2p_ctx* c, int k, double* bytes) {
  if (!c || k < 0 || k >= PK_NCLASS || !bytes) return IA2P_ERR_INVALID;
  prof_fold(c);
  *bytes = c->p_pf[k];
  return IA2P_OK;
}
ia2p_status ia2p_profile_read(ia2p_ctx* c, int k, char* name, int name_len, int64_t* launches, double* ms, double* flops, double* bytes) {
  if (!c || k < 0 || k >= PK_NCLASS) return IA2P_ERR_INVALID;
  prof_fold(c);
  if (name && name_len > 0) { strncpy(name, prof_name(k), name_len - 1); name[name_len - 1] = 0; }
  if (launches) *launches = c->p_n[k];
  if (ms) *ms = c->p_ms[k];
  if (flops) *flops = c->p_fl[k];
  if (bytes) *bytes = c->p_by[k];
  return IA2P_OK;
}

}  // extern "C"
