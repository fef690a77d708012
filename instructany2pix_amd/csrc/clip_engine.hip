// CLIP text-encoder executor and its C ABI (ia2p_clip_*): see include/ia2p.h and DESIGN.md §8. Runtime and operator wrappers: engine_rt.h / engine.hip.
#include "engine_rt.h"

// =====================================================================================================================
// CLIP text encoders: the conditioning side of the path, on the same GEMM kernel.
// transformers CLIPTextModel / CLIPTextModelWithProjection (reference encode_prompt, ddim/sdxl_pipeline.py:202-395).
// Per layer: 4 GEMMs (QKV with layer_norm1 folded, out-proj + residual, fc1 with layer_norm2 folded + activation,
// fc2 + residual) and one causal attention launch; no LayerNorm launches (row statistics travel as in the UNet).
// =====================================================================================================================
struct CLayer { size_t ln1g, ln1b, wqkv, bqkv, wo, bo, ln2g, ln2b, w1, b1, w2, b2, fqkv, cs1, lb1, f1, cs2, lb2; };
struct ia2p_clip : RunCtx {
  ia2p_clip_config cfg;
  size_t tok, pos, lnfg, lnfb, wproj;
  std::vector<CLayer> layers;
};

static ia2p_status clip_plan(ia2p_clip* c) {
  const ia2p_clip_config& g = c->cfg;
  const int H = g.hidden_size, I = g.intermediate_size;
  if (g.num_layers < 1 || H % 64 || g.num_heads * 64 != H || I % 64 || g.vocab_size < 0 || g.max_positions < 1 || g.max_positions > 128)
    return fail(c, IA2P_ERR_SHAPE, "clip: hidden %d must be heads*64, intermediate %d a multiple of 64, 1..128 positions", H, I);
  if (g.hidden_act < 1 || g.hidden_act > 3) return fail(c, IA2P_ERR_INVALID, "clip: hidden_act must be 1 (gelu), 2 (quick_gelu) or 3 (gelu_new)");
  if (g.projection_dim < 0 || g.projection_dim % 8) return fail(c, IA2P_ERR_SHAPE, "clip: projection_dim %d", g.projection_dim);
  size_t cur = 0;
  auto take = [&](size_t e) { size_t o = cur; cur += (e + 127) & ~(size_t)127; return o; };
  auto reg = [&](const std::string& k, size_t off, size_t n) { c->params[k] = Param{off, n, PK_COPY, 0, 0, false, false}; };
  auto par = [&](const std::string& k, size_t n) { size_t o = take(n); reg(k, o, n); return o; };
  const std::string tm = "text_model.";
  c->tok = g.vocab_size ? par(tm + "embeddings.token_embedding.weight", (size_t)g.vocab_size * H) : 0;   // 0: inputs_embeds only (the GPT-2 prior)
  c->pos = par(tm + "embeddings.position_embedding.weight", (size_t)g.max_positions * H);
  for (int i = 0; i < g.num_layers; ++i) {
    const std::string p = tm + "encoder.layers." + std::to_string(i) + ".";
    CLayer l;
    l.ln1g = par(p + "layer_norm1.weight", H); l.ln1b = par(p + "layer_norm1.bias", H);
    l.wqkv = take((size_t)3 * H * H); l.bqkv = take((size_t)3 * H);
    const char* nm[3] = {"q_proj", "k_proj", "v_proj"};
    for (int j = 0; j < 3; ++j) {
      reg(p + "self_attn." + nm[j] + ".weight", l.wqkv + (size_t)j * H * H, (size_t)H * H);
      reg(p + "self_attn." + nm[j] + ".bias", l.bqkv + (size_t)j * H, H);
    }
    l.wo = par(p + "self_attn.out_proj.weight", (size_t)H * H); l.bo = par(p + "self_attn.out_proj.bias", H);
    l.ln2g = par(p + "layer_norm2.weight", H); l.ln2b = par(p + "layer_norm2.bias", H);
    l.w1 = par(p + "mlp.fc1.weight", (size_t)I * H); l.b1 = par(p + "mlp.fc1.bias", I);
    l.w2 = par(p + "mlp.fc2.weight", (size_t)H * I); l.b2 = par(p + "mlp.fc2.bias", H);
    l.fqkv = take((size_t)3 * H * H); l.cs1 = take((size_t)2 * 3 * H); l.lb1 = take((size_t)2 * 3 * H);
    l.f1 = take((size_t)I * H); l.cs2 = take((size_t)2 * I); l.lb2 = take((size_t)2 * I);
    c->layers.push_back(l);
  }
  c->lnfg = par(tm + "final_layer_norm.weight", H); c->lnfb = par(tm + "final_layer_norm.bias", H);
  c->wproj = g.projection_dim ? par("text_projection.weight", (size_t)g.projection_dim * H) : 0;
  c->arena_elems = cur;
  return IA2P_OK;
}

static ia2p_status clip_fold(ia2p_clip* c, hipStream_t stream = nullptr, bool sync = true) {
  const int H = c->cfg.hidden_size, I = c->cfg.intermediate_size;
  hipError_t e = hipSuccess;
  auto Hp = [&](size_t off) { return c->arena + off; };
  auto Fp = [&](size_t off) { return (float*)(c->arena + off); };
  for (const CLayer& l : c->layers) {
    if (e == hipSuccess) e = ia2p_launch_fold_ln(Hp(l.wqkv), Hp(l.ln1g), Hp(l.ln1b), Hp(l.bqkv), Hp(l.fqkv), Fp(l.cs1), Fp(l.lb1), 3 * H, H, stream);
    if (e == hipSuccess) e = ia2p_launch_fold_ln(Hp(l.w1), Hp(l.ln2g), Hp(l.ln2b), Hp(l.b1), Hp(l.f1), Fp(l.cs2), Fp(l.lb2), I, H, stream);
  }
  if (e == hipSuccess && sync) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) return fail_hip(c, e, "clip LayerNorm folding");
  c->fold_dirty = false;
  return IA2P_OK;
}

static ia2p_status clip_run(ia2p_clip* c, const int* ids, const half_t* embeds, int B, int T, half_t* hid2, half_t* last, half_t* pooled) {
  const ia2p_clip_config& g = c->cfg;
  const int H = g.hidden_size, I = g.intermediate_size, M = B * T, L = g.num_layers;
  auto Fp = [&](size_t off) { return (const float*)(c->arena + off); };
  T2 x = wsalloc(c, (size_t)M * H), qkv = wsalloc(c, (size_t)M * 3 * H), att = wsalloc(c, (size_t)M * H), ff = wsalloc(c, (size_t)M * I);
  T2 stt = wsalloc(c, (size_t)M * ((H + 63) / 64) * 2 * 2);
  float* st = (float*)stt.p;
  int slots = 1;
  CHECK_LAUNCH(c, ia2p_launch_clip_embed(ids, W_(c, c->tok), embeds, W_(c, c->pos), x.p, st, M, T, H, g.vocab_size, c->stream), "clip embeddings");
  const bool need_last = last || pooled;
  const int run_layers = need_last ? L : L - 1;
  for (int i = 0; i < run_layers; ++i) {
    const CLayer& l = c->layers[i];
    if (i == L - 1 && hid2 && !c->dry && !c->failed) {     // hidden_states[-2]: what the last layer reads
      hipError_t e = hipMemcpyAsync(hid2, x.p, (size_t)M * H * sizeof(half_t), hipMemcpyDeviceToDevice, c->stream);
      if (e != hipSuccess) fail_hip(c, e, "clip");
    }
    {
      const LnIn ln{st, slots, Fp(l.cs1), Fp(l.lb1), g.layer_norm_eps};
      op_gemm(c, x.p, H, W_(c, l.fqkv), nullptr, nullptr, 0, qkv.p, 3 * H, M, 3 * H, H, 0, 0, 0, 0, 0, &ln);
    }
    CHECK_LAUNCH(c, ia2p_launch_causal_attention_small(qkv.p, att.p, B, T, g.num_heads, c->stream), "clip attention");
    op_gemm(c, att.p, H, W_(c, l.wo), W_(c, l.bo), x.p, H, x.p, H, M, H, H, 0, 0, 0, 0, 0, nullptr, st, &slots);
    {
      const LnIn ln{st, slots, Fp(l.cs2), Fp(l.lb2), g.layer_norm_eps};
      op_gemm(c, x.p, H, W_(c, l.f1), nullptr, nullptr, 0, ff.p, I, M, I, H, 0, 0, 0, 0, 0, &ln, nullptr, nullptr, g.hidden_act);
    }
    op_gemm(c, ff.p, I, W_(c, l.w2), W_(c, l.b2), x.p, H, x.p, H, M, H, I, 0, 0, 0, 0, 0, nullptr, st, &slots);
  }
  if (!need_last && hid2 && !c->dry && !c->failed) {
    hipError_t e = hipMemcpyAsync(hid2, x.p, (size_t)M * H * sizeof(half_t), hipMemcpyDeviceToDevice, c->stream);
    if (e != hipSuccess) fail_hip(c, e, "clip");
  }
  if (last) CHECK_LAUNCH(c, ia2p_launch_layernorm(x.p, H, last, H, W_(c, c->lnfg), W_(c, c->lnfb), M, H, g.layer_norm_eps, c->stream), "clip final_layer_norm");
  if (pooled) {
    T2 pr = wsalloc(c, (size_t)B * H);
    half_t* dst = g.projection_dim ? pr.p : pooled;
    CHECK_LAUNCH(c, ia2p_launch_clip_pool(ids, x.p, W_(c, c->lnfg), W_(c, c->lnfb), dst, B, T, H, g.eos_token_id, g.layer_norm_eps, c->stream), "clip pooling");
    if (g.projection_dim)
      for (int r0 = 0; r0 < B; r0 += 16) {
        const int rows = std::min(16, B - r0);
        CHECK_LAUNCH(c, ia2p_launch_linear_small(c->dry ? nullptr : pr.p + (size_t)r0 * H, H, W_(c, c->wproj), nullptr, nullptr, 0,
                                                 c->dry ? nullptr : pooled + (size_t)r0 * g.projection_dim, g.projection_dim, rows, g.projection_dim, H, 0, 0, c->stream),
                     "clip text_projection");
      }
    wsfree(c, pr);
  }
  wsfree(c, stt); wsfree(c, ff); wsfree(c, att); wsfree(c, qkv); wsfree(c, x);
  return c->failed ? IA2P_ERR_HIP : IA2P_OK;
}

ia2p_status ia2p_clip_create(const ia2p_clip_config* cfg, ia2p_clip** out) {
  if (!cfg || !out) return fail(nullptr, IA2P_ERR_INVALID, "ia2p_clip_create: null argument");
  ia2p_clip* c = new ia2p_clip();
  c->cfg = *cfg;
  if (c->cfg.layer_norm_eps <= 0.f) c->cfg.layer_norm_eps = 1e-5f;
  ia2p_status st = clip_plan(c);
  if (st != IA2P_OK) { g_err = c->err; delete c; *out = nullptr; return st; }
  c->failed = false;
  *out = c;
  return IA2P_OK;
}
void ia2p_clip_destroy(ia2p_clip* c) { delete c; }
const char* ia2p_clip_last_error(ia2p_clip* c) { return c ? c->err.c_str() : g_err.c_str(); }
size_t ia2p_clip_arena_bytes(ia2p_clip* c) { return c ? c->arena_elems * sizeof(half_t) : 0; }
ia2p_status ia2p_clip_bind_arena(ia2p_clip* c, void* dev, size_t bytes) { return rc_bind_arena(c, dev, bytes); }
ia2p_status ia2p_clip_load_tensor(ia2p_clip* c, const char* key, const void* src, const int64_t* shape, int ndim, void* stream) {
  return rc_load_tensor(c, key, src, shape, ndim, stream);
}
ia2p_status ia2p_clip_finalize_weights(ia2p_clip* c) {
  const ia2p_status st = rc_finalize(c, "CLIP text encoder");
  return st == IA2P_OK ? clip_fold(c) : st;
}
static ia2p_status clip_check(ia2p_clip* c, int B, int T) {
  if (B < 1 || T < 1 || T > c->cfg.max_positions) return fail(c, IA2P_ERR_SHAPE, "clip: B=%d, T=%d (1..%d tokens)", B, T, c->cfg.max_positions);
  return IA2P_OK;
}
size_t ia2p_clip_workspace_bytes(ia2p_clip* c, int B, int T) {
  if (!c || clip_check(c, B, T) != IA2P_OK) return 0;
  c->dry = true; c->failed = false; c->record = false;
  c->ws.reset((size_t)1 << 46); c->ws_base = nullptr;
  (void)clip_run(c, nullptr, nullptr, B, T, nullptr, (half_t*)1, c->cfg.vocab_size ? (half_t*)1 : nullptr);
  c->dry = false;
  return c->failed ? 0 : c->ws.high + 256;
}
static ia2p_status clip_encode_impl(ia2p_clip* c, void* stream, const int32_t* ids, const half_t* embeds, int B, int T, void* hid2, void* last, void* pooled,
                                    void* ws, size_t ws_bytes) {
  if (!c->finalized) return fail(c, IA2P_ERR_STATE, "clip_encode before weights were finalized");
  ia2p_status st = clip_check(c, B, T);
  if (st != IA2P_OK) return st;
  if (!zero_page()) return fail(c, IA2P_ERR_HIP, "cannot allocate zero page");
  if (c->fold_dirty) {            // a tensor was reloaded after finalize: re-derive the folded LayerNorm copies, stream-ordered
    st = clip_fold(c, (hipStream_t)stream, false);
    if (st != IA2P_OK) return st;
  }
  const uintptr_t base = ((uintptr_t)ws + 255) & ~(uintptr_t)255;
  const size_t usable = ws_bytes - (base - (uintptr_t)ws);
  const int key = (last || pooled) ? 1 : 2;
  if (c->wseq_key != key) {
    c->wseq.clear();
    c->dry = true; c->record = true; c->failed = false;
    c->ws.reset((size_t)1 << 46); c->ws_base = nullptr;
    (void)clip_run(c, nullptr, nullptr, B, T, nullptr, key == 1 ? (half_t*)1 : nullptr, nullptr);
    c->dry = false; c->record = false; c->wseq_key = key;
  }
  c->widx = 0; c->dry = false; c->failed = false; c->stream = (hipStream_t)stream;
  c->ws.reset(usable); c->ws_base = (char*)base;
  st = clip_run(c, ids, embeds, B, T, (half_t*)hid2, (half_t*)last, (half_t*)pooled);
  if (c->failed && st == IA2P_OK) st = IA2P_ERR_HIP;
  if (c->failed && c->err == "workspace too small") st = IA2P_ERR_NOMEM;
  return st;
}
ia2p_status ia2p_clip_encode(ia2p_clip* c, void* stream, const int32_t* ids, int B, int T, void* hid2, void* last, void* pooled, void* ws, size_t ws_bytes) {
  if (!c || !ids || !ws || (!hid2 && !last && !pooled)) return fail(c, IA2P_ERR_INVALID, "clip_encode: null argument");
  if (!c->cfg.vocab_size) return fail(c, IA2P_ERR_STATE, "clip_encode: this model was created without a token embedding (vocab_size 0); use ia2p_clip_encode_embeds");
  return clip_encode_impl(c, stream, ids, nullptr, B, T, hid2, last, pooled, ws, ws_bytes);
}
ia2p_status ia2p_clip_encode_embeds(ia2p_clip* c, void* stream, const void* inputs_embeds, int B, int T, void* hid2, void* last, void* ws, size_t ws_bytes) {
  if (!c || !inputs_embeds || !ws || (!hid2 && !last)) return fail(c, IA2P_ERR_INVALID, "clip_encode_embeds: null argument");
  return clip_encode_impl(c, stream, nullptr, (const half_t*)inputs_embeds, B, T, hid2, last, nullptr, ws, ws_bytes);
}
