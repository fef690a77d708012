// VAE executor and its C ABI (ia2p_vae_*): see include/ia2p.h and DESIGN.md §8. Runtime and operator wrappers: engine_rt.h / engine.hip.
#include "engine_rt.h"

// =====================================================================================================================
// VAE (diffusers AutoencoderKL, SDXL config): image -> latent moments before DDIM inversion, latent -> image after
// sampling (reference call sites: ddim/pnp_pipeline.py:190-204 via prepare_latents; ddim/sdxl_pipeline.py:859-871).
// First "next" row of SURVEY.md §8f. Same kernels as the UNet: implicit-GEMM 3x3 convs (the stride-2 downsample
// pads only after the image: `pad = 0`), GroupNorm+SiLU, the GEMM kernel for the single-head mid-block attention
// (head_dim = channels: scores are materialised per image, S = Q.K^T, row softmax, O = P.V with V^T produced directly
// as W_v.X^T), direct kernels at the 3/4/8-channel boundaries.
// Range extension instead of the reference's fp32 upcast (ddim/sdxl_pipeline.py:860-865 `upcast_vae`): the residual stream of the SDXL VAE
// grows past the fp16 maximum with the original checkpoint. Every tensor that carries the stream (and each resnet's conv1 output) is STORED
// multiplied by `ss` = cfg.stream_scale, a power of two <= 1 (2^-7 by default: range +-8.4e6): the producing epilogue multiplies its fp32
// accumulator by ss before the single rounding, GroupNorm reads the scaled tensor with eps * ss^2 (exactly the unscaled normalisation),
// linear consumers (shortcuts, resample convolutions) pass the scale through and scale only their bias. Powers of two: no rounding is
// added anywhere; fp16 keeps its 11 significant bits at every magnitude above 6e-5 / ss.
// =====================================================================================================================
struct VRes { int cin, cout; bool shortcut; size_t n1g, n1b, w1, b1, n2g, n2b, w2, b2, wsc, bsc; };
struct VMid { int c; VRes r0, r1; size_t gg, gb, wqk, bqk, wv, bv, wo, bo; };
struct VStage { std::vector<VRes> res; bool resample; size_t rw, rb; int rc; };

struct ia2p_vae : RunCtx {
  ia2p_vae_config cfg;
  size_t e_in_w, e_in_b, e_ng, e_nb, e_out_w, e_out_b, q_w, q_b, pq_w, pq_b, d_in_w, d_in_b, d_ng, d_nb, d_out_w, d_out_b;
  std::vector<VStage> enc, dec;
  VMid emid, dmid;
  float ss = 1.f;        // stream scale (see the header comment)
};

struct VPlanner {
  ia2p_vae* c;
  size_t cur = 0;
  size_t take(size_t e) { size_t o = cur; cur += (e + 127) & ~(size_t)127; return o; }
  size_t vec(const std::string& k, int n) { size_t o = take(n); c->params[k] = Param{o, (size_t)n, PK_COPY, 0, 0, false, false}; return o; }
  size_t mat(const std::string& k, int r, int cc) { size_t o = take((size_t)r * cc); c->params[k] = Param{o, (size_t)r * cc, PK_COPY, 0, 0, false, false}; return o; }
  size_t conv3(const std::string& k, int co, int ci, PKind kind = PK_CONV) { size_t o = take((size_t)co * ci * 9); c->params[k] = Param{o, (size_t)co * ci * 9, kind, co, ci, false, false}; return o; }
  size_t conv_in(const std::string& k, int co, int kt) { size_t o = take((size_t)co * 64); c->params[k] = Param{o, (size_t)co * kt, PK_PAD_CONV_IN, co, kt, false, false}; return o; }
  VRes resnet(const std::string& p, int cin, int cout) {
    VRes r;
    r.cin = cin; r.cout = cout; r.shortcut = cin != cout;
    r.n1g = vec(p + ".norm1.weight", cin); r.n1b = vec(p + ".norm1.bias", cin);
    r.w1 = conv3(p + ".conv1.weight", cout, cin); r.b1 = vec(p + ".conv1.bias", cout);
    r.n2g = vec(p + ".norm2.weight", cout); r.n2b = vec(p + ".norm2.bias", cout);
    r.w2 = conv3(p + ".conv2.weight", cout, cout); r.b2 = vec(p + ".conv2.bias", cout);
    r.wsc = r.bsc = 0;
    if (r.shortcut) { r.wsc = mat(p + ".conv_shortcut.weight", cout, cin); r.bsc = vec(p + ".conv_shortcut.bias", cout); }
    return r;
  }
  VMid mid(const std::string& p, int ch) {
    VMid m;
    m.c = ch;
    m.r0 = resnet(p + ".resnets.0", ch, ch);
    const std::string a = p + ".attentions.0";
    m.gg = vec(a + ".group_norm.weight", ch); m.gb = vec(a + ".group_norm.bias", ch);
    m.wqk = take((size_t)2 * ch * ch); m.bqk = take(2 * ch);      // to_q | to_k stacked: one projection GEMM
    c->params[a + ".to_q.weight"] = Param{m.wqk, (size_t)ch * ch, PK_COPY, 0, 0, false, false};
    c->params[a + ".to_k.weight"] = Param{m.wqk + (size_t)ch * ch, (size_t)ch * ch, PK_COPY, 0, 0, false, false};
    c->params[a + ".to_q.bias"] = Param{m.bqk, (size_t)ch, PK_COPY, 0, 0, false, false};
    c->params[a + ".to_k.bias"] = Param{m.bqk + ch, (size_t)ch, PK_COPY, 0, 0, false, false};
    m.wv = mat(a + ".to_v.weight", ch, ch); m.bv = vec(a + ".to_v.bias", ch);
    m.wo = mat(a + ".to_out.0.weight", ch, ch); m.bo = vec(a + ".to_out.0.bias", ch);
    m.r1 = resnet(p + ".resnets.1", ch, ch);
    return m;
  }
};

static ia2p_status vae_plan(ia2p_vae* c) {
  const ia2p_vae_config& g = c->cfg;
  const int n = g.n_blocks;
  if (n < 1 || n > IA2P_MAX_BLOCKS) return fail(c, IA2P_ERR_INVALID, "n_blocks %d out of range", n);
  for (int i = 0; i < n; ++i)
    if (g.block_out_channels[i] % 64 || g.block_out_channels[i] % g.norm_num_groups)
      return fail(c, IA2P_ERR_SHAPE, "block_out_channels[%d]=%d must be a multiple of 64 and of norm_num_groups", i, g.block_out_channels[i]);
  if (g.in_channels * 9 > 64 || g.latent_channels * 9 > 64 || 2 * g.latent_channels > 8 || g.out_channels > 8)
    return fail(c, IA2P_ERR_SHAPE, "boundary channel counts too large for the direct kernels");
  const int* ch = g.block_out_channels;
  const int z = g.latent_channels;
  VPlanner P{c};
  c->e_in_w = P.conv_in("encoder.conv_in.weight", ch[0], g.in_channels * 9); c->e_in_b = P.vec("encoder.conv_in.bias", ch[0]);
  int cprev = ch[0];
  for (int i = 0; i < n; ++i) {
    VStage st;
    for (int j = 0; j < g.layers_per_block; ++j)
      st.res.push_back(P.resnet("encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? cprev : ch[i], ch[i]));
    cprev = ch[i];
    st.resample = i != n - 1; st.rc = ch[i]; st.rw = st.rb = 0;
    if (st.resample) {
      st.rw = P.conv3("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv.weight", ch[i], ch[i]);
      st.rb = P.vec("encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv.bias", ch[i]);
    }
    c->enc.push_back(st);
  }
  c->emid = P.mid("encoder.mid_block", ch[n - 1]);
  c->e_ng = P.vec("encoder.conv_norm_out.weight", ch[n - 1]); c->e_nb = P.vec("encoder.conv_norm_out.bias", ch[n - 1]);
  c->e_out_w = P.conv3("encoder.conv_out.weight", 2 * z, ch[n - 1], PK_CONV_TAP); c->e_out_b = P.vec("encoder.conv_out.bias", 2 * z);
  c->q_w = P.mat("quant_conv.weight", 2 * z, 2 * z); c->q_b = P.vec("quant_conv.bias", 2 * z);
  c->pq_w = P.mat("post_quant_conv.weight", z, z); c->pq_b = P.vec("post_quant_conv.bias", z);
  c->d_in_w = P.conv_in("decoder.conv_in.weight", ch[n - 1], z * 9); c->d_in_b = P.vec("decoder.conv_in.bias", ch[n - 1]);
  c->dmid = P.mid("decoder.mid_block", ch[n - 1]);
  cprev = ch[n - 1];
  for (int i = 0; i < n; ++i) {
    VStage st;
    const int co = ch[n - 1 - i];
    for (int j = 0; j < g.layers_per_block + 1; ++j)
      st.res.push_back(P.resnet("decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? cprev : co, co));
    cprev = co;
    st.resample = i != n - 1; st.rc = co; st.rw = st.rb = 0;
    if (st.resample) {
      st.rw = P.conv3("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv.weight", co, co);
      st.rb = P.vec("decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv.bias", co);
    }
    c->dec.push_back(st);
  }
  c->d_ng = P.vec("decoder.conv_norm_out.weight", ch[0]); c->d_nb = P.vec("decoder.conv_norm_out.bias", ch[0]);
  c->d_out_w = P.conv3("decoder.conv_out.weight", g.out_channels, ch[0], PK_CONV_TAP); c->d_out_b = P.vec("decoder.conv_out.bias", g.out_channels);
  c->arena_elems = P.cur;
  return IA2P_OK;
}

static T2 vae_resnet(ia2p_vae* c, const VRes& r, T2 x, int B, int H, int Wd, float* gnp) {
  const float ss = c->ss, eps = c->cfg.norm_eps * ss * ss;     // x holds ss * (stream)
  const int HW = H * Wd;
  const long M = (long)B * HW;
  T2 n1 = wsalloc(c, (size_t)M * r.cin);
  op_gn(c, x.p, n1.p, r.n1g, r.n1b, B, HW, r.cin, eps, 1, gnp);
  T2 hh = wsalloc(c, (size_t)M * r.cout);
  c->ep_acc_scale = ss; c->ep_bias_scale = ss;                 // hh = ss * conv1(...)
  op_conv3(c, n1.p, B, H, Wd, r.cin, W_(c, r.w1), W_(c, r.b1), r.cout, 1, 0, nullptr, 0, nullptr, hh.p);
  wsfree(c, n1);
  T2 n2 = wsalloc(c, (size_t)M * r.cout);
  op_gn(c, hh.p, n2.p, r.n2g, r.n2b, B, HW, r.cout, eps, 1, gnp);
  wsfree(c, hh);
  T2 xs{(size_t)-1, nullptr};
  const half_t* resid = x.p;
  if (r.shortcut) {
    xs = wsalloc(c, (size_t)M * r.cout);
    c->ep_acc_scale = 1.f; c->ep_bias_scale = ss;               // linear in the (scaled) stream: only the bias is scaled
    op_gemm(c, x.p, r.cin, W_(c, r.wsc), W_(c, r.bsc), nullptr, 0, xs.p, r.cout, (int)M, r.cout, r.cin);
    resid = xs.p;
  }
  T2 out = wsalloc(c, (size_t)M * r.cout);
  c->ep_acc_scale = ss; c->ep_bias_scale = ss;                 // ss * (conv2 + b) + (scaled) residual
  op_conv3(c, n2.p, B, H, Wd, r.cout, W_(c, r.w2), W_(c, r.b2), r.cout, 1, 0, nullptr, 0, c->dry ? nullptr : resid, out.p);
  wsfree(c, n2);
  if (r.shortcut) wsfree(c, xs);
  return out;
}

static T2 vae_mid(ia2p_vae* c, const VMid& m, T2 x, int B, int H, int Wd, float* gnp) {
  const int HW = H * Wd, C = m.c;
  T2 a = vae_resnet(c, m.r0, x, B, H, Wd, gnp);
  wsfree(c, x);
  // attention (ldm AttnBlock, blocks.py:179-203): x + to_out(softmax(QK^T / sqrt(C)) V), one head of width C
  T2 n = wsalloc(c, (size_t)B * HW * C);
  const float ss = c->ss;
  op_gn(c, a.p, n.p, m.gg, m.gb, B, HW, C, c->cfg.norm_eps * ss * ss, 0, gnp);
  T2 qk = wsalloc(c, (size_t)HW * 2 * C), vt = wsalloc(c, (size_t)C * HW), sc = wsalloc(c, (size_t)HW * HW), o = wsalloc(c, (size_t)HW * C);
  T2 out = wsalloc(c, (size_t)B * HW * C);
  for (int b = 0; b < B; ++b) {
    const half_t* nb = c->dry ? nullptr : n.p + (size_t)b * HW * C;
    op_gemm(c, nb, C, W_(c, m.wqk), W_(c, m.bqk), nullptr, 0, qk.p, 2 * C, HW, 2 * C, C);                  // [q | k]
    op_gemm(c, W_(c, m.wv), C, nb, nullptr, nullptr, 0, vt.p, HW, C, HW, C);                              // V^T = W_v . X^T (bias added after P.V: rows of P sum to 1)
    c->ep_acc_scale = 1.0f / sqrtf((float)C);                                                              // scores leave the accumulator already scaled: |q.k| can pass the fp16 maximum
    op_gemm(c, qk.p, 2 * C, c->dry ? nullptr : qk.p + C, nullptr, nullptr, 0, sc.p, HW, HW, HW, C, 0, 0, 0, 0, 2 * C);   // S = Q . K^T / sqrt(C)
    {
      ProfScope ps(c, PK_ATTN, 0, 4.0 * HW * HW);
      CHECK_LAUNCH(c, ia2p_launch_softmax_rows(sc.p, HW, HW, HW, 1.0f, c->stream), "vae softmax");
    }
    op_gemm(c, sc.p, HW, vt.p, W_(c, m.bv), nullptr, 0, o.p, C, HW, C, HW);                                // O = P . V + b_v
    c->ep_acc_scale = ss; c->ep_bias_scale = ss;
    op_gemm(c, o.p, C, W_(c, m.wo), W_(c, m.bo), c->dry ? nullptr : a.p + (size_t)b * HW * C, C,
            c->dry ? nullptr : out.p + (size_t)b * HW * C, C, HW, C, C);                                  // + residual
  }
  wsfree(c, n); wsfree(c, qk); wsfree(c, vt); wsfree(c, sc); wsfree(c, o); wsfree(c, a);
  T2 r = vae_resnet(c, m.r1, out, B, H, Wd, gnp);
  wsfree(c, out);
  return r;
}

static ia2p_status vae_check(ia2p_vae* c, int B, int h, int w) {
  if (B < 1 || B > 1024) return fail(c, IA2P_ERR_SHAPE, "batch %d outside 1..1024", B);
  if (h < 1 || w < 1 || (long)h * w % 64 || (long)h * w > 16384) return fail(c, IA2P_ERR_SHAPE, "latent %dx%d: h*w must be a multiple of 64 and <= 16384 (mid-block attention over all pixels)", h, w);
  return IA2P_OK;
}

// z [B, zc, h, w] NCHW -> image [B, out, h*2^(n-1), w*2^(n-1)] NCHW
static ia2p_status vae_run_decode(ia2p_vae* c, const half_t* zin, half_t* img, int B, int h, int w) {
  const ia2p_vae_config& g = c->cfg;
  const int n = g.n_blocks, z = g.latent_channels, cm = g.block_out_channels[n - 1];
  T2 gnp = wsalloc(c, (size_t)B * 64 * g.norm_num_groups * 2 * 2);
  float* gp = (float*)gnp.p;
  T2 zq = wsalloc(c, (size_t)B * z * h * w);
  { ProfScope ps(c, PK_CONV_IN, 0, 0);
    CHECK_LAUNCH(c, ia2p_launch_conv1x1_nchw(zin, W_(c, c->pq_w), W_(c, c->pq_b), zq.p, B, z, z, (long)h * w, c->stream), "post_quant_conv");
  }
  T2 x = wsalloc(c, (size_t)B * h * w * cm);
  { ProfScope ps(c, PK_CONV_IN, 0, 0);
    CHECK_LAUNCH(c, ia2p_launch_conv_in(zq.p, W_(c, c->d_in_w), W_(c, c->d_in_b), x.p, B, z, h, w, cm, c->stream, c->ss), "decoder.conv_in");
  }
  wsfree(c, zq);
  x = vae_mid(c, c->dmid, x, B, h, w, gp);
  int H = h, Wd = w;
  for (int i = 0; i < n; ++i) {
    const VStage& st = c->dec[i];
    for (const VRes& r : st.res) { T2 y = vae_resnet(c, r, x, B, H, Wd, gp); wsfree(c, x); x = y; }
    if (st.resample) {
      T2 u = wsalloc(c, (size_t)B * (2 * H) * (2 * Wd) * st.rc);
      c->ep_acc_scale = 1.f; c->ep_bias_scale = c->ss;
      op_conv3(c, x.p, B, H, Wd, st.rc, W_(c, st.rw), W_(c, st.rb), st.rc, 1, 1, nullptr, 0, nullptr, u.p);
      wsfree(c, x); x = u; H *= 2; Wd *= 2;
    }
  }
  const int c0 = g.block_out_channels[0];
  T2 no = wsalloc(c, (size_t)B * H * Wd * c0);
  op_gn(c, x.p, no.p, c->d_ng, c->d_nb, B, H * Wd, c0, g.norm_eps * c->ss * c->ss, 1, gp);
  wsfree(c, x);
  { ProfScope ps(c, PK_CONV_OUT, 0, 0);
    CHECK_LAUNCH(c, ia2p_launch_conv_out(no.p, c0, W_(c, c->d_out_w), W_(c, c->d_out_b), img, B, c0, H, Wd, g.out_channels, c->stream), "decoder.conv_out");
  }
  wsfree(c, no); wsfree(c, gnp);
  return c->failed ? IA2P_ERR_HIP : IA2P_OK;
}

// image [B, in, H, W] NCHW -> moments [B, 2*zc, H/2^(n-1), W/2^(n-1)] NCHW (mean | logvar)
static ia2p_status vae_run_encode(ia2p_vae* c, const half_t* img, half_t* moments, int B, int Hi, int Wi) {
  const ia2p_vae_config& g = c->cfg;
  const int n = g.n_blocks, z = g.latent_channels;
  T2 gnp = wsalloc(c, (size_t)B * 64 * g.norm_num_groups * 2 * 2);
  float* gp = (float*)gnp.p;
  int H = Hi, Wd = Wi;
  T2 x = wsalloc(c, (size_t)B * H * Wd * g.block_out_channels[0]);
  { ProfScope ps(c, PK_CONV_IN, 0, 0);
    CHECK_LAUNCH(c, ia2p_launch_conv_in(img, W_(c, c->e_in_w), W_(c, c->e_in_b), x.p, B, g.in_channels, H, Wd, g.block_out_channels[0], c->stream, c->ss), "encoder.conv_in");
  }
  for (int i = 0; i < n; ++i) {
    const VStage& st = c->enc[i];
    for (const VRes& r : st.res) { T2 y = vae_resnet(c, r, x, B, H, Wd, gp); wsfree(c, x); x = y; }
    if (st.resample) {       // F.pad(x, (0,1,0,1)) + conv stride 2 padding 0 (ldm Downsample, blocks.py:73-77)
      T2 d = wsalloc(c, (size_t)B * (H / 2) * (Wd / 2) * st.rc);
      c->ep_acc_scale = 1.f; c->ep_bias_scale = c->ss;
      op_conv3(c, x.p, B, H, Wd, st.rc, W_(c, st.rw), W_(c, st.rb), st.rc, 2, 0, nullptr, 0, nullptr, d.p, 0);
      wsfree(c, x); x = d; H /= 2; Wd /= 2;
    }
  }
  x = vae_mid(c, c->emid, x, B, H, Wd, gp);
  const int cm = g.block_out_channels[n - 1];
  T2 no = wsalloc(c, (size_t)B * H * Wd * cm);
  op_gn(c, x.p, no.p, c->e_ng, c->e_nb, B, H * Wd, cm, g.norm_eps * c->ss * c->ss, 1, gp);
  wsfree(c, x);
  T2 m0 = wsalloc(c, (size_t)B * 2 * z * H * Wd);
  { ProfScope ps(c, PK_CONV_OUT, 0, 0);
    CHECK_LAUNCH(c, ia2p_launch_conv_out(no.p, cm, W_(c, c->e_out_w), W_(c, c->e_out_b), m0.p, B, cm, H, Wd, 2 * z, c->stream), "encoder.conv_out");
    CHECK_LAUNCH(c, ia2p_launch_conv1x1_nchw(m0.p, W_(c, c->q_w), W_(c, c->q_b), moments, B, 2 * z, 2 * z, (long)H * Wd, c->stream), "quant_conv");
  }
  wsfree(c, no); wsfree(c, m0); wsfree(c, gnp);
  return c->failed ? IA2P_ERR_HIP : IA2P_OK;
}

extern "C" {

ia2p_status ia2p_vae_create(const ia2p_vae_config* cfg, ia2p_vae** out) {
  if (!cfg || !out) return fail(nullptr, IA2P_ERR_INVALID, "ia2p_vae_create: null argument");
  ia2p_vae* c = new ia2p_vae();
  c->cfg = *cfg;
  c->groups = cfg->norm_num_groups;
  c->ss = cfg->stream_scale > 0.f ? cfg->stream_scale : 1.f;
  {
    int ex = 0;
    if (std::frexp(c->ss, &ex) != 0.5f || c->ss > 1.f || c->ss < 1.0f / 65536.f) { delete c; *out = nullptr; return fail(nullptr, IA2P_ERR_INVALID, "stream_scale must be a power of two in [2^-16, 1]"); }
  }
  ia2p_status st = vae_plan(c);
  if (st != IA2P_OK) { g_err = c->err; delete c; *out = nullptr; return st; }
  c->failed = false;
  *out = c;
  return IA2P_OK;
}
void ia2p_vae_destroy(ia2p_vae* c) { delete c; }
const char* ia2p_vae_last_error(ia2p_vae* c) { return c ? c->err.c_str() : g_err.c_str(); }
size_t ia2p_vae_arena_bytes(ia2p_vae* c) { return c ? c->arena_elems * sizeof(half_t) : 0; }
ia2p_status ia2p_vae_bind_arena(ia2p_vae* c, void* dev, size_t bytes) { return rc_bind_arena(c, dev, bytes); }
ia2p_status ia2p_vae_load_tensor(ia2p_vae* c, const char* key, const void* src, const int64_t* shape, int ndim, void* stream) {
  return rc_load_tensor(c, key, src, shape, ndim, stream);
}
ia2p_status ia2p_vae_finalize_weights(ia2p_vae* c) { return rc_finalize(c, "VAE"); }

// h, w: LATENT height/width for both directions (image = latent * 2^(n_blocks-1))
size_t ia2p_vae_workspace_bytes(ia2p_vae* c, int B, int h, int w, int decode) {
  if (!c || vae_check(c, B, h, w) != IA2P_OK) return 0;
  const int f = 1 << (c->cfg.n_blocks - 1);
  c->dry = true; c->failed = false; c->record = false;
  c->ws.reset((size_t)1 << 46); c->ws_base = nullptr;
  if (decode) (void)vae_run_decode(c, nullptr, nullptr, B, h, w); else (void)vae_run_encode(c, nullptr, nullptr, B, h * f, w * f);
  c->dry = false;
  return c->failed ? 0 : c->ws.high + 256;
}

static ia2p_status vae_run(ia2p_vae* c, void* stream, const void* in, void* out, int B, int h, int w, void* ws, size_t ws_bytes, bool decode) {
  if (!c || !in || !out || !ws) return fail(c, IA2P_ERR_INVALID, "vae: null argument");
  if (!c->finalized) return fail(c, IA2P_ERR_STATE, "vae called before weights were finalized");
  ia2p_status st = vae_check(c, B, h, w);
  if (st != IA2P_OK) return st;
  if (!zero_page()) return fail(c, IA2P_ERR_HIP, "cannot allocate zero page");
  const int f = 1 << (c->cfg.n_blocks - 1);
  const uintptr_t base = ((uintptr_t)ws + 255) & ~(uintptr_t)255;
  const size_t usable = ws_bytes - (base - (uintptr_t)ws);
  const int key = decode ? 1 : 2;
  if (c->wseq_key != key) {
    c->wseq.clear();
    c->dry = true; c->record = true; c->failed = false;
    c->ws.reset((size_t)1 << 46); c->ws_base = nullptr;
    if (decode) (void)vae_run_decode(c, nullptr, nullptr, B, h, w); else (void)vae_run_encode(c, nullptr, nullptr, B, h * f, w * f);
    c->dry = false; c->record = false; c->wseq_key = key;
  }
  c->widx = 0; c->dry = false; c->failed = false; c->stream = (hipStream_t)stream;
  c->ws.reset(usable); c->ws_base = (char*)base;
  st = decode ? vae_run_decode(c, (const half_t*)in, (half_t*)out, B, h, w) : vae_run_encode(c, (const half_t*)in, (half_t*)out, B, h * f, w * f);
  if (c->failed && st == IA2P_OK) st = IA2P_ERR_HIP;
  if (c->failed && c->err == "workspace too small") st = IA2P_ERR_NOMEM;
  return st;
}
ia2p_status ia2p_vae_decode(ia2p_vae* c, void* stream, const void* latents, void* image, int B, int h, int w, void* ws, size_t ws_bytes) {
  return vae_run(c, stream, latents, image, B, h, w, ws, ws_bytes, true);
}
ia2p_status ia2p_vae_encode(ia2p_vae* c, void* stream, const void* image, void* moments, int B, int h, int w, void* ws, size_t ws_bytes) {
  return vae_run(c, stream, image, moments, B, h, w, ws, ws_bytes, false);
}

}  // extern "C"
