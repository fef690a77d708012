"""The oracle pinned against the golden vectors generated from the reference's own files
(tests/golden/gen_goldens.py). CPU only."""
import numpy as np
import pytest
import torch

import oracle
from oracle.unet_ref import Attention, ResnetBlock2D, Transformer2DModel, sinusoid

T = torch.from_numpy


def _attn(d, ctx_dim):
    dim = d["to_q"].shape[0]
    a = Attention(dim, ctx_dim, int(d["heads"]), dim // int(d["heads"]))
    a.to_q.weight.data = T(d["to_q"]); a.to_k.weight.data = T(d["to_k"]); a.to_v.weight.data = T(d["to_v"])
    a.to_out[0].weight.data = T(d["to_out_w"]); a.to_out[0].bias.data = T(d["to_out_b"])
    return a


@torch.no_grad()
def test_g1_self_attention_processor(golden):
    d = golden("attn_self.npz")
    a = _attn(d, None)
    o = oracle.AttnProcessor2_0Ref()(a, T(d["x"]))
    assert np.abs(o.numpy() - d["out_2_0"]).max() < 2e-6      # reference AttnProcessor2_0 :205-279
    assert np.abs(o.numpy() - d["out_bmm"]).max() < 2e-6      # reference AttnProcessor (bmm twin) :19-79


@torch.no_grad()
@pytest.mark.parametrize("L", [81, 77])
def test_g2_ip_attention_processor(golden, L):
    d = golden("attn_ip.npz")
    a = _attn(d, d["to_k"].shape[1])
    p = oracle.IPAttnProcessor2_0Ref(d["to_q"].shape[0], d["to_k"].shape[1], num_tokens=4)
    p.to_k_ip.weight.data = T(d["to_k_ip"]); p.to_v_ip.weight.data = T(d["to_v_ip"])
    for s in (0.0, 0.5, 1.0):
        p.scale = s
        o = p(a, T(d["x"]), encoder_hidden_states=T(d[f"ctx{L}"]))
        assert np.abs(o.numpy() - d[f"out{L}_s{s}"]).max() < 2e-6       # IPAttnProcessor2_0 :310-412
        assert np.abs(o.numpy() - d[f"outbmm{L}_s{s}"]).max() < 2e-6    # IPAttnProcessor :107-188
    assert np.abs(p.attn_map.numpy() - d[f"attn_map{L}"]).max() < 1e-5    # side effect :390-391


@torch.no_grad()
def test_g3_image_proj(golden):
    d = golden("image_proj.npz")
    m = oracle.ImageProjModelRef(cross_attention_dim=64, clip_embeddings_dim=48, clip_extra_context_tokens=4)
    m.proj.weight.data = T(d["proj_weight"]); m.proj.bias.data = T(d["proj_bias"])
    m.norm.weight.data = T(d["norm_weight"]); m.norm.bias.data = T(d["norm_bias"]); m.raw_embed.data = T(d["raw_embed"])
    emb = T(d["emb"])
    for mode in ("global", "local", "both"):
        for sl in (1.0, 0.5):
            o = m(emb, mode, scales=(1.0, sl))
            assert np.abs(o.numpy() - d[f"out_{mode}_{sl}"]).max() < 2e-6
    assert np.abs(m(torch.zeros_like(emb), "global").numpy() - d["out_zero_global"]).max() < 2e-6
    with pytest.raises(AssertionError):
        m(emb, "bogus")


def test_g4_backward_ddim_and_schedule(golden):
    d = golden("backward_ddim.npz")
    s = oracle.DDIMSchedulerRef()
    for n in (20, 25, 50):
        s.set_timesteps(n)
        assert np.array_equal(s.timesteps.numpy(), d[f"timesteps{n}"])
        lat = T(d["x0"]).clone()
        prev = None
        for i, t in enumerate(reversed(s.timesteps)):
            a_p = s.alphas_cumprod[prev] if prev is not None else s.final_alpha_cumprod
            lat = oracle.backward_ddim(lat, s.alphas_cumprod[t], a_p, T(d["eps"][i]))
            prev = t
            assert np.abs(lat.numpy() - d[f"traj{n}"][i]).max() <= 1e-6 * max(1.0, np.abs(d[f"traj{n}"][i]).max())
    h = oracle.backward_ddim(T(d["x0"]).half(), s.alphas_cumprod[501], s.alphas_cumprod[481], T(d["eps"][0]).half())
    assert h.dtype == torch.float16 and np.array_equal(h.float().numpy(), d["half_step"])


def test_g4_step_inverts_backward_ddim():
    """DDIM `step` (diffusers formula) and the reference's `_backward_ddim` are exact inverses for fixed eps."""
    s = oracle.DDIMSchedulerRef()
    s.set_timesteps(50)
    g = torch.Generator().manual_seed(0)
    x, e = torch.randn(1, 4, 8, 8, generator=g, dtype=torch.float64), torch.randn(1, 4, 8, 8, generator=g, dtype=torch.float64)
    t = 501
    a_t, a_p = s.alphas_cumprod[t].double(), s.alphas_cumprod[t - 20].double()
    up = oracle.backward_ddim(x, a_t, a_p, e)
    down = a_p ** 0.5 * (up - (1 - a_t) ** 0.5 * e) / a_t ** 0.5 + (1 - a_p) ** 0.5 * e
    assert (down - x).abs().max() < 1e-12
    # and the scheduler's own step agrees with that closed form
    assert (s.step(e.float(), t, up.float()) - x.float()).abs().max() < 1e-5


def test_g5_schedule_tables(golden):
    d = golden("schedule.npz")
    s = oracle.DDIMSchedulerRef()
    assert np.abs(s.betas.numpy() - d["betas"]).max() < 1e-8
    assert np.abs(s.alphas_cumprod.numpy() - d["alphas_cumprod"]).max() < 2e-6
    for n in (20, 25, 50):
        s.set_timesteps(n)
        ts = s.timesteps.numpy()[::-1]                          # ldm lists ascending
        assert np.array_equal(ts, d[f"ts{n}"])
        a = s.alphas_cumprod.numpy()
        assert np.abs(a[ts] - d[f"alphas{n}"]).max() < 2e-6
        prev = [float(s.final_alpha_cumprod)] + [a[t - 1000 // n] for t in ts[1:]]
        assert np.abs(np.array(prev) - d[f"alphas_prev{n}"]).max() < 2e-6


@torch.no_grad()
def test_g6_ldm_blocks(golden):
    d = golden("ldm_blocks.npz")
    assert np.abs(sinusoid(T(d["temb_t"]), 320).numpy() - d["temb_320"]).max() < 1e-6
    assert np.abs(sinusoid(T(d["temb_t"]), 256).numpy() - d["temb_256"]).max() < 1e-6
    rb = ResnetBlock2D(64, 96, 48, 32, 1e-6)                     # ldm Normalize eps 1e-6 (blocks.py:38-39)
    ren = {"temb_proj": "time_emb_proj", "nin_shortcut": "conv_shortcut"}
    sd = {}
    for k in rb.state_dict():
        src = k
        for a, b in ren.items():
            src = src.replace(b, a)
        sd[k] = T(d["rb_" + src.replace(".", "_")])
    rb.load_state_dict(sd)
    o = rb(T(d["rb_x"]), T(d["rb_temb"]))
    assert np.abs(o.numpy() - d["rb_out"]).max() < 2e-5
    st = Transformer2DModel(64, 1, 64, 2, 40, 32)
    sd = {}
    for k in st.state_dict():
        v = T(d["st_" + k.replace(".", "_")])
        sd[k] = v.reshape(v.shape[0], v.shape[1]) if k in ("proj_in.weight", "proj_out.weight") else v   # conv1x1 == Linear
    st.load_state_dict(sd)
    o = st(T(d["st_x"]), T(d["st_ctx"]))
    assert np.abs(o.numpy() - d["st_out"]).max() < 5e-5


def test_g7_misc(golden):
    d = golden("misc.npz")
    a, b = T(d["pa"]), T(d["pb"])
    assert np.abs(oracle.polar_interpolate(a, b, 0.7).numpy() - d["polar_07"]).max() < 1e-6
    assert np.abs(oracle.polar_interpolate(a, b, 0.3).numpy() - d["polar_03"]).max() < 1e-6
    assert np.array_equal(oracle.polar_interpolate(a.half(), b.half(), 0.7).float().numpy(), d["polar_half"])
    ids = oracle.get_add_time_ids((1024, 1024), (0, 0), (1024, 1024), 256, 1280, 2816)
    assert np.array_equal(ids.numpy(), d["time_ids"]) and np.array_equal(ids.numpy(), d["neg_time_ids"])
    assert int(d["bad_dim_raises"]) == 1
    with pytest.raises(ValueError):
        oracle.get_add_time_ids((1024, 1024), (0, 0), (1024, 1024), 256, 1280, 2560)


@torch.no_grad()
def test_g8_unet_with_reference_processors(golden):
    """Oracle UNet + oracle processors == oracle UNet tree + the REFERENCE processor classes."""
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    d = golden("unet_refprocs.npz")
    cfg = tiny()
    sd = synthetic_state_dict(unet_param_specs(cfg), seed=7)
    ipsd = synthetic_state_dict(ip_adapter_specs(cfg, 64)["ip_adapter"], seed=7)
    net = oracle.build_unet(cfg, sd, ipsd)
    added = dict(text_embeds=T(d["text_embeds"]), time_ids=T(d["time_ids"]))
    for L in (81, 77):
        for t in (981, 1):
            for s in (1.0, 0.5):
                for p in net.attn_processors.values():
                    if hasattr(p, "scale"):
                        p.scale = s
                o = net(T(d["x"]), t, T(d[f"ctx{L}"]), added_cond_kwargs=added)[0]
                ref = d[f"out_L{L}_t{t}_s{s}"]
                assert np.abs(o.numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def _ldm_to_diffusers_vae(d, cfg):
    """Key renaming ldm (blocks.py Encoder/Decoder) -> diffusers AutoencoderKL, 1x1 conv -> Linear for the attention."""
    n = len(cfg.block_out_channels)
    sd = {}
    g = torch.Generator().manual_seed(int(d["seed"]))        # weights re-drawn exactly as gen_goldens.py drew them
    for k, shp in zip(d["names"].tolist(), d["shapes"].tolist()):
        shape = tuple(int(x) for x in shp.split(","))
        v = torch.randn(shape, generator=g)
        v = v * (0.5 / max(1, v[0].numel()) ** 0.5 if v.ndim > 1 else 0.2)
        side, rest = ("encoder", k[4:]) if k.startswith("enc.") else ("decoder", k[4:])
        parts = rest.split(".")
        if parts[0] in ("down", "up"):
            lvl = int(parts[1])
            blk = lvl if parts[0] == "down" else n - 1 - lvl        # ldm keeps `up` indexed by resolution level (blocks.py:524)
            grp = "down_blocks" if parts[0] == "down" else "up_blocks"
            if parts[2] == "block":
                name = f"{side}.{grp}.{blk}.resnets.{parts[3]}." + ".".join(parts[4:])
            else:
                name = f"{side}.{grp}.{blk}.{'downsamplers' if parts[2] == 'downsample' else 'upsamplers'}.0." + ".".join(parts[3:])
        elif parts[0] == "mid":
            if parts[1] in ("block_1", "block_2"):
                name = f"{side}.mid_block.resnets.{0 if parts[1] == 'block_1' else 1}." + ".".join(parts[2:])
            else:
                m = {"norm": "group_norm", "q": "to_q", "k": "to_k", "v": "to_v", "proj_out": "to_out.0"}[parts[2]]
                name = f"{side}.mid_block.attentions.0.{m}.{parts[3]}"
                if v.ndim == 4:
                    v = v.reshape(v.shape[0], v.shape[1])
        elif parts[0] == "norm_out":
            name = f"{side}.conv_norm_out.{parts[1]}"
        else:
            name = f"{side}." + rest
        sd[name.replace("nin_shortcut", "conv_shortcut")] = v
    return sd


@torch.no_grad()
def test_g9_vae_against_in_tree_ldm_encoder_decoder(golden):
    from instructany2pix_amd.config import VAEConfig
    from instructany2pix_amd.weights import vae_param_specs
    d = golden("vae_ldm.npz")
    cfg = VAEConfig(block_out_channels=(64, 128, 128), layers_per_block=1).validate()
    sd = _ldm_to_diffusers_vae(d, cfg)
    z = cfg.latent_channels
    sd["quant_conv.weight"] = torch.eye(2 * z).reshape(2 * z, 2 * z, 1, 1); sd["quant_conv.bias"] = torch.zeros(2 * z)
    sd["post_quant_conv.weight"] = torch.eye(z).reshape(z, z, 1, 1); sd["post_quant_conv.bias"] = torch.zeros(z)
    assert set(sd) == {k for k, _, _ in vae_param_specs(cfg)}
    vae = oracle.build_vae(cfg, sd)
    e = vae.encode_moments(T(d["img"]))
    assert np.abs(e.numpy() - d["enc_out"]).max() < 2e-4 * max(1.0, np.abs(d["enc_out"]).max())       # ldm Encoder, blocks.py:435-460
    o = vae.decode(T(d["z"]))
    assert np.abs(o.numpy() - d["dec_out"]).max() < 2e-4 * max(1.0, np.abs(d["dec_out"]).max())       # ldm Decoder, blocks.py:536-569
    mom = torch.cat([torch.ones(1, 4, 2, 2), torch.full((1, 4, 2, 2), 40.0)], dim=1)
    s = oracle.sample_latents(mom, torch.ones(1, 4, 2, 2), 0.13025)
    assert torch.allclose(s, (1 + torch.exp(torch.tensor(10.0))) * 0.13025 * torch.ones(1, 4, 2, 2))    # logvar clamp at 20


def test_g10_refiner_time_ids(golden):
    """the oracle's aesthetics-score ids == the reference's `_get_add_time_ids` (pnp_pipeline.py:23-71, requires_aesthetics_score)"""
    d = golden("misc_refiner.npz")
    ids, neg = oracle.get_add_time_ids_aesthetic((1024, 1024), (0, 0), (1024, 1024), 6.0, 2.5, (1024, 1024), (0, 0), (1024, 1024), 256, 1280, 2560)
    assert np.array_equal(ids.numpy(), d["time_ids"]) and np.array_equal(neg.numpy(), d["neg_time_ids"])
    ids, neg = oracle.get_add_time_ids_aesthetic((768, 512), (8, 16), (768, 512), 7.5, 1.0, (640, 384), (4, 2), (768, 512), 256, 1280, 2560)
    assert np.array_equal(ids.numpy(), d["time_ids_b"]) and np.array_equal(neg.numpy(), d["neg_time_ids_b"])
    for name, requires, feats in (("err_enable", False, 3072), ("err_enable2", True, 2816), ("err_disable", False, 2560), ("err_config", True, 2000)):
        assert int(d[name]) == 1
        with pytest.raises(ValueError):
            oracle.get_add_time_ids_aesthetic((1024, 1024), (0, 0), (1024, 1024), 6.0, 2.5, (1024, 1024), (0, 0), (1024, 1024), 256, 1280, feats, requires)


@pytest.mark.parametrize("proj,act,eos", [(0, "quick_gelu", 2), (64, "gelu", 2), (64, "gelu", 999)])
def test_clip_restatement_matches_transformers(proj, act, eos):
    """The oracle's CLIP text model against the real transformers classes the reference's pipelines hold (`CLIPTextModel`,
    `CLIPTextModelWithProjection`), same random weights: pooled output, last_hidden_state and every hidden state."""
    tf = pytest.importorskip("transformers")
    from instructany2pix_amd.config import tiny_clip
    cfg = tiny_clip(proj, act)
    cfg.eos_token_id = eos
    hf_cfg = tf.CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                               num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                               max_position_embeddings=cfg.max_position_embeddings, hidden_act=act, projection_dim=proj or 512,
                               layer_norm_eps=cfg.layer_norm_eps, eos_token_id=eos, bos_token_id=0, pad_token_id=1, attn_implementation="eager")
    torch.manual_seed(3)
    hf = (tf.CLIPTextModelWithProjection if proj else tf.CLIPTextModel)(hf_cfg).eval()
    sd = {k: v for k, v in hf.state_dict().items() if not k.endswith("position_ids")}
    # checkpoints (and transformers 4.x, the reference's era) name the tower `text_model.*`; transformers 5 flattened CLIPTextModel
    sd = {(k if k.startswith(("text_model.", "text_projection.")) else "text_model." + k): v for k, v in sd.items()}
    ref = oracle.build_clip(cfg, sd)
    ids = torch.randint(3, cfg.vocab_size - 1, (3, 77), generator=torch.Generator().manual_seed(4))
    ids[:, 0] = 0
    for b, n in enumerate((10, 40, 76)):
        ids[b, n] = eos if eos != 2 else cfg.vocab_size - 1          # EOS: the largest id (legacy rule) or the configured id
        ids[b, n + 1:] = 1 if eos != 2 else cfg.vocab_size - 1       # padding (SDXL pads with EOS for encoder 1)
    with torch.no_grad():
        out = hf(input_ids=ids, output_hidden_states=True)
    pooled, last, hidden = ref(ids)
    want_pooled = out.text_embeds if proj else out.pooler_output
    assert (pooled - want_pooled).abs().max() < 2e-5
    assert (last - out.last_hidden_state).abs().max() < 2e-5
    assert len(hidden) == len(out.hidden_states) == cfg.num_hidden_layers + 1
    for a, b in zip(hidden, out.hidden_states):
        assert (a - b).abs().max() < 2e-5
    from instructany2pix_amd.weights import clip_param_specs
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: s for k, s, _ in clip_param_specs(cfg)}


def _clip_pair():
    from instructany2pix_amd.config import tiny_clip
    from instructany2pix_amd.weights import clip_param_specs, synthetic_state_dict
    c1, c2 = tiny_clip(0, "quick_gelu"), tiny_clip(64, "gelu")
    return (c1, oracle.build_clip(c1, synthetic_state_dict(clip_param_specs(c1), seed=31)),
            c2, oracle.build_clip(c2, synthetic_state_dict(clip_param_specs(c2), seed=32)))


def test_g11_encode_prompt_matches_reference_function(golden):
    """oracle.encode_prompt_ref == the reference's vendored `encode_prompt` (ddim/sdxl_pipeline.py:202-395) run on the same stand-in
    tokenizers and CLIP towers: concat of both penultimate states, pooled of tower 2, zero / empty negatives, per-image repeat"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from stub_tokenizer import StubTokenizer
    d = golden("encode_prompt.npz")
    c1, r1, c2, r2 = _clip_pair()
    t1, t2 = StubTokenizer(1, c1.vocab_size), StubTokenizer(2, c1.vocab_size)
    ids = lambda t, x: t(x, padding="max_length", max_length=77, truncation=True).input_ids
    prompts, negs = ["a photo of a cat", "two dogs on the beach at sunset"], ["blurry", "low quality, bad anatomy"]
    pe, ne, pp, npl = oracle.encode_prompt_ref(r1, r2, ids(t1, prompts), ids(t2, prompts), ids(t1, negs), ids(t2, negs), 2)
    for a, k in ((pe, "pair_pe"), (ne, "pair_ne"), (pp, "pair_pp"), (npl, "pair_np")):
        assert np.abs(a.numpy() - d[k]).max() < 1e-5, k
    one = ["a photo of a cat"]
    pe, ne, pp, npl = oracle.encode_prompt_ref(r1, r2, ids(t1, one), ids(t2, one), zero_negative=True)
    assert np.abs(pe.numpy() - d["zeros_pe"]).max() < 1e-5 and not d["zeros_ne"].any() and not d["zeros_np"].any()
    pe, ne, pp, npl = oracle.encode_prompt_ref(r1, r2, ids(t1, one), ids(t2, one), ids(t1, [""]), ids(t2, [""]))
    assert np.abs(ne.numpy() - d["empty_ne"]).max() < 1e-5 and np.abs(npl.numpy() - d["empty_np"]).max() < 1e-5
    assert int(d["err_type"]) == 1 and int(d["err_batch"]) == 1


@torch.no_grad()
def test_g12_inversion_loop_matches_reference_method(golden):
    """oracle.invert_loop == the reference's `SDXLDDIMPipeline.inverse` text (ddim/pnp_pipeline.py:92-278) run over the same oracle
    UNet: ascending timesteps, final_alpha_cumprod on the first move, no guidance, `[H, W, 0, 0, H, W]` ids, 77-token context"""
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    d = golden("inverse_loop.npz")
    cfg = tiny()
    unet = oracle.build_unet(cfg, synthetic_state_dict(unet_param_specs(cfg), seed=7), synthetic_state_dict(ip_adapter_specs(cfg, 64)["ip_adapter"], seed=7), ip_scale=1.0)
    x0, ctx, pooled = T(d["x0"]), T(d["ctx"]), T(d["pooled"])
    tid = torch.tensor([[128.0, 128.0, 0, 0, 128.0, 128.0]] * 2)
    for n in (5, 12):
        out = oracle.invert_loop(unet, oracle.DDIMSchedulerRef(), x0.clone(), ctx, dict(text_embeds=pooled, time_ids=tid), n)
        assert np.abs(out.numpy() - d[f"inv{n}"]).max() < 2e-5 * np.abs(d[f"inv{n}"]).max()


@torch.no_grad()
def test_g13_sampling_loop_matches_reference_methods(golden):
    """oracle.sample_loop (+ ImageProjModelRef, set_scale) == the reference's `IPAdapterXL.generate` (ip_adapter.py:289-356) calling the
    vendored SDXL `__call__` (ddim/sdxl_pipeline.py:544-886) over the same oracle UNet: image tokens appended to both contexts, zero
    embedding for the unconditional branch, cat([latents]*2), CFG combine, DDIM step, `[H,W,0,0,H,W]` ids on both halves"""
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    d = golden("sample_loop.npz")
    cfg = tiny()
    specs = ip_adapter_specs(cfg, 64)
    sd, ipsd = synthetic_state_dict(unet_param_specs(cfg), seed=7), synthetic_state_dict(specs["ip_adapter"], seed=7)
    m = oracle.ImageProjModelRef(cfg.cross_attention_dim, 64, 4)
    m.load_state_dict({k: v.float() for k, v in synthetic_state_dict(specs["image_proj"], seed=7).items()})
    emb, ctx, nctx, pooled, npooled, xT = (T(d[k]) for k in ("emb", "ctx", "nctx", "pooled", "npooled", "xT"))
    e = torch.stack([emb[None], torch.zeros(1, 64)], dim=1)
    p, n = m(e, "global"), m(torch.zeros_like(e), "global")
    tid = torch.tensor([[128.0, 128.0, 0, 0, 128.0, 128.0]])
    for tag, scale, g in (("g4_s07", 0.7, 4.0), ("g10_s10", 1.0, 10.0)):
        unet = oracle.build_unet(cfg, sd, ipsd, ip_scale=scale)
        out = oracle.sample_loop(unet, oracle.DDIMSchedulerRef(), xT.clone(), torch.cat([ctx, p], 1), dict(text_embeds=pooled, time_ids=tid), 6, g,
                                 torch.cat([nctx, n], 1), dict(text_embeds=npooled, time_ids=tid))
        assert np.abs(out.numpy() - d[tag]).max() < 2e-5 * np.abs(d[tag]).max(), tag


# ---- embedding prior (second half) ------------------------------------------------------------------------
@pytest.mark.parametrize("act", ["gelu_new", "gelu"])
def test_gpt2_restatement_matches_transformers(act):
    """The oracle's GPT-2 against the real transformers `GPT2Model` the reference's prior holds (prior/model.py:185), driven the way the
    prior drives it (`inputs_embeds` + all-ones mask), same random weights; and the key inventory of the product's weight specs."""
    tf = pytest.importorskip("transformers")
    from instructany2pix_amd.config import tiny_gpt2
    from instructany2pix_amd.weights import gpt2_param_specs
    cfg = tiny_gpt2()
    cfg.activation_function = act
    torch.manual_seed(8)
    hf = tf.GPT2Model(tf.GPT2Config(vocab_size=cfg.vocab_size, n_positions=cfg.n_positions, n_embd=cfg.n_embd, n_layer=cfg.n_layer, n_head=cfg.n_head,
                                    activation_function=act, bos_token_id=0, eos_token_id=0, attn_implementation="eager")).eval()
    sd = {k: v for k, v in hf.state_dict().items() if not k.endswith((".attn.bias", ".attn.masked_bias"))}
    for k in sd:                                   # transformers initialises biases / LayerNorms trivially: make them count
        if k.endswith(".bias") or ".ln_" in k or k.startswith("ln_f"):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape)
    hf.load_state_dict(sd, strict=False)
    ref = oracle.build_gpt2(cfg, sd)
    x = torch.randn(2, 11, cfg.n_embd, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        want = hf(inputs_embeds=x, attention_mask=torch.ones(2, 11))["last_hidden_state"]
    got = ref(x, torch.ones(2, 11))["last_hidden_state"]
    assert (got - want).abs().max() < 2e-5
    assert {k: tuple(v.shape) for k, v in sd.items()} == {k: s for k, s, _ in gpt2_param_specs(cfg)}


def test_prior_restatement_matches_reference_method_text():
    """G14: `InstructAny2PixPrior.generate_diffusion` and the methods under it, compiled from the reference's own file and run on
    transformers' GPT2Model, vs oracle.PriorRef on the same seeded weights: the live call of pipeline.py:313-317, a 3-step guided run
    (sample in the sequence, DDPM posterior with k1 != 0, per-step noise) and an unguided run."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from stub_tokenizer import StubTokenizer
    from instructany2pix_amd.config import tiny_clip, tiny_gpt2
    from instructany2pix_amd.weights import prior_param_specs, synthetic_state_dict
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prior.npz"))
    gcfg, ccfg = tiny_gpt2(), tiny_clip(0, "gelu")
    sd = synthetic_state_dict(prior_param_specs(gcfg, ccfg, (0, 1024, ccfg.hidden_size, 512, 0, 0, 0)), seed=41, dtype=torch.float32)
    clip = oracle.build_clip(ccfg, {k[len("cond_stage_models.0.model."):]: v for k, v in sd.items() if k.startswith("cond_stage_models.0.model.")})
    tok = StubTokenizer(5, ccfg.vocab_size)

    def text_hidden(prompts):
        b = tok(prompts, max_length=77, padding=True, truncation=True)
        return [clip(b.input_ids)[1], b.attention_mask.float()]
    p = oracle.PriorRef(gcfg, sd, text_hidden)
    assert list(G["sequence_input_key"]) == p.sequence_input_key and len(p.sequence_input_key) == 6      # the missing-comma key list
    src = torch.from_numpy(G["src"])
    for tag, kw, T in (("live", dict(no_diffusion=True, num_inference_steps=25, guidance_scale=10, force_guidence_t0=True, do_classifier_free_guidance=True, score=6.5), 11),
                       ("steps3", dict(no_diffusion=False, num_inference_steps=3, guidance_scale=4, do_classifier_free_guidance=True, score=6.5), 14),
                       ("nocfg", dict(no_diffusion=True, num_inference_steps=25, do_classifier_free_guidance=False, score=6.8), 11)):
        torch.manual_seed(1234)
        y, cond = p.generate_diffusion(3, 0, src, **kw)
        want = torch.from_numpy(G[tag + "_y"])
        assert float((y - want).abs().max()) <= 5e-5 * float(want.abs().max()), tag
        assert G[tag + "_seq0"].shape[1] == T and int(G[tag + "_ncalls"]) == (1 if kw["no_diffusion"] else 3)
