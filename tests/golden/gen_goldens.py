#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE'S OWN FILES.

Run only in the build container (needs /root/reference): `python tests/golden/gen_goldens.py`.
The fixtures are data (inputs, weights, expected outputs as float32 arrays); no reference source travels.

How reference code is reached without its un-installable imports (diffusers, imagebind, ...):
  * torch-only files are imported after registering bare parent packages in sys.modules (so no
    reference `__init__.py` runs): attention_processor.py, llm/model/vae/modules/{blocks,attention,util}.py;
  * pure functions/classes inside files that import diffusers are compiled from their own AST node
    (read from /root/reference at run time, never copied): `_backward_ddim`, `_get_add_time_ids`
    (pnp_pipeline.py), `ImageProjModel` (ip_adapter.py), `polar_intrtpolate` (pipeline.py).

Fixtures:
  G1 attn_self.npz     AttnProcessor2_0 + AttnProcessor on a stand-in `attn`
  G2 attn_ip.npz       IPAttnProcessor2_0 + IPAttnProcessor: 81-token ctx and the 77-token inversion quirk, 3 scales, attn_map
  G3 image_proj.npz    ImageProjModel, modes global/local/both
  G4 backward_ddim.npz _backward_ddim over the 20/25/50-step schedules
  G5 schedule.npz      ldm make_beta_schedule / make_ddim_timesteps / make_ddim_sampling_parameters
  G6 ldm_blocks.npz    ldm ResnetBlock, SpatialTransformer, timestep_embedding
  G7 misc.npz          polar_intrtpolate, _get_add_time_ids
  G8 unet_refprocs.npz tiny UNet (oracle module tree) with the REFERENCE processor classes installed
  G9 vae_ldm.npz       in-tree ldm Encoder / Decoder (the architecture the SDXL VAE descends from), small config
  G10 misc_refiner.npz _get_add_time_ids, requires_aesthetics_score branch (the refiner's 5 micro-conditioning ids) + its error cases
                       (`python gen_goldens.py refiner` writes only this one)
  G12 inverse_loop.npz  the reference's own `SDXLDDIMPipeline.inverse` text (ddim/pnp_pipeline.py:92-278) driving the oracle UNet (tiny config, seeded
                       weights, IP processors installed) with the oracle DDIM tables as the scheduler object (`python gen_goldens.py inverse`)
  G13 sample_loop.npz   the vendored SDXL `__call__` text (ddim/sdxl_pipeline.py:544-886, with its own prepare_latents / _get_add_time_ids /
                       check_inputs / prepare_extra_step_kwargs) driven by the reference's `IPAdapterXL.generate` (ip_adapter.py:289-356) and its
                       AST-extracted `ImageProjModel`, over the oracle UNet + oracle DDIM tables (`python gen_goldens.py sample`)
  G14 prior.npz         the reference's own embedding-prior inference text (`InstructAny2PixPrior.generate_diffusion`, `get_input_sequence_and_mask`,
                       `add_sos_eos_tokens`, `get_eps`, `get_input`, ...; `CLIPTextModelHiddenState.forward / encode_text`; prior/model.py) and its
                       `prior_config` literal (prior/__init__.py), on transformers' real GPT2Model (tiny config, seeded weights), the oracle CLIP tower,
                       the reference's ldm `timestep_embedding` for diffusers' `get_timestep_embedding`, and the oracle's DDPM restatement as the
                       scheduler object (diffusers is not installed) (`python gen_goldens.py prior`)
  G11 encode_prompt.npz the reference's vendored `encode_prompt` (ddim/sdxl_pipeline.py:202-395) driven with stand-in tokenizers and the
                       oracle's CLIP towers on seeded weights (`python gen_goldens.py encode_prompt`)
"""
import ast
import importlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def _stub_pkg(name, path):
    m = types.ModuleType(name)
    m.__path__ = [path]
    sys.modules[name] = m


def import_reference_modules():
    base = os.path.join(REF, "instructany2pix")
    _stub_pkg("instructany2pix", base)
    for sub in ("diffusion", "diffusion/ip_adapter", "llm", "llm/model", "llm/model/vae", "llm/model/vae/modules", "ddim"):
        _stub_pkg("instructany2pix." + sub.replace("/", "."), os.path.join(base, sub))
    ap = importlib.import_module("instructany2pix.diffusion.ip_adapter.attention_processor")
    blocks = importlib.import_module("instructany2pix.llm.model.vae.modules.blocks")
    attention = importlib.import_module("instructany2pix.llm.model.vae.modules.attention")
    util = importlib.import_module("instructany2pix.llm.model.vae.modules.util")
    return ap, blocks, attention, util


def ast_extract(relpath, names, glb):
    """Compile selected top-level defs / class methods of a reference file from its AST."""
    src = open(os.path.join(REF, relpath)).read()
    tree = ast.parse(src)
    out = {}
    for node in ast.walk(tree):
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names and node.name not in out:
            mod = ast.Module(body=[node], type_ignores=[])
            ns = dict(glb)
            exec(compile(mod, relpath, "exec"), ns)
            out[node.name] = ns[node.name]
    missing = set(names) - set(out)
    assert not missing, missing
    return out


def ast_extract_method(relpath, cls, names, glb):
    """Compile selected methods of ONE class of a reference file from its AST (several classes may define the same name)."""
    tree = ast.parse(open(os.path.join(REF, relpath)).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name in names:
                    ns = dict(glb)
                    exec(compile(ast.Module(body=[sub], type_ignores=[]), relpath, "exec"), ns)
                    out[sub.name] = ns[sub.name]
    missing = set(names) - set(out)
    assert not missing, missing
    return out


class StandInAttn(torch.nn.Module):
    """Minimal `attn` object for the processor protocol."""

    def __init__(self, dim, ctx_dim, heads):
        super().__init__()
        self.heads = heads
        self.scale = (dim // heads) ** -0.5
        self.to_q = torch.nn.Linear(dim, dim, bias=False)
        self.to_k = torch.nn.Linear(ctx_dim or dim, dim, bias=False)
        self.to_v = torch.nn.Linear(ctx_dim or dim, dim, bias=False)
        self.to_out = torch.nn.ModuleList([torch.nn.Linear(dim, dim), torch.nn.Dropout(0.0)])
        self.spatial_norm = self.group_norm = self.norm_cross = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0

    def prepare_attention_mask(self, m, *a):
        return m

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        return t.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, n, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        return t.reshape(bh // self.heads, self.heads, n, d).permute(0, 2, 1, 3).reshape(bh // self.heads, n, d * self.heads)

    def get_attention_scores(self, q, k, mask=None):
        return (torch.bmm(q, k.transpose(-1, -2)) * self.scale).softmax(dim=-1)


def npf(t):
    return t.detach().float().numpy()


def main():
    torch.manual_seed(1234)
    torch.set_grad_enabled(False)
    ap, blocks, attention, util = import_reference_modules()
    g = torch.Generator().manual_seed(11)
    rn = lambda *s: torch.randn(*s, generator=g)

    # ---- G1 ------------------------------------------------------------------------------------
    dim, heads, ctxd = 128, 2, 96
    attn = StandInAttn(dim, None, heads)
    for p in attn.parameters():
        p.copy_(rn(*p.shape) * (0.09 if p.ndim == 2 else 0.02))
    x = rn(2, 48, dim)
    o2 = ap.AttnProcessor2_0()(attn, x)
    o1 = ap.AttnProcessor()(attn, x)
    np.savez(os.path.join(HERE, "attn_self.npz"), x=npf(x), heads=heads,
             to_q=npf(attn.to_q.weight), to_k=npf(attn.to_k.weight), to_v=npf(attn.to_v.weight),
             to_out_w=npf(attn.to_out[0].weight), to_out_b=npf(attn.to_out[0].bias),
             out_2_0=npf(o2), out_bmm=npf(o1))
    print("G1 max|2_0 - bmm| =", float((o1 - o2).abs().max()))

    # ---- G2 ------------------------------------------------------------------------------------
    attn = StandInAttn(dim, ctxd, heads)
    for p in attn.parameters():
        p.copy_(rn(*p.shape) * (0.09 if p.ndim == 2 else 0.02))
    proc = ap.IPAttnProcessor2_0(dim, ctxd, scale=1.0, num_tokens=4)
    proc_b = ap.IPAttnProcessor(dim, ctxd, scale=1.0, num_tokens=4)
    for p in proc.parameters():
        p.copy_(rn(*p.shape) * 0.1)
    proc_b.load_state_dict(proc.state_dict())
    x = rn(2, 48, dim)
    d = dict(x=npf(x), heads=heads, to_q=npf(attn.to_q.weight), to_k=npf(attn.to_k.weight), to_v=npf(attn.to_v.weight),
             to_out_w=npf(attn.to_out[0].weight), to_out_b=npf(attn.to_out[0].bias),
             to_k_ip=npf(proc.to_k_ip.weight), to_v_ip=npf(proc.to_v_ip.weight))
    for L in (81, 77):
        ctx = rn(2, L, ctxd)
        d[f"ctx{L}"] = npf(ctx)
        for s in (0.0, 0.5, 1.0):
            proc.scale = proc_b.scale = s
            o = proc(attn, x, encoder_hidden_states=ctx)
            ob = proc_b(attn, x, encoder_hidden_states=ctx)
            d[f"out{L}_s{s}"] = npf(o)
            d[f"outbmm{L}_s{s}"] = npf(ob)
        d[f"attn_map{L}"] = npf(proc.attn_map)
    np.savez(os.path.join(HERE, "attn_ip.npz"), **d)

    # ---- G3 ------------------------------------------------------------------------------------
    IPM = ast_extract("instructany2pix/diffusion/ip_adapter/ip_adapter.py", ["ImageProjModel"], {"torch": torch})["ImageProjModel"]
    m = IPM(cross_attention_dim=64, clip_embeddings_dim=48, clip_extra_context_tokens=4)
    for p in m.parameters():
        p.copy_(rn(*p.shape) * 0.2)
    emb = rn(2, 2, 48)
    d = dict(emb=npf(emb), **{k.replace(".", "_"): npf(v) for k, v in m.state_dict().items()})
    for mode in ("global", "local", "both"):
        for sc in ((1.0, 1.0), (1.0, 0.5)):
            d[f"out_{mode}_{sc[1]}"] = npf(m(emb.clone(), mode=mode, scales=list(sc)))
    d["out_zero_global"] = npf(m(torch.zeros_like(emb), mode="global"))
    np.savez(os.path.join(HERE, "image_proj.npz"), **d)

    # ---- G4 / G7 -------------------------------------------------------------------------------
    fns = ast_extract("instructany2pix/ddim/pnp_pipeline.py", ["_backward_ddim", "_get_add_time_ids"], {"torch": torch})
    bd = fns["_backward_ddim"]
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    acp = torch.cumprod(1 - betas, 0)
    xs = rn(1, 4, 8, 8)
    es = rn(60, 1, 4, 8, 8)
    d = dict(x0=npf(xs), eps=npf(es))
    for n in (20, 25, 50):
        ts = (np.arange(0, n) * (1000 // n)).round()[::-1].copy().astype(np.int64) + 1
        lat = xs.clone()
        prev = None
        traj = []
        for i, t in enumerate(reversed(ts.tolist())):
            a_t = acp[t]
            a_p = acp[prev] if prev is not None else acp[0]
            prev = t
            lat = bd(lat, a_t, a_p, es[i])
            traj.append(npf(lat))
        d[f"traj{n}"] = np.stack(traj)
        d[f"timesteps{n}"] = ts
    # fp16 tensor x fp32 0-dim coefficient, as in the live pipeline
    d["half_step"] = npf(bd(xs.half(), acp[501], acp[481], es[0].half()))
    np.savez(os.path.join(HERE, "backward_ddim.npz"), **d)

    polar = ast_extract("instructany2pix/pipeline.py", ["polar_intrtpolate"], {"torch": torch})["polar_intrtpolate"]
    a, b = rn(1, 4, 16, 16), rn(1, 4, 16, 16)
    dm = dict(pa=npf(a), pb=npf(b), polar_07=npf(polar(None, a, b, 0.7)), polar_03=npf(polar(None, a, b, 0.3)),
              polar_half=npf(polar(None, a.half(), b.half(), 0.7)))
    fake = types.SimpleNamespace(
        config=types.SimpleNamespace(requires_aesthetics_score=False),
        unet=types.SimpleNamespace(config=types.SimpleNamespace(addition_time_embed_dim=256),
                                   add_embedding=types.SimpleNamespace(linear_1=types.SimpleNamespace(in_features=2816))),
        text_encoder_2=types.SimpleNamespace(config=types.SimpleNamespace(projection_dim=1280)))
    ids, nids = fns["_get_add_time_ids"](fake, (1024, 1024), (0, 0), (1024, 1024), 6.0, 2.5, (1024, 1024), (0, 0), (1024, 1024), torch.float32)
    dm["time_ids"] = npf(ids)
    dm["neg_time_ids"] = npf(nids)
    fake.unet.add_embedding.linear_1.in_features = 2560
    try:
        fns["_get_add_time_ids"](fake, (1024, 1024), (0, 0), (1024, 1024), 6.0, 2.5, (1024, 1024), (0, 0), (1024, 1024), torch.float32)
        dm["bad_dim_raises"] = 0
    except ValueError:
        dm["bad_dim_raises"] = 1
    np.savez(os.path.join(HERE, "misc.npz"), **dm)

    # ---- G5 ------------------------------------------------------------------------------------
    b64 = util.make_beta_schedule("linear", 1000, 0.00085, 0.012)
    acp64 = np.cumprod(1 - b64)
    d = dict(betas=b64, alphas_cumprod=acp64)
    for n in (20, 25, 50):
        ts = util.make_ddim_timesteps("uniform", n, 1000, verbose=False)
        _, al, alp = util.make_ddim_sampling_parameters(acp64, ts, 0.0, verbose=False)
        d[f"ts{n}"], d[f"alphas{n}"], d[f"alphas_prev{n}"] = ts, al, alp
    np.savez(os.path.join(HERE, "schedule.npz"), **d)

    # ---- G6 ------------------------------------------------------------------------------------
    d = {}
    rb = blocks.ResnetBlock(in_channels=64, out_channels=96, dropout=0.0, temb_channels=48)
    for p in rb.parameters():
        p.copy_(rn(*p.shape) * (0.06 if p.ndim > 1 else 0.3))
    x, temb = rn(2, 64, 8, 8), rn(2, 48)
    d.update({"rb_" + k.replace(".", "_"): npf(v) for k, v in rb.state_dict().items()})
    d["rb_x"], d["rb_temb"], d["rb_out"] = npf(x), npf(temb), npf(rb(x, temb))
    st = attention.SpatialTransformer(64, 1, 64, depth=2, context_dim=40)
    for p in st.parameters():
        p.copy_(rn(*p.shape) * (0.1 if p.ndim > 1 else 0.3))
    x, ctx = rn(2, 64, 6, 6), rn(2, 9, 40)
    d.update({"st_" + k.replace(".", "_"): npf(v) for k, v in st.state_dict().items()})
    d["st_keys"] = np.array(list(st.state_dict().keys()))
    d["st_x"], d["st_ctx"], d["st_out"] = npf(x), npf(ctx), npf(st(x, ctx))
    tt = torch.tensor([1.0, 481.0, 981.0, 1024.0])
    d["temb_t"], d["temb_320"], d["temb_256"] = npf(tt), npf(util.timestep_embedding(tt, 320)), npf(util.timestep_embedding(tt, 256))
    np.savez(os.path.join(HERE, "ldm_blocks.npz"), **d)

    # ---- G8: oracle module tree driven by the REFERENCE processor classes -----------------------------
    import oracle
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    cfg = tiny()
    sd = synthetic_state_dict(unet_param_specs(cfg), seed=7)
    ipsd = synthetic_state_dict(ip_adapter_specs(cfg, 64)["ip_adapter"], seed=7)
    net = oracle.build_unet(cfg, sd)
    procs = {}
    for name in net.attn_processors.keys():            # reference ip_adapter.py:120-142, verbatim semantics
        if name.endswith("attn1.processor"):
            procs[name] = ap.AttnProcessor2_0()
        else:
            if name.startswith("mid_block"):
                hs = cfg.block_out_channels[-1]
            elif name.startswith("up_blocks"):
                hs = list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])]
            else:
                hs = cfg.block_out_channels[int(name[len("down_blocks.")])]
            procs[name] = ap.IPAttnProcessor2_0(hidden_size=hs, cross_attention_dim=cfg.cross_attention_dim, scale=1.0, num_tokens=4)
    net.set_attn_processor(procs)
    torch.nn.ModuleList(net.attn_processors.values()).load_state_dict({k: v.float() for k, v in ipsd.items()})
    gg = torch.Generator().manual_seed(5)
    B = 2
    x = torch.randn(B, 4, 16, 16, generator=gg)
    te = torch.randn(B, cfg.pooled_dim, generator=gg)
    tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]] * B)
    d = dict(x=npf(x), text_embeds=npf(te), time_ids=npf(tid))
    for L in (81, 77):
        ctx = torch.randn(B, L, cfg.cross_attention_dim, generator=gg)
        d[f"ctx{L}"] = npf(ctx)
        for t in (981, 1):
            for s in (1.0, 0.5):
                for p_ in procs.values():
                    if hasattr(p_, "scale"):
                        p_.scale = s
                d[f"out_L{L}_t{t}_s{s}"] = npf(net(x, t, ctx, added_cond_kwargs=dict(text_embeds=te, time_ids=tid))[0])
    np.savez(os.path.join(HERE, "unet_refprocs.npz"), **d)
    # ---- G9: ldm Encoder / Decoder (blocks.py:369-569) -------------------------------------------------------------------
    import contextlib
    import io
    kw = dict(ch=64, out_ch=3, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=[], dropout=0.0, in_channels=3,
              resolution=32, z_channels=4)
    with contextlib.redirect_stdout(io.StringIO()):
        enc, dec = blocks.Encoder(double_z=True, **kw), blocks.Decoder(**kw)
    # weights are NOT stored (3.7 M values): they are drawn from a seeded CPU generator in parameter order, and the test
    # re-draws them in the recorded order (names/shapes below) -- only inputs and expected outputs travel
    gg = torch.Generator().manual_seed(19)
    names, shapes = [], []
    for side, m_ in (("enc", enc), ("dec", dec)):
        for k_, p_ in m_.named_parameters():
            p_.copy_(torch.randn(p_.shape, generator=gg) * (0.5 / max(1, p_[0].numel()) ** 0.5 if p_.ndim > 1 else 0.2))
            names.append(f"{side}.{k_}")
            shapes.append(",".join(str(x) for x in p_.shape))
    img = torch.randn(2, 3, 32, 32, generator=gg)
    zz = torch.randn(2, 4, 8, 8, generator=gg)
    np.savez(os.path.join(HERE, "vae_ldm.npz"), img=npf(img), z=npf(zz), enc_out=npf(enc(img)), dec_out=npf(dec(zz)),
             names=np.array(names), shapes=np.array(shapes), seed=19)

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


def gen_refiner():
    """G10: the aesthetics-score branch of the reference's own `_get_add_time_ids` (pnp_pipeline.py:23-71)"""
    fn = ast_extract("instructany2pix/ddim/pnp_pipeline.py", ["_get_add_time_ids"], {"torch": torch})["_get_add_time_ids"]

    def fake(requires, in_features):
        return types.SimpleNamespace(
            config=types.SimpleNamespace(requires_aesthetics_score=requires),
            unet=types.SimpleNamespace(config=types.SimpleNamespace(addition_time_embed_dim=256),
                                       add_embedding=types.SimpleNamespace(linear_1=types.SimpleNamespace(in_features=in_features))),
            text_encoder_2=types.SimpleNamespace(config=types.SimpleNamespace(projection_dim=1280)))

    d = {}
    ids, neg = fn(fake(True, 2560), (1024, 1024), (0, 0), (1024, 1024), 6.0, 2.5, (1024, 1024), (0, 0), (1024, 1024), torch.float32)
    d["time_ids"], d["neg_time_ids"] = npf(ids), npf(neg)
    ids, neg = fn(fake(True, 2560), (768, 512), (8, 16), (768, 512), 7.5, 1.0, (640, 384), (4, 2), (768, 512), torch.float32)
    d["time_ids_b"], d["neg_time_ids_b"] = npf(ids), npf(neg)
    msgs = []
    for name, requires, feats in (("err_enable", False, 2560 + 512), ("err_enable2", True, 2816), ("err_disable", False, 2560), ("err_config", True, 2000)):
        try:
            fn(fake(requires, feats), (1024, 1024), (0, 0), (1024, 1024), 6.0, 2.5, (1024, 1024), (0, 0), (1024, 1024), torch.float32)
            d[name] = 0
        except ValueError as e:
            d[name] = 1
            msgs.append(f"{name}: {'enable' if 'to enable' in str(e) else 'disable' if 'to disable' in str(e) else 'config'}")
    d["err_kinds"] = np.array(msgs)
    np.savez(os.path.join(HERE, "misc_refiner.npz"), **d)
    print(d["time_ids"], d["neg_time_ids"], msgs)


def gen_encode_prompt():
    """G11: run the reference's own encode_prompt text (AST-extracted method) on stand-in tokenizers and the oracle CLIP towers"""
    import logging
    from typing import List, Optional
    import oracle
    sys.path.insert(0, os.path.dirname(HERE))
    from stub_tokenizer import StubTokenizer
    from instructany2pix_amd.config import tiny_clip
    from instructany2pix_amd.weights import clip_param_specs, synthetic_state_dict
    fn = ast_extract("instructany2pix/ddim/sdxl_pipeline.py", ["encode_prompt"],
                     {"torch": torch, "List": List, "Optional": Optional, "LoraLoaderMixin": type("L", (), {}), "TextualInversionLoaderMixin": type("T", (), {}),
                      "adjust_lora_scale_text_encoder": lambda *a: None, "logger": logging.getLogger("golden")})["encode_prompt"]
    c1, c2 = tiny_clip(0, "quick_gelu"), tiny_clip(64, "gelu")
    r1 = oracle.build_clip(c1, synthetic_state_dict(clip_param_specs(c1), seed=31))
    r2 = oracle.build_clip(c2, synthetic_state_dict(clip_param_specs(c2), seed=32))

    class Enc:                                     # transformers-style output: [0] = pooled / text_embeds, .hidden_states
        dtype = torch.float32

        def __init__(self, m):
            self.m = m

        def __call__(self, ids, output_hidden_states=True):
            pooled, last, hidden = self.m(ids)
            out = types.SimpleNamespace(hidden_states=hidden)
            return type("O", (), {"__getitem__": lambda s, i: pooled, "hidden_states": hidden})()

    def pipe(force_zeros):
        return types.SimpleNamespace(tokenizer=StubTokenizer(1, c1.vocab_size), tokenizer_2=StubTokenizer(2, c1.vocab_size), text_encoder=Enc(r1), text_encoder_2=Enc(r2),
                                     config=types.SimpleNamespace(force_zeros_for_empty_prompt=force_zeros), _execution_device="cpu")
    prompts, negs = ["a photo of a cat", "two dogs on the beach at sunset"], ["blurry", "low quality, bad anatomy"]
    d = {}
    for tag, args in (("pair", dict(prompt=prompts, negative_prompt=negs, num_images_per_prompt=2)),
                      ("zeros", dict(prompt="a photo of a cat")),
                      ("empty", dict(prompt="a photo of a cat", _force=False)),
                      ("nocfg", dict(prompt=prompts, do_classifier_free_guidance=False))):
        force = args.pop("_force", True)
        pe, ne, pp, npl = fn(pipe(force), **args)
        d[tag + "_pe"], d[tag + "_pp"] = npf(pe), npf(pp)
        if ne is not None:
            d[tag + "_ne"], d[tag + "_np"] = npf(ne), npf(npl)
    for name, kw, exc in (("err_type", dict(prompt="a cat", negative_prompt=["x"]), TypeError), ("err_batch", dict(prompt=["a", "b"], negative_prompt=["x"]), ValueError)):
        try:
            fn(pipe(True), **kw)
            d[name] = 0
        except exc:
            d[name] = 1
    np.savez_compressed(os.path.join(HERE, "encode_prompt.npz"), **d)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items()})


def gen_prior():
    """G14: the reference prior's inference methods, compiled from their own AST, run on stand-ins"""
    import inspect
    import logging
    import transformers
    import oracle
    from oracle.prior_ref import DDPMSchedulerRef
    sys.path.insert(0, os.path.dirname(HERE))
    from stub_tokenizer import StubTokenizer
    from instructany2pix_amd.config import tiny_clip, tiny_gpt2
    from instructany2pix_amd.weights import prior_param_specs, synthetic_state_dict
    _, _, _, util = import_reference_modules()
    tree = ast.parse(open(os.path.join(REF, "instructany2pix/prior/__init__.py")).read())
    prior_config = next(ast.literal_eval(n.value) for n in tree.body if isinstance(n, ast.Assign) and n.targets[0].id == "prior_config")

    class Bar:                      # tqdm(total=) context manager
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def update(self, *a): pass
    glb = {"torch": torch, "nn": torch.nn, "inspect": inspect, "logging": logging, "tqdm": Bar,
           "MODALITY": types.SimpleNamespace(IMAGE=0, AUDIO=1, TEXT=2, VIDEO=3),
           "get_timestep_embedding": lambda timesteps, embedding_dim, flip_sin_to_cos, downscale_freq_shift: util.timestep_embedding(timesteps, embedding_dim)}
    rel = "instructany2pix/prior/model.py"
    P = type("P", (torch.nn.Module,), ast_extract_method(rel, "InstructAny2PixPrior", [
        "get_eps", "add_sos_eos_tokens", "truncate_sequence_and_mask", "get_input_sequence_and_mask", "prepare_extra_step_kwargs", "generate_diffusion",
        "get_input_item", "get_input", "get_learned_conditioning"], glb))
    Hs = type("Hs", (torch.nn.Module,), ast_extract_method(rel, "CLIPTextModelHiddenState", ["forward", "encode_text"], glb))

    gcfg, ccfg = tiny_gpt2(), tiny_clip(0, "gelu")
    E = gcfg.n_embd
    dims = (0, 1024, ccfg.hidden_size, 512, 0, 0, 0)
    sd = synthetic_state_dict(prior_param_specs(gcfg, ccfg, dims), seed=41, dtype=torch.float32)
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}

    clip = oracle.build_clip(ccfg, sub("cond_stage_models.0.model."))

    class ClipShim(torch.nn.Module):          # transformers call form on the (transformers-pinned) oracle tower
        def __init__(self):
            super().__init__()
            self.m = clip
            self.device = torch.device("cpu")

        def forward(self, input_ids=None, attention_mask=None):
            assert bool(attention_mask.all())
            return (self.m(input_ids)[1],)
    hs = Hs()
    hs.freeze_text_encoder, hs.tokenizer, hs.model, hs.device = True, StubTokenizer(5, ccfg.vocab_size), ClipShim(), None
    for p_ in hs.model.parameters():
        p_.requires_grad = False

    gpt = transformers.GPT2Model(transformers.GPT2Config(vocab_size=gcfg.vocab_size, n_positions=gcfg.n_positions, n_embd=E, n_layer=gcfg.n_layer,
                                                         n_head=gcfg.n_head, activation_function=gcfg.activation_function)).eval()
    missing, unexpected = gpt.load_state_dict(sub("model."), strict=False)
    assert not unexpected and all(k.endswith((".attn.bias", ".attn.masked_bias")) for k in missing), (missing, unexpected)

    m = P()
    m.noise_scheduler = DDPMSchedulerRef()
    m.mae_token_num = prior_config["sequence_gen_length"]
    m.sequence_input_key = prior_config["sequence_input_key"]
    m.embed_dim = E
    m.device = "cpu"
    m.start_of_sequence_tokens, m.end_of_sequence_tokens = torch.nn.Embedding(32, E), torch.nn.Embedding(32, E)
    m.modality_embedding = torch.nn.Embedding(10, E)
    m.input_sequence_embed_linear = torch.nn.ModuleList([torch.nn.Linear(d_, E) if d_ else torch.nn.Identity() for d_ in dims])
    m.cond_stage_models = torch.nn.ModuleList([hs])
    m.cond_stage_model_metadata = {"crossattn_clip": {"model_idx": 0, "cond_stage_key": "text", "conditioning_key": "crossattn"}}
    m.model = gpt
    missing, unexpected = m.load_state_dict({k: v for k, v in sd.items() if not k.startswith("cond_stage_models.")}, strict=False)
    assert not unexpected and all(k.startswith("cond_stage_models.") or k.endswith((".attn.bias", ".attn.masked_bias")) for k in missing), (missing, unexpected)
    m.eval()

    seen = []
    gpt.register_forward_pre_hook(lambda mod, a, kw: seen.append(kw["inputs_embeds"].clone()), with_kwargs=True)
    g = torch.Generator().manual_seed(77)
    emb = torch.randn(1, 1024, generator=g)
    src = emb / emb.norm() * 100
    d = dict(src=npf(src), sequence_input_key=np.array(prior_config["sequence_input_key"]))
    # the live call (reference pipeline.py:313-317), then the iterative form and the unguided form of the same method
    for tag, kw in (("live", dict(no_diffusion=True, num_inference_steps=25, guidance_scale=10, force_guidence_t0=True, do_classifier_free_guidance=True, score=6.5)),
                    ("steps3", dict(no_diffusion=False, num_inference_steps=3, guidance_scale=4, do_classifier_free_guidance=True, score=6.5)),
                    ("nocfg", dict(no_diffusion=True, num_inference_steps=25, do_classifier_free_guidance=False, score=6.8))):
        torch.manual_seed(1234)
        seen.clear()
        y, cond = m.generate_diffusion(3, 0, src, device="cpu", image_bind_overwrite=None, dtype=torch.float32, **kw)
        d[tag + "_y"], d[tag + "_seq0"], d[tag + "_ncalls"] = npf(y), npf(seen[0]), len(seen)
    np.savez_compressed(os.path.join(HERE, "prior.npz"), **d)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in d.items()})


def gen_inverse():
    """G12: the reference's inversion loop itself (method text compiled from its AST) over the oracle UNet"""
    from typing import Any, Callable, Dict, List, Optional, Tuple, Union
    import oracle
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    ns = {"torch": torch, "np": np, "List": List, "Optional": Optional, "Union": Union, "Callable": Callable, "Dict": Dict, "Any": Any, "Tuple": Tuple,
          "PIL": types.SimpleNamespace(Image=types.SimpleNamespace(Image=object)), "tqdm": lambda it: it,
          "StableDiffusionXLPipelineOutput": lambda images: types.SimpleNamespace(images=images)}
    fns = ast_extract("instructany2pix/ddim/pnp_pipeline.py", ["_backward_ddim", "_get_add_time_ids"], {"torch": torch})
    ns.update(fns)

    class SchedStandIn(oracle.DDIMSchedulerRef):          # diffusers' DDIMScheduler is absent: the oracle's tables (pinned by G5) stand in
        config = types.SimpleNamespace()

        @classmethod
        def from_config(cls, config):
            return cls()
    ns["DDIMScheduler"] = SchedStandIn
    inverse = ast_extract("instructany2pix/ddim/pnp_pipeline.py", ["inverse"], ns)["inverse"]
    cfg = tiny()
    sd = synthetic_state_dict(unet_param_specs(cfg), seed=7)
    ipsd = synthetic_state_dict(ip_adapter_specs(cfg, 64)["ip_adapter"], seed=7)
    unet = oracle.build_unet(cfg, sd, ipsd, ip_scale=1.0)
    g = torch.Generator().manual_seed(23)
    B, h, w = 2, 16, 16
    x0 = torch.randn(B, 4, h, w, generator=g)
    ctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g)          # the inversion runs with the 77-token context on the IP-enabled UNet
    pooled = torch.randn(B, cfg.pooled_dim, generator=g)
    pipe = types.SimpleNamespace(
        scheduler=SchedStandIn(), unet=unet, _execution_device="cpu", vae_scale_factor=8, check_inputs=lambda *a, **k: None,
        encode_prompt=lambda **k: (k["prompt_embeds"], None, k["pooled_prompt_embeds"], None),
        image_processor=types.SimpleNamespace(preprocess=lambda im: im), prepare_latents=lambda image, *a: image,
        text_encoder_2=types.SimpleNamespace(config=types.SimpleNamespace(projection_dim=cfg.pooled_dim)),
        config=types.SimpleNamespace(requires_aesthetics_score=False))
    d = dict(x0=npf(x0), ctx=npf(ctx), pooled=npf(pooled))
    for n in (5, 12):
        out = inverse(pipe, prompt_embeds=ctx, pooled_prompt_embeds=pooled, image=x0.clone(), num_inference_steps=n)
        d[f"inv{n}"] = npf(out.images)
    np.savez_compressed(os.path.join(HERE, "inverse_loop.npz"), **d)
    print({k: v.shape for k, v in d.items()})


def gen_sample():
    """G13: IPAdapterXL.generate -> StableDiffusionXLPipeline.__call__ (both the reference's own text) over the oracle UNet"""
    import inspect
    from typing import Any, Callable, Dict, List, Optional, Tuple, Union
    import oracle
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    ns = {"torch": torch, "np": np, "inspect": inspect, "List": List, "Optional": Optional, "Union": Union, "Callable": Callable, "Dict": Dict, "Any": Any,
          "Tuple": Tuple, "replace_example_docstring": lambda doc: (lambda f: f), "EXAMPLE_DOC_STRING": "",
          "StableDiffusionXLPipelineOutput": lambda images: types.SimpleNamespace(images=images)}
    ns.update(ast_extract("instructany2pix/ddim/sdxl_pipeline.py", ["rescale_noise_cfg"], {"torch": torch}))
    m = ast_extract("instructany2pix/ddim/sdxl_pipeline.py", ["__call__", "prepare_latents", "_get_add_time_ids", "prepare_extra_step_kwargs", "check_inputs"], ns)

    class Sched(oracle.DDIMSchedulerRef):               # stand-in for diffusers' DDIMScheduler object (absent): oracle tables + step
        order = 1
        config = types.SimpleNamespace(num_train_timesteps=1000)

        def set_timesteps(self, n, device=None):
            super().set_timesteps(n)

        def step(self, model_output, timestep, sample, eta=0.0, return_dict=False):
            return (super().step(model_output, timestep, sample),)

    cfg = tiny()
    sd = synthetic_state_dict(unet_param_specs(cfg), seed=7)
    specs = ip_adapter_specs(cfg, 64)
    ipsd = synthetic_state_dict(specs["ip_adapter"], seed=7)
    unet = oracle.build_unet(cfg, sd, ipsd, ip_scale=1.0)
    unet.add_embedding = types.SimpleNamespace(linear_1=types.SimpleNamespace(in_features=cfg.projection_class_embeddings_input_dim)) if not hasattr(unet.add_embedding, "linear_1") else unet.add_embedding

    class Bar:
        def __init__(self, total): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def update(self): pass

    class Pipe:
        pass
    pipe = Pipe()
    for k, f in m.items():
        setattr(Pipe, k, f)
    pipe.unet, pipe.scheduler, pipe._execution_device, pipe.vae_scale_factor, pipe.default_sample_size = unet, Sched(), "cpu", 8, 16
    pipe.text_encoder_2 = types.SimpleNamespace(config=types.SimpleNamespace(projection_dim=cfg.pooled_dim), dtype=torch.float32)
    Pipe.progress_bar = lambda self, total=None: Bar(total)
    Pipe.maybe_free_model_hooks = lambda self: None
    Pipe.to = lambda self, *a, **k: self

    # IPAdapterXL.generate + get_image_embeds + set_scale: the reference's own text over the reference's ImageProjModel
    ipn = {"torch": torch, "List": List, "Image": types.SimpleNamespace(Image=type("NoImage", (), {})), "get_generator": lambda seed, device: None,
           "IPAttnProcessor": oracle.IPAttnProcessor2_0Ref}
    gen = ast_extract_method("instructany2pix/diffusion/ip_adapter/ip_adapter.py", "IPAdapter", ["get_image_embeds", "set_scale"], ipn)
    gen.update(ast_extract_method("instructany2pix/diffusion/ip_adapter/ip_adapter.py", "IPAdapterXL", ["generate"], ipn))
    IPM = ast_extract("instructany2pix/diffusion/ip_adapter/ip_adapter.py", ["ImageProjModel"], {"torch": torch})["ImageProjModel"]
    proj = IPM(cross_attention_dim=cfg.cross_attention_dim, clip_embeddings_dim=64, clip_extra_context_tokens=4)
    proj.load_state_dict({k: v.float() for k, v in synthetic_state_dict(specs["image_proj"], seed=7).items()})

    class Adapter:
        pass
    for k, f in gen.items():
        setattr(Adapter, k, f)
    ad = Adapter()
    ad.pipe, ad.device, ad.image_proj_model = pipe, "cpu", proj
    g = torch.Generator().manual_seed(29)
    B, h, w, N = 1, 16, 16, 6
    emb = torch.randn(64, generator=g)
    ctx, nctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g), torch.randn(B, 77, cfg.cross_attention_dim, generator=g)
    pooled, npooled = torch.randn(B, cfg.pooled_dim, generator=g), torch.randn(B, cfg.pooled_dim, generator=g)
    xT = torch.randn(B, 4, h, w, generator=g)
    # text encoders are stand-ins: a prompt string maps to fixed embeddings; pre-computed embeddings pass through (as :281 does)
    Pipe.encode_prompt = lambda self, *a, **k: ((k["prompt_embeds"], k["negative_prompt_embeds"], k["pooled_prompt_embeds"], k["negative_pooled_prompt_embeds"])
                                                if k.get("prompt_embeds") is not None else (ctx, nctx, pooled, npooled))
    # generate() runs get_image_embeds in fp16 on the device in the reference; on the CPU stand-in keep fp32
    orig_to = torch.Tensor.to
    d = dict(emb=npf(emb), ctx=npf(ctx), nctx=npf(nctx), pooled=npf(pooled), npooled=npf(npooled), xT=npf(xT))
    try:
        torch.Tensor.to = lambda self, *a, **k: orig_to(self, *[x for x in a if x is not torch.float16], **{kk: vv for kk, vv in k.items() if vv is not torch.float16})
        for tag, kw in (("g4_s07", dict(scale=0.7, guidance_scale=4.0)), ("g10_s10", dict(scale=1.0, guidance_scale=10.0))):
            out = ad.generate(None, clip_image_embeds=emb[None], num_inference_steps=N, mode="global", latents=xT.clone(), height=h * 8, width=w * 8,
                              output_type="latent", **kw)
            d[tag] = npf(out)
    finally:
        torch.Tensor.to = orig_to
    np.savez_compressed(os.path.join(HERE, "sample_loop.npz"), **d)
    print({k: v.shape for k, v in d.items()})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "sample":
        with torch.no_grad():
            gen_sample()
    elif len(sys.argv) > 1 and sys.argv[1] == "inverse":
        with torch.no_grad():
            gen_inverse()
    elif len(sys.argv) > 1 and sys.argv[1] == "encode_prompt":
        with torch.no_grad():
            gen_encode_prompt()
    elif len(sys.argv) > 1 and sys.argv[1] == "prior":
        with torch.no_grad():
            gen_prior()
    elif len(sys.argv) > 1 and sys.argv[1] == "refiner":
        with torch.no_grad():
            gen_refiner()
    else:
        main()
