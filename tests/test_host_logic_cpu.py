"""Host-side logic of the product package that needs no GPU: scheduler tables and DDIM coefficients (against the
golden vectors generated from the reference files and against the oracle), parameter inventory, pipeline argument
checks, config plumbing."""
import math

import numpy as np
import pytest
import torch

import oracle
from instructany2pix_amd.config import sdxl_base, tiny
from instructany2pix_amd.scheduler import DDIMScheduler
from instructany2pix_amd.weights import attn_processor_names, hidden_size_of, ip_adapter_specs, param_count, unet_param_specs


def test_scheduler_tables_match_golden(golden):
    d = golden("schedule.npz")
    s = DDIMScheduler()
    assert np.abs(s.betas.numpy() - d["betas"]).max() < 1e-8
    assert np.abs(s.alphas_cumprod.numpy() - d["alphas_cumprod"]).max() < 2e-6
    g = golden("backward_ddim.npz")
    for n in (20, 25, 50):
        s.set_timesteps(n)
        assert np.array_equal(s.timesteps.numpy(), g[f"timesteps{n}"])          # "leading" spacing + offset 1
        assert np.array_equal(s.timesteps.numpy()[::-1], d[f"ts{n}"])
    assert float(s.final_alpha_cumprod) == float(s.alphas_cumprod[0])            # set_alpha_to_one = False
    s2 = DDIMScheduler.from_config(s.config)
    assert torch.equal(s2.alphas_cumprod, s.alphas_cumprod)
    with pytest.raises(NotImplementedError):
        DDIMScheduler(beta_schedule="linear")
    with pytest.raises(ValueError):
        s.set_timesteps(2000)


def test_inversion_coefficients_reproduce_backward_ddim(golden):
    """out = c_x*x + c_e*eps with inversion_coeffs == reference _backward_ddim (pnp_pipeline.py:73-85) trajectory."""
    d = golden("backward_ddim.npz")
    s = DDIMScheduler()
    for n in (20, 50):
        s.set_timesteps(n)
        lat = torch.from_numpy(d["x0"]).double()
        prev = None
        for i, t in enumerate(reversed(s.timesteps.tolist())):
            a_t = float(s.alphas_cumprod[t])
            a_p = float(s.alphas_cumprod[prev]) if prev is not None else float(s.final_alpha_cumprod)
            prev = t
            cx, ce = DDIMScheduler.inversion_coeffs(a_t, a_p)
            lat = cx * lat + ce * torch.from_numpy(d["eps"][i]).double()
            ref = d[f"traj{n}"][i]
            assert np.abs(lat.numpy() - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max())


def test_step_coefficients_match_oracle_step():
    s, o = DDIMScheduler(), oracle.DDIMSchedulerRef()
    g = torch.Generator().manual_seed(1)
    x, e = torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64), torch.randn(2, 4, 8, 8, generator=g, dtype=torch.float64)
    for n in (20, 25, 50):
        s.set_timesteps(n); o.set_timesteps(n)
        for t in s.timesteps.tolist():
            cx, ce = s.step_coeffs(t)
            ref = o.step(e, t, x)
            assert (cx * x + ce * e - ref).abs().max() < 1e-6
    # the last step lands on final_alpha_cumprod (prev timestep < 0)
    s.set_timesteps(50)
    cx, _ = s.step_coeffs(1)
    assert math.isclose(cx, math.sqrt(float(s.final_alpha_cumprod) / float(s.alphas_cumprod[1])), rel_tol=1e-12)


def test_parameter_inventory():
    cfg = sdxl_base()
    specs = unet_param_specs(cfg)
    assert param_count(specs) == 2_567_463_684 and len(specs) == 1680             # published SDXL-base UNet size
    names = attn_processor_names(cfg)
    assert len(names) == 140 and names[0].startswith("down_blocks.1") and names[-1].startswith("mid_block")
    assert names[0].endswith("attn1.processor") and names[1].endswith("attn2.processor")
    ip = ip_adapter_specs(cfg)
    keys = [k for k, _, _ in ip["ip_adapter"]]
    assert keys[0] == "1.to_k_ip.weight" and keys[-1] == "139.to_v_ip.weight" and len(keys) == 140
    assert param_count(ip["ip_adapter"]) == 340_787_200
    assert hidden_size_of(cfg, "up_blocks.0.attentions.2.transformer_blocks.9.attn2.processor") == 1280
    assert hidden_size_of(cfg, "up_blocks.1.attentions.0.transformer_blocks.0.attn2.processor") == 640
    assert [k for k, _, _ in ip["image_proj"]] == ["proj.weight", "proj.bias", "norm.weight", "norm.bias", "raw_embed"]


def test_oracle_and_product_enumerate_the_same_keys():
    for cfg in (tiny(), tiny(depths=(0, 2, 1))):
        with torch.device("meta"):
            m = oracle.UNet2DConditionModelRef(cfg)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == {k: s for k, s, _ in unet_param_specs(cfg)}
        assert list(m.attn_processors.keys()) == attn_processor_names(cfg)


def test_add_time_ids_and_argument_checks():
    from types import SimpleNamespace
    from instructany2pix_amd.ddim import SDXLDDIMPipeline, StableDiffusionXLPipeline, get_add_time_ids
    cfg = sdxl_base()
    fake = SimpleNamespace(config=cfg, device=torch.device("cpu"),
                           add_embedding=SimpleNamespace(linear_1=SimpleNamespace(in_features=cfg.projection_class_embeddings_input_dim)))
    ids = get_add_time_ids(fake, (1024, 1024), (0, 0), (1024, 1024), 1280, torch.float32)
    assert ids.tolist() == [[1024.0, 1024.0, 0.0, 0.0, 1024.0, 1024.0]]
    with pytest.raises(ValueError):                                  # embed-dim mismatch (reference pnp_pipeline.py:63-66)
        get_add_time_ids(fake, (1024, 1024), (0, 0), (1024, 1024), 1024)
    inv, smp = SDXLDDIMPipeline(fake), StableDiffusionXLPipeline(fake)
    with pytest.raises(ValueError):
        inv.inverse(prompt="a", prompt_embeds=torch.zeros(1, 77, 2048), pooled_prompt_embeds=torch.zeros(1, 1280))
    with pytest.raises(ValueError):
        inv.inverse(latents=torch.zeros(1, 4, 8, 8))                 # neither prompt nor prompt_embeds
    with pytest.raises(ValueError):
        inv.inverse(prompt_embeds=torch.zeros(1, 77, 2048), latents=torch.zeros(1, 4, 8, 8))   # pooled missing
    with pytest.raises(ValueError):
        inv.inverse(prompt_embeds=torch.zeros(1, 77, 2048), pooled_prompt_embeds=torch.zeros(1, 1280), latents=torch.zeros(1, 4, 8, 8), strength=1.5)
    with pytest.raises(NotImplementedError):                         # text encoders are out of scope: needs an injected callable
        inv.inverse(prompt="", latents=torch.zeros(1, 4, 8, 8))
    with pytest.raises(ValueError):
        smp(prompt_embeds=torch.zeros(1, 77, 2048), pooled_prompt_embeds=torch.zeros(1, 1280), guidance_scale=5.0)   # CFG needs negatives
    with pytest.raises(ValueError):
        smp(prompt_embeds=torch.zeros(1, 77, 2048), pooled_prompt_embeds=torch.zeros(1, 1280), guidance_scale=1.0, height=100, width=64)


def test_polar_interpolate_and_fusion_match_golden(golden):
    from instructany2pix_amd.pipeline import fuse_instruction_embedding, polar_intrtpolate
    d = golden("misc.npz")
    a, b = torch.from_numpy(d["pa"]), torch.from_numpy(d["pb"])
    assert np.abs(polar_intrtpolate(a, b, 0.7).numpy() - d["polar_07"]).max() < 1e-6
    assert np.array_equal(polar_intrtpolate(a.half(), b.half(), 0.7).float().numpy(), d["polar_half"])     # fp16 CPU arithmetic kept
    g = torch.Generator().manual_seed(2)
    be, ie, y = torch.randn(1024, generator=g), torch.randn(1, 1, 1024, generator=g), torch.randn(1, 1024, generator=g)
    la = fuse_instruction_embedding(be, ie, y, [0.0, 0.4, 1.0], 20.0)      # reference pipeline.py:322-324
    assert math.isclose(float(la.norm()), 20.0, rel_tol=1e-5)
    ref = be * 0.0 + ie * 0.4 + y / y.norm() * 20.0
    assert torch.allclose(la, ref / ref.norm() * 20.0, atol=1e-6)


# ---- refiner pass: Euler tables, schedule tail, 5-id micro-conditioning ----------------------
def test_refiner_config_inventory():
    from instructany2pix_amd.config import sdxl_refiner, tiny_refiner
    cfg = sdxl_refiner()
    assert param_count(unet_param_specs(cfg)) == 2_259_526_660           # = the 4.52 GB fp16 refiner UNet checkpoint
    assert cfg.pooled_dim == 1280 and cfg.num_time_ids == 5 and cfg.mid_block_transformer_layers == 4
    names = attn_processor_names(cfg)
    assert len(names) == 2 * 4 * (2 * 2 + 3 * 2 + 1)                      # 4 layers x (down 2x2, up 2x3, mid) x (attn1, attn2)
    assert names[-1] == "mid_block.attentions.0.transformer_blocks.3.attn2.processor"
    for c in (cfg, tiny_refiner()):
        with torch.device("meta"):
            ref = oracle.UNet2DConditionModelRef(c)
        assert {k: tuple(v.shape) for k, v in ref.state_dict().items()} == {k: s for k, s, _ in unet_param_specs(c)}
        assert list(ref.attn_processors.keys()) == attn_processor_names(c)


def test_euler_scheduler_tables_and_coefficients_match_oracle():
    from instructany2pix_amd.scheduler import EulerDiscreteScheduler
    s, r = EulerDiscreteScheduler(), oracle.EulerDiscreteSchedulerRef()
    assert torch.equal(s.alphas_cumprod, DDIMScheduler().alphas_cumprod)    # same beta table (pinned by G5)
    for n in (20, 50):
        s.set_timesteps(n)
        r.set_timesteps(n)
        assert np.array_equal(s.timesteps.numpy(), r.timesteps)
        assert np.allclose(s.sigmas.numpy(), r.sigmas, rtol=3e-7, atol=0)        # torch vs numpy float32 pow: 1 ulp
        assert s.timesteps.dtype == torch.float32 and float(s.timesteps[0]) == 1000 // n * (n - 1) + 1 and float(s.sigmas[-1]) == 0.0
        assert abs(s.init_noise_sigma - math.sqrt(float(r.sigmas.max()) ** 2 + 1)) < 1e-6
        x, e = torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(n)), torch.randn(2, 4, 8, 8, generator=torch.Generator().manual_seed(n + 1))
        for i in (0, n // 2, n - 1):
            c_x, c_e = s.step_coeffs(i)
            assert (c_x * x + c_e * e - r.step(e, i, x)).abs().max() < 2e-5
            assert (s.input_scale(i) * x - r.scale_model_input(x, i)).abs().max() < 1e-6
            assert s.index_for_timestep(s.timesteps[i]) == i
    with pytest.raises(ValueError):
        s.index_for_timestep(3.0)
    with pytest.raises(NotImplementedError):
        EulerDiscreteScheduler(use_karras_sigmas=True)


def test_img2img_schedule_tail_and_time_ids_match_reference_golden(golden):
    from types import SimpleNamespace
    from instructany2pix_amd.img2img import StableDiffusionXLImg2ImgPipeline, get_add_time_ids_aesthetic
    d = golden("misc_refiner.npz")

    def unet(in_features):
        return SimpleNamespace(config=SimpleNamespace(addition_time_embed_dim=256), add_embedding=SimpleNamespace(linear_1=SimpleNamespace(in_features=in_features)),
                               device="cpu")
    ids, neg = get_add_time_ids_aesthetic(unet(2560), (1024, 1024), (0, 0), (1024, 1024), 6.0, 2.5, (1024, 1024), (0, 0), (1024, 1024), 1280, True, torch.float32)
    assert np.array_equal(ids.numpy(), d["time_ids"]) and np.array_equal(neg.numpy(), d["neg_time_ids"])
    ids, neg = get_add_time_ids_aesthetic(unet(2560), (768, 512), (8, 16), (768, 512), 7.5, 1.0, (640, 384), (4, 2), (768, 512), 1280, True, torch.float32)
    assert np.array_equal(ids.numpy(), d["time_ids_b"]) and np.array_equal(neg.numpy(), d["neg_time_ids_b"])
    kinds = dict(k.split(": ") for k in d["err_kinds"])
    for name, requires, feats in (("err_enable", False, 3072), ("err_enable2", True, 2816), ("err_disable", False, 2560), ("err_config", True, 2000)):
        assert int(d[name]) == 1
        with pytest.raises(ValueError) as ei:
            get_add_time_ids_aesthetic(unet(feats), (1024, 1024), (0, 0), (1024, 1024), 6.0, 2.5, (1024, 1024), (0, 0), (1024, 1024), 1280, requires, torch.float32)
        msg = str(ei.value)
        assert kinds[name] == ("enable" if "to enable" in msg else "disable" if "to disable" in msg else "config")
    p = StableDiffusionXLImg2ImgPipeline(unet(2560))
    p.scheduler.set_timesteps(50)
    ts, n, t0 = p.get_timesteps(50, 0.5)                         # the reference's default refinement = 0.5
    assert n == 25 and t0 == 25 and float(ts[0]) == 481.0 and float(ts[-1]) == 1.0
    ts, n, t0 = p.get_timesteps(50, 1.0)
    assert n == 50 and t0 == 0
    with pytest.raises(ValueError):
        p(prompt_embeds=torch.zeros(1, 77, 1280), pooled_prompt_embeds=torch.zeros(1, 1280), latents=torch.zeros(1, 4, 8, 8), strength=1.5)
    with pytest.raises(ValueError):
        p(prompt_embeds=torch.zeros(1, 77, 1280), pooled_prompt_embeds=torch.zeros(1, 1280), latents=torch.zeros(1, 4, 8, 8), strength=0.0, guidance_scale=1.0)


def test_inpaint_mask_preparation_and_add_noise_coefficients():
    from instructany2pix_amd.inpaint import prepare_mask
    g = torch.Generator().manual_seed(4)
    soft = torch.rand(128, 128, generator=g)
    m = prepare_mask(soft, 16, 16, "cpu")
    assert m.shape == (1, 1, 16, 16) and m.dtype == torch.float16 and set(m.unique().tolist()) <= {0.0, 1.0}
    ref = torch.nn.functional.interpolate((soft[None, None] >= 0.5).float(), size=(16, 16))        # binarise, then nearest resize
    assert torch.equal(m.float(), ref)
    u8 = (soft * 255).to(torch.uint8)
    assert torch.equal(prepare_mask(u8, 16, 16, "cpu").float(), torch.nn.functional.interpolate((u8.float()[None, None] / 255 >= 0.5).float(), size=(16, 16)))
    assert prepare_mask(torch.ones(2, 1, 32, 32), 8, 8, "cpu").shape == (2, 1, 8, 8)
    with pytest.raises(ValueError):
        prepare_mask(torch.ones(1, 3, 8, 8), 8, 8, "cpu")
    s, r = DDIMScheduler(), oracle.DDIMSchedulerRef()
    x0, n = torch.randn(1, 4, 8, 8, generator=g), torch.randn(1, 4, 8, 8, generator=g)
    for t in (1, 481, 981):
        c0, c1 = s.add_noise_coeffs(t)
        assert (c0 * x0 + c1 * n - oracle.add_noise(r, x0, n, t)).abs().max() < 1e-6 and abs(c0 * c0 + c1 * c1 - 1) < 1e-6


def test_encode_prompt_host_logic_matches_reference_golden(golden):
    """`SDXLTextEncoders.encode_prompt` (the product's host logic, here over CPU stand-in encoders) against fixture G11, the output of
    the reference's own `encode_prompt` text: every branch the reference has (negatives, zeros, empty string, no CFG, both errors)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from stub_tokenizer import StubTokenizer
    from types import SimpleNamespace
    from instructany2pix_amd.clip import SDXLTextEncoders
    from instructany2pix_amd.config import tiny_clip
    from instructany2pix_amd.weights import clip_param_specs, synthetic_state_dict
    d = golden("encode_prompt.npz")
    c1, c2 = tiny_clip(0, "quick_gelu"), tiny_clip(64, "gelu")
    r1 = oracle.build_clip(c1, synthetic_state_dict(clip_param_specs(c1), seed=31))
    r2 = oracle.build_clip(c2, synthetic_state_dict(clip_param_specs(c2), seed=32))

    class Enc:                                            # what HipCLIPTextModel returns, computed by the oracle on the CPU
        def __init__(self, m):
            self.m = m

        def __call__(self, ids, output_hidden_states=True, want_pooled=True):
            pooled, last, hidden = self.m(ids)
            out = SimpleNamespace(hidden_states=hidden)
            return type("O", (), {"__getitem__": lambda s, i: pooled if want_pooled else None, "hidden_states": hidden})()

    mk = lambda force: SDXLTextEncoders(StubTokenizer(1, c1.vocab_size), StubTokenizer(2, c1.vocab_size), Enc(r1), Enc(r2), force_zeros_for_empty_prompt=force)
    prompts, negs = ["a photo of a cat", "two dogs on the beach at sunset"], ["blurry", "low quality, bad anatomy"]
    close = lambda a, k: np.abs(a.numpy() - d[k]).max() < 1e-5
    pe, ne, pp, npl = mk(True).encode_prompt(prompts, negative_prompt=negs, num_images_per_prompt=2)
    assert close(pe, "pair_pe") and close(ne, "pair_ne") and close(pp, "pair_pp") and close(npl, "pair_np")
    pe, ne, pp, npl = mk(True).encode_prompt("a photo of a cat")
    assert close(pe, "zeros_pe") and close(pp, "zeros_pp") and close(ne, "zeros_ne") and close(npl, "zeros_np")
    pe, ne, pp, npl = mk(False).encode_prompt("a photo of a cat")
    assert close(pe, "empty_pe") and close(ne, "empty_ne") and close(npl, "empty_np")
    pe, ne, pp, npl = mk(True).encode_prompt(prompts, do_classifier_free_guidance=False)
    assert close(pe, "nocfg_pe") and close(pp, "nocfg_pp") and ne is None and npl is None
    with pytest.raises(TypeError):
        mk(True).encode_prompt("a cat", negative_prompt=["x"])
    with pytest.raises(ValueError):
        mk(True).encode_prompt(["a", "b"], negative_prompt=["x"])


def test_prior_host_side_tables_and_inventory():
    """DDPM tables / posterior coefficients vs the oracle, the sinusoid vs the oracle's (G6-pinned) form, parameter inventory of the
    released prior (gpt2-medium 354.8 M + CLIP ViT-H text tower 353.0 M + slot tables), the constructor's key list."""
    import oracle
    from instructany2pix_amd.config import gpt2_medium, laion_clip_h_text
    from instructany2pix_amd.prior import MODALITY, get_timestep_embedding, prior_config
    from instructany2pix_amd.scheduler import DDPMScheduler
    from instructany2pix_amd.weights import gpt2_param_specs, clip_param_specs, prior_param_specs, param_count
    sch, ref = DDPMScheduler(), oracle.DDPMSchedulerRef()
    assert torch.equal(sch.alphas_cumprod, ref.alphas_cumprod)
    for n in (1, 3, 25, 50):
        sch.set_timesteps(n); ref.set_timesteps(n)
        assert sch.timesteps.tolist() == ref.timesteps.tolist() and sch.timesteps[-1] == 1
        for t in sch.timesteps.tolist():
            sa, sb, k0, k1, sigma = sch.posterior_coeffs(t)
            x, e = torch.full((1,), 0.7), torch.full((1,), -0.3)
            want = ref.step(e, t, x, generator=torch.Generator().manual_seed(1))[0]
            z = torch.randn(1, generator=torch.Generator().manual_seed(1))
            got = k0 * ((x - sb * e) / sa) + k1 * x + sigma * z
            assert abs(float(got - want)) < 2e-6
    with pytest.raises(ValueError):
        sch.set_timesteps(1001)
    t = torch.tensor([6.5, 981.0])
    assert torch.equal(get_timestep_embedding(t, 512), oracle.timestep_embedding_ref(t, 512))
    with pytest.raises(NotImplementedError):
        get_timestep_embedding(t, 512, flip_sin_to_cos=False)
    assert param_count(gpt2_param_specs(gpt2_medium())) == 354_823_168
    assert param_count(clip_param_specs(laion_clip_h_text())) == 352_984_064
    assert param_count(prior_param_specs(gpt2_medium(), laion_clip_h_text())) == 710_507_520
    assert (MODALITY.IMAGE, MODALITY.AUDIO, MODALITY.TEXT, MODALITY.VIDEO) == (0, 1, 2, 3)
    assert prior_config["sequence_input_key"] == oracle.PriorRef.sequence_input_key and prior_config["sequence_input_embed_dim"] == [0, 1024, 1024, 512, 0, 0, 0]


def test_iter_safetensors_streams_and_validates_a_checkpoint(tmp_path):
    """`weights.iter_safetensors` (the IA2P_UNET_WEIGHTS loader): single file, diffusers directory layout, sharded index; key-set and
    shape validation against the architecture."""
    import json
    from safetensors.torch import save_file
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.weights import iter_safetensors, synthetic_state_dict, unet_param_specs
    cfg = tiny()
    specs = unet_param_specs(cfg)
    sd = synthetic_state_dict(specs, seed=7)
    d = tmp_path / "unet"
    d.mkdir()
    one = str(d / "diffusion_pytorch_model.safetensors")
    save_file(dict(sd), one)
    for src in (one, str(d)):
        got = dict(iter_safetensors(src, specs))
        assert list(got) == [k for k, _, _ in specs] and all(torch.equal(got[k], sd[k]) for k in sd)
    # sharded: two files + index, keys with a prefix
    keys = list(sd)
    sh = tmp_path / "sharded"
    sh.mkdir()
    a, b = {("unet." + k): sd[k] for k in keys[::2]}, {("unet." + k): sd[k] for k in keys[1::2]}
    save_file(a, str(sh / "m-00001-of-00002.safetensors")); save_file(b, str(sh / "m-00002-of-00002.safetensors"))
    json.dump({"weight_map": {**{k: "m-00001-of-00002.safetensors" for k in a}, **{k: "m-00002-of-00002.safetensors" for k in b}}},
              open(sh / "m.safetensors.index.json", "w"))
    got = dict(iter_safetensors(str(sh), specs, prefix="unet."))
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    # validation
    bad = dict(sd)
    bad.pop(keys[3])
    save_file(bad, str(tmp_path / "missing.safetensors"))
    with pytest.raises(KeyError):
        next(iter_safetensors(str(tmp_path / "missing.safetensors"), specs))
    bad = dict(sd)
    bad[keys[3]] = torch.zeros(3, dtype=torch.float16)
    save_file(bad, str(tmp_path / "shape.safetensors"))
    with pytest.raises(ValueError):
        next(iter_safetensors(str(tmp_path / "shape.safetensors"), specs))
    assert len(dict(iter_safetensors(str(tmp_path / "missing.safetensors")))) == len(sd) - 1           # without specs: whatever is there


def test_denoise_batch_validates_guidance_and_group_before_touching_the_device():
    """a batched group always blends eps_u + g (eps_c - eps_u); the reference switches guidance off at cfg <= 1 (one conditional evaluation) -- such a request is
    refused with a message that says where it belongs; groups are 1..8 requests (B_eff <= 16 rows)"""
    import types
    import pytest
    import torch
    from instructany2pix_amd.batch import EditRequest, denoise_batch
    z = torch.zeros
    req = lambda cfg: EditRequest(base_latents=z(1, 4, 8, 8), latent_la=z(64), prompt_embeds=z(1, 77, 64), pooled_prompt_embeds=z(1, 64),
                                  negative_prompt_embeds=z(1, 77, 64), negative_pooled_prompt_embeds=z(1, 64), cfg=cfg)
    pipe = types.SimpleNamespace(ip_adapter_xl=object())
    for bad in (1.0, 0.5, 0.0):
        with pytest.raises(ValueError, match="NO guidance"):
            denoise_batch(pipe, [req(7.5), req(bad)])
    for g in (0, 9, 64):
        with pytest.raises(ValueError, match="group="):
            denoise_batch(pipe, [req(7.5)], group=g)
    with pytest.raises(ValueError, match="at least one request"):
        denoise_batch(pipe, [])


def test_the_reference_import_name_resolves_to_the_native_pipeline():
    """the call surface to keep starts at `from instructany2pix import InstructAny2PixPipeline` (reference instructany2pix/__init__.py:1). The alias package
    forwards every name to instructany2pix_amd, lazily: importing it loads neither torch-side modules nor the HIP library."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; import instructany2pix; assert 'instructany2pix_amd.pipeline' not in sys.modules and 'torch' not in sys.modules; "
            "from instructany2pix import InstructAny2PixPipeline, DDIMScheduler; import instructany2pix_amd.pipeline as p, instructany2pix_amd.scheduler as s; "
            "assert InstructAny2PixPipeline is p.InstructAny2PixPipeline and DDIMScheduler is s.DDIMScheduler; "
            "import inspect; sig = inspect.signature(InstructAny2PixPipeline.__call__); "
            "assert [k for k in sig.parameters][:3] == ['self', 'inst', 'mm_data'] and sig.parameters['num_inference_steps'].default == 25 and sig.parameters['cfg'].default == 10")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=root)


def test_tile_decode_division_is_exact_on_its_domain():
    """`udiv_small` of the kernels' tile decode (csrc/gemm_tile.h): q = int(float(a) * rcp(float(b))), one correction either way. Replayed here in float32 with the
    reciprocal pushed one ulp up and one ulp down (v_rcp_f32 is good to 1 ulp): exact for 0 <= a < 2^21, 0 < b < 2^21 -- the launchers refuse larger grids."""
    import numpy as np
    rng = np.random.default_rng(5)
    b = np.concatenate([rng.integers(1, 1 << 21, 200_000), rng.integers(1, 4096, 200_000), np.array([1, 2, 3, 5, 7, 255, 256, 257, (1 << 21) - 1])]).astype(np.int64)
    k = rng.integers(0, 1 << 21, b.shape[0]).astype(np.int64)
    for delta in (-1, 0, 1):                       # a around a multiple of b, where a wrong reciprocal would flip the floor
        a = np.clip((k // np.maximum(b, 1)) * b + delta, 0, (1 << 21) - 1)
        for ulp in (-1, 0, 1):
            r = (np.float32(1.0) / b.astype(np.float32)).astype(np.float32)
            r = np.nextafter(r, np.float32(np.inf if ulp > 0 else -np.inf)).astype(np.float32) if ulp else r
            q = (a.astype(np.float32) * r).astype(np.float32).astype(np.int64)      # (float -> int conversion truncates, as v_cvt_i32_f32 does)
            rem = a - q * b
            q = q + (rem >= b) - (rem < 0)
            assert np.array_equal(q, a // b), (delta, ulp)
