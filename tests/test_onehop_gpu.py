"""One-hop parity: the HIP path (through the C ABI) against the golden vectors that tests/golden/gen_goldens.py produced by running the
REFERENCE's own code (method text of `SDXLDDIMPipeline.inverse`, `IPAdapterXL.generate` + the vendored SDXL `__call__`,
`ImageProjModel`, `_backward_ddim`) -- no oracle in between. tests/test_oracle_golden.py pins the oracle on the same files on the CPU.

  G12 inverse_loop.npz   <- instructany2pix/ddim/pnp_pipeline.py:92-278
  G13 sample_loop.npz    <- diffusion/ip_adapter/ip_adapter.py:289-356 -> ddim/sdxl_pipeline.py:544-886
  G3  image_proj.npz     <- diffusion/ip_adapter/ip_adapter.py:28-67
  G4  backward_ddim.npz  <- ddim/pnp_pipeline.py:73-85
Tolerances: trajectories rel-L2 <= 3e-2, cosine >= 0.999; ImageProjModel rel-L2 <= 2e-3; a single fp32 DDIM
update with one fp16 rounding <= 1 fp16 ulp of the result.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
T = torch.from_numpy


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def traj_metrics(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))


@pytest.fixture(scope="module")
def tiny_ip():
    """tiny UNet with the IP-Adapter processors installed the reference's way (ip_adapter.py:120-142,168-169), seed-7 weights: the very
    weights gen_goldens.py loaded into the module tree it ran the reference's loops over"""
    from instructany2pix_amd.attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict, hidden_size_of
    cfg = tiny()
    specs = ip_adapter_specs(cfg, 64)
    sd, ipsd = synthetic_state_dict(unet_param_specs(cfg), seed=7), synthetic_state_dict(specs["ip_adapter"], seed=7)
    hip = HipUNet2DConditionModel(cfg, DEV)
    hip.load_state_dict(sd)
    procs = {}
    for n in hip.attn_processors:
        procs[n] = AttnProcessor2_0() if n.endswith("attn1.processor") else \
            IPAttnProcessor2_0(hidden_size_of(cfg, n), cfg.cross_attention_dim, scale=1.0, num_tokens=4).to(DEV, torch.float16)
    hip.set_attn_processor(procs)
    torch.nn.ModuleList(hip.attn_processors.values()).load_state_dict(ipsd)
    return cfg, hip, ipsd, synthetic_state_dict(specs["image_proj"], seed=7)


def test_g12_hip_inversion_vs_reference_method_output(tiny_ip, golden):
    """`SDXLDDIMPipeline.inverse` on the HIP UNet == the output of the reference's own `inverse` text: ascending timesteps,
    `final_alpha_cumprod` on the first move, no guidance, 77-token context on the IP-enabled UNet (the shared-UNet quirk)."""
    from instructany2pix_amd.ddim import SDXLDDIMPipeline
    cfg, hip, _, _ = tiny_ip
    d = golden("inverse_loop.npz")
    for p in hip.attn_processors.values():
        if hasattr(p, "scale"):
            p.scale = 1.0
    for n in (5, 12):
        inv = SDXLDDIMPipeline(hip).inverse(latents=T(d["x0"]).half(), prompt_embeds=T(d["ctx"]).half(), pooled_prompt_embeds=T(d["pooled"]).half(),
                                            num_inference_steps=n).images
        r, c = traj_metrics(inv, T(d[f"inv{n}"]))
        assert r < 3e-2 and c > 0.999, (n, r, c)


def test_g13_hip_generate_vs_reference_method_output(tiny_ip, golden):
    """`IPAdapterXL.generate` on the HIP path (HIP ImageProjModel, set_scale, context assembly, guided DDIM loop) == the output of the
    reference's `IPAdapterXL.generate` -> vendored SDXL `__call__` text."""
    from instructany2pix_amd.ddim import StableDiffusionXLPipeline
    from instructany2pix_amd.ip_adapter import IPAdapterXL
    cfg, hip, ipsd, projsd = tiny_ip
    d = golden("sample_loop.npz")
    ipa = IPAdapterXL(StableDiffusionXLPipeline(hip), "", ip_ckpt={"image_proj": projsd, "ip_adapter": ipsd}, device=DEV, clip_embeddings_dim=64)
    h = lambda k: T(d[k]).half()
    for tag, scale, g in (("g4_s07", 0.7, 4.0), ("g10_s10", 1.0, 10.0)):
        lat = ipa.generate(clip_image_embeds=T(d["emb"]), prompt_embeds=h("ctx"), negative_prompt_embeds=h("nctx"), pooled_prompt_embeds=h("pooled"),
                           negative_pooled_prompt_embeds=h("npooled"), num_inference_steps=6, scale=scale, guidance_scale=g, latents=h("xT"),
                           height=128, width=128, output_type="latent")
        r, c = traj_metrics(lat, T(d[tag]))
        assert r < 3e-2 and c > 0.999, (tag, r, c)


def test_g3_hip_image_proj_vs_reference_class_output(golden):
    """HIP `ImageProjModel` (ia2p_linear_small + ia2p_layernorm) == the reference class on its own fixture: global / local / both modes,
    the local blend scale, the zero embedding of the unconditional branch, and the assertion on an unknown mode."""
    from instructany2pix_amd.ip_adapter import ImageProjModel
    d = golden("image_proj.npz")
    m = ImageProjModel(cross_attention_dim=64, clip_embeddings_dim=48, clip_extra_context_tokens=4)
    m.load_state_dict({"proj.weight": T(d["proj_weight"]), "proj.bias": T(d["proj_bias"]), "norm.weight": T(d["norm_weight"]),
                       "norm.bias": T(d["norm_bias"]), "raw_embed": T(d["raw_embed"])})
    m = m.to(DEV, torch.float16)
    emb = T(d["emb"]).to(DEV)
    for mode in ("global", "local", "both"):
        for sl in (1.0, 0.5):
            o = m(emb, mode, scales=(1.0, sl))
            assert tuple(o.shape) == d[f"out_{mode}_{sl}"].shape
            assert rel_l2(o, T(d[f"out_{mode}_{sl}"])) < 2e-3, (mode, sl)
    assert rel_l2(m(torch.zeros_like(emb), "global"), T(d["out_zero_global"])) < 2e-3
    with pytest.raises(AssertionError):
        m(emb, "bogus")


def test_g4_hip_ddim_step_vs_reference_backward_ddim(golden):
    """`ia2p_ddim_step` with `DDIMScheduler.inversion_coeffs` == the reference's `_backward_ddim` over the 20/25/50-step schedules:
    every single move from the fixture's previous point (one fp16 rounding of an fp32 update), the chained fp16 trajectory, and the
    half-precision call the reference actually makes."""
    from instructany2pix_amd.scheduler import DDIMScheduler, fused_update
    d = golden("backward_ddim.npz")
    for n in (20, 25, 50):
        s = DDIMScheduler()
        s.set_timesteps(n)
        assert np.array_equal(s.timesteps.numpy(), d[f"timesteps{n}"])
        traj = T(d[f"traj{n}"])
        chained = T(d["x0"]).half().to(DEV)
        prev = None
        for i, t in enumerate(reversed(s.timesteps)):
            t = int(t)
            a_p = float(s.alphas_cumprod[prev]) if prev is not None else float(s.final_alpha_cumprod)
            c_x, c_e = DDIMScheduler.inversion_coeffs(float(s.alphas_cumprod[t]), a_p)
            prev = t
            eps = T(d["eps"][i]).half().to(DEV)
            src = (T(d["x0"]) if i == 0 else traj[i - 1]).half()
            out = fused_update(src.to(DEV), eps, None, 1.0, c_x, c_e, torch.empty_like(chained))
            want = c_x * src.float() + c_e * eps.float().cpu()                     # the same move on the same fp16 inputs, fp32
            ulp = torch.maximum(want.abs(), torch.tensor(6.1e-5)) * 2.0 ** -10
            assert ((out.float().cpu() - want).abs() <= ulp).all(), (n, i)
            assert float((out.float().cpu() - traj[i]).abs().max()) <= 4e-3 * float(traj[i].abs().max()), (n, i)   # fp16 inputs vs the fp32 fixture
            nxt = fused_update(chained, eps, None, 1.0, c_x, c_e, torch.empty_like(chained))
            chained = nxt
        r, c = traj_metrics(chained, traj[-1])
        assert r < 1e-2 and c > 0.9999, (n, r, c)
    s = DDIMScheduler()
    c_x, c_e = DDIMScheduler.inversion_coeffs(float(s.alphas_cumprod[501]), float(s.alphas_cumprod[481]))
    x, e = T(d["x0"]).half().to(DEV), T(d["eps"][0]).half().to(DEV)
    got = fused_update(x, e, None, 1.0, c_x, c_e, torch.empty_like(x)).float().cpu()
    ref = T(d["half_step"])                                     # the reference's fp16 tensor arithmetic rounds after every op: <= 3 ulp apart
    assert float(((got - ref).abs() / torch.maximum(ref.abs(), torch.tensor(1e-3))).max()) < 4 * 2.0 ** -10


def test_tiny_50_step_trajectories_vs_oracle():
    """SURVEY.md Appendix A's trajectory bar at its full length: 50-step inversion and 50-step guided sampling on the tiny config."""
    import oracle
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.ddim import SDXLDDIMPipeline, StableDiffusionXLPipeline
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    cfg = tiny()
    torch.set_num_threads(8)                 # the tiny oracle's operators are small: more threads only add synchronisation
    sd = synthetic_state_dict(unet_param_specs(cfg), seed=7)
    ipsd = synthetic_state_dict(ip_adapter_specs(cfg, 64)["ip_adapter"], seed=7)
    hip = HipUNet2DConditionModel(cfg, DEV)
    hip.load_state_dict(sd)
    hip.load_ip_adapter_weights(ipsd, scale=1.0, num_tokens=4)
    ref = oracle.build_unet(cfg, sd, ipsd, ip_scale=1.0)
    B, h, N = 1, 16, 50
    g = torch.Generator().manual_seed(50)
    x0 = torch.randn(B, 4, h, h, generator=g).half()
    ctx77 = torch.randn(B, 77, cfg.cross_attention_dim, generator=g).half()
    ctx81, neg81 = torch.randn(B, 81, cfg.cross_attention_dim, generator=g).half(), torch.randn(B, 81, cfg.cross_attention_dim, generator=g).half()
    pooled, npooled = torch.randn(B, cfg.pooled_dim, generator=g).half(), torch.randn(B, cfg.pooled_dim, generator=g).half()
    tid = torch.tensor([[h * 8.0, h * 8.0, 0, 0, h * 8.0, h * 8.0]] * B)
    inv = SDXLDDIMPipeline(hip).inverse(latents=x0, prompt_embeds=ctx77, pooled_prompt_embeds=pooled, num_inference_steps=N).images
    ref_inv = oracle.invert_loop(ref, oracle.DDIMSchedulerRef(), x0.float(), ctx77.float(), dict(text_embeds=pooled.float(), time_ids=tid), N)
    r, c = traj_metrics(inv, ref_inv)
    assert r < 3e-2 and c > 0.999, ("inversion", r, c)
    xT = torch.randn(B, 4, h, h, generator=g).half()
    out = StableDiffusionXLPipeline(hip)(prompt_embeds=ctx81, negative_prompt_embeds=neg81, pooled_prompt_embeds=pooled, negative_pooled_prompt_embeds=npooled,
                                         num_inference_steps=N, latents=xT, guidance_scale=5.0, height=h * 8, width=h * 8).images
    ref_out = oracle.sample_loop(ref, oracle.DDIMSchedulerRef(), xT.float(), ctx81.float(), dict(text_embeds=pooled.float(), time_ids=tid), N, 5.0,
                                 neg81.float(), dict(text_embeds=npooled.float(), time_ids=tid))
    r, c = traj_metrics(out, ref_out)
    assert r < 3e-2 and c > 0.999, ("sampling", r, c)
