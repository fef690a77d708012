"""Subject-consistency inpainting loop on the MI355X: `ia2p_mask_blend`, the inpainting pipeline behind
`pipe_inpainting` (reference instructany2pix/pipeline.py:132-139, gdino/lib.py:89-102) with 'local' IP-Adapter tokens, HIP path
through the C ABI vs the CPU oracle; plus the size-independent properties of the blend (empty mask returns the input bits, full mask
equals plain sampling bits).

Tolerances as in test_unet_gpu.py: loops rel-L2 <= 3e-2, cos >= 0.999; the blend kernel itself within one fp16 rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _traj_metrics(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))


@pytest.mark.parametrize("B,C,h,w", [(2, 4, 16, 16), (1, 4, 12, 20), (3, 4, 8, 8)])
def test_mask_blend_matches_torch(B, C, h, w):
    from instructany2pix_amd.scheduler import mask_blend
    g = torch.Generator().manual_seed(B * 100 + h)
    x, init, noise = (torch.randn(B, C, h, w, generator=g).half().to(DEV) for _ in range(3))
    mask = (torch.rand(B, 1, h, w, generator=g) > 0.5).half().to(DEV)
    out, out2 = torch.empty_like(x), torch.empty_like(x)
    mask_blend(x, init, noise, mask, 0.8, 0.6, out, out2)
    ref = (1 - mask.float()) * (0.8 * init.float() + 0.6 * noise.float()) + mask.float() * x.float()
    assert torch.equal(out, out2)
    assert ((out.float() - ref).abs() <= 1e-3 * ref.abs().clamp(min=1.0)).all()      # fp32 arithmetic (FMA-contracted), one fp16 rounding
    exact = mask.bool().expand_as(x)
    assert torch.equal(out[exact], x[exact])                  # m = 1: (1-1)*keep + 1*x is x bit for bit
    soft = torch.rand(B, 1, h, w, generator=g).half().to(DEV)   # the kernel also takes soft masks
    mask_blend(x, init, noise, soft, 1.0, 0.0, out)
    assert (out.float() - ((1 - soft.float()) * init.float() + soft.float() * x.float())).abs().max() < 2e-3
    with pytest.raises(AssertionError):
        mask_blend(x, init, noise, soft[:, :, :4].contiguous(), 1.0, 0.0, out)


@pytest.fixture(scope="module")
def tiny_ip_models():
    import oracle
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.ip_adapter import IPAdapterXL
    from instructany2pix_amd.inpaint import StableDiffusionXLInpaintPipeline
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    cfg = tiny()
    sd = synthetic_state_dict(unet_param_specs(cfg), seed=7)
    specs = ip_adapter_specs(cfg, 64)
    ck = {"image_proj": synthetic_state_dict(specs["image_proj"], seed=7), "ip_adapter": synthetic_state_dict(specs["ip_adapter"], seed=7)}
    hip = HipUNet2DConditionModel(cfg, DEV)
    hip.load_state_dict(sd)
    pipe = StableDiffusionXLInpaintPipeline(hip)
    ipa = IPAdapterXL(pipe, "", ip_ckpt=ck, device=DEV, clip_embeddings_dim=64)
    m = oracle.ImageProjModelRef(cfg.cross_attention_dim, 64, 4)
    m.load_state_dict({k: v.float() for k, v in ck["image_proj"].items()})
    return cfg, sd, ck, hip, pipe, ipa, m, oracle


def _conditioning(cfg, B, seed):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    return dict(ctx=rn(B, 77, cfg.cross_attention_dim).half(), nctx=rn(B, 77, cfg.cross_attention_dim).half(), pooled=rn(B, cfg.pooled_dim).half(),
                npooled=rn(B, cfg.pooled_dim).half(), lat=rn(B, 4, 16, 16).half(), noise=rn(B, 4, 16, 16).half(), emb=rn(64), g=g)


@pytest.mark.parametrize("strength,guidance,N", [(0.7, 7.5, 10), (1.0, 4.0, 6), (0.5, 1.0, 8)])
def test_inpaint_with_local_ip_tokens_vs_oracle(tiny_ip_models, strength, guidance, N):
    cfg, sd, ck, hip, pipe, ipa, m, oracle = tiny_ip_models
    c = _conditioning(cfg, 1, int(strength * 10) + N)
    mask = torch.zeros(1, 1, 128, 128)                        # pixel-space mask (8x the latent grid), a rectangle
    mask[:, :, 32:96, 16:80] = 1.0
    out = ipa.generate(latents=c["lat"], mask_image=mask, pil_image=None, strength=strength, clip_image_embeds_local=c["emb"][None], mode="local",
                       num_inference_steps=N, scale=0.8, guidance_scale=guidance, noise=c["noise"], prompt_embeds=c["ctx"],
                       negative_prompt_embeds=c["nctx"], pooled_prompt_embeds=c["pooled"], negative_pooled_prompt_embeds=c["npooled"], output_type="latent")
    torch.cuda.synchronize()
    ref_net = oracle.build_unet(cfg, sd, ck["ip_adapter"], ip_scale=0.8)
    e = torch.stack([torch.zeros(1, 64), c["emb"].half().float()[None]], dim=1)          # global crop zero, local crop = subject (reference :196-198)
    tid = torch.tensor([[128.0, 128.0, 0, 0, 128.0, 128.0]])
    with torch.no_grad():
        p, n_ = m(e, "local", [1.0, 0.5]), m(torch.zeros_like(e), "local")
        ref = oracle.inpaint_loop(ref_net, oracle.DDIMSchedulerRef(), c["lat"].float(), c["noise"].float(), mask, torch.cat([c["ctx"].float(), p], 1),
                                  dict(text_embeds=c["pooled"].float(), time_ids=tid), N, strength, guidance, torch.cat([c["nctx"].float(), n_], 1),
                                  dict(text_embeds=c["npooled"].float(), time_ids=tid))
    r, cs = _traj_metrics(out, ref)
    assert r < 3e-2 and cs > 0.999, (r, cs)
    # outside the mask the known latents come back bit-exactly (no re-noising after the last step)
    keep = torch.nn.functional.interpolate(mask, size=(16, 16)) < 0.5
    assert torch.equal(out.cpu()[keep.expand_as(out)], c["lat"][keep.expand_as(c["lat"])])


def test_inpaint_mask_properties(tiny_ip_models):
    """empty mask: the image latents come back untouched; full mask at strength 1: exactly the plain CFG sampling loop"""
    from instructany2pix_amd.ddim import StableDiffusionXLPipeline
    cfg, sd, ck, hip, pipe, ipa, m, oracle = tiny_ip_models
    c = _conditioning(cfg, 2, 99)
    ipa.set_scale(0.8)
    ip = torch.randn(2, 4, cfg.cross_attention_dim, generator=c["g"]).half()
    kw = dict(prompt_embeds=torch.cat([c["ctx"], ip], 1), negative_prompt_embeds=torch.cat([c["nctx"], ip], 1), pooled_prompt_embeds=c["pooled"],
              negative_pooled_prompt_embeds=c["npooled"], num_inference_steps=6, guidance_scale=5.0, output_type="latent")
    out0 = pipe(latents=c["lat"], noise=c["noise"], mask_image=torch.zeros(16, 16), strength=0.8, **kw).images
    assert torch.equal(out0.cpu(), c["lat"])
    out1 = pipe(latents=c["lat"], noise=c["noise"], mask_image=torch.ones(1, 16, 16), strength=1.0, **kw).images
    plain = StableDiffusionXLPipeline(hip, pipe.scheduler)(latents=c["noise"], height=128, width=128, **kw).images
    assert torch.equal(out1, plain)
    with pytest.raises(ValueError):
        pipe(latents=c["lat"], mask_image=None, **kw)
    with pytest.raises(ValueError):
        pipe(latents=c["lat"], mask_image=torch.ones(16, 16), strength=0.0, **kw)
    with pytest.raises(ValueError):
        pipe(latents=c["lat"], mask_image=torch.ones(2, 3, 16, 16), strength=0.5, **kw)


def test_pipeline_call_runs_subject_consistency():
    """`InstructAny2PixPipeline.__call__(..., subject_strength=0.7)` re-paints each subject mask through `ip_adapter_xl_inpaint`
    (reference pipeline.py:363-368): equals running `subject_consistency` by hand on the unrefined sample; 0 skips it."""
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.inpaint import subject_consistency
    from instructany2pix_amd.pipeline import InstructAny2PixPipeline
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    cfg = tiny()
    base = HipUNet2DConditionModel(cfg, DEV)
    base.load_state_dict(synthetic_state_dict(unet_param_specs(cfg), seed=7))
    specs = ip_adapter_specs(cfg, 64)
    ck = {"image_proj": synthetic_state_dict(specs["image_proj"], seed=7), "ip_adapter": synthetic_state_dict(specs["ip_adapter"], seed=7)}
    g = torch.Generator().manual_seed(8)
    rn = lambda *s: torch.randn(*s, generator=g)
    m1, m2 = torch.zeros(128, 128), torch.zeros(128, 128)
    m1[:64, :64] = 1
    m2[64:, 32:] = 1
    emb = dict(prompt_embeds=rn(1, 77, cfg.cross_attention_dim).half(), pooled_prompt_embeds=rn(1, cfg.pooled_dim).half(),
               negative_prompt_embeds=rn(1, 77, cfg.cross_attention_dim).half(), negative_pooled_prompt_embeds=rn(1, cfg.pooled_dim).half())
    cond = dict(image_embeds=rn(1, 64), base_embed=rn(1, 64), y=rn(1, 64), caption="a photo", base_latents=rn(1, 4, 16, 16).half(), **emb,
                subject_data=[(m1, rn(64)), (m2, rn(64))], subject_noise=rn(1, 4, 16, 16).half(), **{"subject_" + k: v for k, v in emb.items()})
    pipe = InstructAny2PixPipeline(unet=base, ip_ckpt=ck, device=DEV, clip_embeddings_dim=64, conditioner=lambda inst, mm, use_cache=False: cond)
    assert pipe.pipe_inpainting.unet is pipe.pipe.unet and pipe.pipe_inpainting.scheduler is pipe.pipe.scheduler      # shared modules (:132-139)
    torch.manual_seed(3)
    non_refined, out, msg = pipe("add the dog", [], num_inference_steps=5, cfg=4.0, refinement=0.0, subject_strength=0.7)
    assert msg == "SUCCESS!" and torch.isfinite(out).all() and not torch.equal(non_refined, out)
    by_hand = subject_consistency(cond["subject_data"], non_refined, pipe.ip_adapter_xl_inpaint, 0.7, output_type="latent", noise=cond["subject_noise"], **emb)
    assert torch.equal(by_hand, out)
    # like the reference, the next request's inversion sees the IP scale the last pass left behind (0.8 here): restore the start state
    pipe.ip_adapter_xl.set_scale(1.0)
    torch.manual_seed(3)
    a, b, _ = pipe("add the dog", [], num_inference_steps=5, cfg=4.0, refinement=0.0, subject_strength=0.0)
    assert torch.equal(a, b) and torch.equal(a, non_refined)
