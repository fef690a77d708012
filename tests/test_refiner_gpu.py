"""Refiner pass on the MI355X: the second UNet config of the same engine (4 levels, plain outer
blocks, attending mid block, 5 micro-conditioning ids) and the Euler image-to-image loop behind
`StableDiffusionXLImg2ImgPipeline` (reference instructany2pix/pipeline.py:128-131, :358-361), HIP path through the C ABI
vs the CPU oracle on the same seeded inputs and fp16-representable weights.

Tolerances as in test_unet_gpu.py: one UNet forward rel-L2 <= 5e-3, max|d| <= 2e-2 max|ref|; loops rel-L2 <= 3e-2, cos >= 0.999."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def _traj_metrics(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))


@pytest.fixture(scope="module")
def tiny_refiner_models():
    import oracle
    from instructany2pix_amd.config import tiny_refiner
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, synthetic_state_dict
    cfg = tiny_refiner()
    sd = synthetic_state_dict(unet_param_specs(cfg), seed=11)
    hip = HipUNet2DConditionModel(cfg, DEV)
    hip.load_state_dict(sd)
    return cfg, sd, hip, oracle.build_unet(cfg, sd), oracle


def _inputs(cfg, B, h, w, L, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 4, h, w, generator=g).half()
    ctx = torch.randn(B, L, cfg.cross_attention_dim, generator=g).half()
    te = torch.randn(B, cfg.pooled_dim, generator=g).half()
    tid = torch.tensor([[h * 8.0, w * 8.0, 0, 0, 6.0]] * B).half()
    return x, ctx, te, tid


@pytest.mark.parametrize("B,h,w,L,t", [(2, 16, 16, 77, 481.0), (1, 32, 24, 77, 1.0), (3, 8, 8, 20, 961.0)])
def test_refiner_unet_forward_vs_oracle(tiny_refiner_models, B, h, w, L, t):
    cfg, sd, hip, ref_net, oracle = tiny_refiner_models
    x, ctx, te, tid = _inputs(cfg, B, h, w, L, seed=B * 10 + L)
    out = hip(x.to(DEV), t, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=dict(text_embeds=te.to(DEV), time_ids=tid.to(DEV)))[0]
    with torch.no_grad():
        ref = ref_net(x.float(), t, ctx.float(), added_cond_kwargs=dict(text_embeds=te.float(), time_ids=tid.float()))[0]
    assert rel_l2(out, ref) <= 5e-3, rel_l2(out, ref)
    assert float((out.float().cpu() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_refiner_unet_rejects_six_time_ids(tiny_refiner_models):
    cfg, sd, hip, ref_net, oracle = tiny_refiner_models
    x, ctx, te, tid = _inputs(cfg, 1, 8, 8, 77, seed=3)
    six = torch.tensor([[64.0, 64.0, 0, 0, 64.0, 64.0]]).half()
    with pytest.raises(ValueError):
        hip(x.to(DEV), 1.0, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=dict(text_embeds=te.to(DEV), time_ids=six.to(DEV)))
    with pytest.raises(ValueError):                       # h, w must be divisible by 2^(levels-1) = 8
        hip(x[:, :, :4].contiguous().to(DEV), 1.0, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=dict(text_embeds=te.to(DEV), time_ids=tid.to(DEV)))


@pytest.mark.parametrize("guidance,strength,N", [(5.0, 0.5, 20), (1.0, 0.3, 10), (7.5, 1.0, 6)])
def test_img2img_euler_loop_vs_oracle(tiny_refiner_models, guidance, strength, N):
    """clean latents -> add_noise at the first kept timestep -> Euler steps with CFG (the reference's refinement pass)"""
    from instructany2pix_amd.img2img import StableDiffusionXLImg2ImgPipeline
    cfg, sd, hip, ref_net, oracle = tiny_refiner_models
    B, h, w = 2, 16, 16
    g = torch.Generator().manual_seed(int(guidance * 10) + N)
    lat = torch.randn(B, 4, h, w, generator=g).half()
    noise = torch.randn(B, 4, h, w, generator=g).half()
    ctx, nctx = torch.randn(B, 77, cfg.cross_attention_dim, generator=g).half(), torch.randn(B, 77, cfg.cross_attention_dim, generator=g).half()
    pooled, npooled = torch.randn(B, cfg.pooled_dim, generator=g).half(), torch.randn(B, cfg.pooled_dim, generator=g).half()
    pipe = StableDiffusionXLImg2ImgPipeline(hip)
    seen = []
    out = pipe(prompt_embeds=ctx, negative_prompt_embeds=nctx, pooled_prompt_embeds=pooled, negative_pooled_prompt_embeds=npooled, latents=lat,
               noise=noise, strength=strength, num_inference_steps=N, guidance_scale=guidance, output_type="latent",
               callback=lambda i, t, x: seen.append(float(t))).images
    torch.cuda.synchronize()
    H = h * 8
    ids, neg_ids = oracle.get_add_time_ids_aesthetic((H, H), (0, 0), (H, H), 6.0, 2.5, (H, H), (0, 0), (H, H), cfg.addition_time_embed_dim, cfg.pooled_dim,
                                                     cfg.projection_class_embeddings_input_dim)
    sch = oracle.EulerDiscreteSchedulerRef()
    with torch.no_grad():
        ref = oracle.img2img_loop(ref_net, sch, lat.float(), noise.float(), ctx.float(), dict(text_embeds=pooled.float(), time_ids=ids.repeat(B, 1)), N, strength,
                                  guidance, nctx.float(), dict(text_embeds=npooled.float(), time_ids=neg_ids.repeat(B, 1)))
    kept = min(int(N * strength), N)
    assert len(seen) == kept and seen == [float(t) for t in sch.timesteps[N - kept:]]
    r, c = _traj_metrics(out, ref)
    assert r < 3e-2 and c > 0.999, (r, c)
    assert lat.abs().max() > 0 and torch.equal(lat, lat.clone())          # the caller's latents are not written


def test_refiner_full_size_forward_vs_oracle():
    """Full SDXL-refiner architecture (2.26 G parameters: 384/768/1536/1536, 4-layer transformers, 1280-d context) on a 32x32
    latent, B=1: what the CPU oracle finishes in seconds."""
    import os
    import oracle
    from instructany2pix_amd.config import sdxl_refiner
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, iter_synthetic
    cfg = sdxl_refiner()
    us = unet_param_specs(cfg)
    hip = HipUNet2DConditionModel(cfg, DEV)
    hip.load_state_dict(iter_synthetic(us, 5, DEV, torch.float16))
    x, ctx, te, tid = _inputs(cfg, 1, 32, 32, 77, seed=5)
    out = hip(x.to(DEV), 481.0, encoder_hidden_states=ctx.to(DEV), added_cond_kwargs=dict(text_embeds=te.to(DEV), time_ids=tid.to(DEV)))[0]
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ref_net = oracle.build_unet_fast(cfg, ((k, v.cpu()) for k, v in iter_synthetic(us, 5, DEV, torch.float16)))
    with torch.no_grad():
        ref = ref_net(x.float(), 481.0, ctx.float(), added_cond_kwargs=dict(text_embeds=te.float(), time_ids=tid.float()))[0]
    assert rel_l2(out, ref) < 5e-3, rel_l2(out, ref)
    assert float((out.float().cpu() - ref).abs().max()) < 2e-2 * float(ref.abs().max())


def test_refinement_image_to_image_with_vae():
    """pixels -> VAE encode -> noise to strength 0.5 -> 10 Euler CFG steps on the refiner-topology UNet -> VAE decode, all on the
    HIP path, vs the same chain on the CPU oracle (tiny configs)."""
    import oracle
    from instructany2pix_amd.config import tiny_refiner, tiny_vae
    from instructany2pix_amd.img2img import StableDiffusionXLImg2ImgPipeline
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.vae import HipAutoencoderKL
    from instructany2pix_amd.weights import unet_param_specs, vae_param_specs, synthetic_state_dict
    ucfg, vcfg = tiny_refiner(), tiny_vae()
    vsd = synthetic_state_dict(vae_param_specs(vcfg), seed=3)
    vae = HipAutoencoderKL(vcfg, DEV)
    vae.load_state_dict(vsd)
    rvae = oracle.build_vae(vcfg, vsd)
    sd = synthetic_state_dict(unet_param_specs(ucfg), seed=11)
    unet = HipUNet2DConditionModel(ucfg, DEV)
    unet.load_state_dict(sd)
    ref_net = oracle.build_unet(ucfg, sd)
    g = torch.Generator().manual_seed(31)
    img = torch.randn(1, 3, 64, 64, generator=g).half()            # 64x64 image -> 16x16 latent (tiny VAE: factor 4)
    ctx, nctx = torch.randn(1, 77, ucfg.cross_attention_dim, generator=g).half(), torch.randn(1, 77, ucfg.cross_attention_dim, generator=g).half()
    pooled, npooled = torch.randn(1, ucfg.pooled_dim, generator=g).half(), torch.randn(1, ucfg.pooled_dim, generator=g).half()
    noise = torch.randn(1, 4, 16, 16, generator=g).half()
    pipe = StableDiffusionXLImg2ImgPipeline(unet, vae_encode=lambda im: vae.encode_to_latents(im.to(DEV), torch.Generator().manual_seed(1)),
                                            vae_decode=vae.decode_from_latents)
    out = pipe(image=img, prompt_embeds=ctx, negative_prompt_embeds=nctx, pooled_prompt_embeds=pooled, negative_pooled_prompt_embeds=npooled,
               noise=noise, strength=0.5, num_inference_steps=20, guidance_scale=5.0, output_type="pt").images
    torch.cuda.synchronize()
    with torch.no_grad():
        mom = rvae.encode_moments(img.float())
        base = oracle.sample_latents(mom, torch.randn(mom[:, :4].shape, generator=torch.Generator().manual_seed(1)), vcfg.scaling_factor)
        H = 16 * 8
        ids, neg_ids = oracle.get_add_time_ids_aesthetic((H, H), (0, 0), (H, H), 6.0, 2.5, (H, H), (0, 0), (H, H), ucfg.addition_time_embed_dim,
                                                         ucfg.pooled_dim, ucfg.projection_class_embeddings_input_dim)
        rlat = oracle.img2img_loop(ref_net, oracle.EulerDiscreteSchedulerRef(), base, noise.float(), ctx.float(), dict(text_embeds=pooled.float(), time_ids=ids),
                                   20, 0.5, 5.0, nctx.float(), dict(text_embeds=npooled.float(), time_ids=neg_ids))
        rout = rvae.decode(rlat / vcfg.scaling_factor)
    r, c = _traj_metrics(out, rout)
    assert r < 4e-2 and c > 0.999, (r, c)


def test_pipeline_call_runs_refinement_through_piperf():
    """`InstructAny2PixPipeline.__call__(..., refinement=0.5)`: base sample -> `self.piperf` (reference pipeline.py:358-361);
    the refined output equals running the refiner pipeline by hand on the non-refined sample, and refinement=0 skips it."""
    from instructany2pix_amd.config import tiny, tiny_refiner
    from instructany2pix_amd.pipeline import InstructAny2PixPipeline
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    bcfg, rcfg = tiny(), tiny_refiner()
    base = HipUNet2DConditionModel(bcfg, DEV)
    base.load_state_dict(synthetic_state_dict(unet_param_specs(bcfg), seed=7))
    ref = HipUNet2DConditionModel(rcfg, DEV)
    ref.load_state_dict(synthetic_state_dict(unet_param_specs(rcfg), seed=11))
    specs = ip_adapter_specs(bcfg, 64)
    ck = {"image_proj": synthetic_state_dict(specs["image_proj"], seed=7), "ip_adapter": synthetic_state_dict(specs["ip_adapter"], seed=7)}
    g = torch.Generator().manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g)
    cond = dict(image_embeds=rn(1, 64), base_embed=rn(1, 64), y=rn(1, 64), caption="a photo", base_latents=rn(1, 4, 16, 16).half(),
                prompt_embeds=rn(1, 77, bcfg.cross_attention_dim).half(), pooled_prompt_embeds=rn(1, bcfg.pooled_dim).half(),
                negative_prompt_embeds=rn(1, 77, bcfg.cross_attention_dim).half(), negative_pooled_prompt_embeds=rn(1, bcfg.pooled_dim).half(),
                refiner_prompt_embeds=rn(1, 77, rcfg.cross_attention_dim).half(), refiner_pooled_prompt_embeds=rn(1, rcfg.pooled_dim).half(),
                refiner_negative_prompt_embeds=rn(1, 77, rcfg.cross_attention_dim).half(), refiner_negative_pooled_prompt_embeds=rn(1, rcfg.pooled_dim).half(),
                refiner_noise=rn(1, 4, 16, 16).half())
    pipe = InstructAny2PixPipeline(unet=base, ip_ckpt=ck, device=DEV, clip_embeddings_dim=64, conditioner=lambda inst, mm, use_cache=False: cond,
                                   refiner_unet=ref)
    assert pipe.piperf is not None and pipe.piperf.unet is ref
    torch.manual_seed(3)
    non_refined, refined, msg = pipe("make it blue", [], num_inference_steps=6, cfg=4.0, refinement=0.5)
    assert msg == "SUCCESS!" and torch.isfinite(refined).all() and not torch.equal(non_refined, refined)
    by_hand = pipe.piperf(latents=non_refined, strength=0.5, prompt_embeds=cond["refiner_prompt_embeds"], pooled_prompt_embeds=cond["refiner_pooled_prompt_embeds"],
                          negative_prompt_embeds=cond["refiner_negative_prompt_embeds"], negative_pooled_prompt_embeds=cond["refiner_negative_pooled_prompt_embeds"],
                          noise=cond["refiner_noise"], output_type="latent").images
    assert torch.equal(by_hand, refined)
    torch.manual_seed(3)
    a, b, _ = pipe("make it blue", [], num_inference_steps=6, cfg=4.0, refinement=0.0)
    assert torch.equal(a, b) and torch.equal(a, non_refined)


def test_pipeline_refiner_handoff_through_the_vae_like_the_reference():
    """With a VAE attached the base result reaches the refiner the reference's way (pipeline.py:358-361): decoded, quantised to an
    8-bit image, re-encoded by the refiner pipeline (posterior sample, global RNG), then noised and denoised. Equals the same steps
    by hand; the latent shortcut stays available (`refiner_handoff="latent"`) and differs from it; missing conditioner keys are
    reported up front."""
    from instructany2pix_amd.config import tiny, tiny_refiner, tiny_vae
    from instructany2pix_amd.pipeline import InstructAny2PixPipeline, to_8bit_image
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.vae import HipAutoencoderKL
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, vae_param_specs, synthetic_state_dict
    bcfg, rcfg, vcfg = tiny(), tiny_refiner(), tiny_vae()
    base = HipUNet2DConditionModel(bcfg, DEV)
    base.load_state_dict(synthetic_state_dict(unet_param_specs(bcfg), seed=7))
    ref = HipUNet2DConditionModel(rcfg, DEV)
    ref.load_state_dict(synthetic_state_dict(unet_param_specs(rcfg), seed=11))
    vae = HipAutoencoderKL(vcfg, DEV)
    vae.load_state_dict(synthetic_state_dict(vae_param_specs(vcfg), seed=7))
    specs = ip_adapter_specs(bcfg, 64)
    ck = {"image_proj": synthetic_state_dict(specs["image_proj"], seed=7), "ip_adapter": synthetic_state_dict(specs["ip_adapter"], seed=7)}
    g = torch.Generator().manual_seed(5)
    rn = lambda *s: torch.randn(*s, generator=g)
    cond = dict(image_embeds=rn(1, 64), base_embed=rn(1, 64), y=rn(1, 64), caption="a photo", base_latents=rn(1, 4, 16, 16).half(),
                prompt_embeds=rn(1, 77, bcfg.cross_attention_dim).half(), pooled_prompt_embeds=rn(1, bcfg.pooled_dim).half(),
                negative_prompt_embeds=rn(1, 77, bcfg.cross_attention_dim).half(), negative_pooled_prompt_embeds=rn(1, bcfg.pooled_dim).half(),
                refiner_prompt_embeds=rn(1, 77, rcfg.cross_attention_dim).half(), refiner_pooled_prompt_embeds=rn(1, rcfg.pooled_dim).half(),
                refiner_negative_prompt_embeds=rn(1, 77, rcfg.cross_attention_dim).half(), refiner_negative_pooled_prompt_embeds=rn(1, rcfg.pooled_dim).half(),
                refiner_noise=rn(1, 4, 16, 16).half())
    mk = lambda **kw: InstructAny2PixPipeline(unet=base, ip_ckpt=ck, device=DEV, clip_embeddings_dim=64, conditioner=lambda inst, mm, use_cache=False: cond,
                                              refiner_unet=ref, vae_encode=vae.encode_to_latents, vae_decode=vae.decode_from_latents, **kw)
    pipe = mk()
    assert pipe.refiner_handoff == "image"
    torch.manual_seed(3)
    non_refined, refined, _ = pipe("make it blue", [], num_inference_steps=6, cfg=4.0, refinement=0.5)
    # the same hand-over step by step (the base sampler drew one polar-mixing noise from the global RNG before the posterior sample)
    torch.manual_seed(3)
    again, _, _ = pipe("make it blue", [], num_inference_steps=6, cfg=4.0, refinement=0.0)
    assert torch.equal(again, non_refined)
    img8 = to_8bit_image(vae.decode_from_latents(non_refined))
    q = (img8.float() / 2 + 0.5) * 255
    assert float((q - q.round()).abs().max()) < 0.1 and float(img8.min()) >= -1 and float(img8.max()) <= 1        # an 8-bit image in [-1, 1] (held in fp16: ulp 5e-4 * 127.5)
    by_hand = pipe.piperf(image=img8, strength=0.5, prompt_embeds=cond["refiner_prompt_embeds"], pooled_prompt_embeds=cond["refiner_pooled_prompt_embeds"],
                          negative_prompt_embeds=cond["refiner_negative_prompt_embeds"], negative_pooled_prompt_embeds=cond["refiner_negative_pooled_prompt_embeds"],
                          noise=cond["refiner_noise"], output_type="latent").images
    assert torch.equal(by_hand, refined)
    short = mk(refiner_handoff="latent")
    torch.manual_seed(3)
    nr2, refined2, _ = short("make it blue", [], num_inference_steps=6, cfg=4.0, refinement=0.5)
    assert torch.equal(nr2, non_refined) and not torch.equal(refined2, refined)
    with pytest.raises(ValueError):
        mk(refiner_handoff="pil")
    bad = dict(cond)
    del bad["refiner_pooled_prompt_embeds"]
    pipe.conditioner = lambda inst, mm, use_cache=False: bad
    with pytest.raises(KeyError, match="refiner_pooled_prompt_embeds"):
        pipe("make it blue", [], num_inference_steps=2, cfg=4.0, refinement=0.5)
