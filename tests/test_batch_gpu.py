"""Batched serving of independent edit requests with per-request knobs (instructany2pix_amd/batch.py; C ABI: ia2p_unet_forward_v, ia2p_ddim_step_v).

The reference serves one request per call (pipeline.py:303-386), so `num_inference_steps`, `cfg`, `scale`, `alpha` are per-call scalars there;
here N requests share every UNet evaluation. What must hold:
  * a batch element of a HETEROGENEOUS batch gets exactly the bits it gets in a UNIFORM batch of the same size (same kernels, same plans);
  * against the single-request path (`InstructAny2PixPipeline.denoise`, batch 1) only batch-dependent summation orders differ: fp16 tolerance;
  * the per-sample C entry points agree bit for bit with their scalar forms.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _pipe():
    from instructany2pix_amd.config import tiny
    from instructany2pix_amd.pipeline import InstructAny2PixPipeline
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict
    cfg = tiny()
    sd = synthetic_state_dict(unet_param_specs(cfg), seed=7)
    specs = ip_adapter_specs(cfg, 64)
    ck = {"image_proj": synthetic_state_dict(specs["image_proj"], seed=7), "ip_adapter": synthetic_state_dict(specs["ip_adapter"], seed=7)}
    unet = HipUNet2DConditionModel(cfg, DEV)
    unet.load_state_dict(sd)
    return cfg, InstructAny2PixPipeline(unet=unet, ip_ckpt=ck, device=DEV, clip_embeddings_dim=64)


def _requests(cfg, n, seed=3, hw=16):
    from instructany2pix_amd.batch import EditRequest
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    knobs = [dict(num_inference_steps=6, cfg=4.0, scale=1.0, alpha=0.7), dict(num_inference_steps=4, cfg=7.5, scale=0.5, alpha=0.6),
             dict(num_inference_steps=5, cfg=10.0, scale=0.0, alpha=0.8), dict(num_inference_steps=6, cfg=2.0, scale=1.3, alpha=0.5)]
    return [EditRequest(base_latents=r(1, 4, hw, hw).half(), latent_la=r(64), prompt_embeds=r(1, 77, cfg.cross_attention_dim).half(),
                        pooled_prompt_embeds=r(1, cfg.pooled_dim).half(), negative_prompt_embeds=r(1, 77, cfg.cross_attention_dim).half(),
                        negative_pooled_prompt_embeds=r(1, cfg.pooled_dim).half(), noise=r(1, 4, hw, hw).half(), **knobs[i % 4]) for i in range(n)]


def test_heterogeneous_batch_equals_uniform_batches_bit_for_bit():
    cfg, pipe = _pipe()
    reqs = _requests(cfg, 4)
    out, inv = pipe.denoise_batch(reqs, group=4)
    assert out.shape == (4, 4, 16, 16) and inv.shape == (4, 4, 16, 16) and torch.isfinite(out.float()).all()
    for i, r in enumerate(reqs):
        o, v = pipe.denoise_batch([r, r, r, r], group=4)          # all four slots carry request i
        assert torch.equal(o[0], o[3]) and torch.equal(v[1], v[2])       # slots of a uniform batch agree with each other ...
        assert torch.equal(inv[i], v[i]), i                              # ... and slot i of the mixed batch carries exactly those bits
        assert torch.equal(out[i], o[i]), i
    # different knobs really produce different results (the comparison above is not vacuous)
    assert not torch.equal(out[0], out[3])


def test_batch_matches_single_request_path_within_fp16_tolerance():
    cfg, pipe = _pipe()
    reqs = _requests(cfg, 3, seed=5)
    out, inv = pipe.denoise_batch(reqs, group=4)
    for i, r in enumerate(reqs):
        pipe.ip_adapter_xl.set_scale(1.0)              # the inversion sees the processors' current scale (reference: the previous call's)
        lat, linv = pipe.denoise(r.base_latents, r.latent_la, prompt_embeds=r.prompt_embeds, pooled_prompt_embeds=r.pooled_prompt_embeds,
                                 negative_prompt_embeds=r.negative_prompt_embeds, negative_pooled_prompt_embeds=r.negative_pooled_prompt_embeds,
                                 alpha=r.alpha, num_inference_steps=r.num_inference_steps, cfg=r.cfg, scale=r.scale, noise=r.noise)
        for a, b, what in ((inv[i], linv[0], "inverted"), (out[i], lat[0], "sampled")):
            a, b = a.float().flatten(), b.float().flatten()
            rel, cos = float((a - b).norm() / b.norm()), float(torch.dot(a, b) / (a.norm() * b.norm()))
            assert rel < 3e-2 and cos > 0.999, (i, what, rel, cos)       # SURVEY App. A trajectory bar


def test_groups_and_order():
    """5 requests in groups of 2 (2 + 2 + 1) come back in request order and equal the one-group result of the same batch sizes"""
    cfg, pipe = _pipe()
    reqs = _requests(cfg, 5, seed=9)
    out, _ = pipe.denoise_batch(reqs, group=2)
    assert out.shape[0] == 5
    o01, _ = pipe.denoise_batch(reqs[0:2], group=2)
    o4, _ = pipe.denoise_batch(reqs[4:5], group=2)
    assert torch.equal(out[0:2], o01) and torch.equal(out[4:5], o4)


def test_per_sample_entry_points_equal_their_scalar_forms():
    """ia2p_unet_forward_v with equal timesteps / scales == ia2p_unet_forward + set_scale; ia2p_ddim_step_v with equal rows == ia2p_ddim_step"""
    from instructany2pix_amd.scheduler import fused_update, fused_update_v
    cfg, pipe = _pipe()
    unet = pipe.pipe.unet
    g = torch.Generator().manual_seed(1)
    B = 4
    x = torch.randn(B, 4, 16, 16, generator=g).half().to(DEV)
    ctx = torch.randn(B, 81, cfg.cross_attention_dim, generator=g).half().to(DEV)
    added = dict(text_embeds=torch.randn(B, cfg.pooled_dim, generator=g).half().to(DEV), time_ids=torch.tensor([[128.0, 128, 0, 0, 128, 128]] * B).half().to(DEV))
    pipe.ip_adapter_xl.set_scale(0.75)
    ref = unet(x, 481, encoder_hidden_states=ctx, added_cond_kwargs=added)[0].clone()
    pipe.ip_adapter_xl.set_scale(1.0)
    got = unet(x, torch.full((B,), 481.0), encoder_hidden_states=ctx, added_cond_kwargs=added, ip_scales=torch.full((B,), 0.75))[0]
    assert torch.equal(ref, got)
    # mixed timesteps / scales: element b equals the uniform evaluation at its own values
    ts, sc = torch.tensor([981.0, 481.0, 21.0, 481.0]), torch.tensor([1.0, 0.75, 0.0, 0.3])
    mixed = unet(x, ts, encoder_hidden_states=ctx, added_cond_kwargs=added, ip_scales=sc)[0].clone()
    assert torch.equal(mixed[1], ref[1])
    for b in (0, 2, 3):
        uni = unet(x, torch.full((B,), float(ts[b])), encoder_hidden_states=ctx, added_cond_kwargs=added, ip_scales=torch.full((B,), float(sc[b])))[0]
        assert torch.equal(mixed[b], uni[b]), b
    with pytest.raises(ValueError):
        unet(x, torch.zeros(3), encoder_hidden_states=ctx, added_cond_kwargs=added)
    # sampler update
    eu, ec = torch.randn_like(x), torch.randn_like(x)
    o1, o2 = torch.empty_like(x), torch.empty_like(x)
    fused_update(x, eu, ec, 7.5, 0.93, -0.21, o1)
    fused_update_v(x, eu, ec, torch.tensor([[7.5, 0.93, -0.21]] * B).to(DEV), o2)
    assert torch.equal(o1, o2)
    coef = torch.tensor([[7.5, 0.93, -0.21], [1.0, 1.0, 0.0], [0.0, 0.5, 0.5], [10.0, 1.01, -0.4]])
    fused_update_v(x, eu, ec, coef.to(DEV), o2)
    for b in range(B):
        fused_update(x[b:b + 1].contiguous(), eu[b:b + 1].contiguous(), ec[b:b + 1].contiguous(), float(coef[b, 0]), float(coef[b, 1]), float(coef[b, 2]), o1[b:b + 1])
    assert torch.equal(o1, o2)
    assert torch.equal(o2[1], x[1])            # (g, 1, 0): a request that has finished its schedule keeps its latents bit for bit


def test_denoise_batch_sharded_over_two_ranks():
    """world_size 2 on one GPU (gloo): contiguous request shards per rank, groups of 2, all-gather in request order == the single-process result, bit for bit"""
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "tests", "dist_batch_two_ranks.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=root)
    assert r.returncode == 0 and "BATCH_DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
