"""The embedding prior (second half) on the MI355X: GPT-2 stack (`ia2p_clip_encode_embeds`), slot projections,
sampler update (`ia2p_prior_step`) and the whole `InstructAny2PixPrior.generate_diffusion` vs the CPU oracle (oracle/prior_ref.py:
pinned against transformers' GPT2Model and against fixture G14 = the reference's own method text) and vs G14 directly.

Tolerances: the reference computes this stage in fp32; the HIP transformer stacks compute in fp16 with fp32 accumulation.
GPT-2 hidden states rel-L2 <= 5e-3; prior outputs rel-L2 <= 2e-2 (guidance 10 amplifies the fp16 difference of two stack outputs
tenfold) and cosine >= 0.9995 (the pipeline only uses the direction: y / |y| * 20, reference pipeline.py:322). fp32 sampler update
alone: <= 1e-5 relative."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prior.npz")
CASES = (("live", dict(no_diffusion=True, num_inference_steps=25, guidance_scale=10, force_guidence_t0=True, do_classifier_free_guidance=True, score=6.5)),
         ("steps3", dict(no_diffusion=False, num_inference_steps=3, guidance_scale=4, do_classifier_free_guidance=True, score=6.5)),
         ("nocfg", dict(no_diffusion=True, num_inference_steps=25, do_classifier_free_guidance=False, score=6.8)))


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def cosine(a, b):
    a, b = a.float().cpu().flatten(), b.float().cpu().flatten()
    return float(a @ b / (a.norm() * b.norm()))


@pytest.mark.parametrize("B,T", [(2, 11), (1, 14), (3, 64)])
def test_gpt2_stack_vs_oracle(B, T):
    import oracle
    from instructany2pix_amd.config import tiny_gpt2
    from instructany2pix_amd.prior import HipGPT2Model
    from instructany2pix_amd.weights import gpt2_param_specs, synthetic_state_dict
    cfg = tiny_gpt2()
    sd = synthetic_state_dict(gpt2_param_specs(cfg), seed=43)
    hip = HipGPT2Model(cfg, DEV)
    hip.load_state_dict(sd)
    ref = oracle.build_gpt2(cfg, sd)
    x = torch.randn(B, T, cfg.n_embd, generator=torch.Generator().manual_seed(B * 100 + T)).half()
    out = hip(inputs_embeds=x.to(DEV), attention_mask=torch.ones(B, T))["last_hidden_state"]
    torch.cuda.synchronize()
    want = ref(x.float())["last_hidden_state"]
    assert rel_l2(out, want) <= 5e-3
    # causality: later tokens do not change earlier rows, bit for bit
    x2 = x.clone()
    x2[:, T // 2:] += 1.0
    again = hip(inputs_embeds=x2.to(DEV))["last_hidden_state"]
    assert torch.equal(again[:, :T // 2], out[:, :T // 2]) and not torch.equal(again[:, T // 2:], out[:, T // 2:])
    # same rows, same bits, whatever the batch they arrive in
    if B > 1:
        solo = hip(inputs_embeds=x[:1].to(DEV))["last_hidden_state"]
        assert torch.equal(solo, out[:1])


def test_gpt2_rejects_what_the_live_path_never_sends():
    from instructany2pix_amd.config import tiny_gpt2
    from instructany2pix_amd.prior import HipGPT2Model
    from instructany2pix_amd.weights import gpt2_param_specs, synthetic_state_dict
    cfg = tiny_gpt2()
    hip = HipGPT2Model(cfg, DEV)
    with pytest.raises(RuntimeError):                       # weights not finalized
        hip(inputs_embeds=torch.zeros(1, 4, cfg.n_embd))
    hip.load_state_dict(synthetic_state_dict(gpt2_param_specs(cfg), seed=43))
    m = torch.ones(2, 5)
    m[1, 4] = 0
    with pytest.raises(NotImplementedError):
        hip(inputs_embeds=torch.zeros(2, 5, cfg.n_embd), attention_mask=m)
    with pytest.raises(ValueError):
        hip(inputs_embeds=torch.zeros(1, 200, cfg.n_embd))
    with pytest.raises(ValueError):
        hip(inputs_embeds=torch.zeros(1, 4, cfg.n_embd + 8))
    with pytest.raises(RuntimeError):                       # token ids on a model created without a token table
        hip._core(torch.zeros(1, 4, dtype=torch.long), want_pooled=False, want_last_hidden=True)


@pytest.mark.parametrize("cfg_on,noise_on,t,n", [(True, True, 667, 3), (True, False, 1, 1), (False, True, 334, 3), (True, True, 1, 3)])
def test_prior_step_vs_formula(cfg_on, noise_on, t, n):
    """fp32 kernel vs the reference's three-stage arithmetic (get_eps -> guidance -> DDPMScheduler.step) in torch fp32"""
    from instructany2pix_amd.scheduler import DDPMScheduler, prior_update
    import oracle
    sch, ref = DDPMScheduler(), oracle.DDPMSchedulerRef()
    sch.set_timesteps(n); ref.set_timesteps(n)
    assert sch.timesteps.tolist() == ref.timesteps.tolist()
    g = torch.Generator().manual_seed(t)
    E = 1024
    s = torch.randn(1, 1, E, generator=g).to(torch.int64).float()
    oc, ou = torch.randn(1, 1, E, generator=g).half(), torch.randn(1, 1, E, generator=g).half()
    z = torch.randn(1, 1, E, generator=g)
    sa, sb, k0, k1, sigma = sch.posterior_coeffs(t)
    out = torch.empty(1, 1, E, device=DEV)
    prior_update(s.to(DEV), oc.to(DEV) if cfg_on else None, ou.to(DEV), z.to(DEV) if noise_on else None, 10.0 if cfg_on else 1.0, sa, sb, k0, k1, sigma, out)
    a = ref.alphas_cumprod[t]
    eps_u = (s - a ** 0.5 * ou.float()) / (1 - a) ** 0.5
    eps = eps_u
    if cfg_on:
        eps_c = (s - a ** 0.5 * oc.float()) / (1 - a) ** 0.5
        eps = eps_u + 10.0 * (eps_c - eps_u)
    if noise_on:          # the oracle's DDPM step, fed the same noise through an identically seeded generator
        z2 = torch.randn(eps.shape, generator=torch.Generator().manual_seed(5))
        out2 = torch.empty(1, 1, E, device=DEV)
        prior_update(s.to(DEV), oc.to(DEV) if cfg_on else None, ou.to(DEV), z2.to(DEV), 10.0 if cfg_on else 1.0, sa, sb, k0, k1, sigma, out2)
        want = ref.step(eps, t, s, generator=torch.Generator().manual_seed(5))[0]
        assert float((out2.cpu() - want).abs().max() / want.abs().max()) <= 1e-5
    prev = t - 1000 // n
    a_p = ref.alphas_cumprod[prev] if prev >= 0 else ref.one
    x0 = (s - (1 - a) ** 0.5 * eps) / a ** 0.5
    want = (a_p ** 0.5 * (1 - a / a_p)) / (1 - a) * x0 + (a / a_p) ** 0.5 * (1 - a_p) / (1 - a) * s
    if noise_on:
        want = want + torch.clamp((1 - a_p) / (1 - a) * (1 - a / a_p), min=1e-20) ** 0.5 * z
    assert float((out.cpu() - want).abs().max() / want.abs().max()) <= 1e-5
    if prev < 0:          # last step of any schedule: x_prev = x0 (k0 = 1, k1 = 0) and sigma = 1e-10
        assert abs(k0 - 1.0) < 1e-6 and k1 == 0.0 and abs(sigma - 1e-10) < 1e-12


def _prior_pair(seed=41):
    import oracle
    from stub_tokenizer import StubTokenizer
    from instructany2pix_amd.config import tiny_clip, tiny_gpt2
    from instructany2pix_amd.prior import InstructAny2PixPrior, prior_config
    from instructany2pix_amd.weights import prior_param_specs, synthetic_state_dict
    gcfg, ccfg = tiny_gpt2(), tiny_clip(0, "gelu")
    dims = (0, 1024, ccfg.hidden_size, 512, 0, 0, 0)
    sd = synthetic_state_dict(prior_param_specs(gcfg, ccfg, dims), seed=seed, dtype=torch.float32)
    tok = StubTokenizer(5, ccfg.vocab_size)
    kw = dict(prior_config)
    kw.update(sequence_input_embed_dim=list(dims), embed_dim=gcfg.n_embd, output_dim=gcfg.n_embd)
    hip = InstructAny2PixPrior(**kw, device=DEV, gpt_config=gcfg, clip_config=ccfg, tokenizer=tok).eval()
    hip.load_state_dict(sd)
    clip = oracle.build_clip(ccfg, {k[len("cond_stage_models.0.model."):]: v for k, v in sd.items() if k.startswith("cond_stage_models.0.model.")})

    def text_hidden(prompts):
        b = tok(prompts, max_length=77, padding=True, truncation=True)
        return [clip(b.input_ids)[1], b.attention_mask.float()]
    return hip, oracle.PriorRef(gcfg, sd, text_hidden)


@pytest.mark.parametrize("tag,kw", CASES)
def test_generate_diffusion_vs_reference_fixture_and_oracle(tag, kw):
    G = np.load(GOLD)
    hip, ref = _prior_pair()
    src = torch.from_numpy(G["src"])
    torch.manual_seed(1234)
    y, cond = hip.generate_diffusion(3, 0, src, device="cpu", image_bind_overwrite=None, dtype=torch.float32, **kw)
    torch.cuda.synchronize()
    state_after = torch.get_rng_state()
    torch.manual_seed(1234)
    yo, _ = ref.generate_diffusion(3, 0, src, **kw)
    assert torch.equal(state_after, torch.get_rng_state())            # same random numbers consumed, in the same order
    want = torch.from_numpy(G[tag + "_y"])
    assert tuple(y.shape) == tuple(want.shape) and y.dtype == torch.float32
    for w in (want, yo):
        assert rel_l2(y, w) <= 2e-2, rel_l2(y, w)
        assert cosine(y, w) >= 0.9995
    # the sequence the model saw at the first step (slot projections, sos/eos rows, modality row, order)
    key = "noisy_input" if kw["no_diffusion"] else "noisy_inputs"
    assert key in cond and "crossattn_clip" in cond and "noise_level" in cond


def test_sequence_assembly_vs_reference_fixture():
    """first-step `inputs_embeds` of the live call: [modality | sos imagebind eos | sos clip(2 tokens) eos | sos score eos] x (cond, uncond)"""
    G = np.load(GOLD)
    hip, _ = _prior_pair()
    seen = []
    real = hip.model.__call__
    hip.model = type("Spy", (), {"__call__": lambda self, inputs_embeds=None, attention_mask=None: (seen.append((inputs_embeds.clone(), attention_mask.clone())), real(inputs_embeds=inputs_embeds, attention_mask=attention_mask))[1]})()
    torch.manual_seed(1234)
    hip.generate_diffusion(3, 0, torch.from_numpy(G["src"]), **dict(CASES[0][1]))
    assert len(seen) == int(G["live_ncalls"])
    x, m = seen[0]
    want = torch.from_numpy(G["live_seq0"])
    assert tuple(x.shape) == tuple(want.shape) == (2, 11, 128) and bool((m == 1).all())
    assert rel_l2(x, want) <= 3e-3
    assert list(G["sequence_input_key"]) == hip.sequence_input_key


def test_pipeline_calls_the_prior_when_the_conditioner_has_no_y():
    """`InstructAny2PixPipeline.__call__` (reference pipeline.py:313-324): y from the attached prior, fused and renormalised"""
    from instructany2pix_amd.pipeline import fuse_instruction_embedding
    hip, ref = _prior_pair()
    g = torch.Generator().manual_seed(9)
    ie, be = torch.randn(1, 1024, generator=g), torch.randn(1, 1024, generator=g)
    # tiny prior: 128-d output; exercise the call + fusion arithmetic on matching widths
    ie_s, be_s = ie[:, :128].clone(), be[:, :128].clone()
    torch.manual_seed(5)
    y = hip.generate_diffusion(3, 0, ie / ie.norm() * 100, device="cpu", no_diffusion=True, num_inference_steps=25, image_bind_overwrite=None,
                               dtype=torch.float32, guidance_scale=10, force_guidence_t0=True, do_classifier_free_guidance=True, score=6.5)
    torch.manual_seed(5)
    yo = ref.generate_diffusion(3, 0, ie / ie.norm() * 100, no_diffusion=True, num_inference_steps=25, guidance_scale=10, force_guidence_t0=True,
                                do_classifier_free_guidance=True, score=6.5)
    la = fuse_instruction_embedding(be_s, ie_s, y[0].cpu(), [0.0, 0.4, 1.0], 20.0)
    lo = fuse_instruction_embedding(be_s, ie_s, yo[0], [0.0, 0.4, 1.0], 20.0)
    assert abs(float(la.norm()) - 20.0) < 1e-3 and rel_l2(la, lo) <= 1e-2


def test_full_size_prior_runs_and_matches_oracle_direction():
    """gpt2-medium + CLIP ViT-H text tower shapes (710.5 M params, seeded synthetic weights): one live call vs the fp32 oracle"""
    import oracle
    from stub_tokenizer import StubTokenizer
    from instructany2pix_amd.config import gpt2_medium, laion_clip_h_text
    from instructany2pix_amd.prior import InstructAny2PixPrior, prior_config
    from instructany2pix_amd.weights import prior_param_specs, synthetic_state_dict
    gcfg, ccfg = gpt2_medium(), laion_clip_h_text()
    sd = synthetic_state_dict(prior_param_specs(gcfg, ccfg), seed=47)
    tok = StubTokenizer(5, ccfg.vocab_size)
    hip = InstructAny2PixPrior(**prior_config, device=DEV, tokenizer=tok)
    hip.load_state_dict(sd)
    clip = oracle.build_clip(ccfg, {k[len("cond_stage_models.0.model."):]: v for k, v in sd.items() if k.startswith("cond_stage_models.0.model.")})

    def text_hidden(prompts):
        b = tok(prompts, max_length=77, padding=True, truncation=True)
        return [clip(b.input_ids)[1], b.attention_mask.float()]
    ref = oracle.PriorRef(gcfg, sd, text_hidden)
    emb = torch.randn(1, 1024, generator=torch.Generator().manual_seed(3))
    src = emb / emb.norm() * 100
    kw = dict(CASES[0][1])
    torch.manual_seed(11)
    y, _ = hip.generate_diffusion(3, 0, src, **kw)
    torch.manual_seed(11)
    yo, _ = ref.generate_diffusion(3, 0, src, **kw)
    assert tuple(y.shape) == (1, 1, 1024)
    assert cosine(y, yo) >= 0.999 and rel_l2(y, yo) <= 5e-2
