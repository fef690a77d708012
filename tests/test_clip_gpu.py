"""SDXL text encoders and `encode_prompt` on the MI355X: HIP path through the C ABI (`ia2p_clip_*`) vs the
CPU oracle (oracle/clip_ref.py, itself pinned against the transformers classes in tests/test_oracle_golden.py).

Tolerances (fp16 activations vs fp32 oracle): hidden states and pooled outputs rel-L2 <= 5e-3."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def _build(cfg, seed):
    import oracle
    from instructany2pix_amd.clip import HipCLIPTextModel
    from instructany2pix_amd.weights import clip_param_specs, synthetic_state_dict
    sd = synthetic_state_dict(clip_param_specs(cfg), seed=seed)
    hip = HipCLIPTextModel(cfg, DEV)
    hip.load_state_dict(sd)
    return hip, oracle.build_clip(cfg, sd)


def _ids(cfg, B, T, seed, eos_at):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, cfg.vocab_size - 1, (B, T), generator=g)
    ids[:, 0] = 0
    for b, n in enumerate(eos_at):
        ids[b, n] = cfg.vocab_size - 1 if cfg.eos_token_id == 2 else cfg.eos_token_id
        ids[b, n + 1:] = cfg.vocab_size - 1 if cfg.eos_token_id == 2 else 1
    return ids


@pytest.mark.parametrize("proj,act,eos,B,T", [(0, "quick_gelu", 2, 2, 77), (64, "gelu", 2, 3, 77), (64, "gelu", 999, 1, 77), (0, "gelu", 2, 17, 20)])
def test_clip_text_model_vs_oracle(proj, act, eos, B, T):
    from instructany2pix_amd.config import tiny_clip
    cfg = tiny_clip(proj, act)
    cfg.eos_token_id = eos
    hip, ref = _build(cfg, seed=21)
    ids = _ids(cfg, B, T, 5, [(7 * b + 3) % (T - 1) for b in range(B)])
    out = hip(ids, output_hidden_states=True, want_last_hidden=True)
    torch.cuda.synchronize()
    pooled, last, hidden = ref(ids)
    assert rel_l2(out.hidden_states[-2], hidden[-2]) <= 5e-3
    assert rel_l2(out.last_hidden_state, last) <= 5e-3
    assert rel_l2(out.text_embeds if proj else out.pooler_output, pooled) <= 5e-3
    assert len(out.hidden_states) == cfg.num_hidden_layers + 1
    with pytest.raises(IndexError):
        out.hidden_states[0]
    # only hidden_states[-2]: the last layer is skipped, same bits
    only = hip(ids, output_hidden_states=True, want_pooled=False)
    assert only.last_hidden_state is None and torch.equal(only.hidden_states[-2], out.hidden_states[-2])
    # causality: changing tokens after position p leaves hidden states up to p untouched, bit for bit
    ids2 = ids.clone()
    ids2[:, T // 2:] = (ids2[:, T // 2:] + 7) % (cfg.vocab_size - 1)
    again = hip(ids2, output_hidden_states=True, want_pooled=False).hidden_states[-2]
    assert torch.equal(again[:, :T // 2], out.hidden_states[-2][:, :T // 2]) and not torch.equal(again[:, T // 2:], out.hidden_states[-2][:, T // 2:])


def test_clip_validation():
    from instructany2pix_amd.config import tiny_clip
    cfg = tiny_clip()
    hip, _ = _build(cfg, seed=2)
    with pytest.raises(ValueError):
        hip(torch.zeros(1, 78, dtype=torch.long), output_hidden_states=True)          # longer than max_position_embeddings
    with pytest.raises(ValueError):
        hip(torch.zeros(77, dtype=torch.long))
    with pytest.raises(KeyError):
        hip.load_state_dict({"text_model.nope.weight": torch.zeros(4)})
    out = hip(torch.full((1, 77), 5000, dtype=torch.long), output_hidden_states=True)   # out-of-vocabulary ids are clamped, not read out of bounds
    assert torch.isfinite(out.hidden_states[-2]).all()


def test_encode_prompt_vs_oracle():
    """`encode_prompt` with two HIP encoders vs the oracle composition: concat of both penultimate states, pooled of the second,
    zeros for an absent negative prompt, per-image repeat, and the reference's type / batch-size errors."""
    import oracle
    from instructany2pix_amd.clip import SDXLTextEncoders
    from instructany2pix_amd.config import tiny_clip
    c1, c2 = tiny_clip(0, "quick_gelu"), tiny_clip(64, "gelu")
    h1, r1 = _build(c1, seed=31)
    h2, r2 = _build(c2, seed=32)

    class Tok:                                   # stand-in tokenizer: deterministic ids per prompt string
        model_max_length = 77

        def __init__(self, salt):
            self.salt = salt

        def __call__(self, text, padding=None, max_length=77, truncation=True, return_tensors="pt"):
            texts = [text] if isinstance(text, str) else text
            rows = []
            for t in texts:
                n = min(len(t.split()) + 2, max_length)
                g = torch.Generator().manual_seed(sum(map(ord, t)) + self.salt)
                row = torch.full((max_length,), c1.vocab_size - 1, dtype=torch.long)
                row[0] = 0
                row[1:n - 1] = torch.randint(3, c1.vocab_size - 1, (n - 2,), generator=g)
                rows.append(row)
            return type("Enc", (), {"input_ids": torch.stack(rows)})()

    t1, t2 = Tok(1), Tok(2)
    enc = SDXLTextEncoders(t1, t2, h1, h2)
    prompts, negs = ["a photo of a cat", "two dogs on the beach at sunset"], ["blurry", "low quality, bad anatomy"]
    pe, ne, pp, npl = enc.encode_prompt(prompts, num_images_per_prompt=2, negative_prompt=negs)
    rpe, rne, rpp, rnpl = oracle.encode_prompt_ref(r1, r2, t1(prompts).input_ids, t2(prompts).input_ids, t1(negs).input_ids, t2(negs).input_ids, 2)
    assert pe.shape == (4, 77, c1.hidden_size + c2.hidden_size) and pp.shape == (4, 64)
    for a, b in ((pe, rpe), (ne, rne), (pp, rpp), (npl, rnpl)):
        assert rel_l2(a, b) <= 5e-3
    pe0, ne0, pp0, npl0 = enc.encode_prompt("a photo of a cat")              # no negative prompt: zeros (force_zeros_for_empty_prompt)
    assert float(ne0.abs().max()) == 0.0 and float(npl0.abs().max()) == 0.0 and torch.equal(pe0[0], pe[0])
    _, ne1, _, _ = SDXLTextEncoders(t1, t2, h1, h2, force_zeros_for_empty_prompt=False).encode_prompt("a photo of a cat")
    assert float(ne1.abs().max()) > 0                                          # then the empty string is encoded
    with pytest.raises(TypeError):
        enc.encode_prompt("a cat", negative_prompt=["x"])
    with pytest.raises(ValueError):
        enc.encode_prompt(["a", "b"], negative_prompt=["x"])
    pe2, ne2, _, _ = enc.encode_prompt(prompt_embeds=pe0, pooled_prompt_embeds=pp0, negative_prompt_embeds=ne0, negative_pooled_prompt_embeds=npl0)
    assert torch.equal(pe2, pe0)


def test_sdxl_text_encoders_full_size_vs_oracle():
    """Both SDXL text towers at full size (CLIP-L: 12 x 768; OpenCLIP-bigG: 32 x 1280 + projection), 2 prompts of 77 tokens."""
    import os
    from instructany2pix_amd.config import sdxl_text_encoder, sdxl_text_encoder_2
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    for cfg, seed in ((sdxl_text_encoder(), 41), (sdxl_text_encoder_2(), 42)):
        hip, ref = _build(cfg, seed)
        ids = _ids(cfg, 2, 77, seed, [9, 30])
        out = hip(ids, output_hidden_states=True)
        pooled, last, hidden = ref(ids)
        assert rel_l2(out.hidden_states[-2], hidden[-2]) <= 5e-3, rel_l2(out.hidden_states[-2], hidden[-2])
        assert rel_l2(out[0] if cfg.projection_dim else out.pooler_output, pooled) <= 5e-3
