"""ORACLE (test infrastructure): CPU restatement of SDXL's text encoders and of `encode_prompt`.

  CLIPTextModelRef    <- transformers `CLIPTextModel` / `CLIPTextModelWithProjection` (`pipe.text_encoder`, `pipe.text_encoder_2`),
                         the models `encode_prompt` runs (reference instructany2pix/ddim/sdxl_pipeline.py:307-320). transformers is a
                         third-party dependency of the reference, not vendored; it IS installed in this image, so the restatement is
                         pinned against the real classes (tests/test_oracle_golden.py::test_clip_restatement_matches_transformers,
                         same random weights, fp32, <= 2e-5) — parity pinned against the dependency itself rather than a fixture.
  encode_prompt_ref   <- `encode_prompt`, vendored in the reference at sdxl_pipeline.py:202-395 (token ids in, embeddings out)
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Attn(nn.Module):
    def __init__(self, h, heads):
        super().__init__()
        self.heads = heads
        self.k_proj, self.v_proj, self.q_proj, self.out_proj = nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h), nn.Linear(h, h)

    def forward(self, x):
        b, t, h = x.shape
        d = h // self.heads
        q, k, v = (p(x).view(b, t, self.heads, d).transpose(1, 2) for p in (self.q_proj, self.k_proj, self.v_proj))
        s = (q * d ** -0.5) @ k.transpose(-1, -2)
        s = s + torch.full((t, t), float("-inf")).triu(1)             # causal mask
        return self.out_proj((s.softmax(-1) @ v).transpose(1, 2).reshape(b, t, h))


class _MLP(nn.Module):
    def __init__(self, h, i, act):
        super().__init__()
        self.fc1, self.fc2, self.act = nn.Linear(h, i), nn.Linear(i, h), act

    def forward(self, x):
        x = self.fc1(x)
        x = x * torch.sigmoid(1.702 * x) if self.act == "quick_gelu" else F.gelu(x)
        return self.fc2(x)


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        h = cfg.hidden_size
        self.self_attn = _Attn(h, cfg.num_attention_heads)
        self.layer_norm1 = nn.LayerNorm(h, eps=cfg.layer_norm_eps)
        self.mlp = _MLP(h, cfg.intermediate_size, cfg.hidden_act)
        self.layer_norm2 = nn.LayerNorm(h, eps=cfg.layer_norm_eps)

    def forward(self, x):
        x = x + self.self_attn(self.layer_norm1(x))
        return x + self.mlp(self.layer_norm2(x))


class _Emb(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.token_embedding = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.position_embedding = nn.Embedding(cfg.max_position_embeddings, cfg.hidden_size)


class _Enc(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])


class _TextModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embeddings, self.encoder = _Emb(cfg), _Enc(cfg)
        self.final_layer_norm = nn.LayerNorm(cfg.hidden_size, eps=cfg.layer_norm_eps)


class CLIPTextModelRef(nn.Module):
    """module tree = transformers' (state-dict keys match); forward returns (pooled, last_hidden_state, hidden_states)"""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.text_model = _TextModel(cfg)
        if cfg.projection_dim:
            self.text_projection = nn.Linear(cfg.hidden_size, cfg.projection_dim, bias=False)

    @torch.no_grad()
    def forward(self, input_ids):
        tm, cfg = self.text_model, self.config
        t = input_ids.shape[1]
        x = tm.embeddings.token_embedding(input_ids) + tm.embeddings.position_embedding(torch.arange(t))[None]
        hidden = [x]
        for layer in tm.encoder.layers:
            x = layer(x)
            hidden.append(x)
        last = tm.final_layer_norm(x)
        if cfg.eos_token_id == 2:        # legacy rule of the SDXL checkpoints: the EOS token has the largest id
            pos = input_ids.argmax(dim=-1)
        else:
            pos = (input_ids == cfg.eos_token_id).int().argmax(dim=-1)
        pooled = last[torch.arange(last.shape[0]), pos]
        if cfg.projection_dim:
            pooled = self.text_projection(pooled)
        return pooled, last, tuple(hidden)


def build_clip(cfg, state_dict):
    m = CLIPTextModelRef(cfg)
    m.load_state_dict({k: v.float() for k, v in state_dict.items()}, strict=True)
    return m.eval()


def encode_prompt_ref(enc1, enc2, ids1, ids2, neg_ids1=None, neg_ids2=None, num_images_per_prompt=1, zero_negative=False):
    """(prompt_embeds, negative_prompt_embeds, pooled, negative_pooled) from token ids (sdxl_pipeline.py:281-395)"""
    def run(a, b):
        h1, h2 = enc1(a)[2][-2], enc2(b)
        return torch.concat([h1, h2[2][-2]], dim=-1), h2[0]
    pe, pp = run(ids1, ids2)
    if zero_negative:
        ne, npl = torch.zeros_like(pe), torch.zeros_like(pp)
    else:
        ne, npl = run(neg_ids1, neg_ids2)
    rep = lambda e: e.repeat(1, num_images_per_prompt, 1).view(e.shape[0] * num_images_per_prompt, e.shape[1], -1)
    repp = lambda p: p.repeat(1, num_images_per_prompt).view(p.shape[0] * num_images_per_prompt, -1)
    return rep(pe), rep(ne), repp(pp), repp(npl)
