"""ORACLE (test infrastructure): restatement of the reference's attention operator plugins.

Follows reference instructany2pix/diffusion/ip_adapter/attention_processor.py:
  AttnProcessor2_0Ref   <- AttnProcessor2_0.__call__      :205-279
  IPAttnProcessor2_0Ref <- IPAttnProcessor2_0.__call__    :310-412
Written as explicit softmax(QK^T/sqrt(d))V in the caller's dtype instead of F.scaled_dot_product_attention
so the arithmetic is spelled out; pinned against the reference classes themselves by golden fixtures
G1/G2 (tests/golden/attn_*.npz, tests/test_oracle_golden.py).
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _sdpa(q, k, v):
    # reference :259 / :371 / :387 — no mask, no dropout, scale 1/sqrt(head_dim)
    s = (q @ k.transpose(-2, -1)) * (q.shape[-1] ** -0.5)
    return s.softmax(dim=-1) @ v


def _split_heads(t, b, heads):
    return t.view(b, -1, heads, t.shape[-1] // heads).transpose(1, 2)     # reference :252-255


class AttnProcessor2_0Ref(nn.Module):
    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *a, **kw):
        assert hidden_states.ndim == 3 and attention_mask is None
        residual = hidden_states
        b = hidden_states.shape[0]
        q = attn.to_q(hidden_states)                                       # :239
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        k, v = attn.to_k(ctx), attn.to_v(ctx)                              # :246-247
        o = _sdpa(_split_heads(q, b, attn.heads), _split_heads(k, b, attn.heads), _split_heads(v, b, attn.heads))
        o = o.transpose(1, 2).reshape(b, -1, q.shape[-1]).to(q.dtype)      # :263-264
        o = attn.to_out[1](attn.to_out[0](o))                              # :267-269
        if attn.residual_connection:
            o = o + residual
        return o / attn.rescale_output_factor                              # :277


class IPAttnProcessor2_0Ref(nn.Module):
    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4):
        super().__init__()
        self.hidden_size, self.cross_attention_dim = hidden_size, cross_attention_dim
        self.scale, self.num_tokens = scale, num_tokens
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)   # :307
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)   # :308

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *a, **kw):
        assert hidden_states.ndim == 3 and attention_mask is None and encoder_hidden_states is not None
        residual = hidden_states
        b = hidden_states.shape[0]
        q = attn.to_q(hidden_states)                                       # :344
        end = encoder_hidden_states.shape[1] - self.num_tokens             # :350
        text, ip = encoder_hidden_states[:, :end], encoder_hidden_states[:, end:]
        qh = _split_heads(q, b, attn.heads)
        o = _sdpa(qh, _split_heads(attn.to_k(text), b, attn.heads), _split_heads(attn.to_v(text), b, attn.heads))
        o = o.transpose(1, 2).reshape(b, -1, q.shape[-1]).to(q.dtype)      # :375-376
        ipk = _split_heads(self.to_k_ip(ip), b, attn.heads)                # :379,:382
        ipv = _split_heads(self.to_v_ip(ip), b, attn.heads)                # :380,:383
        oi = _sdpa(qh, ipk, ipv)                                           # :387 — its OWN softmax over the ip keys
        # side effect kept for parity (:390-391): softmax binds to ip_k^T (over the token axis) before the matmul
        self.attn_map = qh @ ipk.transpose(-2, -1).softmax(dim=-1)
        oi = oi.transpose(1, 2).reshape(b, -1, q.shape[-1]).to(q.dtype)
        o = o + self.scale * oi                                            # :397
        o = attn.to_out[1](attn.to_out[0](o))                              # :400-402
        if attn.residual_connection:
            o = o + residual
        return o / attn.rescale_output_factor
