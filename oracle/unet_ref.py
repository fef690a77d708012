"""ORACLE (test infrastructure, not product code): CPU restatement of the conditional UNet forward.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this.

What this restates
------------------
The UNet the reference calls at instructany2pix/ddim/pnp_pipeline.py:253-260 and
instructany2pix/ddim/sdxl_pipeline.py:832-839 is `diffusers==0.26.3` `UNet2DConditionModel`
(requirements.txt:3) with the SDXL-base config. diffusers is an un-vendored third-party dependency
that is absent from /root/reference and not installable here, so its published algorithm is restated
below from SURVEY.md Appendix A (A.1 config, A.2 embeddings, A.3 ResnetBlock2D, A.4 Transformer2DModel /
BasicTransformerBlock / Attention / GEGLU, A.6 resampling and skip wiring), keeping diffusers' module
names so state dicts interchange by key. Parity is anchored on the reference's own call sites:

* every attention op goes through the reference's operator-plugin protocol
  `proc(attn, hidden_states, encoder_hidden_states=...)` (attention_processor.py:205-279, 310-412), so the
  reference's OWN processor classes can be installed into this module tree unchanged
  (tests/golden/gen_goldens.py does exactly that to produce the committed fixtures);
* ResnetBlock / SpatialTransformer / GEGLU / timestep-embedding maths are cross-checked against the
  in-tree ldm blocks the reference vendors (llm/model/vae/modules/{blocks,attention,util}.py) by the
  golden fixtures G6.

PARITY PIN STATUS: pinned against golden vectors generated in the build container from the reference's
importable files (tests/golden/*.npz, generator script committed beside them). NOT pinned against a
diffusers checkout (none available): full-UNet agreement with diffusers itself is therefore "parity
unpinned" for the wiring that only diffusers owns (Appendix A provenance note); the parameter inventory
reproduces the published SDXL UNet parameter count (2 567 463 684) exactly.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .attn_processors_ref import AttnProcessor2_0Ref


def sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`; same formula
    as reference llm/model/vae/modules/util.py:271-291 (cos first, then sin)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    e = t[:, None].float() * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(e), torch.sin(e)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_dim, dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_dim, dim)
        self.linear_2 = nn.Linear(dim, dim)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class ResnetBlock2D(nn.Module):
    """SURVEY A.3; in-tree analogue reference llm/model/vae/modules/blocks.py:122-142."""

    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, emb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(emb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class Attention(nn.Module):
    """The `attn` object handed to processors. Attribute set = what the reference processors touch
    (attention_processor.py:320-410): to_q/to_k/to_v/to_out, heads, spatial_norm, group_norm, norm_cross,
    norm_encoder_hidden_states, prepare_attention_mask, residual_connection, rescale_output_factor,
    plus head_to_batch_dim/batch_to_head_dim/get_attention_scores for the bmm twins (:58-64)."""

    def __init__(self, query_dim, cross_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.spatial_norm = None
        self.group_norm = None
        self.norm_cross = None
        self.norm_encoder_hidden_states = None
        self.residual_connection = False
        self.rescale_output_factor = 1.0
        self.processor = AttnProcessor2_0Ref()

    def prepare_attention_mask(self, mask, target_length, batch_size):
        assert mask is None, "the hot path never passes an attention mask"
        return None

    def head_to_batch_dim(self, t):
        b, n, c = t.shape
        return t.reshape(b, n, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, n, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, n, d = t.shape
        b = bh // self.heads
        return t.reshape(b, self.heads, n, d).permute(0, 2, 1, 3).reshape(b, n, d * self.heads)

    def get_attention_scores(self, q, k, mask=None):
        s = torch.baddbmm(torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device),
                          q, k.transpose(-1, -2), beta=0, alpha=self.scale)
        return s.softmax(dim=-1).to(q.dtype)

    def forward(self, hidden_states, encoder_hidden_states=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, **kw)


class GEGLU(nn.Module):
    """a, g = proj(x).chunk(2, -1); a * gelu(g), exact-erf GELU (analogue reference
    llm/model/vae/modules/attention.py:37-44)."""

    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, 4 * dim), nn.Dropout(0.0), nn.Linear(4 * dim, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    """SURVEY A.4 (analogue reference llm/model/vae/modules/attention.py:211-215)."""

    def __init__(self, dim, heads, dim_head, ctx_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-5)
        self.attn1 = Attention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim, eps=1e-5)
        self.attn2 = Attention(dim, ctx_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim, eps=1e-5)
        self.ff = FeedForward(dim)

    def forward(self, x, ctx):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), encoder_hidden_states=ctx) + x
        x = self.ff(self.norm3(x)) + x
        return x


class Transformer2DModel(nn.Module):
    """SURVEY A.4, use_linear_projection=True (analogue reference attention.py:250-261 with 1x1 conv ≡ Linear)."""

    def __init__(self, dim, heads, dim_head, depth, ctx_dim, groups):
        super().__init__()
        self.norm = nn.GroupNorm(groups, dim, eps=1e-6)
        self.proj_in = nn.Linear(dim, dim)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(dim, heads, dim_head, ctx_dim) for _ in range(depth)])
        self.proj_out = nn.Linear(dim, dim)

    def forward(self, x, ctx):
        b, c, h, w = x.shape
        r = x
        x = self.norm(x).permute(0, 2, 3, 1).reshape(b, h * w, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, ctx)
        x = self.proj_out(x)
        return x.reshape(b, h, w, c).permute(0, 3, 1, 2) + r


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)   # symmetric pad 1

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class _Block(nn.Module):
    def __init__(self):
        super().__init__()


class UNet2DConditionModelRef(nn.Module):
    """Module tree and child-registration order mirror diffusers (down_blocks, up_blocks, mid_block;
    attentions before resnets inside a block) so `attn_processors` enumerates in the order the
    IP-Adapter checkpoint indexes."""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        ch = list(cfg.block_out_channels)
        depth = list(cfg.transformer_layers_per_block)
        heads = list(cfg.attention_head_dim)
        n = len(ch)
        g, eps, temb, ctx = cfg.norm_num_groups, cfg.norm_eps, cfg.time_embed_dim, cfg.cross_attention_dim
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(cfg.time_proj_dim, temb)
        self.add_embedding = TimestepEmbedding(cfg.projection_class_embeddings_input_dim, temb)
        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
        cprev = ch[0]
        skip = [ch[0]]
        for i in range(n):
            blk = _Block()
            if depth[i] > 0:
                blk.attentions = nn.ModuleList([Transformer2DModel(ch[i], heads[i], cfg.head_dim, depth[i], ctx, g)
                                                for _ in range(cfg.layers_per_block)])
            blk.resnets = nn.ModuleList([ResnetBlock2D(cprev if j == 0 else ch[i], ch[i], temb, g, eps)
                                         for j in range(cfg.layers_per_block)])
            skip += [ch[i]] * cfg.layers_per_block
            if i != n - 1:
                blk.downsamplers = nn.ModuleList([Downsample2D(ch[i])])
                skip.append(ch[i])
            cprev = ch[i]
            self.down_blocks.append(blk)
        mid = _Block()
        mid.attentions = nn.ModuleList([Transformer2DModel(ch[-1], heads[-1], cfg.head_dim, cfg.mid_block_transformer_layers, ctx, g)])
        mid.resnets = nn.ModuleList([ResnetBlock2D(ch[-1], ch[-1], temb, g, eps) for _ in range(2)])
        self.mid_block = mid
        rch, rdepth, rheads = ch[::-1], depth[::-1], heads[::-1]
        cprev = ch[-1]
        for i in range(n):
            blk = _Block()
            cout = rch[i]
            if rdepth[i] > 0:
                blk.attentions = nn.ModuleList([Transformer2DModel(cout, rheads[i], cfg.head_dim, rdepth[i], ctx, g)
                                                for _ in range(cfg.layers_per_block + 1)])
            res = []
            for j in range(cfg.layers_per_block + 1):
                cs = skip.pop()
                res.append(ResnetBlock2D((cprev if j == 0 else cout) + cs, cout, temb, g, eps))
            blk.resnets = nn.ModuleList(res)
            if i != n - 1:
                blk.upsamplers = nn.ModuleList([Upsample2D(cout)])
            cprev = cout
            self.up_blocks.append(blk)
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    # ---- operator-plugin API the reference uses (ip_adapter.py:123,142,154,168) -------------------
    @property
    def attn_processors(self) -> "OrderedDict[str, nn.Module]":
        procs = OrderedDict()
        for name, m in self.named_modules():
            if isinstance(m, Attention):
                procs[f"{name}.processor"] = m.processor
        return procs

    def set_attn_processor(self, processor):
        for name, m in self.named_modules():
            if isinstance(m, Attention):
                m.processor = processor[f"{name}.processor"] if isinstance(processor, dict) else processor

    # ---- forward --------------------------------------------------------
    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None,
                added_cond_kwargs=None, return_dict=False):
        cfg = self.config
        b = sample.shape[0]
        t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(b)
        emb = self.time_embedding(sinusoid(t, cfg.time_proj_dim).to(sample.dtype))
        text_embeds = added_cond_kwargs["text_embeds"]
        time_ids = added_cond_kwargs["time_ids"]            # "image_embeds", if present, is ignored (text_time)
        tid = sinusoid(time_ids.flatten(), cfg.addition_time_embed_dim).reshape(b, -1)
        aug = self.add_embedding(torch.cat([text_embeds, tid.to(text_embeds.dtype)], dim=-1).to(emb.dtype))
        emb = emb + aug
        ctx = encoder_hidden_states

        x = self.conv_in(sample)
        skips = [x]
        for blk in self.down_blocks:
            for j, res in enumerate(blk.resnets):
                x = res(x, emb)
                if hasattr(blk, "attentions"):
                    x = blk.attentions[j](x, ctx)
                skips.append(x)
            if hasattr(blk, "downsamplers"):
                x = blk.downsamplers[0](x)
                skips.append(x)
        x = self.mid_block.resnets[0](x, emb)
        x = self.mid_block.attentions[0](x, ctx)
        x = self.mid_block.resnets[1](x, emb)
        for blk in self.up_blocks:
            for j, res in enumerate(blk.resnets):
                x = res(torch.cat([x, skips.pop()], dim=1), emb)
                if hasattr(blk, "attentions"):
                    x = blk.attentions[j](x, ctx)
            if hasattr(blk, "upsamplers"):
                x = blk.upsamplers[0](x)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        return (x,)
