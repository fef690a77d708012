"""ORACLE (test infrastructure): CPU restatement of the VAE either side of the denoise loop.

The reference reaches it through diffusers' `AutoencoderKL` (`pipe.vae`): image -> latent before inversion
(instructany2pix/ddim/pnp_pipeline.py:190-204, `prepare_latents` of the img2img base class; scaling 0.13025) and
latent -> image after sampling (instructany2pix/ddim/sdxl_pipeline.py:859-871). diffusers is absent, so the
architecture is restated with diffusers' key names; UNLIKE the UNet it has an in-tree twin to pin against: the
reference vendors the original ldm `Encoder` / `Decoder` / `AttnBlock` / `Downsample` it descends from
(instructany2pix/llm/model/vae/modules/blocks.py:369-460 Encoder, :463-569 Decoder, :151-203 AttnBlock, :61-80
Downsample with the asymmetric (0,1,0,1) pad), and golden fixture G9 (tests/golden/vae_ldm.npz) checks this
restatement against those classes after key renaming.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


class VResnet(nn.Module):
    def __init__(self, cin, cout, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (self.conv_shortcut(x) if self.conv_shortcut is not None else x) + h


class VAttention(nn.Module):
    """single head over all pixels, head_dim = channels (blocks.py:179-203)"""

    def __init__(self, c, groups, eps):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, c, eps=eps)
        self.to_q, self.to_k, self.to_v = nn.Linear(c, c), nn.Linear(c, c), nn.Linear(c, c)
        self.to_out = nn.ModuleList([nn.Linear(c, c)])

    def forward(self, x):
        b, c, h, w = x.shape
        t = self.group_norm(x).reshape(b, c, h * w).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        a = ((q @ k.transpose(1, 2)) * c ** -0.5).softmax(dim=-1) @ v
        return x + self.to_out[0](a).transpose(1, 2).reshape(b, c, h, w)


class _Mid(nn.Module):
    def __init__(self, c, groups, eps):
        super().__init__()
        self.attentions = nn.ModuleList([VAttention(c, groups, eps)])
        self.resnets = nn.ModuleList([VResnet(c, c, groups, eps), VResnet(c, c, groups, eps)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class _Blk(nn.Module):
    pass


class EncoderRef(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g, eps = list(cfg.block_out_channels), cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        cprev = ch[0]
        for i, c in enumerate(ch):
            b = _Blk()
            b.resnets = nn.ModuleList([VResnet(cprev if j == 0 else c, c, g, eps) for j in range(cfg.layers_per_block)])
            if i != len(ch) - 1:
                b.downsamplers = nn.ModuleList([_Blk()])
                b.downsamplers[0].conv = nn.Conv2d(c, c, 3, stride=2, padding=0)
            cprev = c
            self.down_blocks.append(b)
        self.mid_block = _Mid(ch[-1], g, eps)
        self.conv_norm_out = nn.GroupNorm(g, ch[-1], eps=eps)
        self.conv_out = nn.Conv2d(ch[-1], 2 * cfg.latent_channels, 3, padding=1)

    def forward(self, x):
        x = self.conv_in(x)
        for b in self.down_blocks:
            for r in b.resnets:
                x = r(x)
            if hasattr(b, "downsamplers"):
                x = b.downsamplers[0].conv(F.pad(x, (0, 1, 0, 1)))          # asymmetric pad, stride 2, no conv padding
        return self.conv_out(F.silu(self.conv_norm_out(self.mid_block(x))))


class DecoderRef(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, g, eps = list(cfg.block_out_channels), cfg.norm_num_groups, cfg.norm_eps
        self.conv_in = nn.Conv2d(cfg.latent_channels, ch[-1], 3, padding=1)
        self.mid_block = _Mid(ch[-1], g, eps)
        self.up_blocks = nn.ModuleList()
        cprev = ch[-1]
        for i, c in enumerate(ch[::-1]):
            b = _Blk()
            b.resnets = nn.ModuleList([VResnet(cprev if j == 0 else c, c, g, eps) for j in range(cfg.layers_per_block + 1)])
            if i != len(ch) - 1:
                b.upsamplers = nn.ModuleList([_Blk()])
                b.upsamplers[0].conv = nn.Conv2d(c, c, 3, padding=1)
            cprev = c
            self.up_blocks.append(b)
        self.conv_norm_out = nn.GroupNorm(g, ch[0], eps=eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            for r in b.resnets:
                x = r(x)
            if hasattr(b, "upsamplers"):
                x = b.upsamplers[0].conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLRef(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.encoder, self.decoder = EncoderRef(cfg), DecoderRef(cfg)
        z = cfg.latent_channels
        self.quant_conv = nn.Conv2d(2 * z, 2 * z, 1)
        self.post_quant_conv = nn.Conv2d(z, z, 1)

    def encode_moments(self, image):
        """[B, 2z, h, w]: mean | logvar (diffusers DiagonalGaussianDistribution parameters)"""
        return self.quant_conv(self.encoder(image))

    def decode(self, z):
        return self.decoder(self.post_quant_conv(z))


def sample_latents(moments, noise, scaling_factor):
    """latent_dist.sample() * scaling_factor (logvar clamped to [-30, 20] as diffusers does)"""
    mean, logvar = moments.chunk(2, dim=1)
    return (mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise) * scaling_factor


def build_vae(cfg, state_dict):
    m = AutoencoderKLRef(cfg)
    m.load_state_dict({k: v.float() for k, v in state_dict.items()}, strict=True)
    return m.eval()
