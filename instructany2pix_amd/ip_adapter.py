"""IP-Adapter wrapper for the HIP UNet, mirroring reference instructany2pix/diffusion/ip_adapter/ip_adapter.py:
  ImageProjModel            :28-67     1024-d fused instruction embedding -> 4 context tokens
  IPAdapter.set_ip_adapter  :120-148   installs AttnProcessor on every attn1, IPAttnProcessor on every attn2
  IPAdapter.load_ip_adapter :155-169   {"image_proj": ..., "ip_adapter": {"<idx>.to_k_ip.weight": ...}}
  IPAdapter.get_image_embeds:171-209   (clip_image_embeds path only; the CLIP vision tower is off-path)
  IPAdapter.set_scale       :211-214
  IPAdapterXL.generate      :289-356   cat([text, image tokens]) contexts -> SDXL sampling loop
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _ffi
from .attention_processor import AttnProcessor2_0 as AttnProcessor
from .attention_processor import IPAttnProcessor2_0 as IPAttnProcessor
from .weights import hidden_size_of


def get_generator(seed, device):
    """reference diffusion/ip_adapter/utils.py:83-93"""
    if seed is None:
        return None
    if isinstance(seed, list):
        return [torch.Generator(device).manual_seed(s) for s in seed]
    return torch.Generator(device).manual_seed(seed)


class ImageProjModel(nn.Module):
    """Parameters as in the reference (state-dict compatible); the Linear and the LayerNorm run in
    libia2p_hip.so (`ia2p_linear_small`, `ia2p_layernorm`), the per-token blends are a few KB of torch ops."""

    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4, num_crops=2):
        super().__init__()
        self.generator = None
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)
        self.raw_embed = nn.Parameter(torch.zeros(2, cross_attention_dim))
        self.num_crops = num_crops

    @torch.no_grad()
    def forward(self, image_embeds, mode, scales=(1.0, 1.0)):
        L = _ffi.lib()
        bs = image_embeds.shape[0]
        D, T = self.cross_attention_dim, self.clip_extra_context_tokens
        x = image_embeds.to(dtype=torch.float16).reshape(bs * self.num_crops, -1).contiguous()
        if bs * self.num_crops > 16:
            raise ValueError("ImageProjModel: at most 8 requests per call")
        w, b = self.proj.weight.to(torch.float16).contiguous(), self.proj.bias.to(torch.float16).contiguous()
        y = torch.empty(bs * self.num_crops, T * D, dtype=torch.float16, device=x.device)
        _ffi.check(L.ia2p_linear_small(_ffi.current_stream(), _ffi.ptr(x), _ffi.ptr(w), _ffi.ptr(b), _ffi.ptr(y),
                                       bs * self.num_crops, T * D, x.shape[1], 0, 0))
        t = y.float().reshape(bs, self.num_crops, T, D)
        g = t[:, 0:1]
        loc = g * (1 - scales[1]) + t[:, 1:] * scales[1]                     # :49
        g = g + self.raw_embed[0].float()[None, None]                        # :50
        loc = loc + self.raw_embed[1].float()[None, None]                    # :51
        if mode == "global":
            out = g
        elif mode == "local":
            out = loc
        else:
            assert mode == "both", f"Invalid Mode {mode}"
            out = torch.cat([g, loc], dim=1)
        out = out.reshape(-1, D).to(torch.float16).contiguous()
        res = torch.empty_like(out)
        ga, be = self.norm.weight.to(torch.float16).contiguous(), self.norm.bias.to(torch.float16).contiguous()
        _ffi.check(L.ia2p_layernorm(_ffi.current_stream(), _ffi.ptr(out), _ffi.ptr(res), _ffi.ptr(ga), _ffi.ptr(be),
                                    out.shape[0], D, float(self.norm.eps)))
        return res.reshape(bs, -1, D)


class IPAdapter:
    def __init__(self, sd_pipe, image_encoder_path=None, ip_ckpt=None, device="cuda:0", num_tokens=4, clip_embeddings_dim=1024):
        self.device = device
        self.image_encoder_path = image_encoder_path
        self.ip_ckpt = ip_ckpt
        self.num_tokens = num_tokens
        self.clip_embeddings_dim = clip_embeddings_dim
        self.pipe = sd_pipe.to(self.device)
        self.set_ip_adapter()
        self.image_proj_model = self.init_proj()
        self.load_ip_adapter()

    def init_proj(self):
        return ImageProjModel(cross_attention_dim=self.pipe.unet.config.cross_attention_dim,
                              clip_embeddings_dim=self.clip_embeddings_dim,
                              clip_extra_context_tokens=self.num_tokens).to(self.device, dtype=torch.float16)

    def set_ip_adapter(self):
        """Self-attention layers get the plain processor, every cross-attention layer an IP processor sized by
        its block's channel count (same rule as reference :124-141)."""
        unet = self.pipe.unet
        ctx_dim = unet.config.cross_attention_dim
        plugins = {}
        for name in unet.attn_processors:
            if name.endswith("attn1.processor"):
                plugins[name] = AttnProcessor()
                continue
            ip = IPAttnProcessor(hidden_size=hidden_size_of(unet.config, name), cross_attention_dim=ctx_dim,
                                 scale=1.0, num_tokens=self.num_tokens)
            plugins[name] = ip.to(self.device, dtype=torch.float16)
        unet.set_attn_processor(plugins)

    def enable(self):
        self.set_ip_adapter()
        self.load_ip_adapter()

    def disable(self):
        self.pipe.unet.set_attn_processor(AttnProcessor())

    @staticmethod
    def _read_checkpoint(ck):
        """-> {"image_proj": {...}, "ip_adapter": {"<idx>.to_k_ip.weight": ...}} from a dict, .safetensors or .bin"""
        if isinstance(ck, dict):
            return ck
        if str(ck).endswith(".safetensors"):
            from safetensors.torch import load_file
            flat = load_file(ck, device="cpu")
            out = {"image_proj": {}, "ip_adapter": {}}
            for key, val in flat.items():
                group, _, rest = key.partition(".")
                if group in out:
                    out[group][rest] = val
            return out
        return torch.load(ck, map_location="cpu")

    def load_ip_adapter(self):
        sd = self._read_checkpoint(self.ip_ckpt)
        self.image_proj_model.load_state_dict(sd["image_proj"])
        # checkpoint keys are indexed by position in unet.attn_processors (reference :168-169)
        torch.nn.ModuleList(self.pipe.unet.attn_processors.values()).load_state_dict(sd["ip_adapter"])

    @torch.inference_mode()
    def get_image_embeds(self, pil_image=None, clip_image_embeds=None, pil_image_local=None, clip_image_embeds_local=None,
                         mode="global", scale_g=1.0, scale_l=1.0):
        if pil_image is not None or pil_image_local is not None:
            raise NotImplementedError("the CLIP vision tower is outside the denoise hot path: pass clip_image_embeds")
        if clip_image_embeds is not None:
            clip_image_embeds = clip_image_embeds.to(self.device, dtype=torch.float16)
            if clip_image_embeds.ndim == 1:
                clip_image_embeds = clip_image_embeds[None]
        if clip_image_embeds_local is not None:
            clip_image_embeds_local = clip_image_embeds_local.to(self.device, dtype=torch.float16)
            if clip_image_embeds_local.ndim == 1:
                clip_image_embeds_local = clip_image_embeds_local[None]
        if clip_image_embeds is None:
            assert clip_image_embeds_local is not None
            clip_image_embeds = torch.zeros_like(clip_image_embeds_local)
        elif clip_image_embeds_local is None:
            clip_image_embeds_local = torch.zeros_like(clip_image_embeds)
        image_embeds = torch.stack([clip_image_embeds, clip_image_embeds_local], dim=1)       # :203  [N, 2, D]
        image_embeds_neg = torch.zeros_like(image_embeds)                                     # :204
        image_prompt_embeds = self.image_proj_model(image_embeds, mode=mode, scales=[scale_g, scale_l])
        uncond_image_prompt_embeds = self.image_proj_model(image_embeds_neg, mode=mode)
        return image_prompt_embeds, uncond_image_prompt_embeds

    def set_scale(self, scale):
        for attn_processor in self.pipe.unet.attn_processors.values():
            if isinstance(attn_processor, IPAttnProcessor):
                attn_processor.scale = scale


class IPAdapterXL(IPAdapter):
    """SDXL"""

    def generate(self, pil_image=None, prompt=None, negative_prompt=None, scale=1.0, scale_g=1.0, scale_l=0.5, num_samples=1,
                 seed=None, num_inference_steps=30, pil_image_local=None, clip_image_embeds_local=None, clip_image_embeds=None,
                 mode="global", prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, **kwargs):
        self.set_scale(scale)
        image_prompt_embeds, uncond_image_prompt_embeds = self.get_image_embeds(
            pil_image, clip_image_embeds, pil_image_local, clip_image_embeds_local, mode=mode, scale_g=scale_g, scale_l=scale_l)
        bs_embed, seq_len, _ = image_prompt_embeds.shape
        image_prompt_embeds = image_prompt_embeds.repeat(1, num_samples, 1).view(bs_embed * num_samples, seq_len, -1)
        uncond_image_prompt_embeds = uncond_image_prompt_embeds.repeat(1, num_samples, 1).view(bs_embed * num_samples, seq_len, -1)
        if prompt_embeds is None:                                                              # :329-340 (CLIP encoders: off-path)
            num_prompts = bs_embed
            if prompt is None:
                prompt = "best quality, high quality"
            if negative_prompt is None:
                negative_prompt = "monochrome, lowres, bad anatomy, worst quality, low quality"
            if not isinstance(prompt, List):
                prompt = [prompt] * num_prompts
            if not isinstance(negative_prompt, List):
                negative_prompt = [negative_prompt] * num_prompts
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = self.pipe.encode_prompt(
                prompt, num_images_per_prompt=num_samples, do_classifier_free_guidance=True, negative_prompt=negative_prompt)
        dev = image_prompt_embeds.device
        prompt_embeds = torch.cat([prompt_embeds.to(dev, torch.float16), image_prompt_embeds], dim=1)             # :341
        negative_prompt_embeds = torch.cat([negative_prompt_embeds.to(dev, torch.float16), uncond_image_prompt_embeds], dim=1)   # :342
        self.generator = get_generator(seed, "cpu")
        return self.pipe(prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                         pooled_prompt_embeds=pooled_prompt_embeds, negative_pooled_prompt_embeds=negative_pooled_prompt_embeds,
                         num_inference_steps=num_inference_steps, generator=self.generator, **kwargs).images
