"""ORACLE (test infrastructure): CPU restatement of the DDIM loop, scheduler and conditioning helpers.

Reference-owned code restated here (pinned by golden fixtures G3/G4/G7 and, for the loops, G12/G13 = the outputs of the
reference's own `inverse` / `generate` / `__call__` method text run over this oracle's UNet; tests/golden/gen_goldens.py):
  backward_ddim         <- _backward_ddim                 instructany2pix/ddim/pnp_pipeline.py:73-85
  invert_loop           <- SDXLDDIMPipeline.inverse loop  instructany2pix/ddim/pnp_pipeline.py:249-278
  get_add_time_ids      <- _get_add_time_ids              instructany2pix/ddim/pnp_pipeline.py:23-71
  polar_interpolate     <- polar_intrtpolate              instructany2pix/pipeline.py:295-300
  ImageProjModelRef     <- ImageProjModel                 instructany2pix/diffusion/ip_adapter/ip_adapter.py:28-67
  sample_loop / cfg     <- vendored SDXL loop text        instructany2pix/ddim/sdxl_pipeline.py:823-857
  inpaint_loop          <- diffusers 0.26.3 StableDiffusionXLInpaintPipeline.__call__ (4-channel UNet branch), the class behind
                           `pipe_inpainting` (instructany2pix/pipeline.py:132-139; driven from gdino/lib.py:89-102): PARITY UNPINNED
                           (un-vendored dependency, no reference test or vector for it)
Third-party algorithm restated (diffusers==0.26.3 DDIMScheduler, absent from /root/reference; config per
SURVEY.md Appendix A.8). Its schedule tables are pinned against the in-tree ldm schedule utilities
(llm/model/vae/modules/util.py:141-194) by fixture G5.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class DDIMSchedulerRef:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.num_train_timesteps = num_train_timesteps
        # scaled_linear, float32 as diffusers computes it
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]          # set_alpha_to_one = False
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n                       # "leading" spacing
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, x, t=None):
        return x

    def step(self, eps, t, x):
        """eta = 0, epsilon prediction, no clipping/thresholding."""
        t = int(t)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps


def add_noise(sched, x0, noise, t):
    """DDIMScheduler.add_noise: sqrt(abar_t) x0 + sqrt(1 - abar_t) noise"""
    a = sched.alphas_cumprod[int(t)]
    return a ** 0.5 * x0 + (1 - a) ** 0.5 * noise


def backward_ddim(x_tm1, alpha_t, alpha_tm1, eps_xt):
    a, b = alpha_t, alpha_tm1
    return a ** 0.5 * ((1 / b ** 0.5) * x_tm1 + ((1 / a - 1) ** 0.5 - (1 / b - 1) ** 0.5) * eps_xt)


def cfg_combine(eps_uncond, eps_text, g):
    return eps_uncond + g * (eps_text - eps_uncond)                 # sdxl_pipeline.py:842-844


def get_add_time_ids(original_size, crops_coords_top_left, target_size, addition_time_embed_dim,
                     projection_dim, expected_add_embed_dim, dtype=torch.float32):
    ids = list(original_size + crops_coords_top_left + target_size)
    passed = addition_time_embed_dim * len(ids) + projection_dim
    if expected_add_embed_dim != passed:
        raise ValueError(f"Model expects an added time embedding vector of length {expected_add_embed_dim}, "
                         f"but a vector of {passed} was created.")
    return torch.tensor([ids], dtype=dtype)


def polar_interpolate(x, y, alpha):
    n0, n1 = x.norm(), y.norm()
    ll = x * alpha + y * (1 - alpha)
    return ll / ll.norm() * (n0 * alpha + n1 * (1 - alpha))


class ImageProjModelRef(nn.Module):
    def __init__(self, cross_attention_dim=1024, clip_embeddings_dim=1024, clip_extra_context_tokens=4, num_crops=2):
        super().__init__()
        self.cross_attention_dim = cross_attention_dim
        self.clip_extra_context_tokens = clip_extra_context_tokens
        self.proj = nn.Linear(clip_embeddings_dim, clip_extra_context_tokens * cross_attention_dim)
        self.norm = nn.LayerNorm(cross_attention_dim)
        self.raw_embed = nn.Parameter(torch.zeros(2, cross_attention_dim))
        self.num_crops = num_crops

    def forward(self, image_embeds, mode, scales=(1.0, 1.0)):
        bs = image_embeds.shape[0]
        t = self.proj(image_embeds).reshape(bs, self.num_crops, self.clip_extra_context_tokens, self.cross_attention_dim)
        g = t[:, 0:1]
        loc = g * (1 - scales[1]) + t[:, 1:] * scales[1]             # :49
        g = g + self.raw_embed[0][None, None]                        # :50
        loc = loc + self.raw_embed[1][None, None]                    # :51
        if mode == "global":
            out = g
        elif mode == "local":
            out = loc
        else:
            assert mode == "both", f"Invalid Mode {mode}"
            out = torch.cat([g, loc], dim=1)
        out = out.reshape(bs, -1, self.cross_attention_dim)
        return self.norm(out)


def _eps(unet, x, t, ctx, added):
    return unet(x, t, encoder_hidden_states=ctx, cross_attention_kwargs=None,
                added_cond_kwargs=dict(added), return_dict=False)[0]


@torch.no_grad()
def invert_loop(unet, sched: DDIMSchedulerRef, latents, ctx, added, num_inference_steps, trace=None):
    """x0 -> xT: ascending t, no CFG; alpha_prev is final_alpha_cumprod on the first iteration."""
    sched.set_timesteps(num_inference_steps)
    prev_t = None
    for t in reversed(sched.timesteps):
        eps = _eps(unet, latents, t, ctx, added)
        a_t = sched.alphas_cumprod[t]
        a_prev = sched.alphas_cumprod[prev_t] if prev_t is not None else sched.final_alpha_cumprod
        prev_t = t
        latents = backward_ddim(latents, a_t, a_prev, eps)
        if trace is not None:
            trace.append(latents.clone())
    return latents


@torch.no_grad()
def sample_loop(unet, sched: DDIMSchedulerRef, latents, ctx, added, num_inference_steps, guidance_scale=1.0,
                neg_ctx=None, neg_added=None, trace=None):
    """xT -> x0 with optional CFG (cat([uncond, cond]) along batch, sdxl_pipeline.py:800-803,826)."""
    sched.set_timesteps(num_inference_steps)
    do_cfg = neg_ctx is not None
    if do_cfg:
        ctx2 = torch.cat([neg_ctx, ctx], dim=0)
        added2 = {k: torch.cat([neg_added[k], added[k]], dim=0) for k in ("text_embeds", "time_ids")}
    latents = latents * sched.init_noise_sigma
    for t in sched.timesteps:
        if do_cfg:
            e = _eps(unet, torch.cat([latents] * 2), t, ctx2, added2)
            eu, ec = e.chunk(2)
            eps = cfg_combine(eu, ec, guidance_scale)
        else:
            eps = _eps(unet, latents, t, ctx, added)
        latents = sched.step(eps, t, latents)
        if trace is not None:
            trace.append(latents.clone())
    return latents


@torch.no_grad()
def inpaint_loop(unet, sched: DDIMSchedulerRef, image_latents, noise, mask, ctx, added, num_inference_steps, strength, guidance_scale=7.5,
                 neg_ctx=None, neg_added=None):
    """mask: [B,1,H,W] in [0,1] at any resolution; binarised at 0.5 and resized to the latent grid with nearest interpolation"""
    sched.set_timesteps(num_inference_steps)
    init = min(int(num_inference_steps * strength), num_inference_steps)
    ts = sched.timesteps[max(num_inference_steps - init, 0):]
    m = torch.nn.functional.interpolate((mask.float() >= 0.5).float(), size=image_latents.shape[-2:])
    do_cfg = neg_ctx is not None and guidance_scale > 1.0
    if do_cfg:
        ctx2 = torch.cat([neg_ctx, ctx], dim=0)
        added2 = {k: torch.cat([neg_added[k], added[k]], dim=0) for k in ("text_embeds", "time_ids")}
    x = noise * sched.init_noise_sigma if strength == 1.0 else add_noise(sched, image_latents, noise, ts[0])
    for i, t in enumerate(ts):
        if do_cfg:
            eu, ec = _eps(unet, torch.cat([x] * 2), t, ctx2, added2).chunk(2)
            eps = cfg_combine(eu, ec, guidance_scale)
        else:
            eps = _eps(unet, x, t, ctx, added)
        x = sched.step(eps, t, x)
        keep = add_noise(sched, image_latents, noise, ts[i + 1]) if i < len(ts) - 1 else image_latents
        x = (1 - m) * keep + m * x
    return x
