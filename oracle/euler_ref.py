"""ORACLE (test infrastructure): CPU restatement of the refiner pass.

  get_add_time_ids_aesthetic <- `_get_add_time_ids`, requires_aesthetics_score branch, in-tree at
                                instructany2pix/ddim/pnp_pipeline.py:23-71; pinned by golden fixture G10
                                (tests/golden/misc_refiner.npz, generated from that function itself)
  EulerDiscreteSchedulerRef  <- diffusers==0.26.3 `EulerDiscreteScheduler` with the scheduler_config.json of
                                stabilityai/stable-diffusion-xl-refiner-1.0 (the checkpoint the reference loads at
                                instructany2pix/pipeline.py:128-131). diffusers is absent from /root/reference and the
                                reference holds no test or vector for it: PARITY UNPINNED for the sampler formulae
                                (published algorithm: Karras et al. 2022, Alg. 1 with s_churn = 0, epsilon prediction);
                                its beta / alphas_cumprod table is the one fixture G5 pins.
  img2img_loop               <- diffusers 0.26.3 `StableDiffusionXLImg2ImgPipeline.__call__` (get_timesteps,
                                prepare_latents' add_noise, CFG loop), the class behind `self.piperf(...)`
                                (pipeline.py:358-361)
"""
from __future__ import annotations

import numpy as np
import torch


def get_add_time_ids_aesthetic(original_size, crops_coords_top_left, target_size, aesthetic_score, negative_aesthetic_score,
                               negative_original_size, negative_crops_coords_top_left, negative_target_size,
                               addition_time_embed_dim, projection_dim, expected_add_embed_dim, requires_aesthetics_score=True,
                               dtype=torch.float32):
    if requires_aesthetics_score:
        ids = list(original_size + crops_coords_top_left + (aesthetic_score,))
        neg = list(negative_original_size + negative_crops_coords_top_left + (negative_aesthetic_score,))
    else:
        ids = list(original_size + crops_coords_top_left + target_size)
        neg = list(negative_original_size + crops_coords_top_left + negative_target_size)
    passed = addition_time_embed_dim * len(ids) + projection_dim
    if expected_add_embed_dim != passed:
        raise ValueError(f"Model expects an added time embedding vector of length {expected_add_embed_dim}, but a vector of {passed} was created.")
    return torch.tensor([ids], dtype=dtype), torch.tensor([neg], dtype=dtype)


class EulerDiscreteSchedulerRef:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.N = num_train_timesteps
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.steps_offset = steps_offset
        self.timesteps = None
        self.sigmas = None

    def set_timesteps(self, n):
        ac = self.alphas_cumprod.numpy()
        train_sigmas = ((1 - ac) / ac) ** 0.5
        ts = (np.arange(0, n) * (self.N // n)).round()[::-1].copy().astype(np.float32) + self.steps_offset     # "leading"
        sig = np.interp(ts, np.arange(0, len(train_sigmas)), train_sigmas)                                       # linear interpolation
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = ts

    def scale_model_input(self, x, i):
        return x / (float(self.sigmas[i]) ** 2 + 1) ** 0.5

    def step(self, eps, i, x):
        sigma = float(self.sigmas[i])
        pred_original = x - sigma * eps                    # epsilon prediction, gamma = 0 -> sigma_hat = sigma
        derivative = (x - pred_original) / sigma
        return x + derivative * (float(self.sigmas[i + 1]) - sigma)

    def add_noise(self, x0, noise, i):
        return x0 + noise * float(self.sigmas[i])


def img2img_loop(unet, sched: EulerDiscreteSchedulerRef, latents, noise, ctx, added, num_inference_steps, strength,
                 guidance_scale=5.0, neg_ctx=None, neg_added=None, trace=None):
    """clean latents -> noised at the first kept timestep -> denoised (strength selects the tail of the schedule)"""
    sched.set_timesteps(num_inference_steps)
    init = min(int(num_inference_steps * strength), num_inference_steps)
    t_start = max(num_inference_steps - init, 0)
    do_cfg = neg_ctx is not None and guidance_scale > 1.0
    if do_cfg:
        ctx2 = torch.cat([neg_ctx, ctx], dim=0)
        added2 = {k: torch.cat([neg_added[k], added[k]], dim=0) for k in ("text_embeds", "time_ids")}
    x = sched.add_noise(latents, noise, t_start)
    for i in range(t_start, num_inference_steps):
        t = float(sched.timesteps[i])
        if do_cfg:
            e = unet(sched.scale_model_input(torch.cat([x] * 2), i), t, ctx2, added_cond_kwargs=added2)[0]
            eu, ec = e.chunk(2)
            eps = eu + guidance_scale * (ec - eu)
        else:
            eps = unet(sched.scale_model_input(x, i), t, ctx, added_cond_kwargs=added)[0]
        x = sched.step(eps, i, x)
        if trace is not None:
            trace.append(x.clone())
    return x
