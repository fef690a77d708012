"""DDIM scheduler for the denoise hot path (host-side tables + the fused HIP update).

Mirrors the surface of diffusers' `DDIMScheduler` that the reference touches:
  DDIMScheduler.from_config(...)                 instructany2pix/pipeline.py:105,307; ddim/pnp_pipeline.py:133
  .set_timesteps / .timesteps                    pnp_pipeline.py:192,251; ddim/sdxl_pipeline.py:765-767
  .alphas_cumprod / .final_alpha_cumprod         pnp_pipeline.py:262-267
  .step(noise_pred, t, latents, eta=0.0)[0]      sdxl_pipeline.py:851
  .scale_model_input / .init_noise_sigma         sdxl_pipeline.py:828,503
Configuration = SDXL-base `scheduler_config.json` fields that DDIM keeps.
The tensor update itself runs in `ia2p_ddim_step` (one fused kernel: CFG combine + x_{t-1}); coefficient
tables are computed here in float32/float64 exactly as diffusers does (float32 cumprod).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import _ffi


class DDIMScheduler:
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1, timestep_spacing="leading", set_alpha_to_one=False, prediction_type="epsilon",
                 clip_sample=False, **unused):
        if beta_schedule != "scaled_linear" or timestep_spacing != "leading" or prediction_type != "epsilon" or clip_sample:
            raise NotImplementedError("only the SDXL-base DDIM configuration is implemented")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, steps_offset=steps_offset, timestep_spacing=timestep_spacing,
                                      set_alpha_to_one=set_alpha_to_one, prediction_type=prediction_type, clip_sample=clip_sample)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config, **kw):
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        d.update(kw)
        return cls(**d)

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    # ---- coefficients: every DDIM move on this path is  out = c_x * x + c_e * eps  ----------------------------
    def step_coeffs(self, t: int):
        """x_{t-1} from x_t (eta = 0, epsilon prediction, no clipping)."""
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        c_x = math.sqrt(a_p / a_t)
        c_e = math.sqrt(1.0 - a_p) - math.sqrt(a_p) * math.sqrt(1.0 - a_t) / math.sqrt(a_t)
        return c_x, c_e

    @staticmethod
    def inversion_coeffs(alpha_t: float, alpha_tm1: float):
        """`_backward_ddim` (reference pnp_pipeline.py:73-85): sqrt(a) * (x / sqrt(b) + (sqrt(1/a-1) - sqrt(1/b-1)) eps)."""
        a, b = float(alpha_t), float(alpha_tm1)
        return math.sqrt(a / b), math.sqrt(a) * (math.sqrt(1.0 / a - 1.0) - math.sqrt(1.0 / b - 1.0))

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, return_dict: bool = False, **kw):
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not on the reference's path")
        c_x, c_e = self.step_coeffs(int(timestep))
        out = torch.empty_like(sample)
        fused_update(sample, model_output, None, 1.0, c_x, c_e, out)
        return (out,) if not return_dict else SimpleNamespace(prev_sample=out)


def fused_update(x, eps_u, eps_c, guidance, c_x, c_e, out, out2=None):
    """out = c_x*x + c_e*(eps_u + g*(eps_c - eps_u)) on the current stream (ia2p_ddim_step)."""
    for t in (x, eps_u, out):
        assert t.dtype == torch.float16 and t.is_cuda and t.is_contiguous(), "latents must be contiguous fp16 device tensors"
    n = x.numel()
    assert eps_u.numel() == n and out.numel() == n and (eps_c is None or eps_c.numel() == n)
    L = _ffi.lib()
    _ffi.check(L.ia2p_ddim_step(_ffi.current_stream(), _ffi.ptr(x), _ffi.ptr(eps_u), _ffi.ptr(eps_c), float(guidance),
                                float(c_x), float(c_e), _ffi.ptr(out), _ffi.ptr(out2), n))
    return out
