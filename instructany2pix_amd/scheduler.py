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

    def add_noise_coeffs(self, t: int):
        """x_t = sqrt(abar_t) x_0 + sqrt(1 - abar_t) noise (inpainting: start latents and the re-noised known region)"""
        a = float(self.alphas_cumprod[int(t)])
        return math.sqrt(a), math.sqrt(1.0 - a)

    def add_noise(self, original_samples, noise, timesteps):
        t = timesteps.reshape(-1)[0] if torch.is_tensor(timesteps) else timesteps
        c0, c1 = self.add_noise_coeffs(int(t))
        out = torch.empty_like(original_samples)
        return fused_update(original_samples.contiguous(), noise.to(original_samples.dtype).contiguous(), None, 1.0, c0, c1, out)

    def step(self, model_output, timestep, sample, eta: float = 0.0, generator=None, return_dict: bool = False, **kw):
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not on the reference's path")
        c_x, c_e = self.step_coeffs(int(timestep))
        out = torch.empty_like(sample)
        fused_update(sample, model_output, None, 1.0, c_x, c_e, out)
        return (out,) if not return_dict else SimpleNamespace(prev_sample=out)


def mask_blend(x, init, noise, mask, c0, c1, out, out2=None):
    """out = (1 - mask) * (c0*init + c1*noise) + mask * x on the current stream (ia2p_mask_blend); mask is [B,1,h,w]."""
    B, Cc, h, w = x.shape
    for t in (x, init, noise, mask, out):
        assert t.dtype == torch.float16 and t.is_cuda and t.is_contiguous(), "latents and mask must be contiguous fp16 device tensors"
    assert init.shape == x.shape and noise.shape == x.shape and out.shape == x.shape and tuple(mask.shape) == (B, 1, h, w)
    _ffi.check(_ffi.lib().ia2p_mask_blend(_ffi.current_stream(), _ffi.ptr(x), _ffi.ptr(init), _ffi.ptr(noise), _ffi.ptr(mask), float(c0), float(c1),
                                          _ffi.ptr(out), _ffi.ptr(out2), B, Cc, h * w))
    return out


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


def fused_update_v(x, eps_u, eps_c, coef, out, out2=None):
    """The same update with per-request coefficients (ia2p_ddim_step_v): coef = float32 device tensor [B, 3] = (guidance, c_x, c_e) of every
    batch element of x [B, ...]; requests with their own guidance scale at their own step of their own schedule share one launch."""
    for t in (x, eps_u, out):
        assert t.dtype == torch.float16 and t.is_cuda and t.is_contiguous(), "latents must be contiguous fp16 device tensors"
    B = x.shape[0]
    per = x.numel() // B
    assert coef.dtype == torch.float32 and coef.is_cuda and coef.is_contiguous() and tuple(coef.shape) == (B, 3)
    assert eps_u.numel() == x.numel() and out.numel() == x.numel() and (eps_c is None or eps_c.numel() == x.numel())
    L = _ffi.lib()
    _ffi.check(L.ia2p_ddim_step_v(_ffi.current_stream(), _ffi.ptr(x), _ffi.ptr(eps_u), _ffi.ptr(eps_c), _ffi.ptr(coef), _ffi.ptr(out), _ffi.ptr(out2), B, per))
    return out


class EulerDiscreteScheduler:
    """Euler (first-order, sigma-space) sampler of the SDXL refiner: the scheduler `StableDiffusionXLImg2ImgPipeline.
    from_pretrained("stabilityai/stable-diffusion-xl-refiner-1.0")` instantiates for `self.piperf` (reference
    instructany2pix/pipeline.py:128-131, run at :358-361). diffusers 0.26.3 semantics with the refiner's
    `scheduler_config.json` (scaled-linear betas 0.00085..0.012, 1000 train steps, "leading" spacing, steps_offset 1,
    linear sigma interpolation, epsilon prediction, no Karras sigmas, s_churn 0):

        sigma_t = sqrt((1 - abar_t) / abar_t);   model input = x / sqrt(sigma^2 + 1);   x_next = x + (sigma_next - sigma) * eps

    The tensor updates run in `ia2p_ddim_step` like the DDIM ones: every move is  out = c_x * x + c_e * eps.
    """
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                 steps_offset=1, timestep_spacing="leading", prediction_type="epsilon", interpolation_type="linear",
                 use_karras_sigmas=False, **unused):
        if (beta_schedule != "scaled_linear" or timestep_spacing != "leading" or prediction_type != "epsilon"
                or interpolation_type != "linear" or use_karras_sigmas):
            raise NotImplementedError("only the SDXL-refiner Euler configuration is implemented")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
                                      beta_schedule=beta_schedule, steps_offset=steps_offset, timestep_spacing=timestep_spacing,
                                      prediction_type=prediction_type, interpolation_type=interpolation_type,
                                      use_karras_sigmas=use_karras_sigmas)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self._train_sigmas = (((1 - self.alphas_cumprod) / self.alphas_cumprod) ** 0.5).numpy()      # float32, ascending in t
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.linspace(0, num_train_timesteps - 1, num_train_timesteps, dtype=np.float32)[::-1].copy())
        self.sigmas = torch.from_numpy(np.concatenate([self._train_sigmas[::-1], [0.0]]).astype(np.float32))
        self._step_index: Optional[int] = None

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
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.float32) + self.config.steps_offset
        sig = np.interp(ts, np.arange(0, len(self._train_sigmas)), self._train_sigmas)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self._step_index = None

    @property
    def init_noise_sigma(self) -> float:
        return float((self.sigmas.max() ** 2 + 1) ** 0.5)          # "leading" spacing

    def index_for_timestep(self, timestep) -> int:
        idx = (self.timesteps == float(timestep)).nonzero()
        if len(idx) == 0:
            raise ValueError(f"timestep {timestep} is not on the current schedule")
        return int(idx[1 if len(idx) > 1 else 0])                 # diffusers: second match when a timestep repeats

    # ---- coefficients (all updates are  out = c_x * x + c_e * eps) -----------------------------------------------
    def input_scale(self, index: int) -> float:
        """`scale_model_input`: 1 / sqrt(sigma^2 + 1)"""
        s = float(self.sigmas[index])
        return 1.0 / math.sqrt(s * s + 1.0)

    def step_coeffs(self, index: int):
        """x_next = x + (sigma_{i+1} - sigma_i) * eps   (epsilon prediction: the derivative is eps itself)"""
        return 1.0, float(self.sigmas[index + 1]) - float(self.sigmas[index])

    def scale_model_input(self, sample, timestep):
        i = self._step_index if self._step_index is not None else self.index_for_timestep(timestep)
        out = torch.empty_like(sample)
        return fused_update(sample, sample, None, 1.0, self.input_scale(i), 0.0, out)

    def add_noise(self, original_samples, noise, timesteps):
        """x_t = x_0 + sigma_t * noise (img2img start, sdxl img2img `prepare_latents`)"""
        t = timesteps.reshape(-1)[0] if torch.is_tensor(timesteps) else timesteps
        s = float(self.sigmas[self.index_for_timestep(t)])
        out = torch.empty_like(original_samples)
        return fused_update(original_samples.contiguous(), noise.to(original_samples.dtype).contiguous(), None, 1.0, 1.0, s, out)

    def step(self, model_output, timestep, sample, return_dict: bool = False, **kw):
        if self._step_index is None:
            self._step_index = self.index_for_timestep(timestep)
        c_x, c_e = self.step_coeffs(self._step_index)
        out = torch.empty_like(sample)
        fused_update(sample, model_output, None, 1.0, c_x, c_e, out)
        self._step_index += 1
        return (out,) if not return_dict else SimpleNamespace(prev_sample=out)


def prior_update(sample, out_cond, out_uncond, noise, guidance, sqrt_a, sqrt_b, k0, k1, sigma, out):
    """fp32 sampler update of the embedding prior (`ia2p_prior_step`): per element
    eps_i = (sample - sqrt_a o_i) / sqrt_b, guidance on eps, x0 = (sample - sqrt_b eps) / sqrt_a, out = k0 x0 + k1 sample + sigma noise."""
    for t in (sample, noise, out):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda)
    for t in (out_cond, out_uncond):
        assert t is None or (t.dtype == torch.float16 and t.is_contiguous() and t.is_cuda)
    _ffi.check(_ffi.lib().ia2p_prior_step(_ffi.current_stream(), _ffi.ptr(sample), _ffi.ptr(out_cond), _ffi.ptr(out_uncond), _ffi.ptr(noise),
                                          float(guidance), float(sqrt_a), float(sqrt_b), float(k0), float(k1), float(sigma), _ffi.ptr(out),
                                          sample.numel()), None)
    return out


class DDPMScheduler:
    """diffusers 0.26.3 `DDPMScheduler` as the reference's embedding prior configures and calls it: built from the SDXL-base
    `scheduler_config.json` (`DDPMScheduler.from_pretrained("stabilityai/stable-diffusion-xl-base-1.0", subfolder="scheduler")`,
    instructany2pix/prior/model.py:131), then `.set_timesteps`, `.timesteps`, `.alphas_cumprod`, `.one`, `.num_inference_steps`,
    `.config.num_train_timesteps`, `.order` (:208-219, :588-589, :622) and `.step(noise_pred, t, sample, generator=)` (:635).
    Fields DDPM keeps from that config: scaled-linear betas 0.00085..0.012, 1000 steps, "leading" spacing with steps_offset 1,
    epsilon prediction, no clipping; DDPM defaults for the rest (variance_type "fixed_small", no thresholding)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", steps_offset=1,
                 timestep_spacing="leading", prediction_type="epsilon", clip_sample=False, variance_type="fixed_small", **unused):
        if (beta_schedule != "scaled_linear" or timestep_spacing != "leading" or prediction_type != "epsilon" or clip_sample
                or variance_type != "fixed_small"):
            raise NotImplementedError("only the SDXL-base scheduler configuration is implemented")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end, beta_schedule=beta_schedule,
                                      steps_offset=steps_offset, timestep_spacing=timestep_spacing, prediction_type=prediction_type,
                                      clip_sample=clip_sample, variance_type=variance_type)
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.custom_timesteps = False
        self.num_inference_steps: Optional[int] = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config, **kw):
        d = dict(vars(config)) if not isinstance(config, dict) else dict(config)
        d.update(kw)
        return cls(**d)

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.config.train_timesteps`: "
                             f"{self.config.num_train_timesteps} as the unet model trained with this scheduler can only handle maximal "
                             f"{self.config.num_train_timesteps} timesteps.")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def previous_timestep(self, timestep):
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def posterior_coeffs(self, t: int):
        """(sqrt(abar_t), sqrt(1 - abar_t), k0, k1, sigma): x_{t-1} = k0 x0 + k1 x_t + sigma z with the fixed_small variance
        (clamped at 1e-20, as diffusers does), sigma = 0 at t = 0. Float32 arithmetic on the float32 table, as in diffusers."""
        t = int(t)
        prev = int(self.previous_timestep(t))
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.one
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        k0 = (a_p ** 0.5 * cur_b) / b_t
        k1 = cur_a ** 0.5 * b_p / b_t
        var = torch.clamp((1 - a_p) / (1 - a_t) * cur_b, min=1e-20)
        sigma = float(var ** 0.5) if t > 0 else 0.0
        return float(a_t ** 0.5), float(b_t ** 0.5), float(k0), float(k1), sigma

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = False, **kw):
        """Generic form (epsilon in, x_{t-1} out) in torch; the prior's loop uses the fused `prior_update` instead."""
        sa, sb, k0, k1, sigma = self.posterior_coeffs(int(timestep))
        x0 = (sample - sb * model_output) / sa
        prev = k0 * x0 + k1 * sample
        if int(timestep) > 0:
            z = torch.randn(model_output.shape, generator=generator, device=generator.device if generator is not None else "cpu",
                            dtype=model_output.dtype).to(model_output.device)
            prev = prev + sigma * z
        return (prev,) if not return_dict else SimpleNamespace(prev_sample=prev, pred_original_sample=x0)
