"""HIP-backed VAE with the surface the reference uses on diffusers' `AutoencoderKL` (`pipe.vae`):

  vae.encode(image).latent_dist.sample(generator) * vae.config.scaling_factor     img2img `prepare_latents`, reached from
                                                                                   instructany2pix/ddim/pnp_pipeline.py:195-204
  vae.decode(latents / vae.config.scaling_factor, return_dict=False)[0]           instructany2pix/ddim/sdxl_pipeline.py:859-871
  vae.config.{scaling_factor, force_upcast}, vae.dtype

First "next" row of SURVEY.md §8f. All arithmetic runs in libia2p_hip.so (`ia2p_vae_encode` / `ia2p_vae_decode`).
Precision note: the reference upcasts this model to fp32 (`force_upcast`, sdxl_pipeline.py:860-865) because the residual stream of the
original SDXL VAE checkpoint passes the fp16 maximum. The HIP executor keeps fp16 storage (fp32 accumulation and norm statistics) and
extends its range instead: the stream is stored multiplied by `VAEConfig.stream_scale` (2^-7: magnitudes up to 8.4e6), exactly undone by
the GroupNorms (csrc/vae_engine.hip). So `config.force_upcast` reads False (callers need not upcast anything), and every output is checked
for non-finite values: an overflow raises instead of producing black images (tests/test_vae_gpu.py drives both).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Optional

import torch

from . import _ffi
from .config import VAEConfig


class DiagonalGaussianDistribution:
    """diffusers' posterior object: parameters = [mean | logvar] along channels, logvar clamped to [-30, 20]."""

    def __init__(self, parameters: torch.Tensor):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters.float(), 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        dev = generator.device if generator is not None else self.mean.device
        noise = torch.randn(self.mean.shape, generator=generator, device=dev, dtype=torch.float32).to(self.mean.device)
        return (self.mean + self.std * noise).to(self.parameters.dtype)

    def mode(self) -> torch.Tensor:
        return self.mean.to(self.parameters.dtype)


class HipAutoencoderKL:
    dtype = torch.float16

    def __init__(self, config: VAEConfig, device="cuda:0"):
        config.validate()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _ffi.IA2PError("HipAutoencoderKL runs only on an MI355X device (no CPU path exists)")
        self._cfg = config
        self.config = SimpleNamespace(**{**config.__dict__, "force_upcast": False})
        self._lib = _ffi.lib()
        torch.cuda.set_device(self.device)
        if not self._lib.ia2p_device_is_gfx950():
            raise _ffi.IA2PError("libia2p_hip.so is compiled for gfx950 only")
        self._h = C.c_void_p()
        _ffi.check(self._lib.ia2p_vae_create(C.byref(_ffi.make_vae_config(config)), C.byref(self._h)), None, vae=True)
        n = self._lib.ia2p_vae_arena_bytes(self._h)
        self.arena = torch.zeros(n, dtype=torch.uint8, device=self.device)
        _ffi.check(self._lib.ia2p_vae_bind_arena(self._h, _ffi.ptr(self.arena), n), self._h, vae=True)
        self._ws: Optional[torch.Tensor] = None
        self._factor = 2 ** (len(config.block_out_channels) - 1)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.ia2p_vae_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def to(self, *a, **kw):
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        items = state_dict.items() if hasattr(state_dict, "items") else state_dict
        torch.cuda.set_device(self.device)
        for k, v in items:
            t = v.detach().to(device=self.device, dtype=torch.float16).contiguous()
            shape = (C.c_int64 * t.ndim)(*t.shape)
            _ffi.check(self._lib.ia2p_vae_load_tensor(self._h, k.encode(), _ffi.ptr(t), shape, t.ndim, _ffi.current_stream()), self._h, vae=True)
            torch.cuda.current_stream().synchronize()
        if strict:
            _ffi.check(self._lib.ia2p_vae_finalize_weights(self._h), self._h, vae=True)

    def _workspace(self, B, h, w, decode):
        n = self._lib.ia2p_vae_workspace_bytes(self._h, B, h, w, int(decode))
        if n == 0:
            _ffi.check(2, self._h, vae=True)
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._ws

    @torch.no_grad()
    def encode(self, image: torch.Tensor, return_dict: bool = True):
        B, c, H, W = image.shape
        f = self._factor
        if c != self._cfg.in_channels or H % f or W % f:
            raise ValueError(f"image must be [B, {self._cfg.in_channels}, H, W] with H, W divisible by {f}")
        x = image.to(device=self.device, dtype=torch.float16).contiguous()
        h, w = H // f, W // f
        ws = self._workspace(B, h, w, False)
        mom = torch.empty(B, 2 * self._cfg.latent_channels, h, w, dtype=torch.float16, device=self.device)
        _ffi.check(self._lib.ia2p_vae_encode(self._h, _ffi.current_stream(), _ffi.ptr(x), _ffi.ptr(mom), B, h, w, _ffi.ptr(ws), ws.numel()), self._h, vae=True)
        self._check_finite(mom, "encode")
        dist = DiagonalGaussianDistribution(mom)
        return SimpleNamespace(latent_dist=dist) if return_dict else (dist,)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True):
        B, c, h, w = z.shape
        if c != self._cfg.latent_channels:
            raise ValueError(f"latents must have {self._cfg.latent_channels} channels")
        x = z.to(device=self.device, dtype=torch.float16).contiguous()
        f = self._factor
        ws = self._workspace(B, h, w, True)
        img = torch.empty(B, self._cfg.out_channels, h * f, w * f, dtype=torch.float16, device=self.device)
        _ffi.check(self._lib.ia2p_vae_decode(self._h, _ffi.current_stream(), _ffi.ptr(x), _ffi.ptr(img), B, h, w, _ffi.ptr(ws), ws.numel()), self._h, vae=True)
        self._check_finite(img, "decode")
        return SimpleNamespace(sample=img) if return_dict else (img,)

    check_finite = True        # every encode / decode ends with a blocking device-to-host check for non-finite values (an fp16 overflow must not pass
                               # silently where the reference runs fp32); serving loops that check elsewhere may set this to False

    def _check_finite(self, t, what):
        if self.check_finite and not bool(torch.isfinite(t).all()):
            raise _ffi.IA2PError(f"HipAutoencoderKL.{what}: non-finite output -- the activations left the range of fp16 storage at stream_scale="
                                 f"{self._cfg.stream_scale:g} (the reference runs this model in fp32); lower VAEConfig.stream_scale (a power of two)")

    # hooks with the signatures the loops in ddim.py accept (vae_encode= / vae_decode=)
    def encode_to_latents(self, image, generator=None):
        return self.encode(image).latent_dist.sample(generator) * self._cfg.scaling_factor

    def decode_from_latents(self, latents):
        return self.decode(latents / self._cfg.scaling_factor, return_dict=False)[0]
