"""DDIM inversion and sampling loops of the denoise hot path, behind the reference's pipeline surfaces.

  SDXLDDIMPipeline.inverse(...)            <- instructany2pix/ddim/pnp_pipeline.py:92-278 (hot loop :251-275)
  StableDiffusionXLPipeline.__call__(...)  <- the diffusers SDXL loop the reference drives through
                                              IPAdapterXL.generate (ip_adapter.py:346-354); loop text vendored at
                                              instructany2pix/ddim/sdxl_pipeline.py:764-857
Every tensor update runs on the GPU through the C ABI (UNet: ia2p_unet_forward; CFG + DDIM: ia2p_ddim_step).

The stages either side are injectable callables: CLIP text encoders (`encode_prompt`, out of scope) and the VAE
(`vae_encode` / `vae_decode`; `instructany2pix_amd.vae.HipAutoencoderKL` provides HIP implementations). Without them
the loops take `prompt_embeds` / `pooled_prompt_embeds` / `latents` directly, which is also how bench.py and the
parity tests drive them.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Callable, Optional, Tuple

import torch

from .scheduler import DDIMScheduler, fused_update


class StableDiffusionXLPipelineOutput(SimpleNamespace):
    """`.images` like diffusers' output class (reference pnp_pipeline.py:278)."""


def get_add_time_ids(unet, original_size, crops_coords_top_left, target_size, projection_dim, dtype=torch.float16):
    """`_get_add_time_ids` (reference pnp_pipeline.py:23-71, requires_aesthetics_score=False branch;
    same check as sdxl_pipeline.py:506-520)."""
    add_time_ids = list(original_size + crops_coords_top_left + target_size)
    passed = unet.config.addition_time_embed_dim * len(add_time_ids) + projection_dim
    expected = unet.add_embedding.linear_1.in_features
    if expected != passed:
        raise ValueError(f"Model expects an added time embedding vector of length {expected}, but a vector of {passed} was created. "
                         f"The model has an incorrect config. Please check `unet.config.time_embedding_type` and "
                         f"`text_encoder_2.config.projection_dim`.")
    return torch.tensor([add_time_ids], dtype=dtype)


class _PipelineBase:
    vae_scale_factor = 8

    def __init__(self, unet, scheduler: Optional[DDIMScheduler] = None, encode_prompt: Optional[Callable] = None,
                 vae_encode: Optional[Callable] = None, vae_decode: Optional[Callable] = None):
        self.unet = unet
        self.scheduler = scheduler or DDIMScheduler()
        self._encode_prompt = encode_prompt
        self._vae_encode = vae_encode
        self._vae_decode = vae_decode

    @property
    def device(self):
        return self.unet.device

    def to(self, *a, **kw):
        return self

    def encode_prompt(self, prompt=None, num_images_per_prompt=1, do_classifier_free_guidance=True, negative_prompt=None, **kw):
        if self._encode_prompt is None:
            raise NotImplementedError("text encoders are outside the denoise hot path: pass prompt_embeds / pooled_prompt_embeds, "
                                      "or construct the pipeline with encode_prompt=<callable>")
        return self._encode_prompt(prompt=prompt, num_images_per_prompt=num_images_per_prompt,
                                   do_classifier_free_guidance=do_classifier_free_guidance, negative_prompt=negative_prompt, **kw)

    def _check_embeds(self, prompt, prompt_embeds, pooled):
        # same conditions as check_inputs (reference sdxl_pipeline.py:429-486) for the arguments that remain
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt_embeds is not None and pooled is None:
            raise ValueError("If `prompt_embeds` are provided, `pooled_prompt_embeds` also have to be passed.")


class SDXLDDIMPipeline(_PipelineBase):
    """DDIM inversion x0 -> xT. No classifier-free guidance (reference :161)."""

    @torch.no_grad()
    def inverse(self, prompt=None, prompt_2=None, image=None, strength: float = 0.3, num_inference_steps: int = 50,
                guidance_scale: float = 5.0, negative_prompt=None, num_images_per_prompt: int = 1, eta: float = 0.0,
                generator=None, latents: Optional[torch.Tensor] = None, prompt_embeds=None, negative_prompt_embeds=None,
                pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None, output_type="pil", return_dict=True,
                cross_attention_kwargs=None, original_size: Tuple[int, int] = None, crops_coords_top_left=(0, 0),
                target_size: Tuple[int, int] = None, image_embeds=None, callback=None, **unused):
        self.scheduler = DDIMScheduler.from_config(self.scheduler.config)                       # :133
        if strength < 0 or strength > 1:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if num_inference_steps is None or not isinstance(num_inference_steps, int) or num_inference_steps <= 0:
            raise ValueError(f"`num_inference_steps` has to be a positive integer but is {num_inference_steps}")
        self._check_embeds(prompt, prompt_embeds, pooled_prompt_embeds)
        if prompt_embeds is None:
            prompt_embeds, _, pooled_prompt_embeds, _ = self.encode_prompt(
                prompt=prompt, num_images_per_prompt=num_images_per_prompt, do_classifier_free_guidance=False, negative_prompt=negative_prompt)
        dev = self.device
        if latents is None:                                                                     # :190-204 (VAE: "next" row)
            if image is None:
                raise ValueError("inverse() needs `image` (with a vae_encode callable) or `latents`")
            if self._vae_encode is None:
                raise NotImplementedError("VAE encode is outside the denoise hot path: pass latents=, or construct with vae_encode=<callable>")
            latents = self._vae_encode(image)
        latents = latents.to(device=dev, dtype=torch.float16).contiguous().clone()    # the loop ping-pongs buffers: never the caller's tensor
        batch = latents.shape[0]
        self.scheduler.set_timesteps(num_inference_steps, device=dev)                           # :192
        height, width = latents.shape[-2] * self.vae_scale_factor, latents.shape[-1] * self.vae_scale_factor
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        add_time_ids = get_add_time_ids(self.unet, original_size, crops_coords_top_left, target_size,
                                        int(pooled_prompt_embeds.shape[-1])).repeat(batch, 1).to(dev)   # :228-240
        prompt_embeds = prompt_embeds.to(device=dev, dtype=torch.float16)
        if prompt_embeds.shape[0] != batch:
            prompt_embeds = prompt_embeds.expand(batch, -1, -1)
        add_text_embeds = pooled_prompt_embeds.to(device=dev, dtype=torch.float16)
        if add_text_embeds.shape[0] != batch:
            add_text_embeds = add_text_embeds.expand(batch, -1)
        prompt_embeds, add_text_embeds = prompt_embeds.contiguous(), add_text_embeds.contiguous()
        added = {"text_embeds": add_text_embeds, "time_ids": add_time_ids}       # `image_embeds` zeros (:246-248) are ignored by a text_time UNet

        eps = torch.empty_like(latents)
        nxt = torch.empty_like(latents)
        prev_t = None
        acp = self.scheduler.alphas_cumprod
        for i, t in enumerate(reversed(self.scheduler.timesteps)):                              # :251 ascending t
            t = int(t)
            self.unet(latents, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=cross_attention_kwargs,
                      added_cond_kwargs=added, return_dict=False, out=eps)
            a_t = acp[t]
            a_prev = acp[prev_t] if prev_t is not None else self.scheduler.final_alpha_cumprod  # :262-267
            prev_t = t
            c_x, c_e = DDIMScheduler.inversion_coeffs(float(a_t), float(a_prev))
            fused_update(latents, eps, None, 1.0, c_x, c_e, nxt)                                # :270-275 _backward_ddim
            latents, nxt = nxt, latents
            if callback is not None:
                callback(i, t, latents)
        return StableDiffusionXLPipelineOutput(images=latents)                                  # :278 raw latents


class StableDiffusionXLPipeline(_PipelineBase):
    """DDIM sampling xT -> x0 with classifier-free guidance (2x batch UNet evaluation)."""

    @torch.no_grad()
    def __call__(self, prompt=None, height=None, width=None, num_inference_steps: int = 50, guidance_scale: float = 5.0,
                 negative_prompt=None, num_images_per_prompt: int = 1, eta: float = 0.0, generator=None, latents=None,
                 prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                 output_type="latent", return_dict=True, callback=None, callback_steps=1, cross_attention_kwargs=None,
                 guidance_rescale: float = 0.0, original_size=None, crops_coords_top_left=(0, 0), target_size=None, **unused):
        if guidance_rescale != 0.0 or eta != 0.0:
            raise NotImplementedError("guidance_rescale / eta are inactive on the reference's path (sdxl_pipeline.py:846, eta=0)")
        self._check_embeds(prompt, prompt_embeds, pooled_prompt_embeds)
        do_cfg = guidance_scale > 1.0                                                           # sdxl_pipeline.py:735
        if prompt_embeds is None:
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = self.encode_prompt(
                prompt=prompt, num_images_per_prompt=num_images_per_prompt, do_classifier_free_guidance=do_cfg, negative_prompt=negative_prompt)
        if do_cfg and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None):
            raise ValueError("classifier-free guidance needs negative_prompt_embeds and negative_pooled_prompt_embeds")
        dev = self.device
        batch = prompt_embeds.shape[0]
        sample_size = getattr(self.unet.config, "sample_size", 128)
        height = height or sample_size * self.vae_scale_factor
        width = width or sample_size * self.vae_scale_factor
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        self.scheduler.set_timesteps(num_inference_steps, device=dev)                           # :765
        shape = (batch, self.unet.config.in_channels, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if latents is None:                                                                     # prepare_latents :489-504
            g = generator if not isinstance(generator, list) else generator[0]
            gen_dev = g.device if g is not None else torch.device("cpu")
            latents = torch.randn(shape, generator=g, device=gen_dev, dtype=torch.float16)
        else:
            if tuple(latents.shape[-2:]) != shape[-2:]:
                height, width = latents.shape[-2] * self.vae_scale_factor, latents.shape[-1] * self.vae_scale_factor
        latents = (latents.to(dev) * self.scheduler.init_noise_sigma).to(torch.float16).contiguous()
        if latents.shape[0] != batch:
            latents = latents.expand(batch, -1, -1, -1).contiguous()
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        add_time_ids = get_add_time_ids(self.unet, original_size, crops_coords_top_left, target_size,
                                        int(pooled_prompt_embeds.shape[-1]))
        f16 = lambda t: t.to(device=dev, dtype=torch.float16)
        add_text_embeds = f16(pooled_prompt_embeds)
        prompt_embeds = f16(prompt_embeds)
        if do_cfg:                                                                              # :800-803
            prompt_embeds = torch.cat([f16(negative_prompt_embeds), prompt_embeds], dim=0)
            add_text_embeds = torch.cat([f16(negative_pooled_prompt_embeds), add_text_embeds], dim=0)
            add_time_ids = torch.cat([add_time_ids, add_time_ids], dim=0)
        add_time_ids = add_time_ids.to(dev).repeat(batch, 1)                                    # :807
        prompt_embeds, add_text_embeds = prompt_embeds.contiguous(), add_text_embeds.contiguous()
        added = {"text_embeds": add_text_embeds, "time_ids": add_time_ids}

        B = latents.shape[0]
        if do_cfg:
            model_in = torch.cat([latents, latents], dim=0)                                     # :826 cat([latents]*2)
            eps = torch.empty_like(model_in)
            nxt_in = torch.empty_like(model_in)
        else:
            model_in, eps, nxt_in = latents, torch.empty_like(latents), torch.empty_like(latents)
        for i, t in enumerate(self.scheduler.timesteps):                                        # :824 descending t
            t = int(t)
            self.unet(model_in, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=cross_attention_kwargs,
                      added_cond_kwargs=added, return_dict=False, out=eps)
            c_x, c_e = self.scheduler.step_coeffs(t)
            if do_cfg:                                                                          # :842-844 + :851 fused
                fused_update(model_in[:B], eps[:B], eps[B:], guidance_scale, c_x, c_e, nxt_in[:B], nxt_in[B:])
            else:
                fused_update(model_in, eps, None, 1.0, c_x, c_e, nxt_in)
            model_in, nxt_in = nxt_in, model_in
            if callback is not None and i % callback_steps == 0:
                callback(i, t, model_in[:B])
        latents = model_in[:B]
        if output_type == "latent":
            image = latents
        else:                                                                                   # :859-871 (VAE: "next" row)
            if self._vae_decode is None:
                raise NotImplementedError("VAE decode is outside the denoise hot path: use output_type='latent' or pass vae_decode=")
            image = self._vae_decode(latents)
        return StableDiffusionXLPipelineOutput(images=image) if return_dict else (image,)
