"""Refiner pass: image-to-image denoising with the second UNet of the same engine.

  StableDiffusionXLImg2ImgPipeline.__call__(image=, prompt=, strength=)   <- how the reference runs
      `self.piperf` (instructany2pix/pipeline.py:128-131 construction, :358-361 call with strength=refinement and the
      class defaults: 50 steps, guidance 5.0, aesthetic score 6.0 / 2.5, Euler scheduler of the refiner checkpoint)
  get_add_time_ids_aesthetic(...)                                          <- `_get_add_time_ids`, requires_aesthetics_score
      branch, which the reference vendors in-tree (instructany2pix/ddim/pnp_pipeline.py:23-71)

The loop follows diffusers 0.26.3 `StableDiffusionXLImg2ImgPipeline.__call__` (the class the reference imports at
pipeline.py:16): `get_timesteps` (strength -> tail of the schedule), `prepare_latents` (VAE-encode, scale, add noise at the
first kept timestep), then per step `cat([x]*2)` -> `scale_model_input` -> UNet -> CFG -> Euler `step`. Every tensor update
runs on the GPU through the C ABI (`ia2p_unet_forward`, `ia2p_ddim_step`); text encoder 2 and the VAE are injectable
callables as in ddim.py (`instructany2pix_amd.vae.HipAutoencoderKL` is the HIP VAE).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .ddim import StableDiffusionXLPipelineOutput, _PipelineBase
from .scheduler import EulerDiscreteScheduler, fused_update


def get_add_time_ids_aesthetic(unet, original_size, crops_coords_top_left, target_size, aesthetic_score, negative_aesthetic_score,
                               negative_original_size, negative_crops_coords_top_left, negative_target_size, projection_dim,
                               requires_aesthetics_score: bool = True, dtype=torch.float16):
    """(add_time_ids, add_neg_time_ids) with the same three mismatch errors as the reference (pnp_pipeline.py:23-71)."""
    if requires_aesthetics_score:
        ids = list(original_size + crops_coords_top_left + (aesthetic_score,))
        neg = list(negative_original_size + negative_crops_coords_top_left + (negative_aesthetic_score,))
    else:
        ids = list(original_size + crops_coords_top_left + target_size)
        neg = list(negative_original_size + crops_coords_top_left + negative_target_size)
    passed = unet.config.addition_time_embed_dim * len(ids) + projection_dim
    expected = unet.add_embedding.linear_1.in_features
    head = f"Model expects an added time embedding vector of length {expected}, but a vector of {passed} was created."
    if expected > passed and expected - passed == unet.config.addition_time_embed_dim:
        raise ValueError(f"{head} Please make sure to enable `requires_aesthetics_score` with `pipe.register_to_config(requires_aesthetics_score=True)` "
                         f"to make sure `aesthetic_score` {aesthetic_score} and `negative_aesthetic_score` {negative_aesthetic_score} is correctly used by the model.")
    if expected < passed and passed - expected == unet.config.addition_time_embed_dim:
        raise ValueError(f"{head} Please make sure to disable `requires_aesthetics_score` with `pipe.register_to_config(requires_aesthetics_score=False)` "
                         f"to make sure `target_size` {target_size} is correctly used by the model.")
    if expected != passed:
        raise ValueError(f"{head} The model has an incorrect config. Please check `unet.config.time_embedding_type` and `text_encoder_2.config.projection_dim`.")
    return torch.tensor([ids], dtype=dtype), torch.tensor([neg], dtype=dtype)


class StableDiffusionXLImg2ImgPipeline(_PipelineBase):
    """The refiner pipeline object (`InstructAny2PixPipeline.piperf`)."""

    def __init__(self, unet, scheduler: Optional[EulerDiscreteScheduler] = None, encode_prompt=None, vae_encode=None, vae_decode=None,
                 requires_aesthetics_score: bool = True):
        super().__init__(unet, scheduler or EulerDiscreteScheduler(), encode_prompt, vae_encode, vae_decode)
        self.config = type("Config", (), {"requires_aesthetics_score": requires_aesthetics_score})()

    def get_timesteps(self, num_inference_steps: int, strength: float):
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start * self.scheduler.order:], num_inference_steps - t_start, t_start

    @torch.no_grad()
    def __call__(self, prompt=None, image=None, strength: float = 0.3, num_inference_steps: int = 50, guidance_scale: float = 5.0,
                 negative_prompt=None, num_images_per_prompt: int = 1, generator=None, latents: Optional[torch.Tensor] = None,
                 noise: Optional[torch.Tensor] = None, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                 negative_pooled_prompt_embeds=None, output_type="pil", return_dict=True, callback=None, callback_steps=1,
                 cross_attention_kwargs=None, original_size: Tuple[int, int] = None, crops_coords_top_left=(0, 0),
                 target_size: Tuple[int, int] = None, negative_original_size=None, negative_crops_coords_top_left=(0, 0),
                 negative_target_size=None, aesthetic_score: float = 6.0, negative_aesthetic_score: float = 2.5, **unused):
        """`latents` = clean image latents (already scaled by the VAE scaling factor) instead of `image`; `noise` = the
        start noise instead of drawing it from `generator` (both are conveniences for tests and latent-space callers)."""
        if strength < 0 or strength > 1:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if num_inference_steps is None or not isinstance(num_inference_steps, int) or num_inference_steps <= 0:
            raise ValueError(f"`num_inference_steps` has to be a positive integer but is {num_inference_steps} of type {type(num_inference_steps)}.")
        self._check_embeds(prompt, prompt_embeds, pooled_prompt_embeds)
        do_cfg = guidance_scale > 1.0
        if prompt_embeds is None:
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = self.encode_prompt(
                prompt=prompt, num_images_per_prompt=num_images_per_prompt, do_classifier_free_guidance=do_cfg, negative_prompt=negative_prompt)
        if do_cfg and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None):
            raise ValueError("classifier-free guidance needs negative_prompt_embeds and negative_pooled_prompt_embeds")
        dev = self.device
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        timesteps, n_steps, t_start = self.get_timesteps(num_inference_steps, strength)
        if n_steps < 1:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of pipeline steps is "
                             f"{n_steps} which is < 1 and not appropriate for this pipeline.")

        # prepare_latents: encode, scale, add noise at the first kept timestep
        if latents is None:
            if image is None:
                raise ValueError("img2img needs `image` (with a vae_encode callable) or `latents`")
            if self._vae_encode is None:
                raise NotImplementedError("VAE encode is not attached: pass latents=, or construct with vae_encode=<callable>")
            latents = self._vae_encode(image)
        latents = latents.to(device=dev, dtype=torch.float16).contiguous()
        batch = latents.shape[0]
        if noise is None:
            g = generator if not isinstance(generator, list) else generator[0]
            gen_dev = g.device if g is not None else torch.device("cpu")
            noise = torch.randn(latents.shape, generator=g, device=gen_dev, dtype=torch.float16)
        noise = noise.to(device=dev, dtype=torch.float16).contiguous()
        x = self.scheduler.add_noise(latents, noise, timesteps[:1])

        height, width = latents.shape[-2] * self.vae_scale_factor, latents.shape[-1] * self.vae_scale_factor
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        negative_original_size = negative_original_size or original_size
        negative_target_size = negative_target_size or target_size
        ids, neg_ids = get_add_time_ids_aesthetic(self.unet, original_size, crops_coords_top_left, target_size, aesthetic_score,
                                                  negative_aesthetic_score, negative_original_size, negative_crops_coords_top_left,
                                                  negative_target_size, int(pooled_prompt_embeds.shape[-1]),
                                                  self.config.requires_aesthetics_score)
        f16 = lambda t: t.to(device=dev, dtype=torch.float16)
        prompt_embeds, add_text_embeds = f16(prompt_embeds), f16(pooled_prompt_embeds)
        if prompt_embeds.shape[0] != batch:
            prompt_embeds, add_text_embeds = prompt_embeds.expand(batch, -1, -1), add_text_embeds.expand(batch, -1)
        add_time_ids = ids.repeat(batch, 1)
        if do_cfg:
            neg_e, neg_p = f16(negative_prompt_embeds), f16(negative_pooled_prompt_embeds)
            if neg_e.shape[0] != batch:
                neg_e, neg_p = neg_e.expand(batch, -1, -1), neg_p.expand(batch, -1)
            prompt_embeds = torch.cat([neg_e, prompt_embeds], dim=0)
            add_text_embeds = torch.cat([neg_p, add_text_embeds], dim=0)
            add_time_ids = torch.cat([neg_ids.repeat(batch, 1), add_time_ids], dim=0)
        added = {"text_embeds": add_text_embeds.contiguous(), "time_ids": add_time_ids.to(dev)}
        prompt_embeds = prompt_embeds.contiguous()

        B = batch
        model_in = torch.empty((2 * B if do_cfg else B,) + tuple(x.shape[1:]), dtype=torch.float16, device=dev)
        eps = torch.empty_like(model_in)
        nxt = torch.empty_like(x)
        for i, t in enumerate(timesteps):
            idx = t_start + i
            s_in = self.scheduler.input_scale(idx)                       # scale_model_input on cat([x]*2)
            fused_update(x, x, None, 1.0, s_in, 0.0, model_in[:B], model_in[B:] if do_cfg else None)
            self.unet(model_in, float(t), encoder_hidden_states=prompt_embeds, cross_attention_kwargs=cross_attention_kwargs,
                      added_cond_kwargs=added, return_dict=False, out=eps)
            c_x, c_e = self.scheduler.step_coeffs(idx)
            if do_cfg:
                fused_update(x, eps[:B], eps[B:], guidance_scale, c_x, c_e, nxt)
            else:
                fused_update(x, eps, None, 1.0, c_x, c_e, nxt)
            x, nxt = nxt, x
            if callback is not None and i % callback_steps == 0:
                callback(i, t, x)
        if output_type == "latent":
            image_out = x
        else:
            if self._vae_decode is None:
                raise NotImplementedError("VAE decode is not attached: use output_type='latent' or pass vae_decode=")
            image_out = self._vae_decode(x)
        return StableDiffusionXLPipelineOutput(images=image_out) if return_dict else (image_out,)
