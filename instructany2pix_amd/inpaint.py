"""Subject-consistency inpainting loop: the base UNet re-used with 'local' IP-Adapter tokens and a
per-step latent blend under a mask.

  StableDiffusionXLInpaintPipeline.__call__(image=, mask_image=, strength=, prompt_embeds=..., ...)
      <- the object the reference builds as `self.pipe_inpainting` from the base pipeline's own modules
         (instructany2pix/pipeline.py:132-139) and drives through `IPAdapterXL(pipe_inpainting).generate(image=, mask_image=,
         strength=subject_strength, clip_image_embeds_local=emb[None], mode='local', num_inference_steps=50, scale=0.8)`
         (instructany2pix/gdino/lib.py:89-102, called from pipeline.py:363-368)
  subject_consistency(...)   <- the loop over detected subjects of gdino/lib.py:69-104, on given masks (SAM / GroundingDINO
                                produce them in the reference; they are outside this path)

The denoising loop follows diffusers 0.26.3 `StableDiffusionXLInpaintPipeline.__call__` for a 4-channel UNet: `get_timesteps`
(strength -> tail of the schedule), start latents = noise (strength 1) or `add_noise(image_latents, noise, t_0)`, the mask resized to
the latent grid (nearest) after binarising at 0.5, and after every scheduler step
    latents = (1 - mask) * add_noise(image_latents, noise, t_{i+1}) + mask * latents        (no re-noising after the last step)
with the shared DDIM scheduler. UNet: `ia2p_unet_forward`; CFG + DDIM step: `ia2p_ddim_step`; blend: `ia2p_mask_blend`.
"""
from __future__ import annotations

from typing import Optional

import torch

from .ddim import StableDiffusionXLPipelineOutput, _PipelineBase, get_add_time_ids
from .scheduler import DDIMScheduler, fused_update, mask_blend


def prepare_mask(mask_image, h: int, w: int, device) -> torch.Tensor:
    """`mask_processor.preprocess` (grayscale, binarise at 0.5, no normalisation) + `interpolate(size=(h, w))` (nearest) of
    diffusers' `prepare_mask_latents`: any [H,W] / [1,H,W] / [B,1,H,W] array in [0,1] (or uint8 0..255) -> [B,1,h,w] fp16 in {0,1}."""
    m = torch.as_tensor(mask_image)
    if m.dtype == torch.uint8:
        m = m.float() / 255.0
    m = m.float()
    while m.ndim < 4:
        m = m[None]
    if m.shape[1] != 1:
        raise ValueError(f"mask must have one channel, got shape {tuple(m.shape)}")
    m = (m >= 0.5).float()
    m = torch.nn.functional.interpolate(m, size=(h, w))
    return m.to(device=device, dtype=torch.float16).contiguous()


class StableDiffusionXLInpaintPipeline(_PipelineBase):
    def get_timesteps(self, num_inference_steps: int, strength: float):
        init_timestep = min(int(num_inference_steps * strength), num_inference_steps)
        t_start = max(num_inference_steps - init_timestep, 0)
        return self.scheduler.timesteps[t_start * self.scheduler.order:], num_inference_steps - t_start

    @torch.no_grad()
    def __call__(self, prompt=None, image=None, mask_image=None, height=None, width=None, strength: float = 0.9999,
                 num_inference_steps: int = 50, guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: int = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                 prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None,
                 output_type="pil", return_dict=True, callback=None, callback_steps=1, cross_attention_kwargs=None,
                 guidance_rescale: float = 0.0, original_size=None, crops_coords_top_left=(0, 0), target_size=None, **unused):
        """`latents` = clean image latents (scaled) instead of `image`; `noise` = the start / re-noising noise instead of drawing it
        from `generator` (conveniences for tests and latent-space callers, as in img2img.py)."""
        if strength < 0 or strength > 1:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if guidance_rescale != 0.0 or eta != 0.0:
            raise NotImplementedError("guidance_rescale / eta are inactive on the reference's path")
        if mask_image is None:
            raise ValueError("`mask_image` input cannot be undefined.")
        self._check_embeds(prompt, prompt_embeds, pooled_prompt_embeds)
        do_cfg = guidance_scale > 1.0
        if prompt_embeds is None:
            prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds = self.encode_prompt(
                prompt=prompt, num_images_per_prompt=num_images_per_prompt, do_classifier_free_guidance=do_cfg, negative_prompt=negative_prompt)
        if do_cfg and (negative_prompt_embeds is None or negative_pooled_prompt_embeds is None):
            raise ValueError("classifier-free guidance needs negative_prompt_embeds and negative_pooled_prompt_embeds")
        dev = self.device
        self.scheduler.set_timesteps(num_inference_steps, device=dev)
        timesteps, n_steps = self.get_timesteps(num_inference_steps, strength)
        if n_steps < 1:
            raise ValueError(f"After adjusting the num_inference_steps by strength parameter: {strength}, the number of pipeline steps is "
                             f"{n_steps} which is < 1 and not appropriate for this pipeline.")
        if latents is None:
            if image is None:
                raise ValueError("inpainting needs `image` (with a vae_encode callable) or `latents`")
            if self._vae_encode is None:
                raise NotImplementedError("VAE encode is not attached: pass latents=, or construct with vae_encode=<callable>")
            latents = self._vae_encode(image)
        image_latents = latents.to(device=dev, dtype=torch.float16).contiguous()
        B, _, h, w = image_latents.shape
        mask = prepare_mask(mask_image, h, w, dev)
        if mask.shape[0] != B:
            mask = mask.expand(B, -1, -1, -1).contiguous()
        if noise is None:
            g = generator if not isinstance(generator, list) else generator[0]
            gen_dev = g.device if g is not None else torch.device("cpu")
            noise = torch.randn(image_latents.shape, generator=g, device=gen_dev, dtype=torch.float16)
        noise = noise.to(device=dev, dtype=torch.float16).contiguous()
        if strength == 1.0:
            x = (noise * self.scheduler.init_noise_sigma).contiguous()
        else:
            x = self.scheduler.add_noise(image_latents, noise, timesteps[:1])

        height, width = h * self.vae_scale_factor, w * self.vae_scale_factor
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        add_time_ids = get_add_time_ids(self.unet, original_size, crops_coords_top_left, target_size, int(pooled_prompt_embeds.shape[-1]))
        f16 = lambda t: t.to(device=dev, dtype=torch.float16)
        prompt_embeds, add_text_embeds = f16(prompt_embeds), f16(pooled_prompt_embeds)
        if prompt_embeds.shape[0] != B:
            prompt_embeds, add_text_embeds = prompt_embeds.expand(B, -1, -1), add_text_embeds.expand(B, -1)
        add_time_ids = add_time_ids.repeat(B, 1)
        if do_cfg:
            neg_e, neg_p = f16(negative_prompt_embeds), f16(negative_pooled_prompt_embeds)
            if neg_e.shape[0] != B:
                neg_e, neg_p = neg_e.expand(B, -1, -1), neg_p.expand(B, -1)
            prompt_embeds = torch.cat([neg_e, prompt_embeds], dim=0)
            add_text_embeds = torch.cat([neg_p, add_text_embeds], dim=0)
            add_time_ids = torch.cat([add_time_ids, add_time_ids], dim=0)
        added = {"text_embeds": add_text_embeds.contiguous(), "time_ids": add_time_ids.to(dev)}
        prompt_embeds = prompt_embeds.contiguous()

        model_in = torch.cat([x, x], dim=0) if do_cfg else x.clone()
        eps = torch.empty_like(model_in)
        stepped = torch.empty_like(x)
        for i, t in enumerate(timesteps):
            t = int(t)
            self.unet(model_in, t, encoder_hidden_states=prompt_embeds, cross_attention_kwargs=cross_attention_kwargs,
                      added_cond_kwargs=added, return_dict=False, out=eps)
            c_x, c_e = self.scheduler.step_coeffs(t)
            if do_cfg:
                fused_update(model_in[:B], eps[:B], eps[B:], guidance_scale, c_x, c_e, stepped)
            else:
                fused_update(model_in, eps, None, 1.0, c_x, c_e, stepped)
            c0, c1 = self.scheduler.add_noise_coeffs(int(timesteps[i + 1])) if i < len(timesteps) - 1 else (1.0, 0.0)
            mask_blend(stepped, image_latents, noise, mask, c0, c1, model_in[:B], model_in[B:] if do_cfg else None)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, model_in[:B])
        out = model_in[:B].clone()
        if output_type == "latent":
            image_out = out
        else:
            if self._vae_decode is None:
                raise NotImplementedError("VAE decode is not attached: use output_type='latent' or pass vae_decode=")
            image_out = self._vae_decode(out)
        return StableDiffusionXLPipelineOutput(images=image_out) if return_dict else (image_out,)


def subject_consistency(subject_data, latents, ip_adapter_xl_inpaint, subject_strength: float = 0.7, **generate_kwargs):
    """The per-subject loop of reference gdino/lib.py:85-103 on given masks: `subject_data` = [(mask, subject embedding)], each pass
    re-paints the masked region guided by the subject's 'local' image tokens (50 steps, IP scale 0.8)."""
    subject = latents
    for msk, emb in subject_data:
        subject = ip_adapter_xl_inpaint.generate(latents=subject, mask_image=msk, pil_image=None, strength=subject_strength,
                                                 clip_image_embeds_local=emb[None] if emb.ndim == 1 else emb, mode="local",
                                                 num_inference_steps=50, scale=0.8, **generate_kwargs)
    return subject
