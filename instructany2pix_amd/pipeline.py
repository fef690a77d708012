"""`InstructAny2PixPipeline` call surface for the denoise hot path (reference instructany2pix/pipeline.py).

The reference's `__call__` (:303-386) does, in order: LLM + ImageBind (`forward_llm`, off-path), the GPT-2 prior
(off-path), then THE HOT SEGMENT
    :306-307  share one UNet between the pipelines, fresh DDIM scheduler
    :322-324  fuse base / instruction / prior embeddings into `latent_la` and renormalise to `norm`
    :330      latent_inv = pipe_inversion.inverse(num_inference_steps=N, prompt='', image=img_base)
    :331-337  polar interpolation with fresh noise (CPU, fp16, global torch RNG)
    :342-354  ip_adapter_xl.generate(prompt=..., clip_image_embeds=latent_la[0], latents=latent_inv, guidance_scale=cfg, scale=scale)
followed by the refiner pass (:358-361; `self.piperf`, img2img.py — SURVEY.md §8f rank 2, built; with a VAE attached the base result
reaches it the reference's way: decoded, quantised to 8 bits, re-encoded) and the subject-consistency
pass (:363-368; `self.pipe_inpainting` / `ip_adapter_xl_inpaint`, inpaint.py — rank 3, built; its masks come from SAM / GroundingDINO,
which stay outside).

This class keeps the constructor attributes other code touches (`.pipe`, `.pipe_inversion`, `.ip_adapter_xl`,
`.cache`; serve.py:9 assigns `.pipe.scheduler`) and the `__call__` keyword surface. The off-path stages are
injected as callables (`conditioner`, `text_encoder`, `vae`); `denoise()` is the hot segment itself on
already-computed conditioning and is what bench.py and the parity tests drive.
"""
from __future__ import annotations

from typing import Any, Callable, Optional

import torch

from .config import UNetConfig, sdxl_base
from .ddim import SDXLDDIMPipeline, StableDiffusionXLPipeline
from .img2img import StableDiffusionXLImg2ImgPipeline
from .inpaint import StableDiffusionXLInpaintPipeline, subject_consistency
from .ip_adapter import IPAdapterXL
from .prior import MODALITY
from .scheduler import DDIMScheduler
from .unet import HipUNet2DConditionModel


def polar_intrtpolate(x, y, alpha):
    """reference pipeline.py:295-300 (name kept, typo included); runs where its inputs live (CPU fp16 in the reference)."""
    n0 = x.norm()
    n1 = y.norm()
    ll = x * alpha + y * (1 - alpha)
    n = n0 * alpha + n1 * (1 - alpha)
    return ll / ll.norm() * n


def to_8bit_image(image):
    """What the reference's hand-over between its pipelines does to an image TENSOR IN [-1, 1] (what `vae_decode` hooks return here; a hook that
    returns PIL images or [0, 1] tensors is rejected): `postprocess(output_type="pil")`
    (`(x / 2 + 0.5).clamp(0, 1)`, `* 255`, round, uint8) followed by the img2img pipeline's `preprocess` (`/ 255`, `2 x - 1`)."""
    if not torch.is_tensor(image) or not image.is_floating_point():
        raise TypeError("to_8bit_image expects the decode hook's float tensor in [-1, 1]")
    q = ((image.float() / 2 + 0.5).clamp(0, 1) * 255.0).round()
    return (q / 255.0 * 2.0 - 1.0).to(image.dtype)


def _need(c, keys, stage):
    missing = [k for k in keys if k not in c]
    if missing:
        raise KeyError(f"the conditioner returned no {missing} (needed by the {stage}; pass refinement=0 / subject_strength=0 to skip that stage)")


def fuse_instruction_embedding(base_embed, image_embeds, y0, h, norm):
    """reference pipeline.py:322-324"""
    latent_la = base_embed * h[0] + image_embeds * h[1] + y0 / y0.norm() * 20.0 * h[2]
    latent_la = latent_la.detach().clone()
    return latent_la / latent_la.norm() * norm


class InstructAny2PixPipeline:
    def __init__(self, ckpt: str = "ckpts", llm_folder: str = "llm-retrained", *, unet: Optional[HipUNet2DConditionModel] = None,
                 unet_config: Optional[UNetConfig] = None, unet_state_dict=None, ip_ckpt=None, device: str = "cuda:0",
                 conditioner: Optional[Callable] = None, text_encoder: Optional[Callable] = None,
                 vae_encode: Optional[Callable] = None, vae_decode: Optional[Callable] = None, clip_embeddings_dim: int = 1024,
                 refiner_unet: Optional[HipUNet2DConditionModel] = None, refiner_text_encoder: Optional[Callable] = None, prior=None,
                 refiner_handoff: str = "image"):
        # how the base result reaches the refiner: "image" = the reference's route (decode, 8-bit image, VAE re-encode with a posterior
        # sample; needs vae_encode and vae_decode), "latent" = the sampled latents go in directly (no VAE round trip; the only route
        # when no VAE is attached)
        if refiner_handoff not in ("image", "latent"):
            raise ValueError("refiner_handoff must be 'image' or 'latent'")
        self.refiner_handoff = refiner_handoff
        self._warned_handoff = False
        # the embedding prior (reference :97-98,:120-122 `self.model`): prior.py::InstructAny2PixPrior on the HIP kernels, or None when
        # the conditioner supplies `y` itself
        self.model = prior
        if unet is None:
            unet = HipUNet2DConditionModel(unet_config or sdxl_base(), device)
            if unet_state_dict is not None:
                unet.load_state_dict(unet_state_dict)
        self.unet = unet
        new_sch = DDIMScheduler()
        # one shared UNet object for sampling and inversion (reference :106-116)
        self.pipe = StableDiffusionXLPipeline(unet, DDIMScheduler(), encode_prompt=text_encoder, vae_decode=vae_decode)
        self.pipe_inversion = SDXLDDIMPipeline(unet, new_sch, encode_prompt=text_encoder, vae_encode=vae_encode)
        # the refiner pipeline object (:128-131): second UNet config of the same engine, Euler img2img loop (img2img.py)
        self.piperf = StableDiffusionXLImg2ImgPipeline(refiner_unet, encode_prompt=refiner_text_encoder, vae_encode=vae_encode,
                                                       vae_decode=vae_decode) if refiner_unet is not None else None
        # inpainting pipeline assembled from the base pipeline's own modules: same UNet object, same scheduler object (:132-139)
        self.pipe_inpainting = StableDiffusionXLInpaintPipeline(unet, self.pipe.scheduler, encode_prompt=text_encoder, vae_encode=vae_encode,
                                                                vae_decode=vae_decode)
        self.conditioner = conditioner           # stands in for forward_llm + prior (:309-317)
        self.cache = None
        self.mode = "ipa_v2"
        self.ip_adapter_xl = IPAdapterXL(self.pipe, "", ip_ckpt=ip_ckpt, device=device, clip_embeddings_dim=clip_embeddings_dim) if ip_ckpt is not None else None
        # second adapter object over the same UNet (:143-146): it re-installs the same processors and weights
        self.ip_adapter_xl_inpaint = IPAdapterXL(self.pipe_inpainting, "", ip_ckpt=ip_ckpt, device=device,
                                                 clip_embeddings_dim=clip_embeddings_dim) if ip_ckpt is not None else None

    # ---- the hot segment on explicit conditioning ------------------------------------------------------------------
    @torch.no_grad()
    def denoise(self, base_latents, latent_la, *, prompt_embeds, pooled_prompt_embeds, negative_prompt_embeds, negative_pooled_prompt_embeds,
                inv_prompt_embeds=None, inv_pooled_prompt_embeds=None, alpha=0.7, num_inference_steps=25, cfg=10, scale=1.0, noise=None):
        """inversion -> polar mixing -> IP-Adapter guided sampling; returns (sampled latents, inverted latents)."""
        self.pipe_inversion.unet = self.pipe.unet                                              # :306
        self.pipe_inversion.scheduler = DDIMScheduler.from_config(self.pipe.scheduler.config)  # :307
        if inv_prompt_embeds is None:        # reference inverts with prompt='' (:330); callers pass its embedding
            inv_prompt_embeds, inv_pooled_prompt_embeds = negative_prompt_embeds, negative_pooled_prompt_embeds
        latent_inv = self.pipe_inversion.inverse(num_inference_steps=num_inference_steps, latents=base_latents,
                                                 prompt_embeds=inv_prompt_embeds, pooled_prompt_embeds=inv_pooled_prompt_embeds).images
        latent_inv_cpu = latent_inv.cpu()                                                      # :331
        if noise is None:
            noise = torch.randn_like(latent_inv_cpu)                                           # :335 global RNG, CPU, fp16
        mixed = polar_intrtpolate(latent_inv_cpu, noise, alpha)                                # :333-337
        images = self.ip_adapter_xl.generate(pil_image=None, num_samples=1, clip_image_embeds=latent_la, num_inference_steps=num_inference_steps,
                                             scale=scale, mode="global", guidance_scale=cfg, latents=mixed,
                                             prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                                             pooled_prompt_embeds=pooled_prompt_embeds, negative_pooled_prompt_embeds=negative_pooled_prompt_embeds,
                                             output_type="latent")
        return images, latent_inv

    # ---- N independent requests as one batch per evaluation, sharded over the ranks (batch.py) ---------------------------------------------------
    def denoise_batch(self, requests, group: int = 4, shard: bool = True):
        """`denoise` for a list of `batch.EditRequest`s with their own `num_inference_steps` / `cfg` / `scale` / `alpha`: grouped `group` at a time
        (B_eff = 2 x group in the guided loop), contiguous shards per rank when a process group is up, results all-gathered in request order.
        -> (sampled latents [N,4,h,w], inverted latents [N,4,h,w])."""
        from .batch import denoise_batch
        self.pipe_inversion.unet = self.pipe.unet
        return denoise_batch(self, requests, group=group, shard=shard)

    # ---- reference keyword surface ----------------------------------------------------------------------------------
    def __call__(self, inst, mm_data, alpha=0.7, h=[0.0, 0.4, 1.0], norm=20.0, refinement=0.5, llm_only=False, num_inference_steps=25,
                 use_cache=False, debug=False, diffusion_mode="default", subject_strength=0.0, cfg=10, scale=1.0) -> Any:
        if self.conditioner is None:
            raise NotImplementedError("the LLM / ImageBind / prior stages are outside the denoise hot path (SURVEY.md §8): construct with "
                                      "conditioner=<callable returning dict(image_embeds, base_embed, y, caption, base_latents, "
                                      "prompt_embeds, pooled_prompt_embeds, negative_prompt_embeds, negative_pooled_prompt_embeds)> "
                                      "or call .denoise() with explicit conditioning")
        c = self.conditioner(inst, mm_data, use_cache=use_cache)
        self.cache = c
        if llm_only:
            return None, None, c["caption"]
        y0 = c.get("y")
        if y0 is None:                                                                         # :313-317, the prior's one live call
            if self.model is None:
                raise NotImplementedError("the conditioner returned no `y` and no prior= was attached")
            ie = c["image_embeds"]
            y = self.model.generate_diffusion(MODALITY.VIDEO, MODALITY.IMAGE, ie / ie.norm() * 100, device="cpu", no_diffusion=True,
                                              num_inference_steps=25, image_bind_overwrite=None, dtype=torch.float32, guidance_scale=10,
                                              force_guidence_t0=True, do_classifier_free_guidance=True, score=6.5)
            y0 = y[0].to(device=c["base_embed"].device, dtype=c["base_embed"].dtype)
        latent_la = fuse_instruction_embedding(c["base_embed"], c["image_embeds"], y0, h, norm)
        images, latent_inv = self.denoise(c["base_latents"], latent_la.reshape(1, -1)[0], prompt_embeds=c["prompt_embeds"],
                                          pooled_prompt_embeds=c["pooled_prompt_embeds"], negative_prompt_embeds=c["negative_prompt_embeds"],
                                          negative_pooled_prompt_embeds=c["negative_pooled_prompt_embeds"],
                                          inv_prompt_embeds=c.get("inv_prompt_embeds"), inv_pooled_prompt_embeds=c.get("inv_pooled_prompt_embeds"),
                                          alpha=alpha, num_inference_steps=num_inference_steps, cfg=cfg, scale=scale)
        non_refined = images
        oo = images
        if refinement > 0 and self.piperf is not None:                                         # :358-361
            _need(c, ("refiner_prompt_embeds", "refiner_pooled_prompt_embeds", "refiner_negative_prompt_embeds", "refiner_negative_pooled_prompt_embeds"),
                  "refiner pass")
            # Conditioning = text encoder 2 on caption + ',high quality,well-formed,award-winning' (the conditioner supplies its embeddings).
            kw = dict(strength=refinement, prompt_embeds=c["refiner_prompt_embeds"], pooled_prompt_embeds=c["refiner_pooled_prompt_embeds"],
                      negative_prompt_embeds=c["refiner_negative_prompt_embeds"], negative_pooled_prompt_embeds=c["refiner_negative_pooled_prompt_embeds"],
                      noise=c.get("refiner_noise"), output_type="latent")
            vae_route = self.refiner_handoff == "image" and self.pipe._vae_decode is not None and self.piperf._vae_encode is not None
            if self.refiner_handoff == "image" and not vae_route and not self._warned_handoff:
                # the reference's route was asked for (decode, 8-bit image, posterior SAMPLE from the global RNG) but cannot be taken: say so once --
                # the latent route has different numerics and draws nothing from the RNG
                import warnings
                warnings.warn("refiner_handoff='image' needs vae_decode= on the base pipeline and vae_encode= on the refiner pipeline; handing the "
                              "latents over directly instead (pass refiner_handoff='latent' to choose this route explicitly)", RuntimeWarning, stacklevel=2)
                self._warned_handoff = True
            if vae_route:
                # the reference hands a decoded 8-bit image over and the refiner pipeline re-encodes it with the shared VAE
                # (`retrieve_latents(vae.encode(image)) * scaling_factor`: a posterior SAMPLE, global RNG)
                oo = self.piperf(image=to_8bit_image(self.pipe._vae_decode(images)), **kw).images
            else:
                oo = self.piperf(latents=images, **kw).images
        subject_data = c.get("subject_data") or []
        if subject_strength > 0 and len(subject_data) > 0:                                     # :363-368 (masks: SAM / GroundingDINO, off-path)
            _need(c, ("subject_prompt_embeds", "subject_pooled_prompt_embeds", "subject_negative_prompt_embeds", "subject_negative_pooled_prompt_embeds"),
                  "subject-consistency pass")
            oo = subject_consistency(subject_data, oo, self.ip_adapter_xl_inpaint, subject_strength, output_type="latent",
                                     prompt_embeds=c["subject_prompt_embeds"], pooled_prompt_embeds=c["subject_pooled_prompt_embeds"],
                                     negative_prompt_embeds=c["subject_negative_prompt_embeds"],
                                     negative_pooled_prompt_embeds=c["subject_negative_pooled_prompt_embeds"], noise=c.get("subject_noise"))
        msg = "SUCCESS!" if not debug else dict(output_caption=c["caption"], latent_inv=latent_inv, latent_la=latent_la)
        return non_refined, oo, msg
