"""End-to-end latency of ONE edit request at the reference's own defaults, every model at full size with seeded synthetic weights:
1024x1024 image, num_inference_steps=25, cfg=10, refinement=0.5 (reference pipeline.py:303-386). Everything downstream of the LLM / ImageBind
stage runs on the HIP path: VAE encode -> embedding prior (CLIP ViT-H text + GPT-2 medium) -> encode_prompt (CLIP-L + bigG) -> 25-step DDIM
inversion (B=1) -> polar mixing -> 25-step IP-Adapter guided CFG sampling (B_eff=2) -> SDXL-refiner img2img (strength 0.5 of 50 steps, CFG)
-> VAE decode. The LLM / ImageBind outputs (1024-d embeddings, caption) are stand-ins: that stage is out of scope.
Prints per-stage wall times (stream-synchronised) after one warm-up request."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from stub_tokenizer import StubTokenizer
from instructany2pix_amd.clip import HipCLIPTextModel, SDXLTextEncoders
from instructany2pix_amd.config import sdxl_base, sdxl_refiner, sdxl_vae, sdxl_text_encoder, sdxl_text_encoder_2
from instructany2pix_amd.pipeline import InstructAny2PixPipeline
from instructany2pix_amd.prior import InstructAny2PixPrior, prior_config
from instructany2pix_amd.unet import HipUNet2DConditionModel
from instructany2pix_amd.vae import HipAutoencoderKL
from instructany2pix_amd.weights import (unet_param_specs, ip_adapter_specs, vae_param_specs, clip_param_specs, prior_param_specs,
                                         iter_synthetic, synthetic_state_dict)
from instructany2pix_amd.config import gpt2_medium, laion_clip_h_text

DEV = "cuda:0"
PX = int(os.environ.get("PX", 1024))
t_all = time.perf_counter()
bcfg, rcfg, vcfg = sdxl_base(), sdxl_refiner(), sdxl_vae()
base = HipUNet2DConditionModel(bcfg, DEV); base.load_state_dict(iter_synthetic(unet_param_specs(bcfg), 7, DEV, torch.float16))
ref = HipUNet2DConditionModel(rcfg, DEV); ref.load_state_dict(iter_synthetic(unet_param_specs(rcfg), 11, DEV, torch.float16))
vae = HipAutoencoderKL(vcfg, DEV); vae.load_state_dict(iter_synthetic(vae_param_specs(vcfg), 5, DEV, torch.float16))
c1, c2 = sdxl_text_encoder(), sdxl_text_encoder_2()
te1 = HipCLIPTextModel(c1, DEV); te1.load_state_dict(iter_synthetic(clip_param_specs(c1), 7, DEV, torch.float16))
te2 = HipCLIPTextModel(c2, DEV); te2.load_state_dict(iter_synthetic(clip_param_specs(c2), 8, DEV, torch.float16))
enc = SDXLTextEncoders(StubTokenizer(1, c1.vocab_size), StubTokenizer(2, c2.vocab_size), te1, te2)
enc_ref = SDXLTextEncoders(None, StubTokenizer(2, c2.vocab_size), None, te2)           # the refiner checkpoint has text encoder 2 only
prior = InstructAny2PixPrior(**prior_config, device=DEV, tokenizer=StubTokenizer(5, laion_clip_h_text().vocab_size))
prior.load_state_dict(synthetic_state_dict(prior_param_specs(gpt2_medium(), laion_clip_h_text()), seed=47))
specs = ip_adapter_specs(bcfg, 1024)
ck = {"image_proj": synthetic_state_dict(specs["image_proj"], seed=7), "ip_adapter": synthetic_state_dict(specs["ip_adapter"], seed=7)}
print(f"models ready in {time.perf_counter() - t_all:.1f} s (base UNet + IP-Adapter, refiner UNet, VAE, CLIP-L, bigG, prior)", flush=True)

stages = {}


def timed(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(); stages[name] = stages.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
    return out


g = torch.Generator().manual_seed(1)
image = (torch.rand(1, 3, PX, PX, generator=g) * 2 - 1).half().to(DEV)
image_embeds, base_embed = torch.randn(1, 1024, generator=g), torch.randn(1, 1024, generator=g)      # stand-ins for the LLM / ImageBind stage
caption = "a watercolor painting of a fox in the snow"


def conditioner(inst, mm, use_cache=False):
    lat = timed("vae_encode", lambda: vae.encode_to_latents(image, torch.Generator().manual_seed(2)))
    pe, ne, pp, npl = timed("encode_prompt(base)", lambda: enc.encode_prompt(prompt=caption, negative_prompt="", do_classifier_free_guidance=True))
    ipe, _, ipp, _ = timed("encode_prompt(inversion '')", lambda: enc.encode_prompt(prompt="", do_classifier_free_guidance=False))
    rpe, rne, rpp, rnp = timed("encode_prompt(refiner)", lambda: enc_ref.encode_prompt(prompt=caption + ",high quality,well-formed,award-winning", negative_prompt="", do_classifier_free_guidance=True))
    return dict(image_embeds=image_embeds, base_embed=base_embed, caption=caption, base_latents=lat, prompt_embeds=pe, pooled_prompt_embeds=pp,
                negative_prompt_embeds=ne, negative_pooled_prompt_embeds=npl, inv_prompt_embeds=ipe, inv_pooled_prompt_embeds=ipp,
                refiner_prompt_embeds=rpe, refiner_pooled_prompt_embeds=rpp, refiner_negative_prompt_embeds=rne, refiner_negative_pooled_prompt_embeds=rnp)


pipe = InstructAny2PixPipeline(unet=base, ip_ckpt=ck, device=DEV, clip_embeddings_dim=1024, conditioner=conditioner, refiner_unet=ref, prior=prior,
                               vae_encode=vae.encode_to_latents, vae_decode=vae.decode_from_latents)
if os.environ.get("TUNE", "1") == "1":      # measure kernel plans for the three UNet shapes of a request (what bench.py does for its shape)
    t0 = time.perf_counter()
    h = PX // 8
    for net, B, L, cd, pd, nid in ((base, 1, 77, bcfg.cross_attention_dim, bcfg.pooled_dim, 6), (base, 2, 81, bcfg.cross_attention_dim, bcfg.pooled_dim, 6),
                                   (ref, 2, 77, rcfg.cross_attention_dim, rcfg.pooled_dim, 5)):
        x = torch.randn(B, 4, h, h, generator=g).half().to(DEV)
        ehs = torch.randn(B, L, cd, generator=g).half().to(DEV)
        added = dict(text_embeds=torch.randn(B, pd, generator=g).half().to(DEV), time_ids=torch.tensor([[float(PX)] * 2 + [0.0] * 2 + [float(PX)] * (nid - 4)] * B).half().to(DEV))
        net.autotune(x, 500, ehs, added, reps=3)
    print(f"kernel plans measured in {time.perf_counter() - t0:.1f} s", flush=True)
# stage timers around the four loops of a request (synchronising wrappers: they add a few host round trips, nothing else)
def _wrap(obj, name, label):
    fn = getattr(obj, name)
    setattr(obj, name, lambda *a, **k: timed(label, lambda: fn(*a, **k)))


_wrap(pipe.pipe_inversion, "inverse", "inversion loop (25 x B=1)")
_wrap(pipe.ip_adapter_xl, "generate", "guided sampling loop (25 x B_eff=2, incl. image-token projection)")
if getattr(pipe, "model", None) is not None:
    _wrap(pipe.model, "generate_diffusion", "embedding prior")
_piperf_call = pipe.piperf.__call__
pipe.piperf = type("TimedRefiner", (), {"__call__": lambda self, *a, **k: timed("refiner pass (incl. VAE hand-off encode)", lambda: _piperf_call(*a, **k)),
                                        "__getattr__": lambda self, n: getattr(_piperf_call.__self__, n)})()
for rnd in range(2):            # request 0 warms up (workspaces, kernel plans from the cost model), request 1 is reported
    stages.clear()
    torch.manual_seed(3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    non_refined, refined, msg = pipe("turn the fox blue", [], num_inference_steps=25, cfg=10, refinement=0.5)
    out = timed("vae_decode", lambda: vae.decode_from_latents(refined))
    torch.cuda.synchronize(); total = (time.perf_counter() - t0) * 1e3
    assert msg == "SUCCESS!" and torch.isfinite(out.float()).all() and tuple(out.shape) == (1, 3, PX, PX)
    print(f"request {rnd}: {total:.0f} ms total; stages (ms): " + ", ".join(f"{k} {v:.1f}" for k, v in stages.items())
          + f", host glue / the rest {total - sum(stages.values()):.0f}", flush=True)
