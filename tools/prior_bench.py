"""Time the embedding prior's one live call (reference pipeline.py:313-317; full size: gpt2-medium + CLIP ViT-H text tower, synthetic
weights) on the HIP path, and the fp32 CPU oracle of the same call on this box's host cores (the reference runs this stage on the CPU)."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import oracle
from stub_tokenizer import StubTokenizer
from instructany2pix_amd.config import gpt2_medium, laion_clip_h_text
from instructany2pix_amd.prior import InstructAny2PixPrior, prior_config
from instructany2pix_amd.weights import prior_param_specs, synthetic_state_dict

gcfg, ccfg = gpt2_medium(), laion_clip_h_text()
sd = synthetic_state_dict(prior_param_specs(gcfg, ccfg), seed=47)
tok = StubTokenizer(5, ccfg.vocab_size)
hip = InstructAny2PixPrior(**prior_config, device="cuda:0", tokenizer=tok)
hip.load_state_dict(sd)
emb = torch.randn(1, 1024, generator=torch.Generator().manual_seed(3))
src = emb / emb.norm() * 100
kw = dict(no_diffusion=True, num_inference_steps=25, guidance_scale=10, force_guidence_t0=True, do_classifier_free_guidance=True, score=6.5)
for _ in range(3):
    hip.generate_diffusion(3, 0, src, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    y, _ = hip.generate_diffusion(3, 0, src, **kw)
torch.cuda.synchronize()
print(f"HIP prior, live call (CLIP-H text on 2x2 tokens + GPT-2 medium on 2x11 tokens + sampler update): {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")
x = torch.randn(2, 11, 1024, device="cuda:0", dtype=torch.float16)
for _ in range(3):
    hip.model(inputs_embeds=x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    hip.model(inputs_embeds=x)
torch.cuda.synchronize()
print(f"  GPT-2 medium stack alone (2x11 tokens): {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
clip = oracle.build_clip(ccfg, {k[len("cond_stage_models.0.model."):]: v for k, v in sd.items() if k.startswith("cond_stage_models.0.model.")})
ref = oracle.PriorRef(gcfg, sd, lambda p: [clip(tok(p, max_length=77, padding=True, truncation=True).input_ids)[1], torch.ones(len(p), 2)])
ref.generate_diffusion(3, 0, src, **kw)
t0 = time.perf_counter()
for _ in range(3):
    yo, _ = ref.generate_diffusion(3, 0, src, **kw)
print(f"CPU oracle (torch fp32, {torch.get_num_threads()} threads), same call: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
print("cosine(HIP, oracle) =", float(torch.nn.functional.cosine_similarity(y.float().cpu().flatten(), yo.flatten(), dim=0)))
