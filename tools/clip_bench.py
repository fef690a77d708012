"""Time the two SDXL text towers (full size, synthetic weights) through ia2p_clip_encode: what `encode_prompt` costs per request."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd.clip import HipCLIPTextModel
from instructany2pix_amd.config import sdxl_text_encoder, sdxl_text_encoder_2
from instructany2pix_amd.weights import clip_param_specs, iter_synthetic

for name, cfg in (("text_encoder (CLIP-L, 12x768)", sdxl_text_encoder()), ("text_encoder_2 (bigG, 32x1280)", sdxl_text_encoder_2())):
    m = HipCLIPTextModel(cfg, "cuda:0")
    m.load_state_dict(iter_synthetic(clip_param_specs(cfg), 7, "cuda:0", torch.float16))
    for B in (2, 16):
        ids = torch.randint(3, cfg.vocab_size - 1, (B, 77))
        for _ in range(3):
            m(ids, output_hidden_states=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m(ids, output_hidden_states=True)
        torch.cuda.synchronize()
        print(f"{name:34s} B={B:2d}: {(time.perf_counter() - t0) / 20 * 1e3:7.3f} ms per call")
