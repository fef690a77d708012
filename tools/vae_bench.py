"""VAE encode / decode timing on the SDXL VAE architecture (synthetic weights), images resident in HBM."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd.config import sdxl_vae
from instructany2pix_amd.vae import HipAutoencoderKL
from instructany2pix_amd.weights import vae_param_specs, iter_synthetic

cfg = sdxl_vae(); dev = torch.device("cuda:0")
vae = HipAutoencoderKL(cfg, dev)
vae.load_state_dict(iter_synthetic(vae_param_specs(cfg), 7, dev, torch.float16))
for B, h in ((8, 64), (1, 64), (1, 128)):
    z = torch.randn(B, 4, h, h, device=dev).half()
    img = vae.decode(z, return_dict=False)[0]; torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): img = vae.decode(z, return_dict=False)[0]
    torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 3
    mom = vae.encode(img).latent_dist.parameters; torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): mom = vae.encode(img).latent_dist.parameters
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 3
    print(f"B={B} image {h*8}x{h*8}: decode {td*1e3:8.2f} ms  encode {te*1e3:8.2f} ms  finite={bool(torch.isfinite(img).all())} ws={vae._ws.numel()/1e9:.2f} GB")
