"""Does replaying the UNet forward as a HIP graph (stream capture of the ~820 launches) lower the per-launch floor? Fixed timestep, cfg 3 shapes."""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd.config import sdxl_base
from instructany2pix_amd.unet import HipUNet2DConditionModel, import_plans
from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, iter_synthetic, synthetic_state_dict

DEV = "cuda:0"
cfg = sdxl_base()
unet = HipUNet2DConditionModel(cfg, DEV)
unet.load_state_dict(iter_synthetic(unet_param_specs(cfg), 7, DEV, torch.float16))
unet.load_ip_adapter_weights(synthetic_state_dict(ip_adapter_specs(cfg, 1024)["ip_adapter"], seed=7), scale=1.0, num_tokens=4)
if len(sys.argv) > 1:
    import_plans(open(sys.argv[1]).read().strip())
B = int(os.environ.get("B", 8))
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 4, 64, 64, generator=g).half().to(DEV)
ctx = torch.randn(B, 81, 2048, generator=g).half().to(DEV)
added = dict(text_embeds=torch.randn(B, 1280, generator=g).half().to(DEV), time_ids=torch.tensor([[512.0, 512, 0, 0, 512, 512]] * B).half().to(DEV))
out = torch.empty_like(x)
unet.cache_context_kv = False
run = lambda: unet(x, 501.0, encoder_hidden_states=ctx, added_cond_kwargs=added, return_dict=False, out=out)
for _ in range(5):
    run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    run()
torch.cuda.synchronize()
print(f"eager (C++ executor issuing launches): {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per forward")
ref = out.clone()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    run()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        run()
torch.cuda.synchronize()
for _ in range(5):
    graph.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    graph.replay()
torch.cuda.synchronize()
print(f"HIP graph replay: {(time.perf_counter() - t0) / 30 * 1e3:.3f} ms per forward; same bits: {torch.equal(out, ref)}")
