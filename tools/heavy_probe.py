"""Diagnostic (not a test): the full-size UNet on the "heavy" synthetic weights (outlier channels of a few hundred in the residual stream), HIP path against the
fp32 oracle, with the LayerNorm folded into the consuming GEMM (default) and with the explicit fp16 LayerNorm kernel (IA2P_LN_FOLD=0): is the fold what costs
precision on real-weight-shaped activations, or is it the fp16 activation storage both share?  usage: python tools/heavy_probe.py [scale=100]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from bench import make_inputs
from instructany2pix_amd import weights as Wm
from instructany2pix_amd.config import sdxl_base
from instructany2pix_amd.unet import HipUNet2DConditionModel

if len(sys.argv) > 1:
    Wm.HEAVY_SCALE = float(sys.argv[1])
DEV = "cuda:0"
cfg = sdxl_base()
us, ips = Wm.unet_param_specs(cfg), Wm.ip_adapter_specs(cfg)["ip_adapter"]
ids = [3, 23, 43]                 # three input seeds x 8 requests
inputs = [make_inputs(cfg, 8, 64, 81, DEV, cfg_id=i) for i in ids]
outs = {}
for fold in ("1", "0"):
    os.environ["IA2P_LN_FOLD"] = fold
    hip = HipUNet2DConditionModel(cfg, DEV)
    hip.load_state_dict(Wm.iter_synthetic(us, 7, DEV, torch.float16, recipe="heavy"))
    hip.load_ip_adapter_weights(Wm.iter_synthetic(ips, 7, DEV, torch.float16), scale=1.0, num_tokens=4)
    outs[fold] = [hip(lat, 981, encoder_hidden_states=ctx, added_cond_kwargs=dict(text_embeds=pooled, time_ids=tid))[0].float().cpu() for lat, ctx, pooled, tid in inputs]
    del hip
    torch.cuda.empty_cache()
host = lambda it: ((k, v.cpu()) for k, v in it)
torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
ref = oracle.build_unet_fast(cfg, host(Wm.iter_synthetic(us, 7, DEV, torch.float16, recipe="heavy")), host(Wm.iter_synthetic(ips, 7, DEV, torch.float16)), ip_scale=1.0)
seen = {}
h = ref.mid_block.attentions[0].transformer_blocks[-1].register_forward_hook(lambda m, i, o: seen.update(x=(o[0] if isinstance(o, tuple) else o).detach()))      # (returns None: a hook's return value would REPLACE the output)
rl = lambda a, b: float((a - b).norm() / b.norm())
errs = {"1": [], "0": []}
with torch.no_grad():
    for n, (lat, ctx, pooled, tid) in enumerate(inputs):
        want = ref(lat.float().cpu(), 981, ctx.float().cpu(), added_cond_kwargs=dict(text_embeds=pooled.float().cpu(), time_ids=tid.float().cpu()))[0]
        for fold in ("1", "0"):
            errs[fold] += [rl(outs[fold][n][r], want[r]) for r in range(8)]
h.remove()
x = seen["x"].flatten(0, -2)
rms = x.pow(2).mean(0).sqrt()
print(f"scale {Wm.HEAVY_SCALE}: mid-block stream channel rms median {float(rms.median()):.2f} max {float(rms.max()):.1f} |x| max {float(x.abs().max()):.1f}; row |mean|/std max {float((x.mean(1).abs() / x.std(1)).max()):.3f}")
for fold in ("1", "0"):
    e = torch.tensor(errs[fold])
    print(f"IA2P_LN_FOLD={fold}: rel-L2 over {len(e)} requests: mean {float(e.mean()):.5f} median {float(e.median()):.5f} max {float(e.max()):.5f}", [round(float(v), 4) for v in e])
worse = sum(a > b for a, b in zip(errs["1"], errs["0"]))
print(f"fold worse than explicit on {worse} of {len(errs['1'])} requests")
