"""Full-size parity on REAL-WEIGHT-SHAPED activations (the stand-in for the SDXL checkpoint this environment does not have): the "heavy" synthetic recipe
(instructany2pix_amd/weights.py::_make) scales 1 % of the output channels of every residual branch's last layer (attn to_out.0, ff.net.2, resnet conv2,
proj_out; the same channels in every layer) x 60, so the residual stream carries outlier channels of a few hundred (measured: channel rms up to ~120 against a
median of 1.5, |x| up to ~290) -- what the folded LayerNorm (rstd * (acc - mean * colsum) from
{sum x, sum x^2}), the GEGLU gate table (|g| > 8 clamps) and the fp16 activation storage have to survive. One cfg-3 evaluation (BASELINE configs[2],
all 8 requests) against the fp32 CPU oracle on the same weights, at the bound of the tame-weights test (rel-L2 <= 5e-3, max|d| <= 2e-2 max|ref|),
finite everywhere. What the recipe costs (tools/heavy_probe.py, 24 requests): rel-L2 0.6e-3 on the tame weights, 1.6e-3 mean / 2.9e-3 max at x 30,
and at x 100 (|x| up to 436) isolated requests reach 1e-2 ... 3e-2 on BOTH the folded and the explicit-LayerNorm path (IA2P_LN_FOLD=0) -- fp16 activation storage,
not the fold, which is 5 % (x 30) to 25 % (x 100) behind the explicit path in mean error. Reference semantics: diffusers BasicTransformerBlock / ResnetBlock2D behind instructany2pix/ddim/pnp_pipeline.py:253-260.
(A module of its own: the 11.7 GB fp32 oracle of tests/test_fullsize_gpu.py is released before this one is built.)"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / b.norm())


def test_cfg3_heavy_tailed_weights_vs_oracle():
    import oracle
    from bench import make_inputs
    from instructany2pix_amd.config import sdxl_base
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, iter_synthetic
    cfg = sdxl_base()
    us, ips = unet_param_specs(cfg), ip_adapter_specs(cfg)["ip_adapter"]
    hip = HipUNet2DConditionModel(cfg, DEV)
    hip.load_state_dict(iter_synthetic(us, 7, DEV, torch.float16, recipe="heavy"))
    hip.load_ip_adapter_weights(iter_synthetic(ips, 7, DEV, torch.float16), scale=1.0, num_tokens=4)
    lat, ctx, pooled, tid = make_inputs(cfg, 8, 64, 81, DEV, cfg_id=3)
    outs = {}
    for t in (981, 1):
        outs[t] = hip(lat, t, encoder_hidden_states=ctx, added_cond_kwargs=dict(text_embeds=pooled, time_ids=tid))[0].clone()
    torch.cuda.synchronize()
    assert all(torch.isfinite(o).all() for o in outs.values())
    host = lambda it: ((k, v.cpu()) for k, v in it)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    ref = oracle.build_unet_fast(cfg, host(iter_synthetic(us, 7, DEV, torch.float16, recipe="heavy")), host(iter_synthetic(ips, 7, DEV, torch.float16)), ip_scale=1.0)
    # the recipe does what it says: the oracle's residual stream really carries outliers (hook on the last transformer of the mid block)
    seen = {}
    blk = ref.mid_block.attentions[0].transformer_blocks[-1]
    h = blk.register_forward_hook(lambda m, i, o: seen.update(x=(o[0] if isinstance(o, tuple) else o).detach()))      # (returns None: a hook's return value would REPLACE the output)
    with torch.no_grad():
        want = ref(lat.float().cpu(), 981, ctx.float().cpu(), added_cond_kwargs=dict(text_embeds=pooled.float().cpu(), time_ids=tid.float().cpu()))[0]
    h.remove()
    x = seen["x"].flatten(0, -2)
    ch_rms = x.pow(2).mean(0).sqrt()
    print(f"[heavy] mid-block residual stream: channel rms median {float(ch_rms.median()):.2f}, max {float(ch_rms.max()):.1f}, |x| max {float(x.abs().max()):.1f}")
    assert float(ch_rms.max() / ch_rms.median()) > 30.0 and float(x.abs().max()) > 200.0       # outlier channels >> typical ones: a few hundred
    print("[heavy] rel-L2 per request, t = 981:", [round(rel_l2(outs[981][r], want[r]), 5) for r in range(8)])
    for r in range(8):
        e = rel_l2(outs[981][r], want[r])
        assert e < 5e-3, (r, e)
        assert float((outs[981][r].float().cpu() - want[r]).abs().max()) < 2e-2 * float(want[r].abs().max()), r
    with torch.no_grad():
        want1 = ref(lat[:2].float().cpu(), 1, ctx[:2].float().cpu(), added_cond_kwargs=dict(text_embeds=pooled[:2].float().cpu(), time_ids=tid[:2].float().cpu()))[0]
    for r in range(2):
        assert rel_l2(outs[1][r], want1[r]) < 5e-3, r
