"""Experiment: does running two half-batches on two HIP streams (two contexts sharing one weight arena) beat one full batch?"""
import os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd.config import sdxl_base
from instructany2pix_amd.unet import HipUNet2DConditionModel
from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, iter_synthetic
from instructany2pix_amd import _ffi
import bench

dev = torch.device("cuda:0"); cfg = sdxl_base()
u0 = HipUNet2DConditionModel(cfg, dev)
u0.load_state_dict(iter_synthetic(unet_param_specs(cfg), 7, dev, torch.float16))
u0.load_ip_adapter_weights(iter_synthetic(ip_adapter_specs(cfg)["ip_adapter"], 7, dev, torch.float16), 1.0, 4)
u1 = HipUNet2DConditionModel(cfg, dev)
_ffi.check(u1._lib.ia2p_bind_arena(u1._ctx, _ffi.ptr(u0.arena), u0.arena.numel()), u1._ctx)   # share the weights
u1.arena = u0.arena
u1.adopt_arena(); u1.load_ip_adapter_weights([], 1.0, 4)
lat, ctx, pooled, tid = bench.make_inputs(cfg, 8, 64, 81, dev)

def run(unet, sl, stream, steps):
    with torch.cuda.stream(stream):
        x = lat[sl].contiguous(); c = ctx[sl].contiguous(); p = pooled[sl].contiguous(); t = tid[sl].contiguous()
        out = torch.empty_like(x)
        for i in range(steps):
            unet(x, 500, encoder_hidden_states=c, added_cond_kwargs=dict(text_embeds=p, time_ids=t), out=out)

def timed(fn):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0

steps = 20
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
one = timed(lambda: run(u0, slice(0, 8), s0, steps))
def two():
    th = [threading.Thread(target=run, args=(u0, slice(0, 4), s0, steps)), threading.Thread(target=run, args=(u1, slice(4, 8), s1, steps))]
    [t.start() for t in th]; [t.join() for t in th]
both = timed(two)
half = timed(lambda: run(u0, slice(0, 4), s0, steps))
print(f"one stream B=8: {one/steps*1e3:.2f} ms/step | two streams 2xB=4: {both/steps*1e3:.2f} ms/step | one stream B=4 alone: {half/steps*1e3:.2f} ms/step")
