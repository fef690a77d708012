#!/usr/bin/env python3
"""A/B: the headline step (B_eff = 8) as ONE launch stream over the whole batch vs TWO concurrent streams over half the batch each (two engine contexts
bound to the SAME weight arena, each with its own workspace and counter pools). Question: do two independent kernel queues fill each other's
prologue / epilogue bubbles (the step is ~600 dependent launches whose tiles spend ~35 % of their time outside the MFMA loop), or do the doubled
launch count, the second pass over the weights and the halved M per launch cost more? Prints both ms per 8-image step. Not part of the product path."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--skew", type=int, default=0, help="start the second stream this many steps late (de-phases the two queues)")
    args = ap.parse_args()
    import torch
    import bench
    from instructany2pix_amd import _ffi
    from instructany2pix_amd.config import sdxl_base
    from instructany2pix_amd.unet import HipUNet2DConditionModel
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, iter_synthetic

    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    cfg = sdxl_base()
    u1 = HipUNet2DConditionModel(cfg, dev)
    u1.load_state_dict(iter_synthetic(unet_param_specs(cfg), 7, dev, torch.float16))
    u1.load_ip_adapter_weights(iter_synthetic(ip_adapter_specs(cfg)["ip_adapter"], 7, dev, torch.float16), scale=1.0, num_tokens=4)
    torch.cuda.synchronize()
    u2 = HipUNet2DConditionModel(cfg, dev)
    u2.arena = u1.arena                                   # same weights, second context
    _ffi.check(u2._lib.ia2p_bind_arena(u2._ctx, _ffi.ptr(u2.arena), u2.arena.numel()), u2._ctx)
    u2.adopt_arena(True)
    u2.load_ip_adapter_weights([], scale=1.0, num_tokens=4)
    torch.cuda.synchronize()
    for u in (u1, u2):
        u.cache_context_kv = False

    def ms_per_step(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    B = args.batch
    one = bench.Workload(u1, cfg, B, 64, 81, 0.0, dev, cfg_id=3)
    u1.autotune(one.lat, one.ts[0], one.ctx, one.added)
    one.run(3)
    a = [ms_per_step(one.run, args.steps) for _ in range(3)]
    print(f"one stream,  B={B}:            {min(a):.3f} ms/step (runs {', '.join('%.3f' % v for v in a)})", flush=True)

    s0 = torch.cuda.Stream(dev)
    s0.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s0):
        one.run(3)
        a2 = [ms_per_step(one.run, args.steps) for _ in range(3)]
    print(f"one stream (a created stream instead of the default one), B={B}: {min(a2):.3f} ms/step", flush=True)
    torch.cuda.synchronize()

    h = B // 2
    wa, wb = bench.Workload(u1, cfg, h, 64, 81, 0.0, dev, cfg_id=3), bench.Workload(u2, cfg, h, 64, 81, 0.0, dev, cfg_id=3)
    u1.autotune(wa.lat, wa.ts[0], wa.ctx, wa.added)
    wa.run(3)
    b1 = [ms_per_step(wa.run, args.steps) for _ in range(3)]
    print(f"one stream,  B={h}:            {min(b1):.3f} ms/step -> x2 = {2 * min(b1):.3f} ms per {B} images", flush=True)
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    sa.wait_stream(torch.cuda.current_stream())
    sb.wait_stream(torch.cuda.current_stream())

    def both(n):
        for i in range(n + args.skew):
            if i < n:
                with torch.cuda.stream(sa):
                    wa.step()
            if i >= args.skew:
                with torch.cuda.stream(sb):
                    wb.step()

    both(3)
    c = [ms_per_step(both, args.steps) for _ in range(3)]
    print(f"two streams, B={h} each:       {min(c):.3f} ms per {B}-image step (runs {', '.join('%.3f' % v for v in c)})", flush=True)
    assert torch.isfinite(wa.x).all() and torch.isfinite(wb.x).all()


if __name__ == "__main__":
    main()
