#!/usr/bin/env python3
"""Headline benchmark of the denoise hot path: denoise-steps/sec (512x512, 50-step DDIM, batch 8).

  python bench.py --gpus 1 --steps 50 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: conditional-UNet evaluation (SDXL-base architecture,
B_eff = 8 latents of 64x64x4, 81-token context = 77 text + 4 IP-Adapter image tokens) + the fused DDIM
update, i.e. BASELINE.json configs[2]. Inputs and weights are synthetic (seeded) and resident
in HBM before the timed region. With N > 1 every rank runs its own batch of 8 (weak scaling, no per-step
communication; rank 0's weight arena is RCCL-broadcast once, outside the timed region).

Rank 0 prints ONE JSON line; besides the contract fields it carries
  "roofline":     the kernel with the largest share of step time, timed live with HIP events on the launch
                  stream (ia2p_profile_*), algorithmic FLOPs (2*M*N*K) per launch / average launch duration
  "cpu_baseline": the CPU oracle (oracle/, torch fp32 on the host cores) timed on a bounded sample.
"""
import argparse
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0       # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(cfg, B, hw, L, device, cfg_id=3):
    import torch
    def g(seed):
        return torch.Generator("cpu").manual_seed(seed)
    lat = torch.randn(B, 4, hw, hw, generator=g(1000 + cfg_id)).half()
    text = torch.randn(B, 77, cfg.cross_attention_dim, generator=g(2000 + cfg_id))
    ip = torch.nn.functional.layer_norm(torch.randn(B, L - 77, cfg.cross_attention_dim, generator=g(3000 + cfg_id)), (cfg.cross_attention_dim,)) if L > 77 else text[:, :0]
    ctx = torch.cat([text, ip], dim=1).half()
    pooled = torch.randn(B, cfg.pooled_dim, generator=g(4000 + cfg_id)).half()
    tid = torch.tensor([[hw * 8.0, hw * 8.0, 0.0, 0.0, hw * 8.0, hw * 8.0][:cfg.num_time_ids - 1] + ([hw * 8.0] if cfg.num_time_ids == 6 else [6.0])] * B).half()
    return [t.to(device).contiguous() for t in (lat, ctx, pooled, tid)]


def usable_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota, not the host's core count."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, unet_specs, ip_specs, seed, dev, inputs_cpu, L, sample_batch):
    """Oracle UNet step on the host cores: 1 warm-up + 1 timed evaluation of `sample_batch` requests.
    Weights are the very tensors the HIP path was loaded with (same device generator), copied to the host."""
    import torch
    import oracle
    from instructany2pix_amd.weights import iter_synthetic
    cores = usable_cores()
    torch.set_num_threads(cores)
    t0 = time.time()
    host = lambda it: ((k, v.cpu()) for k, v in it)
    net = oracle.build_unet_fast(cfg, host(iter_synthetic(unet_specs, seed, dev, torch.float16)), host(iter_synthetic(ip_specs, seed, dev, torch.float16)))
    log(f"[cpu_baseline] oracle built in {time.time() - t0:.1f}s, {cores} threads")
    lat, ctx, pooled, tid = [t[:sample_batch].float() for t in inputs_cpu]
    added = dict(text_embeds=pooled, time_ids=tid)
    sch = oracle.DDIMSchedulerRef()
    sch.set_timesteps(50)
    times = []
    with torch.no_grad():
        for i in range(2):
            t = int(sch.timesteps[i])
            t0 = time.time()
            eps = net(lat, t, ctx, added_cond_kwargs=added)[0]
            lat = sch.step(eps, t, lat)
            times.append(time.time() - t0)
    log(f"[cpu_baseline] step times {times}")
    return times[-1], cores


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--latent", type=int, default=64, help="latent side (64 = 512x512 pixels)")
    ap.add_argument("--ctx", type=int, default=81, help="context tokens (77 text + 4 image tokens)")
    ap.add_argument("--unet", choices=["base", "refiner"], default="base", help="refiner = the second engine config (non-headline; use --ctx 77)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-autotune", action="store_true", help="use the built-in cost model instead of measured kernel plans")
    ap.add_argument("--plans", default=None, help="import this kernel plan table instead of measuring (profiler runs: keeps the tuning launches out of the trace)")
    ap.add_argument("--save-plans", default=None, help="write the kernel plan table in use to this file")
    ap.add_argument("--cpu-sample-batch", type=int, default=1)
    ap.add_argument("--kernel-table", default=None, help="write the per-kernel timing table (JSON) to this file")
    args = ap.parse_args()

    import torch
    from instructany2pix_amd import dist as D
    from instructany2pix_amd.config import sdxl_base, sdxl_refiner
    from instructany2pix_amd.scheduler import DDIMScheduler, fused_update
    from instructany2pix_amd.unet import HipUNet2DConditionModel, export_plans, import_plans
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, iter_synthetic

    rank, world, local = D.init_distributed()
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}")
    dev = torch.device(f"cuda:{os.environ.get('IA2P_FORCE_DEVICE', local)}")     # (override: tests with several ranks on one GPU)
    torch.cuda.set_device(dev)
    cfg = sdxl_base() if args.unet == "base" else sdxl_refiner()
    seed = 7
    unet_specs, ip_specs = unet_param_specs(cfg), ip_adapter_specs(cfg)["ip_adapter"]
    use_ip = args.ctx > 77

    t0 = time.time()
    unet = HipUNet2DConditionModel(cfg, dev)
    if rank == 0:   # weights are generated once (seeded, on the device for speed) and broadcast
        unet.load_state_dict(iter_synthetic(unet_specs, seed, dev, torch.float16))
        if use_ip:
            unet.load_ip_adapter_weights(iter_synthetic(ip_specs, seed, dev, torch.float16), scale=1.0, num_tokens=args.ctx - 77)
    D.broadcast_weights(unet, src=0)
    if rank != 0 and use_ip:
        unet.load_ip_adapter_weights([], scale=1.0, num_tokens=args.ctx - 77)
    torch.cuda.synchronize()
    log(f"[rank {rank}] weights ready in {time.time() - t0:.1f}s (arena {unet.arena.numel() / 1e9:.2f} GB)")

    B, hw, L = args.batch, args.latent, args.ctx
    lat, ctx, pooled, tid = make_inputs(cfg, B, hw, L, dev)
    added = dict(text_embeds=pooled, time_ids=tid)
    sch = DDIMScheduler()
    sch.set_timesteps(50)
    ts = [int(t) for t in sch.timesteps]
    eps, nxt = torch.empty_like(lat), torch.empty_like(lat)

    def step(i, x, y):
        if i % len(ts) == 0:          # a new request every 50 steps: start again from the seeded latents (keeps any --steps finite)
            x.copy_(lat)
        t = ts[i % len(ts)]
        unet(x, t, encoder_hidden_states=ctx, added_cond_kwargs=added, out=eps)
        c_x, c_e = sch.step_coeffs(t)
        fused_update(x, eps, None, 1.0, c_x, c_e, y)

    # set-up, outside the timed region: measure the candidate (tile, K-split) plans of every GEMM / conv shape once on
    # rank 0 (ia2p_autotune) and hand the table to the other ranks so that all ranks run identical kernels
    plans = "cost model"
    if args.plans:
        text = open(args.plans).read().strip()
        import_plans(text)
        plans = f"imported from {os.path.basename(args.plans)} ({text.count(';')} shapes)"
    elif not args.no_autotune:
        t0 = time.time()
        table = [None]
        if rank == 0:
            n = unet.autotune(lat, ts[0], ctx, added)
            table[0] = export_plans()
            log(f"[rank 0] autotune: {n} GEMM/conv shapes measured in {time.time() - t0:.1f}s")
            if os.environ.get("IA2P_PRINT_PLANS"):
                log("[rank 0] plans: " + table[0])
        if world > 1:
            torch.distributed.broadcast_object_list(table, src=0)
            if rank != 0:
                import_plans(table[0])
        plans = f"measured in place at start-up ({table[0].count(';')} shapes)"
    if args.save_plans and rank == 0:
        with open(args.save_plans, "w") as f:
            f.write(export_plans() + "\n")

    # The headline loop evaluates the WHOLE UNet every step, as the reference does: the context K/V hoisting the pipelines use
    # (same bits, one GEMM less per step) is switched off here and reported separately below.
    unet.cache_context_kv = False
    x, y = lat.clone(), nxt
    for i in range(args.warmup):
        step(i, x, y)
        x, y = y, x
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, x, y)
        x, y = y, x
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, device=dev if world > 1 else "cpu")
    assert torch.isfinite(x).all(), "non-finite latents after the timed run"

    default_cfg = (args.batch, args.latent, args.ctx, args.unet) == (8, 64, 81, "base")
    metric = "denoise-steps/sec (512x512, 50-step DDIM, batch 8)" if default_cfg else \
        f"denoise-steps/sec ({args.latent * 8}x{args.latent * 8}, 50-step DDIM, batch {args.batch}{', SDXL-refiner UNet' if args.unet == 'refiner' else ''}) [non-headline shape]"
    res = {
        "metric": metric, "value": world * args.steps / elapsed, "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"{'BASELINE configs[2]' if default_cfg else 'custom'}: {hw * 8}x{hw * 8} px, latent [{B},4,{hw},{hw}], 50-step DDIM schedule, "
                               f"context {L} tokens ({'77 text + %d IP-Adapter image tokens' % (L - 77) if use_ip else 'text only'}), {'SDXL-base UNet (2.567 G params) + IP-Adapter' if args.unet == 'base' else 'SDXL-refiner UNet (2.260 G params)'}, "
                               f"synthetic seeded weights", "global_batch": B * world, "parallelism": f"dp{world}", "kernel_plans": plans,
                   "image_steps_per_s": world * B * args.steps / elapsed},
    }

    if rank == 0:      # secondary number, not `value`: the same loop with the request's context K/V projected once (what the pipelines do)
        unet.cache_context_kv = True
        step(0, x, y)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nk = min(20, args.steps)
        for i in range(nk):
            step(i, x, y)
            x, y = y, x
        torch.cuda.synchronize()
        res["config"]["ms_per_step_with_context_kv_hoisted"] = 1e3 * (time.perf_counter() - t1) / nk
        unet.cache_context_kv = False
    if rank == 0 and not args.no_roofline:
        unet.profile(True)
        nprof = 3
        for i in range(nprof):
            step(i, x, y)
            x, y = y, x
        torch.cuda.synchronize()
        table = unet.profile_read()
        unet.profile(False)
        tot_ms = sum(v["ms"] for v in table.values())
        dom = max(table, key=lambda k: table[k]["ms"])
        d = table[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        traffic, traffic_src = None, None
        pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
        if pmc:                     # HBM-side bytes per launch of this kernel, from separate rocprofv3 --pmc passes (tools/pmc_traffic.py)
            k = json.load(open(pmc[-1]))["kernels"].get(dom)
            if k:
                traffic, traffic_src = k["traffic_bytes_per_launch"], os.path.basename(pmc[-1])
        res["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                           "traffic": traffic, "traffic_unit": "bytes/launch (L2<->fabric, Infinity-Cache hits included)", "traffic_source": traffic_src,
                           "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
                           "launches_per_step": d["launches"] / nprof, "avg_launch_us": 1e3 * d["ms"] / d["launches"],
                           "flops_per_launch": d["flops"] / d["launches"], "share_of_step": d["ms"] / tot_ms}
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            log(f"  {k:42s} {v['launches'] / nprof:7.1f} launches/step {v['ms'] / nprof:8.3f} ms/step "
                f"{(v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] else 0:8.1f} TFLOP/s {(v['bytes'] / (v['ms'] * 1e-3) / 1e9) if v['ms'] else 0:8.1f} GB/s(alg)")
        if args.kernel_table:
            os.makedirs(os.path.dirname(os.path.abspath(args.kernel_table)), exist_ok=True)
            json.dump({"steps_profiled": nprof, "kernels": table}, open(args.kernel_table, "w"), indent=1)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sb = max(1, min(args.cpu_sample_batch, B))
        inputs_cpu = [t.cpu() for t in (lat, ctx, pooled, tid)]
        t_step, cores = cpu_baseline(cfg, unet_specs, ip_specs, seed, dev, inputs_cpu, L, sb)
        res["cpu_baseline"] = {"value": 1.0 / (t_step * B / sb), "unit": "steps/s", "cores": cores, "kind": "port",
                               "sample": f"oracle (torch fp32, {cores} threads = usable cores of this box): 1 warm-up + 1 timed UNet+DDIM step on "
                                         f"{sb} of the {B} requests of the same workload ({t_step:.2f} s), scaled x{B // sb} to the batch-{B} step"}

    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
