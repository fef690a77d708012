"""Batched, sharded serving of independent edit requests on the denoise hot path.

The reference serves one request at a time on one GPU (instructany2pix/pipeline.py:303-386; serve.py:115 serialises the queue), so every knob of
`__call__` -- `num_inference_steps`, `cfg`, `scale`, `alpha` (pipeline.py:303-304) -- is a per-call scalar. Independent requests share nothing
(GroupNorm / LayerNorm / attention are per sample), so this module runs N of them as ONE batch per UNet evaluation:

  * every batch element carries its own timestep (its own position in its own schedule), its own IP-Adapter scale and its own guidance scale:
    `ia2p_unet_forward_v` (timesteps[B], ip_scales[B]) and `ia2p_ddim_step_v` (per-request {g, c_x, c_e});
  * requests whose schedule is shorter simply stop moving (c_x = 1, c_e = 0) while the others finish;
  * across the GPUs of a node the request list is sharded contiguously (`dist.shard_range`), every rank runs its shard in groups (default 4
    requests = B_eff 8 in the guided sampling loop, the shape the headline metric is quoted on) and the results are all-gathered.

Per request the hot segment is the reference's: DDIM inversion of the base latents (pnp_pipeline.py:92-278, no guidance), polar mixing with fresh
noise on the CPU in fp16 (pipeline.py:331-337), IP-Adapter guided sampling (ip_adapter.py:289-356 -> sdxl_pipeline.py:764-857).
Batch element b of a heterogeneous batch gets the bits it gets in a uniform batch of the same size (tests/test_batch_gpu.py); against a
batch-1 run only the summation order of batch-dependent kernel plans differs (fp16 tolerance).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import dist as D
from .ddim import get_add_time_ids
from .scheduler import DDIMScheduler, fused_update_v


@dataclass
class EditRequest:
    """One edit request on explicit conditioning (what `InstructAny2PixPipeline.denoise` takes), with its own knobs."""
    base_latents: torch.Tensor                 # [1, 4, h, w] x0 of the base image
    latent_la: torch.Tensor                    # [D] fused instruction embedding (IP-Adapter input, reference pipeline.py:322-324)
    prompt_embeds: torch.Tensor                # [1, 77, ctx]
    pooled_prompt_embeds: torch.Tensor         # [1, pooled]
    negative_prompt_embeds: torch.Tensor
    negative_pooled_prompt_embeds: torch.Tensor
    inv_prompt_embeds: Optional[torch.Tensor] = None      # embedding of '' (reference :330); default: the negative ones
    inv_pooled_prompt_embeds: Optional[torch.Tensor] = None
    alpha: float = 0.7
    num_inference_steps: int = 25
    cfg: float = 10.0
    scale: float = 1.0
    noise: Optional[torch.Tensor] = None       # polar-mixing noise (CPU fp16); None: drawn from the global RNG in request order


def _coef_table(rows: Sequence[Sequence[tuple]], device) -> torch.Tensor:
    """[steps][B][3] float32 on the device: (g, c_x, c_e) of every request at every iteration of the joint loop"""
    return torch.tensor(rows, dtype=torch.float32).to(device).contiguous()


@torch.no_grad()
def invert_batch(unet, scheduler_config, latents, prompt_embeds, pooled, steps: Sequence[int], ip_scales=None):
    """DDIM inversion x0 -> xT of B requests at once (reference loop pnp_pipeline.py:251-275 per request), each over its own `steps[b]`-step
    schedule walked in ascending t. latents [B,4,h,w]; prompt_embeds [B,L,ctx]; pooled [B,P]."""
    dev = unet.device
    B = latents.shape[0]
    x = latents.to(device=dev, dtype=torch.float16).contiguous().clone()
    h, w = x.shape[-2] * 8, x.shape[-1] * 8
    tid = get_add_time_ids(unet, (h, w), (0, 0), (h, w), int(pooled.shape[-1])).repeat(B, 1).to(dev)
    ctx = prompt_embeds.to(device=dev, dtype=torch.float16).contiguous()
    added = {"text_embeds": pooled.to(device=dev, dtype=torch.float16).contiguous(), "time_ids": tid}
    nmax = max(steps)
    ts_rows, coef_rows = [[0.0] * B for _ in range(nmax)], [[(0.0, 1.0, 0.0)] * B for _ in range(nmax)]
    for b, n in enumerate(steps):
        sch = DDIMScheduler.from_config(scheduler_config)
        sch.set_timesteps(n)
        acp, prev = sch.alphas_cumprod, None
        asc = [int(t) for t in reversed(sch.timesteps)]
        for i in range(nmax):
            if i < n:
                t = asc[i]
                a_prev = float(acp[prev]) if prev is not None else float(sch.final_alpha_cumprod)      # :262-267
                c_x, c_e = DDIMScheduler.inversion_coeffs(float(acp[t]), a_prev)
                ts_rows[i][b], coef_rows[i][b], prev = float(t), (0.0, c_x, c_e), t
            else:                        # this request is through: evaluated at its last timestep, latents kept as they are
                ts_rows[i][b] = float(asc[-1])
    ts_all = torch.tensor(ts_rows, dtype=torch.float32).to(dev).contiguous()
    coef_all = _coef_table(coef_rows, dev)
    eps, nxt = torch.empty_like(x), torch.empty_like(x)
    for i in range(nmax):
        unet(x, ts_all[i], encoder_hidden_states=ctx, added_cond_kwargs=added, out=eps, ip_scales=ip_scales)
        fused_update_v(x, eps, None, coef_all[i], nxt)
        x, nxt = nxt, x
    return x


@torch.no_grad()
def sample_batch(unet, scheduler_config, latents, cond_ctx, uncond_ctx, cond_pooled, uncond_pooled, steps: Sequence[int], cfg: Sequence[float],
                 scales: Optional[Sequence[float]] = None):
    """Guided DDIM sampling xT -> x0 of n requests at once: one UNet evaluation at B_eff = 2 n per iteration, rows [uncond x n | cond x n]
    (the reference's cat([latents] * 2), sdxl_pipeline.py:826, per request), request r with its own schedule length, guidance and IP scale."""
    dev = unet.device
    n = latents.shape[0]
    x = latents.to(device=dev, dtype=torch.float16).contiguous()
    h, w = x.shape[-2] * 8, x.shape[-1] * 8
    f16 = lambda t: t.to(device=dev, dtype=torch.float16)
    ctx = torch.cat([f16(uncond_ctx), f16(cond_ctx)], dim=0).contiguous()
    pooled = torch.cat([f16(uncond_pooled), f16(cond_pooled)], dim=0).contiguous()
    tid = get_add_time_ids(unet, (h, w), (0, 0), (h, w), int(cond_pooled.shape[-1])).repeat(2 * n, 1).to(dev)
    added = {"text_embeds": pooled, "time_ids": tid}
    nmax = max(steps)
    ts_rows, coef_rows = [[0.0] * (2 * n) for _ in range(nmax)], [[(0.0, 1.0, 0.0)] * n for _ in range(nmax)]
    for r, ns in enumerate(steps):
        sch = DDIMScheduler.from_config(scheduler_config)
        sch.set_timesteps(ns)
        desc = [int(t) for t in sch.timesteps]
        for i in range(nmax):
            t = desc[min(i, ns - 1)]
            ts_rows[i][r] = ts_rows[i][n + r] = float(t)
            if i < ns:
                c_x, c_e = sch.step_coeffs(t)
                coef_rows[i][r] = (float(cfg[r]), c_x, c_e)
    ts_all = torch.tensor(ts_rows, dtype=torch.float32).to(dev).contiguous()
    coef_all = _coef_table(coef_rows, dev)
    sc = None
    if scales is not None:
        sc = torch.tensor(list(scales) * 2, dtype=torch.float32).to(dev)
    model_in = torch.cat([x, x], dim=0).contiguous()
    eps, nxt = torch.empty_like(model_in), torch.empty_like(model_in)
    for i in range(nmax):
        unet(model_in, ts_all[i], encoder_hidden_states=ctx, added_cond_kwargs=added, out=eps, ip_scales=sc)
        fused_update_v(model_in[:n], eps[:n], eps[n:], coef_all[i], nxt[:n], nxt[n:])        # eps_u + g (eps_c - eps_u), x_{t-1} to both halves
        model_in, nxt = nxt, model_in
    return model_in[:n].clone()


@torch.no_grad()
def denoise_batch(pipeline, requests: List[EditRequest], group: int = 4, shard: bool = True):
    """N independent edit requests through the hot segment, batched and (with an initialised process group) sharded over the ranks.
    Returns (sampled latents [N,4,h,w], inverted latents [N,4,h,w]) in request order on every rank.
    `group` requests share one launch sequence (B_eff = 2 * group under guidance): 4 is the headline shape; with a queue of requests 8 is 16 %
    cheaper per image (2.05 vs 2.44 ms per image-step, profiles/r03j_two_stream_probe.txt) at twice the latency of a group."""
    from .pipeline import polar_intrtpolate
    N = len(requests)
    if N == 0:
        raise ValueError("denoise_batch needs at least one request")
    if pipeline.ip_adapter_xl is None:
        raise NotImplementedError("denoise_batch drives the IP-Adapter path (construct the pipeline with ip_ckpt=)")
    shapes = {tuple(r.base_latents.shape[-2:]) for r in requests}
    if len(shapes) != 1:
        raise ValueError(f"requests of one call must share the latent size, got {sorted(shapes)}")
    for r in requests:
        if r.num_inference_steps is None or int(r.num_inference_steps) <= 0:
            raise ValueError(f"`num_inference_steps` has to be a positive integer but is {r.num_inference_steps}")
        if float(r.cfg) <= 1.0:
            # the reference (and pipeline.denoise) switch guidance OFF at guidance_scale <= 1 -- one conditional evaluation, eps = eps_c (sdxl_pipeline.py:842-844
            # behind do_classifier_free_guidance) -- while a batched group always evaluates [uncond | cond] and blends: not the same numbers. Say so instead.
            raise ValueError(f"denoise_batch blends eps_u + g (eps_c - eps_u) for every request; cfg={r.cfg} <= 1 means NO guidance in the reference: run that request "
                             f"through pipeline.denoise / __call__ instead")
    if not 1 <= int(group) <= 8:
        raise ValueError(f"group={group}: a group is one launch sequence of 1..8 requests (B_eff = 2 * group <= 16 rows, the limit of the image-projection and "
                         f"embedding kernels)")
    # polar-mixing noise comes from the global CPU RNG in REQUEST order (what N sequential reference calls would draw), on every rank alike
    noises = [r.noise if r.noise is not None else torch.randn(r.base_latents.shape, dtype=torch.float16) for r in requests]
    world = torch.distributed.get_world_size() if (shard and torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
    rank = torch.distributed.get_rank() if world > 1 else 0
    lo, hi = D.shard_range(N, world, rank)
    unet, cfgs = pipeline.pipe.unet, pipeline.pipe.scheduler.config
    ipa = pipeline.ip_adapter_xl
    dev = unet.device
    outs, invs = [], []
    for g0 in range(lo, hi, group):
        reqs = requests[g0:min(g0 + group, hi)]
        steps = [int(r.num_inference_steps) for r in reqs]
        cat = lambda xs: torch.cat([x.to(dev) for x in xs], dim=0)
        inv_ctx = cat([(r.inv_prompt_embeds if r.inv_prompt_embeds is not None else r.negative_prompt_embeds) for r in reqs])
        inv_pool = cat([(r.inv_pooled_prompt_embeds if r.inv_pooled_prompt_embeds is not None else r.negative_pooled_prompt_embeds) for r in reqs])
        x0 = cat([r.base_latents for r in reqs])
        x_inv = invert_batch(unet, cfgs, x0, inv_ctx, inv_pool, steps)
        x_inv_cpu = x_inv.cpu()                                                                 # pipeline.py:331
        mixed = torch.cat([polar_intrtpolate(x_inv_cpu[i:i + 1], noises[g0 + i].to(x_inv_cpu.dtype), r.alpha) for i, r in enumerate(reqs)], dim=0)
        ip, ip_un = ipa.get_image_embeds(clip_image_embeds=torch.stack([r.latent_la.reshape(-1) for r in reqs]), mode="global")
        cond_ctx = torch.cat([cat([r.prompt_embeds for r in reqs]).to(torch.float16), ip], dim=1)              # ip_adapter.py:341-342
        uncond_ctx = torch.cat([cat([r.negative_prompt_embeds for r in reqs]).to(torch.float16), ip_un], dim=1)
        x = sample_batch(unet, cfgs, mixed, cond_ctx, uncond_ctx, cat([r.pooled_prompt_embeds for r in reqs]), cat([r.negative_pooled_prompt_embeds for r in reqs]),
                         steps, [float(r.cfg) for r in reqs], [float(r.scale) for r in reqs])
        outs.append(x)
        invs.append(x_inv)
    like = requests[0].base_latents
    empty = torch.empty((0,) + tuple(like.shape[1:]), dtype=torch.float16, device=dev)
    out, inv = (torch.cat(outs) if outs else empty), (torch.cat(invs) if invs else empty)
    if world > 1:                    # equal-sized shards for the all-gather: pad, gather, cut
        per = (N + world - 1) // world
        def gathered(t):
            pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            pad[: t.shape[0]] = t
            allr = D.gather_batches(pad).reshape(world, per, *t.shape[1:])
            return torch.cat([allr[r][: D.shard_range(N, world, r)[1] - D.shard_range(N, world, r)[0]] for r in range(world)], dim=0)
        out, inv = gathered(out), gathered(inv)
    return out, inv
