"""Two ranks sharing ONE GPU (gloo transport): rank 0 loads weights, the HEAD of the flat arena (parameters as loaded) is broadcast, rank 1
adopts it and derives the LayerNorm-folded tail locally -- its whole arena must then equal rank 0's bit for bit;
rank 0 measures kernel plans and broadcasts the table;
both evaluate their own shard of a request batch and the gathered result must equal the single-process result
bit for bit. Launched by tests/test_unet_gpu.py::test_two_ranks_broadcast_and_shard through torch.distributed.run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IA2P_DIST_BACKEND"] = "gloo"
from instructany2pix_amd import dist as D
from instructany2pix_amd.config import tiny
from instructany2pix_amd.unet import HipUNet2DConditionModel, export_plans, import_plans
from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict

rank, world, _ = D.init_distributed()
dev = torch.device("cuda:0")
cfg = tiny()
unet = HipUNet2DConditionModel(cfg, dev)
ipsd = synthetic_state_dict(ip_adapter_specs(cfg, 64)["ip_adapter"], seed=7)
if rank == 0:
    unet.load_state_dict(synthetic_state_dict(unet_param_specs(cfg), seed=7))
    unet.load_ip_adapter_weights(ipsd, scale=0.9, num_tokens=4)
raw = unet.arena_raw
assert 0 < raw.numel() < unet.arena.numel()   # the derived tail (LayerNorm folds) is not part of what travels
host = raw.cpu()                             # gloo moves host memory; RCCL broadcasts unet.arena_raw itself (tests/dist_nccl_ranks.py)
D.broadcast_flat(host, src=0, chunk_bytes=16 << 20)
if rank != 0:
    assert not unet.arena[raw.numel():].any()
    raw.copy_(host)
    unet.adopt_arena(with_ip_adapter=True)   # marks the parameters present and runs the fold kernels on this rank
    unet.load_ip_adapter_weights([], scale=0.9, num_tokens=4)
torch.cuda.synchronize()
v = unet.arena.view(torch.int16).to(torch.int64)
sums = D.gather_batches(torch.stack([v.sum(), (v * (torch.arange(v.numel(), device=dev) % 65521)).sum()]).cpu()[None])
assert torch.equal(sums[0], sums[1]), "rank 1's locally folded arena differs from rank 0's"
g = torch.Generator().manual_seed(3)
B = 4
x = torch.randn(B, 4, 16, 16, generator=g).half().to(dev)
ctx = torch.randn(B, 81, cfg.cross_attention_dim, generator=g).half().to(dev)
te = torch.randn(B, cfg.pooled_dim, generator=g).half().to(dev)
tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]] * B).half().to(dev)
lo, hi = D.shard_range(B, world, rank)


def run(a, b):
    return unet(x[a:b].contiguous(), 401, encoder_hidden_states=ctx[a:b].contiguous(),
                added_cond_kwargs=dict(text_embeds=te[a:b].contiguous(), time_ids=tid[a:b].contiguous()))[0]


allo = D.gather_batches(run(lo, hi).cpu())
if rank == 0:
    full = run(0, B).cpu()
    # built-in plans: per-request results do not depend on which rank (or which batch position) computed them
    assert torch.equal(allo, full), float((allo.float() - full.float()).abs().max())
# measured kernel plans: rank 0 tunes on its shard shape and every rank imports the table (bench.py does the same), so that a
# request gives the same bits on every rank (a K-split choice changes fp32 summation order, deterministically per choice)
table = [None]
if rank == 0:
    assert unet.autotune(x[lo:hi].contiguous(), 401, ctx[lo:hi].contiguous(), dict(text_embeds=te[lo:hi].contiguous(), time_ids=tid[lo:hi].contiguous()), reps=2) > 0
    table[0] = export_plans()
torch.distributed.broadcast_object_list(table, src=0)
if rank != 0:
    assert export_plans() == "" and import_plans(table[0]) == table[0].count(";")
assert export_plans() == table[0]
tuned = D.gather_batches(run(lo, hi).cpu())
if rank == 0:
    other = run(B // 2, B).cpu()                      # rank 1's shard, recomputed here with the same table
    assert torch.equal(tuned[B // 2:], other)
    rel = float((tuned.float() - full.float()).norm() / full.float().norm())
    assert rel < 3e-3, rel
    print("DIST_OK")
D.barrier()
torch.distributed.destroy_process_group()
