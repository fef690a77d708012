"""One rank per GPU over RCCL (backend "nccl"): rank 0 loads the weights, `dist.broadcast_weights` sends the head of the arena device to
device, the other ranks adopt it and fold the LayerNorms locally; every rank then runs its shard of a request batch and the gathered
result must equal rank 0's single-process evaluation bit for bit. Needs >= 2 GPUs: launched by
tests/test_dist_gpu.py::test_rccl_broadcast_and_shard through torch.distributed.run when the box has them (skipped on 1-GPU boxes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
from instructany2pix_amd import dist as D
from instructany2pix_amd.config import tiny
from instructany2pix_amd.unet import HipUNet2DConditionModel, export_plans, import_plans
from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, synthetic_state_dict

rank, world, local = D.init_distributed("nccl")
assert torch.distributed.get_backend() == "nccl"
dev = torch.device(f"cuda:{local}")
torch.cuda.set_device(dev)
cfg = tiny()
unet = HipUNet2DConditionModel(cfg, dev)
if rank == 0:
    unet.load_state_dict(synthetic_state_dict(unet_param_specs(cfg), seed=7))
    unet.load_ip_adapter_weights(synthetic_state_dict(ip_adapter_specs(cfg, 64)["ip_adapter"], seed=7), scale=0.9, num_tokens=4)
route = D.broadcast_weights(unet, src=0, with_ip_adapter=True)    # RCCL broadcast of unet.arena_raw + local fold on ranks != 0: ONE C-ABI call per rank (ia2p_bcast_arena)
assert route == "abi", route
if rank != 0:
    unet.load_ip_adapter_weights([], scale=0.9, num_tokens=4)
torch.cuda.synchronize()
v = unet.arena.view(torch.int16).to(torch.int64)
chk = torch.stack([v.sum(), (v * (torch.arange(v.numel(), device=dev) % 65521)).sum()])[None]
allchk = D.gather_batches(chk)
assert all(torch.equal(allchk[0], allchk[r]) for r in range(world)), "arenas differ across ranks"
g = torch.Generator().manual_seed(3)
B = 2 * world
x = torch.randn(B, 4, 16, 16, generator=g).half().to(dev)
ctx = torch.randn(B, 81, cfg.cross_attention_dim, generator=g).half().to(dev)
te = torch.randn(B, cfg.pooled_dim, generator=g).half().to(dev)
tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]] * B).half().to(dev)
lo, hi = D.shard_range(B, world, rank)
run = lambda a, b: unet(x[a:b].contiguous(), 401, encoder_hidden_states=ctx[a:b].contiguous(),
                        added_cond_kwargs=dict(text_embeds=te[a:b].contiguous(), time_ids=tid[a:b].contiguous()))[0]
table = [None]
if rank == 0:
    unet.autotune(x[lo:hi].contiguous(), 401, ctx[lo:hi].contiguous(), dict(text_embeds=te[lo:hi].contiguous(), time_ids=tid[lo:hi].contiguous()), reps=2)
    table[0] = export_plans()
torch.distributed.broadcast_object_list(table, src=0)
if rank != 0:
    import_plans(table[0])
mine = run(lo, hi)
allo = D.gather_batches(mine)                                      # all_gather of the shard results over RCCL
t_max = D.max_over_ranks(1.0 + rank, device=dev)
assert t_max == float(world)
if rank == 0:
    for r in range(world):
        a, b = D.shard_range(B, world, r)
        assert torch.equal(allo[a:b], run(a, b)), f"shard of rank {r} differs from rank 0's evaluation of the same requests"
    print(f"RCCL_OK world={world}")
D.barrier()
torch.distributed.destroy_process_group()
