"""Two ranks sharing ONE GPU (gloo transport): `InstructAny2PixPipeline.denoise_batch` shards a list of heterogeneous edit requests contiguously over
the ranks, every rank runs its shard in groups, the results are all-gathered in request order -- and equal the single-process run of the same list
bit for bit (same groups, same kernels). Launched by tests/test_batch_gpu.py::test_denoise_batch_sharded_over_two_ranks through torch.distributed.run."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["IA2P_DIST_BACKEND"] = "gloo"
from instructany2pix_amd import dist as D
from tests.test_batch_gpu import _pipe, _requests

rank, world, _ = D.init_distributed()
assert world == 2
cfg, pipe = _pipe()                       # same seeded weights on both ranks (the arena broadcast is covered by dist_two_ranks_one_gpu.py)
reqs = _requests(cfg, 8, seed=21)         # 8 requests, 4 different (steps, cfg, scale, alpha) settings
out, inv = pipe.denoise_batch(reqs, group=2, shard=True)          # rank r: requests [4r, 4r+4) as groups of 2
assert out.shape[0] == 8 and inv.shape[0] == 8
both = D.gather_batches(out.cpu()[None])
assert torch.equal(both[0], both[1]), "ranks disagree on the gathered result"
if rank == 0:
    solo_out, solo_inv = pipe.denoise_batch(reqs, group=2, shard=False)
    assert torch.equal(out, solo_out) and torch.equal(inv, solo_inv), float((out.float() - solo_out.float()).abs().max())
    assert not torch.equal(out[0], out[1])
    print("BATCH_DIST_OK")
D.barrier()
torch.distributed.destroy_process_group()
