"""ONE rank over RCCL (backend "nccl") on a 1-GPU box: RCCL initialises on this image and hardware, and every collective `dist.py` uses runs on the
device with the product's dtypes and message sizes -- the 5.8 GB arena-head broadcast in 1 GiB pieces (full SDXL-base arena), the fp16 all_gather of
result batches, the float64 all_reduce(MAX) / all_gather of the step times, the barrier. What a 1-GPU box cannot show is the transport between GPUs
(tests/dist_nccl_ranks.py does, where there are two). Launched by tests/test_dist_gpu.py::test_rccl_one_rank_collectives."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["IA2P_DIST_WORLD1"] = "1"
from instructany2pix_amd import dist as D
from instructany2pix_amd.config import sdxl_base
from instructany2pix_amd.unet import HipUNet2DConditionModel

rank, world, local = D.init_distributed("nccl")
assert (rank, world) == (0, 1) and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
dev = torch.device(f"cuda:{local}")
torch.cuda.set_device(dev)
unet = HipUNet2DConditionModel(sdxl_base(), dev)
unet.arena.view(torch.int16)[::4097] = 12345                      # (recognisable bytes in the head)
before = int(unet.arena_raw.view(torch.int16)[::4097].to(torch.int64).sum())
t0 = time.time()
route = D.broadcast_weights(unet, src=0, with_ip_adapter=True)    # six 1 GiB broadcasts through RCCL, issued by ONE C-ABI call: ia2p_bcast_arena on torch's communicator
torch.cuda.synchronize()
dt = time.time() - t0
assert route == "abi", route                                      # round 6: the broadcast is part of the C ABI (include/ia2p.h), RCCL bound at run time from torch's instance
assert int(unet.arena_raw.view(torch.int16)[::4097].to(torch.int64).sum()) == before
# the entry point's own argument checks (no communicator, a root outside the communicator), and the torch.distributed route kept behind IA2P_BCAST=torch
import ctypes as C
from instructany2pix_amd import _ffi
L = _ffi.lib()
assert L.ia2p_rccl_available() == 1
comm = D.rccl_comm_ptr(dev)
assert comm != 0
assert L.ia2p_bcast_arena(unet._ctx, None, 0, 1, _ffi.current_stream()) == 1                       # IA2P_ERR_INVALID
assert L.ia2p_bcast_arena(unet._ctx, C.c_void_p(comm), 3, 1, _ffi.current_stream()) == 1 and b"root 3" in L.ia2p_last_error(unet._ctx)
assert L.ia2p_bcast_arena(unet._ctx, C.c_void_p(comm), 0, 1, _ffi.current_stream()) == 0          # straight through ctypes, as a host without torch.distributed would
os.environ["IA2P_BCAST"] = "torch"
assert D.broadcast_weights(unet, src=0, with_ip_adapter=True) == "torch"
del os.environ["IA2P_BCAST"]
torch.cuda.synchronize()
assert int(unet.arena_raw.view(torch.int16)[::4097].to(torch.int64).sum()) == before
x = torch.randn(8, 4, 64, 64, device=dev).half()
assert torch.equal(D.gather_batches(x), x)
assert D.max_over_ranks(3.25, device=dev) == 3.25
assert D.gather_floats(1.5, device=dev) == [1.5]
D.barrier()
print(f"RCCL_WORLD1_OK arena_head_bytes={unet.arena_raw.numel()} broadcast_s={dt:.3f}")
torch.distributed.destroy_process_group()
