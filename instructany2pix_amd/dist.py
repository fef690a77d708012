"""Batch-data-parallel execution of independent edit requests across the GPUs of one node.

The reference is single-process, single-GPU (instructany2pix/pipeline.py:124,131); sharding is this build's
addition. One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in
CPU tests). The path has NO per-step exchange: every request's trajectory depends only on its own latents and
conditioning (GroupNorm / LayerNorm / attention are per-sample, CFG pairs stay on one GPU). The only collective
is the one-time broadcast of the head of the flat weight arena (UNet + IP-Adapter parameters as loaded, ~5.8 GB fp16; the LayerNorm-folded
copies behind it are re-derived on every rank) from rank 0 over xGMI,
plus an optional all-gather of the final latents (64 KB per rank at cfg 4).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def _world1_collectives() -> bool:
    """IA2P_DIST_WORLD1=1: a ONE-rank process group is still initialised and the helpers below still issue their collectives (a 1-GPU box can
    then exercise RCCL initialisation and every collective this module uses, with the product's dtypes, on the device:
    tests/test_dist_gpu.py::test_rccl_one_rank_collectives)."""
    return os.environ.get("IA2P_DIST_WORLD1", "") not in ("", "0")


def _active() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or _world1_collectives())


def init_distributed(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment; world_size 1 needs no process group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or _world1_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            # every launcher sets it (torchrun; bench.py's own launcher picks a free port); a fixed default would collide between jobs on one box
            raise RuntimeError("WORLD_SIZE > 1 but MASTER_PORT is not set: start the ranks with a launcher (torchrun, or `bench.py --gpus N`)")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:       # IA2P_DIST_BACKEND=gloo lets two ranks share one GPU in tests (RCCL wants one GPU per rank)
            backend = os.environ.get("IA2P_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        # bounded collectives: a rank that lost a peer fails after IA2P_DIST_TIMEOUT_S (default 10 min; the weight broadcast of 5.8 GB takes seconds)
        # instead of sitting in the collective for the backend's default half hour
        import datetime
        timeout = datetime.timedelta(seconds=float(os.environ.get("IA2P_DIST_TIMEOUT_S", "600")))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local


def shard_range(n_requests: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of the request batch owned by `rank` (remainder spread over the first ranks)."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n_requests, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_flat(buf: torch.Tensor, src: int = 0, chunk_bytes: int = 1 << 30) -> torch.Tensor:
    """Broadcast one flat buffer in <=1 GiB pieces (a few large messages: ring broadcast over xGMI is per-link
    bound, so message count, not size, is what to keep small)."""
    if not _active():
        return buf
    flat = buf.view(-1)
    step = max(1, chunk_bytes // flat.element_size())
    for lo in range(0, flat.numel(), step):
        dist.broadcast(flat[lo:lo + step], src=src)
    return buf


def rccl_comm_ptr(device) -> int:
    """The raw ncclComm_t of the default process group's RCCL communicator on `device` (0 when there is none: another backend, a torch build without the
    accessor). The communicator is created by the first collective: a one-element broadcast makes sure it exists."""
    if not dist.is_initialized() or dist.get_backend() != "nccl":
        return 0
    try:
        t = torch.zeros(1, device=device)
        dist.broadcast(t, src=0)
        torch.cuda.synchronize(device)
        pg = dist.distributed_c10d._get_default_group()._get_backend(torch.device(device))
        return int(pg._comm_ptr())
    except Exception:
        return 0


def broadcast_weights(unet, src: int = 0, with_ip_adapter: bool = True) -> str:
    """Rank `src` holds loaded weights; everyone else receives the HEAD of the arena (the parameters as loaded: 5.8 GB for SDXL-base +
    IP-Adapter, six messages) and derives the LayerNorm-folded tail locally (`ia2p_adopt_arena`), which keeps 2.5 GB off xGMI.
    Under the "nccl" backend (= RCCL) the whole step is ONE C-ABI call, `ia2p_bcast_arena` (include/ia2p.h), on the process group's own communicator -- what a
    host without torch.distributed would call with its ncclComm_t; IA2P_BCAST=torch (or a torch build that does not hand out the communicator) keeps the
    torch.distributed broadcast, as the gloo path always does. Returns which route ran ("abi" / "torch" / "none")."""
    if not _active():
        return "none"
    if dist.get_backend() == "nccl" and os.environ.get("IA2P_BCAST", "abi") != "torch":
        from . import _ffi
        comm = rccl_comm_ptr(unet.device) if _ffi.lib().ia2p_rccl_available() else 0
        ok = torch.tensor([1.0 if comm else 0.0], device=unet.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)               # every rank or none: a collective must not be entered by some ranks only
        if float(ok.item()) > 0:
            torch.cuda.synchronize(unet.device)                 # nothing of torch's is in flight on this communicator
            import ctypes as C
            _ffi.check(_ffi.lib().ia2p_bcast_arena(unet._ctx, C.c_void_p(comm), int(src), int(with_ip_adapter), _ffi.current_stream()), unet._ctx)
            if dist.get_rank() != src:
                unet._weights_gen += 1
            return "abi"
    broadcast_flat(unet.arena_raw, src)
    if dist.get_rank() != src:
        unet.adopt_arena(with_ip_adapter)
    return "torch"


def barrier():
    if _active():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not _active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value: float, device="cpu") -> list:
    """every rank's value, in rank order, on every rank (per-rank step times: a straggler must be visible in the result line)"""
    if not _active():
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def gather_batches(local: torch.Tensor) -> torch.Tensor:
    """Concatenate equally-sized per-rank result batches along dim 0 on every rank."""
    if not _active():
        return local
    out = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(out, local.contiguous())
    return torch.cat(out, dim=0)
