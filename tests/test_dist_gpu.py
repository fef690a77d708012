"""Multi-process paths on GPUs: two ranks sharing one GPU over gloo (runs on every box) and one rank per GPU over RCCL (runs when the box
has >= 2 GPUs; the driver's 1-GPU boxes skip it)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(script, nproc, port, timeout=300):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", script)]
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_rccl_broadcast_and_shard():
    """backend "nccl" (= RCCL): device-to-device broadcast of the arena head, local LayerNorm fold, sharded evaluation, all_gather"""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"RCCL needs one GPU per rank; this box has {n}")
    world = min(n, 8)
    r = _launch("dist_nccl_ranks.py", world, 29541)
    assert r.returncode == 0 and f"RCCL_OK world={world}" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_rccl_one_rank_collectives():
    """RCCL on THIS box: a one-rank "nccl" process group, the full-size arena-head broadcast and every collective dist.py issues, on the device
    (tests/dist_nccl_world1.py). The transport between GPUs needs the two-GPU test above."""
    r = _launch("dist_nccl_world1.py", 1, 29547)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
