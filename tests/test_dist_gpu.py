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


def test_bench_gpus2_real_step_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` end to end on the REAL step (no stub): bench.py launches its own two ranks, rank 0 generates the weights, the
    5.8 GB arena head travels (gloo transport here: both ranks share this box's one GPU, IA2P_FORCE_DEVICE=0), rank 0's measured plans are
    imported by rank 1, barriers and max-over-ranks timing bracket the steps, rank 0 prints ONE line with n_gpus = ranks = 2."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", IA2P_DIST_BACKEND="gloo", IA2P_FORCE_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "1", "--no-cpu-baseline",
                        "--no-secondary", "--no-roofline", "--no-box-probe"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["config"]["dist_backend"] == "gloo"
    assert d["config"]["global_batch"] == 16 and len(d["config"]["per_rank_ms_per_step"]) == 2
    assert d["config"]["weight_broadcast_s"] > 0 and d["config"]["weight_broadcast_route"] == "torch" and d["value"] > 0 and d["ms_per_step"] > 0      # (gloo here; RCCL: "abi")
    assert len(lines[0].encode()) < 8192                                # the line a bounded-tail reader can take (bench.result_line)
    det = json.load(open(os.path.join(ROOT, "bench_detail.json")))     # ... and everything else beside the script
    assert det["self_launched"] is True and det["weight_broadcast"]["bytes"] > 5e9 and det["line"]["n_gpus"] == 2
