"""The N>1 path on CPU: world_size 2, gloo. Covers sharding, the flat weight broadcast, the max-over-ranks
timing reduction bench.py uses, and result gathering."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from instructany2pix_amd import dist as D
    r, w, _ = D.init_distributed("gloo")
    assert (r, w) == (rank, world)
    # weight broadcast: rank 0 owns the bytes
    buf = torch.arange(5000, dtype=torch.uint8) if rank == 0 else torch.zeros(5000, dtype=torch.uint8)
    D.broadcast_flat(buf, src=0, chunk_bytes=1024)
    ok_b = bool(torch.equal(buf, torch.arange(5000, dtype=torch.uint8)))
    # batch sharding: ranks own disjoint contiguous request ranges and reproduce identical per-request results
    lo, hi = D.shard_range(7, world, rank)
    reqs = torch.arange(7, dtype=torch.float32)[lo:hi] * 2.0
    pad = torch.zeros(4)
    pad[: hi - lo] = reqs
    allr = D.gather_batches(pad)
    tmax = D.max_over_ranks(1.0 + rank)
    D.barrier()
    q.put((rank, ok_b, (lo, hi), allr.tolist(), tmax))
    torch.distributed.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert [r[2] for r in res] == [(0, 4), (4, 7)]
    assert res[0][3] == res[1][3] == [0.0, 2.0, 4.0, 6.0, 8.0, 10.0, 12.0, 0.0]
    assert res[0][4] == res[1][4] == 2.0


def test_shard_range_covers_everything():
    from instructany2pix_amd.dist import shard_range
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def test_bench_gpus2_launches_its_own_ranks_gloo_stub():
    """`python bench.py --gpus 2` with no launcher in the environment starts two rank processes itself (free rendezvous port), rendezvous over gloo,
    broadcasts, brackets the timed region with barriers, takes the max over ranks and relays ONE line from rank 0 that says what it saw.
    The step is the sleeping stub (IA2P_BENCH_STUB: no GPU here); rank 1 is 25 % slower by construction, so `ms_per_step` must be rank 1's."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(IA2P_DIST_BACKEND="gloo", IA2P_BENCH_STUB="20")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1 and d["data"] == "stub" and "STUB" in d["metric"]
    c = d["config"]
    assert c["ranks"] == 2 and c["dist_backend"] == "gloo" and c["self_launched"] is True
    assert c["weight_broadcast"]["bytes"] == 4 << 16 and c["weight_broadcast"]["seconds"] >= 0
    pr = c["per_rank_ms_per_step"]
    assert len(pr) == 2 and pr[1] > pr[0] * 1.1                       # the straggler is visible ...
    assert abs(d["ms_per_step"] - max(pr)) < 0.2 * max(pr)            # ... and sets the step time
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"] + 1e-9      # whole-job steps/s = ranks x steps / max-over-ranks time
    # round 5: before anything is timed the ranks verify their collectives for VALUES and the equality of their arenas (preflight), the straggler is called out
    # (spread > 3 %), and every rank pinned itself to its own CPU core slice (two ranks on this box: disjoint halves of the usable cores)
    assert "verified on 2 ranks" in c["preflight"] and "arena checksums equal" in c["preflight"], c["preflight"]
    assert c["per_rank_spread"] > 0.1 and "slowest rank 1" in c["per_rank_spread_warning"], c
    ncores = len(os.sched_getaffinity(0))
    assert c["cores_per_rank"] == [float(max(1, ncores // 2))] * 2, (c["cores_per_rank"], ncores)


def test_rank_core_slices_and_preflight_helpers():
    """bench.py's N > 1 plumbing as pure functions: disjoint, equal core slices per rank (NUMA node of the rank's GPU first when it has a whole share left), the cpulist
    parser of /sys/devices/system/node/nodeN/cpulist, the spread warning."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    sys.modules["bench_mod"] = b
    spec.loader.exec_module(b)
    assert b.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and b.parse_cpulist("") == []
    sl = b.affinity_slices(8, list(range(64)))
    assert [len(x) for x in sl] == [8] * 8 and sorted(c for x in sl for c in x) == list(range(64))
    sl = b.affinity_slices(8, list(range(20)))            # 20 cores, 8 ranks: 2 each, 4 spare, all disjoint
    assert [len(x) for x in sl] == [2] * 8 and len({c for x in sl for c in x}) == 16
    sl = b.affinity_slices(4, [0, 1])                     # fewer cores than ranks: shared round-robin, never empty
    assert sl == [[0], [1], [0], [1]]
    numa = [list(range(32, 64))] * 4 + [list(range(0, 32))] * 4      # GPUs 0-3 on node 1, GPUs 4-7 on node 0
    sl = b.affinity_slices(8, list(range(64)), numa)
    assert all(set(sl[r]) <= set(numa[r]) for r in range(8)) and len({c for x in sl for c in x}) == 64
    sl = b.affinity_slices(2, list(range(8)), [list(range(100, 104)), None])      # a node this job may not use: falls back to the pool
    assert sorted(sl[0] + sl[1]) == list(range(8))
    assert b.rank_spread([10.0, 10.1])[1] is None and b.rank_spread([10.0])[0] == 0.0
    sp, warn = b.rank_spread([10.0, 10.0, 10.9])
    assert abs(sp - 0.09) < 1e-9 and "slowest rank 2" in warn
    assert b.preflight(None, 0, 1, "cpu") is None          # single rank: nothing to verify


def test_bench_self_launch_fails_when_a_rank_fails():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(IA2P_DIST_BACKEND="gloo", IA2P_BENCH_STUB="not-a-number")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_self_launch_fails_fast_when_one_rank_dies_after_rendezvous():
    """Rank 1 exits AFTER the rendezvous and the weight broadcast; rank 0 is left in a wait that would last ten minutes (what a collective with a lost peer
    looks like). The launcher polls every child: it must kill rank 0, report both exit codes and return non-zero within seconds, not after rank 0's wait."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(IA2P_DIST_BACKEND="gloo", IA2P_BENCH_STUB="5", IA2P_BENCH_STUB_DIE_RANK="1", IA2P_BENCH_STUB_HANG_S="600")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert r.returncode != 0
    assert took < 30, f"launcher took {took:.0f} s to notice a dead rank"
    assert "rank exit codes" in r.stderr and "3" in r.stderr and "killed" in r.stderr, r.stderr[-1500:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_self_launch_times_out():
    """every rank alive but stuck: the launcher's own deadline (IA2P_BENCH_LAUNCH_TIMEOUT_S) ends the job"""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(IA2P_DIST_BACKEND="gloo", IA2P_BENCH_STUB="5", IA2P_BENCH_STUB_DIE_RANK="7", IA2P_BENCH_STUB_HANG_S="600", IA2P_BENCH_LAUNCH_TIMEOUT_S="12")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and time.time() - t0 < 60 and "launch timeout" in r.stderr, r.stderr[-1500:]


def test_bench_refuses_a_world_size_that_is_not_what_was_asked():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, IA2P_BENCH_STUB="5", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
