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
