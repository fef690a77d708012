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
    # round 6: the one line is what a bounded-tail reader can take -- under 8 KB, strict JSON (no NaN / Infinity tokens), one line
    assert len(lines[0].encode()) < 8192 and "NaN" not in lines[0] and "Infinity" not in lines[0]


def _bench_module():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    sys.modules["bench_mod"] = b
    spec.loader.exec_module(b)
    return b


def test_result_line_is_bounded_strict_json():
    """bench.result_line: the ONE stdout line stays under 8 KB whatever a run produced (round 5's line was 31.5 KB and the driver's record could not parse it),
    carries `roofline` and `cpu_baseline`, holds no NaN / Infinity tokens, and optional blocks are shed in a fixed order before the contract fields would be."""
    import json
    b = _bench_module()
    res = {"metric": "denoise-steps/sec (512x512, 50-step DDIM, batch 8)", "value": 51.123456789, "unit": "steps/s", "n_gpus": 1, "steps": 20, "warmup": 5,
           "ms_per_step": 19.5601234, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
           "config": {"workload": "BASELINE configs[2]: 512x512, latent [8,4,64,64], 50-step DDIM, ctx 81 (77 text + 4 IP tokens), SDXL-base UNet + IP-Adapter",
                      "global_batch": 8, "parallelism": "dp1", "kernel_plans": "committed table instructany2pix_amd/plans/mi355x_bench.plans sha256:0123456789ab (118 shapes)",
                      "context_kv": "projected in every step (reference schedule, as rounds 1-4)", "ranks": 1, "dist_backend": "none (single process)",
                      "per_rank_ms_per_step": [19.56], "per_rank_spread": 0.0, "box_probe": {"gemm_4096_tflops": 921.0, "copy_1gib_gbs": 5200.0, "launch_floor_us": 7.3},
                      "secondary_ms_per_step": {"cfg2_b1_512": 8.4, "cfg5_768_cfg_b8": 36.6, "ref1024_inv_b1": 14.4, "ref1024_cfg_b2": 20.9}},
           "roofline": {"bound": "mfma", "kernel": "ff_in (GEGLU projection)", "achieved": 841.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.3364, "traffic": 160167060.8,
                        "traffic_source": "static: profiles/r05z_pmc_traffic.json; NOT measured in this run", "algorithmic_bytes_per_launch": 52428800.0,
                        "avg_launch_us": float("nan"), "whole_step": {"algorithmic_tflop": 12.751, "ms": 19.56, "tflops": 652.0, "mfma_frac": 0.26},
                        "conv_blocks": {"ms": 3.61, "mfma_frac": 0.36, "hbm_gbs": 394.0, "hbm_frac": float("inf")}},
           "cpu_baseline": {"value": 0.0772, "unit": "steps/s", "cores": 16, "kind": "port", "sample": "oracle (torch fp32, 16 threads): 1 warm-up + 1 timed step of 1 request, then ONE real batch-8 step"},
           "timing": {"runs_ms_per_step": [19.56, 19.55, 19.57, 19.56, 19.58], "median_ms_per_step": 19.56}}
    line = b.result_line(res)
    d = json.loads(line)
    assert len(line.encode()) < 4096 and "\n" not in line and "NaN" not in line and "Infinity" not in line
    assert d["roofline"]["avg_launch_us"] is None and d["roofline"]["conv_blocks"]["hbm_frac"] is None       # non-finite -> null
    assert d["roofline"]["frac"] == 0.3364 and d["cpu_baseline"]["cores"] == 16 and abs(d["value"] - 51.123456789) < 1e-3
    # a run that produces far too much sheds the optional blocks, never the contract fields
    fat = json.loads(json.dumps(b._finite(res)))
    fat["timing"]["runs_ms_per_step"] = [19.5 + 1e-3 * i for i in range(2000)]
    fat["config"]["per_rank_ms_per_step"] = [19.5] * 1500
    line = b.result_line(fat)
    d = json.loads(line)
    assert len(line.encode()) < b.LINE_LIMIT and "runs_ms_per_step" not in d.get("timing", {}) and "per_rank_ms_per_step" not in d["config"]
    assert d["value"] == fat["value"] and d["roofline"]["frac"] == 0.3364 and d["cpu_baseline"]["value"] == 0.0772
    fat["metric"] = "x" * 10000              # nothing left to shed: refuse to print rather than emit a line a reader cannot take
    with pytest.raises(RuntimeError):
        b.result_line(fat)


def test_gpu_numa_cores_follow_the_kfd_topology_not_the_drm_card_order(tmp_path, monkeypatch):
    """DRM card order is not HIP order (a BMC / VGA card0, *_VISIBLE_DEVICES). bench.gpu_numa_cores maps HIP device r through the KFD topology
    (GPU nodes in node order -> PCI address) to /sys/bus/pci/devices/<bdf>/numa_node; a fake sysfs: 2 CPU nodes + 4 GPU nodes, GPUs 0-1 on NUMA 1, GPUs 2-3 on NUMA 0."""
    b = _bench_module()
    sysfs = tmp_path / "sys"
    bdf = []
    for i, (simd, bus, numa) in enumerate([(0, 0, 0), (0, 0, 1), (1024, 0xc1, 1), (1024, 0xd1, 1), (1024, 0x21, 0), (1024, 0x31, 0)]):
        nd = sysfs / "class/kfd/kfd/topology/nodes" / str(i)
        nd.mkdir(parents=True)
        (nd / "properties").write_text(f"cpu_cores_count {0 if simd else 32}\nsimd_count {simd}\nlocation_id {bus << 8}\ndomain 0\n")
        if simd:
            a = f"0000:{bus:02x}:00.0"
            bdf.append(a)
            (sysfs / "bus/pci/devices" / a).mkdir(parents=True)
            (sysfs / "bus/pci/devices" / a / "numa_node").write_text(f"{numa}\n")
    for n, cl in ((0, "0-31"), (1, "32-63")):
        (sysfs / f"devices/system/node/node{n}").mkdir(parents=True)
        (sysfs / f"devices/system/node/node{n}/cpulist").write_text(cl + "\n")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    assert b.hip_device_bdfs(str(sysfs)) == bdf
    cores = b.gpu_numa_cores(4, str(sysfs))
    assert cores[0] == cores[1] == list(range(32, 64)) and cores[2] == cores[3] == list(range(0, 32))
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "3,0")                     # device 0 is physical GPU 3 now
    cores = b.gpu_numa_cores(2, str(sysfs))
    assert cores == [list(range(0, 32)), list(range(32, 64))]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "GPU-deadbeef")             # a UUID list: unknowable from here -> no NUMA preference, never a wrong one
    assert b.gpu_numa_cores(2, str(sysfs)) == [None, None]
    assert b.gpu_numa_cores(2, str(tmp_path / "nothing")) == [None, None]


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
