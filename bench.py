#!/usr/bin/env python3
"""Headline benchmark of the denoise hot path: denoise-steps/sec (512x512, 50-step DDIM, batch 8).

  python bench.py --gpus 1 --steps 50 --warmup 3
  python bench.py --gpus N ...            (no launcher in the environment: starts its own N ranks, one per GPU, and relays rank 0's line)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: conditional-UNet evaluation (SDXL-base architecture,
B_eff = 8 latents of 64x64x4, 81-token context = 77 text + 4 IP-Adapter image tokens) + the fused DDIM
update, i.e. BASELINE.json configs[2]. Inputs and weights are synthetic (seeded) and resident
in HBM before the timed region. With N > 1 every rank runs its own batch of 8 (weak scaling, no per-step
communication; the head of rank 0's weight arena is RCCL-broadcast once, outside the timed region, and every
other rank derives the LayerNorm-folded tail itself).

Timing: W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on
both sides -> `value` (max over ranks, wall clock). The same region is also bracketed by HIP events on the
launch stream, and `--repeats` (default 5) further K-step runs are timed the same way: `timing` carries every
run and their median.

Rank 0 prints ONE JSON line (< 8 KB, strict JSON, parsed back before it is printed: result_line); besides the
contract fields it carries
  "roofline":     the layer role with the largest share of step time, timed live with HIP events on the launch
                  stream (ia2p_profile_*): algorithmic FLOPs (2*M*N*K) per launch / average launch duration;
                  plus `whole_step` and `conv_blocks` (MFMA and HBM fractions of the conv-block region on the
                  algorithmic bytes of SURVEY.md §8d)
  "cpu_baseline": the CPU oracle (oracle/, torch fp32 on the host cores) timed on a bounded sample
  "config.secondary_ms_per_step": the other single-GPU BASELINE shapes (configs[1]: B=1, 77-token text-only;
                  configs[4]: 768x768 with guidance, B_eff = 8; the reference's 1024x1024 defaults), non-headline.
Everything else (per-role and per-kernel tables, HBM-bound kernels, the secondary shapes' own roofline blocks, the
full box probe) goes to stderr and to `bench_detail.json` beside this script (write_detail).
"""
import argparse
import glob
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0       # dense fp16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0           # spec
HBM_COPY_GBS = 6290.0           # measured float4 copy (same guide)
DEFAULT_PLANS = os.path.join(ROOT, "instructany2pix_amd", "plans", "mi355x_bench.plans")      # the committed kernel plan table (one line "M,N,K,conv,geglu,variant,splitk[,gn];...", # comments)
# algorithmic FLOPs per UNet evaluation and conv-block bytes / FLOPs (17 ResnetBlocks + 4 resample convs + conv_in/out;
# input once, output once, weights once) at the BASELINE shapes, keyed by (B_eff, latent side, with IP-Adapter)
STEP_TFLOP = {(1, 64, False): 1.591, (8, 64, True): 12.751, (8, 96, True): 29.195}
CONV_BLOCK_GB = {(1, 64): 0.792, (8, 64): 1.423, (8, 96): 2.325}
CONV_BLOCK_TFLOP = {(1, 64): 0.406, (8, 64): 3.247, (8, 96): 7.304}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def make_inputs(cfg, B, hw, L, device, cfg_id=3):
    import torch
    def g(seed):
        return torch.Generator("cpu").manual_seed(seed)
    lat = torch.randn(B, 4, hw, hw, generator=g(1000 + cfg_id)).half()
    text = torch.randn(B, 77, cfg.cross_attention_dim, generator=g(2000 + cfg_id))
    ip = torch.nn.functional.layer_norm(torch.randn(B, L - 77, cfg.cross_attention_dim, generator=g(3000 + cfg_id)), (cfg.cross_attention_dim,)) if L > 77 else text[:, :0]
    ctx = torch.cat([text, ip], dim=1).half()
    pooled = torch.randn(B, cfg.pooled_dim, generator=g(4000 + cfg_id)).half()
    tid = torch.tensor([[hw * 8.0, hw * 8.0, 0.0, 0.0, hw * 8.0, hw * 8.0][:cfg.num_time_ids - 1] + ([hw * 8.0] if cfg.num_time_ids == 6 else [6.0])] * B).half()
    return [t.to(device).contiguous() for t in (lat, ctx, pooled, tid)]


def usable_cores():
    """Cores this process may actually use: affinity mask and cgroup CPU quota, not the host's core count."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def conv_block_algorithmic(B, hw):
    """(bytes, flops) of the conv-block region per UNet evaluation: SURVEY.md §8(d)'s figures at the BASELINE shapes, else the same
    definition evaluated by formula (input once, output once, weights once per ResnetBlock / resample conv / boundary conv; SDXL-base)."""
    if (B, hw) in CONV_BLOCK_GB:
        return CONV_BLOCK_GB[(B, hw)] * 1e9, CONV_BLOCK_TFLOP[(B, hw)] * 1e12
    h = hw
    res = [(320, 320, h), (320, 320, h), (320, 640, h // 2), (640, 640, h // 2), (640, 1280, h // 4), (1280, 1280, h // 4), (1280, 1280, h // 4), (1280, 1280, h // 4),
           (2560, 1280, h // 4), (2560, 1280, h // 4), (1920, 1280, h // 4), (1920, 640, h // 2), (1280, 640, h // 2), (960, 640, h // 2), (960, 320, h), (640, 320, h), (640, 320, h)]
    by = fl = 0.0
    for ci, co, s in res:
        sc = ci * co if ci != co else 0
        by += B * s * s * (ci + co) * 2 + (9 * ci * co + 9 * co * co + sc + 1280 * co) * 2
        fl += 2.0 * B * s * s * (9 * ci * co + 9 * co * co + sc)
    for c, si, so in [(320, h, h // 2), (640, h // 2, h // 4), (1280, h // 4, h // 2), (640, h // 2, h)]:
        by += B * (si * si + so * so) * c * 2 + 9 * c * c * 2
        fl += 2.0 * B * so * so * 9 * c * c
    by += B * h * h * (4 + 320) * 2 * 2 + 9 * 4 * 320 * 2 * 2
    fl += 2.0 * B * h * h * 9 * 4 * 320 * 2
    return by, fl


def cpu_baseline(cfg, unet_specs, ip_specs, seed, dev, inputs_cpu, L, B, budget_s=40.0):
    """Oracle UNet + DDIM step on the host cores (every step of the 50-step schedule costs the same). One warm-up step of ONE request (timed: t1), then the REAL
    batch-B step once when B x t1 fits the budget; on a box too slow for that, one step of 2 requests and the measured t2 / t1 scaling extrapolated to B -- said
    so in `sample`. Weights are the very tensors the HIP path was loaded with (same device generator), copied to the host.
    Returns (seconds per batch-B step, cores, description)."""
    import torch
    import oracle
    from instructany2pix_amd.weights import iter_synthetic
    cores = usable_cores()
    torch.set_num_threads(cores)
    t0 = time.time()
    host = lambda it: ((k, v.cpu()) for k, v in it)
    net = oracle.build_unet_fast(cfg, host(iter_synthetic(unet_specs, seed, dev, torch.float16)), host(iter_synthetic(ip_specs, seed, dev, torch.float16)) if L > 77 else None)
    log(f"[cpu_baseline] oracle built in {time.time() - t0:.1f}s, {cores} threads")
    sch = oracle.DDIMSchedulerRef()
    sch.set_timesteps(50)

    def step(nreq, i):
        lat, ctx, pooled, tid = [t[:nreq].float() for t in inputs_cpu]
        t = int(sch.timesteps[i])
        t0 = time.time()
        with torch.no_grad():
            eps = net(lat, t, ctx, added_cond_kwargs=dict(text_embeds=pooled, time_ids=tid))[0]
            sch.step(eps, t, lat)
        return time.time() - t0

    step(1, 0)                                  # (first touch of 11.7 GB of weights: not a timing)
    t1 = step(1, 1)
    if B == 1 or B * t1 <= budget_s:
        tB = step(B, 2) if B > 1 else t1
        what = (f"oracle (torch fp32, {cores} threads = usable cores of this box): one warm-up and one timed step of 1 request ({t1:.2f} s), then ONE timed UNet+DDIM "
                f"step of all {B} requests of the same workload ({tB:.2f} s: measured, batch scaling {tB / t1:.2f}x for {B}x the requests)")
    else:
        t2 = step(2, 2)
        tB = t1 + (t2 - t1) * (B - 1)
        what = (f"oracle (torch fp32, {cores} threads): timed steps of 1 request ({t1:.2f} s) and of 2 requests ({t2:.2f} s); the batch-{B} step is EXTRAPOLATED "
                f"linearly from the two ({tB:.2f} s) because {B} x {t1:.1f} s exceeds the {budget_s:.0f} s budget of the default run")
    log(f"[cpu_baseline] {what}")
    return tB, cores, what



def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def affinity_slices(n_ranks, cores, numa_cores=None):
    """Disjoint CPU core sets, one per rank: eight Python ranks each issue ~27 000 launches/s, and left to the scheduler they migrate and share cores.
    `cores` = the cores this job may use (sorted); `numa_cores[r]` (optional) = cores of the NUMA node GPU r hangs off: a rank takes its share from its
    own node when the node has enough unclaimed cores, else from the remaining pool. Every rank gets floor(len(cores) / n_ranks) cores (at least one;
    with fewer cores than ranks the ranks share round-robin)."""
    cores = sorted(cores)
    per = max(1, len(cores) // n_ranks)
    if len(cores) < n_ranks:
        return [[cores[r % len(cores)]] for r in range(n_ranks)]
    free, out = list(cores), [None] * n_ranks
    for r in range(n_ranks):                      # first pass: ranks whose own node still has a whole share
        mine = [c for c in (numa_cores[r] if numa_cores and numa_cores[r] else []) if c in free]
        if len(mine) >= per:
            out[r] = mine[:per]
            free = [c for c in free if c not in out[r]]
    for r in range(n_ranks):
        if out[r] is None:
            out[r] = free[:per]
            free = free[per:]
    return out


def hip_device_bdfs(sysfs="/sys"):
    """PCI addresses ("0000:c1:00.0") of the GPUs in HIP device order, from the KFD topology: HIP enumerates the KFD nodes that have SIMDs in node order (DRM card
    numbering is NOT that order: a BMC / VGA card0, render-node gaps). ROCR_VISIBLE_DEVICES (applied first, by the runtime below HIP) and then HIP_VISIBLE_DEVICES
    filter / permute the list when they are plain index lists; a UUID form makes the mapping unknowable from here -> None."""
    nodes = []
    for d in glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/[0-9]*")):
        try:
            props = dict(l.split()[:2] for l in open(os.path.join(d, "properties")).read().splitlines() if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue                                   # a CPU node
            loc, dom = int(props["location_id"]), int(props.get("domain", "0"))
            nodes.append((int(os.path.basename(d)), f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"))
        except Exception:
            continue
    bdfs = [b for _, b in sorted(nodes)]
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is None or v.strip() == "":
            continue
        try:
            idx = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:
            return None                                    # GPU-<uuid> form
        bdfs = [bdfs[i] for i in idx if 0 <= i < len(bdfs)]
    return bdfs


def gpu_numa_cores(n_ranks, sysfs="/sys"):
    """cores of the NUMA node of HIP device r, for r < n_ranks, or None where sysfs does not say (no GPU / no NUMA / a container without the KFD topology).
    The device -> PCI address map comes from the KFD topology (hip_device_bdfs), the node from /sys/bus/pci/devices/<bdf>/numa_node."""
    bdfs = hip_device_bdfs(sysfs)
    res = []
    for r in range(n_ranks):
        try:
            node = int(open(os.path.join(sysfs, "bus/pci/devices", bdfs[r], "numa_node")).read())
            res.append(parse_cpulist(open(os.path.join(sysfs, f"devices/system/node/node{node}/cpulist")).read()) if node >= 0 else None)
        except Exception:
            res.append(None)
    return res


LINE_LIMIT = 8192       # bytes of the ONE result line (the driver's record keeps a bounded stdout tail: round 5's 31.5 KB line could not be parsed)


def _finite(x):
    """NaN / +-Infinity are not JSON: a non-finite float becomes null, recursively"""
    if isinstance(x, float):
        return x if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def _round(x, nd=4):
    """floats to `nd` significant-ish decimals (the line is read by people and a parser, not used for arithmetic)"""
    if isinstance(x, float):
        return float(f"{x:.{nd + 2}g}")
    if isinstance(x, dict):
        return {k: _round(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_round(v, nd) for v in x]
    return x


def result_line(res):
    """The one stdout line: strict JSON (no NaN / Infinity tokens), self-checked by parsing it back, under LINE_LIMIT bytes. Optional blocks are dropped in a fixed
    order if a run ever produces more than that (they are all in the detail file)."""
    res = _finite(res)
    res = {k: (_round(v) if isinstance(v, (dict, list)) else v) for k, v in res.items()}      # (the contract scalars -- value, ms_per_step -- keep every digit)
    drop = [("timing", "runs_ms_per_step"), ("config", "secondary_ms_per_step"), ("config", "per_rank_ms_per_step"), ("roofline", "traffic_source"), ("config", "box_probe"),
            ("timing", None), ("roofline", "conv_blocks"), ("roofline", "whole_step")]
    while True:
        line = json.dumps(res, allow_nan=False, separators=(",", ":"))
        if len(line.encode()) < LINE_LIMIT or not drop:
            break
        a, b = drop.pop(0)
        if b is None:
            res.pop(a, None)
        elif isinstance(res.get(a), dict):
            res[a].pop(b, None)
    back = json.loads(line)
    if len(line.encode()) >= LINE_LIMIT or "\n" in line or back.get("metric") != res.get("metric") or back.get("value") != res.get("value"):
        raise RuntimeError(f"result line failed its self-check ({len(line)} bytes)")
    return line


def write_detail(detail, path=None):
    """everything the line does not carry (per-role / per-kernel tables, the secondary shapes' blocks, the full box probe): stderr + a side file beside the script"""
    path = path or os.path.join(ROOT, "bench_detail.json")
    text = json.dumps(_finite(detail), indent=1)
    try:
        with open(path, "w") as f:
            f.write(text + "\n")
    except OSError as e:
        log(f"[bench] could not write {path}: {e}")
    log("[bench detail] " + json.dumps(_finite(detail)))
    return path


def pin_rank(local_rank, local_world):
    """pin THIS rank process to its core slice (before torch starts its threads); returns the slice, or None when pinning is off / impossible.
    Works under any launcher (the driver starts the ranks with torch.distributed.run, not with self_launch)."""
    if local_world <= 1 or os.environ.get("IA2P_BENCH_NO_AFFINITY"):
        return None
    try:
        cores = sorted(os.sched_getaffinity(0))
        mine = affinity_slices(local_world, cores, gpu_numa_cores(local_world))[local_rank]
        os.sched_setaffinity(0, mine)
        return mine
    except Exception as e:          # (never fatal: an unpinned rank still measures)
        log(f"[bench] rank {local_rank}: no CPU pinning ({e})")
        return None


def preflight(D, rank, world, device, arena=None):
    """Before anything is timed on N > 1 ranks: every collective the run uses, checked for VALUES -- a broadcast pattern from rank 0, an all-gather of
    rank-dependent rows, the max reduction -- and, with `arena`, equality of a position-weighted checksum of the weight arena on all ranks (the one-time RCCL
    broadcast of the arena head + the per-rank fold must leave bit-identical arenas). The same checks tests/dist_nccl_ranks.py makes on a tiny model; here
    they validate the first multi-GPU run of the real one before it measures. Raises on any mismatch; returns a short description."""
    import torch
    if world <= 1:
        return None
    pat = (torch.arange(4096, dtype=torch.float32, device=device) * 3 + 1) if rank == 0 else torch.zeros(4096, dtype=torch.float32, device=device)
    D.broadcast_flat(pat, src=0, chunk_bytes=4096)
    if not torch.equal(pat.cpu(), torch.arange(4096, dtype=torch.float32) * 3 + 1):
        raise RuntimeError(f"preflight: rank {rank} received a wrong broadcast pattern")
    rows = D.gather_batches(torch.full((1, 8), float(rank + 1), device=device))
    if rows.shape[0] != world or not all(float(rows[r, 0]) == r + 1 for r in range(world)):
        raise RuntimeError(f"preflight: all_gather returned {rows[:, 0].tolist()} on rank {rank}")
    if D.max_over_ranks(float(rank), device=device) != float(world - 1):
        raise RuntimeError("preflight: max reduction wrong")
    what = f"broadcast / all_gather / max verified on {world} ranks"
    if arena is not None:
        v = arena.view(torch.int16)
        n = v.numel()
        chk = torch.zeros(2, dtype=torch.int64, device=arena.device)
        step = 1 << 26                                 # int64 temporaries of 0.5 GiB at a time
        for lo in range(0, n, step):
            w = v[lo:lo + step].to(torch.int64)
            chk[0] += w.sum()
            chk[1] += (w * ((torch.arange(lo, lo + w.numel(), device=arena.device) % 65521) + 1)).sum()
        allchk = D.gather_batches(chk[None].to(device))
        if not all(torch.equal(allchk[0], allchk[r]) for r in range(world)):
            raise RuntimeError(f"preflight: weight arenas differ across ranks: {allchk.tolist()}")
        what += f"; arena checksums equal ({n * 2 / 1e9:.2f} GB per rank)"
    return what


def rank_spread(per_rank_ms):
    """(max - min) / min of the per-rank step times, and a warning text when a straggler costs more than 3 %"""
    lo, hi = min(per_rank_ms), max(per_rank_ms)
    spread = (hi - lo) / lo if lo > 0 else 0.0
    warn = None
    if len(per_rank_ms) > 1 and spread > 0.03:
        warn = f"per-rank step times spread {100 * spread:.1f} % (slowest rank {per_rank_ms.index(hi)}: {hi:.3f} ms, fastest {lo:.3f} ms): `value` is bound by the straggler"
    return spread, warn


def self_launch(n, argv):
    """`--gpus N` without a launcher in the environment (no WORLD_SIZE): start N fresh rank processes -- one per GPU, RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, a free rendezvous port -- relay rank 0's JSON line, and fail if any rank fails. This parent never touches a GPU (no torch import,
    no HIP call): the ranks are ordinary children, not exec'ed replacements of a process that initialised the device."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):      # (each rank pins itself to its core slice at start-up: pin_rank -- the same code path as under torchrun)
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), IA2P_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread (a blocking read here would sit until rank 0 exits -- and rank 0 sits in a collective until its timeout when a
    # peer has died); the parent polls EVERY child: the first non-zero exit kills the others and fails the run within seconds
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("IA2P_BENCH_LAUNCH_TIMEOUT_S", "3600"))
    rcs = [None] * n
    failed = False
    while any(rc is None for rc in rcs) and not failed:
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        failed = any(rc not in (None, 0) for rc in rcs) or time.time() > deadline
        if not failed and any(rc is None for rc in rcs):
            time.sleep(0.05)
    if failed:
        for r, p in enumerate(procs):
            if p.poll() is None:
                p.kill()                       # (exact children of this process, by handle)
        for r, p in enumerate(procs):
            try:
                rc = p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                rc = -9
            if rcs[r] is None:
                rcs[r] = f"killed({rc})"
        why = "launch timeout" if time.time() > deadline else "a rank failed"
        log(f"[bench] rank exit codes {rcs}: {why}; the other ranks were killed, no result line")
        return 1
    reader.join(timeout=10)
    out = b"".join(chunks).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if not lines:
        log("[bench] rank 0 printed no result line")
        return 1
    print(lines[-1], flush=True)
    return 0


class StubWorkload:
    """IA2P_BENCH_STUB=<ms>: a step that only sleeps. Exercises the launcher, the rendezvous, the barrier / max-over-ranks timing and the result line
    on a box without a GPU (tests/test_dist_cpu.py); its line says "stub" in `metric` and `data` and is never a measurement."""

    def __init__(self, ms, rank):
        self.ms, self.rank = ms, rank

    def run(self, n):
        time.sleep(1e-3 * self.ms * n * (1.0 + 0.25 * self.rank))       # rank r is 25 % slower than rank r-1: max-over-ranks must pick the last one

    def timed(self, n):
        t0 = time.perf_counter()
        self.run(n)
        dt = time.perf_counter() - t0
        return dt, 1e3 * dt


def stub_main(args):
    import torch
    from instructany2pix_amd import dist as D
    rank, world, local = D.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    pf = preflight(D, rank, world, "cpu", arena=torch.arange(1 << 12, dtype=torch.int16))
    wl = StubWorkload(float(os.environ["IA2P_BENCH_STUB"]), rank)
    buf = torch.arange(1 << 16, dtype=torch.float32) if rank == 0 else torch.zeros(1 << 16)
    t_b = time.time()
    D.broadcast_flat(buf, src=0, chunk_bytes=1 << 16)
    bcast_s = time.time() - t_b
    assert float(buf[-1]) == float((1 << 16) - 1), "weight broadcast did not arrive"
    # test hook (tests/test_dist_cpu.py): ONE rank dies after the rendezvous and the broadcast while the others sit where a lost peer would leave them -- in a wait
    # that does not end by itself (a collective with a long timeout); the launcher has to notice the death and end the job
    if os.environ.get("IA2P_BENCH_STUB_DIE_RANK"):
        if rank == int(os.environ["IA2P_BENCH_STUB_DIE_RANK"]):
            os._exit(3)          # (dies on the spot, as a crashed rank does: no interpreter shutdown that would first tear down the process group's threads)
        time.sleep(float(os.environ.get("IA2P_BENCH_STUB_HANG_S", "600")))
    wl.run(args.warmup)
    D.barrier()
    wall, _ = wl.timed(args.steps)
    D.barrier()
    elapsed = D.max_over_ranks(wall)
    per_rank = D.gather_floats(1e3 * wall / args.steps)
    spread, warn = rank_spread(per_rank)
    pins = D.gather_floats(float(len(os.sched_getaffinity(0))))
    if warn and rank == 0:
        log("[bench] WARNING: " + warn)
    if rank == 0:
        print(result_line({"metric": "STUB (sleeping step, no GPU work): launcher / rendezvous / timing plumbing only", "value": world * args.steps / elapsed, "unit": "steps/s",
                           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
                           "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "stub",
                           "config": {"workload": "stub", "ranks": world, "dist_backend": torch.distributed.get_backend() if world > 1 else "none (single process)",
                                      "weight_broadcast": {"bytes": int(buf.numel() * 4), "seconds": bcast_s}, "per_rank_ms_per_step": per_rank,
                                      "per_rank_spread": spread, "per_rank_spread_warning": warn, "preflight": pf, "cores_per_rank": pins,
                                      "self_launched": bool(os.environ.get("IA2P_BENCH_SELF_LAUNCHED"))}}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()



def box_probe(dev):
    """What THIS box does on four fixed micro-workloads, measured in the same process before the timed region: DESCRIPTIVE ONLY. Boxes of the pool differ by up
    to 9 % on identical code, and round 3 showed that this probe does not normalise that away (probe TFLOP/s x ms/step spread 7 % over six boxes: the step is
    bound by launch-level latencies and cache state that a back-to-back GEMM loop does not see). Performance claims rest on same-box interleaved A/Bs
    (tools/ab_env.sh) and on the driver's own number; the probe only tells a reader whether a box was a fast or a slow one.
    (i) a fixed L2-warm 4096^3 fp16 GEMM on the library's 256x128 ping-pong tile (the round-3 probe kernel, pinned so that rounds stay comparable) and on the
    8-phase 256x256 tile, random data; (ii) a 1 GiB device copy; (iii) the back-to-back time of a one-tile GEMM launch = the dependent-launch floor; (iv) the
    K -> 0 intercept of the dominant contraction's shape (2048 x 3840 with a residual, K = 64)."""
    import torch
    from instructany2pix_amd import _ffi
    L = _ffi.lib()
    g = torch.Generator(device=dev).manual_seed(5)
    rnd = lambda *s: (torch.rand(*s, generator=g, device=dev) * 2 - 1).half()
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def timed_gemm(M, N, K, reps, residual=False):
        A, W, out = rnd(M, K), rnd(N, K) * (K ** -0.5), torch.empty(M, N, dtype=torch.half, device=dev)
        R = rnd(M, N) if residual else None
        call = lambda: _ffi.check(L.ia2p_gemm(_ffi.current_stream(), _ffi.ptr(A), _ffi.ptr(W), None, _ffi.ptr(R), _ffi.ptr(out), M, N, K, 0))
        for _ in range(5):
            call()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3            # us per launch

    L.ia2p_debug_set_gemm_tile(12)
    us = timed_gemm(4096, 4096, 4096, 30)
    L.ia2p_debug_set_gemm_tile(22)
    us8 = timed_gemm(4096, 4096, 4096, 30)
    L.ia2p_debug_set_gemm_tile(-1)
    src, dst = torch.empty(1 << 30, dtype=torch.uint8, device=dev), torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    dst.copy_(src)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(5):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    return {"gemm_4096_tflops": 2.0 * 4096 ** 3 / (us * 1e-6) / 1e12, "gemm_4096_tflops_8phase_tile": 2.0 * 4096 ** 3 / (us8 * 1e-6) / 1e12, "copy_1gib_gbs": copy_gbs, "launch_floor_us": timed_gemm(128, 128, 64, 300),
            "k0_intercept_us_2048x3840": timed_gemm(2048, 3840, 64, 200, residual=True),
            "note": "descriptive, not a normaliser (see bench.box_probe); same process, before the timed region; gemm = ia2p_gemm 4096^3 fp16 on random data, L2-warm, 30 launches, "
                    "256x128 ping-pong tile (round 3's probe kernel) / 8-phase 256x256 tile; copy = torch 1 GiB d2d "
                    "(read + write bytes); launch_floor = back-to-back one-tile GEMM launches; k0_intercept = back-to-back 2048x3840x64 launches with a residual"}


class Workload:
    """one denoise loop on resident inputs: UNet evaluation at B_eff + fused (CFG +) DDIM update, ping-ponging two latent buffers"""

    def __init__(self, unet, cfg, B_eff, hw, L, guidance, dev, cfg_id):
        import torch
        from instructany2pix_amd.scheduler import DDIMScheduler
        self.unet, self.B_eff, self.hw, self.L, self.guidance = unet, B_eff, hw, L, guidance
        self.lat, self.ctx, pooled, tid = make_inputs(cfg, B_eff, hw, L, dev, cfg_id)
        if guidance:        # cat([latents] * 2): both halves carry the same latents (sdxl_pipeline.py:826)
            self.lat[B_eff // 2:] = self.lat[:B_eff // 2]
        self.added = dict(text_embeds=pooled, time_ids=tid)
        self.sch = DDIMScheduler()
        self.sch.set_timesteps(50)
        self.ts = [int(t) for t in self.sch.timesteps]
        self.eps = torch.empty_like(self.lat)
        self.x, self.y = self.lat.clone(), torch.empty_like(self.lat)
        self.i = 0
        # context K/V (reference attention_processor.py:358-359,379-380: `to_k/to_v(encoder_hidden_states)`, recomputed every step there although its input is
        # constant over a request's steps): "per_step" = the reference's own schedule (what `value` is quoted on); "per_request" = projected ONCE per request --
        # at the first step of every timed run and at every 50-step request boundary, INSIDE the timed region -- and read from that buffer by the other steps
        # (what the product's pipelines do, bit-identical: tests/test_fullsize_gpu.py).
        self.context_kv = "per_step"

    def step(self):
        from instructany2pix_amd.scheduler import fused_update
        i, ts = self.i, self.ts
        if i % len(ts) == 0:          # a new request every 50 steps: start again from the seeded latents (keeps any --steps finite)
            self.x.copy_(self.lat)
            self.unet.invalidate_context_kv()      # ... with its own conditioning: the context projection runs again (inside whatever region is being timed)
        self.unet.cache_context_kv = self.context_kv == "per_request"
        t = ts[i % len(ts)]
        self.unet(self.x, t, encoder_hidden_states=self.ctx, added_cond_kwargs=self.added, out=self.eps)
        c_x, c_e = self.sch.step_coeffs(t)
        if self.guidance:
            h = self.B_eff // 2       # eps = eps_u + g (eps_c - eps_u), x_{t-1} written to both halves (sdxl_pipeline.py:842-851)
            fused_update(self.x[:h], self.eps[:h], self.eps[h:], self.guidance, c_x, c_e, self.y[:h], self.y[h:])
        else:
            fused_update(self.x, self.eps, None, 1.0, c_x, c_e, self.y)
        self.x, self.y = self.y, self.x
        self.i += 1

    def run(self, n):
        for _ in range(n):
            self.step()

    def timed(self, n):
        """(wall seconds, HIP-event milliseconds) of n steps; the caller brackets with barriers"""
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.unet.invalidate_context_kv()      # no work leaves the timed region: the first timed step projects the context (per_request mode)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        self.run(n)
        e1.record()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, e0.elapsed_time(e1)

    def profile(self, nprof=3):
        """per-kernel-class and per-region HIP-event sums over nprof steps, plus the sampler update timed on its own"""
        import torch
        # the per-launch tables describe the UNet evaluation itself: every profiled step carries the context projection (per_step), so that the "context K/V
        # projection" role shows what ONE projection costs; the headline loop runs it once per request
        mode, self.context_kv = self.context_kv, "per_step"
        self.unet.profile(True)
        self.run(nprof)
        torch.cuda.synchronize()
        table, regions, roles = self.unet.profile_read(), self.unet.profile_read_regions(), self.unet.profile_read_roles()
        self.unet.profile(False)
        self.context_kv = mode
        from instructany2pix_amd.scheduler import fused_update
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            fused_update(self.x, self.eps, None, 1.0, 0.9, 0.1, self.y)
        e1.record()
        torch.cuda.synchronize()
        n = self.x.numel()
        table["ddim_step_kernel"] = dict(launches=nprof, ms=e0.elapsed_time(e1) / reps * nprof, flops=0.0, bytes=3.0 * 2 * n * nprof)
        return table, regions, roles


# layer roles whose launches are MFMA contractions (ia2p_profile_read_role names start with these): the dominant one is `roofline.kernel`
MFMA_ROLES = ("ff_in", "ff_out", "qkv + self-attention", "attention out-projections", "to_q + cross-attention", "conv3x3", "proj_in / proj_out", "context K/V projection")


def roofline_block(table, regions, roles, nprof, B_eff, hw, use_ip, ms_per_step, probe=None):
    """`roofline` is keyed by LAYER ROLE (the executor's call site: FF-in, FF-out, QKV + self-attention, out-projections, to_q + cross-attention, 3x3 convolutions,
    GroupNorm, ...), not by kernel instantiation: the tuner may give one role's launches to different tiles on different boxes (round 4: the same code reported
    gemm<128,160> at 0.21 and gemm<256,160> at 0.36 as "dominant"), a role's work and launch count do not move. `kernel` = the MFMA role with the most time."""
    tot_ms = sum(v["ms"] for k, v in table.items() if k != "ddim_step_kernel")
    gemm = {k: v for k, v in table.items() if v["flops"] > 0 and (k.startswith("gemm_f16_kernel") or k.startswith("conv_halo_f16_kernel") or k.startswith("attention") or k.startswith("qkv_sattn") or k.startswith("qproj_xattn"))}
    mfma_roles = {k: v for k, v in roles.items() if k.startswith(MFMA_ROLES) and v["flops"] > 0 and v["ms"] > 0}
    dom = max(mfma_roles, key=lambda k: mfma_roles[k]["ms"])
    d = roles[dom]
    ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
    traffic, traffic_src = None, None
    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if pmc:                     # HBM-side bytes per launch, from separate rocprofv3 --pmc passes (tools/pmc_traffic.py): per layer role when the file carries roles
        main_k = max(d["kernels"], key=lambda q: d["kernels"][q]["ms"]) if d.get("kernels") else None      # the instantiation that carries most of the role's time on this plan table
        k = json.load(open(pmc[-1]))["kernels"].get(main_k)
        if k:
            traffic = k["traffic_bytes_per_launch"]
            traffic_src = (f"static: profiles/{os.path.basename(pmc[-1])}, kernel {main_k} ({100 * d['kernels'][main_k]['ms'] / d['ms']:.0f} % of the role's time in this run; separate rocprofv3 --pmc "
                           f"passes on the builder's box, tools/pmc_traffic.py; NOT measured in this run)")
    out = {"bound": "mfma", "kernel": dom, "keyed_by": "layer role (executor call site), not kernel instantiation", "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
           "traffic": traffic, "traffic_unit": "bytes/launch (L2<->fabric, Infinity-Cache hits included)", "traffic_source": traffic_src,
           "algorithmic_bytes_per_launch": d["bytes"] / d["launches"],
           # the PMC figure counts what the launch moved, including the NEXT contraction's weights its trailing workgroups prefetched: priced apart
           "prefetch_bytes_per_launch": d.get("prefetch_bytes", 0.0) / d["launches"],
           "traffic_over_algorithmic": (traffic / (d["bytes"] / d["launches"])) if traffic else None,
           "traffic_minus_prefetch_over_algorithmic": ((traffic - d.get("prefetch_bytes", 0.0) / d["launches"]) / (d["bytes"] / d["launches"])) if traffic else None,
           "launches_per_step": d["launches"] / nprof, "avg_launch_us": 1e3 * d["ms"] / d["launches"],
           "flops_per_launch": d["flops"] / d["launches"], "share_of_step": d["ms"] / tot_ms}
    # the MFMA kernel INSTANTIATIONS by time, for cross-reference with the rocprofv3 summary (which instantiation comes first depends on the plan table the tuner measured on
    # this box: a tile shared by several layer shapes lumps them -- the role table above does not)
    out["kernels_of_role"] = {q: {"launches_per_step": w["launches"] / nprof, "ms_per_step": w["ms"] / nprof} for q, w in d.get("kernels", {}).items()}
    out["roles"] = {k: {"ms_per_step": v["ms"] / nprof, "launches_per_step": v["launches"] / nprof, "avg_launch_us": 1e3 * v["ms"] / v["launches"],
                        **({"achieved_tflops": v["flops"] / (v["ms"] * 1e-3) / 1e12, "mfma_frac": v["flops"] / (v["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS} if k.startswith(MFMA_ROLES) and v["flops"] > 0 else
                           {"algorithmic_gbs": v["bytes"] / (v["ms"] * 1e-3) / 1e9, "hbm_frac": v["bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS})}
                    for k, v in sorted(roles.items(), key=lambda kv: -kv[1]["ms"]) if v["ms"] > 0}
    out["top_kernels"] = [{"kernel": k, "ms_per_step": v["ms"] / nprof, "launches_per_step": v["launches"] / nprof, "achieved": v["flops"] / (v["ms"] * 1e-3) / 1e12,
                           "frac": v["flops"] / (v["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS} for k, v in sorted(gemm.items(), key=lambda kv: -kv[1]["ms"])[:5]]
    launches = sum(v["launches"] for k, v in table.items() if k != "ddim_step_kernel") / nprof + 1
    out["launches_per_step_all_kernels"] = launches
    if probe:       # what the step spends on per-launch fixed cost: launches x measured dependent-launch floor (box_probe), as a share of the step
        out["fixed_share"] = {"launches_per_step": launches, "launch_floor_us": probe["launch_floor_us"], "share_of_step": launches * probe["launch_floor_us"] * 1e-3 / ms_per_step,
                              "k0_intercept_us_2048x3840": probe["k0_intercept_us_2048x3840"],
                              "note": "launch_floor = back-to-back one-tile GEMM launches on this box; the K -> 0 intercept of a full-width launch (C store + residual read included) is the larger figure"}
    step_tf = STEP_TFLOP.get((B_eff, hw, use_ip))
    if step_tf is None:
        step_tf = sum(v["flops"] for v in table.values()) / nprof / 1e12
    out["whole_step"] = {"algorithmic_tflop": step_tf, "ms": ms_per_step, "tflops": step_tf / (ms_per_step * 1e-3), "mfma_frac": step_tf / (ms_per_step * 1e-3) / MFMA_PEAK_TFLOPS}
    cb = regions["conv_blocks"]
    cb_bytes, cb_flops = conv_block_algorithmic(B_eff, hw)
    cb_ms = cb["ms"] / nprof
    out["conv_blocks"] = {"ms": cb_ms, "launches_per_step": cb["launches"] / nprof, "algorithmic_bytes": cb_bytes, "algorithmic_flops": cb_flops,
                          "tflops": cb_flops / (cb_ms * 1e-3) / 1e12, "mfma_frac": cb_flops / (cb_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                          "hbm_gbs": cb_bytes / (cb_ms * 1e-3) / 1e9, "hbm_frac": cb_bytes / (cb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "hbm_frac_of_measured_copy_peak": cb_bytes / (cb_ms * 1e-3) / 1e9 / HBM_COPY_GBS,
                          "note": "region = conv_in/out, 17 ResnetBlock2D (GroupNorm+SiLU, 3x3 convs, 1x1 shortcuts), 4 resample convs, skip concats; "
                                  "HIP events around every launch of the region (adds ~2 us per launch)"}
    hbm = {}
    for name in ("gn_stats_kernel+gn_apply_kernel", "concat_kernel", "splitk_reduce_kernel", "ddim_step_kernel", "conv_in_kernel", "conv_out_kernel"):
        v = table.get(name)
        if v and v["ms"] > 0:
            gbs = v["bytes"] / (v["ms"] * 1e-3) / 1e9
            hbm[name] = {"ms_per_step": v["ms"] / nprof, "launches_per_step": v["launches"] / nprof, "avg_launch_us": 1e3 * v["ms"] / v["launches"],
                         "algorithmic_gbs": gbs, "frac_of_8000": gbs / HBM_PEAK_GBS, "frac_of_6290": gbs / HBM_COPY_GBS}
    out["hbm_kernels"] = hbm
    # MFMA-pipe busy fraction of the role's main instantiation (SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): separate rocprofv3 --pmc passes,
    # tools/pmc_mfma.py), static like `traffic`: from the latest profiles/r*_pmc_mfma.json, not measured in this run
    pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")))
    out["mfma_busy_frac"], out["mfma_busy_source"] = None, None
    if pm and d.get("kernels"):
        main_k = max(d["kernels"], key=lambda q: d["kernels"][q]["ms"])
        k = json.load(open(pm[-1])).get("kernels", {}).get(main_k)
        if k:
            out["mfma_busy_frac"] = k.get("mfma_busy_frac")
            out["mfma_busy_source"] = f"static: profiles/{os.path.basename(pm[-1])}, kernel {main_k}; NOT measured in this run"
    return out


def roofline_summary(rb):
    """the part of roofline_block() that rides in the result line (scalars and two small blocks); the per-role / per-kernel tables go to the detail file"""
    keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch", "prefetch_bytes_per_launch",
            "traffic_over_algorithmic", "traffic_minus_prefetch_over_algorithmic", "avg_launch_us", "launches_per_step", "flops_per_launch", "share_of_step",
            "launches_per_step_all_kernels", "mfma_busy_frac")
    out = {k: rb[k] for k in keep if k in rb}
    if not out.get("prefetch_bytes_per_launch"):      # (the role records carry no prefetch bytes: the two prefetch-corrected figures would only repeat the plain ones)
        out.pop("prefetch_bytes_per_launch", None); out.pop("traffic_minus_prefetch_over_algorithmic", None)
    if out.get("traffic_source"):
        out["traffic_source"] = out["traffic_source"].split(" (")[0] + "; NOT measured in this run"
    main_k = max(rb["kernels_of_role"], key=lambda q: rb["kernels_of_role"][q]["ms_per_step"]) if rb.get("kernels_of_role") else None
    out["main_kernel_of_role"] = main_k
    ws, cb = rb["whole_step"], rb["conv_blocks"]
    out["whole_step"] = {"algorithmic_tflop": ws["algorithmic_tflop"], "ms": ws["ms"], "tflops": ws["tflops"], "mfma_frac": ws["mfma_frac"]}
    out["conv_blocks"] = {"ms": cb["ms"], "mfma_frac": cb["mfma_frac"], "hbm_gbs": cb["hbm_gbs"], "hbm_frac": cb["hbm_frac"]}
    return out


def log_table(table, nprof):
    for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
        log(f"  {k:42s} {v['launches'] / nprof:7.1f} launches/step {v['ms'] / nprof:8.3f} ms/step "
            f"{(v['flops'] / (v['ms'] * 1e-3) / 1e12) if v['ms'] else 0:8.1f} TFLOP/s {(v['bytes'] / (v['ms'] * 1e-3) / 1e9) if v['ms'] else 0:8.1f} GB/s(alg)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=5, help="K-step runs timed in all (the first one is `value`); their median goes to `timing`")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--latent", type=int, default=64, help="latent side (64 = 512x512 pixels)")
    ap.add_argument("--ctx", type=int, default=81, help="context tokens (77 text + 4 image tokens)")
    ap.add_argument("--guidance", type=float, default=0.0, help="> 0: classifier-free guidance, the batch is cat([uncond, cond]) (B_eff = --batch)")
    ap.add_argument("--unet", choices=["base", "refiner"], default="base", help="refiner = the second engine config (non-headline; use --ctx 77)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-box-probe", action="store_true", help="skip the calibrated box-speed probe (4096^3 GEMM, 1 GiB copy, launch floor)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the non-headline BASELINE shapes (configs[1], configs[4])")
    ap.add_argument("--no-autotune", action="store_true", help="use the built-in cost model instead of measured kernel plans")
    ap.add_argument("--tune", action="store_true", help="measure the kernel plans in place at start-up instead of importing the committed table")
    ap.add_argument("--plans", default=None, help="import this kernel plan table (default: the committed instructany2pix_amd/plans/mi355x_bench.plans)")
    ap.add_argument("--detail", default=None, help="where the detail JSON goes (default: bench_detail.json beside this script)")
    ap.add_argument("--save-plans", default=None, help="write the kernel plan table in use to this file")
    ap.add_argument("--kernel-table", default=None, help="write the per-kernel timing table (JSON) to this file")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:      # no launcher around us: be the launcher (before anything touches a GPU)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    # N > 1: this rank process pins itself to its own CPU core slice before torch starts its threads (any launcher: torchrun, self_launch)
    pinned = pin_rank(int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))), int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))) if args.gpus > 1 else None
    if os.environ.get("IA2P_BENCH_STUB"):
        return stub_main(args)

    import torch
    from instructany2pix_amd import dist as D
    from instructany2pix_amd.attention_processor import AttnProcessor2_0
    from instructany2pix_amd.config import sdxl_base, sdxl_refiner
    from instructany2pix_amd.unet import HipUNet2DConditionModel, export_plans, import_plans
    from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, iter_synthetic

    rank, world, local = D.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: refusing to print a line whose n_gpus would not be what was asked for")
    dev = torch.device(f"cuda:{os.environ.get('IA2P_FORCE_DEVICE', local)}")     # (override: tests with several ranks on one GPU)
    torch.cuda.set_device(dev)
    cfg = sdxl_base() if args.unet == "base" else sdxl_refiner()
    seed = 7
    unet_specs, ip_specs = unet_param_specs(cfg), ip_adapter_specs(cfg)["ip_adapter"]
    use_ip = args.ctx > 77
    backend = torch.distributed.get_backend() if world > 1 else None

    t0 = time.time()
    unet = HipUNet2DConditionModel(cfg, dev)
    real = os.environ.get("IA2P_UNET_WEIGHTS")       # optional: a diffusers `unet/` checkpoint (directory or .safetensors); not present on the driver's boxes
    if rank == 0:   # weights are generated once (seeded, on the device for speed) and broadcast
        if real:
            from instructany2pix_amd.weights import load_unet_safetensors
            load_unet_safetensors(unet, real)
        else:
            unet.load_state_dict(iter_synthetic(unet_specs, seed, dev, torch.float16))
        if args.unet == "base":      # the IP-Adapter tensors travel with the arena whatever the headline context is (secondary shapes use them)
            unet.load_ip_adapter_weights(iter_synthetic(ip_specs, seed, dev, torch.float16), scale=1.0, num_tokens=4)
    t_b = time.time()
    bcast_route = D.broadcast_weights(unet, src=0, with_ip_adapter=args.unet == "base")      # nccl backend: ia2p_bcast_arena (C ABI) on the process group's communicator
    torch.cuda.synchronize()
    bcast_s = time.time() - t_b
    pf_note = None
    if world > 1:          # before anything is timed: the collectives carry the right VALUES and every rank holds the same arena (raises otherwise)
        pf_note = preflight(D, rank, world, dev, arena=unet.arena)
        log(f"[rank {rank}] preflight: {pf_note}")
    if use_ip:
        unet.load_ip_adapter_weights([], scale=1.0, num_tokens=args.ctx - 77)      # descriptors only: the weights are in the arena
    else:
        unet.set_attn_processor(AttnProcessor2_0())
    torch.cuda.synchronize()
    log(f"[rank {rank}] weights ready in {time.time() - t0:.1f}s (arena {unet.arena.numel() / 1e9:.2f} GB, of which {unet.arena_raw.numel() / 1e9:.2f} GB travel; "
        f"{world} rank(s){', backend ' + backend if backend else ''})")

    probe = None
    if rank == 0 and not args.no_box_probe:
        probe = box_probe(dev)
        log(f"[box probe] 4096^3 GEMM {probe['gemm_4096_tflops']:.0f} TFLOP/s, 1 GiB copy {probe['copy_1gib_gbs']:.0f} GB/s, launch floor {probe['launch_floor_us']:.2f} us, "
            f"K->0 intercept {probe['k0_intercept_us_2048x3840']:.2f} us")

    B, hw, L = args.batch, args.latent, args.ctx
    wl = Workload(unet, cfg, B, hw, L, args.guidance, dev, cfg_id=3)

    # set-up, outside the timed region: measure the candidate (tile, K-split) plans of every GEMM / conv shape once on
    # rank 0 (ia2p_autotune) and hand the table to the other ranks so that all ranks run identical kernels
    def tune(w, what):
        t0 = time.time()
        table = [None]
        if rank == 0:
            n = unet.autotune(w.lat, w.ts[0], w.ctx, w.added)
            table[0] = export_plans()
            log(f"[rank 0] autotune ({what}): {n} GEMM/conv shapes measured in {time.time() - t0:.1f}s")
            if os.environ.get("IA2P_PRINT_PLANS"):
                log("[rank 0] plans: " + table[0])
        if world > 1:
            torch.distributed.broadcast_object_list(table, src=0)
            if rank != 0:
                import_plans(table[0])
        return table[0]

    # Kernel plans ((tile, K-split, GroupNorm-fused) per GEMM / conv shape). Default: the COMMITTED table (instructany2pix_amd/plans/mi355x_bench.plans, measured once on
    # an MI355X with `--tune --save-plans`): every box runs the same kernel instantiations and the same K-splits -- the same bits, and the per-kernel tables under
    # profiles/ describe the kernels the driver times. `--tune` measures in place instead (the round-1..5 default); `--no-autotune` leaves every pick to the cost model.
    plans, tuning = "cost model", False
    if args.plans or (os.path.exists(DEFAULT_PLANS) and not args.tune and not args.no_autotune):
        path = args.plans or DEFAULT_PLANS
        text = "".join(l for l in open(path).read().splitlines() if not l.startswith("#")).strip()
        import_plans(text)
        import hashlib
        plans = f"{'imported from' if args.plans else 'committed table'} {os.path.relpath(path, ROOT)} sha256:{hashlib.sha256(text.encode()).hexdigest()[:12]} ({text.count(';')} shapes; --tune measures in place)"
    elif not args.no_autotune:
        tuning = True
        plans = f"measured in place at start-up ({tune(wl, 'headline').count(';')} shapes)"

    # `value` is quoted on the REFERENCE's schedule: the context K/V projection (`to_k/to_v(encoder_hidden_states)`, attention_processor.py:358-359,379-380) runs in
    # EVERY step, as rounds 1-4 measured it. The loop the product's pipelines run (projection once per request, the other steps read the buffer; bit-identical,
    # tests/test_fullsize_gpu.py) is reported beside it as config.ms_per_step_context_kv_once_per_request.
    wl.context_kv = "per_step"
    wl.run(args.warmup)
    D.barrier()
    wall, ev_ms = wl.timed(args.steps)
    D.barrier()
    elapsed = D.max_over_ranks(wall, device=dev if world > 1 else "cpu")
    per_rank_ms = D.gather_floats(1e3 * wall / args.steps, device=dev if world > 1 else "cpu")      # a straggler shows here
    spread, spread_warn = rank_spread(per_rank_ms)
    if spread_warn and rank == 0:
        log("[bench] WARNING: " + spread_warn)
    assert torch.isfinite(wl.x).all(), "non-finite latents after the timed run"
    runs = [ev_ms / args.steps]
    for _ in range(max(0, args.repeats - 1)):
        D.barrier()
        _, ms = wl.timed(args.steps)
        runs.append(ms / args.steps)
    D.barrier()

    default_cfg = (args.batch, args.latent, args.ctx, args.unet, args.guidance) == (8, 64, 81, "base", 0.0)
    metric = "denoise-steps/sec (512x512, 50-step DDIM, batch 8)" if default_cfg else \
        f"denoise-steps/sec ({args.latent * 8}x{args.latent * 8}, 50-step DDIM, batch {args.batch}{', SDXL-refiner UNet' if args.unet == 'refiner' else ''}) [non-headline shape]"
    res = {
        "metric": metric, "value": world * args.steps / elapsed, "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic inputs, checkpoint weights" if real else "synthetic",
        "config": {"workload": f"{'BASELINE configs[2]' if default_cfg else 'custom'}: {hw * 8}x{hw * 8}, latent [{B},4,{hw},{hw}], 50-step DDIM, ctx {L} "
                               f"({'77 text + %d IP tokens' % (L - 77) if use_ip else 'text only'}), {'SDXL-base UNet + IP-Adapter' if args.unet == 'base' else 'SDXL-refiner UNet'}",
                   "global_batch": B * world, "parallelism": f"dp{world}", "kernel_plans": plans,
                   "context_kv": "projected in every step (reference schedule, as rounds 1-4)",
                   "ranks": world, "dist_backend": backend or "none (single process)", "per_rank_ms_per_step": per_rank_ms, "per_rank_spread": spread,
                   "weight_broadcast_s": bcast_s if world > 1 else None, "weight_broadcast_route": bcast_route, "detail_file": "bench_detail.json"},
        "timing": {"runs_ms_per_step": runs, "median_ms_per_step": statistics.median(runs)},
    }
    # what the line does not carry goes to stderr and to bench_detail.json beside this script
    detail = {"metric": metric, "timing_method": "rank 0: HIP events on the launch stream around each K-step run; run 0 is the region `value` is quoted on (wall clock, max over ranks)",
              "image_steps_per_s": world * (B // 2 if args.guidance else B) * args.steps / elapsed, "self_launched": bool(os.environ.get("IA2P_BENCH_SELF_LAUNCHED")),
              "preflight": pf_note, "cpu_cores_pinned": pinned, "per_rank_spread_warning": spread_warn,
              "weight_broadcast": {"bytes": int(unet.arena_raw.numel()), "seconds": bcast_s, "note": "head of the arena only; LayerNorm-folded tail derived per rank"} if world > 1 else None}

    if probe:
        detail["box_probe"] = probe
        res["config"]["box_probe"] = {k: probe[k] for k in ("gemm_4096_tflops", "copy_1gib_gbs", "launch_floor_us")}      # tells a fast box from a slow one
    if rank == 0:      # side note, not `value`: the loop the product's pipelines run -- one whole 50-step request per timed run, its context projected once, inside
        wl.context_kv = "per_request"
        wl.run(1)
        _, ms = wl.timed(50)
        res["config"]["ms_per_step_context_kv_once_per_request"] = ms / 50
        wl.context_kv = "per_step"
    if rank == 0 and not args.no_roofline:
        nprof = 3
        table, regions, roles = wl.profile(nprof)
        rb = roofline_block(table, regions, roles, nprof, B, hw, use_ip, statistics.median(runs), probe)
        detail["roofline"] = rb
        res["roofline"] = roofline_summary(rb)
        log_table(table, nprof)
        if args.kernel_table:
            os.makedirs(os.path.dirname(os.path.abspath(args.kernel_table)), exist_ok=True)
            json.dump({"steps_profiled": nprof, "kernels": table, "regions": regions, "roles": roles}, open(args.kernel_table, "w"), indent=1)

    # ---- the other single-GPU BASELINE shapes, same process, non-headline ----------------------------------------------------
    if rank == 0 and world == 1 and default_cfg and not args.no_secondary:
        sec = {}
        for name, (b2, hw2, L2, g2, cid) in {"configs[1]: 512x512, batch 1, 77-token text-only context": (1, 64, 77, 0.0, 2),
                                              "configs[4]: 768x768, 4 requests with classifier-free guidance 10 (B_eff 8), 81-token contexts": (8, 96, 81, 10.0, 5),
                                              # the reference's own default request (pipeline.py:303, serve.py:80): 1024^2, inversion at batch 1 / guided sampling at B_eff 2
                                              "reference default, inversion: 1024x1024, batch 1, 77-token context (IP processors installed: the last 4 text tokens are split off)": (1, 128, 77, 0.0, 6),
                                              "reference default, guided sampling: 1024x1024, 1 request with classifier-free guidance 10 (B_eff 2), 81-token contexts": (2, 128, 81, 10.0, 7)}.items():
            inv_quirk = "inversion" in name
            if L2 > 77 or inv_quirk:
                unet.load_ip_adapter_weights([], scale=1.0, num_tokens=4)
            else:
                unet.set_attn_processor(AttnProcessor2_0())
            w2 = Workload(unet, cfg, b2, hw2, L2, g2, dev, cfg_id=cid)
            w2.context_kv = "per_request"      # (secondary shapes: the product's loop, one whole 50-step request per timed run)
            if tuning:
                tune(w2, name.split(":")[0])
            w2.run(3)
            n2 = 50      # one whole request per timed run: its context projection is inside, amortised over the 50 steps it serves
            r2 = [w2.timed(n2)[1] / n2 for _ in range(3)]
            ms2 = statistics.median(r2)
            entry = {"ms_per_step": ms2, "steps_per_s": 1e3 / ms2, "runs_ms_per_step": r2, "B_eff": b2, "latent": hw2, "context_tokens": L2, "guidance": g2, "context_kv": "once per request"}
            if not args.no_roofline:
                t2, rg2, ro2 = w2.profile(3)
                rb2 = roofline_block(t2, rg2, ro2, 3, b2, hw2, L2 > 77, ms2, probe)
                entry["roofline"] = {k: rb2[k] for k in ("kernel", "achieved", "frac", "roles", "top_kernels", "whole_step", "conv_blocks", "hbm_kernels", "launches_per_step_all_kernels", "fixed_share") if k in rb2}
                if b2 == 1:
                    wbytes = 5.817e9 if inv_quirk else 5.135e9           # compulsory weight bytes of an evaluation with / without the IP-Adapter projections
                    entry["weight_streaming"] = {"bytes": wbytes, "bound_ms_at_6290": wbytes / HBM_COPY_GBS / 1e6, "frac_of_bound": wbytes / HBM_COPY_GBS / 1e6 / ms2}
            sec[name] = entry
            log(f"[secondary] {name}: {ms2:.2f} ms/step")
        detail["secondary"] = sec
        res["config"]["secondary_ms_per_step"] = {("cfg2_b1_512" if "configs[1]" in k else "cfg5_768_cfg_b8" if "configs[4]" in k else "ref1024_inv_b1" if "inversion" in k else "ref1024_cfg_b2"): v["ms_per_step"]
                                                  for k, v in sec.items()}
        unet.load_ip_adapter_weights([], scale=1.0, num_tokens=4)
    if args.save_plans and rank == 0:      # (after the secondary shapes: a --tune run leaves ONE table that covers every workload of this file)
        with open(args.save_plans, "w") as f:
            f.write(export_plans() + "\n")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        inputs_cpu = [t.cpu() for t in (wl.lat, wl.ctx, wl.added["text_embeds"], wl.added["time_ids"])]
        t_step, cores, what = cpu_baseline(cfg, unet_specs, ip_specs, seed, dev, inputs_cpu, L, B)
        detail["cpu_baseline_sample"] = what
        res["cpu_baseline"] = {"value": 1.0 / t_step, "unit": "steps/s", "cores": cores, "kind": "port",
                               "sample": f"oracle (torch fp32, {cores} threads): 1 warm-up + 1 timed step of 1 request, then ONE real batch-{B} UNet+DDIM step ({t_step:.1f} s)"
                                         if "EXTRAPOLATED" not in what else f"oracle (torch fp32, {cores} threads): steps of 1 and 2 requests, batch-{B} step EXTRAPOLATED ({t_step:.1f} s)"}

    if rank == 0:
        detail["line"] = res
        write_detail(detail, args.detail)
        print(result_line(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
