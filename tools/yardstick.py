"""Yardstick (NOT used by the product): ia2p_gemm against the vendor GEMM library (torch.nn.functional.linear -> hipBLASLt / rocBLAS) on the
contraction shapes of the denoise step and on 4096^3, BOTH columns under ONE protocol, on the same random fp16 data:

  cold : the autotuner's protocol (csrc/engine.hip::tune_site) -- before every timed launch the L2s are flushed (memset of a 48 MiB region) and the
         activations are read back in (in the real step they were written by the launch just before); ONE launch between two events; fastest of
         REPS rounds, candidates interleaved round-robin (cdna guide §5.4 rule 24). Epilogue-free call (no bias, no residual).
  warm : 20 back-to-back launches between two events (what round 1's yardstick file did for the vendor column only).

For ia2p every (tile variant, K split) candidate is measured and the fastest reported, which is what the autotuner does in place.
Usage: python tools/yardstick.py [out.json]      env: REPS (5), SHAPES=idx,idx  VARIANTS=..  (subset runs)
"""
import ctypes as C
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi

L = _ffi.lib()
SHAPES = [(2048, 3840, 1280, "L2 qkv"), (2048, 1280, 1280, "L2 proj"), (2048, 10240, 1280, "L2 ff-in"), (2048, 1280, 5120, "L2 ff-out"),
          (8192, 1920, 640, "L1 qkv"), (8192, 640, 640, "L1 proj"), (8192, 5120, 640, "L1 ff-in"), (8192, 640, 2560, "L1 ff-out"),
          (616, 166400, 2048, "ctx kv"), (32768, 320, 2880, "conv320@64 as GEMM"), (2048, 1280, 11520, "conv1280@16 as GEMM"), (4096, 4096, 4096, "4096^3")]
REPS = int(os.environ.get("REPS", "5"))


def tile_table():
    """(bm, bn, stages, pp) per variant, from the library when it exports the query, else round 3's table"""
    if hasattr(L, "ia2p_debug_gemm_tile_info"):
        out = []
        v = 0
        while True:
            t = (C.c_int * 4)()
            if L.ia2p_debug_gemm_tile_info(v, t) != 0:
                break
            out.append(tuple(t))
            v += 1
        return out
    return [(128, 128, 2, 0), (128, 128, 3, 0), (128, 64, 2, 0), (128, 64, 3, 0), (64, 64, 2, 0), (64, 64, 3, 0), (64, 160, 2, 0), (64, 160, 3, 0), (128, 160, 2, 0),
            (128, 160, 3, 0), (160, 128, 2, 0), (160, 160, 2, 0), (256, 128, 3, 1), (64, 64, 4, 0), (64, 64, 6, 0), (128, 64, 4, 0), (128, 80, 2, 0), (128, 80, 4, 0),
            (256, 160, 3, 1), (128, 160, 3, 1), (32, 64, 3, 0), (32, 128, 3, 0)]


TILES = tile_table()


def name(v, sk):
    bm, bn, st, pp = TILES[v]
    return f"{bm}x{bn}s{st}{'p' * pp}" + (f"/k{sk}" if sk > 1 else "")


def main():
    s = _ffi.current_stream()
    flush = torch.empty(48 << 20, dtype=torch.uint8, device="cuda")
    sink = torch.zeros(1, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pick = [int(i) for i in os.environ["SHAPES"].split(",")] if os.environ.get("SHAPES") else range(len(SHAPES))
    only = [int(v) for v in os.environ["VARIANTS"].split(",")] if os.environ.get("VARIANTS") else None
    rows = []
    print(f"{'shape':40s} {'vendor cold':>12s} {'ia2p cold':>12s} {'ratio':>6s}   {'vendor warm':>12s} {'ia2p warm':>12s} {'ratio':>6s}   ia2p plan (cold / warm)")
    for si in pick:
        M, N, K, label = SHAPES[si]
        g = torch.Generator(device="cuda").manual_seed(si)
        a = torch.randn(M, K, device="cuda", generator=g).half()
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
        out = torch.empty(M, N, device="cuda", dtype=torch.half)
        fl = 2.0 * M * N * K
        cands = []
        for v, (bm, bn, st, pp) in enumerate(TILES):
            if (only is not None and v not in only) or pp == 3:      # (schedule 3: the halo-staged convolution tiles; a linear problem runs their gathered twins)
                continue
            tiles = -(-M // bm) * -(-N // bn)
            if tiles > 16384 or (bm <= 64 and bn <= 64 and fl > 2e11):
                continue                                        # (hopeless candidates: tiny tiles on the large problems)
            for sk in (1, 2, 3, 4, 6):
                if sk > 1 and (tiles * sk > 1024 or K // 64 // sk < 4 or sk * M * N * 4 > (256 << 20)):
                    continue
                cands.append((v, sk))
        partial = torch.empty(max(1, max(sk for _, sk in cands)) * M * N if any(sk > 1 for _, sk in cands) else 1, device="cuda", dtype=torch.float32)

        def run_ia2p(v, sk):
            L.ia2p_debug_set_gemm_tile(v)
            if sk > 1:
                return L.ia2p_gemm_splitk(s, _ffi.ptr(a), _ffi.ptr(w), None, None, _ffi.ptr(out), M, N, K, sk, C.c_void_p(partial.data_ptr()))
            return L.ia2p_gemm(s, _ffi.ptr(a), _ffi.ptr(w), None, None, _ffi.ptr(out), M, N, K, 0)      # (the plain entry never splits K)

        def run_vendor():
            torch.nn.functional.linear(a, w, out=None)

        arms = [("vendor", run_vendor)] + [((v, sk), (lambda v=v, sk=sk: run_ia2p(v, sk))) for v, sk in cands]
        cold = {k: 1e30 for k, _ in arms}
        warm = {k: 1e30 for k, _ in arms}
        dead = set()
        for r in range(-1, REPS):
            for k, fn in arms:
                if k in dead:
                    continue
                flush.fill_(r & 1)
                sink.add_(a.view(torch.int32).sum())                  # reads all of A back in (the launch before it wrote A in the real step); no temporaries
                e0.record()
                rc = fn()
                e1.record()
                torch.cuda.synchronize()
                if rc not in (None, 0):
                    dead.add(k)
                    continue
                if r >= 0:
                    cold[k] = min(cold[k], e0.elapsed_time(e1))
        # warm: only the vendor and the 6 best cold candidates (plus their K-split-free siblings)
        order = sorted((k for k in cold if k != "vendor" and k not in dead), key=lambda k: cold[k])
        short = ["vendor"] + order[:8]
        fns = dict(arms)
        for r in range(3):
            for k in short:
                fns[k]()
                torch.cuda.synchronize()
                e0.record()
                for _ in range(20):
                    fns[k]()
                e1.record()
                torch.cuda.synchronize()
                warm[k] = min(warm[k], e0.elapsed_time(e1) / 20)
        L.ia2p_debug_set_gemm_tile(-1)
        bc, bw = order[0], min(order[:8], key=lambda k: warm[k])
        vc, vw = cold["vendor"], warm["vendor"]
        tf = lambda ms: fl / ms / 1e9
        print(f"{label + f' {M}x{N}x{K}':40s} {vc*1e3:6.1f}us {tf(vc):4.0f} {cold[bc]*1e3:6.1f}us {tf(cold[bc]):4.0f} {vc/cold[bc]:6.3f}   "
              f"{vw*1e3:6.1f}us {tf(vw):4.0f} {warm[bw]*1e3:6.1f}us {tf(warm[bw]):4.0f} {vw/warm[bw]:6.3f}   {name(*bc)} / {name(*bw)}", flush=True)
        rows.append(dict(label=label, M=M, N=N, K=K, vendor_cold_us=vc * 1e3, ia2p_cold_us=cold[bc] * 1e3, ia2p_cold_plan=name(*bc), vendor_warm_us=vw * 1e3,
                         ia2p_warm_us=warm[bw] * 1e3, ia2p_warm_plan=name(*bw), top_cold={name(*k): round(cold[k] * 1e3, 2) for k in order[:8]},
                         top_warm={name(*k): round(warm[k] * 1e3, 2) for k in short[1:]}))
        del a, w, out, partial
    if len(sys.argv) > 1:
        json.dump(dict(device=torch.cuda.get_device_name(0), reps=REPS, rows=rows), open(sys.argv[1], "w"), indent=1)


main()
