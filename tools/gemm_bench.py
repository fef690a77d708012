"""Micro-benchmark of the GEMM / implicit-GEMM-conv kernel variants on the shapes of the denoise step (cfg 3).
Random data (cdna guide §5.4 rule 25), interleaved rounds in one process (rule 24). Prints TFLOP/s per variant."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi

L = _ffi.lib()
VARS = [int(v) for v in os.environ.get("VARIANTS", "0,2,4").split(",")]
TILES = [(128, 128, 2), (128, 128, 3), (128, 64, 2), (128, 64, 3), (64, 64, 2), (64, 64, 3), (64, 160, 2), (64, 160, 3), (128, 160, 2), (128, 160, 3),
         (160, 128, 2), (160, 160, 2), (256, 128, 3)]      # = IA2P_GEMM_TILES (csrc/common.h); the last one is the 8-wave ping-pong tile
NAMES = {v: "%dx%ds%d" % t for v, t in enumerate(TILES)}
LIN = [  # (M, N, K, count/step, label)
    (2048, 3840, 1280, 60, "L2 qkv"), (2048, 1280, 1280, 192, "L2 proj"), (2048, 10240, 1280, 60, "L2 ff-in(geglu)"),
    (2048, 1280, 5120, 60, "L2 ff-out"), (8192, 1920, 640, 10, "L1 qkv"), (8192, 640, 640, 40, "L1 proj"),
    (8192, 5120, 640, 10, "L1 ff-in"), (8192, 640, 2560, 10, "L1 ff-out"), (616, 166400, 2048, 1, "ctx kv")]
CONV = [  # (B, H, W, Cin, Co, count, label)
    (8, 64, 64, 320, 320, 6, "c320@64"), (8, 64, 64, 960, 320, 1, "c960>320@64"), (8, 64, 64, 640, 320, 2, "c640>320@64"),
    (8, 32, 32, 640, 640, 8, "c640@32"), (8, 32, 32, 1920, 640, 1, "c1920>640@32"), (8, 32, 32, 1280, 640, 1, "c1280>640@32"),
    (8, 16, 16, 1280, 1280, 12, "c1280@16"), (8, 16, 16, 2560, 1280, 2, "c2560>1280@16"), (8, 16, 16, 1920, 1280, 1, "c1920>1280@16")]
if os.environ.get("QUICK"):
    LIN, CONV = LIN[:4], CONV[:1] + CONV[6:7]


def time_it(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    s = _ffi.current_stream()
    print("%-18s %14s " % ("shape", "GFLOP") + " ".join(f"{NAMES[v]:>10s}" for v in VARS))
    tot = {v: 0.0 for v in VARS}
    best_tot = 0.0
    for (M, N, K, cnt, label) in LIN:
        A = torch.randn(M, K, device="cuda").half()
        # COLD=1: cycle through enough distinct weight copies to exceed the 256 MiB Infinity Cache, as in the real step
        ncopy = max(1, min(200, int(600e6 // (N * K * 2)))) if os.environ.get("COLD") else 1
        Wp = (torch.randn(ncopy, N, K, device="cuda") * K ** -0.5).half()
        out = torch.empty(M, N, device="cuda", dtype=torch.half)
        fl = 2.0 * M * N * K
        row, times = [], {}
        it = [0]
        def run_lin():
            W = Wp[it[0] % ncopy]
            it[0] += 1
            L.ia2p_gemm(s, _ffi.ptr(A), _ffi.ptr(W), None, None, _ffi.ptr(out), M, N, K, 0)
        for v in VARS:
            L.ia2p_debug_set_gemm_tile(v)
            ms = time_it(run_lin, max(20, ncopy))
            times[v] = ms
            tot[v] += ms * cnt
            row.append(f"{fl / ms / 1e9:10.0f}")
        best_tot += min(times.values()) * cnt
        print("%-18s %14.1f " % (f"{label} {M}x{N}x{K}"[:18], fl / 1e9) + " ".join(row) + f"   best {NAMES[min(times, key=times.get)]} {min(times.values())*1e3:.1f}us")
    for (B, H, Wd, Ci, Co, cnt, label) in CONV:
        x = torch.randn(B, H, Wd, Ci, device="cuda").half()
        w = (torch.randn(Co, 9 * Ci, device="cuda") * (9 * Ci) ** -0.5).half()
        y = torch.empty(B, H, Wd, Co, device="cuda", dtype=torch.half)
        fl = 2.0 * B * H * Wd * Co * 9 * Ci
        row, times = [], {}
        for v in VARS:
            L.ia2p_debug_set_gemm_tile(v)
            ms = time_it(lambda: L.ia2p_conv3x3(s, _ffi.ptr(x), _ffi.ptr(w), None, None, None, _ffi.ptr(y), B, H, Wd, Ci, Co, 1, 0), 10)
            times[v] = ms
            tot[v] += ms * cnt
            row.append(f"{fl / ms / 1e9:10.0f}")
        best_tot += min(times.values()) * cnt
        print("%-18s %14.1f " % (label, fl / 1e9) + " ".join(row) + f"   best {NAMES[min(times, key=times.get)]} {min(times.values())*1e3:.1f}us")
    L.ia2p_debug_set_gemm_tile(-1)
    print("ms/step if one variant everywhere: " + " ".join(f"{NAMES[v]}={tot[v]:.2f}" for v in VARS) + f" | best-per-shape {best_tot:.2f}")


main()
