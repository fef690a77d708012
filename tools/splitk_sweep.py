"""Tile x split-K sweep on the weak small-M linear shapes of the denoise step (L2 proj / ff-out)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi

L = _ffi.lib()
TILES = [(128, 128, 2), (128, 128, 3), (128, 64, 2), (128, 64, 3), (64, 64, 2), (64, 64, 3), (64, 160, 2), (64, 160, 3), (128, 160, 2), (128, 160, 3),
         (160, 128, 2), (160, 160, 2)]      # = IA2P_GEMM_TILES (csrc/common.h)
NAMES = {v: "%dx%ds%d" % t for v, t in enumerate(TILES)}
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(x) for x in sh.split("x")) + (sh,) for sh in os.environ["SHAPES"].split(",")]
else:
  SHAPES = [(2048, 1280, 1280, "L2 proj"), (2048, 1280, 5120, "L2 ff-out"), (8192, 640, 640, "L1 proj"), (8192, 640, 2560, "L1 ff-out")]


def time_it(fn, reps=40):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


s = _ffi.current_stream()
for (M, N, K, label) in SHAPES:
    A = torch.randn(M, K, device="cuda").half()
    ncopy = max(1, min(200, int(600e6 // (N * K * 2))))
    Wp = (torch.randn(ncopy, N, K, device="cuda") * K ** -0.5).half()
    R = torch.randn(M, N, device="cuda").half()
    out = torch.empty(M, N, device="cuda", dtype=torch.half)
    part = torch.empty(8 * M * N, device="cuda", dtype=torch.float32)
    fl = 2.0 * M * N * K
    it = [0]
    print(f"{label} {M}x{N}x{K}")
    for v in [int(x) for x in os.environ.get('VARIANTS', '0,2,4').split(',')]:
        L.ia2p_debug_set_gemm_tile(v)
        row = []
        for sk in [int(x) for x in os.environ.get('SPLITS', '1,2,3,4').split(',')]:
            def run():
                W = Wp[it[0] % ncopy]
                it[0] += 1
                L.ia2p_gemm_splitk(s, _ffi.ptr(A), _ffi.ptr(W), None, _ffi.ptr(R), _ffi.ptr(out), M, N, K, sk, _ffi.ptr(part))
            ms = time_it(run)
            row.append(f"sk{sk}: {ms*1e3:6.1f}us {fl/ms/1e9:5.0f}TF")
        print("  %-10s " % NAMES.get(v, "v%d" % v) + "  ".join(row))
L.ia2p_debug_set_gemm_tile(-1)
