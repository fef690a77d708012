"""Fixed cost vs per-k-tile cost of the FF-in launch (LayerNorm folded in, GEGLU epilogue): M x N fixed, K swept; timing only (random operands).
usage (GPU box): [M=2048 N=10240 VARIANTS=18,12,22,23] python tools/ffin_ksweep.py"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi

L = _ffi.lib()
s = _ffi.current_stream()
M, N = int(os.environ.get("M", 2048)), int(os.environ.get("N", 10240))
VARS = [int(v) for v in os.environ.get("VARIANTS", "18,12,22,23").split(",")]
GEGLU = int(os.environ.get("GEGLU", 1))
LN = int(os.environ.get("LN", 1))
KS = [64, 128, 256, 640, 1280, 2560]
print(f"M={M} N={N} geglu={GEGLU} ln={LN}  us per launch; rows = variant, cols = K {KS}; last: us per k-tile between K = 640 and 2560")
for v in VARS:
    L.ia2p_debug_set_gemm_tile(v)
    row = []
    for K in KS:
        A = torch.randn(M, K, device="cuda").half()
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        slots = K // 64
        st = torch.randn(slots, M, 2, device="cuda").abs().float().contiguous()
        cs, fb = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
        ln = _ffi.LnFoldC(st.data_ptr(), slots, cs.data_ptr(), fb.data_ptr(), 1e-5)
        out = torch.empty(M, N // 2 if GEGLU else N, device="cuda", dtype=torch.half)
        bias = torch.randn(N, device="cuda").half()
        fn = lambda: _ffi.check(L.ia2p_gemm_ex(s, _ffi.ptr(A), _ffi.ptr(W), None if LN else _ffi.ptr(bias), None, _ffi.ptr(out), M, N, K, GEGLU, C.addressof(ln) if LN else None, None, None, 1, None))
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 20)
    print(f"variant {v:2d}: " + " ".join(f"{t:7.2f}" for t in row) + f"   {(row[-1] - row[3]) / 30:6.3f}")
L.ia2p_debug_set_gemm_tile(-1)
