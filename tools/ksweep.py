"""K-sweep of one GEMM shape: where does the time of the small transformer GEMMs go (fixed cost vs per-k-step cost)?
Warm, back-to-back launches of the same problem; prints us per launch for each tile variant and K."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi

L = _ffi.lib()
s = _ffi.current_stream()
M, N = int(os.environ.get("M", 2048)), int(os.environ.get("N", 1280))
VARS = [int(v) for v in os.environ.get("VARIANTS", "0,2,3,4,5").split(",")]
KS = [64, 128, 256, 512, 1024, 1280, 2560, 5120]
print(f"M={M} N={N}   us per launch; rows = variant, cols = K {KS}")
for v in VARS:
    L.ia2p_debug_set_gemm_tile(v)
    L.ia2p_debug_set_gemm_splitk(1)
    row = []
    for K in KS:
        A = torch.randn(M, K, device="cuda").half()
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        R = torch.randn(M, N, device="cuda").half()
        b = torch.randn(N, device="cuda").half()
        out = torch.empty(M, N, device="cuda", dtype=torch.half)
        fn = lambda: L.ia2p_gemm(s, _ffi.ptr(A), _ffi.ptr(W), _ffi.ptr(b), _ffi.ptr(R), _ffi.ptr(out), M, N, K, 0)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) * 10)
    print(f"variant {v:2d}: " + " ".join(f"{t:7.2f}" for t in row))
