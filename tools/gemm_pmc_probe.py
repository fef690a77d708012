"""One GEMM shape, one tile variant, N launches: the target of the rocprofv3 --pmc passes that ask what the kernel waits on.
usage: python tools/gemm_pmc_probe.py M N K variant [launches]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi

M, N, K, v = [int(x) for x in sys.argv[1:5]]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
L = _ffi.lib()
s = _ffi.current_stream()
A = torch.randn(M, K, device="cuda").half()
W = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
out = torch.empty(M, N, device="cuda", dtype=torch.half)
L.ia2p_debug_set_gemm_tile(v)
for _ in range(reps):
    _ffi.check(L.ia2p_gemm(s, _ffi.ptr(A), _ffi.ptr(W), None, None, _ffi.ptr(out), M, N, K, 0))
torch.cuda.synchronize()
