"""One attention shape, N launches, for rocprofv3 --pmc passes (tools/attn_pmc.sh). argv: B heads Nq reps"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi
L = _ffi.lib()
B, h, N, reps = (int(a) for a in sys.argv[1:5])
Cc = h * 64
qkv = torch.randn(B, N, 3 * Cc, device="cuda").half()
out = torch.empty(B, N, Cc, device="cuda", dtype=torch.half)
base = qkv.data_ptr()
s = _ffi.current_stream()
for _ in range(reps):
    L.ia2p_attention(s, _ffi.ptr(qkv), 3 * Cc, _ffi.ptr(out), Cc, B, h, N, 1, C.c_void_p(base + 2 * Cc), C.c_void_p(base + 4 * Cc), 3 * Cc, N, 1.0, None, None, 0, 0, 0.0)
torch.cuda.synchronize()
