"""Micro-benchmark: fused QKV projection + self-attention (ia2p_qkv_self_attention) against the two launches it replaces, on the step's shape
(B = 8 images x 256 tokens, 20 heads, K = 1280), LayerNorm folded, random data, interleaved rounds in one process."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi
L = _ffi.lib()
B, heads, K = int(os.environ.get("B", 8)), 20, 1280
C_, M = heads * 64, B * 256
g = torch.Generator(device="cuda").manual_seed(1)
t = (torch.randn(M, K, device="cuda", generator=g) * 1.5).half()
Wf = (torch.randn(3 * C_, K, device="cuda", generator=g) * K ** -0.5).half()
cs, fb = torch.randn(3 * C_, device="cuda"), torch.randn(3 * C_, device="cuda")
tf = t.float()
st = torch.stack([tf.view(M, 20, 64).sum(2), (tf * tf).view(M, 20, 64).sum(2)], dim=2).permute(1, 0, 2).contiguous()
ln = _ffi.LnFoldC(st.data_ptr(), 20, cs.data_ptr(), fb.data_ptr(), 1e-5)
qkv = torch.empty(M, 3 * C_, dtype=torch.half, device="cuda")
o1, o2 = torch.empty(M, C_, dtype=torch.half, device="cuda"), torch.empty(M, C_, dtype=torch.half, device="cuda")
s = _ffi.current_stream()


def two():
    L.ia2p_gemm_ex(s, _ffi.ptr(t), _ffi.ptr(Wf), None, None, _ffi.ptr(qkv), M, 3 * C_, K, 0, C.addressof(ln), None, None, 1, None)
    L.ia2p_attention(s, _ffi.ptr(qkv), 3 * C_, _ffi.ptr(o1), C_, B, heads, 256, 1, C.c_void_p(qkv.data_ptr() + 2 * C_), C.c_void_p(qkv.data_ptr() + 4 * C_), 3 * C_, 256, 1.0, None, None, 0, 0, 0.0)


def gemm_only():
    L.ia2p_gemm_ex(s, _ffi.ptr(t), _ffi.ptr(Wf), None, None, _ffi.ptr(qkv), M, 3 * C_, K, 0, C.addressof(ln), None, None, 1, None)


def one():
    L.ia2p_qkv_self_attention(s, _ffi.ptr(t), _ffi.ptr(Wf), None, C.addressof(ln), _ffi.ptr(o2), C_, B, heads, K)


def timeit(fn, n=200):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for tile in (0, 12):
    L.ia2p_debug_set_gemm_tile(tile)
    for r in range(3):
        print(f"tile {tile}: QKV GEMM {timeit(gemm_only):.1f} us, GEMM + attention {timeit(two):.1f} us, fused {timeit(one):.1f} us")
L.ia2p_debug_set_gemm_tile(-1)
print("same bits:", bool(torch.equal(o1, o2)))
