"""Micro-benchmark of the fused attention kernel on the shapes of the denoise step (cfg 3), random data."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi
L = _ffi.lib()


def time_it(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


s = _ffi.current_stream()
tot = 0.0
for (B, h, N, cnt, label) in [(8, 20, 256, 60, "self L2"), (8, 10, 1024, 10, "self L1"), (16, 20, 256, 0, "self L2 B16"), (8, 20, 576, 0, "self L2 768px"),
                             (2, 10, 4096, 0, "self L1 1024px"), (2, 20, 1024, 0, "self L2 1024px"), (1, 10, 4096, 0, "self L1 1024px B1")]:
    Cc = h * 64
    qkv = torch.randn(B, N, 3 * Cc, device="cuda").half()
    out = torch.empty(B, N, Cc, device="cuda", dtype=torch.half)
    base = qkv.data_ptr()
    us = time_it(lambda: L.ia2p_attention(s, _ffi.ptr(qkv), 3 * Cc, _ffi.ptr(out), Cc, B, h, N, 1, C.c_void_p(base + 2 * Cc), C.c_void_p(base + 4 * Cc), 3 * Cc, N, 1.0, None, None, 0, 0, 0.0))
    fl = 4.0 * B * h * N * N * 64
    tot += us * cnt
    print(f"{label:14s} B{B} h{h} N{N}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
for (B, h, N, cnt, label) in [(8, 20, 256, 60, "cross L2"), (8, 10, 1024, 10, "cross L1")]:
    Cc = h * 64
    ld = 166400
    q = torch.randn(B, N, Cc, device="cuda").half()
    kv = torch.randn(B * 77, ld, device="cuda").half()
    kvi = torch.randn(B * 4, ld, device="cuda").half()
    out = torch.empty(B, N, Cc, device="cuda", dtype=torch.half)
    us = time_it(lambda: L.ia2p_attention(s, _ffi.ptr(q), Cc, _ffi.ptr(out), Cc, B, h, N, 2, _ffi.ptr(kv), C.c_void_p(kv.data_ptr() + 2 * Cc), ld, 77, 1.0,
                                          _ffi.ptr(kvi), C.c_void_p(kvi.data_ptr() + 2 * Cc), ld, 4, 1.0))
    fl = 4.0 * B * h * N * 81 * 64
    tot += us * cnt
    print(f"{label:14s} B{B} h{h} N{N}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s")
print(f"attention ms/step (cfg 3 launch counts): {tot / 1e3:.2f}")
