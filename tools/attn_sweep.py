import ctypes as C, os, sys, torch
sys.path.insert(0, "/root/repo")
from instructany2pix_amd import _ffi
L = _ffi.lib(); s = _ffi.current_stream()
def time_it(fn, reps=50):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
B, h = 8, 20; Cc = h * 64
for Nq in (32, 128, 256):
    for Nk in (64, 128, 256, 512, 1024):
        q = torch.randn(B, Nq, Cc, device="cuda").half()
        kv = torch.randn(B, Nk, 2 * Cc, device="cuda").half()
        out = torch.empty(B, Nq, Cc, device="cuda", dtype=torch.half)
        us = time_it(lambda: L.ia2p_attention(s, _ffi.ptr(q), Cc, _ffi.ptr(out), Cc, B, h, Nq, 1, _ffi.ptr(kv), C.c_void_p(kv.data_ptr() + 2 * Cc), 2 * Cc, Nk, 1.0, None, None, 0, 0, 0.0))
        print(f"Nq {Nq:4d} Nk {Nk:5d} blocks {((Nq+127)//128)*B*h:4d}: {us:7.1f} us")
