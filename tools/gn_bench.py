"""GroupNorm(+SiLU) on the step's own shapes (cfg 3), back to back, us per call (stats + apply launches) and algorithmic GB/s.
Tuning hook: IA2P_GN_STATS_WGS (workgroup target of the statistics launch) -- an EXPERIMENT knob, read only by libraries built with
IA2P_EXTRA_FLAGS=-DIA2P_EXPERIMENTS (the product build ignores it and says so on stderr); the apply launch's target became a constant in round 4."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi
L = _ffi.lib()
s = _ffi.current_stream()
shapes = [(8, 4096, 320, 11), (8, 4096, 640, 2), (8, 4096, 960, 1), (8, 1024, 640, 9), (8, 1024, 1280, 1), (8, 1024, 1920, 1), (8, 1024, 320, 1), (8, 1024, 960, 1),
          (8, 256, 1280, 17), (8, 256, 2560, 2), (8, 256, 1920, 1), (8, 256, 640, 1)]
tot = 0.0
for B, HW, Cc, cnt in shapes:
    x = torch.randn(B, HW, Cc, device="cuda").half()
    y = torch.empty_like(x)
    ga, be = torch.ones(Cc, device="cuda").half(), torch.zeros(Cc, device="cuda").half()
    part = torch.empty(B * 64 * 32 * 2, dtype=torch.float32, device="cuda")
    fn = lambda: L.ia2p_groupnorm_silu(s, _ffi.ptr(x), _ffi.ptr(y), _ffi.ptr(ga), _ffi.ptr(be), B, HW, Cc, 32, C.c_float(1e-5), 1, C.c_void_p(part.data_ptr()))
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    tot += us * cnt
    print(f"B{B} HW{HW} C{Cc}: {us:6.2f} us  {4.0 * B * HW * Cc / us / 1e3:7.0f} GB/s (x{cnt})")
print(f"sum over the step's 46 calls (approx. counts): {tot / 1e3:.3f} ms")
