"""GroupNorm-fused halo-staged convolution (ia2p_conv3x3_gn, conv_halo_kernel.h GN = 1) against the plain halo-staged convolution on the step's ResnetBlock2D shapes
(batch 8, 512 x 512): us per launch, interleaved rounds, L2 flushed between launches (a 96 MiB memset) and the operand re-warmed, as the executor's tuner does.
Columns: plain | plain + output statistics | fused | fused + output statistics, and the stand-alone GroupNorm launch the fusion removes.
usage (GPU box): python tools/conv_gn_probe.py [tile ...]      (tile variants: 24 = 160 wide, 25 = 128, 26 = 80; default: all three)"""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi

L = _ffi.lib()
s = _ffi.current_stream()
SHAPES = [  # B, H, W, C0, C1, Co, Ca, splitk
    (8, 64, 64, 320, 0, 320, 0, 0), (8, 64, 64, 640, 320, 320, 0, 0), (8, 64, 64, 320, 0, 320, 960, 0), (8, 32, 32, 640, 0, 640, 0, 0), (8, 32, 32, 320, 0, 640, 0, 0),
    (8, 32, 32, 1280, 640, 640, 0, 0), (8, 16, 16, 1280, 0, 1280, 0, 2), (8, 16, 16, 1280, 0, 1280, 0, 3), (8, 16, 16, 1280, 1280, 1280, 0, 2), (8, 16, 16, 640, 0, 1280, 0, 2)]
TILES = [int(a) for a in sys.argv[1:]] or [24, 25, 26]
flush = torch.empty(96 << 20, dtype=torch.uint8, device="cuda")


def bench(fn, touch, reps=12):
    ts = []
    for r in range(reps + 2):
        flush.fill_(r & 1)
        for t in touch:
            t.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 4]


for (B, H, W, C0, C1, Co, Ca, sk) in SHAPES:
    g = torch.Generator().manual_seed(C0 + Co)
    M, HW, Cin = B * H * W, H * W, C0 + C1
    rn = lambda *sh: torch.randn(*sh, generator=g).half().cuda()
    x0, x1 = rn(M, C0), (rn(M, C1) if C1 else None)
    gamma, beta, b, tv = rn(Cin) * 0.2 + 1, rn(Cin) * 0.2, rn(Co), rn(B, Co)
    wp = (rn(Co, 9 * Cin + Ca) * (9 * Cin) ** -0.5).contiguous()
    xa = rn(M, Ca) if Ca else None
    n = rn(M, Cin)
    st0 = torch.empty(M // 256, C0, 2, dtype=torch.float64, device="cuda")
    _ffi.check(L.ia2p_gn_colstats(s, _ffi.ptr(x0), M, C0, 256, C.c_void_p(st0.data_ptr())))
    st1 = None
    if C1:
        st1 = torch.empty(M // 256, C1, 2, dtype=torch.float64, device="cuda")
        _ffi.check(L.ia2p_gn_colstats(s, _ffi.ptr(x1), M, C1, 256, C.c_void_p(st1.data_ptr())))
    y = torch.empty(M, Co, dtype=torch.half, device="cuda")
    part = torch.empty(max(sk, 1) * M * Co, dtype=torch.float32, device="cuda")
    gout = torch.empty(M // 16, Co, 2, dtype=torch.float64, device="cuda")
    gnp = torch.empty(B * 64 * 32 * 2, dtype=torch.float32, device="cuda")

    def call(fused, stats):
        d = _ffi.ConvGnC()
        p = lambda t: None if t is None else t.data_ptr()
        if fused:
            d.x0, d.C0, d.st0, d.rows0, d.x1, d.C1, d.st1, d.rows1 = p(x0), C0, p(st0), 256, p(x1), C1, p(st1), 256
            d.gamma, d.beta, d.groups, d.eps = p(gamma), p(beta), 32, 1e-5
        else:
            d.x0, d.C0 = p(n), Cin
        d.Wp, d.bias, d.rowvec, d.y, d.B, d.H, d.W, d.Co, d.xa, d.Ca, d.splitk, d.partial = p(wp), p(b), p(tv), p(y), B, H, W, Co, p(xa), Ca, sk, p(part)
        d.gn_out = p(gout) if stats else None
        rows = C.c_int(0)
        return lambda: _ffi.check(L.ia2p_conv3x3_gn(s, C.byref(d), C.byref(rows)))

    gn = lambda: _ffi.check(L.ia2p_groupnorm_silu(s, _ffi.ptr(n), _ffi.ptr(y[:, :Cin] if Co >= Cin else n), _ffi.ptr(gamma), _ffi.ptr(beta), B, HW, Cin, 32, 1e-5, 1, C.c_void_p(gnp.data_ptr()))) if Co >= Cin else None
    t_gn = bench(lambda: _ffi.check(L.ia2p_groupnorm_silu(s, _ffi.ptr(n), _ffi.ptr(n), _ffi.ptr(gamma), _ffi.ptr(beta), B, HW, Cin, 32, 1e-5, 1, C.c_void_p(gnp.data_ptr()))), [n])
    for tile in TILES:
        L.ia2p_debug_set_gemm_tile(tile)
        try:
            src = [x0] + ([x1] if C1 else [])
            r = [bench(call(False, False), [n]), bench(call(False, True), [n]), bench(call(True, False), src), bench(call(True, True), src)]
        finally:
            L.ia2p_debug_set_gemm_tile(-1)
        fl = 2.0 * M * Co * (9 * Cin + Ca)
        print(f"{B}x{H}x{W} {C0}+{C1}->{Co} (+{Ca}) sk{sk} tile {tile}: plain {r[0]:6.1f} us ({fl / r[0] / 1e6:5.0f} TF) | +stats {r[1]:6.1f} | fused {r[2]:6.1f} | fused+stats {r[3]:6.1f} | GroupNorm launch {t_gn:5.1f} us"
              f" | fused - plain = {r[2] - r[0]:+5.1f}, stats = {r[1] - r[0]:+4.1f}", flush=True)
