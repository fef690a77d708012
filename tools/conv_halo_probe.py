"""The halo-staged 3x3 convolution (tile variants 24 / 25, conv_halo_f16_kernel) against the gathered-operand tiles on the step's convolution shapes:
correctness against an fp32 convolution, then interleaved timing rounds (warm operands; K splits 1 and, on the 16 x 16 maps, 2 / 3 / 4).
usage (GPU box): python tools/conv_halo_probe.py [--quick]"""
import ctypes as C
import os
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi

L = _ffi.lib()
s = _ffi.current_stream()
SHAPES = [  # B, H, W, Cin, Cin2 (appended 1x1 block), Co
    (8, 64, 64, 320, 0, 320), (8, 64, 64, 640, 0, 320), (8, 64, 64, 320, 960, 320), (8, 32, 32, 640, 0, 640), (8, 32, 32, 640, 1920, 640), (8, 32, 32, 320, 0, 640),
    (8, 16, 16, 1280, 0, 1280), (8, 16, 16, 1280, 2560, 1280), (8, 16, 16, 640, 0, 1280), (1, 16, 16, 64, 0, 160), (2, 32, 48, 128, 64, 192)]
if "--quick" in sys.argv:
    SHAPES = [SHAPES[0], SHAPES[3], SHAPES[6], SHAPES[9], SHAPES[10]]


def rel_l2(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm()).item()


def run(B, H, W, Cin, Cin2, Co, tile, splitk, x, x2, wcat, b, y, part):
    L.ia2p_debug_set_gemm_tile(tile)
    if Cin2:
        assert splitk == 1
        _ffi.check(L.ia2p_conv3x3_cat(s, _ffi.ptr(x), _ffi.ptr(x2), _ffi.ptr(wcat), _ffi.ptr(b), _ffi.ptr(y), B, H, W, Cin, Cin2, Co))
    else:
        _ffi.check(L.ia2p_conv3x3_splitk(s, _ffi.ptr(x), _ffi.ptr(wcat), _ffi.ptr(b), None, None, _ffi.ptr(y), B, H, W, Cin, Co, splitk, C.c_void_p(part.data_ptr())))


for (B, H, W, Cin, Cin2, Co) in SHAPES:
    g = torch.Generator().manual_seed(B * 1000 + Cin + Co)
    x = torch.randn(B, H, W, Cin, generator=g).half().cuda()
    x2 = torch.randn(B, H, W, max(Cin2, 64), generator=g).half().cuda()
    w = (torch.randn(Co, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).half().cuda()
    wsc = (torch.randn(Co, max(Cin2, 64), generator=g) * max(Cin2, 64) ** -0.5).half().cuda()
    b = torch.randn(Co, generator=g).half().cuda()
    wp = torch.empty(Co, 9 * Cin, dtype=torch.half, device="cuda")
    _ffi.check(L.ia2p_pack_conv3x3(s, _ffi.ptr(w), _ffi.ptr(wp), Co, Cin))
    wcat = torch.cat([wp, wsc[:, :Cin2]], dim=1).contiguous() if Cin2 else wp
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b.float(), padding=1)
    if Cin2:
        ref = ref + F.conv2d(x2.float().permute(0, 3, 1, 2), wsc.float()[:, :, None, None])
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, Co)
    M = B * H * W
    part = torch.empty(4 * M * Co, dtype=torch.float32, device="cuda")
    plans = [(18, 1), (24, 1), (12, 1), (25, 1), (26, 1), (0, 1)]
    if not Cin2 and M <= 2048:
        plans += [(0, 3), (12, 3), (24, 2), (24, 3), (24, 4), (25, 2), (25, 3), (25, 4), (26, 2), (26, 3)]
    outs, err = {}, {}
    for (tile, sk) in plans:
        y = torch.full((M, Co), float("nan"), dtype=torch.half, device="cuda")
        run(B, H, W, Cin, Cin2, Co, tile, sk, x, x2, wcat, b, y, part)
        torch.cuda.synchronize()
        outs[(tile, sk)] = y
        err[(tile, sk)] = rel_l2(y, ref)
    bad = {k: v for k, v in err.items() if not v < 1e-3}
    same = torch.equal(outs[(24, 1)], outs[(25, 1)]) and torch.equal(outs[(24, 1)], outs[(26, 1)])
    # timing: interleaved rounds, best of 7
    best = {p: 1e9 for p in plans}
    y = torch.empty(M, Co, dtype=torch.half, device="cuda")
    for r in range(8):
        for p in plans:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            run(B, H, W, Cin, Cin2, Co, p[0], p[1], x, x2, wcat, b, y, part)
            e0.record()
            for _ in range(5):
                run(B, H, W, Cin, Cin2, Co, p[0], p[1], x, x2, wcat, b, y, part)
            e1.record()
            torch.cuda.synchronize()
            if r:
                best[p] = min(best[p], e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * M * Co * (9 * Cin + Cin2)
    print(f"B{B} {H}x{W} Cin {Cin}+{Cin2} Co {Co}: max rel-L2 {max(err.values()):.2e} {'BAD ' + str(bad) if bad else 'ok'}; halo 160 == 128 == 80 bits: {same}")
    print("    " + "  ".join(f"v{t}/k{k} {best[(t, k)]:.1f}us {fl / best[(t, k)] / 1e6:.0f}TF" for (t, k) in plans))
L.ia2p_debug_set_gemm_tile(-1)
