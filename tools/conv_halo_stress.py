"""Randomised shapes through the halo-staged convolution tiles (variants 24 / 25): batch, map size (multiples of 16), channel blocks, appended 1x1 blocks, output widths that
take the 16-byte and the 8-byte epilogue route, K splits -- against an fp32 convolution. usage (GPU box): python tools/conv_halo_stress.py"""
import ctypes as C, random, sys, os, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi
L = _ffi.lib(); s = _ffi.current_stream()
random.seed(7)
def rel(a, b): return ((a.float() - b.float()).norm() / b.float().norm()).item()
bad = 0
for it in range(60):
    B = random.choice([1, 2, 3, 5]); H = 16 * random.choice([1, 2, 3]); W = 16 * random.choice([1, 2, 4])
    Cin = 64 * random.choice([1, 2, 3, 5, 10]); Cin2 = 64 * random.choice([0, 0, 1, 3, 7]); Co = random.choice([64, 100, 128, 160, 192, 320, 328])
    tile = random.choice([24, 25, 26]); sk = random.choice([1, 1, 2, 3, 4]) if not Cin2 else 1
    g = torch.Generator().manual_seed(it)
    x = torch.randn(B, H, W, Cin, generator=g).half().cuda()
    w = (torch.randn(Co, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5).half().cuda()
    b = torch.randn(Co, generator=g).half().cuda()
    wp = torch.empty(Co, 9 * Cin, dtype=torch.half, device="cuda")
    _ffi.check(L.ia2p_pack_conv3x3(s, _ffi.ptr(w), _ffi.ptr(wp), Co, Cin))
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b.float(), padding=1)
    M = B * H * W
    y = torch.full((M, Co), float("nan"), dtype=torch.half, device="cuda")
    L.ia2p_debug_set_gemm_tile(tile)
    if Cin2:
        x2 = torch.randn(B, H, W, Cin2, generator=g).half().cuda()
        wsc = (torch.randn(Co, Cin2, generator=g) * Cin2 ** -0.5).half().cuda()
        ref = ref + F.conv2d(x2.float().permute(0, 3, 1, 2), wsc.float()[:, :, None, None])
        wcat = torch.cat([wp, wsc], dim=1).contiguous()
        _ffi.check(L.ia2p_conv3x3_cat(s, _ffi.ptr(x), _ffi.ptr(x2), _ffi.ptr(wcat), _ffi.ptr(b), _ffi.ptr(y), B, H, W, Cin, Cin2, Co))
    else:
        sk = min(sk, 9 * Cin // 64)
        part = torch.full((sk * M * Co,), float("nan"), dtype=torch.float32, device="cuda")
        _ffi.check(L.ia2p_conv3x3_splitk(s, _ffi.ptr(x), _ffi.ptr(wp), _ffi.ptr(b), None, None, _ffi.ptr(y), B, H, W, Cin, Co, sk, C.c_void_p(part.data_ptr())))
    torch.cuda.synchronize()
    ref = ref.permute(0, 2, 3, 1).reshape(M, Co)
    e = rel(y, ref)
    if not e < 1e-3:
        bad += 1
        print("BAD", it, B, H, W, Cin, Cin2, Co, tile, sk, e)
L.ia2p_debug_set_gemm_tile(-1)
print("done, bad =", bad)
