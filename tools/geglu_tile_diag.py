"""where the 256 x 320 GEGLU tile differs from the other tiles (diagnostic for tests/test_ops_gpu.py::test_geglu_projection_on_the_256x320_tile): coordinates of every mismatch
inside the tile (row % 256 -> wave row, fragment row i; packed column % 320 -> column half, fragment column j)"""
import ctypes as C, os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi
L = _ffi.lib(); f = _ffi
s = lambda: f.current_stream()
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().cuda()
LN = int(os.environ.get("LN", 1))
for M, C_, seed in ((2048, 640, 180), (2048, 640, 181), (2048, 640, 182), (2048, 1280, 183), (4096, 320, 184), (1024, 320, 175), (2048, 640, 190), (2048, 640, 191)):
    N = 8 * C_
    X = rnd(M, C_, seed=seed)
    gamma, beta = (1.0 + 0.3 * torch.randn(C_, generator=torch.Generator().manual_seed(seed + 3))).half().cuda(), rnd(C_, seed=seed + 4, scale=0.2)
    W, b = rnd(N, C_, seed=seed + 5, scale=C_ ** -0.5), rnd(N, seed=seed + 6, scale=0.3)
    Wpk, bpk = torch.empty_like(W), torch.empty_like(b)
    f.check(L.ia2p_pack_geglu(s(), f.ptr(W), f.ptr(Wpk), N, C_)); f.check(L.ia2p_pack_geglu(s(), f.ptr(b), f.ptr(bpk), N, 1))
    Wf = torch.empty_like(W); cs = torch.empty(N, dtype=torch.float32, device="cuda"); fb = torch.empty(N, dtype=torch.float32, device="cuda")
    f.check(L.ia2p_fold_layernorm(s(), f.ptr(Wpk), f.ptr(gamma), f.ptr(beta), f.ptr(bpk), f.ptr(Wf), f.ptr(cs), f.ptr(fb), N, C_))
    t = (X.float() * 2 + 0.25).half(); tf = t.float(); slots = C_ // 64
    st = torch.stack([tf.view(M, slots, 64).sum(2), (tf * tf).view(M, slots, 64).sum(2)], dim=2).permute(1, 0, 2).contiguous()
    lnc = f.LnFoldC(st.data_ptr(), slots, cs.data_ptr(), fb.data_ptr(), 1e-5)
    outs = {}
    for tile in (18, 27):
        out = torch.full((M, N // 2), float("nan"), dtype=torch.half, device="cuda")
        L.ia2p_debug_set_gemm_tile(tile)
        if LN:
            f.check(L.ia2p_gemm_ex(s(), f.ptr(t), f.ptr(Wf), None, None, f.ptr(out), M, N, C_, 1, C.addressof(lnc), None, None, 1, None))
        else:
            f.check(L.ia2p_gemm_ex(s(), f.ptr(t), f.ptr(Wpk), f.ptr(bpk), None, f.ptr(out), M, N, C_, 1, None, None, None, 1, None))
        torch.cuda.synchronize()
        outs[tile] = out
    L.ia2p_debug_set_gemm_tile(-1)
    d = (outs[18] != outs[27]).nonzero().tolist()
    print(f"M={M} C={C_} seed={seed} ln={LN}: {len(d)} of {M * N // 2} differ")
    for i, j in d[:12]:
        pc = (j // 16) * 32 + (j % 16)                # packed value column
        r, c = i % 256, pc % 320
        print(f"   out[{i},{j}] {float(outs[18][i, j])!r} vs {float(outs[27][i, j])!r}: tile row {r} (wave row {r // 64}, i {(r % 64) // 16}, frow {r % 16}), value column {c} (half {c // 160}, j {(c % 160) // 16}, fq {(c % 16) // 4}, e {c % 4})")
