import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from instructany2pix_amd import _ffi as f
L = f.lib()
def rnd(*s, seed=0, scale=1.0):
    return (torch.randn(*s, generator=torch.Generator().manual_seed(seed)) * scale).half().cuda()
for (M, C) in [(256, 128), (130, 640)]:
    A, W, b = rnd(M, C, seed=8), rnd(8 * C, C, seed=9, scale=C ** -0.5), rnd(8 * C, seed=10, scale=0.1)
    Wp, bp = torch.empty_like(W), torch.empty_like(b)
    f.check(L.ia2p_pack_geglu(f.current_stream(), f.ptr(W), f.ptr(Wp), 8 * C, C))
    f.check(L.ia2p_pack_geglu(f.current_stream(), f.ptr(b), f.ptr(bp), 8 * C, 1))
    h = A.float() @ W.float().t() + b.float()
    a, g = h.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    import ctypes as Cc
    v, sk = Cc.c_int(), Cc.c_int()
    L.ia2p_debug_gemm_plan(M, 8 * C, C, 0, 1, Cc.addressof(v), Cc.addressof(sk))
    print("auto plan", M, C, v.value, sk.value)
    for tile in range(13):
        out = torch.zeros(M, 4 * C, dtype=torch.half, device="cuda")
        L.ia2p_debug_set_gemm_tile(tile)
        st = L.ia2p_gemm(f.current_stream(), f.ptr(A), f.ptr(Wp), f.ptr(bp), None, f.ptr(out), M, 8 * C, C, 1)
        torch.cuda.synchronize()
        e = float((out.float() - ref).norm() / ref.norm())
        bad = ((out.float() - ref).abs() > 0.05).nonzero()
        print(tile, st, f"{e:.4f}", bad[:3].tolist(), bad[-2:].tolist(), len(bad))
    L.ia2p_debug_set_gemm_tile(-1)
