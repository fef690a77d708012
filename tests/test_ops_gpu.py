"""Per-kernel parity on the MI355X: each hand-written HIP kernel, called through the C ABI, against a plain
torch fp32 reference of the same operator on identical fp16 inputs. Tolerances are stated per test
(per-kernel rel-L2 <= 2e-3; norms/elementwise <= 1e-3)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
NTILES = 28      # entries of IA2P_GEMM_TILES (csrc/common.h); tests/test_abi_cpu.py checks it against the library's table


@pytest.fixture(scope="module")
def L():
    from instructany2pix_amd import _ffi
    assert torch.cuda.is_available(), "GPU tests need a device"
    lib = _ffi.lib()
    assert lib.ia2p_device_is_gfx950() == 1
    return lib


def _ffi():
    from instructany2pix_amd import _ffi as f
    return f


def rel_l2(a, b):
    a, b = a.float(), b.float()
    return float((a - b).norm() / (b.norm() + 1e-12))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().cuda()


def run(L, name, *args):
    f = _ffi()
    f.check(getattr(L, name)(f.current_stream(), *args))
    torch.cuda.synchronize()


@pytest.mark.parametrize("tile", [-1] + list(range(NTILES)))        # every entry of IA2P_GEMM_TILES (csrc/common.h)
@pytest.mark.parametrize("M,N,K", [(256, 320, 320), (2048, 1280, 1280), (616, 2560, 2048), (100, 64, 64), (37, 132, 128), (8192, 640, 2560)])
def test_gemm_bias_residual(L, M, N, K, tile):
    f = _ffi()
    A, W, b, R = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3), rnd(M, N, seed=4)
    out = torch.empty(M, N, dtype=torch.half, device="cuda")
    L.ia2p_debug_set_gemm_tile(tile)
    try:
        run(L, "ia2p_gemm", f.ptr(A), f.ptr(W), f.ptr(b), f.ptr(R), f.ptr(out), M, N, K, 0)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    ref = A.float() @ W.float().t() + b.float() + R.float()
    assert rel_l2(out, ref) < 1e-3, (M, N, K, tile, rel_l2(out, ref))
    # identity check with an asymmetric weight: catches a transposed accumulator layout
    if K == N:
        eye = torch.eye(K, dtype=torch.half, device="cuda")
        run(L, "ia2p_gemm", f.ptr(A), f.ptr(eye), None, None, f.ptr(out), M, N, K, 0)
        assert torch.equal(out, A)


@pytest.mark.parametrize("M,N,K", [(2048, 1280, 1280), (333, 200, 192)])
def test_gemm_tile_choice_never_changes_the_bits(L, M, N, K):
    """every tile variant accumulates each output element over the same 32-deep MFMA chunks in the same order"""
    f = _ffi()
    A, W, b, R = rnd(M, K, seed=51), rnd(N, K, seed=52, scale=K ** -0.5), rnd(N, seed=53), rnd(M, N, seed=54)
    outs = []
    try:
        for tile in range(NTILES):
            out = torch.empty(M, N, dtype=torch.half, device="cuda")
            L.ia2p_debug_set_gemm_tile(tile)
            run(L, "ia2p_gemm", f.ptr(A), f.ptr(W), f.ptr(b), f.ptr(R), f.ptr(out), M, N, K, 0)
            outs.append(out)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("M,N,K,S", [(256, 1280, 5120, 8), (256, 1280, 1280, 4), (2048, 1280, 11520, 3), (2048, 1280, 5120, 3), (77, 192, 640, 5), (300, 64, 128, 2)])
def test_gemm_splitk_deterministic(L, M, N, K, S):
    """A K split gives the same bits however it is finished: by the last-arriving K-slice inside the launch (ticket counters, own partial sums
    from its accumulators, the other slabs read back in slab order) or by the separate splitk_reduce_kernel launch; and again on a second run."""
    f = _ffi()
    A, W, b, R = rnd(M, K, seed=41), rnd(N, K, seed=42, scale=K ** -0.5), rnd(N, seed=43), rnd(M, N, seed=44)
    ref = A.float() @ W.float().t() + b.float() + R.float()
    part = torch.empty(S * M * N, dtype=torch.float32, device="cuda")
    outs = {}
    try:
        for route, limit in (("in-launch", 1 << 40), ("reduce launch", 0)):
            L.ia2p_debug_set_splitk_inkernel(limit)
            for tile in (-1, 0, 4, 8, 12, 18, 19, 20, 21, 22, 23):      # auto, 128x128, 64x64, 128x160 (two epilogue chunks), ping-pong 256x128 / 256x160 / 128x160, 32-row tiles, 8-phase 256x256 / 256x128
                L.ia2p_debug_set_gemm_tile(tile)
                out, out2 = torch.empty(M, N, dtype=torch.half, device="cuda"), torch.empty(M, N, dtype=torch.half, device="cuda")
                run(L, "ia2p_gemm_splitk", f.ptr(A), f.ptr(W), f.ptr(b), f.ptr(R), f.ptr(out), M, N, K, S, C.c_void_p(part.data_ptr()))
                assert rel_l2(out, ref) < 1e-3, (route, tile, rel_l2(out, ref))
                part.fill_(float("nan"))                 # slabs are fully rewritten; summation order is fixed -> same bits
                run(L, "ia2p_gemm_splitk", f.ptr(A), f.ptr(W), f.ptr(b), f.ptr(R), f.ptr(out2), M, N, K, S, C.c_void_p(part.data_ptr()))
                assert torch.equal(out, out2), (route, tile)
                outs[(route, tile)] = out
    finally:
        L.ia2p_debug_set_splitk_inkernel(-1)
        L.ia2p_debug_set_gemm_tile(-1)
    first = outs[("in-launch", -1)]
    for key, o in outs.items():
        assert torch.equal(o, first), key


def test_gemm_splitk_in_launch_back_to_back_reuses_slabs_and_counters(L):
    """stress of the in-launch hand-off: the same slab region and ticket counters serve launches with different data back to back (stale lines of
    the previous launch's slabs may sit in L1 / L2 of the reading CU); every result must equal the reduce-launch route bit for bit"""
    f = _ffi()
    M, N, K, S = 2048, 1280, 5120, 3
    part = torch.empty(S * M * N, dtype=torch.float32, device="cuda")
    W = rnd(N, K, seed=61, scale=K ** -0.5)
    As = [rnd(M, K, seed=70 + i) for i in range(6)]
    try:
        L.ia2p_debug_set_splitk_inkernel(0)
        refs = []
        for A in As:
            o = torch.empty(M, N, dtype=torch.half, device="cuda")
            run(L, "ia2p_gemm_splitk", f.ptr(A), f.ptr(W), None, None, f.ptr(o), M, N, K, S, C.c_void_p(part.data_ptr()))
            refs.append(o)
        L.ia2p_debug_set_splitk_inkernel(1 << 40)
        outs = [torch.empty(M, N, dtype=torch.half, device="cuda") for _ in As]
        for rep in range(5):
            for A, o in zip(As, outs):          # no synchronisation between the launches
                f.check(L.ia2p_gemm_splitk(f.current_stream(), f.ptr(A), f.ptr(W), None, None, f.ptr(o), M, N, K, S, C.c_void_p(part.data_ptr())))
            torch.cuda.synchronize()
            for i, (o, r) in enumerate(zip(outs, refs)):
                assert torch.equal(o, r), (rep, i)
    finally:
        L.ia2p_debug_set_splitk_inkernel(-1)


def test_gemm_no_bias_inplace_residual(L):
    f = _ffi()
    M, N, K = 512, 640, 640
    A, W = rnd(M, K, seed=5), rnd(N, K, seed=6, scale=K ** -0.5)
    X = rnd(M, N, seed=7)
    ref = A.float() @ W.float().t() + X.float()
    run(L, "ia2p_gemm", f.ptr(A), f.ptr(W), None, f.ptr(X), f.ptr(X), M, N, K, 0)   # residual aliases the output
    assert rel_l2(X, ref) < 1e-3


@pytest.mark.parametrize("tile", [-1, 0, 2, 4, 8, 10, 12, 18, 19, 20, 21])
@pytest.mark.parametrize("M,C", [(256, 128), (2048, 1280), (130, 640)])
def test_gemm_geglu(L, M, C, tile):
    f = _ffi()
    A, W, b = rnd(M, C, seed=8), rnd(8 * C, C, seed=9, scale=C ** -0.5), rnd(8 * C, seed=10, scale=0.1)
    Wp, bp = torch.empty_like(W), torch.empty_like(b)
    run(L, "ia2p_pack_geglu", f.ptr(W), f.ptr(Wp), 8 * C, C)
    run(L, "ia2p_pack_geglu", f.ptr(b), f.ptr(bp), 8 * C, 1)
    out = torch.empty(M, 4 * C, dtype=torch.half, device="cuda")
    L.ia2p_debug_set_gemm_tile(tile)
    try:
        run(L, "ia2p_gemm", f.ptr(A), f.ptr(Wp), f.ptr(bp), None, f.ptr(out), M, 8 * C, C, 1)
        if tile == 0:       # every tile width is a multiple of the 32-column (value, gate) block: all tiles give the same bits
            first = out.clone()
            for other in (6, 8, 11):
                L.ia2p_debug_set_gemm_tile(other)
                run(L, "ia2p_gemm", f.ptr(A), f.ptr(Wp), f.ptr(bp), None, f.ptr(out), M, 8 * C, C, 1)
                assert torch.equal(out, first), other
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    h = A.float() @ W.float().t() + b.float()
    a, g = h.chunk(2, dim=-1)
    ref = a * F.gelu(g)                                   # exact-erf GELU
    assert rel_l2(out, ref) < 1.5e-3, rel_l2(out, ref)


def _ln_fold_setup(L, M, C_, N, seed, bias=True):
    """x -> (producer GEMM with stats) -> t; then LayerNorm(t) . W^T + b through the folded path"""
    f = _ffi()
    X, Wp, R = rnd(M, C_, seed=seed), rnd(C_, C_, seed=seed + 1, scale=C_ ** -0.5), rnd(M, C_, seed=seed + 2, scale=2.0)
    gamma, beta = (1.0 + 0.3 * torch.randn(C_, generator=torch.Generator().manual_seed(seed + 3))).half().cuda(), rnd(C_, seed=seed + 4, scale=0.2)
    W, b = rnd(N, C_, seed=seed + 5, scale=C_ ** -0.5), (rnd(N, seed=seed + 6, scale=0.3) if bias else None)
    Wf = torch.empty_like(W)
    cs, fb = torch.empty(N, dtype=torch.float32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda")
    run(L, "ia2p_fold_layernorm", f.ptr(W), f.ptr(gamma), f.ptr(beta), f.ptr(b), f.ptr(Wf), f.ptr(cs), f.ptr(fb), N, C_)
    return f, X, Wp, R, gamma, beta, W, b, Wf, cs, fb


@pytest.mark.parametrize("tile", [-1, 0, 4, 6, 8, 11, 12, 20, 21])
@pytest.mark.parametrize("M,C_,N,psplit,csplit", [(2048, 1280, 3840, 1, 1), (300, 256, 768, 1, 1), (256, 1280, 1280, 3, 2), (77, 128, 132, 2, 1)])
def test_layernorm_folded_into_gemm(L, M, C_, N, psplit, csplit, tile):
    """producer GEMM emits row statistics, consumer GEMM reads the raw rows against gamma-folded weights: equals
    LayerNorm(eps 1e-5) -> Linear in torch, for every tile shape and with either side K-split"""
    f, X, Wp, R, gamma, beta, W, b, Wf, cs, fb = _ln_fold_setup(L, M, C_, N, seed=60 + M % 7)
    t = torch.empty(M, C_, dtype=torch.half, device="cuda")
    stats = torch.zeros((C_ // 64 + 1) * M * 2, dtype=torch.float32, device="cuda")
    part = torch.empty(4 * M * max(N, C_), dtype=torch.float32, device="cuda")
    slots = C.c_int(0)
    L.ia2p_debug_set_gemm_tile(tile)
    try:
        run(L, "ia2p_gemm_ex", f.ptr(X), f.ptr(Wp), None, f.ptr(R), f.ptr(t), M, C_, C_, 0, None, f.ptr(stats), C.addressof(slots), psplit, f.ptr(part))
        tf = t.float()
        st = stats.view(-1, M, 2)[:slots.value].sum(0)
        assert slots.value >= 1 and torch.allclose(st[:, 0], tf.sum(1), rtol=1e-4, atol=1e-2) and torch.allclose(st[:, 1], (tf * tf).sum(1), rtol=1e-4, atol=1e-2)
        ln = f.LnFoldC(stats.data_ptr(), slots.value, cs.data_ptr(), fb.data_ptr(), 1e-5)
        out = torch.empty(M, N, dtype=torch.half, device="cuda")
        run(L, "ia2p_gemm_ex", f.ptr(t), f.ptr(Wf), None, None, f.ptr(out), M, N, C_, 0, C.addressof(ln), None, None, csplit, f.ptr(part))
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    ref = F.layer_norm(tf, (C_,), gamma.float(), beta.float(), 1e-5) @ W.float().t() + b.float()
    assert rel_l2(out, ref) < 1.5e-3, rel_l2(out, ref)
    # the separate-kernel path (fp16 LayerNorm output, then the plain GEMM) is no closer to fp32 than the folded one
    y = torch.empty_like(t)
    run(L, "ia2p_layernorm", f.ptr(t), f.ptr(y), f.ptr(gamma), f.ptr(beta), M, C_, 1e-5)
    sep = torch.empty_like(out)
    run(L, "ia2p_gemm", f.ptr(y), f.ptr(W), f.ptr(b), None, f.ptr(sep), M, N, C_, 0)
    assert rel_l2(out, ref) <= rel_l2(sep, ref) * 1.2 + 1e-4


def test_layernorm_fold_bits_do_not_depend_on_the_tile(L):
    """mean and rstd of the folded LayerNorm are derived inside every tile instantiation from the same {sum, sum of squares}: the roundings are spelled out
    (common.h ln_mean_rstd_f), so -- like the plain GEMM -- the folded output is the same bits on every tile (round 3 left `s2/K - mean^2` to the compiler's
    contraction and two instantiations differed in the last bit)."""
    f, X, Wp, R, gamma, beta, W, b, Wf, cs, fb = _ln_fold_setup(L, 1024, 1280, 1920, seed=91)
    M, C_, N = 1024, 1280, 1920
    t = (X.float() * 3 + 0.5).half()
    tf = t.float()
    slots = C_ // 64
    st = torch.stack([tf.view(M, slots, 64).sum(2), (tf * tf).view(M, slots, 64).sum(2)], dim=2).permute(1, 0, 2).contiguous()
    ln = f.LnFoldC(st.data_ptr(), slots, cs.data_ptr(), fb.data_ptr(), 1e-5)
    outs = []
    try:
        for tile in range(NTILES):
            out = torch.empty(M, N, dtype=torch.half, device="cuda")
            L.ia2p_debug_set_gemm_tile(tile)
            run(L, "ia2p_gemm_ex", f.ptr(t), f.ptr(Wf), None, None, f.ptr(out), M, N, C_, 0, C.addressof(ln), None, None, 1, None)
            outs.append(out)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    for i, o in enumerate(outs[1:]):
        assert torch.equal(o, outs[0]), i + 1


@pytest.mark.parametrize("M,C_,ln", [(2048, 640, True), (512, 1280, True), (256, 640, False), (1024, 320, True)])
def test_geglu_projection_on_the_256x320_tile(L, M, C_, ln):
    """Round 6: `ff.net.0` (GEGLU projection, norm3 folded in; diffusers FeedForward / GEGLU behind pnp_pipeline.py:253-260, in-tree twin
    llm/model/vae/modules/attention.py:37-44) on tile variant 27 -- 256 x 320, 8 waves of 64 x 160, ping-pong on 32-deep sub-steps over two k-tile slots, the epilogue in
    two column halves -- against fp32 torch and BIT-IDENTICAL to the other GEGLU-capable tiles (128 x 160, 256 x 160 ping-pong, 128 x 160 ping-pong, 8-phase 256 x 256):
    same 32-deep accumulation order, same epilogue formulas. With and without the folded LayerNorm (plain bias); K from 5 to 20 k-tiles."""
    f, X, Wp, R, gamma, beta, W, b, _, _, _ = _ln_fold_setup(L, M, C_, 8 * C_, seed=170 + C_ // 64)
    N = 8 * C_
    Wpk, bpk = torch.empty_like(W), torch.empty_like(b)
    run(L, "ia2p_pack_geglu", f.ptr(W), f.ptr(Wpk), N, C_)
    run(L, "ia2p_pack_geglu", f.ptr(b), f.ptr(bpk), N, 1)
    t = (X.float() * 2 + 0.25).half()
    if ln:
        Wf = torch.empty_like(W)
        cs, fb = torch.empty(N, dtype=torch.float32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda")
        run(L, "ia2p_fold_layernorm", f.ptr(Wpk), f.ptr(gamma), f.ptr(beta), f.ptr(bpk), f.ptr(Wf), f.ptr(cs), f.ptr(fb), N, C_)
        tf = t.float()
        slots = C_ // 64
        st = torch.stack([tf.view(M, slots, 64).sum(2), (tf * tf).view(M, slots, 64).sum(2)], dim=2).permute(1, 0, 2).contiguous()
        lnc = f.LnFoldC(st.data_ptr(), slots, cs.data_ptr(), fb.data_ptr(), 1e-5)
        h = F.layer_norm(t.float(), (C_,), gamma.float(), beta.float(), 1e-5) @ W.float().t() + b.float()
    else:
        h = t.float() @ W.float().t() + b.float()
    a, g = h.chunk(2, dim=-1)
    ref = a * F.gelu(g)
    info = (C.c_int * 4)()
    assert L.ia2p_debug_gemm_tile_info(27, info) == 0 and tuple(info) == (256, 320, 2, 4)
    outs = {}
    try:
        for tile in (27, 8, 18, 19, 22):
            out = torch.full((M, N // 2), float("nan"), dtype=torch.half, device="cuda")
            L.ia2p_debug_set_gemm_tile(tile)
            v, sk = C.c_int(-1), C.c_int(-1)
            L.ia2p_debug_gemm_plan(M, N, C_, 0, 1, C.addressof(v), C.addressof(sk))
            assert (v.value, sk.value) == (tile, 1)                      # the forced tile is what runs (whole tiles: the GEGLU tile takes the launch)
            if ln:
                run(L, "ia2p_gemm_ex", f.ptr(t), f.ptr(Wf), None, None, f.ptr(out), M, N, C_, 1, C.addressof(lnc), None, None, 1, None)
            else:
                run(L, "ia2p_gemm_ex", f.ptr(t), f.ptr(Wpk), f.ptr(bpk), None, f.ptr(out), M, N, C_, 1, None, None, None, 1, None)
            outs[tile] = out
        # a launch the tile does not take (rows not a multiple of 256): the force is ignored, the plan falls back to the table / the cost model
        L.ia2p_debug_set_gemm_tile(27)
        v = C.c_int(-1)
        L.ia2p_debug_gemm_plan(130, N, C_, 0, 1, C.addressof(v), None)
        assert v.value != 27 and v.value >= 0
        L.ia2p_debug_gemm_plan(M, N, C_, 0, 0, C.addressof(v), None)       # ... and so is a non-GEGLU launch of the same shape
        assert v.value != 27 and v.value >= 0
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    assert torch.isfinite(outs[27]).all()
    assert rel_l2(outs[27], ref) < 2e-3, rel_l2(outs[27], ref)
    for tile, o in outs.items():
        assert torch.equal(o, outs[27]), (tile, int((o != outs[27]).sum()))


@pytest.mark.parametrize("M,N,K", [(256, 640, 64), (512, 320, 128), (256, 960, 192)])
def test_geglu_tile_shortest_k_loops(L, M, N, K):
    """the 256 x 320 GEGLU tile with ONE, two and three k-tiles (its k-loop peels the first sub-step -- the MFMAs that define the accumulators -- and the last one: the
    shortest loops run only peeled code), plain bias: same bits as the 128 x 160 tile, fp32 torch within tolerance"""
    f = _ffi()
    A, W, b = rnd(M, K, seed=201), rnd(N, K, seed=202, scale=K ** -0.5), rnd(N, seed=203, scale=0.3)
    Wpk, bpk = torch.empty_like(W), torch.empty_like(b)
    run(L, "ia2p_pack_geglu", f.ptr(W), f.ptr(Wpk), N, K)
    run(L, "ia2p_pack_geglu", f.ptr(b), f.ptr(bpk), N, 1)
    outs = {}
    try:
        for tile in (27, 8):
            out = torch.full((M, N // 2), float("nan"), dtype=torch.half, device="cuda")
            L.ia2p_debug_set_gemm_tile(tile)
            v = C.c_int(-1)
            L.ia2p_debug_gemm_plan(M, N, K, 0, 1, C.addressof(v), None)
            assert v.value == tile
            run(L, "ia2p_gemm_ex", f.ptr(A), f.ptr(Wpk), f.ptr(bpk), None, f.ptr(out), M, N, K, 1, None, None, None, 1, None)
            outs[tile] = out
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    h = A.float() @ W.float().t() + b.float()
    a, g = h.chunk(2, dim=-1)
    assert torch.isfinite(outs[27]).all() and rel_l2(outs[27], a * F.gelu(g)) < 2e-3
    assert torch.equal(outs[27], outs[8])


def test_geglu_tile_gate_range_and_special_values_match_the_family(L):
    """the in-register GEGLU of the 256 x 320 tile on gates anywhere in [-12, 12] (exactly +-8, the table's last cell, beyond it) and on NaN / +-Inf in a value or a gate:
    the same bits as the 128 x 160 tile, which sends the projected tile through the LDS (test_geglu_gate_table_range_and_special_values pins that route to exact-erf GELU);
    NaN / Inf never come out as a finite number. One row of A selects one (value, gate) pair."""
    f = _ffi()
    M, N, K = 256, 320, 64
    gates = torch.tensor([-12.0, -9.5, -8.0, -7.99, -7.5, -4.0, -1.0, -0.03125, 0.0, 0.015625, 1.0, 3.0, 6.0, 7.96875, 7.99, 8.0, 9.5, 12.0])
    gates = torch.cat([gates, torch.linspace(-12, 12, 64 - len(gates))])
    vals = torch.linspace(-3.0, 3.0, 64)
    A = torch.eye(64, K).repeat(4, 1).half().cuda()                            # row m selects k = m % 64
    W = torch.zeros(N, K)
    for blk in range(N // 32):
        W[blk * 32:blk * 32 + 16, :] = vals
        W[blk * 32 + 16:blk * 32 + 32, :] = gates
    bias = torch.zeros(N).half().cuda()
    for special, row, col in ((None, 0, 0), (float("nan"), 16, 5), (float("nan"), 0, 5), (float("inf"), 16, 7), (float("-inf"), 16, 9), (float("inf"), 0, 11), (float("nan"), 9 * 32 + 31, 63)):
        W2 = W.clone()
        if special is not None:
            W2[row, col] = special
        W2 = W2.half().cuda()
        outs = {}
        try:
            for tile in (27, 8):
                out = torch.zeros(M, N // 2, dtype=torch.half, device="cuda")
                L.ia2p_debug_set_gemm_tile(tile)
                run(L, "ia2p_gemm", f.ptr(A), f.ptr(W2), f.ptr(bias), None, f.ptr(out), M, N, K, 1)
                outs[tile] = out
        finally:
            L.ia2p_debug_set_gemm_tile(-1)
        a, b = outs[27], outs[8]
        assert torch.equal(torch.isnan(a), torch.isnan(b)) and torch.equal(torch.nan_to_num(a.float(), nan=7.0), torch.nan_to_num(b.float(), nan=7.0)), (special, row, col)
        if special is not None:
            oc = (row // 32) * 16 + (row % 16)                                  # the output column the touched value / gate row feeds
            hit = a[col::64, oc].float()
            assert not torch.isfinite(hit).any(), (special, row, col, hit)      # NaN / Inf propagate (Inf in a far-negative... the rows chosen have finite non-zero partners)


@pytest.mark.parametrize("M,C_", [(2048, 640), (130, 128)])
def test_layernorm_folded_into_geglu(L, M, C_):
    f, X, Wp, R, gamma, beta, W, b, _, _, _ = _ln_fold_setup(L, M, C_, 8 * C_, seed=70)
    Wpk, bpk = torch.empty_like(W), torch.empty_like(b)
    run(L, "ia2p_pack_geglu", f.ptr(W), f.ptr(Wpk), 8 * C_, C_)
    run(L, "ia2p_pack_geglu", f.ptr(b), f.ptr(bpk), 8 * C_, 1)
    Wf = torch.empty_like(W)
    cs, fb = torch.empty(8 * C_, dtype=torch.float32, device="cuda"), torch.empty(8 * C_, dtype=torch.float32, device="cuda")
    run(L, "ia2p_fold_layernorm", f.ptr(Wpk), f.ptr(gamma), f.ptr(beta), f.ptr(bpk), f.ptr(Wf), f.ptr(cs), f.ptr(fb), 8 * C_, C_)   # row-wise: packing first is fine
    t = torch.empty(M, C_, dtype=torch.half, device="cuda")
    stats = torch.zeros((C_ // 64 + 1) * M * 2, dtype=torch.float32, device="cuda")
    slots = C.c_int(0)
    run(L, "ia2p_gemm_ex", f.ptr(X), f.ptr(Wp), None, f.ptr(R), f.ptr(t), M, C_, C_, 0, None, f.ptr(stats), C.addressof(slots), 1, None)
    ln = f.LnFoldC(stats.data_ptr(), slots.value, cs.data_ptr(), fb.data_ptr(), 1e-5)
    out = torch.empty(M, 4 * C_, dtype=torch.half, device="cuda")
    run(L, "ia2p_gemm_ex", f.ptr(t), f.ptr(Wf), None, None, f.ptr(out), M, 8 * C_, C_, 1, C.addressof(ln), None, None, 1, None)
    h = F.layer_norm(t.float(), (C_,), gamma.float(), beta.float(), 1e-5) @ W.float().t() + b.float()
    a, g = h.chunk(2, dim=-1)
    assert rel_l2(out, a * F.gelu(g)) < 2e-3
    with pytest.raises(ValueError):
        f.check(L.ia2p_gemm_ex(f.current_stream(), f.ptr(t), f.ptr(Wf), None, None, f.ptr(out), M, 8 * C_, C_, 1, None, None, None, 1, None))   # GEGLU without bias or fold


@pytest.mark.parametrize("tile", [-1, 0, 12, 18, 22])
@pytest.mark.parametrize("ratio", [10.0, 50.0])
@pytest.mark.parametrize("geglu", [0, 1])
def test_layernorm_fold_on_real_weight_shaped_rows(L, ratio, tile, geglu):
    """The folded LayerNorm computes rstd * (acc - mean * colsum) from {sum x, sum x^2}: both steps cancel when |mean| >> std. Rows like a trained SDXL residual
    stream's: per-row mean = ratio x the row's std (both signs), four x 100 outlier channels, a quarter of the rows with all three. Against fp32
    LayerNorm -> Linear (-> GEGLU) in torch at the bound of the tame test, and no worse than the explicit fp16 LayerNorm kernel + plain GEMM."""
    f = _ffi()
    M, C_ = 1024, 1280
    N = 8 * C_ if geglu else 3 * C_
    g = torch.Generator().manual_seed(int(ratio) + 7 * geglu)
    t = torch.randn(M, C_, generator=g)
    sign = torch.where(torch.rand(M, 1, generator=g) < 0.5, -1.0, 1.0)
    t[: M // 2] += ratio * sign[: M // 2]                                  # rows with |mean| / std = ratio
    out_ch = torch.tensor([3, 640, 641, 1279])
    t[M // 4: 3 * M // 4, out_ch] *= 100.0                                  # outlier channels (rows M/4 .. M/2 have both)
    t = t.half().cuda()
    gamma = (1.0 + 0.3 * torch.randn(C_, generator=g)).half().cuda()
    beta = (0.2 * torch.randn(C_, generator=g)).half().cuda()
    W = (torch.randn(N, C_, generator=g) * C_ ** -0.5).half().cuda()
    b = (0.3 * torch.randn(N, generator=g)).half().cuda()
    if geglu:
        Wp, bp = torch.empty_like(W), torch.empty_like(b)
        run(L, "ia2p_pack_geglu", f.ptr(W), f.ptr(Wp), N, C_)
        run(L, "ia2p_pack_geglu", f.ptr(b), f.ptr(bp), N, 1)
    else:
        Wp, bp = W, b
    Wf = torch.empty_like(W)
    cs, fb = torch.empty(N, dtype=torch.float32, device="cuda"), torch.empty(N, dtype=torch.float32, device="cuda")
    run(L, "ia2p_fold_layernorm", f.ptr(Wp), f.ptr(gamma), f.ptr(beta), f.ptr(bp), f.ptr(Wf), f.ptr(cs), f.ptr(fb), N, C_)
    tf = t.float()
    # row statistics as a producer epilogue writes them: {sum, sum of squares} of the fp16 row over 64-column slots, fp32
    slots = C_ // 64
    st = torch.stack([tf.view(M, slots, 64).sum(2), (tf * tf).view(M, slots, 64).sum(2)], dim=2).permute(1, 0, 2).contiguous()      # [slot][M][2]
    ln = f.LnFoldC(st.data_ptr(), slots, cs.data_ptr(), fb.data_ptr(), 1e-5)
    No = N // 2 if geglu else N
    out = torch.empty(M, No, dtype=torch.half, device="cuda")
    L.ia2p_debug_set_gemm_tile(tile)
    try:
        run(L, "ia2p_gemm_ex", f.ptr(t), f.ptr(Wf), None, None, f.ptr(out), M, N, C_, geglu, C.addressof(ln), None, None, 1, None)
        y = torch.empty_like(t)
        run(L, "ia2p_layernorm", f.ptr(t), f.ptr(y), f.ptr(gamma), f.ptr(beta), M, C_, 1e-5)
        sep = torch.empty_like(out)
        run(L, "ia2p_gemm", f.ptr(y), f.ptr(Wp), f.ptr(bp), None, f.ptr(sep), M, N, C_, geglu)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    h = F.layer_norm(tf.double(), (C_,), gamma.double(), beta.double(), 1e-5) @ W.double().t() + b.double()
    if geglu:
        a_, g_ = h.chunk(2, dim=-1)
        h = a_ * F.gelu(g_)
    ref = h.float()
    assert torch.isfinite(out).all()
    e_fold, e_sep = rel_l2(out, ref), rel_l2(sep, ref)
    groups = {"shifted": slice(0, M // 4), "shifted + outliers": slice(M // 4, M // 2), "outliers": slice(M // 2, 3 * M // 4), "plain": slice(3 * M // 4, M)}
    per = {k: rel_l2(out[v], ref[v]) for k, v in groups.items()}
    assert e_fold < 2e-3, (ratio, tile, geglu, e_fold, per)
    assert max(per.values()) < 3e-3, per                                   # every kind of row by itself, not just the average
    assert e_fold <= 1.2 * e_sep + 1e-4, (e_fold, e_sep)


def test_geglu_gate_table_range_and_special_values(L):
    """The GEGLU gate goes through a normal-CDF table on [-8, 8) (csrc/common.h gelu_lut_f). Gates anywhere in [-12, 12] -- exactly +-8, the table's last cell,
    beyond it on both sides -- and values times such gates against exact-erf GELU; NaN / +Inf in the value or the gate come out as NaN / Inf, never as a
    finite number. One row of A selects one (value, gate) pair: W = identity-like rows, so acc = the chosen numbers exactly."""
    f = _ffi()
    K, M = 64, 64
    gates = torch.tensor([-12.0, -9.5, -8.0, -7.99, -7.5, -4.0, -1.0, -0.03125, 0.0, 0.015625, 1.0, 3.0, 6.0, 7.96875, 7.99, 8.0, 9.5, 12.0])
    gates = torch.cat([gates, torch.linspace(-12, 12, 64 - len(gates))])
    vals = torch.linspace(-3.0, 3.0, 64)
    # A[m] = e_m (one-hot, exact); W rows: 32 packed columns per block = [16 values | 16 gates]; value row j and gate row j both pick k = j
    A = torch.eye(M, K).half().cuda()
    N = 8 * 16                                                               # 4 blocks of (16 values + 16 gates)
    W = torch.zeros(N, K)
    bias = torch.zeros(N)
    for blk in range(4):
        for j in range(16):
            c = blk * 16 + j                                                  # output column c gets value vals[m] * scale_c and gate gates[m]
            W[blk * 32 + j, :] = vals                                        # value row: acc = vals[m]
            W[blk * 32 + 16 + j, :] = gates                                  # gate row:  acc = gates[m]
    W = W.half().cuda()
    bias = bias.half().cuda()
    out = torch.empty(M, N // 2, dtype=torch.half, device="cuda")
    for tile in (0, 18, 22):
        L.ia2p_debug_set_gemm_tile(tile)
        try:
            run(L, "ia2p_gemm", f.ptr(A), f.ptr(W), f.ptr(bias), None, f.ptr(out), M, N, K, 1)
        finally:
            L.ia2p_debug_set_gemm_tile(-1)
        v, g_ = W[0].double().cpu(), W[16].double().cpu()                  # as rounded to fp16
        ref = (v * F.gelu(g_)).float()
        got = out[:, 0].float().cpu()
        assert torch.equal(out, out[:, :1].expand_as(out)), tile              # every column computes the same thing
        err = (got - ref).abs()
        assert float((err / (ref.abs() + 1e-3)).max()) < 2e-3, (tile, got, ref)
        assert float(got[g_ <= -9.0].abs().max()) == 0.0                      # far negative gates switch the value off entirely
    # special values: NaN / Inf must propagate
    for special, where in ((float("nan"), "gate"), (float("nan"), "value"), (float("inf"), "gate"), (float("inf"), "value")):
        W2 = W.clone()
        row = 16 if where == "gate" else 0
        W2[row, 5] = special                                                 # A row 5 selects k = 5
        run(L, "ia2p_gemm", f.ptr(A), f.ptr(W2), f.ptr(bias), None, f.ptr(out), M, N, K, 1)
        got = out[5, 0].float().item()
        v5 = float("nan") if (where == "value" and special != special) else (special if where == "value" else float(W[0, 5]))
        g5 = special if where == "gate" else float(W[16, 5])
        want = (torch.tensor(v5, dtype=torch.float64) * F.gelu(torch.tensor(g5, dtype=torch.float64))).item()
        if want != want:
            assert got != got, (special, where, got)
        else:
            assert got == want or (abs(want) == float("inf") and got == want), (special, where, got, want)
        assert torch.isfinite(out[:, 1:].float()).all()                       # only the output column fed by the poisoned weight row is affected


def test_gemm_splitk_two_streams_run_concurrently(L):
    """K-split launches of two streams overlap on the device (each stream has its own ticket buffer): 60 interleaved launches per stream of two different
    problems -- one of them the FF-out shape of the step -- give, launch for launch, the bits of the same launches run alone."""
    f = _ffi()
    probs = [(2048, 1280, 5120, 3, 0), (1024, 640, 2560, 4, 12)]           # (M, N, K, split, tile)
    data, alone = [], []
    for i, (M, N, K, S, tile) in enumerate(probs):
        A, W, b, R = rnd(M, K, seed=91 + i), rnd(N, K, seed=93 + i, scale=K ** -0.5), rnd(N, seed=95 + i), rnd(M, N, seed=97 + i)
        part = torch.empty(S * M * N, dtype=torch.float32, device="cuda")
        out = torch.empty(M, N, dtype=torch.half, device="cuda")
        data.append((A, W, b, R, part, out))
        L.ia2p_debug_set_gemm_tile(tile)
        run(L, "ia2p_gemm_splitk", f.ptr(A), f.ptr(W), f.ptr(b), f.ptr(R), f.ptr(out), M, N, K, S, C.c_void_p(part.data_ptr()))
        alone.append(out.clone())
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    bad = []
    try:
        for rep in range(60):
            for i, (M, N, K, S, tile) in enumerate(probs):
                A, W, b, R, part, out = data[i]
                with torch.cuda.stream(streams[i]):
                    if rep % 10 == 0:
                        out.fill_(float("nan"))
                    L.ia2p_debug_set_gemm_tile(tile)
                    f.check(L.ia2p_gemm_splitk(C.c_void_p(streams[i].cuda_stream), f.ptr(A), f.ptr(W), f.ptr(b), f.ptr(R), f.ptr(out), M, N, K, S, C.c_void_p(part.data_ptr())))
            if rep % 10 == 9:
                torch.cuda.synchronize()
                bad += [(rep, i) for i in range(2) if not torch.equal(data[i][5], alone[i])]
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    torch.cuda.synchronize()
    assert not bad, bad


@pytest.mark.parametrize("tile", [-1, 0, 4, 12, 18, 22])
def test_epilogue_rounds_where_the_reference_does(L, tile):
    """The reference's fp16 modules round a Linear's output (bias included) to fp16 BEFORE the residual is added, and round the sum again. With an identity weight the
    accumulators are exact, so the two rounding points are checked to the bit -- on the register epilogue (no K split), on the K-split route finished inside the launch and
    on the one finished by the reduce launch; a single rounding of (x + b + r) differs from this in a few per cent of the elements at these magnitudes."""
    f = _ffi()
    M, N = 300, 256
    A, b, R = rnd(M, N, seed=81), rnd(N, seed=82), rnd(M, N, seed=83)
    eye = torch.eye(N, dtype=torch.half, device="cuda")
    want = ((A.float() + b.float()).half().float() + R.float()).half()
    once = (A.float() + b.float() + R.float()).half()
    assert not torch.equal(want, once)                      # the test can tell the two apart
    part = torch.empty(2 * M * N, dtype=torch.float32, device="cuda")
    L.ia2p_debug_set_gemm_tile(tile)
    try:
        out = torch.empty(M, N, dtype=torch.half, device="cuda")
        run(L, "ia2p_gemm", f.ptr(A), f.ptr(eye), f.ptr(b), f.ptr(R), f.ptr(out), M, N, N, 0)
        assert torch.equal(out, want), "register epilogue"
        for route, limit in (("in-launch", 1 << 40), ("reduce launch", 0)):
            L.ia2p_debug_set_splitk_inkernel(limit)
            out = torch.empty(M, N, dtype=torch.half, device="cuda")
            run(L, "ia2p_gemm_splitk", f.ptr(A), f.ptr(eye), f.ptr(b), f.ptr(R), f.ptr(out), M, N, N, 2, C.c_void_p(part.data_ptr()))
            assert torch.equal(out, want), route
    finally:
        L.ia2p_debug_set_splitk_inkernel(-1)
        L.ia2p_debug_set_gemm_tile(-1)


def test_gemm_splitk_recovers_from_poisoned_tickets(L):
    """A launch that dies mid-flight leaves tickets behind: the next launch's "last arriver" of a tile is then an early one and combines slabs that are not written
    yet. Poisoned here the same way (every ticket of the stream's buffer set to splitk - 1, through ia2p_debug_fill_splitk_counters): the launch after it is WRONG, which
    is the failure the epoch guards against; ia2p_debug_invalidate_splitk_counters() -- what the library calls on IA2P_ERR_HIP and on context creation -- puts a
    re-zeroing memset on the stream in front of the next K-split launch, and that launch and the one after it are right again."""
    f = _ffi()
    M, N, K, S = 2048, 1280, 5120, 8          # 160 tiles x 8 slices = 1280 workgroups: more than are resident at once, so the first slices finish before the last start
    A, W, b, R = rnd(M, K, seed=71), rnd(N, K, seed=72, scale=K ** -0.5), rnd(N, seed=73), rnd(M, N, seed=74)
    part = torch.empty(S * M * N, dtype=torch.float32, device="cuda")
    ref = A.float() @ W.float().t() + b.float() + R.float()
    st = torch.cuda.Stream()
    sp = C.c_void_p(st.cuda_stream)
    out = torch.empty(M, N, dtype=torch.half, device="cuda")

    def launch():
        out.fill_(float("nan"))
        part.fill_(float("nan"))
        torch.cuda.synchronize()
        f.check(L.ia2p_gemm_splitk(sp, f.ptr(A), f.ptr(W), f.ptr(b), f.ptr(R), f.ptr(out), M, N, K, S, C.c_void_p(part.data_ptr())))
        st.synchronize()
        return out.clone()

    L.ia2p_debug_set_gemm_tile(0)
    try:
        good = launch()
        assert rel_l2(good, ref) < 1e-3
        assert L.ia2p_debug_fill_splitk_counters(sp, S - 1) == 0      # the FIRST slice to arrive draws "last": it combines slabs nobody has written (NaN here)
        bad = launch()
        assert not torch.equal(bad, good)                             # the hazard is real
        assert L.ia2p_debug_fill_splitk_counters(sp, S - 1) == 0      # (that launch left its own mess; poison again to a known state)
        L.ia2p_debug_invalidate_splitk_counters()
        assert torch.equal(launch(), good) and torch.equal(launch(), good)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)


def test_gemm_rejects_bad_shapes(L):
    f = _ffi()
    A, W, out = rnd(64, 96), rnd(64, 96), torch.empty(64, 64, dtype=torch.half, device="cuda")
    with pytest.raises(ValueError):
        f.check(L.ia2p_gemm(f.current_stream(), f.ptr(A), f.ptr(W), None, None, f.ptr(out), 64, 64, 96, 0))   # K % 64
    with pytest.raises(ValueError):
        f.check(L.ia2p_gemm(f.current_stream(), None, f.ptr(W), None, None, f.ptr(out), 64, 64, 64, 0))


@pytest.mark.parametrize("B,H,W,Cin,Co,stride,up", [
    (2, 16, 16, 64, 128, 1, 0), (1, 32, 32, 320, 320, 1, 0), (2, 16, 16, 128, 128, 2, 0), (2, 8, 8, 128, 64, 1, 1),
    (1, 12, 20, 64, 64, 1, 0), (1, 10, 14, 64, 64, 2, 0), (8, 16, 16, 1280, 1280, 1, 0)])
def test_conv3x3(L, B, H, W, Cin, Co, stride, up):
    f = _ffi()
    x = rnd(B, H, W, Cin, seed=11)                                  # channels-last
    w = rnd(Co, Cin, 3, 3, seed=12, scale=(9 * Cin) ** -0.5)
    b, tv = rnd(Co, seed=13), rnd(B, Co, seed=14)
    wp = torch.empty(Co, 9 * Cin, dtype=torch.half, device="cuda")
    run(L, "ia2p_pack_conv3x3", f.ptr(w), f.ptr(wp), Co, Cin)
    xn = x.permute(0, 3, 1, 2).float()
    if up:
        xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xn, w.float(), b.float(), stride=stride, padding=1) + tv.float()[:, :, None, None]
    Ho, Wo = ref.shape[-2:]
    res = rnd(B, Ho, Wo, Co, seed=15)
    ref = (ref + res.permute(0, 3, 1, 2).float()).permute(0, 2, 3, 1)
    y = torch.empty(B, Ho, Wo, Co, dtype=torch.half, device="cuda")
    run(L, "ia2p_conv3x3", f.ptr(x), f.ptr(wp), f.ptr(b), f.ptr(tv), f.ptr(res), f.ptr(y), B, H, W, Cin, Co, stride, up)
    assert rel_l2(y, ref) < 1e-3, rel_l2(y, ref)


@pytest.mark.parametrize("tile", list(range(NTILES)))
@pytest.mark.parametrize("B,H,W,Cin,Co,stride,up", [(1, 32, 32, 320, 320, 1, 0), (2, 12, 20, 128, 192, 2, 0), (1, 10, 14, 64, 320, 1, 1), (2, 8, 16, 128, 160, 1, 1)])      # (the last one: an upsampled view in whole 16 x 16 patches -- the halo-staged tiles take it)
def test_conv3x3_every_tile(L, tile, B, H, W, Cin, Co, stride, up):
    L.ia2p_debug_set_gemm_tile(tile)
    try:
        test_conv3x3(L, B, H, W, Cin, Co, stride, up)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)


@pytest.mark.parametrize("B,Cin,H,W,Co", [(8, 4, 64, 64, 320), (1, 4, 128, 128, 320), (2, 4, 13, 9, 64), (1, 3, 40, 24, 128), (1, 7, 5, 5, 8), (3, 4, 16, 16, 384)])
def test_conv_in_boundary(L, B, Cin, H, W, Co):
    """latent-boundary conv (NCHW in, channels-last out): every pixel count incl. ones that do not fill a 16-pixel wave tile"""
    f = _ffi()
    x, w, b = rnd(B, Cin, H, W, seed=41), rnd(Co, Cin, 3, 3, seed=42, scale=(Cin * 9) ** -0.5), rnd(Co, seed=43, scale=0.2)
    y = torch.full((B * H * W, Co), float("nan"), dtype=torch.half, device="cuda")
    ws = torch.empty(Co * 64, dtype=torch.half, device="cuda")
    run(L, "ia2p_conv_in", f.ptr(x), f.ptr(w), f.ptr(b), f.ptr(y), f.ptr(ws), B, Cin, H, W, Co)
    ref = F.conv2d(x.float(), w.float(), b.float(), padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Co)
    assert rel_l2(y, ref) < 1e-3, rel_l2(y, ref)


@pytest.mark.parametrize("B,C_,H,W,Co", [(8, 320, 64, 64, 4), (1, 320, 128, 128, 4), (2, 128, 13, 9, 3), (1, 384, 24, 40, 4), (1, 512, 8, 8, 8), (2, 64, 5, 7, 1), (1, 96, 16, 16, 4)])
def test_conv_out_boundary(L, B, C_, H, W, Co):
    """latent-boundary conv (channels-last in, NCHW out): the unrolled channel counts (128 / 320 / 384 / 512) and the generic loop"""
    f = _ffi()
    x, w, b = rnd(B * H * W, C_, seed=44), rnd(Co, C_, 3, 3, seed=45, scale=(C_ * 9) ** -0.5), rnd(Co, seed=46, scale=0.2)
    wp = torch.empty(Co * 9 * C_, dtype=torch.half, device="cuda")
    run(L, "ia2p_pack_conv_out", f.ptr(w), f.ptr(wp), Co, C_)
    y = torch.full((B, Co, H, W), float("nan"), dtype=torch.half, device="cuda")
    run(L, "ia2p_conv_out", f.ptr(x), f.ptr(wp), f.ptr(b), f.ptr(y), B, C_, H, W, Co)
    ref = F.conv2d(x.float().reshape(B, H, W, C_).permute(0, 3, 1, 2), w.float(), b.float(), padding=1)
    assert rel_l2(y, ref) < 1e-3, rel_l2(y, ref)


def test_boundary_convs_reject_bad_shapes(L):
    f = _ffi()
    x = rnd(1, 8, 4, 4, seed=1)
    assert L.ia2p_conv_in(f.current_stream(), f.ptr(x), f.ptr(x), f.ptr(x), f.ptr(x), f.ptr(x), 1, 8, 4, 4, 64) != 0      # Cin * 9 > 64
    assert L.ia2p_conv_out(f.current_stream(), f.ptr(x), f.ptr(x), f.ptr(x), f.ptr(x), 1, 48, 4, 4, 4) != 0              # C % 32
    assert L.ia2p_conv_out(f.current_stream(), f.ptr(x), f.ptr(x), f.ptr(x), f.ptr(x), 1, 64, 4, 4, 9) != 0              # Co > 8


@pytest.mark.parametrize("tile", [-1, 0, 4, 6, 8, 12, 16, 20, 24, 25, 26])      # 24 .. 26: halo-staged patches (the ragged 9 x 7 map runs their gathered twins)
@pytest.mark.parametrize("B,H,W,Cin,Cin2,Co", [(2, 16, 16, 1280, 640, 1280), (1, 32, 32, 320, 960, 320), (2, 9, 7, 128, 64, 192), (1, 32, 48, 64, 128, 96)])
def test_conv3x3_with_appended_shortcut(L, B, H, W, Cin, Cin2, Co, tile):
    """conv2(h) + conv_shortcut(x) of a ResnetBlock2D as ONE implicit GEMM (K = 9 Cin + Cin2), every tile family incl. K-split plans"""
    f = _ffi()
    h, x = rnd(B, H, W, Cin, seed=51), rnd(B, H, W, Cin2, seed=52)
    w2, wsc = rnd(Co, Cin, 3, 3, seed=53, scale=(9 * Cin) ** -0.5), rnd(Co, Cin2, seed=54, scale=Cin2 ** -0.5)
    b = rnd(Co, seed=55, scale=0.2)
    wp = torch.empty(Co, 9 * Cin, dtype=torch.half, device="cuda")
    run(L, "ia2p_pack_conv3x3", f.ptr(w2), f.ptr(wp), Co, Cin)
    wcat = torch.cat([wp, wsc], dim=1).contiguous()
    y = torch.full((B * H * W, Co), float("nan"), dtype=torch.half, device="cuda")
    L.ia2p_debug_set_gemm_tile(tile)
    try:
        run(L, "ia2p_conv3x3_cat", f.ptr(h), f.ptr(x), f.ptr(wcat), f.ptr(b), f.ptr(y), B, H, W, Cin, Cin2, Co)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    ref = F.conv2d(h.float().permute(0, 3, 1, 2), w2.float(), None, padding=1) + F.conv2d(x.float().permute(0, 3, 1, 2), wsc.float()[:, :, None, None]) + b.float()[None, :, None, None]
    ref = ref.permute(0, 2, 3, 1).reshape(B * H * W, Co)
    assert rel_l2(y, ref) < 1.5e-3, rel_l2(y, ref)


@pytest.mark.parametrize("B,H,W,Cin,Co", [(8, 16, 16, 1280, 1280), (2, 32, 32, 640, 640), (1, 64, 64, 320, 320), (1, 16, 48, 64, 100), (3, 16, 16, 192, 256)])
def test_conv3x3_halo_staged_patches(L, B, H, W, Cin, Co):
    """conv_halo_f16_kernel (tile variants 24 / 25 / 26 = 160 / 128 / 80 wide): the activation operand is staged once per block of 64 channels as a 16 x 16 pixel patch with its border and
    the nine taps are read out of that image; K is walked block-major. Against the fp32 convolution; all three tile widths give the same bits (same accumulation
    order); a K split gives the same bits finished in the launch or by the reduce launch, and on a second run; splits are cut on whole blocks (more slices than
    blocks run the gathered twin tile); time-embedding row, bias and residual go through the patch-to-row map of the epilogue."""
    f = _ffi()
    x = rnd(B, H, W, Cin, seed=61)
    w = rnd(Co, Cin, 3, 3, seed=62, scale=(9 * Cin) ** -0.5)
    b, tv, res = rnd(Co, seed=63), rnd(B, Co, seed=64), rnd(B, H, W, Co, seed=65)
    wp = torch.empty(Co, 9 * Cin, dtype=torch.half, device="cuda")
    run(L, "ia2p_pack_conv3x3", f.ptr(w), f.ptr(wp), Co, Cin)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float(), b.float(), padding=1) + tv.float()[:, :, None, None]
    ref = (ref + res.permute(0, 3, 1, 2).float()).permute(0, 2, 3, 1)
    M = B * H * W
    part = torch.empty(4 * M * Co, dtype=torch.float32, device="cuda")
    outs = {}
    try:
        for route, limit in (("in-launch", 1 << 40), ("reduce launch", 0)):
            L.ia2p_debug_set_splitk_inkernel(limit)
            for tile in (24, 25, 26):
                L.ia2p_debug_set_gemm_tile(tile)
                for sk in (1, 2, 3, 4):
                    if sk > 9 * Cin // 64:
                        continue
                    y = torch.full((B, H, W, Co), float("nan"), dtype=torch.half, device="cuda")
                    part.fill_(float("nan"))
                    run(L, "ia2p_conv3x3_splitk", f.ptr(x), f.ptr(wp), f.ptr(b), f.ptr(tv), f.ptr(res), f.ptr(y), B, H, W, Cin, Co, sk, C.c_void_p(part.data_ptr()))
                    assert rel_l2(y, ref) < 1e-3, (route, tile, sk, rel_l2(y, ref))
                    outs[(route, tile, sk)] = y
    finally:
        L.ia2p_debug_set_splitk_inkernel(-1)
        L.ia2p_debug_set_gemm_tile(-1)
    for (route, tile, sk), y in outs.items():
        assert torch.equal(y, outs[("in-launch", 24, sk)]), (route, tile, sk)


def _sdpa(q, k, v):
    s = (q.float() @ k.float().transpose(-1, -2)) / 8.0
    return s.softmax(-1) @ v.float()


@pytest.mark.parametrize("B,heads,N", [(2, 2, 256), (1, 10, 1024), (2, 4, 100), (1, 1, 576), (1, 2, 33), (1, 2, 2304), (1, 1, 4096), (2, 1, 192), (1, 3, 320)])
def test_self_attention(L, B, heads, N):
    f = _ffi()
    C_ = heads * 64
    qkv = rnd(B, N, 3 * C_, seed=16)
    out = torch.empty(B, N, C_, dtype=torch.half, device="cuda")
    base = qkv.data_ptr()
    run(L, "ia2p_attention", f.ptr(qkv), 3 * C_, f.ptr(out), C_, B, heads, N, 1,
        C.c_void_p(base + 2 * C_), C.c_void_p(base + 4 * C_), 3 * C_, N, 1.0, None, None, 0, 0, 0.0)
    q, k, v = [t.reshape(B, N, heads, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = _sdpa(q, k, v).transpose(1, 2).reshape(B, N, C_)
    assert rel_l2(out, ref) < 2e-3, rel_l2(out, ref)


def test_self_attention_online_softmax_rescale(L):
    """Force the running-max rescale path: one late key dominates one query (cdna guide §5.4 rule 26)."""
    f = _ffi()
    B, heads, N = 1, 1, 256
    qkv = rnd(B, N, 192, seed=17)
    qkv[0, 5, 0:64] = 3.0
    qkv[0, 200, 64:128] = 3.0            # key 200 (4th tile) aligned with query 5
    out = torch.empty(B, N, 64, dtype=torch.half, device="cuda")
    base = qkv.data_ptr()
    run(L, "ia2p_attention", f.ptr(qkv), 192, f.ptr(out), 64, B, heads, N, 1,
        C.c_void_p(base + 128), C.c_void_p(base + 256), 192, N, 1.0, None, None, 0, 0, 0.0)
    q, k, v = [t.reshape(B, N, 1, 64).transpose(1, 2) for t in qkv.chunk(3, dim=-1)]
    ref = _sdpa(q, k, v).transpose(1, 2).reshape(B, N, 64)
    assert (out.float() - ref).abs().max() < 6e-3


@pytest.mark.parametrize("Lt,Li,scale", [(77, 4, 1.0), (73, 4, 0.5), (77, 0, 0.0), (128, 16, 0.7), (77, 64, 1.3), (300, 4, 0.9), (300, 100, 0.5), (64, 65, 2.0)])
def test_cross_attention_two_softmaxes(L, Lt, Li, scale):
    """text + image-token branches, each with its own softmax (reference attention_processor.py:371,387,397)."""
    f = _ffi()
    B, heads, N = 2, 4, 256
    C_ = heads * 64
    q, kv = rnd(B, N, C_, seed=18), rnd(B, Lt, 2 * C_, seed=19)
    kvi = rnd(B, max(Li, 1), 2 * C_, seed=20)
    out = torch.empty(B, N, C_, dtype=torch.half, device="cuda")
    run(L, "ia2p_attention", f.ptr(q), C_, f.ptr(out), C_, B, heads, N, 2 if Li else 1,
        f.ptr(kv), C.c_void_p(kv.data_ptr() + 2 * C_), 2 * C_, Lt, 1.0,
        f.ptr(kvi), C.c_void_p(kvi.data_ptr() + 2 * C_), 2 * C_, Li, scale)
    sp = lambda t: t.reshape(B, -1, heads, 64).transpose(1, 2)
    ref = _sdpa(sp(q), sp(kv[..., :C_]), sp(kv[..., C_:]))
    if Li:
        ref = ref + scale * _sdpa(sp(q), sp(kvi[..., :C_]), sp(kvi[..., C_:]))
    ref = ref.transpose(1, 2).reshape(B, N, C_)
    assert rel_l2(out, ref) < 2e-3, rel_l2(out, ref)


@pytest.mark.parametrize("Lt,Li,scale,folds", [(77, 4, 1.0, True), (73, 4, 0.5, True), (77, 4, 0.0, True), (13, 4, 0.9, True), (141, 16, 0.7, True), (100, 28, 1.2, True), (60, 4, 1.0, True),
                                               (64, 4, 1.0, False), (77, 33, 1.0, False), (120, 16, 0.8, False), (300, 4, 0.9, False)])
def test_image_token_keys_fold_into_the_last_text_tile(L, Lt, Li, scale, folds):
    """Round 6: IPAttnProcessor2_0's two scaled_dot_product_attention calls (reference attention_processor.py:371 text, :387 image tokens, :397
    `text + scale * ip`) walk 81 keys as TWO key tiles -- the 4 image-token keys ride in the free slots of the second text tile, a key-index mask keeps the two softmaxes
    apart -- instead of three. Against fp32 torch, and against the three-tile form (ia2p_debug_set_attn_fold(0)) to one fp16 ulp: the softmaxes are the same numbers, the
    probability sums and the P.V products are added up in another order. Stand-alone kernel and the fused to_q + cross-attention launch; per-request scales too.
    Shapes that do not fold (a full last text tile, more than 32 image tokens, no room, a non-resident context) must not change at all."""
    f = _ffi()
    B, heads, N = 2, 4, 256
    C_ = heads * 64
    q, kv, kvi = rnd(B, N, C_, seed=118), rnd(B, Lt, 2 * C_, seed=119), rnd(B, Li, 2 * C_, seed=120)
    sp = lambda t: t.reshape(B, -1, heads, 64).transpose(1, 2)
    ref = (_sdpa(sp(q), sp(kv[..., :C_]), sp(kv[..., C_:])) + scale * _sdpa(sp(q), sp(kvi[..., :C_]), sp(kvi[..., C_:]))).transpose(1, 2).reshape(B, N, C_)
    segs = (2, f.ptr(kv), C.c_void_p(kv.data_ptr() + 2 * C_), 2 * C_, Lt, 1.0, f.ptr(kvi), C.c_void_p(kvi.data_ptr() + 2 * C_), 2 * C_, Li, scale)
    outs = {}
    X, W, b = rnd(B * N, C_, seed=133), rnd(C_, C_, seed=134, scale=C_ ** -0.5), rnd(C_, seed=135, scale=0.3)
    fused_ok = (Lt + 63) // 64 + (Li + 63) // 64 <= 3            # (the fused launch keeps a resident context of at most three key tiles)
    try:
        for mode in (1, 0):
            L.ia2p_debug_set_attn_fold(mode)
            out = torch.full((B, N, C_), float("nan"), dtype=torch.half, device="cuda")
            run(L, "ia2p_attention", f.ptr(q), C_, f.ptr(out), C_, B, heads, N, *segs)
            outs[("attention", mode)] = out
            if fused_ok:
                o2 = torch.full((B, N, C_), float("nan"), dtype=torch.half, device="cuda")
                run(L, "ia2p_qproj_attention", f.ptr(X), f.ptr(W), f.ptr(b), None, f.ptr(o2), C_, B, heads, N, C_, *segs)
                outs[("fused", mode)] = o2
    finally:
        L.ia2p_debug_set_attn_fold(-1)
    assert rel_l2(outs[("attention", 1)], ref) < 2e-3 and rel_l2(outs[("attention", 0)], ref) < 2e-3
    kinds = ["attention"] + (["fused"] if fused_ok else [])
    for kind in kinds:
        a, c = outs[(kind, 1)], outs[(kind, 0)]
        assert torch.isfinite(a).all()
        if not folds:
            assert torch.equal(a, c), kind                                          # nothing to fold: the switch changes nothing
            continue
        # one fp16 ulp at the scale of the query's output: |a - c| <= the spacing of fp16 numbers at the largest |O| of the query's head (an output is a sum of
        # probability x value terms whose probabilities are rounded to fp16 before the P.V product: where the image tokens' probabilities round the other way -- their
        # common factor l_text / l_ip is summed in another order -- an output moves by 2^-11 of a TERM, which is many ulps of an output that cancels to near zero)
        d = (a.float() - c.float()).abs().reshape(B, N, heads, 64)
        scale_ = torch.maximum(a.float().abs(), c.float().abs()).reshape(B, N, heads, 64).amax(dim=-1, keepdim=True).clamp_min(2.0 ** -14)
        ulp = torch.exp2(torch.floor(torch.log2(scale_)) - 10)
        assert bool((d <= ulp).all()), (kind, float((d / ulp).max()))
        assert rel_l2(a, c) < 2e-4, (kind, rel_l2(a, c))
        assert float((a != c).float().mean()) < 0.25, kind                          # ... and most outputs are the same bits
    if fused_ok:
        qref = (X.float() @ W.float().t() + b.float()).half()
        two = torch.empty(B, N, C_, dtype=torch.half, device="cuda")
        L.ia2p_debug_set_gemm_tile(2)
        try:
            qq = torch.empty(B * N, C_, dtype=torch.half, device="cuda")
            run(L, "ia2p_gemm_ex", f.ptr(X), f.ptr(W), f.ptr(b), None, f.ptr(qq), B * N, C_, C_, 0, None, None, None, 1, None)
            run(L, "ia2p_attention", f.ptr(qq), C_, f.ptr(two), C_, B, heads, N, *segs)
        finally:
            L.ia2p_debug_set_gemm_tile(-1)
        assert torch.equal(two, outs[("fused", 1)])                                  # fused and two-launch forms share the core: same bits, folded or not


@pytest.mark.parametrize("folded", [True, False])
@pytest.mark.parametrize("B,heads,Nq,Lt,Li,scale", [(2, 4, 256, 77, 4, 1.0), (1, 20, 256, 77, 4, 0.6), (1, 10, 1024, 77, 0, 0.0), (2, 2, 128, 128, 4, 0.9),
                                                    (1, 3, 384, 64, 65, 2.0), (1, 2, 256, 77, 64, 1.3), (1, 2, 256, 13, 0, 0.0), (1, 1, 128, 192, 0, 0.0)])
def test_qproj_fused_with_cross_attention(L, B, heads, Nq, Lt, Li, scale, folded):
    """ia2p_qproj_attention = to_q (optionally behind a folded LayerNorm) + the two-softmax cross-attention in ONE launch
    (reference attention_processor.py:344, :371, :387, :397): same BITS as ia2p_gemm_ex on the 128 x 64 tile followed by
    ia2p_attention, and within the attention tolerance of torch fp32."""
    f = _ffi()
    C_ = heads * 64
    M = B * Nq
    kv, kvi = rnd(B, Lt, 2 * C_, seed=31), rnd(B, max(Li, 1), 2 * C_, seed=32)
    att = lambda q_ptr, out: run(L, "ia2p_attention", q_ptr, C_, f.ptr(out), C_, B, heads, Nq, 2 if Li else 1,
                                 f.ptr(kv), C.c_void_p(kv.data_ptr() + 2 * C_), 2 * C_, Lt, 1.0,
                                 f.ptr(kvi), C.c_void_p(kvi.data_ptr() + 2 * C_), 2 * C_, Li, scale)
    segs = (2 if Li else 1, f.ptr(kv), C.c_void_p(kv.data_ptr() + 2 * C_), 2 * C_, Lt, 1.0,
            f.ptr(kvi), C.c_void_p(kvi.data_ptr() + 2 * C_), 2 * C_, Li, scale)
    q = torch.empty(M, C_, dtype=torch.half, device="cuda")
    two, one = torch.empty(B, Nq, C_, dtype=torch.half, device="cuda"), torch.full((B, Nq, C_), float("nan"), dtype=torch.half, device="cuda")
    L.ia2p_debug_set_gemm_tile(2)                      # 128 x 64 x 2 stages: the tile the fused kernel is built on
    try:
        if folded:
            _, X, Wp, R, gamma, beta, W, b, Wf, cs, fb = _ln_fold_setup(L, M, C_, C_, seed=70 + heads, bias=False)
            t = torch.empty(M, C_, dtype=torch.half, device="cuda")
            stats = torch.zeros((C_ // 64 + 1) * M * 2, dtype=torch.float32, device="cuda")
            slots = C.c_int(0)
            run(L, "ia2p_gemm_ex", f.ptr(X), f.ptr(Wp), None, f.ptr(R), f.ptr(t), M, C_, C_, 0, None, f.ptr(stats), C.addressof(slots), 1, None)
            ln = f.LnFoldC(stats.data_ptr(), slots.value, cs.data_ptr(), fb.data_ptr(), 1e-5)
            run(L, "ia2p_gemm_ex", f.ptr(t), f.ptr(Wf), None, None, f.ptr(q), M, C_, C_, 0, C.addressof(ln), None, None, 1, None)
            att(f.ptr(q), two)
            run(L, "ia2p_qproj_attention", f.ptr(t), f.ptr(Wf), None, C.addressof(ln), f.ptr(one), C_, B, heads, Nq, C_, *segs)
            qref = F.layer_norm(t.float(), (C_,), gamma.float(), beta.float(), 1e-5) @ W.float().t()
        else:
            X, W, b = rnd(M, C_, seed=33), rnd(C_, C_, seed=34, scale=C_ ** -0.5), rnd(C_, seed=35, scale=0.3)
            run(L, "ia2p_gemm_ex", f.ptr(X), f.ptr(W), f.ptr(b), None, f.ptr(q), M, C_, C_, 0, None, None, None, 1, None)
            att(f.ptr(q), two)
            run(L, "ia2p_qproj_attention", f.ptr(X), f.ptr(W), f.ptr(b), None, f.ptr(one), C_, B, heads, Nq, C_, *segs)
            qref = X.float() @ W.float().t() + b.float()
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    assert torch.equal(one, two), float((one.float() - two.float()).abs().max())
    sp = lambda x: x.reshape(B, -1, heads, 64).transpose(1, 2)
    ref = _sdpa(sp(qref), sp(kv[..., :C_]), sp(kv[..., C_:]))
    if Li:
        ref = ref + scale * _sdpa(sp(qref), sp(kvi[..., :C_]), sp(kvi[..., C_:]))
    ref = ref.transpose(1, 2).reshape(B, Nq, C_)
    assert rel_l2(one, ref) < 3e-3, rel_l2(one, ref)     # Q is rounded to fp16 between the projection and the scores, as in the reference


@pytest.mark.parametrize("folded", [True, False])
@pytest.mark.parametrize("B,heads,K", [(8, 20, 1280), (2, 4, 256), (1, 1, 64), (3, 10, 640)])
def test_qkv_fused_with_self_attention(L, B, heads, K, folded):
    """ia2p_qkv_self_attention = the stacked Q | K | V projection (optionally behind a folded LayerNorm) + the self-attention of AttnProcessor2_0 over the 256
    tokens of an image in ONE launch (reference attention_processor.py:239, :246-247, :259): the same BITS as ia2p_gemm_ex (N = 3 C) followed by ia2p_attention
    -- whatever tile the stand-alone GEMM runs on -- and within the attention tolerance of torch fp32."""
    f = _ffi()
    C_ = heads * 64
    Nq, M = 256, B * 256
    qkv = torch.empty(M, 3 * C_, dtype=torch.half, device="cuda")
    two, one = torch.empty(B, Nq, C_, dtype=torch.half, device="cuda"), torch.full((B, Nq, C_), float("nan"), dtype=torch.half, device="cuda")
    att = lambda out: run(L, "ia2p_attention", f.ptr(qkv), 3 * C_, f.ptr(out), C_, B, heads, Nq, 1,
                          C.c_void_p(qkv.data_ptr() + 2 * C_), C.c_void_p(qkv.data_ptr() + 4 * C_), 3 * C_, Nq, 1.0, None, None, 0, 0, 0.0)
    if folded:
        g = torch.Generator().manual_seed(100 + heads)
        t = (torch.randn(M, K, generator=g) * 1.5 + 0.3).half().cuda()
        gamma, beta = (1.0 + 0.3 * torch.randn(K, generator=g)).half().cuda(), (0.2 * torch.randn(K, generator=g)).half().cuda()
        W, b = (torch.randn(3 * C_, K, generator=g) * K ** -0.5).half().cuda(), None
        Wf = torch.empty_like(W)
        cs, fb = torch.empty(3 * C_, dtype=torch.float32, device="cuda"), torch.empty(3 * C_, dtype=torch.float32, device="cuda")
        run(L, "ia2p_fold_layernorm", f.ptr(W), f.ptr(gamma), f.ptr(beta), None, f.ptr(Wf), f.ptr(cs), f.ptr(fb), 3 * C_, K)
        tf = t.float()
        slots = K // 64
        st = torch.stack([tf.view(M, slots, 64).sum(2), (tf * tf).view(M, slots, 64).sum(2)], dim=2).permute(1, 0, 2).contiguous()
        ln = f.LnFoldC(st.data_ptr(), slots, cs.data_ptr(), fb.data_ptr(), 1e-5)
        for tile in (0, 12, 4):
            L.ia2p_debug_set_gemm_tile(tile)
            try:
                run(L, "ia2p_gemm_ex", f.ptr(t), f.ptr(Wf), None, None, f.ptr(qkv), M, 3 * C_, K, 0, C.addressof(ln), None, None, 1, None)
            finally:
                L.ia2p_debug_set_gemm_tile(-1)
            att(two)
            one.fill_(float("nan"))
            run(L, "ia2p_qkv_self_attention", f.ptr(t), f.ptr(Wf), None, C.addressof(ln), f.ptr(one), C_, B, heads, K)
            assert torch.equal(one, two), (tile, float((one.float() - two.float()).abs().max()))
        ref_qkv = F.layer_norm(tf, (K,), gamma.float(), beta.float(), 1e-5) @ W.float().t()
    else:
        X, W, b = rnd(M, K, seed=33), rnd(3 * C_, K, seed=34, scale=K ** -0.5), rnd(3 * C_, seed=35, scale=0.3)
        run(L, "ia2p_gemm_ex", f.ptr(X), f.ptr(W), f.ptr(b), None, f.ptr(qkv), M, 3 * C_, K, 0, None, None, None, 1, None)
        att(two)
        run(L, "ia2p_qkv_self_attention", f.ptr(X), f.ptr(W), f.ptr(b), None, f.ptr(one), C_, B, heads, K)
        assert torch.equal(one, two), float((one.float() - two.float()).abs().max())
        ref_qkv = X.float() @ W.float().t() + b.float()
    sp = lambda x: x.reshape(B, Nq, heads, 64).transpose(1, 2)
    q, k, v = ref_qkv[:, :C_], ref_qkv[:, C_:2 * C_], ref_qkv[:, 2 * C_:]
    ref = _sdpa(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(B, Nq, C_)
    assert rel_l2(one, ref) < 3e-3, rel_l2(one, ref)
    # wrong shapes are refused, not mis-computed
    assert L.ia2p_qkv_self_attention(f.current_stream(), f.ptr(one), f.ptr(W), None, None, f.ptr(one), C_, B, heads, 100) != 0      # K % 64
    assert L.ia2p_qkv_self_attention(f.current_stream(), None, f.ptr(W), None, None, f.ptr(one), C_, B, heads, K) != 0


def test_qproj_attention_rejects_bad_shapes(L):
    f = _ffi()
    x, w, kv = rnd(200, 128, seed=1), rnd(128, 128, seed=2), rnd(77, 256, seed=3)
    o = torch.empty(200, 128, dtype=torch.half, device="cuda")
    rc = L.ia2p_qproj_attention(f.current_stream(), f.ptr(x), f.ptr(w), None, None, f.ptr(o), 128, 1, 2, 200, 128, 1,
                                f.ptr(kv), C.c_void_p(kv.data_ptr() + 256), 256, 77, 1.0, None, None, 0, 0, 0.0)
    assert rc != 0                                       # Nq = 200 is not a multiple of 128
    kl = rnd(300, 256, seed=4)
    rc = L.ia2p_qproj_attention(f.current_stream(), f.ptr(x), f.ptr(w), None, None, f.ptr(o), 128, 1, 2, 128, 128, 1,
                                f.ptr(kl), C.c_void_p(kl.data_ptr() + 256), 256, 300, 1.0, None, None, 0, 0, 0.0)
    assert rc != 0                                       # 300 keys = 5 tiles: only contexts of <= 3 key tiles are fused (callers use the two launches)
    rc = L.ia2p_qproj_attention(f.current_stream(), None, f.ptr(w), None, None, f.ptr(o), 128, 1, 2, 128, 128, 1,
                                f.ptr(kv), C.c_void_p(kv.data_ptr() + 256), 256, 77, 1.0, None, None, 0, 0, 0.0)
    assert rc != 0


@pytest.mark.parametrize("B,HW,C_,silu,eps", [(8, 4096, 320, 1, 1e-5), (2, 256, 1280, 0, 1e-6), (1, 1024, 1920, 1, 1e-5),
                                             (2, 256, 2560, 1, 1e-5), (2, 64, 64, 1, 1e-5), (1, 576, 960, 1, 1e-5),
                                             # the UNet's (channels, pixels) shapes at 512^2 / 768^2 / 1024^2 and the refiner's widths
                                             (2, 4096, 640, 1, 1e-5), (2, 4096, 960, 1, 1e-5), (3, 1024, 640, 1, 1e-5), (2, 1024, 1280, 0, 1e-6),
                                             (1, 9216, 320, 1, 1e-5), (1, 9216, 640, 1, 1e-5), (1, 2304, 1280, 1, 1e-5), (1, 16384, 320, 1, 1e-5),
                                             (2, 256, 1920, 1, 1e-5), (1, 1000, 384, 1, 1e-5), (2, 1024, 768, 0, 1e-6), (1, 4096, 1152, 1, 1e-5),
                                             (1, 1024, 1536, 1, 1e-5), (2, 256, 2304, 1, 1e-5), (1, 256, 3072, 1, 1e-5), (1, 70, 320, 1, 1e-5)])
def test_groupnorm_silu(L, B, HW, C_, silu, eps):
    f = _ffi()
    x = (rnd(B, HW, C_, seed=21) * 2 + 0.7).half()
    x[:, :, : C_ // 32] += 40.0                # one group far from zero: E[x^2] - mean^2 would lose the variance there
    ga, be = (1 + 0.1 * rnd(C_, seed=22)).half(), (0.05 * rnd(C_, seed=23)).half()
    y = torch.empty_like(x)
    part = torch.empty(B * 64 * 32 * 2, dtype=torch.float32, device="cuda")
    run(L, "ia2p_groupnorm_silu", f.ptr(x), f.ptr(y), f.ptr(ga), f.ptr(be), B, HW, C_, 32, eps, silu, C.c_void_p(part.data_ptr()))
    ref = F.group_norm(x.float().transpose(1, 2), 32, ga.float(), be.float(), eps)
    if silu:
        ref = F.silu(ref)
    ref = ref.transpose(1, 2)
    assert rel_l2(y, ref) < 1e-3, rel_l2(y, ref)


@pytest.mark.parametrize("M,C_", [(2048, 1280), (8192, 640), (77, 128), (5, 2048)])
def test_layernorm(L, M, C_):
    f = _ffi()
    x = (rnd(M, C_, seed=24) * 3 - 0.5).half()
    ga, be = (1 + 0.1 * rnd(C_, seed=25)).half(), (0.05 * rnd(C_, seed=26)).half()
    y = torch.empty_like(x)
    run(L, "ia2p_layernorm", f.ptr(x), f.ptr(y), f.ptr(ga), f.ptr(be), M, C_, 1e-5)
    ref = F.layer_norm(x.float(), (C_,), ga.float(), be.float(), 1e-5)
    assert rel_l2(y, ref) < 1e-3, rel_l2(y, ref)


@pytest.mark.parametrize("M,N,K", [(8, 1280, 320), (16, 1280, 2816), (1, 13760, 1280), (3, 50, 64)])
def test_linear_small(L, M, N, K):
    f = _ffi()
    X, W, b = rnd(M, K, seed=27), rnd(N, K, seed=28, scale=K ** -0.5), rnd(N, seed=29)
    out = torch.empty(M, N, dtype=torch.half, device="cuda")
    run(L, "ia2p_linear_small", f.ptr(X), f.ptr(W), f.ptr(b), f.ptr(out), M, N, K, 1, 1)
    ref = F.silu(F.silu(X.float()) @ W.float().t() + b.float())
    assert rel_l2(out, ref) < 1e-3, rel_l2(out, ref)


def test_ddim_step_cfg(L):
    f = _ffi()
    n = 8 * 4 * 64 * 64
    x, eu, ec = rnd(n, seed=30), rnd(n, seed=31), rnd(n, seed=32)
    out, out2 = torch.empty_like(x), torch.empty_like(x)
    g, cx, ce = 10.0, 1.0123, -0.0456
    f.check(L.ia2p_ddim_step(f.current_stream(), f.ptr(x), f.ptr(eu), f.ptr(ec), g, cx, ce, f.ptr(out), f.ptr(out2), n))
    torch.cuda.synchronize()
    ref = (cx * x.float() + ce * (eu.float() + g * (ec.float() - eu.float()))).half()   # one rounding, like the kernel
    assert (out.float() - ref.float()).abs().max() <= 2e-3 * ref.float().abs().max()
    assert torch.equal(out, out2)


@pytest.mark.parametrize("M,C,S", [(2048, 1280, 3), (2048, 1280, 1), (256, 1280, 4), (8192, 640, 1), (1024, 320, 2)])
def test_ffn_operator(L, M, C, S):
    """The GEGLU feed-forward of a BasicTransformerBlock as one operator call (ff.net.0 GEGLU -> H, ff.net.2 + bias + residual, two launches under the library's
    plans): against torch, the same bits whichever tiles the plans name, and call after call."""
    f = _ffi()
    X, R = rnd(M, C, seed=81), rnd(M, C, seed=82)
    W1, b1 = rnd(8 * C, C, seed=83, scale=C ** -0.5), rnd(8 * C, seed=84, scale=0.1)
    W2, b2 = rnd(C, 4 * C, seed=85, scale=(4 * C) ** -0.5), rnd(C, seed=86, scale=0.1)
    W1p, b1p = torch.empty_like(W1), torch.empty_like(b1)
    run(L, "ia2p_pack_geglu", f.ptr(W1), f.ptr(W1p), 8 * C, C)
    run(L, "ia2p_pack_geglu", f.ptr(b1), f.ptr(b1p), 8 * C, 1)
    part = torch.empty(max(S, 1) * M * C, dtype=torch.float32, device="cuda")
    H = torch.empty(M, 4 * C, dtype=torch.half, device="cuda")
    outs = {}
    for tile_a in (8, 0, 18):                     # FF-in on 128x160, 128x128 and the 256x160 ping-pong tile; FF-out on 128x128
        o = torch.empty(M, C, dtype=torch.half, device="cuda")
        for rep in range(2):
            o.fill_(float("nan")); H.fill_(float("nan"))
            _force_plans(L, M, C, tile_a, S)
            f.check(L.ia2p_ffn(f.current_stream(), f.ptr(X), f.ptr(W1p), f.ptr(b1p), f.ptr(W2), f.ptr(b2), f.ptr(R), f.ptr(H), f.ptr(o), M, C, S, ctypes_ptr(part)))
            torch.cuda.synchronize()
            if tile_a in outs:
                assert torch.equal(o, outs[tile_a])
            outs[tile_a] = o.clone()
    L.ia2p_plan_clear()
    h = X.float() @ W1.float().t() + b1.float()
    a, g = h.chunk(2, dim=-1)
    ref = (a * F.gelu(g)).half().float() @ W2.float().t() + b2.float() + R.float()
    assert rel_l2(outs[8], ref) < 2e-3
    assert torch.equal(outs[8], outs[0]) and torch.equal(outs[8], outs[18])


def C_int():
    import ctypes
    return ctypes.c_int(0)


def ctypes_ptr(t):
    import ctypes
    return ctypes.c_void_p(t.data_ptr())


def ctypes_byref(x):
    import ctypes
    return ctypes.byref(x)


def _force_plans(L, M, C, tile_a, S):
    """pin the two plans of the feed-forward pair through the plan table (what ia2p_autotune fills): FF-in on `tile_a`, FF-out on 128x128 with K split S"""
    txt = f"{M},{8 * C},{C},0,1,{tile_a},1;{M},{C},{4 * C},0,0,0,{max(S, 1)};"
    assert L.ia2p_plan_import(txt.encode()) == 2
