"""GroupNorm + SiLU fused into the halo-staged 3x3 convolution (round 5), operator by operator through the C ABI.

Reference op: diffusers ResnetBlock2D `conv1(nonlinearity(norm1(x)))` / `conv2(nonlinearity(norm2(h)))` behind
/root/reference/instructany2pix/ddim/pnp_pipeline.py:253-260 (in-tree twin llm/model/vae/modules/blocks.py:122-142): here
`F.conv2d(F.silu(F.group_norm(x, 32, gamma, beta, eps)), w, b, padding=1)` in fp32 on the same fp16 inputs.

What is pinned:
  * producer side -- the column sums a GEMM / conv epilogue leaves for its own output are the numbers ia2p_gn_colstats computes for the stored tensor,
    whatever tile, K split or finishing route produced it; and ia2p_gn_colstats is, to the bit, the canonical definition (fp32 over aligned runs of
    16 rows in row order, fp64 from there on) evaluated on the host;
  * consumer side -- ia2p_gn_apply_stats against fp32 GroupNorm (+SiLU), incl. the two-source input whose groups straddle the boundary (the up path's
    1280 + 640 channels: groups of 60, group 21 = 20 channels of the first tensor + 40 of the second);
  * the fused convolution against the fp32 pipeline at every halo tile width, with K splits, appended 1x1 block, time-embedding row, residual, image
    borders (border pixels must stay exactly zero: the reference pads the ACTIVATED tensor) -- and BIT-IDENTICAL to its unfused pair
    (ia2p_gn_apply_stats, then the plain halo-staged convolution)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


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


def rnd(*shape, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale + shift).half().cuda()


def run(L, name, *args):
    f = _ffi()
    f.check(getattr(L, name)(f.current_stream(), *args))
    torch.cuda.synchronize()


def colstats(L, x2d, rows):
    """ia2p_gn_colstats of a [M, C] fp16 tensor -> float64 [M / rows, C, 2]"""
    f = _ffi()
    M, Cc = x2d.shape
    out = torch.full((M // rows, Cc, 2), float("nan"), dtype=torch.float64, device="cuda")
    run(L, "ia2p_gn_colstats", f.ptr(x2d), M, Cc, rows, C.c_void_p(out.data_ptr()))
    return out


def canonical_colstats(x2d, rows):
    """the definition, on the host: per slot and channel, fp32 {sum, fma-accumulated sum of squares} over aligned runs of 16 rows taken in row order, the runs added in
    fp64 (run order is immaterial there: 16 ... 64 fp32 values of like magnitude add exactly in fp64)"""
    x = x2d.float().cpu().numpy()
    M, Cc = x.shape
    s = np.zeros((M // 16, Cc), np.float32)
    q = np.zeros((M // 16, Cc), np.float32)
    xs = x.reshape(M // 16, 16, Cc)
    for i in range(16):
        v = xs[:, i, :]
        s = (s + v).astype(np.float32)
        q = (v.astype(np.float64) * v.astype(np.float64) + q.astype(np.float64)).astype(np.float32)      # fma: the product is exact in fp64, one rounding
    s64 = s.astype(np.float64).reshape(M // rows, rows // 16, Cc).sum(1)
    q64 = q.astype(np.float64).reshape(M // rows, rows // 16, Cc).sum(1)
    return np.stack([s64, q64], axis=-1)


@pytest.mark.parametrize("M,Cc,rows", [(512, 64, 256), (2048, 320, 256), (1024, 128, 64), (576, 192, 576), (4096, 640, 1024), (256, 1280, 16)])
def test_colstats_is_the_canonical_definition(L, M, Cc, rows):
    x = rnd(M, Cc, seed=3, scale=2.0, shift=0.5)
    got = colstats(L, x, rows).cpu().numpy()
    ref = canonical_colstats(x, rows)
    assert np.array_equal(got, ref), float(np.abs(got - ref).max())
    exact = torch.stack([x.double().reshape(M // rows, rows, Cc).sum(1), (x.double() ** 2).reshape(M // rows, rows, Cc).sum(1)], dim=-1).cpu().numpy()
    assert np.allclose(got, exact, rtol=2e-6, atol=1e-3)


def pack3(L, w):
    f = _ffi()
    Co, Cin = w.shape[:2]
    wp = torch.empty(Co, 9 * Cin, dtype=torch.half, device="cuda")
    run(L, "ia2p_pack_conv3x3", f.ptr(w), f.ptr(wp), Co, Cin)
    return wp


def conv_gn(L, *, x0, st0=None, rows0=0, x1=None, st1=None, rows1=0, gamma=None, beta=None, groups=32, eps=1e-5, wp, bias=None, rowvec=None, residual=None, B, H, W, Co, xa=None,
            splitk=0, want_stats=False, expect_fail=False):
    """ia2p_conv3x3_gn -> (y [B*H*W, Co], stats [slots, Co, 2] or None, rows)"""
    f = _ffi()
    M = B * H * W
    d = f.ConvGnC()
    p = lambda t: None if t is None else t.data_ptr()
    d.x0, d.C0, d.st0, d.rows0 = p(x0), x0.shape[-1], p(st0), rows0
    d.x1, d.C1, d.st1, d.rows1 = p(x1), (x1.shape[-1] if x1 is not None else 0), p(st1), rows1
    d.gamma, d.beta, d.groups, d.eps = p(gamma), p(beta), groups, eps
    d.Wp, d.bias, d.rowvec, d.residual = p(wp), p(bias), p(rowvec), p(residual)
    y = torch.full((M, Co), float("nan"), dtype=torch.half, device="cuda")
    d.y, d.B, d.H, d.W, d.Co = p(y), B, H, W, Co
    d.xa, d.Ca = p(xa), (xa.shape[-1] if xa is not None else 0)
    part = torch.empty(max(1, splitk) * M * Co if splitk > 1 else 1, dtype=torch.float32, device="cuda").fill_(float("nan"))
    d.splitk, d.partial = splitk, p(part)
    stats = torch.full((M // 16, Co, 2), float("nan"), dtype=torch.float64, device="cuda") if want_stats else None
    d.gn_out = p(stats)
    rows = C.c_int(0)
    rc = L.ia2p_conv3x3_gn(f.current_stream(), C.byref(d), C.byref(rows))
    torch.cuda.synchronize()
    if expect_fail:
        assert rc != 0
        return None, None, 0
    f.check(rc)
    if want_stats and rows.value:
        stats = stats.reshape(-1)[: (M // rows.value) * Co * 2].reshape(M // rows.value, Co, 2)
    return y, stats, rows.value


@pytest.mark.parametrize("B,H,W,Cin,Co", [(2, 32, 32, 320, 320), (8, 16, 16, 1280, 640), (1, 64, 64, 64, 160), (3, 16, 48, 128, 96)])
def test_epilogue_statistics_equal_colstats_of_the_stored_output(L, B, H, W, Cin, Co):
    """every tile family, K splits on both finishing routes: the column sums the launch leaves are those of the fp16 tensor it stored"""
    x, w = rnd(B, H, W, Cin, seed=11), rnd(Co, Cin, 3, 3, seed=12, scale=(9 * Cin) ** -0.5)
    b, tv, res = rnd(Co, seed=13), rnd(B, Co, seed=14), rnd(B, H, W, Co, seed=15)
    wp = pack3(L, w)
    produced = 0
    try:
        for limit in (1 << 40, 0):
            L.ia2p_debug_set_splitk_inkernel(limit)
            for tile in (24, 25, 26, 18, 12, 0, 4, 8, 16, 19, 22):
                L.ia2p_debug_set_gemm_tile(tile)
                for sk in (0, 2, 3):
                    if sk > 9 * Cin // 64 or (limit == 0 and sk == 0):
                        continue
                    y, st, rows = conv_gn(L, x0=x.reshape(-1, Cin), wp=wp, bias=b, rowvec=tv, residual=res.reshape(-1, Co), B=B, H=H, W=W, Co=Co, splitk=sk, want_stats=True)
                    if limit == 0 and sk > 1:
                        assert rows == 0          # finished by the reduce launch: no epilogue statistics (the executor runs ia2p_gn_colstats)
                        continue
                    if rows == 0:
                        continue                  # (a tile that does not divide the image, e.g. 256 rows on a 16 x 48 map's 768 pixels is fine, 160 is not)
                    produced += 1
                    # a slot of a halo-staged tile is a 16 x 16 PATCH, a slot of a linear tile 256 consecutive pixels: per image both cover the same aligned runs of
                    # 16 pixels, and the per-image sums (what the consumer folds; exact in fp64) must agree to the bit
                    ref = colstats(L, y, rows)
                    per_image = lambda t: t.reshape(B, H * W // rows, Co, 2).sum(1)
                    assert torch.equal(per_image(st), per_image(ref)), (tile, sk, rows, float((per_image(st) - per_image(ref)).abs().max()))
                    if tile not in (24, 25, 26):
                        assert torch.equal(st, ref), (tile, sk, rows)
    finally:
        L.ia2p_debug_set_splitk_inkernel(-1)
        L.ia2p_debug_set_gemm_tile(-1)
    assert produced >= 8


@pytest.mark.parametrize("M,N,K,HW", [(2048, 1280, 1280, 256), (8192, 640, 640, 1024), (512, 320, 64, 256)])
def test_gemm_epilogue_statistics(L, M, N, K, HW):
    """a Transformer2DModel's proj_out (+ residual) in front of the next ResnetBlock2D: linear tiles of 64 ... 256 rows, with and without a K split"""
    f = _ffi()
    A, W, b, R = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5), rnd(N, seed=23), rnd(M, N, seed=24)
    produced = 0
    try:
        for tile in (-1, 0, 2, 4, 8, 12, 16, 18, 19, 20, 22):
            L.ia2p_debug_set_gemm_tile(tile)
            for sk in (0, 2):
                if sk > K // 64:
                    continue
                out = torch.full((M, N), float("nan"), dtype=torch.half, device="cuda")
                st = torch.full((M // 16, N, 2), float("nan"), dtype=torch.float64, device="cuda")
                part = torch.empty(max(1, sk) * M * N, dtype=torch.float32, device="cuda")
                rows = C.c_int(0)
                run(L, "ia2p_gemm_gnstats", f.ptr(A), f.ptr(W), f.ptr(b), f.ptr(R), f.ptr(out), M, N, K, sk, C.c_void_p(part.data_ptr()), HW, C.c_void_p(st.data_ptr()), C.byref(rows))
                ref = A.float() @ W.float().t() + b.float() + R.float()
                assert rel_l2(out, ref) < 1e-3
                if rows.value:
                    produced += 1
                    got = st.reshape(-1)[: (M // rows.value) * N * 2].reshape(M // rows.value, N, 2)
                    assert torch.equal(got, colstats(L, out, rows.value)), (tile, sk, rows.value)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    assert produced >= 6


def gn_ref(xcat, gamma, beta, groups, eps, B, H, W):
    """fp32 silu(group_norm) of a channels-last [B*H*W, C] tensor -> NCHW fp32"""
    Cc = xcat.shape[-1]
    xn = xcat.float().reshape(B, H, W, Cc).permute(0, 3, 1, 2)
    return F.silu(F.group_norm(xn, groups, gamma.float(), beta.float(), eps))


@pytest.mark.parametrize("B,H,W,C0,C1,rows0,rows1", [(2, 16, 16, 1280, 640, 256, 64), (1, 32, 32, 320, 0, 256, 0), (2, 16, 16, 128, 64, 128, 256), (1, 64, 64, 320, 320, 256, 1024), (3, 8, 8, 64, 0, 64, 0)])
def test_apply_from_producer_statistics_vs_fp32_groupnorm(L, B, H, W, C0, C1, rows0, rows1):
    """the fold (slots in slot order, channels in channel order, fp64) + scale / shift + SiLU against fp32 GroupNorm; two sources with different slot sizes, groups
    straddling the boundary (1280 + 640: groups of 60)"""
    f = _ffi()
    HW, Cc = H * W, C0 + C1
    x0 = rnd(B * HW, C0, seed=31, scale=1.5, shift=0.7)
    x1 = rnd(B * HW, C1, seed=32, scale=0.6, shift=-0.4) if C1 else None
    gamma, beta = rnd(Cc, seed=33, scale=0.3, shift=1.0), rnd(Cc, seed=34, scale=0.2)
    st0 = colstats(L, x0, rows0)
    st1 = colstats(L, x1, rows1) if C1 else None
    for silu in (1, 0):
        y = torch.full((B * HW, Cc), float("nan"), dtype=torch.half, device="cuda")
        run(L, "ia2p_gn_apply_stats", f.ptr(x0), C0, C.c_void_p(st0.data_ptr()), rows0, f.ptr(x1), C1, C.c_void_p(st1.data_ptr()) if C1 else None, rows1,
            f.ptr(gamma), f.ptr(beta), f.ptr(y), B, HW, 32, 1e-5, silu)
        xcat = torch.cat([x0, x1], dim=1) if C1 else x0
        xn = xcat.float().reshape(B, H, W, Cc).permute(0, 3, 1, 2)
        ref = F.group_norm(xn, 32, gamma.float(), beta.float(), 1e-5)
        ref = (F.silu(ref) if silu else ref).permute(0, 2, 3, 1).reshape(B * HW, Cc)
        assert rel_l2(y, ref) < 1e-3, (silu, rel_l2(y, ref))


CASES = [  # B, H, W, C0, C1, Co, Ca (appended 1x1 block), producer slot rows (source 0, source 1)
    (2, 16, 16, 1280, 640, 1280, 0, 256, 64),        # up_blocks.0 resnet 2 conv1: [hidden 1280 | skip 640], groups of 60 straddle the boundary by 40 channels
    (1, 32, 32, 320, 0, 320, 0, 256, 0),
    (2, 32, 32, 640, 0, 640, 320, 256, 0),           # conv2 + appended conv_shortcut block
    (1, 64, 64, 320, 320, 320, 0, 256, 1024),        # up_blocks.2: two sources, 16 and 4 slots per image
    (3, 16, 16, 128, 64, 96, 64, 128, 256),
    (1, 16, 48, 64, 0, 160, 0, 256, 0),
    (1, 96, 96, 320, 0, 320, 0, 1024, 0),            # BASELINE configs[4] (768 x 768): 9 216 pixels per image = 9 slots of 1 024 rows (the canonical pass: 36 tile slots would be too many)
    (2, 48, 48, 640, 320, 640, 0, 256, 576),         # its 48 x 48 level: 9 slots of 256 rows and 4 slots of 576
]


@pytest.mark.parametrize("case", CASES)
def test_fused_groupnorm_conv_vs_fp32_and_its_unfused_pair(L, case):
    B, H, W, C0, C1, Co, Ca, rows0, rows1 = case
    f = _ffi()
    HW, Cin, M = H * W, C0 + C1, B * H * W
    x0 = rnd(M, C0, seed=41, scale=1.3, shift=0.5)
    x1 = rnd(M, C1, seed=42, scale=0.7, shift=-0.3) if C1 else None
    gamma, beta = rnd(Cin, seed=43, scale=0.3, shift=1.0), rnd(Cin, seed=44, scale=0.2)
    w = rnd(Co, Cin, 3, 3, seed=45, scale=(9 * Cin) ** -0.5)
    b, tv, res = rnd(Co, seed=46), rnd(B, Co, seed=47), rnd(M, Co, seed=48)
    xa = rnd(M, Ca, seed=49) if Ca else None
    wsc = rnd(Co, Ca, seed=50, scale=max(Ca, 1) ** -0.5) if Ca else None
    wp = pack3(L, w)
    wfull = torch.cat([wp, wsc], dim=1).contiguous() if Ca else wp
    st0 = colstats(L, x0, rows0)
    st1 = colstats(L, x1, rows1) if C1 else None
    xcat = torch.cat([x0, x1], dim=1) if C1 else x0
    act = gn_ref(xcat, gamma, beta, 32, 1e-5, B, H, W)
    ref = F.conv2d(act, w.float(), b.float(), padding=1) + tv.float()[:, :, None, None]
    if Ca:
        ref = ref + F.conv2d(xa.float().reshape(B, H, W, Ca).permute(0, 3, 1, 2), wsc.float()[:, :, None, None])
    ref = ref.permute(0, 2, 3, 1).reshape(M, Co) + res.float()
    # the unfused pair: the normalisation as a pass of its own (same statistics), then the plain halo-staged convolution
    n = torch.empty(M, Cin, dtype=torch.half, device="cuda")
    run(L, "ia2p_gn_apply_stats", f.ptr(x0), C0, C.c_void_p(st0.data_ptr()), rows0, f.ptr(x1), C1, C.c_void_p(st1.data_ptr()) if C1 else None, rows1, f.ptr(gamma), f.ptr(beta), f.ptr(n), B, HW, 32, 1e-5, 1)
    outs = {}
    try:
        for tile in (24, 25, 26):
            L.ia2p_debug_set_gemm_tile(tile)
            for sk in (0, 2, 3):
                if sk > Cin // 64:
                    continue
                kw = dict(wp=wfull, bias=b, rowvec=tv, residual=res, B=B, H=H, W=W, Co=Co, xa=xa, splitk=sk)
                fused, st, rows = conv_gn(L, x0=x0, st0=st0, rows0=rows0, x1=x1, st1=st1, rows1=rows1, gamma=gamma, beta=beta, want_stats=True, **kw)
                assert rel_l2(fused, ref) < 2e-3, (tile, sk, rel_l2(fused, ref))
                pair, _, _ = conv_gn(L, x0=n, **kw)
                assert torch.equal(fused, pair), (tile, sk, float((fused.float() - pair.float()).abs().max()))
                if rows:
                    per_image = lambda t: t.reshape(B, HW // rows, Co, 2).sum(1)
                    assert torch.equal(per_image(st), per_image(colstats(L, fused, rows))), (tile, sk)
                outs[(tile, sk)] = fused
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    for (tile, sk), y in outs.items():       # same accumulation order in every halo tile width
        assert torch.equal(y, outs[(24, sk)]), (tile, sk)


def test_fused_border_pixels_are_padded_after_the_activation(L):
    """silu(norm(0)) != 0: a kernel that normalised its zero padding would add beta-dependent garbage along the image border. A constant input makes every output pixel
    depend on the border handling alone: interior pixels see nine taps of silu(beta), edge pixels six, corner pixels four."""
    B, H, W, Cc, Co = 1, 16, 16, 64, 64
    x = torch.full((B * H * W, Cc), 0.5, dtype=torch.half, device="cuda")
    gamma, beta = torch.ones(Cc, dtype=torch.half, device="cuda"), torch.full((Cc,), 1.5, dtype=torch.half, device="cuda")
    w = torch.zeros(Co, Cc, 3, 3, dtype=torch.half, device="cuda")
    w[:, 0] = 1.0 / 16
    wp = pack3(L, w)
    st = colstats(L, x, 256)
    L.ia2p_debug_set_gemm_tile(25)
    try:
        y, _, _ = conv_gn(L, x0=x, st0=st, rows0=256, gamma=gamma, beta=beta, groups=32, wp=wp, B=B, H=H, W=W, Co=Co)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    v = float(F.silu(torch.tensor(1.5))) / 16       # (x - mean) = 0 everywhere: the activation is silu(beta)
    img = y.float().reshape(H, W, Co)[:, :, 0]
    assert abs(float(img[8, 8]) - 9 * v) < 2e-2 and abs(float(img[0, 8]) - 6 * v) < 2e-2 and abs(float(img[0, 0]) - 4 * v) < 2e-2, (float(img[8, 8]), float(img[0, 8]), float(img[0, 0]), v)


def test_fused_form_refuses_what_it_cannot_take(L):
    x = rnd(256, 64, seed=1)
    w = pack3(L, rnd(64, 64, 3, 3, seed=2))
    st = colstats(L, x, 256)
    g, b = rnd(64, seed=3), rnd(64, seed=4)
    L.ia2p_debug_set_gemm_tile(0)          # a gathered tile: no fused form
    try:
        conv_gn(L, x0=x, st0=st, rows0=256, gamma=g, beta=b, wp=w, B=1, H=16, W=16, Co=64, expect_fail=True)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
    L.ia2p_debug_set_gemm_tile(24)
    try:
        conv_gn(L, x0=x, st0=st, rows0=96, gamma=g, beta=b, wp=w, B=1, H=16, W=16, Co=64, expect_fail=True)       # slots that do not divide the image
        conv_gn(L, x0=x, st0=st, rows0=256, gamma=None, beta=b, wp=w, B=1, H=16, W=16, Co=64, expect_fail=True)
    finally:
        L.ia2p_debug_set_gemm_tile(-1)
