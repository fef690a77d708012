"""Where a launch of one layer role spends its time INSIDE a denoise step, next to the same launch back to back on warm operands (the attention
out-projections run 19.4 us in the step against 8.9 us warm). Needs a DIAGNOSTIC build of the library:
    IA2P_EXTRA_FLAGS="-DIA2P_CLOCK_STAMP -DIA2P_STAMP_AT=4" python tools/insitu_stamps.py [role index, default 4 = attention out-projections] [--plans FILE]
(the build stamps s_memrealtime -- a 100 MHz counter shared by the whole chip -- at workgroup entry, k-loop start, first k-tile landed, k-loop end, last C store issued, stores
drained; the product library is built WITHOUT the stamps). Reference op: `attn.to_out[0]` of both attention processors, attention_processor.py:267,400."""
import ctypes as C
import os
import sys
import statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from instructany2pix_amd import _ffi
from instructany2pix_amd.config import sdxl_base
from instructany2pix_amd.unet import HipUNet2DConditionModel, import_plans
from instructany2pix_amd.weights import unet_param_specs, ip_adapter_specs, iter_synthetic
import bench

role = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].lstrip("-").isdigit() else 4
plans = sys.argv[sys.argv.index("--plans") + 1] if "--plans" in sys.argv else bench.DEFAULT_PLANS
if os.environ.get("IA2P_STAMP_LIB"):      # a diagnostic library built beside the product one (built in the build container: saves GPU minutes)
    _ffi.LIB_PATH = os.path.abspath(os.environ["IA2P_STAMP_LIB"])
L = _ffi.lib()
if not hasattr(L, "ia2p_debug_stamp_begin"):
    raise SystemExit("this library has no stamps: rebuild with IA2P_EXTRA_FLAGS='-DIA2P_CLOCK_STAMP -DIA2P_STAMP_AT=4'")
L.ia2p_debug_stamp_begin.restype = C.c_int
L.ia2p_debug_stamp_begin.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
L.ia2p_debug_stamp_read.restype = C.c_int
L.ia2p_debug_stamp_read.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
cfg = sdxl_base()
unet = HipUNet2DConditionModel(cfg, dev)
unet.load_state_dict(iter_synthetic(unet_param_specs(cfg), 7, dev, torch.float16))
unet.load_ip_adapter_weights(iter_synthetic(ip_adapter_specs(cfg)["ip_adapter"], 7, dev, torch.float16), scale=1.0, num_tokens=4)
if os.path.exists(plans):
    import_plans("".join(l for l in open(plans).read().splitlines() if not l.startswith("#")).strip())
wl = bench.Workload(unet, cfg, 8, 64, 81, 0.0, dev, cfg_id=3)
if not os.path.exists(plans):
    unet.autotune(wl.lat, wl.ts[0], wl.ctx, wl.added)
wl.run(5)
torch.cuda.synchronize()
SLOTS = 256
WG = L.ia2p_debug_stamp_begin(unet._ctx, -1, None, 0)
buf = torch.zeros(SLOTS * WG * 8, dtype=torch.int64, device=dev)
L.ia2p_debug_stamp_begin(unet._ctx, role, C.c_void_p(buf.data_ptr()), SLOTS)
wl.run(1)
torch.cuda.synchronize()
meta = (C.c_int * (5 * SLOTS))()
n = L.ia2p_debug_stamp_read(unet._ctx, meta, SLOTS)
L.ia2p_debug_stamp_begin(unet._ctx, -1, None, 0)
h = buf.cpu().view(SLOTS, WG, 8)


def breakdown(rec, tiles):
    r = rec[:tiles].double() * 0.01          # us (100 MHz counter)
    ok = rec[:tiles, 3] != 0
    r = r[ok]
    if r.shape[0] == 0:
        return None
    entry, l0, l1, left, issued, first = r[:, 2], r[:, 3], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
    med = lambda t: float(t.median())
    return {"wgs": int(r.shape[0]), "span": float(left.max() - entry.min()), "entry_ramp": float(entry.max() - entry.min()), "entry_to_loop": med(l0 - entry),
            "loop_to_first_tile": med(first - l0) if float(first.max()) > 0 else float("nan"), "first_tile_to_loop_end": med(l1 - first) if float(first.max()) > 0 else med(l1 - l0),
            "loop_end_to_last_store": med(issued - l1), "drain": med(left - issued), "wg_lifetime": med(left - entry), "exit_skew": float(left.max() - left.min())}


def breakdown_fused(rec, tiles):
    """the fused QKV + self-attention tile (variant -4): slot 7 = Q / K / V images in LDS, slot 6 = attention core done (wave 0), slot 5 = O stored and drained"""
    r = rec[:tiles].double() * 0.01
    r = r[rec[:tiles, 3] != 0]
    if r.shape[0] == 0:
        return None
    entry, l0, l1, left, core, img = r[:, 2], r[:, 3], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
    med = lambda t: float(t.median())
    return {"wgs": int(r.shape[0]), "span": float(left.max() - entry.min()), "entry_ramp": float(entry.max() - entry.min()), "entry_to_loop": med(l0 - entry), "k_loop": med(l1 - l0),
            "loop_end_to_images": med(img - l1), "attention_core": med(core - img), "store_O": med(left - core), "wg_lifetime": med(left - entry), "exit_skew": float(left.max() - left.min())}


def breakdown_xattn(rec, tiles):
    """the fused to_q + cross-attention tile (variant -5): slot 6 = Q fragments built, K / V images in LDS, slot 7 = attention core done, slot 5 = O stored and drained"""
    r = rec[:tiles].double() * 0.01
    r = r[rec[:tiles, 3] != 0]
    if r.shape[0] == 0:
        return None
    entry, l0, l1, left, img, core = r[:, 2], r[:, 3], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
    med = lambda t: float(t.median())
    return {"wgs": int(r.shape[0]), "span": float(left.max() - entry.min()), "entry_ramp": float(entry.max() - entry.min()), "entry_to_loop": med(l0 - entry), "k_loop": med(l1 - l0),
            "k_loop_p90": float((l1 - l0).quantile(0.9)), "loop_end_to_q_and_kv_images": med(img - l1), "attention_core": med(core - img), "store_O": med(left - core),
            "wg_lifetime": med(left - entry), "wg_lifetime_p90": float((left - entry).quantile(0.9)), "exit_skew": float(left.max() - left.min())}


def breakdown_split(rec, tiles):
    """a launch with an in-launch K split (variant -(100 * split + v)): slot 0 = this slice's slab has drained (in front of the ticket); slots 6 / 5 are written by the LAST
    arriver of a tile only (epilogue's last store issued / drained), 0 for the slices that leave behind the ticket"""
    r = rec[:tiles].double() * 0.01
    r = r[rec[:tiles, 3] != 0]
    if r.shape[0] == 0:
        return None
    slab, entry, l0, l1, left, issued, first = r[:, 0], r[:, 2], r[:, 3], r[:, 4], r[:, 5], r[:, 6], r[:, 7]
    last = left > 0
    med = lambda t: float(t.median()) if t.numel() else float("nan")
    end = torch.where(last, left, slab)
    return {"wgs": int(r.shape[0]), "last_arrivers": int(last.sum()), "span": float(end.max() - entry.min()), "entry_ramp": float(entry.max() - entry.min()), "entry_to_loop": med(l0 - entry),
            "loop_to_first_tile": med(first - l0) if float(first.min()) > 0 and float((l1 - first).min()) > 0 else float("nan"), "k_loop": med(l1 - l0), "k_loop_p90": float((l1 - l0).quantile(0.9)),
            "loop_end_to_slab_drained": med(slab - l1), "last_arriver_ticket_to_last_store": med((issued - slab)[last]),
            "last_arriver_slab_drained_to_slot7(STAMP_AT=1: slabs summed; STAMP_AT=5: through the ticket and the acquire fence)": med((first - slab)[last]) if last.any() and float((first - l1)[last].min()) > 0 else float("nan"), "last_arriver_drain": med((left - issued)[last]),
            "lifetime_slice_that_leaves": med((slab - entry)[~last]), "lifetime_last_arriver": med((left - entry)[last]), "first_exit": float(end.min() - entry.min())}


cols = ("wgs", "span", "entry_ramp", "entry_to_loop", "loop_to_first_tile", "first_tile_to_loop_end", "loop_end_to_last_store", "drain", "wg_lifetime", "exit_skew")
print(f"# role {role}: {n} launches stamped inside one step (batch 8, 512 x 512, 81-token context); us; medians over the launch's workgroups unless noted")
print("# span = first workgroup entry -> last workgroup's stores drained; entry_ramp = first -> last workgroup entry; exit_skew = first -> last workgroup exit")
by_shape = {}
fused = {}
split = {}
xat = {}
for i in range(min(n, SLOTS)):
    M, N, K, v, tiles = meta[5 * i:5 * i + 5]
    if v == -4:
        b = breakdown_fused(h[i], tiles)
        if b:
            fused.setdefault((M, N, K), []).append(b)
        continue
    if v == -5:
        b = breakdown_xattn(h[i], tiles)
        if b:
            xat.setdefault((M, N, K), []).append(b)
        continue
    if v <= -100:
        b = breakdown_split(h[i], tiles)
        if b:
            split.setdefault((M, N, K, (-v) // 100, (-v) % 100), []).append(b)
        continue
    b = breakdown(h[i], tiles)
    if b:
        by_shape.setdefault((M, N, K, v), []).append(b)
# the 256 x 320 GEGLU tile in a -DIA2P_G320_PHASES build: per interval of a k-tile, the cycles a wave works and the cycles it then waits in the barrier (records behind the 8-slot ones)
if os.environ.get("IA2P_G320_PHASES"):
    what = (("I(4t)   read kk0 + 5 W pieces", "I(4t+1) 40 MFMAs", "I(4t+2) read kk1 + 4 A pieces", "I(4t+3) 40 MFMAs"),
            ("I(4t)   40 MFMAs (t-1, kk1)", "I(4t+1) read kk0 + 5 W pieces", "I(4t+2) 40 MFMAs", "I(4t+3) read kk1 + 4 A pieces"))
    for i in range(min(n, SLOTS)):
        M, N, K, v, tiles = meta[5 * i:5 * i + 5]
        if v != 27 or (M, N, K) != (2048, 10240, 1280):
            continue
        ph = h[i].reshape(-1)[8 * tiles:8 * tiles + 16 * tiles].reshape(tiles, 16).double() / (K // 64)
        print(f"launch {i}: {M} x {N} x {K}, cycles per k-tile (median over workgroups)")
        for grp in range(2):
            for k in range(4):
                print(f"    group {grp}  {what[grp][k]:32s} works {float(ph[:, 8 * grp + k].median()):6.0f}, waits in the barrier {float(ph[:, 8 * grp + 4 + k].median()):6.0f}")
        break
for (M, N, K), bs in sorted(xat.items()):
    print(f"fused to_q + cross-attention (128 x 64 tile), {M} x {N} x {K}, {len(bs)} launches in the step: " + ", ".join(f"{c} {statistics.median(b[c] for b in bs):.2f}" for c in bs[0]))
for (M, N, K, sk, v), bs in sorted(split.items()):
    print(f"K split {sk} in the launch (tile variant {v}), {M} x {N} x {K}, {len(bs)} launches in the step: " + ", ".join(f"{c} {statistics.median(b[c] for b in bs):.2f}" for c in bs[0]))
for (M, N, K), bs in sorted(fused.items()):
    fc = list(bs[0])
    print(f"fused QKV projection + self-attention (256 x 192 tile per image and head), {M} x {N} x {K}, {len(bs)} launches in the step: " + ", ".join(f"{c} {statistics.median(b[c] for b in bs):.2f}" for c in fc))
print(f"{'where':28s} {'M x N x K (variant)':28s} {'n':>4s} " + " ".join(f"{c:>22s}" for c in cols))
for (M, N, K, v), bs in sorted(by_shape.items()):
    print(f"{'in the step':28s} {f'{M} x {N} x {K} ({v})':28s} {len(bs):4d} " + " ".join(f"{statistics.median(b[c] for b in bs):22.2f}" for c in cols))
    # the same launch (bias, residual, row statistics of the output) back to back on warm operands, same tile variant
    A = (torch.randn(M, K, device=dev) * 0.5).half(); W = (torch.randn(N, K, device=dev) * K ** -0.5).half(); bias = torch.randn(N, device=dev).half(); R = torch.randn(M, N, device=dev).half()
    out = torch.empty(M, N, device=dev, dtype=torch.half); stats = torch.zeros(64 * M * 2, device=dev); slots = C.c_int(0)
    wb = torch.zeros(WG * 8, dtype=torch.int64, device=dev)
    L.ia2p_debug_set_gemm_tile(v)
    call = lambda: _ffi.check(L.ia2p_gemm_ex(_ffi.current_stream(), _ffi.ptr(A), _ffi.ptr(W), _ffi.ptr(bias), _ffi.ptr(R), _ffi.ptr(out), M, N, K, 0, None, _ffi.ptr(stats), C.addressof(slots), 1,
                                             C.c_void_p(wb.data_ptr())))
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        call()
    e1.record()
    torch.cuda.synchronize()
    L.ia2p_debug_set_gemm_tile(-1)
    tiles = bs[0]["wgs"]
    b = breakdown(wb.cpu().view(WG, 8), WG)
    print(f"{'back to back, warm':28s} {f'{M} x {N} x {K} ({v})':28s} {100:4d} " + " ".join(f"{b[c]:22.2f}" for c in cols) + f"   ({e0.elapsed_time(e1) * 10:.2f} us per launch by HIP events)")
