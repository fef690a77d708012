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


cols = ("wgs", "span", "entry_ramp", "entry_to_loop", "loop_to_first_tile", "first_tile_to_loop_end", "loop_end_to_last_store", "drain", "wg_lifetime", "exit_skew")
print(f"# role {role}: {n} launches stamped inside one step (batch 8, 512 x 512, 81-token context); us; medians over the launch's workgroups unless noted")
print("# span = first workgroup entry -> last workgroup's stores drained; entry_ramp = first -> last workgroup entry; exit_skew = first -> last workgroup exit")
by_shape = {}
for i in range(min(n, SLOTS)):
    M, N, K, v, tiles = meta[5 * i:5 * i + 5]
    b = breakdown(h[i], tiles)
    if b:
        by_shape.setdefault((M, N, K, v), []).append(b)
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
