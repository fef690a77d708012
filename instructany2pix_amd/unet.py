"""HIP-backed conditional UNet with the call surface the reference uses on diffusers' UNet2DConditionModel.

What callers in the reference touch, and where it lives here:
  unet(sample, t, encoder_hidden_states=, cross_attention_kwargs=None, added_cond_kwargs={...}, return_dict=False)[0]
        instructany2pix/ddim/pnp_pipeline.py:253-260; ddim/sdxl_pipeline.py:832-839          -> __call__
  unet.config.{in_channels, addition_time_embed_dim, cross_attention_dim, block_out_channels, sample_size}
        pnp_pipeline.py:44-47; diffusion/ip_adapter/ip_adapter.py:114,124-132                   -> .config
  unet.add_embedding.linear_1.in_features       pnp_pipeline.py:47                            -> .add_embedding
  unet.attn_processors / unet.set_attn_processor(...)   ip_adapter.py:123,142,154,168          -> same names
All arithmetic runs in libia2p_hip.so (`ia2p_unet_forward`); this class owns the flat weight arena and the
activation workspace as torch tensors (device memory plumbing only) and keeps no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from types import SimpleNamespace
from typing import Dict, Iterable, Optional, Tuple, Union

import torch

from . import _ffi
from .attention_processor import AttnProcessor2_0, IPAttnProcessor2_0
from .config import UNetConfig
from .weights import attn_processor_names


class HipUNet2DConditionModel:
    dtype = torch.float16

    def __init__(self, config: UNetConfig, device: Union[str, torch.device] = "cuda:0"):
        config.validate()
        self.config = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _ffi.IA2PError("HipUNet2DConditionModel runs only on an MI355X device (no CPU path exists)")
        self._lib = _ffi.lib()
        torch.cuda.set_device(self.device)
        if not self._lib.ia2p_device_is_gfx950():
            raise _ffi.IA2PError("libia2p_hip.so is compiled for gfx950 only")
        self._ctx = C.c_void_p()
        _ffi.check(self._lib.ia2p_create(C.byref(_ffi.make_config(config)), C.byref(self._ctx)))
        nbytes = self._lib.ia2p_arena_bytes(self._ctx)
        self.arena = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)     # the ONE flat weight buffer
        _ffi.check(self._lib.ia2p_bind_arena(self._ctx, _ffi.ptr(self.arena), nbytes), self._ctx)
        self._workspace: Optional[torch.Tensor] = None
        self._ws_key = None
        # context K/V hoisting: the cross-attention K/V projections of a context are computed once and reused while the SAME context
        # tensor object (unmodified: torch's version counter) keeps arriving, i.e. over the denoise steps of a request
        self.cache_context_kv = True
        self._kv = None                 # (context tensor kept alive, its _version, ip (enabled, tokens), weight generation, kv buffer)
        self._weights_gen = 0
        self._names = attn_processor_names(config)
        self._procs: "OrderedDict[str, torch.nn.Module]" = OrderedDict((n, AttnProcessor2_0()) for n in self._names)
        self._ip_sig = None
        self.add_embedding = SimpleNamespace(linear_1=SimpleNamespace(in_features=config.projection_class_embeddings_input_dim))

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self._lib.ia2p_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    # ---- torch.nn.Module look-alikes the pipelines call --------------------------------------------------------
    def to(self, *a, **kw):
        return self

    def eval(self):
        return self

    # ---- weights --------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict: Union[Dict[str, torch.Tensor], Iterable[Tuple[str, torch.Tensor]]], strict: bool = True):
        """Load parameters by diffusers key. Accepts a dict or an iterator of (key, tensor) so 5 GB of weights
        never need to be resident twice."""
        items = state_dict.items() if hasattr(state_dict, "items") else state_dict
        torch.cuda.set_device(self.device)
        for k, v in items:
            self._load(k, v)
        torch.cuda.synchronize(self.device)
        if strict:
            _ffi.check(self._lib.ia2p_finalize_weights(self._ctx), self._ctx)

    def _load(self, key: str, v: torch.Tensor):
        self._weights_gen += 1
        t = v.detach().to(device=self.device, dtype=torch.float16).contiguous()
        shape = (C.c_int64 * t.ndim)(*t.shape)
        _ffi.check(self._lib.ia2p_load_tensor(self._ctx, key.encode(), _ffi.ptr(t), shape, t.ndim, _ffi.current_stream()), self._ctx)
        torch.cuda.current_stream().synchronize()      # `t` may be a temporary

    @property
    def arena_raw(self) -> torch.Tensor:
        """Head of the arena: the parameters as loaded (5.8 GB for SDXL-base + IP-Adapter). The tail behind it holds data derived at
        finalize (LayerNorm-folded weight copies, 2.5 GB) that every rank can recompute: only this view needs to travel."""
        return self.arena[: self._lib.ia2p_arena_raw_bytes(self._ctx)]

    def adopt_arena(self, with_ip_adapter: bool = True):
        """The arena HEAD was produced elsewhere (RCCL broadcast of `arena_raw` from rank 0): mark parameters present and derive the
        tail (LayerNorm folds) locally with the same kernel rank 0 used, so ranks hold bit-identical arenas."""
        _ffi.check(self._lib.ia2p_adopt_arena_on(self._ctx, int(with_ip_adapter), _ffi.current_stream()), self._ctx)   # ordered behind the broadcast
        self._weights_gen += 1

    # ---- operator-plugin API (reference ip_adapter.py:120-154) ----------------------------------------------------
    @property
    def attn_processors(self) -> "OrderedDict[str, torch.nn.Module]":
        return OrderedDict(self._procs)

    def set_attn_processor(self, processor):
        if isinstance(processor, dict):
            if set(processor) != set(self._names):
                raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match "
                                 f"the number of attention layers: {len(self._names)}.")
            self._procs = OrderedDict((n, processor[n]) for n in self._names)
        else:
            self._procs = OrderedDict((n, processor) for n in self._names)
        self._ip_sig = None

    def _sync_processors(self):
        """Push the installed plugins into the HIP context when they changed (weights, scale or topology)."""
        ips = [(n, p) for n, p in self._procs.items() if isinstance(p, IPAttnProcessor2_0)]
        if not ips:
            sig = ("off",)
            if sig != self._ip_sig:
                _ffi.check(self._lib.ia2p_set_ip_adapter(self._ctx, 0, 4, 1.0), self._ctx)
                self._ip_sig = sig
            return
        bad = [n for n, p in ips if n.endswith("attn1.processor")]
        if bad or len(ips) != len(self._names) // 2:
            raise NotImplementedError("IP processors must sit on every attn2 and only there (reference ip_adapter.py:123-141)")
        scales = {float(p.scale) for _, p in ips}
        toks = {int(p.num_tokens) for _, p in ips}
        if len(scales) != 1 or len(toks) != 1:
            raise NotImplementedError("per-layer IP scales / token counts are not supported (the reference sets one value, ip_adapter.py:211-214)")
        if all(p.to_k_ip.weight.is_meta for _, p in ips):
            wsig = "arena"           # descriptors only: weights were streamed straight into the arena (load_ip_adapter_weights)
        else:
            wsig = hash(tuple((p.to_k_ip.weight.data_ptr(), p.to_k_ip.weight._version, p.to_v_ip.weight.data_ptr(), p.to_v_ip.weight._version) for _, p in ips))
        sig = ("on", scales.pop(), toks.pop(), wsig)
        if sig == self._ip_sig:
            return
        if wsig != "arena" and (self._ip_sig is None or self._ip_sig[0] != "on" or self._ip_sig[3] != sig[3]):
            for n, p in ips:
                idx = self._names.index(n)
                self._load(f"ip_adapter.{idx}.to_k_ip.weight", p.to_k_ip.weight)
                self._load(f"ip_adapter.{idx}.to_v_ip.weight", p.to_v_ip.weight)
        _ffi.check(self._lib.ia2p_set_ip_adapter(self._ctx, 1, sig[2], sig[1]), self._ctx)
        self._ip_sig = sig

    def load_ip_adapter_weights(self, items, scale: float = 1.0, num_tokens: int = 4):
        """Stream `{"<idx>.to_k_ip.weight": tensor, ...}` (the "ip_adapter" group of the checkpoint, reference
        ip_adapter.py:165-169) straight into the arena and install storage-free IP processor descriptors: the
        0.68 GB of adapter weights then exist once, in the arena, not twice."""
        from .weights import hidden_size_of
        it = items.items() if hasattr(items, "items") else items
        for k, v in it:
            self._load("ip_adapter." + k, v)
        procs = {}
        for n in self._names:
            if n.endswith("attn1.processor"):
                procs[n] = AttnProcessor2_0()
            else:
                with torch.device("meta"):
                    procs[n] = IPAttnProcessor2_0(hidden_size_of(self.config, n), self.config.cross_attention_dim, scale=scale, num_tokens=num_tokens)
        self.set_attn_processor(procs)

    # ---- forward -----------------------------------------------------------------------------------------------------
    def workspace_for(self, B: int, h: int, w: int, L: int) -> torch.Tensor:
        # (ip on/off, scale, tokens); the kernel plan table decides the K-split slabs, so its generation is part of the key
        key = (B, h, w, L, self._ip_sig[:3] if self._ip_sig else None, self._lib.ia2p_plan_generation())
        if self._ws_key != key:
            n = self._lib.ia2p_workspace_bytes(self._ctx, B, h, w, L)
            if n == 0:
                _ffi.check(2, self._ctx)
            if self._workspace is None or self._workspace.numel() < n:
                self._workspace = torch.empty(n, dtype=torch.uint8, device=self.device)
            self._ws_key = key
        return self._workspace

    def __call__(self, sample, timestep, encoder_hidden_states=None, cross_attention_kwargs=None, added_cond_kwargs=None,
                 return_dict: bool = False, out: Optional[torch.Tensor] = None, _tune_reps: Optional[int] = None,
                 ip_scales: Optional[torch.Tensor] = None, **unused):
        """`timestep`: a number / 0-dim tensor as the reference passes it, or a [B] tensor -- one timestep per batch element, as diffusers' UNet
        accepts (`ia2p_unet_forward_v`); `ip_scales` (or cross_attention_kwargs={"ip_scales": ...}): optional [B] tensor, the IP-Adapter scale of
        every batch element (the reference sets one value per call, ip_adapter.py:211-214). Both let independent requests share one evaluation."""
        if encoder_hidden_states is None or added_cond_kwargs is None:
            raise ValueError("encoder_hidden_states and added_cond_kwargs (text_embeds, time_ids) are required (text_time UNet)")
        if "text_embeds" not in added_cond_kwargs or "time_ids" not in added_cond_kwargs:
            raise ValueError("added_cond_kwargs must carry `text_embeds` and `time_ids`")      # diffusers raises ValueError here too
        self._sync_processors()
        B, c_in, h, w = sample.shape
        if c_in != self.config.in_channels:
            raise ValueError(f"sample has {c_in} channels, expected {self.config.in_channels}")
        f16 = lambda t: t.to(device=self.device, dtype=torch.float16).contiguous()
        ctx_in = encoder_hidden_states
        sample, ctx = f16(sample), f16(encoder_hidden_states)
        te, tid = f16(added_cond_kwargs["text_embeds"]), f16(added_cond_kwargs["time_ids"])
        if ctx.shape[0] != B or ctx.shape[2] != self.config.cross_attention_dim:
            raise ValueError(f"encoder_hidden_states must be [B, L, {self.config.cross_attention_dim}]")
        if te.shape != (B, self.config.pooled_dim) or tid.shape != (B, self.config.num_time_ids):
            raise ValueError(f"Model expects an added time embedding vector of length {self.config.projection_class_embeddings_input_dim}, "
                             f"but a vector of {te.shape[-1] + tid.shape[-1] * self.config.addition_time_embed_dim} was created.")
        L = ctx.shape[1]
        ws = self.workspace_for(B, h, w, L)
        if out is None:
            out = torch.empty(B, self.config.out_channels, h, w, dtype=torch.float16, device=self.device)
        if ip_scales is None and cross_attention_kwargs:
            ip_scales = cross_attention_kwargs.get("ip_scales")
        per_sample = (torch.is_tensor(timestep) and timestep.numel() > 1) or ip_scales is not None
        if per_sample:
            if _tune_reps is not None:
                raise ValueError("autotune takes a scalar timestep")
            ts = timestep if torch.is_tensor(timestep) else torch.full((B,), float(timestep))
            ts = ts.to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
            if ts.numel() == 1:
                ts = ts.expand(B).contiguous()
            if ts.numel() != B:
                raise ValueError(f"timestep tensor has {ts.numel()} elements for a batch of {B}")
            sc = None
            if ip_scales is not None:
                sc = torch.as_tensor(ip_scales).to(device=self.device, dtype=torch.float32).reshape(-1).contiguous()
                if sc.numel() != B:
                    raise ValueError(f"ip_scales has {sc.numel()} elements for a batch of {B}")
            kv = self._context_kv(ctx_in, ctx, B, L, ws) if self.cache_context_kv else None
            _ffi.check(self._lib.ia2p_unet_forward_v(self._ctx, _ffi.current_stream(), _ffi.ptr(sample), _ffi.ptr(ts), _ffi.ptr(sc) if sc is not None else None,
                                                     None if kv is not None else _ffi.ptr(ctx), _ffi.ptr(kv) if kv is not None else None, L, _ffi.ptr(te), _ffi.ptr(tid),
                                                     _ffi.ptr(out), B, h, w, _ffi.ptr(ws), ws.numel()), self._ctx)
            return (out,) if not return_dict else SimpleNamespace(sample=out)
        t = float(timestep.item()) if torch.is_tensor(timestep) else float(timestep)
        args = (self._ctx, _ffi.current_stream(), _ffi.ptr(sample), t, _ffi.ptr(ctx), L, _ffi.ptr(te), _ffi.ptr(tid), _ffi.ptr(out), B, h, w,
                _ffi.ptr(ws), ws.numel())
        if _tune_reps is not None:
            sites = C.c_int(0)
            _ffi.check(self._lib.ia2p_autotune(*args, int(_tune_reps), C.addressof(sites)), self._ctx)
            return sites.value
        if self.cache_context_kv:
            kv = self._context_kv(ctx_in, ctx, B, L, ws)
            _ffi.check(self._lib.ia2p_unet_forward_kv(self._ctx, _ffi.current_stream(), _ffi.ptr(sample), t, _ffi.ptr(kv), L, _ffi.ptr(te), _ffi.ptr(tid),
                                                      _ffi.ptr(out), B, h, w, _ffi.ptr(ws), ws.numel()), self._ctx)
        else:
            _ffi.check(self._lib.ia2p_unet_forward(*args), self._ctx)
        return (out,) if not return_dict else SimpleNamespace(sample=out)

    def set_gn_fuse(self, mode: int):
        """GroupNorm + SiLU of the ResnetBlock2Ds: 1 (default) inside the halo-staged 3x3 convolutions that consume them (statistics from the producers' epilogues),
        0 as GroupNorm launches of their own, 2 the fused path's unfused twin (same statistics, same bits as 1; tests). `ia2p_set_gn_fuse`."""
        _ffi.check(self._lib.ia2p_set_gn_fuse(self._ctx, int(mode)), self._ctx)

    def invalidate_context_kv(self):
        """Forget the cached context K/V projections (a new request begins: the next evaluation projects its context again); the buffer is kept."""
        if self._kv is not None:
            self._kv = (None, -1, None, -1, self._kv[4], self._kv[5])

    def _context_kv(self, ctx_in, ctx, B, L, ws):
        """K/V projections of `ctx` (ia2p_project_context), cached while the same unmodified tensor object arrives with the same weights
        and IP-Adapter topology. The source tensor is kept alive so its address cannot be handed to another tensor."""
        ip = self._ip_sig[:1] + self._ip_sig[2:3] if self._ip_sig else None          # (on/off, tokens): the split of the context rows
        k = self._kv
        if k is not None and k[0] is ctx_in and k[1] == ctx_in._version and k[2] == ip and k[3] == self._weights_gen and k[4] == (B, L):
            return k[5]
        n = self._lib.ia2p_context_kv_bytes(self._ctx, B, L)
        if n == 0:
            _ffi.check(2, self._ctx)
        buf = k[5] if (k is not None and k[5].numel() == n) else torch.empty(n, dtype=torch.uint8, device=self.device)
        self._kv = None
        _ffi.check(self._lib.ia2p_project_context(self._ctx, _ffi.current_stream(), _ffi.ptr(ctx), L, B, _ffi.ptr(buf), n, _ffi.ptr(ws), ws.numel()), self._ctx)
        self._kv = (ctx_in, ctx_in._version, ip, self._weights_gen, (B, L), buf)
        return buf

    def autotune(self, sample, timestep, encoder_hidden_states, added_cond_kwargs, reps: int = 7) -> int:
        """Measure the candidate (tile, K-split) plans of every GEMM / conv shape of this call in place and keep the fastest
        (ia2p_autotune; process-wide table, see export_plans / import_plans). Returns the number of shapes measured.
        Use inputs shaped like the real ones (random values, not zeros). Optional: without it the built-in cost model picks."""
        for _ in range(4):          # bring the clocks up first: candidates measured on a cold GPU would all look slow, the first ones most
            self(sample, timestep, encoder_hidden_states=encoder_hidden_states, added_cond_kwargs=added_cond_kwargs)
        return self(sample, timestep, encoder_hidden_states=encoder_hidden_states, added_cond_kwargs=added_cond_kwargs, _tune_reps=reps)

    # ---- per-kernel-class timing for the roofline leg of bench.py ------------------------------------------------
    def profile(self, on: bool):
        _ffi.check(self._lib.ia2p_profile_enable(self._ctx, int(on)), self._ctx)

    def profile_read_regions(self):
        """{"other" | "conv_blocks" | "transformer": dict(launches, ms, flops, bytes)} summed since profile(True): which part of the network
        the launches belonged to (conv blocks = conv_in/out, ResnetBlock2D incl. GroupNorm+SiLU and shortcuts, resample convs, concat)."""
        res = {}
        for r, name in enumerate(("other", "conv_blocks", "transformer")):
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            _ffi.check(self._lib.ia2p_profile_read_region(self._ctx, r, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), self._ctx)
            res[name] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value)
        return res

    def profile_read_roles(self):
        """{layer role: dict(launches, ms, flops, bytes)} summed since profile(True): keyed by the executor's call site (FF-in, FF-out, QKV + self-attention,
        out-projections, to_q + cross-attention, 3x3 convolutions, GroupNorm, ...), whatever kernel instantiation the plan table picked."""
        res = {}
        for r in range(self._lib.ia2p_profile_roles()):
            name = C.create_string_buffer(128)
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            _ffi.check(self._lib.ia2p_profile_read_role(self._ctx, r, name, 128, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), self._ctx)
            if n.value:
                kernels = {}
                for k in range(self._lib.ia2p_profile_classes()):
                    kn, kms = C.c_int64(), C.c_double()
                    _ffi.check(self._lib.ia2p_profile_read_role_class(self._ctx, r, k, C.byref(kn), C.byref(kms)), self._ctx)
                    if kn.value:
                        kname = C.create_string_buffer(96)
                        _ffi.check(self._lib.ia2p_profile_read(self._ctx, k, kname, 96, None, None, None, None), self._ctx)
                        kernels[kname.value.decode()] = dict(launches=kn.value, ms=kms.value)
                res[name.value.decode()] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value, kernels=kernels)
        return res

    def profile_read(self):
        """{kernel name: dict(launches, ms, flops, bytes)} summed since profile(True)."""
        res = {}
        for k in range(self._lib.ia2p_profile_classes()):
            name = C.create_string_buffer(96)
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            _ffi.check(self._lib.ia2p_profile_read(self._ctx, k, name, 96, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), self._ctx)
            if n.value:
                pf = C.c_double()
                _ffi.check(self._lib.ia2p_profile_read_prefetch(self._ctx, k, C.byref(pf)), self._ctx)
                res[name.value.decode()] = dict(launches=n.value, ms=ms.value, flops=fl.value, bytes=by.value, prefetch_bytes=pf.value)
        return res


def export_plans() -> str:
    """measured kernel plans as text ("M,N,K,conv,geglu,variant,splitk;..."), e.g. to broadcast from rank 0"""
    lib = _ffi.lib()
    n = lib.ia2p_plan_export(None, 0)
    buf = C.create_string_buffer(n + 1)
    lib.ia2p_plan_export(buf, n + 1)
    return buf.value.decode()


def import_plans(text: str) -> int:
    n = _ffi.lib().ia2p_plan_import(text.encode())
    if n < 0:
        raise ValueError("malformed kernel plan table")
    return n


def clear_plans() -> None:
    _ffi.lib().ia2p_plan_clear()


def build_unet(config: UNetConfig, state_dict=None, device="cuda:0") -> HipUNet2DConditionModel:
    m = HipUNet2DConditionModel(config, device)
    if state_dict is not None:
        m.load_state_dict(state_dict)
    return m
