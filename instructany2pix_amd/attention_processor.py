"""Attention operator plugins on the HIP library, mirroring the reference's classes by name, constructor and call protocol
(instructany2pix/diffusion/ip_adapter/attention_processor.py:191-279 AttnProcessor2_0, :282-412 IPAttnProcessor2_0) so the reference's
installation code (ip_adapter.py:120-142) and checkpoint loading (`ModuleList(unet.attn_processors.values()).load_state_dict`, :168-169)
work unchanged.

Two ways these objects are used:
  * inside `HipUNet2DConditionModel` they are DESCRIPTORS: the executor in libia2p_hip.so runs the whole block and only reads `scale`,
    `num_tokens` and the `to_k_ip` / `to_v_ip` weights from them (unet.py::_sync_processors);
  * a host that keeps diffusers' own module tree installs them with `unet.set_attn_processor(...)` exactly like the reference does, and
    diffusers' `Attention.forward` then CALLS them: `proc(attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None)`.
    `__call__` below runs that protocol on the C ABI: projections through `ia2p_gemm` (fused q/k/v and k/v weight stacks, bias and
    residual in the epilogue), the softmax(QK^T/8)V cores -- two key segments with their own softmaxes and `text + scale * ip` for the
    IP variant -- through `ia2p_attention`, the optional `attn_map` side effect (:390-391) through `ia2p_ip_attn_map`.
There is no torch fallback: on a non-device tensor `_ffi.ptr` refuses. head_dim is 64 (every attention layer of SDXL).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _ffi


def _p(t, off_elems=0):
    return C.c_void_p(t.data_ptr() + 2 * off_elems)


def _f16c(t):
    return t.detach().to(dtype=torch.float16).contiguous()


class _HipAttnBase(nn.Module):
    """weight staging shared by the two processors: fp16 device copies of the projections the `attn` object holds, stacked for fused
    launches and zero-padded along K to the GEMM's 64-element k-tile; rebuilt when a source tensor is replaced or modified in place"""

    def _staged(self, tag, tensors, build):
        cache = self.__dict__.setdefault("_stage", {})
        key = tuple((t.data_ptr(), t._version, t.device, t.dtype) for t in tensors if t is not None)
        hit = cache.get(tag)
        if hit is None or hit[0] != key:
            cache[tag] = hit = (key, build())
        return hit[1]

    @staticmethod
    def _pad_k(w):
        k = w.shape[1]
        if k % 64 == 0:
            return w
        return torch.nn.functional.pad(w, (0, 64 - k % 64))

    def _stack(self, tag, linears, dev):
        """[sum N_i, Kpad] fp16 weight stack (+ bias stack or None) of a list of nn.Linear"""
        ws = [l.weight for l in linears]
        bs = [l.bias for l in linears]

        def build():
            w = self._pad_k(torch.cat([_f16c(x.to(dev)) for x in ws], dim=0)).contiguous()
            b = None
            if any(x is not None for x in bs):
                b = torch.cat([_f16c(x.to(dev)) if x is not None else torch.zeros(l.weight.shape[0], dtype=torch.float16, device=dev)
                               for x, l in zip(bs, linears)]).contiguous()
            return w, b
        return self._staged(tag, ws + bs, build)

    @staticmethod
    def _rows(x2d, kpad):
        """activation rows as a contiguous fp16 [M, kpad] device tensor"""
        x2d = x2d.to(torch.float16)
        if x2d.shape[1] != kpad:
            x2d = torch.nn.functional.pad(x2d, (0, kpad - x2d.shape[1]))
        return x2d.contiguous()

    @staticmethod
    def _gemm(x2d, w, bias, residual=None):
        M, K = x2d.shape
        N = w.shape[0]
        if N % 4:
            raise ValueError(f"projection width {N} must be a multiple of 4")
        out = torch.empty(M, N, dtype=torch.float16, device=x2d.device)
        _ffi.check(_ffi.lib().ia2p_gemm(_ffi.current_stream(), _ffi.ptr(x2d), _ffi.ptr(w), _ffi.ptr(bias), _ffi.ptr(residual), _ffi.ptr(out), M, N, K, 0))
        return out

    @staticmethod
    def _check(attn, hidden_states, attention_mask, temb):
        if attention_mask is not None:
            raise NotImplementedError("attention masks are not on the reference's path (no caller passes one)")
        if getattr(attn, "spatial_norm", None) is not None or getattr(attn, "group_norm", None) is not None or getattr(attn, "norm_cross", None):
            raise NotImplementedError("spatial_norm / group_norm / norm_cross are None on every SDXL attention layer")
        inner = attn.to_q.weight.shape[0]
        if inner != attn.heads * 64:
            raise ValueError(f"head_dim must be 64 (inner dim {inner}, {attn.heads} heads)")
        if not hidden_states.is_cuda:
            raise _ffi.IA2PError("the HIP attention processors run on device tensors only (no CPU path exists)")

    @staticmethod
    def _enter(hidden_states):
        """reference :223-228 / :328-333: 4-D inputs are flattened to tokens"""
        shape4 = None
        if hidden_states.ndim == 4:
            shape4 = hidden_states.shape
            b, c, h, w = shape4
            hidden_states = hidden_states.view(b, c, h * w).transpose(1, 2)
        return hidden_states, shape4

    @staticmethod
    def _leave(out, attn, shape4, residual4=None):
        if shape4 is not None:                                              # :271-272
            b, c, h, w = shape4
            out = out.transpose(-1, -2).reshape(b, c, h, w)
            if residual4 is not None:                                        # :274-275 on the 4-D form (3-D inputs: fused into the out-projection)
                out = out + residual4
        f = float(getattr(attn, "rescale_output_factor", 1.0))
        return out if f == 1.0 else out / f                                  # :277


class AttnProcessor2_0(_HipAttnBase):
    """reference attention_processor.py:191-279"""

    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()

    @torch.no_grad()
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *args, **kwargs):
        self._check(attn, hidden_states, attention_mask, temb)
        residual = hidden_states
        hidden_states, shape4 = self._enter(hidden_states)
        B, Nq, _ = hidden_states.shape
        dev, heads, inner = hidden_states.device, attn.heads, attn.to_q.weight.shape[0]
        L = _ffi.lib()
        o = torch.empty(B * Nq, inner, dtype=torch.float16, device=dev)
        if encoder_hidden_states is None:                                    # self-attention: one [3C, C] projection (:239,246-247)
            w, b = self._stack("qkv", [attn.to_q, attn.to_k, attn.to_v], dev)
            qkv = self._gemm(self._rows(hidden_states.reshape(B * Nq, -1), w.shape[1]), w, b)
            _ffi.check(L.ia2p_attention(_ffi.current_stream(), _p(qkv), 3 * inner, _p(o), inner, B, heads, Nq, 1,
                                        _p(qkv, inner), _p(qkv, 2 * inner), 3 * inner, Nq, 1.0, None, None, 0, 0, 0.0))
        else:
            wq, bq = self._stack("q", [attn.to_q], dev)
            wkv, bkv = self._stack("kv", [attn.to_k, attn.to_v], dev)
            Nk = encoder_hidden_states.shape[1]
            q = self._gemm(self._rows(hidden_states.reshape(B * Nq, -1), wq.shape[1]), wq, bq)
            kv = self._gemm(self._rows(encoder_hidden_states.reshape(B * Nk, -1), wkv.shape[1]), wkv, bkv)
            _ffi.check(L.ia2p_attention(_ffi.current_stream(), _p(q), inner, _p(o), inner, B, heads, Nq, 1,
                                        _p(kv), _p(kv, inner), 2 * inner, Nk, 1.0, None, None, 0, 0, 0.0))
        wo, bo = self._stack("out", [attn.to_out[0]], dev)                   # :267 (+ dropout(0) :269)
        res = None
        if getattr(attn, "residual_connection", False) and shape4 is None:  # :274-275 fused into the epilogue
            res = _f16c(residual).reshape(B * Nq, -1)
        out = self._gemm(self._rows(o, wo.shape[1]), wo, bo, res).reshape(B, Nq, -1)
        return self._leave(out, attn, shape4, residual if getattr(attn, "residual_connection", False) else None)


AttnProcessor = AttnProcessor2_0


class IPAttnProcessor2_0(_HipAttnBase):
    """reference attention_processor.py:282-412. `store_attn_map = True` also produces the `attn_map` side effect (:390-391)."""

    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self.num_tokens = num_tokens
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.store_attn_map = False
        self.attn_map = None
        self.fuse_to_q = True

    @torch.no_grad()
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, temb=None, *args, **kwargs):
        self._check(attn, hidden_states, attention_mask, temb)
        residual = hidden_states
        hidden_states, shape4 = self._enter(hidden_states)
        B, Nq, _ = hidden_states.shape
        dev, heads, inner = hidden_states.device, attn.heads, attn.to_q.weight.shape[0]
        if encoder_hidden_states is None:                                    # :346-347
            encoder_hidden_states = hidden_states
        end = encoder_hidden_states.shape[1] - self.num_tokens               # :350 (a 77-token context loses its last 4 text rows here)
        if end < 1:
            raise ValueError(f"context length {encoder_hidden_states.shape[1]} must exceed the {self.num_tokens} image tokens")
        text, ip = encoder_hidden_states[:, :end], encoder_hidden_states[:, end:]
        L = _ffi.lib()
        wq, bq = self._stack("q", [attn.to_q], dev)
        wkv, bkv = self._stack("kv", [attn.to_k, attn.to_v], dev)            # :358-359
        wip, _ = self._stack("kvip", [self.to_k_ip, self.to_v_ip], dev)      # :379-380
        x = self._rows(hidden_states.reshape(B * Nq, -1), wq.shape[1])
        kv = self._gemm(self._rows(text.reshape(B * end, -1), wkv.shape[1]), wkv, bkv)
        kvip = self._gemm(self._rows(ip.reshape(B * self.num_tokens, -1), wip.shape[1]), wip, None)
        o = torch.empty(B * Nq, inner, dtype=torch.float16, device=dev)
        segs = (2, _p(kv), _p(kv, inner), 2 * inner, end, 1.0, _p(kvip), _p(kvip, inner), 2 * inner, self.num_tokens, float(self.scale))
        # to_q (:344) and the two SDPA calls in ONE launch when a 128 x 64 tile of to_q is 128 queries of one head and the context is short
        # (what the executor does, DESIGN.md §4 qproj_xattn_kernel); attn_map needs Q in memory, so it keeps the two launches
        fused = self.fuse_to_q and not self.store_attn_map and Nq % 128 == 0 and (end + 63) // 64 + (self.num_tokens + 63) // 64 <= 3
        if fused:
            _ffi.check(L.ia2p_qproj_attention(_ffi.current_stream(), _p(x), _p(wq), _ffi.ptr(bq), None, _p(o), inner, B, heads, Nq, wq.shape[1], *segs))
        else:
            q = self._gemm(x, wq, bq)
            # :371 SDPA(q, k, v) + scale * :387 SDPA(q, ip_k, ip_v) -- two softmaxes, one launch (:397)
            _ffi.check(L.ia2p_attention(_ffi.current_stream(), _p(q), inner, _p(o), inner, B, heads, Nq, *segs))
        if self.store_attn_map:
            amap = torch.empty(B, heads, Nq, self.num_tokens, dtype=torch.float16, device=dev)
            _ffi.check(L.ia2p_ip_attn_map(_ffi.current_stream(), _p(q), inner, _p(kvip), 2 * inner, _p(amap), B, heads, Nq, self.num_tokens))
            self.attn_map = amap
        wo, bo = self._stack("out", [attn.to_out[0]], dev)                   # :400 (+ dropout(0) :402)
        res = None
        if getattr(attn, "residual_connection", False) and shape4 is None:  # :407-408
            res = _f16c(residual).reshape(B * Nq, -1)
        out = self._gemm(self._rows(o, wo.shape[1]), wo, bo, res).reshape(B, Nq, -1)
        return self._leave(out, attn, shape4, residual if getattr(attn, "residual_connection", False) else None)


IPAttnProcessor = IPAttnProcessor2_0
