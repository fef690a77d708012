"""Attention operator plugins for the HIP UNet, mirroring the reference's classes by name and constructor
(instructany2pix/diffusion/ip_adapter/attention_processor.py:191-203 AttnProcessor2_0, :282-308
IPAttnProcessor2_0) so the reference's installation code (ip_adapter.py:120-142) and checkpoint loading
(`ModuleList(unet.attn_processors.values()).load_state_dict`, :168-169) work unchanged.

On this path a processor is a DESCRIPTOR: the arithmetic of `__call__` (QKV projections, the two softmaxes,
`text + scale * ip`, out-projection) runs inside libia2p_hip.so (csrc/attention.hip, csrc/gemm.hip); the
UNet reads `scale`, `num_tokens` and the `to_k_ip` / `to_v_ip` weights from these objects.
"""
import torch
import torch.nn as nn


class AttnProcessor2_0(nn.Module):
    def __init__(self, hidden_size=None, cross_attention_dim=None):
        super().__init__()

    def __call__(self, *a, **kw):
        raise RuntimeError("HIP-path processors are descriptors: the UNet executes attention inside libia2p_hip.so")


AttnProcessor = AttnProcessor2_0


class IPAttnProcessor2_0(nn.Module):
    def __init__(self, hidden_size, cross_attention_dim=None, scale=1.0, num_tokens=4):
        super().__init__()
        self.hidden_size = hidden_size
        self.cross_attention_dim = cross_attention_dim
        self.scale = scale
        self.num_tokens = num_tokens
        self.to_k_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)
        self.to_v_ip = nn.Linear(cross_attention_dim or hidden_size, hidden_size, bias=False)

    def __call__(self, *a, **kw):
        raise RuntimeError("HIP-path processors are descriptors: the UNet executes attention inside libia2p_hip.so")


IPAttnProcessor = IPAttnProcessor2_0
