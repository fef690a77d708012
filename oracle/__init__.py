"""ORACLE — test infrastructure only.

CPU (torch fp32) restatement of the reference's denoise hot path: conditional UNet forward with the
IP-Adapter attention plugins + DDIM inversion/sampling loop. It is the CHECKER for the HIP path:
only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and the
product package `instructany2pix_amd` never does (tests/test_abi_cpu.py::test_product_never_imports_oracle enforces that).

Parity pin: golden vectors in tests/golden/ generated from the reference's own importable files
(generator script tests/golden/gen_goldens.py). diffusers==0.26.3 itself is absent: wiring that only
diffusers owns is "parity unpinned" (see oracle/unet_ref.py header and DESIGN.md).
"""
from .unet_ref import UNet2DConditionModelRef, sinusoid
from .attn_processors_ref import AttnProcessor2_0Ref, IPAttnProcessor2_0Ref
from .vae_ref import AutoencoderKLRef, build_vae, sample_latents
from .ddim_ref import (DDIMSchedulerRef, backward_ddim, cfg_combine, get_add_time_ids, polar_interpolate,
                       ImageProjModelRef, invert_loop, sample_loop, inpaint_loop, add_noise)
from .clip_ref import CLIPTextModelRef, build_clip, encode_prompt_ref
from .euler_ref import EulerDiscreteSchedulerRef, get_add_time_ids_aesthetic, img2img_loop


def build_unet_fast(cfg, items, ip_items=None, ip_scale=1.0, num_tokens=4):
    """Same as build_unet for multi-GB weights: modules are created on the meta device and the fp32 tensors
    yielded by `items` / `ip_items` are ASSIGNED (no default init pass, no second copy)."""
    import torch
    with torch.device("meta"):
        m = UNet2DConditionModelRef(cfg)
    m.load_state_dict({k: v.float() for k, v in items}, strict=True, assign=True)
    if ip_items is not None:
        procs = {}
        with torch.device("meta"):
            for name in m.attn_processors.keys():
                if name.endswith("attn1.processor"):
                    procs[name] = AttnProcessor2_0Ref()
                else:
                    blk = name.split(".")
                    ch = cfg.block_out_channels
                    hs = ch[-1] if blk[0] == "mid_block" else (list(reversed(ch))[int(blk[1])] if blk[0] == "up_blocks" else ch[int(blk[1])])
                    procs[name] = IPAttnProcessor2_0Ref(hs, cfg.cross_attention_dim, scale=ip_scale, num_tokens=num_tokens)
        m.set_attn_processor(procs)
        torch.nn.ModuleList(m.attn_processors.values()).load_state_dict({k: v.float() for k, v in ip_items}, assign=True)
    return m.eval()


def build_unet(cfg, state_dict, ip_state=None, ip_scale=1.0, num_tokens=4, dtype=None):
    """Oracle UNet with weights loaded by diffusers key; optionally installs the IP-Adapter plugins the way
    reference ip_adapter.py:120-142,168-169 does (self-attn -> AttnProcessor, cross-attn -> IPAttnProcessor,
    weights loaded through a ModuleList over `attn_processors.values()`)."""
    import torch
    m = UNet2DConditionModelRef(cfg)
    m.load_state_dict({k: v.float() for k, v in state_dict.items()}, strict=True)
    if ip_state is not None:
        procs = {}
        for name in m.attn_processors.keys():
            if name.endswith("attn1.processor"):
                procs[name] = AttnProcessor2_0Ref()
            else:
                if name.startswith("mid_block"):
                    hs = cfg.block_out_channels[-1]
                elif name.startswith("up_blocks"):
                    hs = list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])]
                else:
                    hs = cfg.block_out_channels[int(name[len("down_blocks.")])]
                procs[name] = IPAttnProcessor2_0Ref(hs, cfg.cross_attention_dim, scale=ip_scale, num_tokens=num_tokens)
        m.set_attn_processor(procs)
        torch.nn.ModuleList(m.attn_processors.values()).load_state_dict({k: v.float() for k, v in ip_state.items()})
    if dtype is not None:
        m = m.to(dtype)
    return m.eval()
from .prior_ref import GPT2ModelRef, build_gpt2, DDPMSchedulerRef, PriorRef, timestep_embedding_ref  # noqa: E402,F401
