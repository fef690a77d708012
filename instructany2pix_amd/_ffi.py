"""ctypes binding of libia2p_hip.so (C ABI declared in include/ia2p.h; test hooks and the profile interface in include/ia2p_debug.h).

The product path has no CPU fallback: if the library is missing this module raises at import of the
symbols, and every compute entry point needs device pointers on an MI355X.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libia2p_hip.so")

IA2P_OK = 0
_STATUS_NAMES = {1: "INVALID", 2: "SHAPE", 3: "KEY", 4: "STATE", 5: "NOMEM", 6: "HIP", 7: "ARCH"}
MAX_BLOCKS = 4


class UNetConfigC(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int), ("out_channels", C.c_int), ("n_blocks", C.c_int),
        ("block_out_channels", C.c_int * MAX_BLOCKS), ("transformer_layers_per_block", C.c_int * MAX_BLOCKS),
        ("num_heads", C.c_int * MAX_BLOCKS), ("layers_per_block", C.c_int), ("cross_attention_dim", C.c_int),
        ("norm_num_groups", C.c_int), ("norm_eps", C.c_float), ("addition_time_embed_dim", C.c_int),
        ("projection_class_embeddings_input_dim", C.c_int), ("time_embed_dim", C.c_int), ("time_proj_dim", C.c_int),
        ("mid_transformer_layers", C.c_int), ("num_time_ids", C.c_int),
    ]


class LnFoldC(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("slots", C.c_int), ("colsum", C.c_void_p), ("fbias", C.c_void_p), ("eps", C.c_float)]


class CLIPConfigC(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("hidden_size", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int), ("intermediate_size", C.c_int),
                ("max_positions", C.c_int), ("projection_dim", C.c_int), ("hidden_act", C.c_int), ("eos_token_id", C.c_int), ("layer_norm_eps", C.c_float)]


class VAEConfigC(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("out_channels", C.c_int), ("latent_channels", C.c_int), ("n_blocks", C.c_int),
                ("block_out_channels", C.c_int * MAX_BLOCKS), ("layers_per_block", C.c_int), ("norm_num_groups", C.c_int),
                ("norm_eps", C.c_float), ("stream_scale", C.c_float)]


class ConvGnC(C.Structure):
    """ia2p_conv_gn (include/ia2p.h): a 3x3 convolution with the GroupNorm + SiLU in front of it applied inside the kernel"""
    _fields_ = [("x0", C.c_void_p), ("C0", C.c_int), ("st0", C.c_void_p), ("rows0", C.c_int),
                ("x1", C.c_void_p), ("C1", C.c_int), ("st1", C.c_void_p), ("rows1", C.c_int),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("groups", C.c_int), ("eps", C.c_float),
                ("Wp", C.c_void_p), ("bias", C.c_void_p), ("rowvec", C.c_void_p), ("residual", C.c_void_p), ("y", C.c_void_p),
                ("B", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Co", C.c_int),
                ("xa", C.c_void_p), ("Ca", C.c_int), ("splitk", C.c_int), ("partial", C.c_void_p), ("gn_out", C.c_void_p)]


_P, _I, _F, _SZ, _I64 = C.c_void_p, C.c_int, C.c_float, C.c_size_t, C.c_int64
# every symbol include/ia2p.h and include/ia2p_debug.h declare: name -> (restype, argtypes)
SIGNATURES = {
    "ia2p_create": (_I, [C.POINTER(UNetConfigC), C.POINTER(_P)]),
    "ia2p_destroy": (None, [_P]),
    "ia2p_last_error": (C.c_char_p, [_P]),
    "ia2p_device_is_gfx950": (_I, []),
    "ia2p_arena_bytes": (_SZ, [_P]),
    "ia2p_bind_arena": (_I, [_P, _P, _SZ]),
    "ia2p_load_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I, _P]),
    "ia2p_finalize_weights": (_I, [_P]),
    "ia2p_adopt_arena": (_I, [_P, _I]),
    "ia2p_adopt_arena_on": (_I, [_P, _I, _P]),
    "ia2p_arena_raw_bytes": (_SZ, [_P]),
    "ia2p_bcast_arena": (_I, [_P, _P, _I, _I, _P]),
    "ia2p_rccl_available": (_I, []),
    "ia2p_set_ip_adapter": (_I, [_P, _I, _I, _F]),
    "ia2p_workspace_bytes": (_SZ, [_P, _I, _I, _I, _I]),
    "ia2p_unet_forward": (_I, [_P, _P, _P, _F, _P, _I, _P, _P, _P, _I, _I, _I, _P, _SZ]),
    "ia2p_unet_forward_v": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _P, _SZ]),
    "ia2p_context_kv_bytes": (_SZ, [_P, _I, _I]),
    "ia2p_project_context": (_I, [_P, _P, _P, _I, _I, _P, _SZ, _P, _SZ]),
    "ia2p_unet_forward_kv": (_I, [_P, _P, _P, _F, _P, _I, _P, _P, _P, _I, _I, _I, _P, _SZ]),
    "ia2p_autotune": (_I, [_P, _P, _P, _F, _P, _I, _P, _P, _P, _I, _I, _I, _P, _SZ, _I, _P]),
    "ia2p_plan_export": (_SZ, [_P, _SZ]),
    "ia2p_plan_import": (_I, [C.c_char_p]),
    "ia2p_plan_clear": (None, []),
    "ia2p_plan_generation": (C.c_ulonglong, []),
    "ia2p_ddim_step": (_I, [_P, _P, _P, _P, _F, _F, _F, _P, _P, _I64]),
    "ia2p_ddim_step_v": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I64]),
    "ia2p_mask_blend": (_I, [_P, _P, _P, _P, _P, _F, _F, _P, _P, _I, _I, _I64]),
    "ia2p_groupnorm_silu": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "ia2p_layernorm": (_I, [_P, _P, _P, _P, _P, _I, _I, _F]),
    "ia2p_gemm": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    "ia2p_gemm_splitk": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "ia2p_ffn": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "ia2p_fold_layernorm": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I]),
    "ia2p_gemm_ex": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "ia2p_debug_set_gemm_splitk": (None, [_I]),
    "ia2p_conv3x3": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I]),
    "ia2p_conv3x3_splitk": (_I, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "ia2p_pack_conv3x3": (_I, [_P, _P, _P, _I, _I]),
    "ia2p_pack_conv_out": (_I, [_P, _P, _P, _I, _I]),
    "ia2p_conv3x3_cat": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I]),
    "ia2p_pack_geglu": (_I, [_P, _P, _P, _I, _I]),
    "ia2p_conv_in": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "ia2p_conv_out": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "ia2p_attention": (_I, [_P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _I, _I, _F, _P, _P, _I, _I, _F]),
    "ia2p_qkv_self_attention": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I]),
    "ia2p_qproj_attention": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _F, _P, _P, _I, _I, _F]),
    "ia2p_linear_small": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I]),
    "ia2p_ip_attn_map": (_I, [_P, _P, _I, _P, _I, _P, _I, _I, _I, _I]),
    "ia2p_debug_set_gemm_tile": (None, [_I]),
    "ia2p_debug_gemm_tile_info": (_I, [_I, _P]),
    "ia2p_debug_set_xattn_min_tiles": (None, [_I]),
    "ia2p_debug_set_attn_fold": (None, [_I]),
    "ia2p_debug_set_splitk_inkernel": (None, [C.c_longlong]),
    "ia2p_debug_invalidate_splitk_counters": (None, []),
    "ia2p_debug_fill_splitk_counters": (_I, [_P, _I]),
    "ia2p_debug_gemm_plan": (None, [_I, _I, _I, _I, _I, _P, _P]),
    "ia2p_clip_create": (_I, [C.POINTER(CLIPConfigC), C.POINTER(_P)]),
    "ia2p_clip_destroy": (None, [_P]),
    "ia2p_clip_last_error": (C.c_char_p, [_P]),
    "ia2p_clip_arena_bytes": (_SZ, [_P]),
    "ia2p_clip_bind_arena": (_I, [_P, _P, _SZ]),
    "ia2p_clip_load_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I, _P]),
    "ia2p_clip_finalize_weights": (_I, [_P]),
    "ia2p_clip_workspace_bytes": (_SZ, [_P, _I, _I]),
    "ia2p_clip_encode": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _SZ]),
    "ia2p_clip_encode_embeds": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _SZ]),
    "ia2p_prior_step": (_I, [_P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _F, _P, _I64]),
    "ia2p_vae_create": (_I, [C.POINTER(VAEConfigC), C.POINTER(_P)]),
    "ia2p_vae_destroy": (None, [_P]),
    "ia2p_vae_last_error": (C.c_char_p, [_P]),
    "ia2p_vae_arena_bytes": (_SZ, [_P]),
    "ia2p_vae_bind_arena": (_I, [_P, _P, _SZ]),
    "ia2p_vae_load_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(_I64), _I, _P]),
    "ia2p_vae_finalize_weights": (_I, [_P]),
    "ia2p_vae_workspace_bytes": (_SZ, [_P, _I, _I, _I, _I]),
    "ia2p_vae_decode": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _SZ]),
    "ia2p_vae_encode": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _SZ]),
    "ia2p_profile_enable": (_I, [_P, _I]),
    "ia2p_profile_classes": (_I, []),
    "ia2p_profile_read_region": (_I, [_P, _I, C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ia2p_profile_read_prefetch": (_I, [_P, _I, _P]),
    "ia2p_set_gn_fuse": (_I, [_P, _I]),
    "ia2p_debug_set_gn_plan": (None, [_I]),
    "ia2p_gn_colstats": (_I, [_P, _P, _I, _I, _I, _P]),
    "ia2p_gn_apply_stats": (_I, [_P, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _F, _I]),
    "ia2p_conv3x3_gn": (_I, [_P, C.POINTER(ConvGnC), _P]),
    "ia2p_gemm_gnstats": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _I, _P, _P]),
    "ia2p_profile_roles": (_I, []),
    "ia2p_profile_read_role_class": (_I, [_P, _I, _I, C.POINTER(_I64), C.POINTER(C.c_double)]),
    "ia2p_profile_read_role": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ia2p_profile_read": (_I, [_P, _I, C.c_char_p, _I, C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

_lib = None


class IA2PError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load the HIP library. Fails loudly when it has not been built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IA2PError(f"{LIB_PATH} is missing: build it with `python -m instructany2pix_amd.build` "
                            f"(the denoise path has no non-HIP fallback)")
        # torch bundles its own HIP runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7). It MUST be in the
        # process before this library is dlopen'ed so both bind to the SAME runtime instance; loaded the other way
        # round, a second runtime comes up and one of the two sees "No HIP GPUs" (observed on the MI355X boxes).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)           # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(status: int, ctx=None, vae=False, clip=False):
    """Map ia2p_status to the exception types the reference raises at the same conditions
    (ValueError from check_inputs/_get_add_time_ids, reference pnp_pipeline.py:49-66)."""
    if status == IA2P_OK:
        return
    msg = lib().ia2p_clip_last_error(ctx) if clip else lib().ia2p_vae_last_error(ctx) if vae else lib().ia2p_last_error(ctx)
    msg = msg.decode() if msg else ""
    text = f"ia2p {_STATUS_NAMES.get(status, status)}: {msg}"
    if status in (1, 2):
        raise ValueError(text)
    if status == 3:
        raise KeyError(text)
    if status == 5:
        raise MemoryError(text)
    raise IA2PError(text)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL). Tensors must be contiguous fp16 CUDA tensors."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "ia2p needs contiguous device tensors"
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_config(cfg) -> UNetConfigC:
    c = UNetConfigC()
    c.in_channels, c.out_channels = cfg.in_channels, cfg.out_channels
    c.n_blocks = len(cfg.block_out_channels)
    if c.n_blocks > MAX_BLOCKS:
        raise ValueError(f"at most {MAX_BLOCKS} resolution levels are supported")
    for i in range(c.n_blocks):
        c.block_out_channels[i] = cfg.block_out_channels[i]
        c.transformer_layers_per_block[i] = cfg.transformer_layers_per_block[i]
        c.num_heads[i] = cfg.attention_head_dim[i]
    c.layers_per_block = cfg.layers_per_block
    c.cross_attention_dim = cfg.cross_attention_dim
    c.norm_num_groups, c.norm_eps = cfg.norm_num_groups, cfg.norm_eps
    c.addition_time_embed_dim = cfg.addition_time_embed_dim
    c.projection_class_embeddings_input_dim = cfg.projection_class_embeddings_input_dim
    c.time_embed_dim, c.time_proj_dim = cfg.time_embed_dim, cfg.time_proj_dim
    c.mid_transformer_layers, c.num_time_ids = cfg.mid_block_transformer_layers, cfg.num_time_ids
    return c


def make_clip_config(cfg) -> CLIPConfigC:
    c = CLIPConfigC()
    c.vocab_size, c.hidden_size, c.num_layers, c.num_heads = cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads
    c.intermediate_size, c.max_positions, c.projection_dim = cfg.intermediate_size, cfg.max_position_embeddings, cfg.projection_dim
    c.hidden_act = {"gelu": 1, "quick_gelu": 2, "gelu_new": 3}[cfg.hidden_act]
    c.eos_token_id, c.layer_norm_eps = cfg.eos_token_id, cfg.layer_norm_eps
    return c


def make_vae_config(cfg) -> VAEConfigC:
    c = VAEConfigC()
    c.in_channels, c.out_channels, c.latent_channels = cfg.in_channels, cfg.out_channels, cfg.latent_channels
    c.n_blocks = len(cfg.block_out_channels)
    if c.n_blocks > MAX_BLOCKS:
        raise ValueError(f"at most {MAX_BLOCKS} resolution levels are supported")
    for i in range(c.n_blocks):
        c.block_out_channels[i] = cfg.block_out_channels[i]
    c.layers_per_block, c.norm_num_groups, c.norm_eps = cfg.layers_per_block, cfg.norm_num_groups, cfg.norm_eps
    c.stream_scale = float(getattr(cfg, "stream_scale", 1.0))
    return c
