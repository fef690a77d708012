"""Parameter inventory (diffusers key layout) and synthetic weights for the denoise hot path.

The UNet the reference drives is diffusers' `UNet2DConditionModel` loaded from the SDXL-base checkpoint
(reference: instructany2pix/pipeline.py:101); its state-dict keys are the interchange format here
(SURVEY.md Appendix A.7), so a real `unet/diffusion_pytorch_model.safetensors` drops in unchanged.
The IP-Adapter file layout `{"image_proj": ..., "ip_adapter": {"<idx>.to_k_ip.weight": ...}}` follows
reference instructany2pix/diffusion/ip_adapter/ip_adapter.py:155-169.

There is no network on the build/bench machines, so benchmarks and tests use seeded synthetic weights
of exactly these names and shapes (recipe: SURVEY.md §8d).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Iterator, List, Tuple

import torch

from .config import UNetConfig

Spec = Tuple[str, Tuple[int, ...], str]  # (key, shape, kind)


def _resnet(prefix: str, cin: int, cout: int, temb: int) -> List[Spec]:
    s: List[Spec] = [
        (f"{prefix}.norm1.weight", (cin,), "gamma"), (f"{prefix}.norm1.bias", (cin,), "beta"),
        (f"{prefix}.conv1.weight", (cout, cin, 3, 3), "w"), (f"{prefix}.conv1.bias", (cout,), "b"),
        (f"{prefix}.time_emb_proj.weight", (cout, temb), "w"), (f"{prefix}.time_emb_proj.bias", (cout,), "b"),
        (f"{prefix}.norm2.weight", (cout,), "gamma"), (f"{prefix}.norm2.bias", (cout,), "beta"),
        (f"{prefix}.conv2.weight", (cout, cout, 3, 3), "w_res"), (f"{prefix}.conv2.bias", (cout,), "b"),
    ]
    if cin != cout:
        s += [(f"{prefix}.conv_shortcut.weight", (cout, cin, 1, 1), "w"), (f"{prefix}.conv_shortcut.bias", (cout,), "b")]
    return s


def _transformer(prefix: str, c: int, depth: int, ctx: int) -> List[Spec]:
    s: List[Spec] = [
        (f"{prefix}.norm.weight", (c,), "gamma"), (f"{prefix}.norm.bias", (c,), "beta"),
        (f"{prefix}.proj_in.weight", (c, c), "w"), (f"{prefix}.proj_in.bias", (c,), "b"),
    ]
    for k in range(depth):
        p = f"{prefix}.transformer_blocks.{k}"
        s += [
            (f"{p}.norm1.weight", (c,), "gamma"), (f"{p}.norm1.bias", (c,), "beta"),
            (f"{p}.attn1.to_q.weight", (c, c), "w"), (f"{p}.attn1.to_k.weight", (c, c), "w"),
            (f"{p}.attn1.to_v.weight", (c, c), "w"),
            (f"{p}.attn1.to_out.0.weight", (c, c), "w_res"), (f"{p}.attn1.to_out.0.bias", (c,), "b"),
            (f"{p}.norm2.weight", (c,), "gamma"), (f"{p}.norm2.bias", (c,), "beta"),
            (f"{p}.attn2.to_q.weight", (c, c), "w"), (f"{p}.attn2.to_k.weight", (c, ctx), "w"),
            (f"{p}.attn2.to_v.weight", (c, ctx), "w"),
            (f"{p}.attn2.to_out.0.weight", (c, c), "w_res"), (f"{p}.attn2.to_out.0.bias", (c,), "b"),
            (f"{p}.norm3.weight", (c,), "gamma"), (f"{p}.norm3.bias", (c,), "beta"),
            (f"{p}.ff.net.0.proj.weight", (8 * c, c), "w"), (f"{p}.ff.net.0.proj.bias", (8 * c,), "b"),
            (f"{p}.ff.net.2.weight", (c, 4 * c), "w_res"), (f"{p}.ff.net.2.bias", (c,), "b"),
        ]
    s += [(f"{prefix}.proj_out.weight", (c, c), "w_res"), (f"{prefix}.proj_out.bias", (c,), "b")]
    return s


def unet_param_specs(cfg: UNetConfig) -> List[Spec]:
    """Every UNet parameter as (diffusers key, shape, init kind), in module-registration order."""
    ch = list(cfg.block_out_channels)
    depth = list(cfg.transformer_layers_per_block)
    temb = cfg.time_embed_dim
    ctx = cfg.cross_attention_dim
    n = len(ch)
    s: List[Spec] = [
        ("conv_in.weight", (ch[0], cfg.in_channels, 3, 3), "w"), ("conv_in.bias", (ch[0],), "b"),
        ("time_embedding.linear_1.weight", (temb, cfg.time_proj_dim), "w"), ("time_embedding.linear_1.bias", (temb,), "b"),
        ("time_embedding.linear_2.weight", (temb, temb), "w"), ("time_embedding.linear_2.bias", (temb,), "b"),
        ("add_embedding.linear_1.weight", (temb, cfg.projection_class_embeddings_input_dim), "w"),
        ("add_embedding.linear_1.bias", (temb,), "b"),
        ("add_embedding.linear_2.weight", (temb, temb), "w"), ("add_embedding.linear_2.bias", (temb,), "b"),
    ]
    # down path
    cprev = ch[0]
    skip_ch = [ch[0]]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            cin = cprev if j == 0 else ch[i]
            s += _resnet(f"down_blocks.{i}.resnets.{j}", cin, ch[i], temb)
            if depth[i] > 0:
                s += _transformer(f"down_blocks.{i}.attentions.{j}", ch[i], depth[i], ctx)
            skip_ch.append(ch[i])
        cprev = ch[i]
        if i != n - 1:
            s += [(f"down_blocks.{i}.downsamplers.0.conv.weight", (ch[i], ch[i], 3, 3), "w"),
                  (f"down_blocks.{i}.downsamplers.0.conv.bias", (ch[i],), "b")]
            skip_ch.append(ch[i])
    # mid
    cm = ch[-1]
    s += _resnet("mid_block.resnets.0", cm, cm, temb)
    s += _transformer("mid_block.attentions.0", cm, cfg.mid_block_transformer_layers, ctx)
    s += _resnet("mid_block.resnets.1", cm, cm, temb)
    # up path
    rch = list(reversed(ch))
    rdepth = list(reversed(depth))
    cprev = cm
    for i in range(n):
        cout = rch[i]
        for j in range(cfg.layers_per_block + 1):
            cskip = skip_ch.pop()
            cin = (cprev if j == 0 else cout) + cskip
            s += _resnet(f"up_blocks.{i}.resnets.{j}", cin, cout, temb)
            if rdepth[i] > 0:
                s += _transformer(f"up_blocks.{i}.attentions.{j}", cout, rdepth[i], ctx)
        cprev = cout
        if i != n - 1:
            s += [(f"up_blocks.{i}.upsamplers.0.conv.weight", (cout, cout, 3, 3), "w"),
                  (f"up_blocks.{i}.upsamplers.0.conv.bias", (cout,), "b")]
    s += [("conv_norm_out.weight", (ch[0],), "gamma"), ("conv_norm_out.bias", (ch[0],), "beta"),
          ("conv_out.weight", (cfg.out_channels, ch[0], 3, 3), "w_out"), ("conv_out.bias", (cfg.out_channels,), "b")]
    return s


def attn_processor_names(cfg: UNetConfig) -> List[str]:
    """Keys of `unet.attn_processors` in diffusers' module-registration order: down_blocks, up_blocks,
    mid_block. The position in this list is the `<idx>` of the IP-Adapter
    checkpoint keys (reference ip_adapter.py:168-169 loads them through a ModuleList of the values)."""
    depth = list(cfg.transformer_layers_per_block)
    n = len(depth)
    names: List[str] = []

    def add(prefix, d):
        for k in range(d):
            names.append(f"{prefix}.transformer_blocks.{k}.attn1.processor")
            names.append(f"{prefix}.transformer_blocks.{k}.attn2.processor")

    for i in range(n):
        for j in range(cfg.layers_per_block):
            add(f"down_blocks.{i}.attentions.{j}", depth[i])
    rdepth = list(reversed(depth))
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            add(f"up_blocks.{i}.attentions.{j}", rdepth[i])
    add("mid_block.attentions.0", cfg.mid_block_transformer_layers)
    return names


def hidden_size_of(cfg: UNetConfig, name: str) -> int:
    """Same rule the reference uses to size IP processors (ip_adapter.py:125-132)."""
    if name.startswith("mid_block"):
        return cfg.block_out_channels[-1]
    if name.startswith("up_blocks"):
        return list(reversed(cfg.block_out_channels))[int(name[len("up_blocks.")])]
    return cfg.block_out_channels[int(name[len("down_blocks.")])]


def ip_adapter_specs(cfg: UNetConfig, clip_embeddings_dim: int = 1024, num_tokens: int = 4) -> Dict[str, List[Spec]]:
    ctx = cfg.cross_attention_dim
    image_proj: List[Spec] = [
        ("proj.weight", (num_tokens * ctx, clip_embeddings_dim), "w"), ("proj.bias", (num_tokens * ctx,), "b"),
        ("norm.weight", (ctx,), "gamma"), ("norm.bias", (ctx,), "beta"),
        ("raw_embed", (2, ctx), "b"),
    ]
    ip: List[Spec] = []
    for idx, name in enumerate(attn_processor_names(cfg)):
        if name.endswith("attn2.processor"):
            c = hidden_size_of(cfg, name)
            ip += [(f"{idx}.to_k_ip.weight", (c, ctx), "w"), (f"{idx}.to_v_ip.weight", (c, ctx), "w")]
    return {"image_proj": image_proj, "ip_adapter": ip}


def _fan_in(shape: Tuple[int, ...]) -> int:
    f = 1
    for d in shape[1:]:
        f *= d
    return max(f, 1)


HEAVY_SCALE = 60.0


def heavy_tail_rows(rows: int) -> "torch.Tensor":
    """Output channels of a residual branch's last layer that the "heavy" recipe scales x HEAVY_SCALE: 1 % of the rows, the SAME channels in every layer of a
    given width (as in trained transformers, where a few fixed channels of the residual stream carry the outliers), so the branches add up in them."""
    idx = [r for r in range(rows) if ((r * 2654435761) & 0xFFFFFFFF) % 100 == 0]
    return torch.tensor(idx or [rows // 2], dtype=torch.long)


def _make(key: str, shape, kind: str, seed: int, device, dtype, recipe: str = "tame") -> torch.Tensor:
    """recipe "tame": N(0, 1 / fan_in) weights, residual branches x 0.3 (activations stay O(1) everywhere).
    recipe "heavy": the same draw, but 1 % of the output channels of every residual branch's last layer (attn to_out.0, ff.net.2, resnet conv2,
    proj_out -- kind "w_res") are scaled x HEAVY_SCALE = 60 (the same channels in every layer), so the residual stream carries outlier channels of a few hundred like a trained SDXL UNet's does:
    the stand-in for the checkpoint nobody has here (tests/test_fullsize_heavy_gpu.py)."""
    g = torch.Generator(device=device)
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    r = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
    if kind == "w":
        r.mul_(_fan_in(shape) ** -0.5)
    elif kind == "w_res":        # output layer of a residual branch
        r.mul_(0.3 * _fan_in(shape) ** -0.5)
    elif kind == "w_out":        # conv_out: input is ~N(0,1) after GroupNorm+SiLU (rms ~0.6)
        r.mul_(1.6 * _fan_in(shape) ** -0.5)
    elif kind == "wt":           # transformers Conv1D layout [in, out]
        r.mul_(shape[0] ** -0.5)
    elif kind == "wt_res":
        r.mul_(0.3 * shape[0] ** -0.5)
    elif kind == "emb":          # embedding tables: unit-scale rows would swamp nothing; CLIP uses ~0.02, use 0.5 to keep LayerNorm honest
        r.mul_(0.5)
    elif kind == "b":
        r.mul_(0.02)
    elif kind == "gamma":
        r.mul_(0.1).add_(1.0)
    elif kind == "beta":
        r.mul_(0.05)
    else:
        raise ValueError(kind)
    if recipe == "heavy" and kind == "w_res":
        r[heavy_tail_rows(shape[0]).to(r.device)] *= HEAVY_SCALE
    elif recipe not in ("tame", "heavy"):
        raise ValueError(recipe)
    return r.to(dtype)


def synthetic_state_dict(specs: List[Spec], seed: int = 7, device="cpu", dtype=torch.float16, recipe: str = "tame") -> "OrderedDict[str, torch.Tensor]":
    """Seeded synthetic parameters. Each tensor depends only on (key, seed, recipe), so any subset can be
    regenerated independently and CPU/GPU processes agree when `device` is the same kind."""
    return OrderedDict((k, _make(k, shp, kind, seed, device, dtype, recipe)) for k, shp, kind in specs)


def iter_synthetic(specs: List[Spec], seed: int = 7, device="cpu", dtype=torch.float16, recipe: str = "tame") -> Iterator[Tuple[str, torch.Tensor]]:
    for k, shp, kind in specs:
        yield k, _make(k, shp, kind, seed, device, dtype, recipe)


def iter_safetensors(path: str, specs: "List[Spec] | None" = None, prefix: str = "") -> Iterator[Tuple[str, torch.Tensor]]:
    """Stream a checkpoint tensor by tensor: `path` is a `.safetensors` file, a directory holding `diffusion_pytorch_model*.safetensors`
    (diffusers' layout of `unet/`, sharded or not, `.fp16` variant included), or a `*.safetensors.index.json`. Yields (key, tensor) with
    `prefix` stripped, for `load_state_dict(...)` of the HIP modules: 5 GB of UNet weights are never resident twice on the host.
    With `specs` (e.g. `unet_param_specs(cfg)`) the key set and every shape are checked against the architecture first -- the loader of
    SURVEY.md §7.1's `IA2P_UNET_WEIGHTS`; the reference gets the same tensors through `from_pretrained` (pipeline.py:101,128)."""
    import glob
    import json
    import os
    from safetensors import safe_open
    if os.path.isdir(path):
        idx = sorted(glob.glob(os.path.join(path, "*.safetensors.index.json")))
        files = [idx[0]] if idx else sorted(glob.glob(os.path.join(path, "diffusion_pytorch_model*.safetensors"))) or sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            raise FileNotFoundError(f"no .safetensors checkpoint under {path}")
        if len(files) > 1 and not idx:
            plain = [f for f in files if os.path.basename(f) == "diffusion_pytorch_model.safetensors"]
            if not plain:       # several variants (fp16 / non-ema / ...) and nothing says which one is meant: do not pick silently
                raise ValueError(f"{path} holds several .safetensors variants ({[os.path.basename(f) for f in files]}) and no "
                                 f"diffusion_pytorch_model.safetensors: pass the file to load")
            files = plain
        path = files[0]
    if path.endswith(".index.json"):
        shards = sorted(set(json.load(open(path))["weight_map"].values()))
        files = [os.path.join(os.path.dirname(path), f) for f in shards]
    else:
        files = [path]
    import contextlib
    with contextlib.ExitStack() as stack:          # the mmaps / file handles close when the generator is exhausted or closed
        handles = [stack.enter_context(safe_open(f, framework="pt", device="cpu")) for f in files]
        yield from _iter_open_safetensors(handles, prefix, specs)


def _iter_open_safetensors(handles, prefix, specs):
    where = {}
    for h in handles:
        for k in h.keys():
            if k.startswith(prefix):
                where[k[len(prefix):]] = (h, k)
    if specs is not None:
        want = {k: tuple(shp) for k, shp, _ in specs}
        missing, extra = sorted(set(want) - set(where)), sorted(set(where) - set(want))
        if missing or extra:
            raise KeyError(f"checkpoint does not match the architecture: {len(missing)} missing (e.g. {missing[:3]}), {len(extra)} unexpected (e.g. {extra[:3]})")
        for k, shp in want.items():
            h, full = where[k]
            got = tuple(h.get_slice(full).get_shape())
            if got != shp:
                raise ValueError(f"checkpoint tensor '{k}' has shape {got}, the architecture expects {shp}")
        order = [k for k, _, _ in specs]
    else:
        order = list(where)
    for k in order:
        h, full = where[k]
        yield k, h.get_tensor(full)


def load_unet_safetensors(unet, path: str, strict: bool = True):
    """`unet.load_state_dict` from a diffusers UNet checkpoint on disk (see iter_safetensors); returns the UNet."""
    unet.load_state_dict(iter_safetensors(path, unet_param_specs(unet.config) if strict else None), strict=strict)
    return unet


def param_count(specs: List[Spec]) -> int:
    t = 0
    for _, shp, _ in specs:
        n = 1
        for d in shp:
            n *= d
        t += n
    return t


# ---- VAE (diffusers AutoencoderKL key layout) ---------------------------------------------------------------------------
def _vae_resnet(prefix: str, cin: int, cout: int) -> List[Spec]:
    s: List[Spec] = [
        (f"{prefix}.norm1.weight", (cin,), "gamma"), (f"{prefix}.norm1.bias", (cin,), "beta"),
        (f"{prefix}.conv1.weight", (cout, cin, 3, 3), "w"), (f"{prefix}.conv1.bias", (cout,), "b"),
        (f"{prefix}.norm2.weight", (cout,), "gamma"), (f"{prefix}.norm2.bias", (cout,), "beta"),
        (f"{prefix}.conv2.weight", (cout, cout, 3, 3), "w_res"), (f"{prefix}.conv2.bias", (cout,), "b"),
    ]
    if cin != cout:
        s += [(f"{prefix}.conv_shortcut.weight", (cout, cin, 1, 1), "w"), (f"{prefix}.conv_shortcut.bias", (cout,), "b")]
    return s


def _vae_mid(prefix: str, c: int) -> List[Spec]:
    a = f"{prefix}.attentions.0"
    return (_vae_resnet(f"{prefix}.resnets.0", c, c) + [
        (f"{a}.group_norm.weight", (c,), "gamma"), (f"{a}.group_norm.bias", (c,), "beta"),
        (f"{a}.to_q.weight", (c, c), "w"), (f"{a}.to_q.bias", (c,), "b"),
        (f"{a}.to_k.weight", (c, c), "w"), (f"{a}.to_k.bias", (c,), "b"),
        (f"{a}.to_v.weight", (c, c), "w"), (f"{a}.to_v.bias", (c,), "b"),
        (f"{a}.to_out.0.weight", (c, c), "w_res"), (f"{a}.to_out.0.bias", (c,), "b"),
    ] + _vae_resnet(f"{prefix}.resnets.1", c, c))


def vae_param_specs(cfg) -> List[Spec]:
    ch = list(cfg.block_out_channels)
    n = len(ch)
    z = cfg.latent_channels
    s: List[Spec] = [("encoder.conv_in.weight", (ch[0], cfg.in_channels, 3, 3), "w"), ("encoder.conv_in.bias", (ch[0],), "b")]
    cprev = ch[0]
    for i in range(n):
        for j in range(cfg.layers_per_block):
            s += _vae_resnet(f"encoder.down_blocks.{i}.resnets.{j}", cprev if j == 0 else ch[i], ch[i])
        cprev = ch[i]
        if i != n - 1:
            s += [(f"encoder.down_blocks.{i}.downsamplers.0.conv.weight", (ch[i], ch[i], 3, 3), "w"),
                  (f"encoder.down_blocks.{i}.downsamplers.0.conv.bias", (ch[i],), "b")]
    s += _vae_mid("encoder.mid_block", ch[-1])
    s += [("encoder.conv_norm_out.weight", (ch[-1],), "gamma"), ("encoder.conv_norm_out.bias", (ch[-1],), "beta"),
          ("encoder.conv_out.weight", (2 * z, ch[-1], 3, 3), "w_out"), ("encoder.conv_out.bias", (2 * z,), "b"),
          ("quant_conv.weight", (2 * z, 2 * z, 1, 1), "w"), ("quant_conv.bias", (2 * z,), "b"),
          ("post_quant_conv.weight", (z, z, 1, 1), "w"), ("post_quant_conv.bias", (z,), "b"),
          ("decoder.conv_in.weight", (ch[-1], z, 3, 3), "w"), ("decoder.conv_in.bias", (ch[-1],), "b")]
    s += _vae_mid("decoder.mid_block", ch[-1])
    rch = ch[::-1]
    cprev = ch[-1]
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            s += _vae_resnet(f"decoder.up_blocks.{i}.resnets.{j}", cprev if j == 0 else rch[i], rch[i])
        cprev = rch[i]
        if i != n - 1:
            s += [(f"decoder.up_blocks.{i}.upsamplers.0.conv.weight", (rch[i], rch[i], 3, 3), "w"),
                  (f"decoder.up_blocks.{i}.upsamplers.0.conv.bias", (rch[i],), "b")]
    s += [("decoder.conv_norm_out.weight", (ch[0],), "gamma"), ("decoder.conv_norm_out.bias", (ch[0],), "beta"),
          ("decoder.conv_out.weight", (cfg.out_channels, ch[0], 3, 3), "w_out"), ("decoder.conv_out.bias", (cfg.out_channels,), "b")]
    return s


def clip_param_specs(cfg) -> List[Tuple[str, Tuple[int, ...], str]]:
    """transformers `CLIPTextModel(WithProjection).state_dict()` keys and shapes, in module-registration order."""
    H, I = cfg.hidden_size, cfg.intermediate_size
    s = [("text_model.embeddings.token_embedding.weight", (cfg.vocab_size, H), "emb"),
         ("text_model.embeddings.position_embedding.weight", (cfg.max_position_embeddings, H), "emb")]
    for i in range(cfg.num_hidden_layers):
        p = f"text_model.encoder.layers.{i}."
        for n in ("k_proj", "v_proj", "q_proj"):
            s += [(p + f"self_attn.{n}.weight", (H, H), "w"), (p + f"self_attn.{n}.bias", (H,), "b")]
        s += [(p + "self_attn.out_proj.weight", (H, H), "w_res"), (p + "self_attn.out_proj.bias", (H,), "b"),
              (p + "layer_norm1.weight", (H,), "gamma"), (p + "layer_norm1.bias", (H,), "beta"),
              (p + "mlp.fc1.weight", (I, H), "w"), (p + "mlp.fc1.bias", (I,), "b"),
              (p + "mlp.fc2.weight", (H, I), "w_res"), (p + "mlp.fc2.bias", (H,), "b"),
              (p + "layer_norm2.weight", (H,), "gamma"), (p + "layer_norm2.bias", (H,), "beta")]
    s += [("text_model.final_layer_norm.weight", (H,), "gamma"), ("text_model.final_layer_norm.bias", (H,), "beta")]
    if cfg.projection_dim:
        s += [("text_projection.weight", (cfg.projection_dim, H), "w")]
    return s


# ---- embedding prior (reference instructany2pix/prior/model.py) --------------------------------------------------------------
def gpt2_param_specs(cfg, prefix: str = "") -> List[Spec]:
    """transformers `GPT2Model.state_dict()` keys and shapes (Conv1D weights are [in, out])."""
    E, I = cfg.n_embd, cfg.inner
    s: List[Spec] = [(prefix + "wte.weight", (cfg.vocab_size, E), "emb"), (prefix + "wpe.weight", (cfg.n_positions, E), "emb")]
    for i in range(cfg.n_layer):
        p = f"{prefix}h.{i}."
        s += [(p + "ln_1.weight", (E,), "gamma"), (p + "ln_1.bias", (E,), "beta"),
              (p + "attn.c_attn.weight", (E, 3 * E), "wt"), (p + "attn.c_attn.bias", (3 * E,), "b"),
              (p + "attn.c_proj.weight", (E, E), "wt_res"), (p + "attn.c_proj.bias", (E,), "b"),
              (p + "ln_2.weight", (E,), "gamma"), (p + "ln_2.bias", (E,), "beta"),
              (p + "mlp.c_fc.weight", (E, I), "wt"), (p + "mlp.c_fc.bias", (I,), "b"),
              (p + "mlp.c_proj.weight", (I, E), "wt_res"), (p + "mlp.c_proj.bias", (E,), "b")]
    s += [(prefix + "ln_f.weight", (E,), "gamma"), (prefix + "ln_f.bias", (E,), "beta")]
    return s


def prior_param_specs(gpt_cfg, clip_cfg, sequence_input_embed_dim=(0, 1024, 1024, 512, 0, 0, 0), output_dim=None) -> List[Spec]:
    """`InstructAny2PixPrior.state_dict()` (prior/model.py:159-185): sos/eos tables, one Linear per sequence slot with a non-zero
    input width, the modality table, the CLIP text tower of the conditioning stage and the GPT-2 sequence model."""
    E = gpt_cfg.n_embd
    s: List[Spec] = [("start_of_sequence_tokens.weight", (32, E), "emb"), ("end_of_sequence_tokens.weight", (32, E), "emb")]
    for i, d in enumerate(sequence_input_embed_dim):
        if d:
            s += [(f"input_sequence_embed_linear.{i}.weight", (E, d), "w"), (f"input_sequence_embed_linear.{i}.bias", (E,), "b")]
    s += [("modality_embedding.weight", (10, E), "emb")]
    if output_dim is not None and output_dim != E:
        s += [("output_proj.weight", (output_dim, E), "w"), ("output_proj.bias", (output_dim,), "b")]
    s += [("cond_stage_models.0.model." + k, shp, kind) for k, shp, kind in clip_param_specs(clip_cfg)]
    s += gpt2_param_specs(gpt_cfg, "model.")
    return s
