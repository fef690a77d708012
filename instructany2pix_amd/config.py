"""UNet configuration for the denoise hot path.

Field names and the SDXL-base values follow the `unet/config.json` fields the reference reads
through `unet.config` (reference: instructany2pix/ddim/pnp_pipeline.py:44-47,
instructany2pix/diffusion/ip_adapter/ip_adapter.py:114,124-132) and SURVEY.md Appendix A.1.
"""
from dataclasses import dataclass, field, asdict
from typing import List, Optional, Tuple


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    sample_size: int = 128
    block_out_channels: Tuple[int, ...] = (320, 640, 1280)
    # per down block: number of transformer layers (0 = plain DownBlock2D / UpBlock2D)
    transformer_layers_per_block: Tuple[int, ...] = (0, 2, 10)
    # diffusers calls this attention_head_dim but for SDXL it holds HEAD COUNTS (head_dim is 64)
    attention_head_dim: Tuple[int, ...] = (5, 10, 20)
    layers_per_block: int = 2
    cross_attention_dim: int = 2048
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    addition_time_embed_dim: int = 256
    projection_class_embeddings_input_dim: int = 2816
    time_embed_dim: int = 1280           # block_out_channels[0] * 4
    time_proj_dim: int = 320             # block_out_channels[0]
    head_dim: int = 64
    # diffusers takes the mid block's depth from transformer_layers_per_block[-1]; here a 0 in that tuple means "plain
    # block", so a config whose last block is plain but whose mid block attends (the SDXL refiner) states it explicitly
    mid_transformer_layers: int = -1     # -1 = transformer_layers_per_block[-1]
    num_time_ids: int = 6                # 6 = (orig size, crop, target size); 5 = (orig size, crop, aesthetic score): refiner

    @property
    def mid_block_transformer_layers(self) -> int:
        return self.mid_transformer_layers if self.mid_transformer_layers >= 0 else self.transformer_layers_per_block[-1]

    def __getitem__(self, k):            # diffusers FrozenDict-style access
        return getattr(self, k)

    def to_dict(self):
        return asdict(self)

    @property
    def pooled_dim(self) -> int:
        return self.projection_class_embeddings_input_dim - self.num_time_ids * self.addition_time_embed_dim

    def validate(self):
        n = len(self.block_out_channels)
        assert len(self.transformer_layers_per_block) == n and len(self.attention_head_dim) == n
        for c, h, d in zip(self.block_out_channels, self.attention_head_dim, self.transformer_layers_per_block):
            assert c % self.norm_num_groups == 0
            assert c % 64 == 0, "channel counts must be multiples of 64 (MFMA K tile)"
            if d > 0:
                assert h * self.head_dim == c, "inner dim must equal channels"
        assert self.mid_block_transformer_layers >= 1 and self.attention_head_dim[-1] * self.head_dim == self.block_out_channels[-1]
        assert self.cross_attention_dim % 64 == 0
        assert self.time_embed_dim % 64 == 0 and self.projection_class_embeddings_input_dim % 64 == 0
        return self


def sdxl_base() -> UNetConfig:
    return UNetConfig().validate()


def sdxl_refiner() -> UNetConfig:
    """`stabilityai/stable-diffusion-xl-refiner-1.0` UNet, the second config of the same engine: the reference runs it as
    `self.piperf` after sampling (instructany2pix/pipeline.py:128-131,358-361). 4 levels, attention on
    the two middle ones and in the mid block (4 layers each), OpenCLIP-bigG context (1280), 5 micro-conditioning ids."""
    return UNetConfig(
        block_out_channels=(384, 768, 1536, 1536), transformer_layers_per_block=(0, 4, 4, 0), mid_transformer_layers=4,
        attention_head_dim=(6, 12, 24, 24), cross_attention_dim=1280, projection_class_embeddings_input_dim=2560, num_time_ids=5,
        time_embed_dim=1536, time_proj_dim=384,
    ).validate()


def tiny_refiner() -> UNetConfig:
    """the refiner's topology at toy width (4 levels, plain outer blocks, attending mid block, 5 time ids)"""
    return UNetConfig(
        block_out_channels=(64, 128, 256, 256), transformer_layers_per_block=(0, 1, 2, 0), mid_transformer_layers=2,
        attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128, sample_size=32, addition_time_embed_dim=64,
        projection_class_embeddings_input_dim=5 * 64 + 64, num_time_ids=5, time_embed_dim=256, time_proj_dim=64,
    ).validate()


def tiny(depths=(0, 1, 2)) -> UNetConfig:
    """Same topology at toy width: used by the parity tests so the CPU oracle finishes in seconds."""
    return UNetConfig(
        block_out_channels=(64, 128, 256), transformer_layers_per_block=tuple(depths),
        attention_head_dim=(1, 2, 4), cross_attention_dim=128, sample_size=32,
        addition_time_embed_dim=32, projection_class_embeddings_input_dim=6 * 32 + 64,
        time_embed_dim=256, time_proj_dim=64,
    ).validate()


@dataclass
class VAEConfig:
    """diffusers `AutoencoderKL` fields of the SDXL VAE (the object behind `pipe.vae`, reference
    instructany2pix/pipeline.py:109,134; ddim/sdxl_pipeline.py:859-871). First row of SURVEY.md §8f."""
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.13025
    force_upcast: bool = True        # the reference upcasts the VAE to fp32 (sdxl_pipeline.py:860-865); see vae.py
    # range extension that replaces the upcast on the HIP path: the residual stream is stored multiplied by this power of two (vae_engine.hip);
    # 2^-7 covers magnitudes up to 8.4e6 (the original SDXL VAE checkpoint passes the fp16 maximum of 65504); 1.0 = plain fp16 storage
    stream_scale: float = 2.0 ** -7

    def __getitem__(self, k):
        return getattr(self, k)

    def validate(self):
        for c in self.block_out_channels:
            assert c % 64 == 0 and c % self.norm_num_groups == 0
        assert self.in_channels * 9 <= 64 and self.latent_channels * 9 <= 64 and 2 * self.latent_channels <= 8 and self.out_channels <= 8
        return self


def sdxl_vae() -> VAEConfig:
    return VAEConfig().validate()


def tiny_vae() -> VAEConfig:
    return VAEConfig(block_out_channels=(64, 128, 128), layers_per_block=1).validate()


@dataclass
class CLIPTextConfig:
    """transformers `CLIPTextConfig` fields of SDXL's two text encoders (`pipe.text_encoder`: CLIP ViT-L/14 text tower;
    `pipe.text_encoder_2`: OpenCLIP ViT-bigG/14 text tower with projection) — the models behind `encode_prompt`
    (reference instructany2pix/ddim/sdxl_pipeline.py:202-395). Last row of SURVEY.md §8f."""
    vocab_size: int = 49408
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    projection_dim: int = 0              # 0 = CLIPTextModel (no text_projection); > 0 = CLIPTextModelWithProjection
    eos_token_id: int = 2                # SDXL checkpoints keep the legacy value: pooled row = position of the largest token id
    layer_norm_eps: float = 1e-5

    def __getitem__(self, k):
        return getattr(self, k)

    def validate(self):
        assert self.hidden_size == 64 * self.num_attention_heads and self.intermediate_size % 64 == 0
        assert self.hidden_act in ("gelu", "quick_gelu", "gelu_new") and self.max_position_embeddings <= 128 and self.projection_dim % 8 == 0
        return self


def sdxl_text_encoder() -> CLIPTextConfig:
    return CLIPTextConfig().validate()


def sdxl_text_encoder_2() -> CLIPTextConfig:
    return CLIPTextConfig(hidden_size=1280, num_hidden_layers=32, num_attention_heads=20, intermediate_size=5120, hidden_act="gelu",
                          projection_dim=1280).validate()


def laion_clip_h_text() -> CLIPTextConfig:
    """Text tower of laion/CLIP-ViT-H-14-laion2B-s32B-b79K: the conditioning encoder inside the embedding prior
    (`CLIPTextModelHiddenState`, reference instructany2pix/prior/model.py:28-34)."""
    return CLIPTextConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096, hidden_act="gelu").validate()


@dataclass
class GPT2Config:
    """transformers `GPT2Config` fields the prior's sequence model reads (`GPT2Model(GPT2Config.from_pretrained('gpt2-medium'))`,
    reference instructany2pix/prior/model.py:185 with prior/__init__.py `pretrained_name`)."""
    vocab_size: int = 50257
    n_positions: int = 1024
    n_embd: int = 1024
    n_layer: int = 24
    n_head: int = 16
    n_inner: Optional[int] = None          # None = 4 * n_embd
    activation_function: str = "gelu_new"
    layer_norm_epsilon: float = 1e-5

    def __getitem__(self, k):
        return getattr(self, k)

    @property
    def inner(self) -> int:
        return self.n_inner or 4 * self.n_embd

    def validate(self):
        assert self.n_embd == 64 * self.n_head and self.inner % 64 == 0 and self.activation_function in ("gelu_new", "gelu")
        return self


def gpt2_medium() -> GPT2Config:
    return GPT2Config().validate()


def tiny_gpt2() -> GPT2Config:
    return GPT2Config(vocab_size=100, n_positions=64, n_embd=128, n_layer=3, n_head=2).validate()


def tiny_clip(projection_dim: int = 0, hidden_act: str = "quick_gelu") -> CLIPTextConfig:
    return CLIPTextConfig(vocab_size=1000, hidden_size=128, num_hidden_layers=3, num_attention_heads=2, intermediate_size=256,
                          hidden_act=hidden_act, projection_dim=projection_dim).validate()
