"""The embedding prior in front of the denoise loop, on the HIP kernels (second half).

  MODALITY, prior_config                 <- instructany2pix/prior/model.py:6-10, prior/__init__.py (the constructor arguments)
  InstructAny2PixPrior(**prior_config)   <- prior/model.py:109-206; held as `InstructAny2PixPipeline.model` (pipeline.py:97-98,120-122)
      .generate_diffusion(src_type, tgt_type, src, ...) -> (latents, cond_dict)      prior/model.py:528-658; the one call on the live
                                            path is pipeline.py:313-317 (VIDEO -> IMAGE, no_diffusion=True, guidance 10, score 6.5)
      .get_input_sequence_and_mask / .add_sos_eos_tokens / .get_eps / .get_input     :299-381, :272-287, :208-240, :717-741
  CLIPTextModelHiddenState               <- prior/model.py:20-105 (conditioning stage: CLIP ViT-H text tower, last_hidden_state + mask)
  HipGPT2Model                           <- transformers `GPT2Model` as the prior drives it: `model(inputs_embeds=, attention_mask=)
                                            ["last_hidden_state"]` (:611-613)

What runs where: the GPT-2 stack and the CLIP tower go through `ia2p_clip_encode_embeds` / `ia2p_clip_encode` (LayerNorm-folded GEMMs,
causal attention, gelu_new / gelu epilogues), the three slot projections through `ia2p_linear_small`, the sampler update (get_eps,
guidance, DDPM posterior) through `ia2p_prior_step` in fp32. Host side: table lookups, sequence concatenation, the sinusoidal score
embedding (512 numbers) and the RNG draws, which stay on the CPU generator exactly as in the reference (it runs this stage with
device='cpu'), so the same seed consumes the same random numbers. The reference computes this stage in fp32 on the CPU; here the
transformer stacks compute in fp16 with fp32 accumulation (tolerances in tests/test_prior_gpu.py).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Dict, List, Optional

import torch

from . import _ffi
from .clip import HipCLIPTextModel
from .config import CLIPTextConfig, GPT2Config, gpt2_medium, laion_clip_h_text
from .scheduler import DDPMScheduler, prior_update


class MODALITY:
    IMAGE = 0
    AUDIO = 1
    TEXT = 2
    VIDEO = 3


# Constructor arguments of the released prior. The reference's list literal has no comma between its last two entries, so they are ONE key
# that no conditioning dict ever holds: `noise_level` and `tgt_type` never enter the sequence. Kept, because the trained weights assume it.
prior_config = {
    "sequence_gen_length": 1,
    "diffusion": True,
    "sequence_input_key": ["src_type", "imagebind", "crossattn_clip", "score", "noisy_inputs", "noise_leveltgt_type"],
    "sequence_input_embed_dim": [0, 1024, 1024, 512, 0, 0, 0],
    "pretrained_name": "gpt2-medium",
    "embed_dim": 1024,
    "output_dim": 1024,
    "cond_stage_config": {"crossattn_clip": {"cond_stage_key": "text", "conditioning_key": "crossattn"}},
}


def get_timestep_embedding(timesteps: torch.Tensor, embedding_dim: int, flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0):
    """diffusers' sinusoid in the one form the prior asks for (cos half first)."""
    if not flip_sin_to_cos or downscale_freq_shift != 0:
        raise NotImplementedError("the prior only uses flip_sin_to_cos=True, downscale_freq_shift=0")
    half = embedding_dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class HipGPT2Model:
    """GPT-2 stack on the text-encoder executor: same pre-LayerNorm causal layer, fused `c_attn` split into the q/k/v slots, Conv1D
    weights ([in, out]) transposed at load, `wpe` as the position table (first 128 rows: sequences here are 11..~90 tokens), no token
    table (the prior feeds `inputs_embeds`)."""

    def __init__(self, config: GPT2Config, device="cuda:0"):
        self.config = config.validate()
        self.max_positions = min(config.n_positions, 128)
        self._core = HipCLIPTextModel(CLIPTextConfig(vocab_size=0, hidden_size=config.n_embd, num_hidden_layers=config.n_layer,
                                                     num_attention_heads=config.n_head, intermediate_size=config.inner,
                                                     max_position_embeddings=self.max_positions, hidden_act=config.activation_function,
                                                     layer_norm_eps=config.layer_norm_epsilon), device)
        self.device = self._core.device
        self.dtype = torch.float16

    def to(self, *a, **kw):
        return self

    def eval(self):
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        E = self.config.n_embd
        out: Dict[str, torch.Tensor] = {}
        for k, v in (state_dict.items() if hasattr(state_dict, "items") else state_dict):
            if k == "wte.weight" or k.endswith((".attn.bias", ".attn.masked_bias")):      # token table unused; causal-mask buffers of old checkpoints
                continue
            if k == "wpe.weight":
                out["text_model.embeddings.position_embedding.weight"] = v[:self.max_positions]
            elif k.startswith("ln_f."):
                out["text_model.final_layer_norm." + k[5:]] = v
            elif k.startswith("h."):
                _, i, rest = k.split(".", 2)
                p = f"text_model.encoder.layers.{i}."
                kind = rest.rsplit(".", 1)[1]
                if rest.startswith("ln_1."):
                    out[p + "layer_norm1." + kind] = v
                elif rest.startswith("ln_2."):
                    out[p + "layer_norm2." + kind] = v
                elif rest.startswith("attn.c_attn."):
                    for j, n in enumerate(("q_proj", "k_proj", "v_proj")):
                        out[p + f"self_attn.{n}.{kind}"] = v[:, j * E:(j + 1) * E].t() if kind == "weight" else v[j * E:(j + 1) * E]
                elif rest.startswith("attn.c_proj."):
                    out[p + "self_attn.out_proj." + kind] = v.t() if kind == "weight" else v
                elif rest.startswith("mlp.c_fc."):
                    out[p + "mlp.fc1." + kind] = v.t() if kind == "weight" else v
                elif rest.startswith("mlp.c_proj."):
                    out[p + "mlp.fc2." + kind] = v.t() if kind == "weight" else v
                else:
                    raise KeyError(f"unexpected GPT-2 parameter {k}")
            else:
                raise KeyError(f"unexpected GPT-2 parameter {k}")
        self._core.load_state_dict(out, strict=strict)

    @torch.no_grad()
    def __call__(self, inputs_embeds: torch.Tensor = None, attention_mask: Optional[torch.Tensor] = None, **unused):
        if inputs_embeds is None or inputs_embeds.ndim != 3:
            raise ValueError("inputs_embeds must be [batch, tokens, n_embd]")
        if attention_mask is not None and not bool((attention_mask != 0).all()):
            raise NotImplementedError("padded sequences (zeros in attention_mask) are not on the reference's live path")
        B, T, E = inputs_embeds.shape
        if E != self.config.n_embd or T > self.max_positions:
            raise ValueError(f"inputs_embeds {tuple(inputs_embeds.shape)}: width must be {self.config.n_embd}, at most {self.max_positions} tokens")
        core = self._core
        x = inputs_embeds.to(device=self.device, dtype=torch.float16).contiguous()
        n = core._lib.ia2p_clip_workspace_bytes(core._h, B, T)
        if n == 0:
            _ffi.check(2, core._h, clip=True)
        if core._ws is None or core._ws.numel() < n:
            core._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        last = torch.empty(B, T, E, dtype=torch.float16, device=self.device)
        _ffi.check(core._lib.ia2p_clip_encode_embeds(core._h, _ffi.current_stream(), _ffi.ptr(x), B, T, None, _ffi.ptr(last), _ffi.ptr(core._ws),
                                                     core._ws.numel()), core._h, clip=True)
        return {"last_hidden_state": last}


class CLIPTextModelHiddenState:
    """Conditioning stage: tokenize (`max_length=77, padding=True, truncation=True`), run the text tower, return
    `[last_hidden_state, attention_mask.float()]`. The tokenizer is an injected callable (checkpoint vocabulary)."""

    def __init__(self, tokenizer: Callable, model: HipCLIPTextModel):
        self.tokenizer, self.model = tokenizer, model
        self.empty_hidden_state_cfg = None
        self.device = model.device

    def get_unconditional_condition(self, batchsize):
        if self.empty_hidden_state_cfg is None:
            self.empty_hidden_state_cfg, _ = self([""])
        hidden = torch.cat([self.empty_hidden_state_cfg] * batchsize).float()
        return [hidden, torch.ones((batchsize, hidden.size(1)), device=hidden.device).float()]

    def encode_text(self, prompt):
        batch = self.tokenizer(prompt, max_length=77, padding=True, truncation=True, return_tensors="pt")
        mask = batch.attention_mask
        if not bool((mask != 0).all()):
            raise NotImplementedError("prompts of different token counts in one batch need key masking; the live path passes [''] only")
        out = self.model(batch.input_ids, want_last_hidden=True, want_pooled=False)
        return [out.last_hidden_state, mask.to(self.device).float()]

    __call__ = forward = encode_text


def _linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """y = x W^T + b over the last axis, fp16, via ia2p_linear_small (<= 16 rows per launch)."""
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1]).to(torch.float16).contiguous()
    M, K = x2.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float16, device=x2.device)
    L = _ffi.lib()
    for r0 in range(0, M, 16):
        rows = min(16, M - r0)
        _ffi.check(L.ia2p_linear_small(_ffi.current_stream(), _ffi.ptr(x2[r0:r0 + rows]), _ffi.ptr(w), _ffi.ptr(b), _ffi.ptr(y[r0:r0 + rows]),
                                       rows, N, K, 0, 0), None)
    return y.reshape(*lead, N)


class InstructAny2PixPrior:
    def __init__(self, sequence_gen_length=1, sequence_input_key=None, sequence_input_embed_dim=None, cond_stage_config=None,
                 pretrained_name="gpt2-medium", embed_dim=1024, output_dim=None, diffusion=True, *, device="cuda:0",
                 gpt_config: Optional[GPT2Config] = None, clip_config: Optional[CLIPTextConfig] = None, tokenizer: Optional[Callable] = None,
                 **unused):
        """Keyword surface of the reference constructor (training-only arguments are accepted and ignored). `gpt_config` / `clip_config`
        replace the checkpoint names the reference resolves online; `tokenizer` is the CLIP tokenizer callable of the text stage."""
        if pretrained_name != "gpt2-medium" and gpt_config is None:
            raise NotImplementedError(f"pass gpt_config= for pretrained_name={pretrained_name!r}")
        self.device = torch.device(device)
        self.diffusion = diffusion
        self.noise_scheduler = DDPMScheduler() if diffusion else None
        self.mae_token_num = sequence_gen_length
        self.sequence_input_key = list(sequence_input_key if sequence_input_key is not None else prior_config["sequence_input_key"])
        self.sequence_input_embed_dim = list(sequence_input_embed_dim if sequence_input_embed_dim is not None else prior_config["sequence_input_embed_dim"])
        self.cond_stage_config = cond_stage_config if cond_stage_config is not None else prior_config["cond_stage_config"]
        gcfg = gpt_config or gpt2_medium()
        if embed_dim != gcfg.n_embd:
            raise ValueError(f"embed_dim {embed_dim} != GPT-2 width {gcfg.n_embd}")
        self.embed_dim = embed_dim
        self.output_dim = output_dim
        if output_dim is not None and output_dim != embed_dim:
            raise NotImplementedError("output_proj is an Identity in the released configuration (output_dim == embed_dim)")
        self.model = HipGPT2Model(gcfg, device)
        self.cond_stage_models = [CLIPTextModelHiddenState(tokenizer, HipCLIPTextModel(clip_config or laion_clip_h_text(), device))]
        self.cond_stage_model_metadata = {k: {"model_idx": i, "cond_stage_key": v["cond_stage_key"], "conditioning_key": v["conditioning_key"]}
                                          for i, (k, v) in enumerate(self.cond_stage_config.items())}
        self._p: Dict[str, torch.Tensor] = {}

    def eval(self):
        return self

    def to(self, *a, **kw):
        return self

    # ---- weights ------------------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True):
        gpt, clip = {}, {}
        want = {"start_of_sequence_tokens.weight", "end_of_sequence_tokens.weight", "modality_embedding.weight"}
        want |= {f"input_sequence_embed_linear.{i}.{k}" for i, d in enumerate(self.sequence_input_embed_dim) if d for k in ("weight", "bias")}
        for k, v in state_dict.items():
            if k.startswith("model."):
                gpt[k[6:]] = v
            elif k.startswith("cond_stage_models.0.model."):
                clip[k[len("cond_stage_models.0.model."):]] = v
            elif k in want:
                self._p[k] = v.detach().to(device=self.device, dtype=torch.float16).contiguous()
            elif strict:
                raise KeyError(f"unexpected key {k} in the prior state dict")
        if strict and want - set(self._p):
            raise KeyError(f"missing keys in the prior state dict: {sorted(want - set(self._p))}")
        self.model.load_state_dict(gpt, strict=strict)
        self.cond_stage_models[0].model.load_state_dict(clip, strict=strict)

    # ---- conditioning ---------------------------------------------------------------------------------------------------------
    def get_input_item(self, batch, k):
        return list(batch[k]) if k == "text" else batch[k]

    def get_learned_conditioning(self, c, key, unconditional_cfg):
        stage = self.cond_stage_models[self.cond_stage_model_metadata[key]["model_idx"]]
        if not unconditional_cfg:
            return stage(c)
        if isinstance(c, torch.Tensor):
            return stage.get_unconditional_condition(c.size(0))
        if isinstance(c, list):
            return stage.get_unconditional_condition(len(c))
        raise NotImplementedError()

    def get_input(self, batch):
        cond = {}
        for key, meta in self.cond_stage_model_metadata.items():
            xc = self.get_input_item(batch, meta["cond_stage_key"])
            if isinstance(xc, torch.Tensor):
                xc = xc.to(self.device)
            cond[key] = self.get_learned_conditioning(xc, key=key, unconditional_cfg=False)
        return cond

    # ---- sequence assembly ------------------------------------------------------------------------------------------------------
    def add_sos_eos_tokens(self, _id, sequence, attn_mask):
        b = sequence.size(0)
        one = torch.ones((b, 1), device=sequence.device)
        sos = self._p["start_of_sequence_tokens.weight"][_id].view(1, 1, -1).expand(b, 1, -1)
        eos = self._p["end_of_sequence_tokens.weight"][_id].view(1, 1, -1).expand(b, 1, -1)
        return torch.cat([sos, sequence.to(sos.dtype), eos], dim=1), torch.cat([one, attn_mask, one], dim=1)

    def truncate_sequence_and_mask(self, sequence, mask, max_len=512):
        if sequence.size(1) > max_len:
            print("The input sequence length to GPT-2 model is too long:", sequence.size(1))
            return sequence[:, :max_len], mask[:, :max_len]
        return sequence, mask

    def _slot_linear(self, _id, x):
        if not self.sequence_input_embed_dim[_id]:
            return x                                                   # nn.Identity slot
        return _linear(x.to(self.device), self._p[f"input_sequence_embed_linear.{_id}.weight"], self._p[f"input_sequence_embed_linear.{_id}.bias"])

    def get_input_sequence_and_mask(self, cond_dict):
        seqs: List[torch.Tensor] = []
        masks: List[torch.Tensor] = []
        for _id, key in enumerate(self.sequence_input_key):
            if key not in cond_dict:
                continue
            v = cond_dict[key]
            if key in ("src_type", "tgt_type"):
                v = v[:, None] if v.ndim == 1 else v
                seqs.append(self._p["modality_embedding.weight"][v.to(self.device)])
                masks.append(torch.ones((v.size(0), v.size(1)), device=self.device))
            elif isinstance(v, list):
                assert len(v) == 2, "The crossattn returned list should have length 2, including embed and attn_mask"
                s, m = self.add_sos_eos_tokens(_id, self._slot_linear(_id, v[0]), v[1].to(self.device))
                seqs.append(s); masks.append(m)
            else:
                assert isinstance(v, torch.Tensor)
                e = self._slot_linear(_id, v)
                s, m = self.add_sos_eos_tokens(_id, e, torch.ones((e.size(0), e.size(1)), device=self.device))
                seqs.append(s); masks.append(m)
        assert seqs
        x, m = self.truncate_sequence_and_mask(torch.cat(seqs, dim=1), torch.cat(masks, dim=1), int(1024 - self.mae_token_num))
        return x, m, x.size(1)

    # ---- sampler ----------------------------------------------------------------------------------------------------------------
    def get_eps(self, timestep, sample, model_output):
        a = self.noise_scheduler.alphas_cumprod[int(timestep)]
        return (sample - a ** 0.5 * model_output) / (1 - a) ** 0.5

    @torch.no_grad()
    def generate_diffusion(self, src_type, tgt_type, src, no_grad=False, num_inference_steps=25, eta: float = 0.0, generator=None,
                           image_bind_overwrite=None, guidance_scale=5, score=6.8, negative_score=2.0, do_classifier_free_guidance=True,
                           device="cuda", dtype=torch.float16, no_diffusion=False, force_guidence_t0=False):
        """Same arguments as the reference; `device` / `dtype` are accepted and ignored (tensors live on this object's GPU, the
        transformer stacks compute in fp16, the returned latents are fp32 on the GPU)."""
        dev = self.device
        if no_diffusion:
            num_inference_steps = 1
        bs = raw_bs = len(src)
        src_key = "text" if src_type == MODALITY.TEXT else "imagebind"
        if image_bind_overwrite is None:
            image_bind_overwrite = torch.zeros(bs, 1, 1024)
        cond_dict = dict(
            src_type=torch.tensor(src_type).view(1, 1).repeat(bs, 1).to(dev),
            tgt_type=torch.tensor(tgt_type).view(1, 1).repeat(bs, 1).to(dev),
            score=get_timestep_embedding(torch.tensor([score], dtype=torch.float32), 512).view(1, 1, -1).repeat(bs, 1, 1).to(dev),
            text=[""],
            imagebind=image_bind_overwrite.to(dev).float(),
        )
        if src_key == "text":
            cond_dict[src_key] = src
        else:
            cond_dict[src_key] = src.view(bs, 1, -1).to(dev).float()
        if do_classifier_free_guidance:
            cond_dict["src_type"] = cond_dict["src_type"].repeat(2, 1)
            cond_dict["tgt_type"] = cond_dict["tgt_type"].repeat(2, 1)
            cond_dict["text"] = cond_dict["text"] + [""] * len(cond_dict["text"])
            cond_dict["imagebind"] = torch.cat([cond_dict["imagebind"], cond_dict["imagebind"] * 0.0], dim=0)
            cond_dict["score"] = torch.cat([cond_dict["score"], cond_dict["score"] * 0.0 + negative_score], dim=0)   # a constant vector, as in the reference
            bs = bs * 2
        sch = self.noise_scheduler
        sch.set_timesteps(num_inference_steps, device=dev)
        cond_dict.update(self.get_input(cond_dict))
        key = "noisy_input" if no_diffusion else "noisy_inputs"      # the no_diffusion key is not a sequence slot: the model never sees the sample
        # reference :601 `torch.randn(...).to(device).to(src_type)` with src_type the int64 modality tensor: the start noise is truncated
        # toward zero to integers. Drawn from the global CPU RNG as there.
        sample = torch.randn(raw_bs, 1, self.embed_dim).to(torch.int64).to(device=dev, dtype=torch.float32)
        cond_dict[key] = sample.repeat(2, 1, 1) if do_classifier_free_guidance else sample
        n = raw_bs * self.embed_dim
        for t in sch.timesteps:
            rows = cond_dict[key].shape[0]
            cond_dict["noise_level"] = get_timestep_embedding(torch.ones((rows,), device=dev) * t, self.embed_dim)
            x, m, end = self.get_input_sequence_and_mask(cond_dict)
            for _ in range(self.mae_token_num):
                out = self.model(inputs_embeds=x, attention_mask=m)["last_hidden_state"]
                x = torch.cat([x.to(out.dtype), out[:, -1:, :]], dim=1)
                m = torch.cat([m, torch.ones((m.size(0), 1), device=dev)], dim=1)
            output = x[:, end:].contiguous()                                   # [bs, mae_token_num, E] fp16
            if sch.config.num_train_timesteps // sch.num_inference_steps >= 0 or force_guidence_t0:
                if self.mae_token_num != 1:
                    raise NotImplementedError("the sampler update is written for sequence_gen_length = 1 (the released configuration)")
                sa, sb, k0, k1, sigma = sch.posterior_coeffs(int(t))
                noise = None
                if int(t) > 0:                                                 # DDPMScheduler.step draws its variance noise on the reference's device (CPU)
                    gdev = generator.device if generator is not None else "cpu"
                    noise = torch.randn((raw_bs, 1, self.embed_dim), generator=generator, device=gdev, dtype=torch.float32).to(dev).contiguous()
                cur = cond_dict[key][:raw_bs].to(torch.float32).contiguous()
                latents = torch.empty_like(cur)
                if do_classifier_free_guidance:
                    prior_update(cur, output[:raw_bs].contiguous(), output[raw_bs:].contiguous(), noise, guidance_scale, sa, sb, k0, k1, sigma, latents)
                    latents = latents.repeat(2, 1, 1)
                else:
                    prior_update(cur, None, output, noise, 1.0, sa, sb, k0, k1, sigma, latents)
            else:
                latents = output[:raw_bs].float()
            cond_dict[key] = latents
        return cond_dict[key][:raw_bs], cond_dict

    def generate(self, batch, cond_dict=None, no_grad=False):
        raise NotImplementedError("the auto-regressive (non-diffusion) `generate` is not called by the reference pipeline; use generate_diffusion")
