"""SDXL text encoders on the HIP kernels and the `encode_prompt` stage in front of the denoise loop.

  HipCLIPTextModel(config)           <- transformers `CLIPTextModel` / `CLIPTextModelWithProjection` (`pipe.text_encoder`,
                                        `pipe.text_encoder_2`; reference instructany2pix/pipeline.py:132-139 passes both on)
  SDXLTextEncoders.encode_prompt(..) <- `encode_prompt`, vendored at instructany2pix/ddim/sdxl_pipeline.py:202-395: both encoders on
                                        the max-length token ids, `hidden_states[-2]` of each concatenated to the 2048-d context,
                                        pooled `[0]` of the second one, zeros for an absent negative prompt
                                        (`force_zeros_for_empty_prompt`), repeat per image
Everything after tokenisation runs through `ia2p_clip_encode`. Tokenizers are injected callables (the BPE vocabularies are
checkpoint data): `tokenizer(prompts, padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids`.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace
from typing import Callable, List, Optional

import torch

from . import _ffi
from .config import CLIPTextConfig


class _PenultimateOnly(tuple):
    """`hidden_states` of the output object: only [-2] is materialised (the only entry the reference reads, :318)."""

    def __new__(cls, h, n):
        t = super().__new__(cls, (h,))
        t._n = n
        return t

    def __getitem__(self, i):
        if i in (-2, self._n - 2):
            return tuple.__getitem__(self, 0)
        raise IndexError("only hidden_states[-2] is computed on the HIP path")

    def __len__(self):
        return self._n


class CLIPTextOutput(SimpleNamespace):
    def __getitem__(self, i):          # tuple-style access as on transformers' ModelOutput (fields that were not computed are None)
        return [getattr(self, k) for k in self._order][i]


class HipCLIPTextModel:
    def __init__(self, config: CLIPTextConfig, device="cuda:0"):
        self.config = config.validate()
        self.device = torch.device(device)
        self._lib = _ffi.lib()
        self._h = C.c_void_p()
        _ffi.check(self._lib.ia2p_clip_create(C.byref(_ffi.make_clip_config(config)), C.byref(self._h)), None, clip=True)
        with torch.cuda.device(self.device):
            self.arena = torch.zeros(self._lib.ia2p_clip_arena_bytes(self._h), dtype=torch.uint8, device=self.device)
        _ffi.check(self._lib.ia2p_clip_bind_arena(self._h, _ffi.ptr(self.arena), self.arena.numel()), self._h, clip=True)
        self._ws = None
        self.dtype = torch.float16

    def __del__(self):
        try:
            if self._h:
                self._lib.ia2p_clip_destroy(self._h)
        except Exception:
            pass

    def to(self, *a, **kw):
        return self

    def load_state_dict(self, state_dict, strict: bool = True):
        items = state_dict.items() if hasattr(state_dict, "items") else state_dict
        for k, v in items:
            if k.endswith("position_ids"):          # buffer in older transformers checkpoints
                continue
            t = v.detach().to(device=self.device, dtype=torch.float16).contiguous()
            shape = (C.c_int64 * t.ndim)(*t.shape)
            _ffi.check(self._lib.ia2p_clip_load_tensor(self._h, k.encode(), _ffi.ptr(t), shape, t.ndim, _ffi.current_stream()), self._h, clip=True)
            torch.cuda.current_stream().synchronize()
        if strict:
            _ffi.check(self._lib.ia2p_clip_finalize_weights(self._h), self._h, clip=True)

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, output_hidden_states: bool = False, want_last_hidden: Optional[bool] = None,
                 want_pooled: bool = True, **unused):
        """want_last_hidden / want_pooled = False skip work the caller does not read (the last layer, when only hidden_states[-2]
        is wanted: what `encode_prompt` needs from the first encoder); defaults compute what transformers returns."""
        if input_ids.ndim != 2:
            raise ValueError("input_ids must be [batch, tokens]")
        B, T = input_ids.shape
        ids = input_ids.to(device=self.device, dtype=torch.int32).contiguous()
        n = self._lib.ia2p_clip_workspace_bytes(self._h, B, T)
        if n == 0:
            _ffi.check(2, self._h, clip=True)
        if self._ws is None or self._ws.numel() < n:
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        cfg = self.config
        hid2 = torch.empty(B, T, cfg.hidden_size, dtype=torch.float16, device=self.device) if output_hidden_states else None
        if want_last_hidden is None:
            want_last_hidden = not output_hidden_states
        last = torch.empty(B, T, cfg.hidden_size, dtype=torch.float16, device=self.device) if want_last_hidden else None
        pooled = torch.empty(B, cfg.projection_dim or cfg.hidden_size, dtype=torch.float16, device=self.device) if want_pooled else None
        if hid2 is None and last is None and pooled is None:
            raise ValueError("nothing requested")
        _ffi.check(self._lib.ia2p_clip_encode(self._h, _ffi.current_stream(), _ffi.ptr(ids), B, T, _ffi.ptr(hid2), _ffi.ptr(last), _ffi.ptr(pooled),
                                              _ffi.ptr(self._ws), self._ws.numel()), self._h, clip=True)
        hs = _PenultimateOnly(hid2, cfg.num_hidden_layers + 1) if hid2 is not None else None
        if cfg.projection_dim:            # CLIPTextModelWithProjection: (text_embeds, last_hidden_state, hidden_states)
            out = CLIPTextOutput(text_embeds=pooled, last_hidden_state=last, hidden_states=hs)
            out._order = ("text_embeds", "last_hidden_state", "hidden_states")
        else:                             # CLIPTextModel: (last_hidden_state, pooler_output, hidden_states)
            out = CLIPTextOutput(last_hidden_state=last, pooler_output=pooled, hidden_states=hs)
            out._order = ("last_hidden_state", "pooler_output", "hidden_states")
        return out


class SDXLTextEncoders:
    """`encode_prompt` of the SDXL pipelines with the reference's argument meaning and error behaviour (:202-395)."""

    def __init__(self, tokenizer: Optional[Callable], tokenizer_2: Callable, text_encoder: Optional[HipCLIPTextModel], text_encoder_2: HipCLIPTextModel,
                 force_zeros_for_empty_prompt: bool = True):
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self.config = SimpleNamespace(force_zeros_for_empty_prompt=force_zeros_for_empty_prompt)

    def _run(self, texts, max_length=None):
        tokenizers = [self.tokenizer, self.tokenizer_2] if self.tokenizer is not None else [self.tokenizer_2]
        encoders = [self.text_encoder, self.text_encoder_2] if self.text_encoder is not None else [self.text_encoder_2]
        embeds, pooled = [], None
        for j, (text, tok, enc) in enumerate(zip(texts, tokenizers, encoders)):
            ids = tok(text, padding="max_length", max_length=max_length or getattr(tok, "model_max_length", 77), truncation=True, return_tensors="pt").input_ids
            final = j == len(encoders) - 1
            out = enc(ids, output_hidden_states=True, want_pooled=final)
            if final:                                         # "We are only ALWAYS interested in the pooled output of the final text encoder"
                pooled = out[0]
            embeds.append(out.hidden_states[-2])
        return torch.concat(embeds, dim=-1), pooled

    @torch.no_grad()
    def encode_prompt(self, prompt=None, prompt_2=None, device=None, num_images_per_prompt: int = 1, do_classifier_free_guidance: bool = True,
                      negative_prompt=None, negative_prompt_2=None, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                      negative_pooled_prompt_embeds=None, lora_scale=None):
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]
        if prompt_embeds is None:
            prompt_2 = prompt_2 or prompt
            prompt_embeds, pooled_prompt_embeds = self._run([prompt, prompt_2])
        zero_out = negative_prompt is None and self.config.force_zeros_for_empty_prompt
        if do_classifier_free_guidance and negative_prompt_embeds is None and zero_out:
            negative_prompt_embeds = torch.zeros_like(prompt_embeds)
            negative_pooled_prompt_embeds = torch.zeros_like(pooled_prompt_embeds)
        elif do_classifier_free_guidance and negative_prompt_embeds is None:
            negative_prompt = negative_prompt or ""
            negative_prompt_2 = negative_prompt_2 or negative_prompt
            if prompt is not None and type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} != {type(prompt)}.")
            if not isinstance(negative_prompt, str) and batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`: {prompt} has batch size "
                                 f"{batch_size}. Please make sure that passed `negative_prompt` matches the batch size of `prompt`.")
            negative_prompt_embeds, negative_pooled_prompt_embeds = self._run([negative_prompt, negative_prompt_2], max_length=prompt_embeds.shape[1])
        bs_embed, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs_embed * num_images_per_prompt, seq_len, -1)
        if do_classifier_free_guidance:
            seq_len = negative_prompt_embeds.shape[1]
            negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1).view(batch_size * num_images_per_prompt, seq_len, -1)
        pooled_prompt_embeds = pooled_prompt_embeds.repeat(1, num_images_per_prompt).view(bs_embed * num_images_per_prompt, -1)
        if do_classifier_free_guidance:
            negative_pooled_prompt_embeds = negative_pooled_prompt_embeds.repeat(1, num_images_per_prompt).view(bs_embed * num_images_per_prompt, -1)
        return prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds
