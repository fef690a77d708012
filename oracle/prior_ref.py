"""ORACLE (test infrastructure): CPU fp32 restatement of the reference's embedding prior (second half).

  GPT2ModelRef            <- transformers `GPT2Model` driven with `inputs_embeds` (reference instructany2pix/prior/model.py:185,
                             :493-495, :611-613). Third-party dependency of the reference, not vendored; transformers IS installed in
                             this image, so the restatement is pinned against the real class
                             (tests/test_oracle_golden.py::test_gpt2_restatement_matches_transformers: same random weights, <= 2e-5).
  DDPMSchedulerRef        <- diffusers 0.26.3 `DDPMScheduler` (un-vendored, NOT installed: restated from its published algorithm,
                             `set_timesteps` "leading" + offset, `step` with the fixed_small posterior variance clamped at 1e-20).
                             Pinned only through the reference's own `generate_diffusion` text run on top of it (fixture G14); its
                             tables are the DDIM ones pinned by G5.
  timestep_embedding_ref  <- diffusers `get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)` = the ldm
                             `timestep_embedding` pinned by fixture G6.
  PriorRef                <- `InstructAny2PixPrior.generate_diffusion` / `get_input_sequence_and_mask` / `add_sos_eos_tokens` /
                             `get_eps` (prior/model.py:208-240, :272-381, :528-658), pinned by fixture G14 (tests/golden/prior.npz:
                             the reference's own method text executed on stand-ins, tests/golden/gen_goldens.py::gen_prior).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class _Conv1D(nn.Module):          # transformers Conv1D: weight [in, out], y = x @ W + b
    def __init__(self, nin, nout):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.zeros(nin, nout)), nn.Parameter(torch.zeros(nout))

    def forward(self, x):
        return x @ self.weight + self.bias


class _GAttn(nn.Module):
    def __init__(self, e, heads):
        super().__init__()
        self.heads = heads
        self.c_attn, self.c_proj = _Conv1D(e, 3 * e), _Conv1D(e, e)

    def forward(self, x):
        b, t, e = x.shape
        d = e // self.heads
        q, k, v = (y.view(b, t, self.heads, d).transpose(1, 2) for y in self.c_attn(x).split(e, dim=2))
        s = q @ k.transpose(-1, -2) / math.sqrt(d)
        s = s + torch.full((t, t), float("-inf")).triu(1)
        return self.c_proj((s.softmax(-1) @ v).transpose(1, 2).reshape(b, t, e))


class _GMLP(nn.Module):
    def __init__(self, e, i, act):
        super().__init__()
        self.c_fc, self.c_proj, self.act = _Conv1D(e, i), _Conv1D(i, e), act

    def forward(self, x):
        x = self.c_fc(x)
        if self.act == "gelu_new":
            x = 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))
        else:
            x = F.gelu(x)
        return self.c_proj(x)


class _GBlock(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        e = cfg.n_embd
        self.ln_1 = nn.LayerNorm(e, eps=cfg.layer_norm_epsilon)
        self.attn = _GAttn(e, cfg.n_head)
        self.ln_2 = nn.LayerNorm(e, eps=cfg.layer_norm_epsilon)
        self.mlp = _GMLP(e, cfg.inner, cfg.activation_function)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class GPT2ModelRef(nn.Module):
    """module tree = transformers' GPT2Model (state-dict keys match)"""

    def __init__(self, cfg):
        super().__init__()
        self.config = cfg
        self.wte = nn.Embedding(cfg.vocab_size, cfg.n_embd)
        self.wpe = nn.Embedding(cfg.n_positions, cfg.n_embd)
        self.h = nn.ModuleList([_GBlock(cfg) for _ in range(cfg.n_layer)])
        self.ln_f = nn.LayerNorm(cfg.n_embd, eps=cfg.layer_norm_epsilon)

    @torch.no_grad()
    def forward(self, inputs_embeds, attention_mask=None):
        if attention_mask is not None and not bool((attention_mask != 0).all()):
            raise NotImplementedError("padded sequences are not on the reference's live path")
        x = inputs_embeds + self.wpe(torch.arange(inputs_embeds.shape[1]))[None]
        for blk in self.h:
            x = blk(x)
        return {"last_hidden_state": self.ln_f(x)}


def build_gpt2(cfg, state_dict, prefix=""):
    m = GPT2ModelRef(cfg)
    sd = {k[len(prefix):]: v.float() for k, v in state_dict.items() if k.startswith(prefix)}
    m.load_state_dict({k: v for k, v in sd.items() if not k.endswith((".attn.bias", ".attn.masked_bias"))}, strict=True)
    return m.eval()


def timestep_embedding_ref(timesteps, dim):
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half)
    args = timesteps[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class DDPMSchedulerRef:
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.config = type("C", (), dict(num_train_timesteps=num_train_timesteps, steps_offset=steps_offset))()
        self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)
        self.one = torch.tensor(1.0)
        self.num_inference_steps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        ratio = self.config.num_train_timesteps // n
        self.timesteps = torch.tensor([i * ratio + self.config.steps_offset for i in range(n)][::-1], dtype=torch.int64)

    def step(self, model_output, timestep, sample, generator=None, return_dict=False):
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.one
        b_t, b_p = 1 - a_t, 1 - a_p
        cur_a = a_t / a_p
        cur_b = 1 - cur_a
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        prev_sample = (a_p ** 0.5 * cur_b) / b_t * x0 + cur_a ** 0.5 * b_p / b_t * sample
        if t > 0:
            z = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            var = torch.clamp((1 - a_p) / (1 - a_t) * cur_b, min=1e-20)
            prev_sample = prev_sample + (var ** 0.5) * z
        return (prev_sample,)


class PriorRef:
    """fp32 restatement of the reference prior's inference path. `sd` = InstructAny2PixPrior state dict; `text_hidden(prompts)` returns
    [last_hidden_state, attention_mask.float()] of the conditioning text tower (CLIPTextModelHiddenState, prior/model.py:80-105)."""
    # the reference's key list, with its missing comma between the last two entries (prior/__init__.py:13-22)
    sequence_input_key = ["src_type", "imagebind", "crossattn_clip", "score", "noisy_inputs", "noise_level" "tgt_type"]

    def __init__(self, gpt_cfg, sd, text_hidden, sequence_gen_length=1):
        self.sd = {k: v.float() for k, v in sd.items()}
        self.model = build_gpt2(gpt_cfg, sd, "model.")
        self.text_hidden = text_hidden
        self.embed_dim = gpt_cfg.n_embd
        self.mae_token_num = sequence_gen_length
        self.noise_scheduler = DDPMSchedulerRef()

    def _linear(self, i, x):
        if f"input_sequence_embed_linear.{i}.weight" not in self.sd:        # nn.Identity slot (input width 0)
            return x
        return F.linear(x, self.sd[f"input_sequence_embed_linear.{i}.weight"], self.sd[f"input_sequence_embed_linear.{i}.bias"])

    def _sos_eos(self, i, seq, mask):
        b = seq.shape[0]
        one = torch.ones(b, 1)
        sos = self.sd["start_of_sequence_tokens.weight"][i][None, None].expand(b, 1, -1)
        eos = self.sd["end_of_sequence_tokens.weight"][i][None, None].expand(b, 1, -1)
        return torch.cat([sos, seq, eos], dim=1), torch.cat([one, mask, one], dim=1)

    def get_input_sequence_and_mask(self, cond):
        seqs, masks = [], []
        for i, key in enumerate(self.sequence_input_key):
            if key not in cond:
                continue
            v = cond[key]
            if key in ("src_type", "tgt_type"):
                v = v[:, None] if v.ndim == 1 else v
                seqs.append(self.sd["modality_embedding.weight"][v])
                masks.append(torch.ones(v.shape[0], v.shape[1]))
            elif isinstance(v, list):
                s, m = self._sos_eos(i, self._linear(i, v[0]), v[1])
                seqs.append(s); masks.append(m)
            else:
                e = self._linear(i, v)
                s, m = self._sos_eos(i, e, torch.ones(e.shape[0], e.shape[1]))
                seqs.append(s); masks.append(m)
        x, m = torch.cat(seqs, dim=1), torch.cat(masks, dim=1)
        lim = 1024 - self.mae_token_num
        return x[:, :lim], m[:, :lim], min(x.shape[1], lim)

    def get_eps(self, t, sample, model_output):
        a = self.noise_scheduler.alphas_cumprod[t]
        return (sample - a ** 0.5 * model_output) / (1 - a) ** 0.5

    @torch.no_grad()
    def generate_diffusion(self, src_type, tgt_type, src, num_inference_steps=25, generator=None, image_bind_overwrite=None, guidance_scale=5,
                           score=6.8, negative_score=2.0, do_classifier_free_guidance=True, no_diffusion=False, force_guidence_t0=False):
        if no_diffusion:
            num_inference_steps = 1
        bs = raw = len(src)
        src_key = "text" if src_type == 2 else "imagebind"
        if image_bind_overwrite is None:
            image_bind_overwrite = torch.zeros(bs, 1, 1024)
        cond = dict(src_type=torch.tensor(src_type).view(1, 1).repeat(bs, 1), tgt_type=torch.tensor(tgt_type).view(1, 1).repeat(bs, 1),
                    score=timestep_embedding_ref(torch.tensor([score], dtype=torch.float32), 512).view(1, 1, -1).repeat(bs, 1, 1),
                    text=[""], imagebind=image_bind_overwrite.float())
        cond[src_key] = src if src_key == "text" else src.view(bs, 1, -1).float()
        if do_classifier_free_guidance:
            cond["src_type"], cond["tgt_type"] = cond["src_type"].repeat(2, 1), cond["tgt_type"].repeat(2, 1)
            cond["text"] = cond["text"] + [""] * len(cond["text"])
            cond["imagebind"] = torch.cat([cond["imagebind"], cond["imagebind"] * 0.0], dim=0)
            cond["score"] = torch.cat([cond["score"], cond["score"] * 0.0 + negative_score], dim=0)
        sch = self.noise_scheduler
        sch.set_timesteps(num_inference_steps)
        cond["crossattn_clip"] = self.text_hidden(list(cond["text"]))
        key = "noisy_input" if no_diffusion else "noisy_inputs"
        # `.to(src_type)` in the reference converts to the dtype of the (int64) modality tensor: the start noise is truncated to integers
        cond[key] = torch.randn(raw, 1, self.embed_dim).to(torch.int64)          # global RNG, as in the reference (:601)
        if do_classifier_free_guidance:
            cond[key] = cond[key].repeat(2, 1, 1)
        for t in sch.timesteps:
            cond["noise_level"] = timestep_embedding_ref(torch.ones(cond[key].shape[0]) * t, cond[key].shape[-1])
            x, m, end = self.get_input_sequence_and_mask(cond)
            for _ in range(self.mae_token_num):
                out = self.model(x, m)["last_hidden_state"]
                x = torch.cat([x, out[:, -1:, :]], dim=1)
                m = torch.cat([m, torch.ones(m.shape[0], 1)], dim=1)
            output = x[:, end:]
            if sch.config.num_train_timesteps // sch.num_inference_steps >= 0 or force_guidence_t0:
                eps = self.get_eps(t, cond[key], output)
                if do_classifier_free_guidance:
                    e_text, e_uncond = eps.chunk(2)
                    eps = e_uncond + guidance_scale * (e_text - e_uncond)
                latents = sch.step(eps, t, cond[key][:raw], generator=generator)[0]
                if do_classifier_free_guidance:
                    latents = latents.repeat(2, 1, 1)
            else:
                latents = output[:raw]
            cond[key] = latents
        return cond[key][:raw], cond
