"""Deterministic stand-in for transformers' CLIPTokenizer, shared by tests/golden/gen_goldens.py (fixture G11) and the tests: the BPE
vocabularies are checkpoint data that is not available here, and `encode_prompt` only needs `tokenizer(...).input_ids`."""
import types

import torch


class StubTokenizer:
    """deterministic stand-in for CLIPTokenizer (the BPE vocabularies are checkpoint data): ids from the characters of the prompt,
    BOS 0, EOS / padding = vocab_size - 1 (SDXL pads with EOS), `padding="longest"` / `True` returns the unpadded row length;
    `attention_mask` marks the unpadded tokens (read by the prior's text stage, reference prior/model.py:82-91)"""
    model_max_length = 77

    def __init__(self, salt, vocab):
        self.salt, self.vocab = salt, vocab

    def __call__(self, text, padding=None, max_length=None, truncation=False, return_tensors="pt"):
        texts = [text] if isinstance(text, str) else list(text)
        rows = []
        for t in texts:
            n = min(len(t.split()) + 2, 77)
            g = torch.Generator().manual_seed(sum(map(ord, t)) + self.salt)
            body = torch.randint(3, self.vocab - 1, (n - 2,), generator=g)
            rows.append(torch.cat([torch.zeros(1, dtype=torch.long), body, torch.full((1,), self.vocab - 1, dtype=torch.long)]))
        L = max_length if padding == "max_length" else max(len(r) for r in rows)
        out = torch.full((len(rows), L), self.vocab - 1, dtype=torch.long)
        mask = torch.zeros((len(rows), L), dtype=torch.long)
        for i, r in enumerate(rows):
            out[i, :min(len(r), L)] = r[:L]
            mask[i, :min(len(r), L)] = 1
        return types.SimpleNamespace(input_ids=out, attention_mask=mask)

    def batch_decode(self, ids):
        return ["<truncated>"] * len(ids)
