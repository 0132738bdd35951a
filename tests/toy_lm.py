"""Tiny duck-typed models used to drive BOTH the reference sample() (fixture
generation, build container only) and this repo's loops (tests, smoke).

* ToyVLM   — an order/position-sensitive 1-layer causal-attention LM with a real KV
             cache and a LLaVA-style image splice (n_img patch embeddings replace the
             single -200 placeholder, as experiments/llava/model/llava_arch.py:122-163
             does with 576), so branch bookkeeping bugs change the tokens.
* BankModel — ignores its inputs and replays pre-generated logit rows, one per forward
             call, so per-step kernel vectors can be pushed through a full loop.

Both expose the protocol the reference loop needs.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch

IMAGE_TOKEN_INDEX = -200


class _GenCfg(SimpleNamespace):
    pass


def _gen_cfg(pad, eos):
    return _GenCfg(pad_token_id=pad, eos_token_id=eos, output_scores=False, output_attentions=False,
                   output_hidden_states=False, return_dict_in_generate=False)


class _Proto:
    """4.31-era GenerationMixin hooks the loop calls."""
    config = SimpleNamespace(is_encoder_decoder=False)

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None,
                                      inputs_embeds=None, **kw):
        if past_key_values:
            input_ids = input_ids[:, -1:]
        return {"input_ids": input_ids, "past_key_values": past_key_values, "use_cache": kw.get("use_cache"),
                "attention_mask": attention_mask, "images": kw.get("images", None)}

    def prepare_inputs_for_generation_cd(self, input_ids, past_key_values=None, attention_mask=None,
                                         inputs_embeds=None, **kw):
        if past_key_values:
            input_ids = input_ids[:, -1:]
        return {"input_ids": input_ids, "past_key_values": past_key_values, "use_cache": kw.get("use_cache"),
                "attention_mask": attention_mask, "images": kw.get("images_cd", None)}

    def _update_model_kwargs_for_generation(self, outputs, model_kwargs, is_encoder_decoder=False, **_):
        model_kwargs["past_key_values"] = outputs.past_key_values
        am = model_kwargs.get("attention_mask")
        if am is not None:
            model_kwargs["attention_mask"] = torch.cat([am, am.new_ones((am.shape[0], 1))], dim=-1)
        return model_kwargs


class ToyVLM(_Proto):
    def __init__(self, vocab=97, d=32, n_img=5, img_dim=12, seed=0, logit_dtype=torch.float32,
                 pad=0, eos=None, device="cpu", logit_scale=6.0):
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g)
        self.vocab, self.d, self.n_img = vocab, d, n_img
        self.emb = r(vocab, d)
        self.pos = r(1024, d) * 0.7
        self.wq, self.wk, self.wv, self.wo = (r(d, d) / math.sqrt(d) for _ in range(4))
        self.head = r(d, vocab) * (logit_scale / math.sqrt(d))
        self.img_proj = r(img_dim, n_img * d) / math.sqrt(img_dim)
        self.logit_dtype = logit_dtype
        self.generation_config = _gen_cfg(pad, eos)
        self.device = torch.device(device)
        for n in ("emb", "pos", "wq", "wk", "wv", "wo", "head", "img_proj"):
            setattr(self, n, getattr(self, n).to(self.device))
        self.calls = []

    def _embed(self, input_ids, images):
        B, L = input_ids.shape
        if images is None or L == 1:
            return self.emb[input_ids]                      # image-free branches: ids must be >= 0
        rows = []
        for b in range(B):
            ids = input_ids[b]
            at = torch.where(ids == IMAGE_TOKEN_INDEX)[0]
            if at.numel() == 0:
                rows.append(self.emb[ids])
                continue
            s = int(at[0])
            feat = (images[b].reshape(-1).float()[: self.img_proj.shape[0]] @ self.img_proj).view(self.n_img, self.d)
            rows.append(torch.cat([self.emb[ids[:s]], feat, self.emb[ids[s + 1:]]], dim=0))
        return torch.stack(rows, dim=0)

    def __call__(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None,
                 use_cache=None, images=None, image_sizes=None, return_dict=True,
                 output_attentions=None, output_hidden_states=None, **_):
        past_len = int(past_key_values[0][0].shape[-2]) if past_key_values else 0
        self.calls.append((tuple(input_ids.shape), tuple(attention_mask.shape) if attention_mask is not None else None,
                           images is not None, past_len))
        x = self._embed(input_ids, images)
        T = x.shape[1]
        x = x + self.pos[past_len:past_len + T]
        q, k, v = x @ self.wq, x @ self.wk, x @ self.wv
        if past_key_values:
            k = torch.cat([past_key_values[0][0][:, 0], k], dim=1)
            v = torch.cat([past_key_values[0][1][:, 0], v], dim=1)
        att = (q @ k.transpose(1, 2)) / math.sqrt(self.d)
        S = k.shape[1]
        causal = torch.ones(T, S, dtype=torch.bool, device=x.device).tril(diagonal=S - T)
        att = att.masked_fill(~causal, -float("inf")).softmax(-1)
        h = x + (att @ v) @ self.wo
        logits = (torch.tanh(h) @ self.head).to(self.logit_dtype)
        return SimpleNamespace(logits=logits, past_key_values=((k[:, None], v[:, None]),),
                               attentions=None, hidden_states=None)


class BankModel(_Proto):
    """Replays bank[i] ([B, V]) on the i-th forward call as the last-position logits."""

    def __init__(self, bank, pad=0, eos=None):
        self.bank = bank
        self.i = 0
        self.generation_config = _gen_cfg(pad, eos)

    def __call__(self, input_ids=None, past_key_values=None, **_):
        row = self.bank[self.i]
        self.i += 1
        n = (int(past_key_values[0][0].shape[-2]) if past_key_values else 0) + 1
        dummy = torch.zeros(1, 1, n, 1)
        return SimpleNamespace(logits=row[:, None, :], past_key_values=((dummy, dummy),),
                               attentions=None, hidden_states=None)


def pope_like_ids(rng, n_sys=35, n_text=24, vocab=97, batch=1):
    """[n_sys sys tokens] + [-200] + [n_text question tokens]."""
    import numpy as np
    rows = []
    for _ in range(batch):
        sys_ = rng.integers(3, vocab, size=n_sys)
        txt = rng.integers(3, vocab, size=n_text)
        rows.append(np.concatenate([[1], sys_[1:], [IMAGE_TOKEN_INDEX], txt]))
    return torch.tensor(np.stack(rows), dtype=torch.long)
