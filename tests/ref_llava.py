"""Plain-PyTorch fp32 reference of the LLaVA-1.5 forward (CLIP ViT -> mlp2x_gelu projector ->
multimodal splice -> Llama), written against the reference's model protocol
(llava_llama.py:58-174, llava_arch.py:82-204) so that oracle.reference_loop can drive it exactly
like the reference's sample() drives LlavaLlamaForCausalLM.  Test infrastructure only."""
from __future__ import annotations

import math
from types import SimpleNamespace

import torch
import torch.nn.functional as F

from toy_lm import IMAGE_TOKEN_INDEX, _Proto, _gen_cfg


def splice(embed, row, feat):
    """llava_arch.py:122-163 for one sequence with one image slot: the text chunks embedded around the projected patch features
    (pinned to the reference's own function by tests/golden/splice.npz, tests/test_splice_golden.py)."""
    s = int(torch.where(row == IMAGE_TOKEN_INDEX)[0][0])
    return torch.cat([embed[row[:s]], feat, embed[row[s + 1:]]], 0)


class RefLlava(_Proto):
    def __init__(self, weights, device="cpu", logit_dtype=None, pad=0, eos=None, dtype=torch.float32,
                 output_attentions=False, logits_on_device=False, store=None):
        """dtype=float32: numerics reference.  dtype=bfloat16/float16: what the reference's eager HF stack executes
        (weights and matmuls in the model dtype, fp32 softmax / norm statistics, KV cache grown by torch.cat)."""
        self.cfg = weights.cfg
        self.dtype = dtype
        # what the engine under test stores pixels / embeddings in (its weights' dtype): inputs are rounded through it first
        self.store = store if store is not None else (weights.dtype if getattr(weights, "dtype", None) in (torch.bfloat16, torch.float16) else torch.bfloat16)
        self.logits_on_device = logits_on_device              # the eager GPU pipeline keeps the logits where they were computed
        self.materialize_attn = output_attentions            # llava_calibrate.py:175 asks for the [H, T, S] maps every step
        self.w = {k: v.detach().to(device=device, dtype=dtype) for k, v in weights.t.items()}
        self.device = torch.device(device)
        self.logit_dtype = logit_dtype if logit_dtype is not None else self.store      # the model dtype (vcd_sample.py:119 reads outputs.logits)
        self.generation_config = _gen_cfg(pad, eos)
        lm = self.cfg.lm
        inv = 1.0 / (lm.rope_theta ** (torch.arange(0, lm.head_dim, 2, dtype=torch.float32) / lm.head_dim))
        ang = torch.arange(lm.max_pos, dtype=torch.float32)[:, None] * inv[None, :]
        self.cos, self.sin = ang.cos().to(device), ang.sin().to(device)
        self.calls = []

    # ---- vision ----
    def encode_images(self, images):
        v, w = self.cfg.vision, self.w
        x = images.to(self.device, self.store).to(self.dtype)
        n = x.shape[0]
        P, G = v.patch, v.image // v.patch
        patches = x.view(n, 3, G, P, G, P).permute(0, 2, 4, 1, 3, 5).reshape(n, G * G, 3 * P * P)
        h = patches @ w["v.patch"][:, : 3 * P * P].t()
        h = torch.cat([w["v.cls"].view(1, 1, -1).expand(n, 1, -1), h], 1) + w["v.pos"][None]
        h = F.layer_norm(h, (v.width,), w["v.pre_ln.w"], w["v.pre_ln.b"], v.eps)
        sm = lambda t: t.float().softmax(-1).to(self.dtype)
        H, D = v.heads, v.width // v.heads
        for i in range(v.run_layers):
            p = f"v{i}."
            a = F.layer_norm(h, (v.width,), w[p + "ln1.w"], w[p + "ln1.b"], v.eps)
            qkv = (a @ w[p + "wqkv"].t() + w[p + "bqkv"]).view(n, -1, 3, H, D)
            q, k, val = (qkv[:, :, j].transpose(1, 2) for j in range(3))
            att = sm((q @ k.transpose(-1, -2)) / math.sqrt(D)) @ val
            h = h + att.transpose(1, 2).reshape(n, -1, v.width) @ w[p + "wo"].t() + w[p + "bo"]
            a = F.layer_norm(h, (v.width,), w[p + "ln2.w"], w[p + "ln2.b"], v.eps)
            f = a @ w[p + "fc1"].t() + w[p + "b1"]
            f = f * torch.sigmoid(1.702 * f)
            h = h + f @ w[p + "fc2"].t() + w[p + "b2"]
        z = F.gelu(h[:, 1:] @ w["mm.w1"].t() + w["mm.b1"])
        return z @ w["mm.w2"].t() + w["mm.b2"]

    # ---- language model ----
    def _rope(self, x, pos):            # x [1, H, T, D]
        D = x.shape[-1]
        c, s = self.cos[pos][None, None].to(x.dtype), self.sin[pos][None, None].to(x.dtype)
        a, b = x[..., : D // 2], x[..., D // 2:]
        return torch.cat([a * c - b * s, b * c + a * s], -1)

    def _lm(self, emb, past):
        lm, w = self.cfg.lm, self.w
        T = emb.shape[1]
        p0 = past[0][0].shape[-2] if past else 0
        pos = torch.arange(p0, p0 + T, device=self.device)
        h = emb
        new = []
        rms = lambda x, g: (x.float() * torch.rsqrt(x.float().pow(2).mean(-1, keepdim=True) + lm.eps)).to(x.dtype) * g
        H, Hkv, D = lm.n_heads, lm.n_kv_heads, lm.head_dim
        for i in range(lm.n_layers):
            p = f"l{i}."
            a = rms(h, w[p + "ln1"])
            qkv = a @ w[p + "wqkv"].t()
            if lm.qkv_bias:
                qkv = qkv + w[p + "bqkv_lm"]
            q = qkv[..., : H * D].view(1, T, H, D).transpose(1, 2)
            k = qkv[..., H * D: (H + Hkv) * D].view(1, T, Hkv, D).transpose(1, 2)
            v = qkv[..., (H + Hkv) * D:].view(1, T, Hkv, D).transpose(1, 2)
            q, k = self._rope(q, pos), self._rope(k, pos)
            if past:
                k, v = torch.cat([past[i][0], k], 2), torch.cat([past[i][1], v], 2)
            new.append((k, v))
            kk, vv = k.repeat_interleave(H // Hkv, 1), v.repeat_interleave(H // Hkv, 1)
            s = (q @ kk.transpose(-1, -2)) / math.sqrt(D)
            S = kk.shape[2]
            mask = torch.ones(T, S, dtype=torch.bool, device=self.device).tril(diagonal=S - T)
            pw = s.masked_fill(~mask, -float("inf")).float().softmax(-1).to(self.dtype)     # materialised [1, H, T, S]
            if i == lm.n_layers - 1:
                self.last_attn = pw                        # what HF hands back as attentions[step][-1] (llava_calibrate.py:180)
            att = pw @ vv
            h = h + att.transpose(1, 2).reshape(1, T, H * D) @ w[p + "wo"].t()
            a = rms(h, w[p + "ln2"])
            gu = a @ w[p + "wgu"].t()
            h = h + (F.silu(gu[..., : lm.ffn]) * gu[..., lm.ffn:]) @ w[p + "wd"].t()
        if lm.n_layers == 0:                    # depth-0 model (bench.py's fixed-cost measurement): keep a length-carrying dummy cache
            z = torch.zeros(1, 1, p0 + T, 1, device=self.device)
            new = [(z, z)]
        return rms(h, w["norm"]) @ w["lm_head"].t(), tuple(new)

    def __call__(self, input_ids=None, attention_mask=None, past_key_values=None, inputs_embeds=None, use_cache=None,
                 images=None, image_sizes=None, return_dict=True, output_attentions=None, output_hidden_states=None, **_):
        past_len = int(past_key_values[0][0].shape[-2]) if past_key_values else 0
        if inputs_embeds is not None:                           # LAVIS Llama (modeling_llama.py:764-792): prompt as embeddings
            self.calls.append((tuple(inputs_embeds.shape[:2]), False, past_len))
            emb = inputs_embeds.to(self.device, self.store).to(self.dtype)
            logits, past = self._lm(emb, past_key_values)
            return SimpleNamespace(logits=logits.to(self.logit_dtype) if self.logits_on_device else logits.to(self.logit_dtype).cpu(), past_key_values=past, attentions=None, hidden_states=None)
        ids = input_ids.to(self.device)
        self.calls.append((tuple(ids.shape), images is not None, past_len))
        if images is None or ids.shape[1] == 1:                 # llava_arch.py:91-94
            emb = self.w["embed"][ids]
        else:                                                   # llava_arch.py:122-163 (batch 1)
            emb = splice(self.w["embed"], ids[0], self.encode_images(images)[0])[None]
        emb = emb.to(self.store).to(self.dtype)
        logits, past = self._lm(emb, past_key_values)
        return SimpleNamespace(logits=logits.to(self.logit_dtype) if self.logits_on_device else logits.to(self.logit_dtype).cpu(), past_key_values=past, attentions=None, hidden_states=None)


class RefLavisLM(RefLlava):
    """The LM side of InstructBLIP as the reference drives it: generate(inputs_embeds=..., images_cd=inputs_embeds_cd)
    with prepare_inputs_for_generation[_cd] of experiments/lavis/models/blip2_models/modeling_llama.py:734-792."""

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kw):
        if past_key_values:
            input_ids = input_ids[:, -1:]
        if inputs_embeds is not None and past_key_values is None:
            d = {"inputs_embeds": inputs_embeds}
        else:
            d = {"input_ids": input_ids}
        d.update({"past_key_values": past_key_values, "use_cache": kw.get("use_cache"), "attention_mask": attention_mask})
        return d

    def prepare_inputs_for_generation_cd(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kw):
        if past_key_values:
            input_ids = input_ids[:, -1:]
        if inputs_embeds is not None and past_key_values is None:
            d = {"inputs_embeds": kw.get("images_cd")}             # modeling_llama.py:778-782
        else:
            d = {"input_ids": input_ids}
        d.update({"past_key_values": past_key_values, "use_cache": kw.get("use_cache"), "attention_mask": attention_mask})
        return d
