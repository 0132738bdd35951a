"""InstructBLIP front-end for the VCD branch (BASELINE config #5): builds `inputs_embeds` and `inputs_embeds_cd` - what the reference
hands to `llm_model.generate(inputs_embeds=..., images_cd=inputs_embeds_cd, ...)` - from an image, its noised copy and the prompt:

    image -> EVA-ViT-g -> ln_vision -> Q-Former (32 learned queries + the instruction text; cross-attention to the image every
    2nd layer) -> llm_proj -> 32 LLM-width embeddings, concatenated in front of the prompt's token embeddings
    (experiments/lavis/models/blip2_models/blip2_vicuna_instruct.py:333-388; the towers: lavis/models/eva_vit.py:64-342,
    blip2_models/Qformer.py:51-108,378-484, blip2.py:48-62,194-200).

Every op is a kernel of this package (GEMMs with fused bias / GELU / residual epilogues, LayerNorm, flash attention, the ViT
glue kernels); torch only holds the buffers and assembles inputs (slice copies).  EVA's 88-wide attention heads run on the
128-wide attention kernel: q/k/v projection rows and out-projection columns are zero-padded per head ONCE at weight load, so the
padded dimensions are exactly zero and scores / outputs are unchanged (scale stays 88^-0.5).
LAVIS itself (dataset builders, processors, registry, training) is out of scope; tokenisers stay with the caller.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch

from . import ops


@dataclass
class EvaVitConfig:
    image: int = 224
    patch: int = 14
    width: int = 1408
    layers: int = 39
    heads: int = 16
    head_dim: int = 88
    mlp: int = 6144              # int(1408 * 4.3637)
    eps: float = 1e-6
    ln_vision_eps: float = 1e-5

    @property
    def n_tokens(self):
        return (self.image // self.patch) ** 2 + 1


@dataclass
class QFormerConfig:
    hidden: int = 768
    layers: int = 12
    heads: int = 12
    inter: int = 3072
    n_query: int = 32
    cross_freq: int = 2
    vocab: int = 30523
    max_pos: int = 512
    eps: float = 1e-12


@dataclass
class BlipConfig:
    vit: EvaVitConfig = field(default_factory=EvaVitConfig)
    qf: QFormerConfig = field(default_factory=QFormerConfig)
    d_llm: int = 4096


def tiny_blip_config() -> BlipConfig:
    """Test-sized towers with the real structure: 88-wide ViT heads (padded to 128), cross-attention every 2nd Q-Former layer."""
    return BlipConfig(EvaVitConfig(image=56, width=256, layers=2, heads=2, head_dim=88, mlp=512),
                      QFormerConfig(hidden=128, layers=4, heads=2, inter=256, n_query=8, vocab=500, max_pos=64), d_llm=256)


PAD_D = 128                       # head width the attention kernel runs EVA's heads at


class BlipWeights:
    """Device tensors of one 16-bit dtype (bf16, or fp16 as LAVIS loads the checkpoint) under LAVIS's parameter names (`visual_encoder.*`, `ln_vision.*`, `Qformer.bert.*`, `query_tokens`,
    `llm_proj.*`) plus the derived, head-padded attention weights of the ViT (`vit{i}.wqkv/bqkv/wo`)."""

    def __init__(self, cfg: BlipConfig, device, dtype=torch.bfloat16):
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        self.t: Dict[str, torch.Tensor] = {}

    @staticmethod
    def random(cfg: BlipConfig, device, seed: int = 0, std: float = 0.02, dtype=torch.bfloat16) -> "BlipWeights":
        g = torch.Generator(device=device).manual_seed(seed)
        rnd = lambda *s, sc=std: torch.randn(*s, device=device, generator=g, dtype=torch.float32) * sc
        one = lambda n: 1.0 + torch.randn(n, device=device, generator=g) * 0.02
        v, q = cfg.vit, cfg.qf
        sd = {}
        ahd = v.heads * v.head_dim
        sd["visual_encoder.patch_embed.proj.weight"] = rnd(v.width, 3, v.patch, v.patch)
        sd["visual_encoder.patch_embed.proj.bias"] = rnd(v.width)
        sd["visual_encoder.cls_token"] = rnd(1, 1, v.width)
        sd["visual_encoder.pos_embed"] = rnd(1, v.n_tokens, v.width)
        for i in range(v.layers):
            p = f"visual_encoder.blocks.{i}."
            sd[p + "norm1.weight"], sd[p + "norm1.bias"] = one(v.width), rnd(v.width)
            sd[p + "norm2.weight"], sd[p + "norm2.bias"] = one(v.width), rnd(v.width)
            sd[p + "attn.qkv.weight"] = rnd(3 * ahd, v.width)
            sd[p + "attn.q_bias"], sd[p + "attn.v_bias"] = rnd(ahd), rnd(ahd)
            sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = rnd(v.width, ahd), rnd(v.width)
            sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = rnd(v.mlp, v.width), rnd(v.mlp)
            sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = rnd(v.width, v.mlp), rnd(v.width)
        sd["ln_vision.weight"], sd["ln_vision.bias"] = one(v.width), rnd(v.width)
        sd["query_tokens"] = rnd(1, q.n_query, q.hidden)
        e = "Qformer.bert.embeddings."
        sd[e + "word_embeddings.weight"], sd[e + "position_embeddings.weight"] = rnd(q.vocab, q.hidden), rnd(q.max_pos, q.hidden)
        sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"] = one(q.hidden), rnd(q.hidden)
        for i in range(q.layers):
            p = f"Qformer.bert.encoder.layer.{i}."
            blocks = [("attention", q.hidden)] + ([("crossattention", v.width)] if i % q.cross_freq == 0 else [])
            for name, kv_in in blocks:
                a = p + name + "."
                sd[a + "self.query.weight"], sd[a + "self.query.bias"] = rnd(q.hidden, q.hidden), rnd(q.hidden)
                sd[a + "self.key.weight"], sd[a + "self.key.bias"] = rnd(q.hidden, kv_in), rnd(q.hidden)
                sd[a + "self.value.weight"], sd[a + "self.value.bias"] = rnd(q.hidden, kv_in), rnd(q.hidden)
                sd[a + "output.dense.weight"], sd[a + "output.dense.bias"] = rnd(q.hidden, q.hidden), rnd(q.hidden)
                sd[a + "output.LayerNorm.weight"], sd[a + "output.LayerNorm.bias"] = one(q.hidden), rnd(q.hidden)
            for suf in ("", "_query"):
                sd[p + f"intermediate{suf}.dense.weight"], sd[p + f"intermediate{suf}.dense.bias"] = rnd(q.inter, q.hidden), rnd(q.inter)
                sd[p + f"output{suf}.dense.weight"], sd[p + f"output{suf}.dense.bias"] = rnd(q.hidden, q.inter), rnd(q.hidden)
                sd[p + f"output{suf}.LayerNorm.weight"], sd[p + f"output{suf}.LayerNorm.bias"] = one(q.hidden), rnd(q.hidden)
        sd["llm_proj.weight"], sd["llm_proj.bias"] = rnd(cfg.d_llm, q.hidden), rnd(cfg.d_llm)
        return BlipWeights.from_state_dict(cfg, sd, device, dtype)

    @staticmethod
    def from_state_dict(cfg: BlipConfig, sd: Dict[str, torch.Tensor], device, dtype=torch.bfloat16) -> "BlipWeights":
        w = BlipWeights(cfg, device, dtype)
        v = cfg.vit
        for k, t in sd.items():
            w.t[k] = t.detach().to(device=device, dtype=dtype).contiguous()
        # patch convolution as a [width, 3*P*P -> padded to 128] GEMM operand
        pw = w.t["visual_encoder.patch_embed.proj.weight"].reshape(v.width, -1)
        kp = (pw.shape[1] + 127) // 128 * 128
        w.t["vit.patch"] = torch.nn.functional.pad(pw, (0, kp - pw.shape[1])).contiguous()
        w.t["vit.cls"] = w.t["visual_encoder.cls_token"].reshape(-1).contiguous()
        w.t["vit.pos"] = w.t["visual_encoder.pos_embed"].reshape(-1, v.width).contiguous()
        H, hd = v.heads, v.head_dim
        for i in range(v.layers):
            p = f"visual_encoder.blocks.{i}.attn."
            qkv = w.t[p + "qkv.weight"].view(3, H, hd, v.width)
            wq = torch.zeros(3, H, PAD_D, v.width, dtype=dtype, device=device)
            wq[:, :, :hd] = qkv
            bq = torch.zeros(3, H, PAD_D, dtype=dtype, device=device)
            bq[0, :, :hd] = w.t[p + "q_bias"].view(H, hd)                  # qkv bias = (q_bias, 0, v_bias), eva_vit.py:125-126
            bq[2, :, :hd] = w.t[p + "v_bias"].view(H, hd)
            wo = torch.zeros(v.width, H, PAD_D, dtype=dtype, device=device)
            wo[:, :, :hd] = w.t[p + "proj.weight"].view(v.width, H, hd)
            w.t[f"vit{i}.wqkv"], w.t[f"vit{i}.bqkv"], w.t[f"vit{i}.wo"] = wq.view(3 * H * PAD_D, v.width), bq.view(-1), wo.view(v.width, H * PAD_D)
        q = cfg.qf
        for i in range(q.layers):                      # fused q/k/v of the self-attention; k/v of the cross-attention
            p = f"Qformer.bert.encoder.layer.{i}."
            a = p + "attention.self."
            w.t[f"qf{i}.wqkv"] = torch.cat([w.t[a + "query.weight"], w.t[a + "key.weight"], w.t[a + "value.weight"]], 0).contiguous()
            w.t[f"qf{i}.bqkv"] = torch.cat([w.t[a + "query.bias"], w.t[a + "key.bias"], w.t[a + "value.bias"]], 0).contiguous()
            if i % q.cross_freq == 0:
                c = p + "crossattention.self."
                w.t[f"qf{i}.wkv_x"] = torch.cat([w.t[c + "key.weight"], w.t[c + "value.weight"]], 0).contiguous()
                w.t[f"qf{i}.bkv_x"] = torch.cat([w.t[c + "key.bias"], w.t[c + "value.bias"]], 0).contiguous()
        w.t["qf.query_tokens"] = w.t["query_tokens"].reshape(q.n_query, q.hidden).contiguous()
        return w


class InstructBlipFrontEnd:
    def __init__(self, weights: BlipWeights):
        self.w, self.cfg = weights, weights.cfg
        self.device = weights.device
        self._cache: Dict[tuple, tuple] = {}

    def _kv(self, key, n, H, T, D):
        c = self._cache.get(key)
        if c is None or c[0].shape[0] < n or c[0].shape[2] < T:
            mk = lambda: torch.zeros(n, H, T, D, dtype=self.w.dtype, device=self.device)
            c = self._cache[key] = (mk(), mk())
        return c

    # ---- EVA ViT + ln_vision: eva_vit.py:318-342 (forward_features), blip2_vicuna_instruct.py:333
    @torch.no_grad()
    def image_embeds(self, images: torch.Tensor) -> torch.Tensor:
        v, t = self.cfg.vit, self.w.t
        n = images.shape[0]
        x = images.to(self.device)
        if x.dtype not in (torch.float32, torch.float16, torch.bfloat16):
            x = x.float()
        T, H = v.n_tokens, v.heads
        patches = ops.vit_im2col(x.contiguous(), v.patch, t["vit.patch"].shape[1], dtype=self.w.dtype)
        emb = ops.gemm(patches, t["vit.patch"], bias=t["visual_encoder.patch_embed.proj.bias"], epi=ops.EPI_BIAS)
        h = ops.vit_assemble(emb, t["vit.cls"], t["vit.pos"], n, T)
        kc, vc = self._kv("vit", n, H, T, PAD_D)
        seqs = torch.tensor([[i * T, T, 0, i, 0, 0] for i in range(n)], dtype=torch.int32, device=self.device)
        for i in range(v.layers):
            p = f"visual_encoder.blocks.{i}."
            a = ops.layernorm(h, t[p + "norm1.weight"], t[p + "norm1.bias"], v.eps)
            qkv = ops.gemm(a, t[f"vit{i}.wqkv"], bias=t[f"vit{i}.bqkv"], epi=ops.EPI_BIAS)
            q = ops.vit_qkv_split(qkv, kc, vc, n, T, H, PAD_D)
            att = ops.flash_attention(q, kc, vc, seqs, n, T, H, H, PAD_D, causal=False, scale=v.head_dim ** -0.5)
            h = ops.gemm(att, t[f"vit{i}.wo"], bias=t[p + "attn.proj.bias"], resid=h, epi=ops.EPI_BIAS_RESID)
            a = ops.layernorm(h, t[p + "norm2.weight"], t[p + "norm2.bias"], v.eps)
            f = ops.gemm(a, t[p + "mlp.fc1.weight"], bias=t[p + "mlp.fc1.bias"], epi=ops.EPI_BIAS_GELU)
            h = ops.gemm(f, t[p + "mlp.fc2.weight"], bias=t[p + "mlp.fc2.bias"], resid=h, epi=ops.EPI_BIAS_RESID)
        return ops.layernorm(h, t["ln_vision.weight"], t["ln_vision.bias"], v.ln_vision_eps).view(n, T, v.width)

    # ---- Q-Former: Qformer.py BertModel with query_embeds + text, cross-attention to the image every cross_freq layers
    @torch.no_grad()
    def qformer(self, image_embeds: torch.Tensor, text_ids: Optional[Sequence[Sequence[int]]]) -> torch.Tensor:
        """image_embeds [n, Ti, width]; text_ids: per sample the Q-Former tokenizer's ids of the instruction (unpadded), or None
        for qformer_text_input=False.  Returns the query outputs [n, n_query, hidden]."""
        q, t, dev = self.cfg.qf, self.w.t, self.device
        n, Ti, _ = image_embeds.shape
        NQ, Hd, H = q.n_query, q.hidden, q.heads
        D = Hd // H
        lens = [len(r) for r in text_ids] if text_ids is not None else [0] * n
        L = max(lens) if lens else 0
        e = "Qformer.bert.embeddings."
        hq = t["qf.query_tokens"].repeat(n, 1)                                     # [n*NQ, Hd]
        ht = None
        if L > 0:
            ids = torch.zeros(n, L, dtype=torch.long)
            for i, r in enumerate(text_ids):
                ids[i, : len(r)] = torch.tensor(list(r), dtype=torch.long)
            ids = ids.to(dev).view(-1)
            pos = torch.arange(L, device=dev).repeat(n)
            ht = ops.add(ops.embed(ids, t[e + "word_embeddings.weight"]), ops.embed(pos, t[e + "position_embeddings.weight"]))   # :95-99
            ht = ops.layernorm(ht, t[e + "LayerNorm.weight"], t[e + "LayerNorm.bias"], q.eps)
        hq = ops.layernorm(hq, t[e + "LayerNorm.weight"], t[e + "LayerNorm.bias"], q.eps)                                        # :106
        kc, vc = self._kv("qf_self", n, H, NQ + max(L, 1), D)
        kx, vx = self._kv("qf_cross", n, H, Ti, D)
        img2d = image_embeds.reshape(n * Ti, -1)
        # a sample's keys: its NQ queries then its text tokens (padding rows sit behind the valid ones and are never attended)
        seq_q = torch.tensor([[i * NQ, NQ, lens[i], i, 0, 0] for i in range(n)], dtype=torch.int32, device=dev)
        seq_t = torch.tensor([[i * L, lens[i], NQ, i, 0, 0] for i in range(n)], dtype=torch.int32, device=dev) if L > 0 else None
        seq_x = torch.tensor([[i * NQ, NQ, Ti - NQ, i, 0, 0] for i in range(n)], dtype=torch.int32, device=dev)
        assert Ti >= NQ, "cross-attention descriptor assumes at least as many image tokens as queries"
        for i in range(q.layers):
            p = f"Qformer.bert.encoder.layer.{i}."
            last = i == q.layers - 1
            # self-attention over [queries ; text]
            qq = ops.vit_qkv_split(ops.gemm(hq, t[f"qf{i}.wqkv"], bias=t[f"qf{i}.bqkv"], epi=ops.EPI_BIAS), kc, vc, n, NQ, H, D)
            if ht is not None:
                qt = ops.vit_qkv_split(ops.gemm(ht, t[f"qf{i}.wqkv"], bias=t[f"qf{i}.bqkv"], epi=ops.EPI_BIAS), kc[:, :, NQ:], vc[:, :, NQ:], n, L, H, D)
            aq = ops.flash_attention(qq, kc, vc, seq_q, n, NQ, H, H, D, causal=False)
            a = p + "attention.output."
            hq = ops.layernorm(ops.gemm(aq, t[a + "dense.weight"], bias=t[a + "dense.bias"], resid=hq, epi=ops.EPI_BIAS_RESID),
                               t[a + "LayerNorm.weight"], t[a + "LayerNorm.bias"], q.eps)
            if ht is not None and not last:                       # the text stream only feeds later layers' keys
                at = ops.flash_attention(qt, kc, vc, seq_t, n, L, H, H, D, causal=False)
                ht = ops.layernorm(ops.gemm(at, t[a + "dense.weight"], bias=t[a + "dense.bias"], resid=ht, epi=ops.EPI_BIAS_RESID),
                                   t[a + "LayerNorm.weight"], t[a + "LayerNorm.bias"], q.eps)
            if i % q.cross_freq == 0:                             # queries attend the image (Qformer.py:432-444)
                c = p + "crossattention."
                ops.vit_qkv_split(ops.gemm(img2d, t[f"qf{i}.wkv_x"], bias=t[f"qf{i}.bkv_x"], epi=ops.EPI_BIAS), kx, vx, n, Ti, H, D, kv_only=True)
                qx = ops.gemm(hq, t[c + "self.query.weight"], bias=t[c + "self.query.bias"], epi=ops.EPI_BIAS)
                ax = ops.flash_attention(qx, kx, vx, seq_x, n, NQ, H, H, D, causal=False)
                hq = ops.layernorm(ops.gemm(ax, t[c + "output.dense.weight"], bias=t[c + "output.dense.bias"], resid=hq, epi=ops.EPI_BIAS_RESID),
                                   t[c + "output.LayerNorm.weight"], t[c + "output.LayerNorm.bias"], q.eps)
            f = ops.gemm(hq, t[p + "intermediate_query.dense.weight"], bias=t[p + "intermediate_query.dense.bias"], epi=ops.EPI_BIAS_GELU)
            hq = ops.layernorm(ops.gemm(f, t[p + "output_query.dense.weight"], bias=t[p + "output_query.dense.bias"], resid=hq, epi=ops.EPI_BIAS_RESID),
                               t[p + "output_query.LayerNorm.weight"], t[p + "output_query.LayerNorm.bias"], q.eps)
            if ht is not None and not last:
                f = ops.gemm(ht, t[p + "intermediate.dense.weight"], bias=t[p + "intermediate.dense.bias"], epi=ops.EPI_BIAS_GELU)
                ht = ops.layernorm(ops.gemm(f, t[p + "output.dense.weight"], bias=t[p + "output.dense.bias"], resid=ht, epi=ops.EPI_BIAS_RESID),
                                   t[p + "output.LayerNorm.weight"], t[p + "output.LayerNorm.bias"], q.eps)
        return hq.view(n, NQ, Hd)

    @torch.no_grad()
    def embeds_to_llm(self, image_embeds: torch.Tensor, text_ids: Optional[Sequence[Sequence[int]]]) -> torch.Tensor:
        """ln_vision(ViT(image)) [n, Ti, width] -> Q-Former -> llm_proj -> [n, n_query, d_llm] (blip2_vicuna_instruct.py:339-366).  Split from
        image_embeds() so that a driver can run the ViT once per DISTINCT image (POPE asks 6 questions per image; the `zeros` prior image is
        the same for every question) while the Q-Former, which also reads the instruction, runs per question."""
        hq = self.qformer(image_embeds, text_ids)
        n, NQ, Hd = hq.shape
        return ops.gemm(hq.reshape(n * NQ, Hd), self.w.t["llm_proj.weight"], bias=self.w.t["llm_proj.bias"], epi=ops.EPI_BIAS).view(n, NQ, -1)

    @torch.no_grad()
    def inputs_llm(self, images: torch.Tensor, text_ids: Optional[Sequence[Sequence[int]]]) -> torch.Tensor:
        """image -> [n, n_query, d_llm] (blip2_vicuna_instruct.py:333-366)."""
        return self.embeds_to_llm(self.image_embeds(images), text_ids)

    @torch.no_grad()
    def assemble(self, llm_in: torch.Tensor, prompt_ids: Sequence[Sequence[int]], embed_table: torch.Tensor) -> List[torch.Tensor]:
        """[n, n_query, d_llm] ++ the LLM's token embeddings of each prompt (:377-388) -> per sample [n_query + len(prompt), d_llm]."""
        out = []
        for i in range(llm_in.shape[0]):
            tok = ops.embed(torch.tensor(list(prompt_ids[i]), dtype=torch.long, device=self.device), embed_table)
            out.append(torch.cat([llm_in[i], tok], 0))
        return out

    @torch.no_grad()
    def build(self, images: torch.Tensor, prompt_ids: Sequence[Sequence[int]], embed_table: torch.Tensor,
              qformer_text_ids: Optional[Sequence[Sequence[int]]] = None, images_cd: Optional[torch.Tensor] = None):
        """-> (inputs_embeds, inputs_embeds_cd): per sample [n_query + len(prompt), d_llm] = Q-Former output ++ the LLM's token
        embeddings of the prompt (:377-388); inputs_embeds_cd from the noised image, None without images_cd.  These go straight
        into VddLlavaEngine.generate(inputs_embeds=..., images_cd=...)."""
        n = images.shape[0]
        main = self.inputs_llm(images, qformer_text_ids)
        cd = self.inputs_llm(images_cd, qformer_text_ids) if images_cd is not None else None
        out, out_cd = [], []
        for i in range(n):
            tok = ops.embed(torch.tensor(list(prompt_ids[i]), dtype=torch.long, device=self.device), embed_table)
            out.append(torch.cat([main[i], tok], 0))
            if cd is not None:
                out_cd.append(torch.cat([cd[i], tok], 0))
        return out, (out_cd if cd is not None else None)
