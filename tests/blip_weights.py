"""Seeded InstructBLIP front-end weights under LAVIS's parameter names (test infrastructure).

The golden fixtures (tests/golden/blip_vectors.npz, made by tests/golden/make_blip_golden.py from the REAL LAVIS modules) store
outputs only; the weights and inputs are regenerated here from `numpy.random.RandomState` (a frozen stream: same numbers under
every numpy version, on the build container and on the GPU box).  Every tensor is pre-rounded to bf16 (returned as fp32), so the
LAVIS fp32 run, tests/ref_blip.py and the bf16 kernels all see the same parameter values.

Names follow experiments/lavis/models/eva_vit.py (visual_encoder.*), blip2_models/Qformer.py (Qformer.bert.*),
blip2_models/blip2_vicuna_instruct.py:58-110 (ln_vision, query_tokens, llm_proj)."""
import numpy as np
import torch


def _bf16(a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float()


def blip_state_dict(cfg, seed: int, std: float = 0.05):
    """cfg: llava_align_amd.blip_frontend.BlipConfig (or anything with the same .vit / .qf / .d_llm fields)."""
    rs = np.random.RandomState(seed)
    rnd = lambda *s, sc=std: _bf16(rs.standard_normal(s) * sc)
    lin = lambda o, i: _bf16(rs.standard_normal((o, i)) * (0.7 / np.sqrt(i)))     # fan-in scaled: activations stay O(1) at any width
    one = lambda n: _bf16(1.0 + rs.standard_normal((n,)) * 0.05)
    v, q = cfg.vit, cfg.qf
    ahd = v.heads * v.head_dim
    sd = {}
    sd["visual_encoder.patch_embed.proj.weight"] = rnd(v.width, 3, v.patch, v.patch, sc=0.7 / np.sqrt(3 * v.patch * v.patch))
    sd["visual_encoder.patch_embed.proj.bias"] = rnd(v.width)
    sd["visual_encoder.cls_token"] = rnd(1, 1, v.width)
    sd["visual_encoder.pos_embed"] = rnd(1, v.n_tokens, v.width)
    for i in range(v.layers):
        p = f"visual_encoder.blocks.{i}."
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = one(v.width), rnd(v.width)
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = one(v.width), rnd(v.width)
        sd[p + "attn.qkv.weight"] = lin(3 * ahd, v.width)
        sd[p + "attn.q_bias"], sd[p + "attn.v_bias"] = rnd(ahd), rnd(ahd)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = lin(v.width, ahd), rnd(v.width)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = lin(v.mlp, v.width), rnd(v.mlp)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = lin(v.width, v.mlp), rnd(v.width)
    sd["ln_vision.weight"], sd["ln_vision.bias"] = one(v.width), rnd(v.width)
    sd["query_tokens"] = rnd(1, q.n_query, q.hidden, sc=0.5)
    e = "Qformer.bert.embeddings."
    sd[e + "word_embeddings.weight"] = rnd(q.vocab, q.hidden, sc=0.5)
    sd[e + "position_embeddings.weight"] = rnd(q.max_pos, q.hidden, sc=0.5)
    sd[e + "LayerNorm.weight"], sd[e + "LayerNorm.bias"] = one(q.hidden), rnd(q.hidden)
    for i in range(q.layers):
        p = f"Qformer.bert.encoder.layer.{i}."
        blocks = [("attention", q.hidden)] + ([("crossattention", v.width)] if i % q.cross_freq == 0 else [])
        for name, kv_in in blocks:
            a = p + name + "."
            sd[a + "self.query.weight"], sd[a + "self.query.bias"] = lin(q.hidden, q.hidden), rnd(q.hidden)
            sd[a + "self.key.weight"], sd[a + "self.key.bias"] = lin(q.hidden, kv_in), rnd(q.hidden)
            sd[a + "self.value.weight"], sd[a + "self.value.bias"] = lin(q.hidden, kv_in), rnd(q.hidden)
            sd[a + "output.dense.weight"], sd[a + "output.dense.bias"] = lin(q.hidden, q.hidden), rnd(q.hidden)
            sd[a + "output.LayerNorm.weight"], sd[a + "output.LayerNorm.bias"] = one(q.hidden), rnd(q.hidden)
        for suf in ("", "_query"):
            sd[p + f"intermediate{suf}.dense.weight"], sd[p + f"intermediate{suf}.dense.bias"] = lin(q.inter, q.hidden), rnd(q.inter)
            sd[p + f"output{suf}.dense.weight"], sd[p + f"output{suf}.dense.bias"] = lin(q.hidden, q.inter), rnd(q.hidden)
            sd[p + f"output{suf}.LayerNorm.weight"], sd[p + f"output{suf}.LayerNorm.bias"] = one(q.hidden), rnd(q.hidden)
    sd["llm_proj.weight"], sd["llm_proj.bias"] = lin(cfg.d_llm, q.hidden), rnd(cfg.d_llm)
    return sd


def blip_inputs(cfg, seed: int, n: int):
    """Seeded images [n, 3, S, S] (bf16-representable fp32) and ragged Q-Former instruction ids (101 ... 102, as the BERT tokenizer
    frames them)."""
    rs = np.random.RandomState(seed)
    imgs = _bf16(rs.standard_normal((n, 3, cfg.vit.image, cfg.vit.image)))
    text = []
    for i in range(n):
        L = int(rs.randint(0, 9)) if i else 6
        text.append([101] + [int(t) for t in rs.randint(103, cfg.qf.vocab, size=L)] + [102])
    return imgs, text


# The cases of the fixture file: name -> (config factory, weight seed, input seed, batch).  `real_widths` keeps the published
# EVA-ViT-g / Q-Former widths (1408 = 16 x 88, MLP 6144, 257 tokens at 224 px; 768 = 12 x 64, FFN 3072, 32 queries, vocabulary
# 30523) with fewer layers; `tiny` is the GPU smoke size with the same structure (88-wide heads, cross-attention every 2nd layer).
def cases():
    from llava_align_amd.blip_frontend import BlipConfig, EvaVitConfig, QFormerConfig, tiny_blip_config
    real = lambda: BlipConfig(EvaVitConfig(layers=2), QFormerConfig(layers=3), d_llm=4096)
    return {"tiny": (tiny_blip_config, 11, 12, 3), "real_widths": (real, 21, 22, 2)}
