"""Plain-PyTorch fp32 restatement of the InstructBLIP front-end as LAVIS computes it (test infrastructure only; PINNED to outputs
of the reference's own LAVIS modules by tests/test_blip_golden.py / tests/golden/make_blip_golden.py):
EVA-ViT (lavis/models/eva_vit.py:64-342: pre-LN blocks, qkv bias = (q_bias, 0, v_bias), absolute position embedding, no final norm)
-> ln_vision -> Q-Former BertModel with query_embeds + text (blip2_models/Qformer.py:51-108 embeddings, :378-484 layers: joint
self-attention over [queries ; text] under the padding mask, cross-attention of the queries to the image every cross_freq
layers, separate FFNs for queries and text, post-LN) -> llm_proj (blip2_vicuna_instruct.py:333-366)."""
import math

import torch
import torch.nn.functional as F


def eva_vit(sd, cfg, images):
    v = cfg.vit
    w = lambda k: sd[k].float()
    x = images.to(torch.bfloat16).float()
    n = x.shape[0]
    h = F.conv2d(x, w("visual_encoder.patch_embed.proj.weight"), w("visual_encoder.patch_embed.proj.bias"), stride=v.patch).flatten(2).transpose(1, 2)
    h = torch.cat([w("visual_encoder.cls_token").expand(n, -1, -1), h], 1) + w("visual_encoder.pos_embed")
    H, hd = v.heads, v.head_dim
    for i in range(v.layers):
        p = f"visual_encoder.blocks.{i}."
        a = F.layer_norm(h, (v.width,), w(p + "norm1.weight"), w(p + "norm1.bias"), v.eps)
        bias = torch.cat([w(p + "attn.q_bias"), torch.zeros_like(w(p + "attn.v_bias")), w(p + "attn.v_bias")])
        qkv = F.linear(a, w(p + "attn.qkv.weight"), bias).reshape(n, -1, 3, H, hd).permute(2, 0, 3, 1, 4)
        att = ((qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1)).softmax(-1) @ qkv[2]
        h = h + F.linear(att.transpose(1, 2).reshape(n, -1, H * hd), w(p + "attn.proj.weight"), w(p + "attn.proj.bias"))
        a = F.layer_norm(h, (v.width,), w(p + "norm2.weight"), w(p + "norm2.bias"), v.eps)
        h = h + F.linear(F.gelu(F.linear(a, w(p + "mlp.fc1.weight"), w(p + "mlp.fc1.bias"))), w(p + "mlp.fc2.weight"), w(p + "mlp.fc2.bias"))
    return F.layer_norm(h, (v.width,), w("ln_vision.weight"), w("ln_vision.bias"), v.ln_vision_eps)


def _attn(sd, pre, x_q, x_kv, mask, H):
    w = lambda k: sd[pre + k].float()
    B, Tq, Hd = x_q.shape
    D = Hd // H
    sp = lambda t: t.view(B, -1, H, D).transpose(1, 2)
    q, k, v = sp(F.linear(x_q, w("self.query.weight"), w("self.query.bias"))), sp(F.linear(x_kv, w("self.key.weight"), w("self.key.bias"))), \
        sp(F.linear(x_kv, w("self.value.weight"), w("self.value.bias")))
    s = q @ k.transpose(-1, -2) / math.sqrt(D)
    if mask is not None:
        s = s + (1.0 - mask[:, None, None, :].float()) * -10000.0          # BertModel.get_extended_attention_mask
    ctx = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, Tq, Hd)
    return ctx


def qformer(sd, cfg, image_embeds, text_ids):
    q = cfg.qf
    w = lambda k: sd[k].float()
    n = image_embeds.shape[0]
    if text_ids is None:                                  # qformer_text_input=False (blip2_vicuna_instruct.py:358-363): queries only
        text_ids = [[] for _ in range(n)]
    lens = [len(r) for r in text_ids]
    L = max(lens)
    ids = torch.zeros(n, L, dtype=torch.long, device=image_embeds.device)
    mask = torch.zeros(n, q.n_query + L, device=image_embeds.device)
    mask[:, : q.n_query] = 1
    for i, r in enumerate(text_ids):
        ids[i, : len(r)] = torch.tensor(list(r), dtype=torch.long)
        mask[i, q.n_query: q.n_query + len(r)] = 1
    e = "Qformer.bert.embeddings."
    emb = w(e + "word_embeddings.weight")[ids] + w(e + "position_embeddings.weight")[torch.arange(L, device=ids.device)][None]
    h = torch.cat([w("query_tokens").expand(n, -1, -1), emb], 1)
    h = F.layer_norm(h, (q.hidden,), w(e + "LayerNorm.weight"), w(e + "LayerNorm.bias"), q.eps)
    NQ = q.n_query

    def out_ln(pre, x, resid):
        return F.layer_norm(F.linear(x, w(pre + "dense.weight"), w(pre + "dense.bias")) + resid, (q.hidden,), w(pre + "LayerNorm.weight"),
                            w(pre + "LayerNorm.bias"), q.eps)
    for i in range(q.layers):
        p = f"Qformer.bert.encoder.layer.{i}."
        a = out_ln(p + "attention.output.", _attn(sd, p + "attention.", h, h, mask, q.heads), h)
        qa = a[:, :NQ]
        if i % q.cross_freq == 0:
            qa = out_ln(p + "crossattention.output.", _attn(sd, p + "crossattention.", qa, image_embeds, None, q.heads), qa)
        ffn = lambda x, suf: out_ln(p + f"output{suf}.", F.gelu(F.linear(x, w(p + f"intermediate{suf}.dense.weight"), w(p + f"intermediate{suf}.dense.bias"))), x)
        h = torch.cat([ffn(qa, "_query"), ffn(a[:, NQ:], "")], 1)
    return h[:, :NQ]


def inputs_llm(sd, cfg, images, text_ids):
    hq = qformer(sd, cfg, eva_vit(sd, cfg, images), text_ids)
    return F.linear(hq, sd["llm_proj.weight"].float(), sd["llm_proj.bias"].float())
