"""Build-container-only: runs the REAL LAVIS modules of the reference (read-only import by file path from /root/reference,
nothing copied) on seeded weights / inputs and commits their OUTPUTS as tests/golden/blip_vectors.npz.

    python tests/golden/make_blip_golden.py

What is executed unmodified:
  experiments/lavis/models/eva_vit.py         VisionTransformer (+ Block / Attention / Mlp / PatchEmbed), forward_features
  experiments/lavis/models/blip2_models/Qformer.py   BertModel with query_embeds + text + cross-attention (Qformer.bert)
and, composed exactly as blip2_vicuna_instruct.py:333-366 composes them: ln_vision(visual_encoder(image)) -> Qformer.bert(
text ids, attention_mask = [query_atts ; text mask], query_embeds = query_tokens, encoder_hidden_states = image_embeds) ->
llm_proj(last_hidden_state[:, :n_query]).

Import shims (runtime only): `timm.models.layers` (drop_path / to_2tuple / trunc_normal_) and `timm.models.registry`
(register_model) - timm is not installed; `lavis.common.dist_utils.download_cached_file` (never called: no checkpoint is
downloaded); three helper names that moved out of `transformers.modeling_utils` after the reference's transformers era.
Weights and inputs come from tests/blip_weights.py (numpy RandomState: regenerated identically wherever the tests run)."""
import importlib.util
import os
import sys
import types
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.dont_write_bytecode = True
REF = os.path.join(os.environ.get("VDD_REFERENCE_ROOT", "/root/reference"), "experiments/lavis/models")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_lavis():
    import transformers  # noqa: F401  (must be imported BEFORE the timm stub exists: it probes for a real timm)
    import transformers.modeling_utils as MU
    import transformers.pytorch_utils as PU
    for n in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(MU, n):
            setattr(MU, n, getattr(PU, n))
    if not hasattr(MU, "find_pruneable_heads_and_indices"):           # only used by prune_heads(), which nothing here calls
        MU.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError)
    stubs = {}
    tl = types.ModuleType("timm.models.layers")
    tl.drop_path = lambda x, p=0.0, training=False: x                # eval mode, drop_path_rate 0
    tl.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
    tl.trunc_normal_ = lambda t, std=1.0, **kw: torch.nn.init.trunc_normal_(t, std=std)
    tr = types.ModuleType("timm.models.registry")
    tr.register_model = lambda f: f
    ld = types.ModuleType("lavis.common.dist_utils")
    ld.download_cached_file = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("no network in this container"))
    for n, m in (("timm", types.ModuleType("timm")), ("timm.models", types.ModuleType("timm.models")), ("timm.models.layers", tl),
                 ("timm.models.registry", tr), ("lavis", types.ModuleType("lavis")), ("lavis.common", types.ModuleType("lavis.common")),
                 ("lavis.common.dist_utils", ld)):
        stubs[n] = sys.modules.get(n)
        sys.modules[n] = m
    try:
        eva = _load("ref_eva_vit", os.path.join(REF, "eva_vit.py"))
        qf = _load("ref_qformer", os.path.join(REF, "blip2_models/Qformer.py"))
    finally:
        for n, old in stubs.items():
            if old is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = old
    return eva, qf


def build_reference(eva, qf, cfg, sd):
    """The reference's modules at `cfg`'s sizes, loaded with `sd`."""
    from transformers.models.bert.configuration_bert import BertConfig
    v, q = cfg.vit, cfg.qf
    # create_eva_vit_g (eva_vit.py:427-440) with the widths of cfg; blocks whose heads are not width / heads wide (the tiny test
    # config keeps EVA's 88-wide heads at width 256) get the reference's own Attention(attn_head_dim=...)
    vit = eva.VisionTransformer(img_size=v.image, patch_size=v.patch, use_mean_pooling=False, embed_dim=v.width, depth=v.layers,
                                num_heads=v.heads, mlp_ratio=v.mlp / v.width, qkv_bias=True, drop_path_rate=0.0,
                                norm_layer=partial(torch.nn.LayerNorm, eps=v.eps), use_checkpoint=False)
    assert vit.blocks[0].mlp.fc1.out_features == v.mlp
    if v.width // v.heads != v.head_dim:
        for blk in vit.blocks:
            blk.attn = eva.Attention(v.width, num_heads=v.heads, qkv_bias=True, attn_head_dim=v.head_dim)
    vit.load_state_dict({k[len("visual_encoder."):]: t for k, t in sd.items() if k.startswith("visual_encoder.")}, strict=True)
    ln_vision = torch.nn.LayerNorm(v.width, eps=v.ln_vision_eps)
    ln_vision.load_state_dict({"weight": sd["ln_vision.weight"], "bias": sd["ln_vision.bias"]})
    # init_Qformer (blip2.py:48-62): bert-base-uncased config + encoder_width / add_cross_attention / cross_attention_freq / query_length
    bc = BertConfig(vocab_size=q.vocab, hidden_size=q.hidden, num_hidden_layers=q.layers, num_attention_heads=q.heads,
                    intermediate_size=q.inter, max_position_embeddings=q.max_pos, layer_norm_eps=q.eps, hidden_act="gelu",
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pad_token_id=0)
    bc.encoder_width, bc.add_cross_attention, bc.cross_attention_freq, bc.query_length = v.width, True, q.cross_freq, q.n_query
    for name, default in (("chunk_size_feed_forward", 0), ("position_embedding_type", "absolute"), ("is_decoder", False),
                          ("output_attentions", False), ("output_hidden_states", False), ("use_return_dict", True)):
        if not hasattr(bc, name):
            setattr(bc, name, default)
    # transformers 5.x's PreTrainedModel.init_weights() wants post_init() bookkeeping the reference's era did not have; random
    # initialisation is irrelevant here (every parameter is loaded below), so it is skipped - forward() is untouched
    qf.BertModel.init_weights = lambda self: None
    if not hasattr(qf.BertModel, "get_head_mask"):        # removed from PreTrainedModel in 5.x; head_mask=None meant "no mask per layer"
        qf.BertModel.get_head_mask = lambda self, head_mask, n_layers, *a, **k: [None] * n_layers
    bert = qf.BertModel(bc, add_pooling_layer=False)
    bsd = {k[len("Qformer.bert."):]: t for k, t in sd.items() if k.startswith("Qformer.bert.")}
    missing, unexpected = bert.load_state_dict(bsd, strict=False)
    assert not unexpected and all(m.endswith("position_ids") for m in missing), (missing, unexpected)
    llm_proj = torch.nn.Linear(q.hidden, cfg.d_llm)
    llm_proj.load_state_dict({"weight": sd["llm_proj.weight"], "bias": sd["llm_proj.bias"]})
    for m in (vit, ln_vision, bert, llm_proj):
        m.eval()
    return vit, ln_vision, bert, llm_proj


@torch.no_grad()
def run_reference(mods, cfg, sd, imgs, text):
    vit, ln_vision, bert, llm_proj = mods
    n = imgs.shape[0]
    image_embeds = ln_vision(vit(imgs))                                                     # blip2_vicuna_instruct.py:331
    image_atts = torch.ones(image_embeds.size()[:-1], dtype=torch.long)
    query_tokens = sd["query_tokens"].expand(n, -1, -1)                                     # :312
    L = max(len(r) for r in text)
    ids = torch.zeros(n, L, dtype=torch.long)
    tmask = torch.zeros(n, L, dtype=torch.long)
    for i, r in enumerate(text):                                                            # tokenizer(padding='longest'): right-padded
        ids[i, : len(r)] = torch.tensor(r)
        tmask[i, : len(r)] = 1
    query_atts = torch.ones(query_tokens.size()[:-1], dtype=torch.long)
    atts = torch.cat([query_atts, tmask], dim=1)                                            # :322-323
    out = bert(ids, attention_mask=atts, query_embeds=query_tokens, encoder_hidden_states=image_embeds,
               encoder_attention_mask=image_atts, return_dict=True)                         # :340-347
    hq = out.last_hidden_state[:, : query_tokens.size(1), :]
    out_nt = bert(query_embeds=query_tokens, encoder_hidden_states=image_embeds, encoder_attention_mask=image_atts,
                  return_dict=True)                                                         # qformer_text_input=False, :358-363
    return dict(image_embeds=image_embeds, query_out=hq, inputs_llm=llm_proj(hq), query_out_notext=out_nt.last_hidden_state)


ROWS = (0, 1, -1)          # image-token rows stored in full (class token, first patch, last patch); all rows as feature sums


def main():
    from blip_weights import blip_inputs, blip_state_dict, cases
    eva, qf = load_lavis()
    out = {}
    for name, (mk, wseed, iseed, n) in cases().items():
        cfg = mk()
        sd = blip_state_dict(cfg, wseed)
        imgs, text = blip_inputs(cfg, iseed, n)
        r = run_reference(build_reference(eva, qf, cfg, sd), cfg, sd, imgs, text)
        ie = r["image_embeds"]
        out[f"{name}.image_embeds_rows"] = ie[:, list(ROWS)].numpy()
        out[f"{name}.image_embeds_rowsum"] = ie.sum(-1).numpy()
        out[f"{name}.image_embeds_rowabs"] = ie.abs().sum(-1).numpy()
        out[f"{name}.query_out"] = r["query_out"].numpy()
        out[f"{name}.query_out_notext"] = r["query_out_notext"].numpy()
        il = r["inputs_llm"]
        out[f"{name}.inputs_llm_head"] = il[:, :, :64].numpy()               # first 64 of d_llm columns in full, all as sums
        out[f"{name}.inputs_llm_rowsum"] = il.sum(-1).numpy()
        out[f"{name}.inputs_llm_rowabs"] = il.abs().sum(-1).numpy()
        print(name, {k: tuple(t.shape) for k, t in r.items()}, "text lens", [len(t) for t in text],
              "|query_out|", float(r["query_out"].abs().mean()))
    path = os.path.join(HERE, "blip_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
