"""The native engine behind the object the reference's eval scripts actually call.

`experiments/eval/calibrate/llava_calibrate.py:120` builds a `LlavaLlamaForCausalLM` through `load_pretrained_model`
(`experiments/llava/model/builder.py:26-148`: fp16 weights, CLIP tower moved to cuda / fp16 at :137-141) and calls ITS
`generate` once per question (`llava_calibrate.py:161-177`).  `attach_engine(model)` keeps that object and that call:

    tokenizer, model, image_processor, _ = load_pretrained_model(...)     # unchanged
    evolve_vcd_sampling()                                                 # unchanged (the generic loop stays installed)
    attach_engine(model)                                                  # + this line
    out = model.generate(input_ids, images=img.unsqueeze(0).half().cuda(), use_dd_unk=True, cd_alpha=1, cd_beta=0.1,
                         do_sample=True, temperature=0.2, max_new_tokens=64, output_scores=True, return_dict_in_generate=True)
    out['sequences'], out['scores'][0]                                    # as llava_calibrate.py:178-179 reads them

It reads the architecture from `model.config` / the vision tower's config, maps the LIVE parameters into the engine's layout
(`LlavaWeights.from_state_dict`: zero-copy for everything stored as the kernels read it; q/k/v and gate/up are fused into one
tensor each and, with `share_storage`, the HF modules are re-pointed at views of the fused tensors, so the model does not grow),
builds a `VddLlavaEngine` in the MODEL'S dtype (fp16 for every released driver) and routes `model.generate(...)` to
`engine.generate(...)`, resolving defaults the way HF's `generate()` does (explicit kwargs, `None` included, win over
`model.generation_config`).  The object returned for `return_dict_in_generate=True` reads like HF's `GenerateDecoderOnlyOutput`:
`['sequences']` [B, L + new] (prompt ids first, -200 kept, as HF's sample() returns them: vcd_sample.py:262,304-321) and `['scores']`.
"""
from __future__ import annotations

import types
import warnings
from typing import Optional

import torch

from .engine import IMAGE_TOKEN_INDEX, LlavaConfig, LlavaWeights, LMConfig, VddLlavaEngine, VisionConfig

_NO_MAPS = ("{}: not produced for this call - the native engine's flash-style kernels never materialise the [heads, T, T] maps; what the "
            "reference's driver reads (llava_calibrate.py:180-182: model_outputs['attentions'][0][-1], step 0 / last layer) is computed on "
            "request for ONE question per call with output_attentions=True; hidden states and the maps of a batch are not - "
            "detach_engine(model) gives HF's generate (with the generic evolve_vcd_sampling() loop) back")


def _rope_theta(c) -> float:
    th = getattr(c, "rope_theta", None)
    if th is None:                                    # transformers >= 5: rope_parameters = {"rope_theta": ..., "rope_type": ...}
        rp = getattr(c, "rope_parameters", None) or {}
        if rp.get("rope_type", "default") not in ("default", None):
            raise ValueError(f"rope_type {rp.get('rope_type')!r}: the engine implements the plain rotary embedding of LLaVA-1.5 / Vicuna")
        th = rp.get("rope_theta", 10000.0)
    return float(th)


def config_from_hf(model) -> LlavaConfig:
    """LlavaConfig of a loaded `LlavaLlamaForCausalLM`-shaped module (llava_llama.py:46-56; tower: clip_encoder.py:8-37;
    projector: multimodal_projector/builder.py:33-46).  Raises ValueError for anything the kernels are not built for."""
    c = model.config
    heads = int(c.num_attention_heads)
    head_dim = int(getattr(c, "head_dim", None) or c.hidden_size // heads)
    if head_dim != 128:
        raise ValueError(f"head_dim {head_dim}: the attention kernels are instantiated for 128 (Llama / Vicuna 7B, 13B)")
    if getattr(c, "attention_bias", False) or getattr(c, "mlp_bias", False):
        raise ValueError("attention_bias / mlp_bias: not a LLaVA-1.5 language model")
    lm = LMConfig(d=int(c.hidden_size), n_layers=int(c.num_hidden_layers), n_heads=heads,
                  n_kv_heads=int(getattr(c, "num_key_value_heads", None) or heads), head_dim=head_dim, ffn=int(c.intermediate_size),
                  vocab=int(model.lm_head.weight.shape[0]), rope_theta=_rope_theta(c), eps=float(c.rms_norm_eps),
                  max_pos=int(c.max_position_embeddings))
    tower = model.get_vision_tower() if hasattr(model, "get_vision_tower") else None
    clip = getattr(tower, "vision_tower", None)
    if clip is None:
        raise ValueError("attach_engine needs the loaded CLIP tower (model.get_vision_tower().vision_tower; call load_model() first)")
    if getattr(tower, "select_feature", "patch") != "patch":
        raise ValueError("mm_vision_select_feature != 'patch': LLaVA-1.5 drops the class token (clip_encoder.py:33-37)")
    vc = clip.config
    vc = getattr(vc, "vision_config", vc)
    if getattr(vc, "hidden_act", "quick_gelu") != "quick_gelu":
        raise ValueError(f"CLIP hidden_act {vc.hidden_act!r}: the ViT MLP epilogue is quick_gelu")
    if vc.hidden_size // vc.num_attention_heads != 64:
        raise ValueError("the ViT attention kernel is instantiated for 64-wide heads (CLIP ViT-L/14)")
    ptype = getattr(c, "mm_projector_type", "mlp2x_gelu")
    if ptype != "mlp2x_gelu":
        raise ValueError(f"mm_projector_type {ptype!r}: LLaVA-1.5 uses mlp2x_gelu")
    vis = VisionConfig(image=int(vc.image_size), patch=int(vc.patch_size), width=int(vc.hidden_size), layers=int(vc.num_hidden_layers),
                       select_layer=int(getattr(tower, "select_layer", getattr(c, "mm_vision_select_layer", -2))),
                       heads=int(vc.num_attention_heads), mlp=int(vc.intermediate_size), eps=float(vc.layer_norm_eps))
    return LlavaConfig(lm, vis, getattr(c, "_name_or_path", "") or "hf-llava")


def weights_from_hf(model, cfg: Optional[LlavaConfig] = None, share_storage: bool = True) -> LlavaWeights:
    """The live parameters of `model` in the engine's layout, in the model's dtype, on the model's device.  Tensors the kernels
    read as HF stores them ([N, K] row-major: embeddings, o / down projections, lm_head, norms, the ViT's out / fc weights) are
    the SAME storage; q/k/v and gate/up are concatenated once, and with share_storage the HF parameters become views of the
    concatenated tensors (HF's eager forward keeps working on them; the model's memory does not grow by a second copy).
    SIDE EFFECT of share_storage: those parameters are then non-contiguous slices of one storage per layer - `save_pretrained` /
    safetensors refuse such tensors.  `detach_engine(model)` gives every re-pointed parameter its own contiguous storage back
    (`model._vdd_shared` lists them); `share_storage=False` never touches the model and costs one extra copy of q/k/v and gate/up."""
    cfg = cfg if cfg is not None else config_from_hf(model)
    p0 = model.lm_head.weight
    if not p0.is_cuda or p0.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"attach_engine needs the model on the GPU in fp16 or bf16 (got {p0.dtype} on {p0.device}): "
                         f"builder.py:40 loads fp16, llava_calibrate.py:163 moves it to cuda")
    sd = model.state_dict()                                           # references, not copies
    w = LlavaWeights.from_state_dict(cfg, sd, p0.device, dtype=p0.dtype)
    if share_storage:
        lm, v = cfg.lm, cfg.vision
        shared = model.__dict__.setdefault("_vdd_shared", [])
        nq, nkv = lm.n_heads * lm.head_dim, lm.n_kv_heads * lm.head_dim
        for i, layer in enumerate(model.model.layers):
            a, m = layer.self_attn, layer.mlp
            qkv, gu = w.t[f"l{i}.wqkv"], w.t[f"l{i}.wgu"]
            a.q_proj.weight.data, a.k_proj.weight.data, a.v_proj.weight.data = qkv[:nq], qkv[nq:nq + nkv], qkv[nq + nkv:]
            m.gate_proj.weight.data, m.up_proj.weight.data = gu[:lm.ffn], gu[lm.ffn:]
            shared += [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, m.gate_proj.weight, m.up_proj.weight]
        enc = model.get_vision_tower().vision_tower
        enc = getattr(enc, "vision_model", enc)
        for i in range(v.run_layers):
            a = enc.encoder.layers[i].self_attn
            qkv, b = w.t[f"v{i}.wqkv"], w.t[f"v{i}.bqkv"]
            W = v.width
            a.q_proj.weight.data, a.k_proj.weight.data, a.v_proj.weight.data = qkv[:W], qkv[W:2 * W], qkv[2 * W:]
            a.q_proj.bias.data, a.k_proj.bias.data, a.v_proj.bias.data = b[:W], b[W:2 * W], b[2 * W:]
            shared += [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.q_proj.bias, a.k_proj.bias, a.v_proj.bias]
    return w


class NativeGenerateOutput(dict):
    """What `model.generate(..., return_dict_in_generate=True)` returns on the native path: `['sequences']` / `.sequences`
    [B, L + new] int64 (prompt first, -200 kept) and `['scores']` (tuple of [B, V] post-warp rows, when output_scores) like HF's
    GenerateDecoderOnlyOutput; plus the engine's extras (`tokens`, `top_prob`, `top_tok`, `stats`)."""

    def __getitem__(self, k):
        if k in ("attentions", "hidden_states") and k not in self:
            raise KeyError(_NO_MAPS.format(k))
        return super().__getitem__(k)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(*e.args) from None


_UNSET = object()


def _resolve_generate_kwargs(gc, kw):
    """`GenerationMixin.generate`'s keyword resolution for what the reference's drivers pass: explicit keywords (None included:
    llava_calibrate.py:170-171 pass top_p=None, top_k=None) override `generation_config`, as `generation_config.update(**kwargs)` does in
    HF.  Consumes the resolved names from `kw`; returns (engine kwargs, return_dict_in_generate, output_attentions)."""
    def opt(name, default=None):
        v = kw.pop(name, _UNSET)
        return getattr(gc, name, default) if v is _UNSET else v
    am = kw.pop("attention_mask", None)
    if am is not None and not bool(torch.as_tensor(am).ne(0).all()):
        raise ValueError("attention_mask with zeros (a padded batch): pass each question's own ids - the engine batches ragged prompts "
                         "itself, and the reference's drivers call generate() with one un-padded question (llava_calibrate.py:130-177)")
    args = dict(do_sample=bool(opt("do_sample", False)), temperature=opt("temperature", 1.0), top_p=opt("top_p", 1.0), top_k=opt("top_k", 50),
                repetition_penalty=opt("repetition_penalty", None), min_new_tokens=opt("min_new_tokens", None),
                min_length=opt("min_length", None), num_beams=opt("num_beams", 1), num_return_sequences=opt("num_return_sequences", 1),
                output_scores=bool(opt("output_scores", False)))
    if "top_k" in args and args["top_k"] == 0:
        args["top_k"] = None                                          # HF: top_k = 0 disables the warper
    max_new = opt("max_new_tokens", None)
    if max_new is not None:
        args["max_new_tokens"] = int(max_new)
    else:
        ml = opt("max_length", None)
        args["max_length"] = int(ml) if ml is not None else 20       # HF's default when nothing is given
    eos, pad = opt("eos_token_id", None), opt("pad_token_id", None)
    if eos is not None and pad is None:                               # HF: "Setting `pad_token_id` to `eos_token_id`" (generate(), utils.py [ext])
        pad = eos[0] if isinstance(eos, (list, tuple)) else int(eos)
        warnings.warn(f"Setting `pad_token_id` to `eos_token_id`:{pad} for open-end generation.")
    args.update(eos_token_id=eos, pad_token_id=pad)
    return_dict = bool(opt("return_dict_in_generate", False))
    want_attn = bool(opt("output_attentions", False))
    for k in ("output_hidden_states", "use_cache", "synced_gpus", "length_penalty"):     # accepted without effect (length_penalty: beam search only)
        opt(k, None)
    return args, return_dict, want_attn


def _set_guard(model) -> None:
    """Remember where the model's lm_head lives (and its in-place version) at attach time.  The engine may hold a COPY of it (a tied or
    non-contiguous lm_head goes through .contiguous()), so comparing against the engine's tensor would raise on every call; what has to be
    noticed is the MODEL's parameter being re-allocated (model.to() / .half() / resize_token_embeddings) or overwritten in place."""
    w = model.lm_head.weight
    model._vdd_guard = (w.data_ptr(), w._version, tuple(w.shape), w.dtype)


def _check_guard(model, message: str) -> None:
    w = model.lm_head.weight
    if getattr(model, "_vdd_guard", None) != (w.data_ptr(), w._version, tuple(w.shape), w.dtype):
        raise RuntimeError(message)


def _native_generate(model, inputs=None, generation_config=None, **kw):
    """`GenerationMixin.generate`'s argument handling for the keywords the reference's drivers use, in front of
    `VddLlavaEngine.generate`.  Explicit keywords (None included: llava_calibrate.py:170-171 pass top_p=None, top_k=None) override
    `model.generation_config`, as `generation_config.update(**kwargs)` does in HF."""
    eng: VddLlavaEngine = model._vdd_engine
    _check_guard(model, "the model's parameters moved since attach_engine(model) (model.to() / .half() / resize_token_embeddings re-allocate "
                        "them): the engine would decode with the old weights - call attach_engine(model) again")
    gc = generation_config if generation_config is not None else model.generation_config
    input_ids = inputs if inputs is not None else kw.pop("input_ids", None)
    if input_ids is None and kw.get("inputs_embeds") is None:
        raise ValueError("generate() needs input_ids")
    if input_ids is not None and (not torch.is_tensor(input_ids) or input_ids.dim() != 2):
        raise ValueError("generate() takes input_ids as a [batch, length] tensor (llava_calibrate.py:143 passes [1, L])")

    args, return_dict, want_attn = _resolve_generate_kwargs(gc, kw)
    out = eng.generate(input_ids, output_attentions=want_attn, **args, **kw)
    seqs = torch.stack(list(out.sequences)) if input_ids is not None else out.tokens       # embeddings prompts: no ids to echo (HF)
    if not return_dict:
        return seqs
    res = NativeGenerateOutput(sequences=seqs, tokens=out.tokens, stats=out.stats)
    if args["output_scores"]:
        res["scores"] = tuple(out.scores)
    if out.top_prob is not None:
        res["top_prob"], res["top_tok"] = out.top_prob, out.top_tok
    if getattr(out, "attentions", None) is not None:                  # one question + output_attentions: ['attentions'][0][-1] as llava_calibrate.py:180 reads it
        res["attentions"] = out.attentions
    return res


def attach_engine(model, share_storage: bool = True, use_graph: bool = True, max_questions: int = 64) -> VddLlavaEngine:
    """Puts a `VddLlavaEngine` built from `model`'s own parameters behind `model.generate`.  Returns the engine (also at
    `model._vdd_engine`).  `detach_engine(model)` restores HF's generate (with the generic evolve_vcd_sampling() loop if installed).
    After a weight update (LoRA merge, resize_token_embeddings) attach again."""
    cfg = config_from_hf(model)
    w = weights_from_hf(model, cfg, share_storage=share_storage)
    eng = VddLlavaEngine(cfg, weights=w, device=w.device, use_graph=use_graph, max_questions=max_questions)
    model._vdd_engine = eng
    _set_guard(model)
    model.generate = types.MethodType(_native_generate, model)
    return eng


# ------------------------------------------------------------------ language-model-only objects (InstructBLIP's Vicuna, Qwen-VL's LM)
def _is_qwen(lm) -> bool:
    return hasattr(lm, "transformer") and hasattr(lm.transformer, "wte")


def lm_config_from_hf(lm) -> LlavaConfig:
    """LMConfig of a Llama-shaped (`model.layers[i].self_attn.{q,k,v,o}_proj`: Vicuna inside InstructBLIP,
    blip2_vicuna_instruct.py:95-103) or Qwen-shaped (`transformer.h[i].attn.c_attn`, experiments/Qwen_VL/modeling_qwen.py:112-140,
    319-336, 440-500) causal LM.  The vision side of the LlavaConfig is a placeholder: such an engine takes `inputs_embeds` / text ids."""
    c = lm.config
    if _is_qwen(lm):
        heads, hd = int(c.num_attention_heads), int(c.kv_channels)
        if hd != 128 or heads * hd != int(c.hidden_size):
            raise ValueError("Qwen LM: the attention kernels are instantiated for 128-wide heads with hidden = heads x 128")
        if float(getattr(c, "rotary_pct", 1.0)) != 1.0:
            raise ValueError("rotary_pct != 1: partial rotary embeddings are not implemented")
        cfg = LMConfig(d=int(c.hidden_size), n_layers=int(c.num_hidden_layers), n_heads=heads, n_kv_heads=heads, head_dim=hd,
                       ffn=int(c.intermediate_size) // 2, vocab=int(lm.lm_head.weight.shape[0]), rope_theta=float(getattr(c, "rotary_emb_base", 10000.0)),
                       eps=float(c.layer_norm_epsilon), qkv_bias=True,
                       # plain rotary embedding up to the training length: beyond it the reference switches to dynamic NTK scaling and
                       # log-n attention (modeling_qwen.py:137-138, 645-659), which this engine does not implement - longer prompts raise
                       max_pos=int(getattr(c, "seq_length", 2048)))
    else:
        heads = int(c.num_attention_heads)
        hd = int(getattr(c, "head_dim", None) or c.hidden_size // heads)
        if hd != 128:
            raise ValueError(f"head_dim {hd}: the attention kernels are instantiated for 128")
        if getattr(c, "attention_bias", False) or getattr(c, "mlp_bias", False):
            raise ValueError("attention_bias / mlp_bias: not a Vicuna / Llama-2 language model")
        cfg = LMConfig(d=int(c.hidden_size), n_layers=int(c.num_hidden_layers), n_heads=heads,
                       n_kv_heads=int(getattr(c, "num_key_value_heads", None) or heads), head_dim=hd, ffn=int(c.intermediate_size),
                       vocab=int(lm.lm_head.weight.shape[0]), rope_theta=_rope_theta(c), eps=float(c.rms_norm_eps),
                       max_pos=int(c.max_position_embeddings))
    return LlavaConfig(cfg, VisionConfig(), getattr(c, "_name_or_path", "") or ("hf-qwen-lm" if _is_qwen(lm) else "hf-llama-lm"))


def lm_weights_from_hf(lm, cfg: Optional[LlavaConfig] = None) -> LlavaWeights:
    """The live LM parameters in the engine's layout (no vision tower, no projector).  Zero-copy for everything the kernels read as HF
    stores it; Llama's q/k/v and gate/up (Qwen: w2 = gate, w1 = up, `a1 * silu(a2)`, modeling_qwen.py:331-335) are concatenated once.
    Qwen's c_attn is already the fused [3 d, d] q|k|v projection (:128, :265-269) and is used in place."""
    cfg = cfg if cfg is not None else lm_config_from_hf(lm)
    p0 = lm.lm_head.weight
    if not p0.is_cuda or p0.dtype not in (torch.float16, torch.bfloat16):
        raise ValueError(f"the language model must be on the GPU in fp16 or bf16 (got {p0.dtype} on {p0.device})")
    sd, c = lm.state_dict(), cfg.lm
    w = LlavaWeights(cfg, p0.device, p0.dtype)
    get = lambda k: sd[k].detach().to(device=p0.device, dtype=p0.dtype).contiguous()
    if _is_qwen(lm):
        w.t["embed"], w.t["norm"], w.t["lm_head"] = get("transformer.wte.weight"), get("transformer.ln_f.weight"), get("lm_head.weight")
        for i in range(c.n_layers):
            p, q = f"l{i}.", f"transformer.h.{i}."
            w.t[p + "ln1"], w.t[p + "ln2"] = get(q + "ln_1.weight"), get(q + "ln_2.weight")
            w.t[p + "wqkv"], w.t[p + "bqkv_lm"] = get(q + "attn.c_attn.weight"), get(q + "attn.c_attn.bias")
            w.t[p + "wo"] = get(q + "attn.c_proj.weight")
            w.t[p + "wgu"] = torch.cat([get(q + "mlp.w2.weight"), get(q + "mlp.w1.weight")], 0).contiguous()
            w.t[p + "wd"] = get(q + "mlp.c_proj.weight")
            for b in ("attn.c_proj.bias", "mlp.w1.bias", "mlp.w2.bias", "mlp.c_proj.bias"):
                if q + b in sd:
                    raise ValueError(f"{q + b}: Qwen-VL's LM is built with no_bias=True (only c_attn carries a bias)")
    else:
        w.t["embed"], w.t["norm"], w.t["lm_head"] = get("model.embed_tokens.weight"), get("model.norm.weight"), get("lm_head.weight")
        for i in range(c.n_layers):
            p, q = f"l{i}.", f"model.layers.{i}."
            w.t[p + "ln1"], w.t[p + "ln2"] = get(q + "input_layernorm.weight"), get(q + "post_attention_layernorm.weight")
            w.t[p + "wqkv"] = torch.cat([get(q + f"self_attn.{n}_proj.weight") for n in ("q", "k", "v")], 0).contiguous()
            w.t[p + "wo"] = get(q + "self_attn.o_proj.weight")
            w.t[p + "wgu"] = torch.cat([get(q + "mlp.gate_proj.weight"), get(q + "mlp.up_proj.weight")], 0).contiguous()
            w.t[p + "wd"] = get(q + "mlp.down_proj.weight")
    return w


def _rows(x, mask):
    """[B, T, ...] (+ an attention mask with leading / trailing zeros: LAVIS pads left, blip2_vicuna_instruct.py:264) -> list of
    per-question rows without their padding."""
    if mask is None:
        return [x[i] for i in range(x.shape[0])]
    m = torch.as_tensor(mask).to(x.device).ne(0)
    return [x[i][m[i]] for i in range(x.shape[0])]


def _native_lm_generate(lm, inputs=None, generation_config=None, **kw):
    """`llm_model.generate(inputs_embeds=[B, T, d], attention_mask, do_sample, top_p, temperature, num_beams, max_length, min_length,
    repetition_penalty, length_penalty, num_return_sequences, images_cd=<embeddings of the noised image's prompt>, cd_beta, cd_alpha,
    use_dd, use_dd_unk, return_dict_in_generate, output_scores)` (blip2_vicuna_instruct.py:390-410) and Qwen's
    `generate(input_ids / inputs_embeds, stop_words_ids, min_new_tokens, ...)` (modeling_qwen.py:1044-1087) on the native engine.
    HF returns only the NEW tokens for an embeddings prompt; so does this."""
    eng: VddLlavaEngine = lm._vdd_engine
    _check_guard(lm, "the language model's parameters moved since the engine was attached: attach again")
    gc = generation_config if generation_config is not None else lm.generation_config
    input_ids = inputs if inputs is not None else kw.pop("input_ids", None)
    embeds, mask = kw.pop("inputs_embeds", None), kw.pop("attention_mask", None)
    if (input_ids is None) == (embeds is None):
        raise ValueError("generate() takes input_ids or inputs_embeds")
    sw = kw.pop("stop_words_ids", None)                                # Qwen: keyword, else generation_config (modeling_qwen.py:1061-1066)
    if sw is None:
        sw = getattr(gc, "stop_words_ids", None)
    if sw is not None:
        kw["stop_words_ids"] = sw
    cd = kw.pop("images_cd", None)
    if embeds is not None:
        kw["inputs_embeds"] = _rows(embeds, mask)
        if cd is not None:
            kw["images_cd"] = _rows(cd, mask)                           # the reference passes ONE mask for both prompts (:391, :402)
        ids = None
    else:
        if cd is not None:
            raise ValueError("images_cd with input_ids: a language-model-only engine has no image path; pass embeddings")
        ids = _rows(input_ids, mask)
    args, return_dict, _ = _resolve_generate_kwargs(gc, kw)
    out = eng.generate(ids, **args, **kw)
    seqs = out.tokens if ids is None else torch.nn.utils.rnn.pad_sequence(
        [torch.cat([r.to(out.tokens.device), t]) for r, t in zip(ids, out.tokens)], batch_first=True, padding_value=args["pad_token_id"] or 0)
    if not return_dict:
        return seqs
    res = NativeGenerateOutput(sequences=seqs, tokens=out.tokens, stats=out.stats)
    if args["output_scores"]:
        res["scores"] = tuple(out.scores)
    return res


def attach_lm_engine(lm, use_graph: bool = True, max_questions: int = 64) -> VddLlavaEngine:
    """A language-model-only `VddLlavaEngine` behind `lm.generate` for a Llama-shaped or Qwen-shaped causal LM: what
    `Blip2VicunaInstruct.generate` calls as `self.llm_model.generate(inputs_embeds=..., images_cd=inputs_embeds_cd, ...)`
    (blip2_vicuna_instruct.py:390-410 - the reference's own EVA-ViT / Q-Former code then runs unchanged in front of the native LM)."""
    cfg = lm_config_from_hf(lm)
    eng = VddLlavaEngine(cfg, weights=lm_weights_from_hf(lm, cfg), device=lm.lm_head.weight.device, use_graph=use_graph, max_questions=max_questions)
    lm._vdd_engine = eng
    _set_guard(lm)
    lm.generate = types.MethodType(_native_lm_generate, lm)
    return eng


# ------------------------------------------------------------------ InstructBLIP (lavis/models/blip2_models/blip2_vicuna_instruct.py)
def blip_config_from_model(model, sd=None):
    """BlipConfig of a `Blip2VicunaInstruct`-shaped object from its parameter shapes (+ the head counts its modules carry)."""
    from .blip_frontend import BlipConfig, EvaVitConfig, QFormerConfig
    sd = sd if sd is not None else model.state_dict()
    pe, pos = sd["visual_encoder.patch_embed.proj.weight"], sd["visual_encoder.pos_embed"]
    width, patch = int(pe.shape[0]), int(pe.shape[-1])
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("visual_encoder.blocks."))
    blocks = getattr(getattr(model, "visual_encoder", None), "blocks", None)
    heads = int(blocks[0].attn.num_heads) if blocks is not None and hasattr(blocks[0].attn, "num_heads") else max(1, width // 88)
    hd = int(sd["visual_encoder.blocks.0.attn.qkv.weight"].shape[0]) // 3 // heads
    vit = EvaVitConfig(image=int(round((pos.shape[1] - 1) ** 0.5)) * patch, patch=patch, width=width, layers=n_layers, heads=heads, head_dim=hd,
                       mlp=int(sd["visual_encoder.blocks.0.mlp.fc1.weight"].shape[0]))
    qt = sd["query_tokens"]
    q_layers = 1 + max(int(k.split(".")[4]) for k in sd if k.startswith("Qformer.bert.encoder.layer."))
    bert_cfg = getattr(getattr(model, "Qformer", None), "config", None)
    q_heads = int(getattr(bert_cfg, "num_attention_heads", max(1, qt.shape[2] // 64)))
    cross = [int(k.split(".")[4]) for k in sd if ".crossattention.self.key.weight" in k]
    freq = min((i for i in cross if i > 0), default=q_layers + 1)
    qf = QFormerConfig(hidden=int(qt.shape[2]), layers=q_layers, heads=q_heads, inter=int(sd["Qformer.bert.encoder.layer.0.intermediate.dense.weight"].shape[0]),
                       n_query=int(qt.shape[1]), cross_freq=freq, vocab=int(sd["Qformer.bert.embeddings.word_embeddings.weight"].shape[0]),
                       max_pos=int(sd["Qformer.bert.embeddings.position_embeddings.weight"].shape[0]))
    return BlipConfig(vit, qf, d_llm=int(sd["llm_proj.weight"].shape[0]))


def _native_blip_generate(model, samples, use_nucleus_sampling=False, num_beams=5, max_length=256, min_length=1, top_p=0.9,
                          repetition_penalty=1.5, length_penalty=1, num_captions=1, temperature=1, images_cd=None, cd_beta=None,
                          cd_alpha=None, use_dd_unk=None, use_dd=None, use_image=True, **engine_kw):
    """`Blip2VicunaInstruct.generate` (blip2_vicuna_instruct.py:233-418: same signature and defaults, same `(output_text, scores)`
    return) on the native path: EVA-ViT -> ln_vision -> Q-Former (queries + instruction) -> llm_proj for the image and, with
    images_cd, its noised copy (`blip_frontend`), then the engine's `inputs_embeds` path with the LAVIS keyword set (:390-410), token id
    0 -> 2 (:414), `llm_tokenizer.batch_decode(..., skip_special_tokens=True)`.  Tokenisers are the model's own (`tokenizer` = BERT with
    `max_txt_len` truncation for the Q-Former, `llm_tokenizer` for the LLM).  engine_kw: seed, cd_greedy, sync_every (not in LAVIS)."""
    eng, front = model._vdd_engine, model._vdd_front
    prompt = samples["prompt"] if "prompt" in samples.keys() else model.prompt
    image = samples["image"]
    if image is None or image.dim() != 4:
        raise ValueError("samples['image']: [batch, 3, S, S] (the video form, :299-324, is not on the contrastive-decoding path)")
    bs = image.size(0)
    if isinstance(prompt, str):
        prompt = [prompt] * bs
    elif len(prompt) != bs:
        raise AssertionError("The number of prompts must be equal to the batch size.")          # :255
    if "ocr_tokens" in samples.keys() and "{}" in prompt[0]:
        prompt = [p.format(", ".join(samples["ocr_tokens"][i][:30])) for i, p in enumerate(prompt)]
    qf_ids = None
    if getattr(model, "qformer_text_input", True):
        tq = model.tokenizer(prompt, padding="longest", truncation=True, max_length=model.max_txt_len, return_tensors="pt")
        qf_ids = [r.tolist() for r in _rows(tq.input_ids, tq.attention_mask)]
    model.llm_tokenizer.padding_side = "left"                                                  # :249
    tl = model.llm_tokenizer(prompt, padding="longest", return_tensors="pt")
    llm_ids = [r.tolist() for r in _rows(tl.input_ids, tl.attention_mask)]
    emb, emb_cd = front.build(image.to(eng.device), llm_ids, eng.w.t["embed"], qf_ids, images_cd.to(eng.device) if images_cd is not None else None)
    gc = model.llm_model.generation_config                                                    # (`# eos_token_id=self.eos_token_id`, :398: HF's defaults apply)
    eos, pad = getattr(gc, "eos_token_id", None), getattr(gc, "pad_token_id", None)
    if eos is not None and pad is None:
        pad = eos[0] if isinstance(eos, (list, tuple)) else int(eos)
    out = eng.generate(None, inputs_embeds=emb, images_cd=emb_cd, do_sample=bool(use_nucleus_sampling), top_p=top_p, temperature=temperature,
                       num_beams=num_beams, max_length=max_length, min_length=min_length, repetition_penalty=repetition_penalty,
                       num_return_sequences=num_captions, cd_beta=cd_beta, cd_alpha=cd_alpha, use_dd=bool(use_dd), use_dd_unk=bool(use_dd_unk),
                       eos_token_id=eos, pad_token_id=pad, output_scores=True, top_k=getattr(gc, "top_k", 50) if use_nucleus_sampling else None,
                       **engine_kw)
    outputs = out.tokens.clone()
    outputs[outputs == 0] = 2                                                                  # :414
    text = [t.strip() for t in model.llm_tokenizer.batch_decode(outputs, skip_special_tokens=True)]
    return text, out.scores[0]


def attach_blip_engine(model, use_graph: bool = True, max_questions: int = 64):
    """The native front-end + engine behind a live `Blip2VicunaInstruct`-shaped object (`visual_encoder`, `ln_vision`, `Qformer`,
    `query_tokens`, `llm_proj`, `llm_model`, `tokenizer`, `llm_tokenizer`, `max_txt_len`): the driver's three calls per question
    (experiments/eval/calibrate/blip_calibrate.py:84-98) then run `model.generate({"image", "prompt"}, use_nucleus_sampling=True,
    num_beams=1, top_p=..., repetition_penalty=1, images_cd=..., cd_beta=...)` unchanged.  Returns (engine, front-end)."""
    from .blip_frontend import BlipWeights, InstructBlipFrontEnd
    lm = model.llm_model
    cfg = lm_config_from_hf(lm)
    w = lm_weights_from_hf(lm, cfg)
    sd = {k: v for k, v in model.state_dict().items() if not k.startswith("llm_model.")}
    bcfg = blip_config_from_model(model, sd)
    if bcfg.d_llm != cfg.lm.d:
        raise ValueError(f"llm_proj maps to {bcfg.d_llm}, the language model is {cfg.lm.d} wide")
    front = InstructBlipFrontEnd(BlipWeights.from_state_dict(bcfg, sd, w.device, dtype=w.dtype))
    eng = VddLlavaEngine(cfg, weights=w, device=w.device, use_graph=use_graph, max_questions=max_questions)
    model._vdd_engine, model._vdd_front = eng, front
    model.generate = types.MethodType(_native_blip_generate, model)
    return eng, front


# ------------------------------------------------------------------ Qwen-VL (experiments/Qwen_VL/modeling_qwen.py)
def qwen_spliced_embeddings(model, input_ids: torch.Tensor, images: Optional[torch.Tensor], feats: Optional[torch.Tensor] = None):
    """What `QWenModel.forward` feeds its decoder for a prompt with image spans (modeling_qwen.py:545-575, 631-640, 688-693): token
    embeddings with the positions between every <img> (config.visual['image_start_id']) and </img> (+1) replaced by the rows the
    CALLER'S ViT + resampler (`model.transformer.visual`: out of scope here) returns for the image - from the
    `images` tensor when given (:565-566), else from the path spelled out in the ids (`visual.encode`, :567-568); `feats`: rows the caller already has from that tower (one [rows, d] per span).  -> list of [T, d]."""
    tr = model.transformer
    hidden = tr.wte(input_ids)
    start = int(model.config.visual["image_start_id"])
    if not bool((input_ids == start).any()):
        return [hidden[i] for i in range(hidden.shape[0])]
    bos = torch.where(input_ids == start)
    eos = torch.where(input_ids == start + 1)
    if not bool((bos[0] == eos[0]).all()):
        raise ValueError("unbalanced <img> ... </img> spans")
    pos = torch.stack((bos[0], bos[1], eos[1]), dim=1).tolist()
    if feats is not None:
        pass
    elif images is not None:
        feats = tr.visual(images)
    else:
        paths = []
        for i, a, b in pos:
            img = input_ids[i][a + 1: b - 1].tolist()
            paths.append(bytes(img[: img.index(start + 2)]).decode("utf-8"))
        feats = tr.visual.encode(paths)
    hidden = hidden.clone()
    for idx, (i, a, b) in enumerate(pos):
        hidden[i][a + 1: b] = feats[idx].to(hidden.dtype)
    return [hidden[i] for i in range(hidden.shape[0])]


def _native_qwen_generate(model, inputs=None, generation_config=None, **kw):
    """`model.generate(input_ids=..., attention_mask=..., images=image_tensor, images_cd=image_tensor_cd, use_dd, use_dd_unk, cd_alpha,
    cd_beta, min_new_tokens=1, max_new_tokens=20, eos / pad = tokenizer.eod_id, ...)` (experiments/eval/MME/run_qwen.py:190-213;
    modeling_qwen.py:1044-1087 for stop_words_ids).  The visual embeddings come from the caller's own tower; the language model, the
    branch passes (image-free branches of a Qwen prompt re-run the SAME inputs) and the sampling tail are native."""
    input_ids = inputs if inputs is not None else kw.pop("input_ids", None)
    if input_ids is None:
        raise ValueError("generate() needs input_ids")
    mask = kw.pop("attention_mask", None)
    images, images_cd = kw.pop("images", None), kw.pop("images_cd", None)
    rows = _rows(input_ids, mask)
    keep = None if mask is None else torch.as_tensor(mask).to(input_ids.device).ne(0)
    emb = qwen_spliced_embeddings(model, input_ids, images)
    emb = emb if keep is None else [e[keep[i]] for i, e in enumerate(emb)]
    kw["inputs_embeds"] = torch.nn.utils.rnn.pad_sequence(emb, batch_first=True) if len({e.shape[0] for e in emb}) == 1 else None
    if kw["inputs_embeds"] is None:
        raise ValueError("prompts of different lengths in one call: the reference's drivers call generate() per question")
    if images_cd is not None:
        cd = qwen_spliced_embeddings(model, input_ids, images_cd)
        kw["images_cd"] = torch.stack(cd if keep is None else [e[keep[i]] for i, e in enumerate(cd)])
    return_dict = kw.get("return_dict_in_generate", getattr(generation_config or model.generation_config, "return_dict_in_generate", False))
    out = _native_lm_generate(model, None, generation_config, **dict(kw, return_dict_in_generate=True))
    gc = generation_config if generation_config is not None else model.generation_config
    pad = kw.get("pad_token_id", getattr(gc, "pad_token_id", None))
    pad = pad if pad is not None else kw.get("eos_token_id", getattr(gc, "eos_token_id", None))      # Qwen: pad = eos = eod (run_qwen.py:196-197)
    pad = (pad[0] if isinstance(pad, (list, tuple)) else pad) if pad is not None else 0
    toks = out["tokens"]
    seqs = torch.nn.utils.rnn.pad_sequence([torch.cat([r.to(toks.device), t]) for r, t in zip(rows, toks)], batch_first=True,
                                           padding_value=int(pad))        # HF echoes the prompt ids in front (ids were given; CPU ids too)
    out["sequences"] = seqs
    return out if return_dict else seqs


def attach_qwen_engine(model, use_graph: bool = True, max_questions: int = 64) -> VddLlavaEngine:
    """The native LM engine behind a `QWenLMHeadModel`-shaped object's `generate`; `model.transformer.visual` (the caller's ViT +
    resampler) keeps producing the 256 image rows per <img> span."""
    cfg = lm_config_from_hf(model)
    eng = VddLlavaEngine(cfg, weights=lm_weights_from_hf(model, cfg), device=model.lm_head.weight.device, use_graph=use_graph,
                         max_questions=max_questions)
    model._vdd_engine = eng
    _set_guard(model)
    model.generate = types.MethodType(_native_qwen_generate, model)
    return eng


def detach_engine(model) -> None:
    """HF's own generate back, and (after attach_engine(..., share_storage=True)) every parameter that was re-pointed at a slice of a
    fused tensor gets a contiguous storage of its own again, so that save_pretrained / safetensors take the model as before."""
    model.__dict__.pop("generate", None)
    model.__dict__.pop("_vdd_engine", None)
    model.__dict__.pop("_vdd_front", None)
    model.__dict__.pop("_vdd_guard", None)
    for p in model.__dict__.pop("_vdd_shared", []):
        p.data = p.data.clone(memory_format=torch.contiguous_format)


__all__ = ["attach_engine", "attach_lm_engine", "attach_blip_engine", "attach_qwen_engine", "detach_engine", "config_from_hf", "weights_from_hf",
           "lm_config_from_hf", "lm_weights_from_hf", "blip_config_from_model", "qwen_spliced_embeddings", "NativeGenerateOutput",
           "IMAGE_TOKEN_INDEX"]
