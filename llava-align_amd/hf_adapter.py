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


def _native_generate(model, inputs=None, generation_config=None, **kw):
    """`GenerationMixin.generate`'s argument handling for the keywords the reference's drivers use, in front of
    `VddLlavaEngine.generate`.  Explicit keywords (None included: llava_calibrate.py:170-171 pass top_p=None, top_k=None) override
    `model.generation_config`, as `generation_config.update(**kwargs)` does in HF."""
    eng: VddLlavaEngine = model._vdd_engine
    if model.lm_head.weight.data_ptr() != eng.w.t["lm_head"].data_ptr():
        raise RuntimeError("the model's parameters moved since attach_engine(model) (model.to() / .half() / resize_token_embeddings re-allocate "
                           "them): the engine would decode with the old weights - call attach_engine(model) again")
    gc = generation_config if generation_config is not None else model.generation_config
    input_ids = inputs if inputs is not None else kw.pop("input_ids", None)
    if input_ids is None and kw.get("inputs_embeds") is None:
        raise ValueError("generate() needs input_ids")
    if input_ids is not None and (not torch.is_tensor(input_ids) or input_ids.dim() != 2):
        raise ValueError("generate() takes input_ids as a [batch, length] tensor (llava_calibrate.py:143 passes [1, L])")

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
    for k in ("output_hidden_states", "use_cache", "synced_gpus"):     # accepted without effect
        opt(k, None)
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
    model.generate = types.MethodType(_native_generate, model)
    return eng


def detach_engine(model) -> None:
    """HF's own generate back, and (after attach_engine(..., share_storage=True)) every parameter that was re-pointed at a slice of a
    fused tensor gets a contiguous storage of its own again, so that save_pretrained / safetensors take the model as before."""
    model.__dict__.pop("generate", None)
    model.__dict__.pop("_vdd_engine", None)
    for p in model.__dict__.pop("_vdd_shared", []):
        p.data = p.data.clone(memory_format=torch.contiguous_format)


__all__ = ["attach_engine", "detach_engine", "config_from_hf", "weights_from_hf", "NativeGenerateOutput", "IMAGE_TOKEN_INDEX"]
