"""Checkpoint directories -> engines, for the command-line entry points of the drivers (pope_driver / mme_driver / blip_driver `main()`):
what `load_pretrained_model(model_path, model_base, model_name)` does for the reference's scripts (experiments/llava/model/builder.py:26-148:
tokenizer, model weights in fp16, CLIP image processor from the vision tower), without instantiating an HF model - the safetensors /
.bin shards go straight into `LlavaWeights.from_state_dict`.

A LLaVA-1.5 directory holds config.json (hidden_size, num_hidden_layers, num_attention_heads, num_key_value_heads, intermediate_size,
vocab_size, rms_norm_eps, rope_theta, max_position_embeddings, mm_vision_tower, mm_vision_select_layer), the weight shards, the tokenizer
files, and - released checkpoints keep the CLIP tower OUTSIDE (`mm_vision_tower` names it; builder.py:139-143 loads it separately) - either
the tower's weights inside the shards or a `--vision-tower` directory with its own config.json / weights / preprocessor_config.json.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional, Tuple

import torch

from .engine import LlavaConfig, LlavaWeights, LMConfig, VddLlavaEngine, VisionConfig, preset


def load_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Every *.safetensors shard of a directory (or one file); pytorch_model*.bin / *.pth as the fallback the older releases need."""
    files = [path] if os.path.isfile(path) else sorted(os.path.join(path, f) for f in os.listdir(path))
    sd: Dict[str, torch.Tensor] = {}
    st = [f for f in files if f.endswith(".safetensors")]
    if st:
        from safetensors.torch import load_file
        for f in st:
            sd.update(load_file(f))
        return sd
    for f in files:
        if f.endswith((".bin", ".pth", ".pt")) and os.path.basename(f) != "training_args.bin":
            part = torch.load(f, map_location="cpu", weights_only=True)
            sd.update(part.get("model", part) if isinstance(part, dict) else part)
    if not sd:
        raise FileNotFoundError(f"no *.safetensors / *.bin weights under {path}")
    return sd


def _vision_config(d: Optional[dict], select_layer: int) -> Optional[VisionConfig]:
    if not d:
        return None
    d = d.get("vision_config", d)
    if "hidden_size" not in d:
        return None
    return VisionConfig(image=int(d.get("image_size", 336)), patch=int(d.get("patch_size", 14)), width=int(d["hidden_size"]),
                        layers=int(d["num_hidden_layers"]), select_layer=select_layer, heads=int(d["num_attention_heads"]),
                        mlp=int(d["intermediate_size"]), eps=float(d.get("layer_norm_eps", 1e-5)))


def config_from_dir(model_path: str, vision_tower: Optional[str] = None, fallback: Optional[str] = None) -> LlavaConfig:
    """LlavaConfig from <model_path>/config.json (+ the vision tower's config.json: `vision_tower` or a local `mm_vision_tower`; a
    `vision_config` entry inside config.json also serves).  `fallback`: a preset name for whatever the files do not say (the released
    checkpoints name their tower on the hub - `openai/clip-vit-large-patch14-336` - which is the preset's)."""
    base = preset(fallback) if fallback else None
    cj = os.path.join(model_path, "config.json")
    if not os.path.exists(cj):
        if base is None:
            raise FileNotFoundError(f"{cj} is missing and no --preset was given")
        return base
    c = json.load(open(cj))
    heads = int(c["num_attention_heads"])
    lm = LMConfig(d=int(c["hidden_size"]), n_layers=int(c["num_hidden_layers"]), n_heads=heads, n_kv_heads=int(c.get("num_key_value_heads") or heads),
                  head_dim=int(c.get("head_dim") or int(c["hidden_size"]) // heads), ffn=int(c["intermediate_size"]), vocab=int(c["vocab_size"]),
                  eps=float(c.get("rms_norm_eps", 1e-5)), rope_theta=float(c.get("rope_theta", 10000.0)),
                  max_pos=int(c.get("max_position_embeddings", 4096)))
    sel = int(c.get("mm_vision_select_layer", -2))
    if c.get("mm_vision_select_feature", "patch") != "patch":
        raise ValueError("mm_vision_select_feature must be 'patch' (clip_encoder.py:33-38: LLaVA-1.5 drops the CLS token)")
    vis = _vision_config(c if "vision_config" in c else None, sel)
    for cand in (vision_tower, c.get("mm_vision_tower")):
        if vis is None and cand and os.path.exists(os.path.join(cand, "config.json")):
            vis = _vision_config(json.load(open(os.path.join(cand, "config.json"))), sel)
    if vis is None:
        if base is None:
            raise FileNotFoundError("no vision tower configuration: pass --vision-tower DIR (the CLIP checkpoint `mm_vision_tower` names) or --preset")
        vis = base.vision
    return LlavaConfig(lm, vis, os.path.basename(model_path.rstrip("/")) or "llava")


def load_tokenizer(path: str):
    """builder.py:45,105: AutoTokenizer.from_pretrained(model_path, use_fast=False); directories that only hold a tokenizer.json load fast."""
    from transformers import AutoTokenizer
    try:
        return AutoTokenizer.from_pretrained(path, use_fast=False)
    except Exception:
        return AutoTokenizer.from_pretrained(path)


def load_llava(model_path: str, device, dtype=torch.float16, vision_tower: Optional[str] = None, fallback_preset: Optional[str] = None,
               use_graph: bool = True) -> Tuple[VddLlavaEngine, object, object]:
    """-> (engine, tokenizer, CLIP image processor) of a LLaVA-1.5 checkpoint directory; dtype fp16 as the reference loads it (builder.py:40)."""
    from transformers import CLIPImageProcessor
    cfg = config_from_dir(model_path, vision_tower, fallback_preset)
    sd = load_state_dict(model_path)
    has_tower = any(k.endswith("embeddings.patch_embedding.weight") for k in sd)
    cj = os.path.join(model_path, "config.json")
    tower_dir = vision_tower or (json.load(open(cj)).get("mm_vision_tower") if os.path.exists(cj) else None)
    vision_sd = None
    if not has_tower:
        if not tower_dir or not os.path.isdir(tower_dir):
            raise FileNotFoundError(f"the checkpoint holds no CLIP tower and its mm_vision_tower ({tower_dir!r}) is not a local directory: pass --vision-tower DIR")
        vision_sd = load_state_dict(tower_dir)
    w = LlavaWeights.from_state_dict(cfg, sd, device, vision_sd=vision_sd, dtype=dtype)
    eng = VddLlavaEngine(cfg, weights=w, device=device, use_graph=use_graph)
    proc_dir = next((d for d in (model_path, tower_dir) if d and os.path.exists(os.path.join(d, "preprocessor_config.json"))), None)
    if proc_dir is None:
        raise FileNotFoundError("no preprocessor_config.json (CLIPImageProcessor) in the checkpoint or the vision tower directory")
    return eng, load_tokenizer(model_path), CLIPImageProcessor.from_pretrained(proc_dir)


def qwen_embed_prompt(model, tokenize, device, keep_images: int = 16):
    """The Qwen front-end the drivers call (mme_driver.qwen_mme_inputs, qwen_driver.run_qwen_pope): `embed_prompt(text, image)` -> [T, d] for
    prompts without an <img> span (image = None), else ([T, d], n_shared) with n_shared = the rows up to and including </img>.  image: a path
    string (the tower reads the file the prompt spells out, modeling_qwen.py:567-568) or a [3, S, S] tensor (`images=`, :565-566); the tower's
    rows for the last `keep_images` distinct tensor OBJECTS are kept, so the six POPE questions about one image cost one ViT pass (the
    reference: one per generate() call).  tokenize(text) -> ids."""
    from .hf_adapter import qwen_spliced_embeddings
    start = int(model.config.visual["image_start_id"])
    kept: Dict[int, tuple] = {}                                   # id(tensor) -> (tensor, rows): holding the tensor keeps its id unique

    @torch.no_grad()
    def embed_prompt(text, image):
        ids = torch.tensor([tokenize(text)], device=device)
        feats = None
        if torch.is_tensor(image):
            hit = kept.get(id(image))
            if hit is None:
                if len(kept) >= keep_images:
                    kept.pop(next(iter(kept)))
                hit = kept[id(image)] = (image, model.transformer.visual(image[None].to(device)))
            feats = hit[1]
        e = qwen_spliced_embeddings(model, ids, None, feats=feats)[0]
        if image is None:
            return e
        return e, int((ids[0] == start + 1).nonzero()[0]) + 1
    return embed_prompt


def load_qwen(model_path: str, device, dtype=torch.bfloat16, use_graph: bool = True):
    """-> (engine, tokenizer, HF model, embed_prompt) of a Qwen-VL directory, as the reference's drivers load it
    (`AutoTokenizer / QWenLMHeadModel.from_pretrained(model_path, trust_remote_code=True)`, MME/run_qwen.py:146-157, qwen_calibrate.py:75-86:
    the caller's environment must provide what modeling_qwen.py imports).  The language model runs natively (hf_adapter.lm_weights_from_hf);
    the ViT + resampler stay the model's own `transformer.visual` behind
    `embed_prompt(text, image) -> [T, d]` or `([T, d], n_shared)`: image = None for prompts without an <img> span, a path string (the tower
    reads the file the prompt spells out, modeling_qwen.py:567-568) or a [3, S, S] tensor (`images=`, :565-566); n_shared = the rows up to and
    including </img>, which every prompt about the same image starts with."""
    from transformers import AutoModelForCausalLM, AutoTokenizer
    from .hf_adapter import lm_config_from_hf, lm_weights_from_hf, qwen_spliced_embeddings
    tok = AutoTokenizer.from_pretrained(model_path, trust_remote_code=True)
    tok.padding_side = "left"
    tok.pad_token_id = tok.eod_id
    model = AutoModelForCausalLM.from_pretrained(model_path, trust_remote_code=True, torch_dtype=dtype).to(device).eval()
    cfg = lm_config_from_hf(model)
    eng = VddLlavaEngine(cfg, weights=lm_weights_from_hf(model, cfg), device=device, use_graph=use_graph)
    embed_prompt = qwen_embed_prompt(model, lambda text: tok(text).input_ids, device)
    return eng, tok, model, embed_prompt


def tokenizer_image_token(tok, prompt: str, image_token_index: int = -200):
    """experiments/llava/mm_utils.py tokenizer_image_token: split at '<image>', tokenise the chunks, join with -200, one BOS in front."""
    chunks = [tok(c).input_ids for c in prompt.split("<image>")]
    has_bos = len(chunks[0]) > 0 and chunks[0][0] == tok.bos_token_id
    ids = list(chunks[0])
    for c in chunks[1:]:
        ids += [image_token_index] + (c[1:] if has_bos else c)        # later chunks lose their BOS
    return ids


def clip_preprocess(proc, path: str) -> torch.Tensor:
    """`image_processor.preprocess(image, return_tensors='pt')['pixel_values'][0]` of an RGB image file (llava_calibrate.py:146-147)."""
    from PIL import Image
    return proc.preprocess(Image.open(path).convert("RGB"), return_tensors="pt")["pixel_values"][0]
