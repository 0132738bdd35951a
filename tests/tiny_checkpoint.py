"""Writes a test-sized LLaVA-1.5 checkpoint DIRECTORY the way a release lays it out (experiments/llava/model/builder.py:26-148 reads exactly
this): config.json, model.safetensors (HF parameter names, CLIP tower inside), tokenizer files, preprocessor_config.json - plus a POPE-like
question file and an image folder.  Weights are those of tests/hf_llava.build (random, tiny widths); the tokenizer is a generated 1,000-word
WordLevel vocabulary with Llama's special ids (<unk> 0, <s> 1, </s> 2) and a BOS in front of every encoding."""
import json
import os

import numpy as np
import torch


def write_checkpoint(root: str, seed: int = 0) -> dict:
    import hf_llava
    from PIL import Image
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import CLIPImageProcessor, PreTrainedTokenizerFast
    ckpt, imgs = os.path.join(root, "tiny-llava"), os.path.join(root, "images")
    os.makedirs(ckpt, exist_ok=True); os.makedirs(imgs, exist_ok=True)
    torch.manual_seed(seed)
    m = hf_llava.build("cpu", torch.float16)
    with torch.no_grad():
        m.lm_head.weight.mul_(8.0)                          # a peaked next-token distribution, as a trained model has
    save_file({k: v.contiguous() for k, v in m.state_dict().items()}, os.path.join(ckpt, "model.safetensors"))
    c, v = m.config, m.get_vision_tower().vision_tower.config
    cfg = {"architectures": ["LlavaLlamaForCausalLM"], "model_type": "llava", "hidden_size": c.hidden_size, "num_hidden_layers": c.num_hidden_layers,
           "num_attention_heads": c.num_attention_heads, "num_key_value_heads": c.num_key_value_heads, "intermediate_size": c.intermediate_size,
           "vocab_size": c.vocab_size, "rms_norm_eps": c.rms_norm_eps, "rope_theta": 10000.0, "max_position_embeddings": c.max_position_embeddings,
           "head_dim": c.hidden_size // c.num_attention_heads, "mm_vision_tower": "openai/clip-tiny-test", "mm_vision_select_layer": -2,
           "mm_vision_select_feature": "patch", "mm_projector_type": "mlp2x_gelu",
           "vision_config": {"hidden_size": v.hidden_size, "num_hidden_layers": v.num_hidden_layers, "num_attention_heads": v.num_attention_heads,
                             "intermediate_size": v.intermediate_size, "image_size": v.image_size, "patch_size": v.patch_size,
                             "layer_norm_eps": v.layer_norm_eps}}
    json.dump(cfg, open(os.path.join(ckpt, "config.json"), "w"), indent=1)
    words = ["<unk>", "<s>", "</s>", "yes", "no"] + [f"w{i}" for i in range(c.vocab_size - 5)]
    t = Tokenizer(models.WordLevel({w: i for i, w in enumerate(words)}, unk_token="<unk>"))
    t.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    t.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
    PreTrainedTokenizerFast(tokenizer_object=t, unk_token="<unk>", bos_token="<s>", eos_token="</s>").save_pretrained(ckpt)
    CLIPImageProcessor(size={"shortest_edge": v.image_size}, crop_size={"height": v.image_size, "width": v.image_size}).save_pretrained(ckpt)
    rng = np.random.default_rng(seed)
    n_img, per = 5, 3
    for i in range(n_img):
        Image.fromarray(rng.integers(0, 255, size=(70 + 3 * i, 90, 3), dtype=np.uint8)).save(os.path.join(imgs, f"im{i}.png"))
    qfile = os.path.join(root, "questions.json")
    with open(qfile, "w") as f:
        for q in range(n_img * per):
            im = (q * 2) % n_img                                      # images interleaved in file order
            f.write(json.dumps({"question_id": 100 + q, "image": f"im{im}.png", "text": f"w{q} w{q + 7} the w{3 * q} ?", "label": ("yes", "no")[q % 2]}) + "\n")
    return {"ckpt": ckpt, "images": imgs, "questions": qfile, "n_questions": n_img * per, "config": cfg}
