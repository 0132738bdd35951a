"""A `LlavaLlamaForCausalLM`-shaped module built from the INSTALLED transformers' LlamaForCausalLM and CLIPVisionModel, composed the
way the reference composes them (experiments/llava/model/language_model/llava_llama.py:46-174, llava_arch.py:31-204,
multimodal_encoder/clip_encoder.py:8-51, multimodal_projector/builder.py:33-46): what `load_pretrained_model` (builder.py:26-148)
hands to the eval scripts, minus the checkpoint.  Test infrastructure: the object `attach_engine()` is tested on, and the HF-eager
comparator the drop-in loop runs on.  Batch 1 per call, like the reference's drivers (llava_calibrate.py:130-177)."""
import torch
import torch.nn as nn
from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM

IMAGE_TOKEN_INDEX = -200


class ClipTower(nn.Module):
    """clip_encoder.py:8-51: hidden_states[select_layer] of the CLIP ViT, class token dropped ('patch')."""

    def __init__(self, clip_cfg, select_layer=-2):
        super().__init__()
        self.vision_tower = CLIPVisionModel(clip_cfg)
        self.select_layer, self.select_feature, self.is_loaded = select_layer, "patch", True

    @property
    def dtype(self):
        return next(self.vision_tower.parameters()).dtype

    @torch.no_grad()
    def forward(self, images):
        out = self.vision_tower(images.to(dtype=self.dtype), output_hidden_states=True)          # :44-46
        return out.hidden_states[self.select_layer][:, 1:].to(images.dtype)                      # :29-37


def splice(embed_tokens, row, feat):
    """llava_arch.py:122-163, one sequence, one image slot (pinned by tests/golden/splice.npz)."""
    s = int(torch.where(row == IMAGE_TOKEN_INDEX)[0][0])
    return torch.cat([embed_tokens(row[:s]), feat, embed_tokens(row[s + 1:])], 0)


class HfLlava(LlamaForCausalLM):
    def __init__(self, config, clip_cfg):
        super().__init__(config)
        self.model.vision_tower = ClipTower(clip_cfg)
        self.model.mm_projector = nn.Sequential(nn.Linear(clip_cfg.hidden_size, config.hidden_size), nn.GELU(),
                                                nn.Linear(config.hidden_size, config.hidden_size))   # mlp2x_gelu
        self.config.mm_projector_type = "mlp2x_gelu"

    def get_model(self):
        return self.model

    def get_vision_tower(self):
        return self.model.vision_tower

    def encode_images(self, images):                                                                 # llava_arch.py:82-85
        return self.model.mm_projector(self.model.vision_tower(images))

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None, labels=None,
                use_cache=None, images=None, images_cd=None, cd_alpha=None, cd_beta=None, use_dd=None, use_dd_unk=None,
                cd_greedy=None, image_sizes=None, **kw):
        # prepare_inputs_labels_for_multimodal for ONE sequence (llava_arch.py:87-204): no images or a decode step -> ids as they are
        # (:91-94); else the text chunks are embedded around the projected patch features (:122-163)
        if images is not None and input_ids is not None and input_ids.shape[1] != 1:
            assert input_ids.shape[0] == 1, "one question per call"
            emb = self.model.embed_tokens
            inputs_embeds = splice(emb, input_ids[0], self.encode_images(images)[0].to(emb.weight.dtype))[None]
            input_ids = None
        # un-padded single sequences: the all-ones masks of the loop (re-sized by the reference at :92-93 / :199-202) carry nothing, and
        # positions follow the CACHE length (the spliced sequence is 575 longer than the ids the generation loop counts)
        kw.pop("cache_position", None)
        return super().forward(input_ids=input_ids, attention_mask=None, position_ids=None, past_key_values=past_key_values,
                               inputs_embeds=inputs_embeds, use_cache=use_cache, **kw)

    def prepare_inputs_for_generation_cd(self, input_ids, **kw):                                     # llava_llama.py:153-174
        d = self.prepare_inputs_for_generation(input_ids, **kw)
        d["images"] = kw.get("images_cd")
        return d


class HfProto:
    """The 4.31-era GenerationMixin hooks the reference's sample() calls (vcd_sample.py:106-114,150,161,266-277; llava_llama.py:130-174),
    over an HfLlava: lets oracle.reference_loop - the restatement of the reference's OWN loop - and this package's drop-in `sample()`
    drive the installed transformers' eager Llama + CLIP stack exactly like the reference's scripts drive LlavaLlamaForCausalLM
    (bench.py `eager_gpu` / `dropin_gpu`).  output_attentions as the driver passes it (llava_calibrate.py:175): with the eager attention
    implementation every layer materialises its [1, H, T, S] map, as the reference era's LlamaAttention always did."""

    def __init__(self, model, output_attentions=True, logits_on_device=False):
        from types import SimpleNamespace
        self.m, self.attn, self.on_dev = model, output_attentions, logits_on_device
        self.config = SimpleNamespace(is_encoder_decoder=False)
        self.generation_config = model.generation_config
        self.device = model.lm_head.weight.device

    def prepare_inputs_for_generation(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kw):
        if past_key_values:
            input_ids = input_ids[:, -1:]
        return {"input_ids": input_ids, "past_key_values": past_key_values, "use_cache": kw.get("use_cache"), "attention_mask": attention_mask,
                "images": kw.get("images", None)}

    def prepare_inputs_for_generation_cd(self, input_ids, past_key_values=None, attention_mask=None, inputs_embeds=None, **kw):
        d = self.prepare_inputs_for_generation(input_ids, past_key_values=past_key_values, attention_mask=attention_mask, **kw)
        d["images"] = kw.get("images_cd", None)                                                   # llava_llama.py:170
        return d

    def _update_model_kwargs_for_generation(self, outputs, model_kwargs, is_encoder_decoder=False, **_):
        model_kwargs["past_key_values"] = outputs.past_key_values
        am = model_kwargs.get("attention_mask")
        if am is not None:
            model_kwargs["attention_mask"] = torch.cat([am, am.new_ones((am.shape[0], 1))], dim=-1)
        return model_kwargs

    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None, past_key_values=None, use_cache=None, images=None, return_dict=True,
                 output_attentions=None, output_hidden_states=None, **_):
        from types import SimpleNamespace
        dev = self.device
        out = self.m(input_ids=input_ids.to(dev), past_key_values=past_key_values, use_cache=True,
                     images=images.to(dev, self.m.dtype) if images is not None else None, output_attentions=self.attn)
        logits = out.logits if self.on_dev else out.logits.cpu()
        return SimpleNamespace(logits=logits, past_key_values=out.past_key_values, attentions=out.attentions, hidden_states=None)


def build(device, dtype, d=256, layers=2, heads=2, ffn=512, vocab=1000, clip_width=128, clip_layers=3, clip_heads=2, clip_mlp=256,
          image=56, patch=14, max_pos=512, lm_head_gain=6.0, seed=0, attn_implementation=None):
    """Random-init model on `device` in `dtype`.  Defaults = the engine's 'tiny' preset; 7B widths: d 4096, heads 32, ffn 11008,
    vocab 32000, clip 1024 / 16 heads / mlp 4096 / image 336."""
    torch.manual_seed(seed)
    cfg = LlamaConfig(vocab_size=vocab, hidden_size=d, intermediate_size=ffn, num_hidden_layers=layers, num_attention_heads=heads,
                      num_key_value_heads=heads, head_dim=d // heads, rms_norm_eps=1e-5, max_position_embeddings=max_pos,
                      attention_bias=False, mlp_bias=False, tie_word_embeddings=False, pad_token_id=0, eos_token_id=None, bos_token_id=1,
                      **({"attn_implementation": attn_implementation} if attn_implementation else {}))
    ccfg = CLIPVisionConfig(hidden_size=clip_width, intermediate_size=clip_mlp, num_hidden_layers=clip_layers, num_attention_heads=clip_heads,
                            image_size=image, patch_size=patch, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                            **({"attn_implementation": attn_implementation} if attn_implementation else {}))
    with torch.device(device):
        m = HfLlava(cfg, ccfg)
    with torch.no_grad():
        m.lm_head.weight.mul_(lm_head_gain)                 # N(0, 0.02) everywhere gives flat logits: clear top-1 margins need a gain
        for p in m.model.vision_tower.parameters():         # CLIP's default init is tiny: make every ViT path contribute
            p.mul_(2.0)
    return m.to(dtype=dtype).eval()
