"""Test doubles of the two model objects the reference's remaining drivers call (test infrastructure, like tests/hf_llava.py):

* `BlipDouble`: a `Blip2VicunaInstruct`-shaped object (experiments/lavis/models/blip2_models/blip2_vicuna_instruct.py:58-110): `llm_model` =
  the INSTALLED transformers' LlamaForCausalLM, the EVA-ViT / Q-Former / projection parameters under LAVIS's names (tests/blip_weights.py:
  the weights the golden fixtures of tests/golden/blip_vectors.npz were made with the real LAVIS modules on), `tokenizer` /
  `llm_tokenizer` / `max_txt_len` / `qformer_text_input`.  LAVIS itself does not travel to the GPU box; what `attach_blip_engine`
  reads of the object (state_dict names, head counts, tokenisers, generation_config) is all here.
* `QwenDouble`: a `QWenLMHeadModel`-shaped object (experiments/Qwen_VL/modeling_qwen.py:112-140, 319-336, 440-500, 747-790): `transformer.wte /
  h[i].{ln_1, attn.c_attn, attn.c_proj, ln_2, mlp.w1, mlp.w2, mlp.c_proj} / ln_f / visual`, `lm_head`, a config with Qwen's field names,
  and a stand-in `visual` (the ViT + resampler are out of scope: anything that maps an image tensor to [n, n_img_rows, d])."""
from types import SimpleNamespace

import torch
import torch.nn as nn
from transformers import GenerationConfig, LlamaConfig, LlamaForCausalLM

from blip_weights import blip_state_dict


def word_ids(text, vocab, lo=3):
    return [(sum(ord(c) * (i + 1) for i, c in enumerate(w)) % (vocab - lo)) + lo for w in text.split()]


class ToyTokenizer:
    """The call shape of an HF tokenizer: `tok(prompts, padding='longest', truncation=..., max_length=..., return_tensors='pt')` ->
    (.input_ids, .attention_mask), `padding_side`, `batch_decode(ids, skip_special_tokens=True)`."""

    def __init__(self, vocab, bos=None, eos=None, pad=0, padding_side="right"):
        self.vocab, self.bos, self.eos, self.pad, self.padding_side = vocab, bos, eos, pad, padding_side

    def encode(self, text):
        return ([self.bos] if self.bos is not None else []) + word_ids(text, self.vocab) + ([self.eos] if self.eos is not None else [])

    def __call__(self, prompts, padding="longest", truncation=False, max_length=None, return_tensors="pt"):
        rows = [self.encode(p) for p in ([prompts] if isinstance(prompts, str) else prompts)]
        if truncation and max_length is not None:
            rows = [r[:max_length] for r in rows]
        L = max(len(r) for r in rows)
        ids, mask = torch.full((len(rows), L), self.pad, dtype=torch.long), torch.zeros(len(rows), L, dtype=torch.long)
        for i, r in enumerate(rows):
            sl = slice(L - len(r), L) if self.padding_side == "left" else slice(0, len(r))
            ids[i, sl], mask[i, sl] = torch.tensor(r), 1
        return SimpleNamespace(input_ids=ids, attention_mask=mask, to=lambda *_a, **_k: SimpleNamespace(input_ids=ids, attention_mask=mask))

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(f"t{int(t)}" for t in row.tolist() if not (skip_special_tokens and int(t) < 3)) for row in ids]


class BlipDouble(nn.Module):
    def __init__(self, bcfg, llm, sd):
        super().__init__()
        self.llm_model = llm
        self._lavis = sd
        self.bcfg = bcfg
        self.tokenizer = ToyTokenizer(bcfg.qf.vocab, bos=101, eos=102, pad=0)          # BERT: [CLS] ... [SEP], right-padded
        self.llm_tokenizer = ToyTokenizer(llm.config.vocab_size, bos=1, pad=0, padding_side="right")
        self.max_txt_len, self.prompt, self.qformer_text_input = 32, "", True
        self.visual_encoder = SimpleNamespace(blocks=[SimpleNamespace(attn=SimpleNamespace(num_heads=bcfg.vit.heads))])
        self.Qformer = SimpleNamespace(config=SimpleNamespace(num_attention_heads=bcfg.qf.heads, cross_attention_freq=bcfg.qf.cross_freq))

    def state_dict(self, *a, **k):
        out = dict(self._lavis)
        out.update({"llm_model." + n: t for n, t in self.llm_model.state_dict().items()})
        return out


def build_blip(device, dtype, seed=0, lm_head_gain=6.0):
    """Tiny InstructBLIP: the `tiny` towers of blip_frontend.tiny_blip_config (88-wide ViT heads, cross-attention every 2nd layer) in front
    of a 2-layer Llama of width 256 (2 heads of 128)."""
    from llava_align_amd.blip_frontend import tiny_blip_config
    bcfg = tiny_blip_config()
    torch.manual_seed(seed)
    cfg = LlamaConfig(vocab_size=1000, hidden_size=bcfg.d_llm, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
                      num_key_value_heads=2, head_dim=128, rms_norm_eps=1e-5, max_position_embeddings=512, attention_bias=False, mlp_bias=False,
                      tie_word_embeddings=False, pad_token_id=0, eos_token_id=2, bos_token_id=1)
    with torch.device(device):
        llm = LlamaForCausalLM(cfg)
    with torch.no_grad():
        llm.lm_head.weight.mul_(lm_head_gain)
        llm.model.embed_tokens.weight.mul_(15.0)               # token embeddings on the scale of the Q-Former rows they sit beside
    llm = llm.to(dtype=dtype).eval()
    llm.generation_config = GenerationConfig(eos_token_id=2, pad_token_id=0, bos_token_id=1, top_k=50)
    sd = {k: v.to(device=device, dtype=dtype) for k, v in blip_state_dict(bcfg, seed=11 + seed).items()}
    return BlipDouble(bcfg, llm, sd)


class _QwenAttn(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.c_attn, self.c_proj = nn.Linear(d, 3 * d), nn.Linear(d, d, bias=False)


class _QwenMLP(nn.Module):
    def __init__(self, d, ffn):
        super().__init__()
        self.w1, self.w2, self.c_proj = nn.Linear(d, ffn, bias=False), nn.Linear(d, ffn, bias=False), nn.Linear(ffn, d, bias=False)


class _RMS(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))


class _QwenBlock(nn.Module):
    def __init__(self, d, ffn):
        super().__init__()
        self.ln_1, self.attn, self.ln_2, self.mlp = _RMS(d), _QwenAttn(d), _RMS(d), _QwenMLP(d, ffn)


class _Visual(nn.Module):
    """Stand-in for Qwen-VL's ViT + resampler (out of scope): image tensor [n, 3, S, S] -> [n, rows, d]."""

    def __init__(self, d, rows, image):
        super().__init__()
        self.rows, self.proj = rows, nn.Linear(3 * image * image // rows, d)

    def forward(self, images):
        n = images.shape[0]
        return self.proj(images.reshape(n, self.rows, -1).to(self.proj.weight.dtype))


class QwenDouble(nn.Module):
    def __init__(self, d=256, layers=2, heads=2, ffn=512, vocab=1200, img_rows=8, image=16, seq_length=512):
        super().__init__()
        self.transformer = nn.Module()
        self.transformer.wte = nn.Embedding(vocab, d)
        self.transformer.h = nn.ModuleList([_QwenBlock(d, ffn) for _ in range(layers)])
        self.transformer.ln_f = _RMS(d)
        self.transformer.visual = _Visual(d, img_rows, image)
        self.lm_head = nn.Linear(d, vocab, bias=False)
        self.img_rows = img_rows
        self.config = SimpleNamespace(hidden_size=d, num_hidden_layers=layers, num_attention_heads=heads, kv_channels=d // heads,
                                      intermediate_size=2 * ffn, layer_norm_epsilon=1e-6, rotary_emb_base=10000.0, rotary_pct=1.0,
                                      seq_length=seq_length, no_bias=True, vocab_size=vocab, use_dynamic_ntk=True, use_logn_attn=True,
                                      visual={"image_start_id": vocab - 10}, _name_or_path="qwen-double")
        self.generation_config = GenerationConfig(eos_token_id=vocab - 20, pad_token_id=vocab - 20, top_k=0, do_sample=True)

    def image_prompt(self, text_ids):
        """'<img>' + img_rows pad slots + '</img>' + the question's ids (run_qwen.py:176-177 as the tokenizer expands it)."""
        st = self.config.visual["image_start_id"]
        return [st] + [st + 2] * self.img_rows + [st + 1] + list(text_ids)


def build_qwen(device, dtype, seed=0, lm_head_gain=6.0):
    torch.manual_seed(seed)
    with torch.device(device):
        m = QwenDouble()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.06)
        for b in m.transformer.h:
            b.attn.c_attn.bias.normal_(0, 0.1)
        m.lm_head.weight.mul_(lm_head_gain / 3)
    return m.to(dtype=dtype).eval()
