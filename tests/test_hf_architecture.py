"""Ties the chain of references to HuggingFace's own modules (CPU, fp32, random init):

    HF LlamaForCausalLM + CLIPVisionModel + mlp2x_gelu projector, composed the way the reference composes them
    (llava_arch.py:82-204, clip_encoder.py:39-51, builder.py:33-46)
        ==  tests/ref_llava.RefLlava on weights mapped by LlavaWeights.from_state_dict      (this file)
        ==  the HIP engine on the same weights                                               (tests/test_engine_gpu.py)

so the engine's architecture (RoPE pairing, norm placement/eps, quick_gelu, CLS + position embeddings,
hidden_states[-2] without CLS, splice order) and the checkpoint-name mapping are checked against the real thing."""
import torch
from transformers import CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM

from llava_align_amd.engine import LlavaWeights, preset
from ref_llava import RefLlava


def build():
    torch.manual_seed(0)
    cfg = preset("tiny")
    lm, v = cfg.lm, cfg.vision
    llama = LlamaForCausalLM(LlamaConfig(vocab_size=lm.vocab, hidden_size=lm.d, intermediate_size=lm.ffn, num_hidden_layers=lm.n_layers,
                                         num_attention_heads=lm.n_heads, num_key_value_heads=lm.n_kv_heads, head_dim=lm.head_dim,
                                         rms_norm_eps=lm.eps, rope_theta=lm.rope_theta, max_position_embeddings=lm.max_pos,
                                         attention_bias=False, mlp_bias=False, tie_word_embeddings=False)).eval()
    clip = CLIPVisionModel(CLIPVisionConfig(hidden_size=v.width, intermediate_size=v.mlp, num_hidden_layers=v.layers,
                                            num_attention_heads=v.heads, image_size=v.image, patch_size=v.patch,
                                            hidden_act="quick_gelu", layer_norm_eps=v.eps)).eval()
    proj = torch.nn.Sequential(torch.nn.Linear(v.width, lm.d), torch.nn.GELU(), torch.nn.Linear(lm.d, lm.d)).eval()
    for m in (llama, clip, proj):
        for p in m.parameters():
            p.data.mul_(2.0)                     # away from the tiny default init: make every path contribute
    sd = {("model." + k[len("model."):] if k.startswith("model.") else k): t for k, t in llama.state_dict().items()}
    sd.update({"model.mm_projector.0.weight": proj[0].weight, "model.mm_projector.0.bias": proj[0].bias,
               "model.mm_projector.2.weight": proj[2].weight, "model.mm_projector.2.bias": proj[2].bias})
    return cfg, llama, clip, proj, sd


def test_checkpoint_mapping_and_reference_match_hf_modules():
    cfg, llama, clip, proj, sd = build()
    w = LlavaWeights.from_state_dict(cfg, sd, "cpu", vision_sd=clip.state_dict(), dtype=torch.float32)
    ref = RefLlava(w, device="cpu", logit_dtype=torch.float32, dtype=torch.float32)
    # also accept the tower stored inside the checkpoint under LLaVA's prefix
    sd2 = dict(sd)
    sd2.update({"model.vision_tower.vision_tower." + k: t for k, t in clip.state_dict().items()})
    w2 = LlavaWeights.from_state_dict(cfg, sd2, "cpu", dtype=torch.float32)
    assert all(torch.equal(w.t[k], w2.t[k]) for k in w.t)

    img = torch.randn(2, 3, cfg.vision.image, cfg.vision.image).to(torch.bfloat16).float()    # the reference path feeds bf16-rounded pixels
    with torch.no_grad():
        # clip_encoder.py:39-51: hidden_states[select_layer = -2], drop CLS; builder.py: mlp2x_gelu
        hs = clip(pixel_values=img, output_hidden_states=True).hidden_states[-2][:, 1:]
        feat_hf = proj(hs)
        feat_ref = ref.encode_images(img)
    assert torch.allclose(feat_ref, feat_hf, rtol=1e-4, atol=1e-4), (feat_ref - feat_hf).abs().max()

    ids = torch.tensor([[1, 17, 250, 33, -200, 400, 401, 77, 12]])
    s = 4
    with torch.no_grad():
        emb = llama.get_input_embeddings()
        x = torch.cat([emb(ids[0, :s]), feat_hf[0], emb(ids[0, s + 1:])], 0)[None]               # llava_arch.py:142-158
        x = x.to(torch.bfloat16).float()
        out_hf = llama(inputs_embeds=x, use_cache=True)
        out_ref = ref(input_ids=ids, images=img[:1])
        assert torch.allclose(out_ref.logits, out_hf.logits, rtol=2e-3, atol=2e-3), (out_ref.logits - out_hf.logits).abs().max()
        # one cached decode step
        nxt = torch.tensor([[55]])
        r16 = lambda t: t.to(torch.bfloat16).float()                 # the reference path (and the engine) hold embeddings in bf16
        step_hf = llama(inputs_embeds=r16(emb(nxt)), past_key_values=out_hf.past_key_values, use_cache=True).logits
        step_ref = ref(input_ids=nxt, images=img[:1], past_key_values=out_ref.past_key_values).logits
        assert torch.allclose(step_ref, step_hf, rtol=2e-3, atol=2e-3)
        # image-free branch: <unk> (token 0) embedded as one ordinary token
        unk = ids.clone(); unk[unk == -200] = 0
        assert torch.allclose(ref(input_ids=unk, images=None).logits, llama(inputs_embeds=r16(emb(unk))).logits, rtol=2e-3, atol=2e-3)
