"""InstructBLIP front-end (EVA-ViT with 88-wide heads -> Q-Former with text input and cross-attention -> llm_proj) against the
plain-torch fp32 restatement of the LAVIS modules (tests/ref_blip.py), then end to end into the engine's VCD decoding
(generate(inputs_embeds=..., images_cd=embeddings), blip2_vicuna_instruct.py:380-410)."""
import pytest
import torch

import ref_blip

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def cos(a, b):
    return torch.nn.functional.cosine_similarity(a.float().flatten(), b.float().flatten(), dim=0).item()


@pytest.fixture(scope="module")
def front():
    from llava_align_amd.blip_frontend import BlipWeights, InstructBlipFrontEnd, tiny_blip_config
    cfg = tiny_blip_config()
    return InstructBlipFrontEnd(BlipWeights.random(cfg, DEV, seed=4, std=0.05))


def test_eva_vit_with_padded_heads_matches_reference(front):
    imgs = torch.randn(3, 3, 56, 56, generator=torch.Generator().manual_seed(1))
    got = front.image_embeds(imgs).float()
    want = ref_blip.eva_vit(front.w.t, front.cfg, imgs.to(DEV))
    assert got.shape == want.shape == (3, 17, 256)
    assert cos(got, want) > 0.9995 and (got - want).abs().max().item() <= 0.05 * want.abs().max().item()


def test_qformer_text_input_cross_attention_and_projection_match_reference(front):
    imgs = torch.randn(4, 3, 56, 56, generator=torch.Generator().manual_seed(2))
    text = [[101, 7, 45, 300, 102], [101, 9, 102], [101, 11, 12, 13, 14, 15, 16, 102], [101, 102]]        # ragged instructions
    got = front.inputs_llm(imgs, text).float()
    want = ref_blip.inputs_llm(front.w.t, front.cfg, imgs.to(DEV), text)
    assert got.shape == want.shape == (4, 8, 256)
    assert cos(got, want) > 0.999 and (got - want).abs().max().item() <= 0.06 * want.abs().max().item()
    # the instruction really conditions the queries (a wrong mask / missing text stream would not)
    other = front.inputs_llm(imgs, [[101, 400, 401, 402, 102]] * 4).float()
    assert (other - got).abs().max().item() > 10 * (got - want).abs().max().item()


def test_front_end_feeds_vcd_decoding_of_the_engine(front):
    from llava_align_amd import add_diffusion_noise
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=False)
    imgs = torch.randn(3, 3, 56, 56, generator=torch.Generator().manual_seed(3)).to(DEV)
    noisy = torch.stack([add_diffusion_noise(im, 500, seed=9) for im in imgs])                 # blip_calibrate.py: the VCD branch input
    prompts = [[1, 20, 21, 22], [1, 30, 31], [1, 40, 41, 42, 43, 44]]
    text = [[101, 5, 102], [101, 6, 7, 102], [101, 8, 102]]
    emb, emb_cd = front.build(imgs, prompts, eng.w.t["embed"], qformer_text_ids=text, images_cd=noisy)
    assert [e.shape for e in emb] == [(8 + len(p), 256) for p in prompts] and len(emb_cd) == 3
    assert all(torch.equal(a[8:], b[8:]) and not torch.equal(a[:8], b[:8]) for a, b in zip(emb, emb_cd))      # same prompt, other image
    out = eng.generate(None, inputs_embeds=emb, images_cd=emb_cd, cd_alpha=0.5, cd_beta=0.1, temperature=1.0, max_new_tokens=5,
                       cd_greedy=True, output_scores=True)
    plain = eng.generate(None, inputs_embeds=emb, temperature=1.0, max_new_tokens=5, cd_greedy=True, output_scores=True)
    assert out.tokens.shape == (3, 5) and all(torch.isfinite(s).any(-1).all() for s in out.scores)
    # step 0 is contrasted against the noisy-image branch (masked entries appear), later steps have c == v
    assert torch.isinf(out.scores[0]).any() and not torch.isinf(plain.scores[0]).any()
