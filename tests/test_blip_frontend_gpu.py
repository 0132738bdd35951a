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


# ---- against OUTPUTS OF THE REFERENCE'S LAVIS MODULES (tests/golden/blip_vectors.npz, see tests/test_blip_golden.py), at the tiny
# size and at the published EVA-ViT-g / Q-Former widths (1408 = 16 x 88 padded to 128-wide heads, MLP 6144, 257 tokens; 768 = 12 x 64,
# FFN 3072, 32 queries, vocabulary 30523; 2 + 3 layers)
import os

import numpy as np

from blip_weights import blip_inputs, blip_state_dict, cases

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "blip_vectors.npz"))


@pytest.mark.parametrize("name", list(cases()))
def test_front_end_matches_the_lavis_fixture(name):
    from llava_align_amd.blip_frontend import BlipWeights, InstructBlipFrontEnd
    mk, wseed, iseed, n = cases()[name]
    cfg = mk()
    sd = blip_state_dict(cfg, wseed)
    imgs, text = blip_inputs(cfg, iseed, n)
    fe = InstructBlipFrontEnd(BlipWeights.from_state_dict(cfg, sd, DEV))

    def check(got, want, what, rel=0.04):
        want = torch.from_numpy(np.asarray(want)).to(got.device)
        err, scale = (got.float() - want).abs().max().item(), want.abs().max().item()
        assert cos(got, want) > 0.9995 and err <= rel * scale, (what, cos(got, want), err, scale)

    ie = fe.image_embeds(imgs.to(DEV))
    assert tuple(ie.shape) == (n, cfg.vit.n_tokens, cfg.vit.width)
    check(ie[:, [0, 1, -1]], GOLD[f"{name}.image_embeds_rows"], "image_embeds rows")
    want_abs = torch.from_numpy(GOLD[f"{name}.image_embeds_rowabs"]).to(DEV)
    assert ((ie.float().abs().sum(-1) - want_abs).abs() <= 0.01 * want_abs).all()              # every token row, via its L1 norm
    hq = fe.qformer(ie, text)
    check(hq, GOLD[f"{name}.query_out"], "query_out")
    check(fe.qformer(ie, None), GOLD[f"{name}.query_out_notext"], "query_out_notext")
    il = fe.inputs_llm(imgs.to(DEV), text)
    assert tuple(il.shape) == (n, cfg.qf.n_query, cfg.d_llm)
    check(il[:, :, :64], GOLD[f"{name}.inputs_llm_head"], "inputs_llm head")
    want_abs = torch.from_numpy(GOLD[f"{name}.inputs_llm_rowabs"]).to(DEV)
    assert ((il.float().abs().sum(-1) - want_abs).abs() <= 0.02 * want_abs).all()
