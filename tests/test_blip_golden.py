"""Pins tests/ref_blip.py (the fp32 restatement the GPU front-end tests compare against) to OUTPUTS OF THE REFERENCE'S OWN LAVIS
MODULES: tests/golden/blip_vectors.npz was written by tests/golden/make_blip_golden.py, which executes
experiments/lavis/models/eva_vit.py (VisionTransformer) and blip2_models/Qformer.py (BertModel with queries + text +
cross-attention) unmodified, composed as blip2_vicuna_instruct.py:333-366 composes them.  Weights / inputs are regenerated from
numpy RandomState seeds (tests/blip_weights.py), so only outputs are stored.  CPU, fp32, tolerance = fp32 summation-order noise."""
import os

import numpy as np
import pytest
import torch

import ref_blip
from blip_weights import blip_inputs, blip_state_dict, cases

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "blip_vectors.npz"))
ROWS = (0, 1, -1)


def close(got, want, tol=2e-4):
    want = torch.from_numpy(np.asarray(want))
    scale = max(1.0, float(want.abs().max()))
    err = float((got.float() - want).abs().max())
    assert err <= tol * scale, (err, scale)


@pytest.fixture(scope="module", params=list(cases()))
def case(request):
    mk, wseed, iseed, n = cases()[request.param]
    cfg = mk()
    sd = blip_state_dict(cfg, wseed)
    imgs, text = blip_inputs(cfg, iseed, n)
    return request.param, cfg, sd, imgs, text


def test_fixture_cases_cover_real_eva_and_qformer_widths():
    cfg = cases()["real_widths"][0]()
    assert (cfg.vit.width, cfg.vit.heads, cfg.vit.head_dim, cfg.vit.mlp, cfg.vit.n_tokens) == (1408, 16, 88, 6144, 257)
    assert (cfg.qf.hidden, cfg.qf.heads, cfg.qf.inter, cfg.qf.n_query, cfg.qf.vocab, cfg.qf.cross_freq) == (768, 12, 3072, 32, 30523, 2)


def test_ref_blip_matches_the_lavis_modules(case):
    name, cfg, sd, imgs, text = case
    with torch.no_grad():
        ie = ref_blip.eva_vit(sd, cfg, imgs)
        close(ie[:, list(ROWS)], GOLD[f"{name}.image_embeds_rows"])
        close(ie.sum(-1), GOLD[f"{name}.image_embeds_rowsum"], tol=2e-4 * 8)
        close(ie.abs().sum(-1), GOLD[f"{name}.image_embeds_rowabs"], tol=2e-4)
        hq = ref_blip.qformer(sd, cfg, ie, text)
        close(hq, GOLD[f"{name}.query_out"])
        close(ref_blip.qformer(sd, cfg, ie, None), GOLD[f"{name}.query_out_notext"])
        il = ref_blip.inputs_llm(sd, cfg, imgs, text)
        close(il[:, :, :64], GOLD[f"{name}.inputs_llm_head"])
        close(il.sum(-1), GOLD[f"{name}.inputs_llm_rowsum"], tol=2e-4 * 8)
        close(il.abs().sum(-1), GOLD[f"{name}.inputs_llm_rowabs"])


def test_text_and_padding_matter_in_the_fixture(case):
    """The stored outputs really depend on the instruction and on the padding mask (so a restatement that drops either fails
    the test above): text vs no text differ, and the shortest instruction sits in a right-padded batch."""
    name, cfg, sd, imgs, text = case
    assert np.abs(GOLD[f"{name}.query_out"] - GOLD[f"{name}.query_out_notext"]).max() > 0.05
    assert len(set(len(t) for t in text)) > 1
