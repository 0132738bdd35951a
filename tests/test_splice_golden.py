"""The multimodal splice against the reference's OWN function: tests/golden/splice.npz holds what
`LlavaMetaForCausalLM.prepare_inputs_labels_for_multimodal` / `encode_images` (experiments/llava/model/llava_arch.py:82-204) returned on
seeded inputs (made by tests/golden/make_splice_golden.py in the build container, the module loaded by path).  Checked here: the
splice of tests/ref_llava.py - the fp32 LLaVA every engine test is compared with - and of tests/hf_llava.py, the branch-input rules
the engine and the drop-in loop rely on (image-free branches and decode steps pass their ids through untouched; the mask of an
un-padded question stays all ones, left-extended by the patch count), and - on the GPU - the engine's own packed prefill matrix."""
import os
import types

import numpy as np
import pytest
import torch

import hf_llava
import ref_llava

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "splice.npz"))
T = lambda k: torch.from_numpy(G[k])
IMG = -200


def test_fixture_covers_the_paths_of_the_reference_function():
    assert int(G["main.embeds_returned"]) == 1 and int(G["main.ids_returned"]) == 0           # :204 returns (None, mask, past, embeds, labels)
    for tag in ("unk", "none", "decode"):                                                     # :91-94 early return: ids untouched, no embeds
        assert int(G[tag + ".ids_returned"]) == 1 and int(G[tag + ".embeds_returned"]) == 0
    assert G["encode_images"].shape == (3, 5, 8)
    proj = T("feats") @ T("proj_w").t() + T("proj_b")                                         # :82-85: projector(tower(images))
    assert torch.allclose(T("encode_images"), proj, atol=1e-6)


def test_ref_llava_and_hf_llava_splice_like_the_reference():
    table, feat = T("table"), T("encode_images")
    row = T("ids_main")[0]
    want = T("main.embeds")[0]
    assert want.shape[0] == row.numel() - 1 + feat.shape[1]                                   # the slot is REPLACED by the patches
    close = lambda a, b: a.shape == b.shape and torch.allclose(a, b, rtol=0, atol=1e-6)      # (the projector ran on 1 vs 3 images: last-bit fp32 noise)
    assert close(ref_llava.splice(table, row, feat[1]), want)
    s = int(torch.where(row == IMG)[0][0])
    assert torch.equal(ref_llava.splice(table, row, want[s: s + feat.shape[1]]), want)       # rows in the reference's order, bit for bit
    emb = torch.nn.Embedding.from_pretrained(table)
    assert close(hf_llava.splice(emb, row, feat[1]), want)
    for q, im in ((0, 0), (1, 2)):                                                            # a batch of two questions, one image each
        assert close(ref_llava.splice(table, T("ids_b2")[q], feat[im]), T("batch2.embeds")[q])
    # images given but no slot in the row (:106-117): plain token embeddings, nothing spliced
    unk = T("unk.ids")[0]
    assert torch.equal(T("no_slot.embeds")[0], table[unk])


def test_branch_ids_and_masks_follow_the_reference():
    ids = T("ids_main")
    unk = ids.clone(); unk[unk == IMG] = 0                                                    # vcd_sample.py:154-155
    assert torch.equal(T("unk.ids"), unk) and torch.equal(T("none.ids"), ids[ids != IMG][None])   # :160
    # un-padded single questions: every mask the function returns is all ones (what ref_llava / hf_llava / the engine assume at B = 1)
    for tag in ("main", "unk", "none", "decode", "batch2", "no_slot"):
        assert (G[tag + ".mask"] == 1).all(), tag
    assert G["main.mask"].shape[1] == G["main.embeds"].shape[1] == ids.shape[1] + 4           # left-extended by (patches - 1), :199-202
    assert G["decode.mask"].shape == (1, 13) and torch.equal(T("decode.ids"), torch.tensor([[5]]))   # :92-93: past length + 1


@pytest.mark.gpu
def test_engine_packs_the_prefill_matrix_like_the_reference_splices():
    """VddLlavaEngine._plan + _pack (prefix = [tokens before the slot | patches], suffix = the tokens behind it) lay out exactly the
    rows the reference's function concatenates - with and without prefix sharing."""
    from llava_align_amd.engine import LlavaConfig, LlavaWeights, LMConfig, VddLlavaEngine, VisionConfig
    dev = "cuda:0"
    cfg = LlavaConfig(LMConfig(d=8, vocab=50, n_layers=0), VisionConfig(), "splice")
    for dt in (torch.bfloat16, torch.float16):
        table, P = T("table").to(dt), G["encode_images"].shape[1]
        eng = VddLlavaEngine.__new__(VddLlavaEngine)                                           # only what _plan / _pack touch
        eng.cfg, eng.device, eng.dtype = cfg, torch.device(dev), dt
        eng.w = LlavaWeights(cfg, dev, dt)
        eng.w.t["embed"] = table.to(dev)
        for ids, tag in ((T("ids_main"), "main"), (T("ids_b2"), "batch2")):
            rows = [r.tolist() for r in ids]
            # the patch features of question q = the rows the reference itself put behind the slot
            feats = [T(tag + ".embeds")[q][r.index(IMG): r.index(IMG) + P].to(dt).to(dev) for q, r in enumerate(rows)]
            for share in (True, False):
                plan = eng._plan([("main", rows, feats)], P, share)
                got = {}
                for phase in ("prefix", "suffix"):
                    if plan[phase]:
                        x, pos, cpos, slot, seqs, _ = eng._pack(plan[phase])
                        for s in plan[phase]:
                            got[(phase, s["slot"])] = (x[s["q_row0"]: s["q_row0"] + s["T"]].cpu(), s)
                for q in range(len(rows)):
                    suf, s = got[("suffix", q)]
                    full = torch.cat([got[("prefix", s["pslot"])][0], suf]) if s["plen"] > 0 else suf
                    want = T(tag + ".embeds")[q].to(dt)                                        # gather / copy only: exact in any dtype
                    assert torch.equal(full, want), (tag, q, share)
