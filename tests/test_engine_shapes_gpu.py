"""The engine against the fp32 reference LLaVA at REAL model dimensions (the `tiny` presets of test_engine_gpu.py exercise every
code path but at d = 256 / 2 heads):
  * LLaVA-1.5-7B widths (d 4096, 32 heads, ffn 11008, CLIP width 1024 / 336 px -> 611-token shared prefixes), 2 decoder layers,
    8 images x 6 questions x 2 branches = 96 decode rows: the grouped MFMA prefix pass with several 64-key chunks per work item,
    the own-token merge pass, the persistent GEMM at decode and prefill sizes and the captured HIP graph all run at their real
    head counts and sequence lengths;
  * LLaVA-1.5-13B widths (d 5120, 40 heads, ffn 13824; BASELINE config #3: use_dd + use_dd_unk = 3 branches, top-p 0.9, T = 1),
    2 decoder layers.
Reference = tests/ref_llava.py in fp32 driven by the oracle restatement of the reference loop, one question at a time."""
import numpy as np
import pytest
import torch

from oracle import vdd_oracle as O
from ref_llava import RefLlava

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine(lm_kw, n_layers=2, vit_layers=3, use_graph=True, seed=5):
    from llava_align_amd.engine import LlavaConfig, LlavaWeights, LMConfig, VddLlavaEngine, VisionConfig
    cfg = LlavaConfig(LMConfig(n_layers=n_layers, max_pos=1024, **lm_kw), VisionConfig(layers=vit_layers), "shapes-test")
    w = LlavaWeights.random(cfg, DEV, seed=seed, std=0.02, lm_head_gain=2.0)
    return VddLlavaEngine(cfg, weights=w, device=DEV, use_graph=use_graph)


def _prompts(n_img, per_img, vocab, seed, image=336):
    rng = np.random.default_rng(seed)
    sys_tok = [1] + rng.integers(3, vocab, size=34).tolist()
    ids, imgs = [], []
    for i in range(n_img):
        im = torch.randn(3, image, image, generator=torch.Generator().manual_seed(500 + i))
        for _ in range(per_img):
            ids.append(torch.tensor(sys_tok + [-200] + rng.integers(3, vocab, size=int(rng.integers(19, 29))).tolist()))
            imgs.append(im)
    return ids, imgs


def _compare(eng, ref, ids, imgs, mode_kw, warp_kw, n_new, questions, tol):
    out = eng.generate(ids, images=imgs, cd_alpha=1.0, cd_beta=0.1, max_new_tokens=n_new, cd_greedy=True, output_scores=True, **mode_kw, **warp_kw)
    checked = 0
    for q in questions:
        kw = dict(images=imgs[q][None], attention_mask=torch.ones(1, ids[q].numel(), dtype=torch.long), use_cache=True, cd_alpha=1.0,
                  cd_beta=0.1, **mode_kw)
        r = O.reference_loop(ref, ids[q][None].clone(), warp=O.WarpConfig(**warp_kw), max_length=ids[q].numel() + n_new, pad_token_id=None,
                             eos_token_id=None, pick=O.pick_argmax, **kw)
        want, got = r.sequences[0, ids[q].numel():].tolist(), out.tokens[q].tolist()
        for step in range(n_new):
            s_got, s_want = out.scores[step][q].float().cpu(), r.scores[step][0].float().cpu()
            fin = torch.isfinite(s_got) & torch.isfinite(s_want)
            # tokens whose main-branch logit (or cumulative mass) sits within bf16 noise of a cutoff may fall on either side
            flip = torch.isfinite(s_got) ^ torch.isfinite(s_want)
            assert fin.sum() >= 1, (q, step)
            if flip.sum() > 3 + 0.25 * int(fin.sum()):               # candidate density x logit noise
                # top-p on a knife edge (the kept mass crosses p within noise of a token boundary): one side stops, the other
                # goes on through the flat tail - whatever it adds carries at most the (1 - p) mass top-p may remove
                assert "top_p" in warp_kw, (q, step)
                big = s_got if torch.isfinite(s_got).sum() > torch.isfinite(s_want).sum() else s_want
                assert torch.softmax(big, -1)[flip].sum().item() <= (1.0 - warp_kw["top_p"]) + 0.02, (q, step)
            tol_s = tol + 0.02 * s_want[fin].abs().max().item()                  # the scores are bf16: 2-3 ulps of their own magnitude
            assert (s_got[fin] - s_want[fin]).abs().max().item() <= tol_s, (q, step)
            top2 = torch.topk(s_want, 2).values
            # (a candidate whose main-branch logit sits within noise of the plausibility cutoff is kept by one side only - the `flip`
            #  entries above - and may carry the largest contrasted score: then the picks differ although every common score agrees)
            edge = bool(flip[got[step]]) or bool(flip[want[step]])
            if (top2[0] - top2[1]).item() > 2 * tol_s and not edge:
                assert got[step] == want[step], (q, step)
                checked += 1
            if got[step] != want[step]:
                break
    return out, checked


def test_llava_7b_widths_grouped_decode_and_graph_match_reference():
    eng = _engine(dict(d=4096, n_heads=32, n_kv_heads=32, head_dim=128, ffn=11008, vocab=32000))
    ref = RefLlava(eng.w, device=DEV)
    ids, imgs = _prompts(8, 6, 32000, seed=21)                       # 48 questions, 96 rows, 8 + 1 shared prefixes of 611 / 36 tokens
    # bf16 engine vs fp32 reference; the contrast multiplies logit noise by ((1+a) + a) / T = 6
    out, checked = _compare(eng, ref, ids, imgs, dict(use_dd_unk=True), dict(temperature=0.5), n_new=4, questions=range(0, 48, 5), tol=0.5)
    assert out.stats["graph"] and out.stats["n_rows"] == 96 and checked >= 10
    assert out.stats["prefill_tokens"] < 0.3 * out.stats["unshared_prefill_tokens"]
    # same batch through the eager step: the captured graph replays exactly that
    eng2 = _engine(dict(d=4096, n_heads=32, n_kv_heads=32, head_dim=128, ffn=11008, vocab=32000), use_graph=False)
    o2 = eng2.generate(ids, images=imgs, cd_alpha=1.0, cd_beta=0.1, max_new_tokens=4, cd_greedy=True, use_dd_unk=True, temperature=0.5)
    assert torch.equal(o2.tokens, out.tokens)


def test_llava_13b_widths_three_branches_top_p_match_reference():
    eng = _engine(dict(d=5120, n_heads=40, n_kv_heads=40, head_dim=128, ffn=13824, vocab=32000))
    ref = RefLlava(eng.w, device=DEV)
    ids, imgs = _prompts(6, 1, 32000, seed=22)                       # BASELINE config #3: one image per question, nothing to group
    out, checked = _compare(eng, ref, ids, imgs, dict(use_dd=True, use_dd_unk=True), dict(top_p=0.9), n_new=4, questions=range(6), tol=0.25)
    assert out.stats["n_rows"] == 18 and checked >= 6


def test_qwen_7b_widths_bias_epilogue_and_151936_vocab_match_reference():
    """BASELINE config #4 at Qwen-7B LM widths (d 4096, 32 heads, qkv bias, V = 151,936: the 151936 x 4096 lm_head through the MFMA
    GEMM with 594 column tiles, the bias epilogue at N = 12,288, the sampling kernel's global-row path), 2 layers, 24 questions x 2
    branches = 48 rows.  Prompts arrive as embeddings (256 image slots + text, run_qwen.py:176-177); use_dd_unk re-runs the same
    inputs in the second branch (modeling_qwen.py:1089-1118); min_new_tokens = 1 and pad = eos = eod (run_qwen.py:190-213)."""
    from ref_llava import RefLavisLM
    eng = _engine(dict(d=4096, n_heads=32, n_kv_heads=32, head_dim=128, ffn=11008, vocab=151936, qkv_bias=True, eps=1e-6))

    class RefQwenLM(RefLavisLM):
        def prepare_inputs_for_generation_cd(self, input_ids, **kw):        # Qwen: the cd branch gets the SAME inputs
            return self.prepare_inputs_for_generation(input_ids, **kw)
    ref = RefQwenLM(eng.w, device=DEV)
    g = torch.Generator().manual_seed(77)
    embs = [torch.randn(256 + int(n), 4096, generator=g) * 0.02 for n in torch.randint(30, 50, (24,), generator=g)]
    eod, n_new = 151643, 3
    out = eng.generate(None, inputs_embeds=embs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, max_new_tokens=n_new,
                       min_new_tokens=1, eos_token_id=eod, pad_token_id=eod, cd_greedy=True, output_scores=True, sync_every=1)
    assert out.stats["n_rows"] == 48 and out.stats["graph"] and torch.isneginf(out.scores[0][:, eod]).all()
    checked = 0
    for q in range(0, 24, 2):
        kw = dict(inputs_embeds=embs[q][None], attention_mask=torch.ones(1, embs[q].shape[0], dtype=torch.long), use_cache=True,
                  cd_alpha=1.0, cd_beta=0.1, use_dd_unk=True)
        r = O.reference_loop(ref, torch.zeros(1, 0, dtype=torch.long), warp=O.WarpConfig(temperature=1.0), max_length=n_new, pad_token_id=eod,
                             eos_token_id=eod, pick=O.pick_argmax, processors=O.ProcessorList([O.MinNewTokens(0, 1, [eod])]), **kw)
        for step in range(len(r.scores)):
            s_got, s_want = out.scores[step][q].float().cpu(), r.scores[step][0].float().cpu()
            fin = torch.isfinite(s_got) & torch.isfinite(s_want)
            assert fin.sum() >= 1 and (torch.isfinite(s_got) ^ torch.isfinite(s_want)).sum() <= 3 + 0.25 * int(fin.sum()), (q, step)
            tol_s = 0.25 + 0.02 * s_want[fin].abs().max().item()
            assert (s_got[fin] - s_want[fin]).abs().max().item() <= tol_s, (q, step)
            top2 = torch.topk(s_want, 2).values
            if (top2[0] - top2[1]).item() > 2 * tol_s:
                assert out.tokens[q, step].item() == r.sequences[0, step].item(), (q, step)
                checked += 1
            if out.tokens[q, step].item() != r.sequences[0, step].item():
                break
    assert checked >= 4


@pytest.mark.parametrize("n_q", [1, 5, 8])
def test_few_row_layer_at_7b_widths_equals_the_unfused_layer(n_q):
    """One / five / eight questions in flight at LLaVA-1.5-7B widths (2, 10, 16 rows): the few-row decode layer - projections that
    normalise their input once per workgroup (two 4-wave blocks per CU at 2 rows, one 8-wave block at 10 and 16), residual + sum of
    squares in the d-wide projections, the one-launch attention with its keys cut over 4 workgroups (2 rows) - against the layer of
    separate RMSNorm / projection / attention launches on the same weights."""
    eng = _engine(dict(d=4096, n_heads=32, n_kv_heads=32, head_dim=128, ffn=11008, vocab=32000))
    ids, imgs = _prompts((n_q + 5) // 6, 6, 32000, seed=23)
    ids, imgs = ids[:n_q], imgs[:n_q]
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=6, cd_greedy=True, output_scores=True)
    a = eng.generate(ids, **kw)
    from llava_align_amd import ops
    try:
        eng.lm.fuse_norms, ops.FUSED_ATTN_SPLIT = False, False
        b = eng.generate(ids, **kw)
    finally:
        eng.lm.fuse_norms, ops.FUSED_ATTN_SPLIT = True, True
    for step, (sa, sb) in enumerate(zip(a.scores, b.scores)):
        fin = torch.isfinite(sa) & torch.isfinite(sb)
        assert (torch.isfinite(sa) ^ torch.isfinite(sb)).sum() <= 2 * n_q, step
        assert (sa[fin].float() - sb[fin].float()).abs().max().item() <= 0.25, step
    assert (a.tokens == b.tokens).float().mean().item() >= 0.75


def test_unseen_shape_gets_measured_forms_a_monotone_step_curve_and_the_reference_tokens():
    """A shape nobody tuned a constant for: d = 8192, 64 heads with 8 KV heads (GQA), ffn 28672 - Llama-70B-like widths, 2
    layers.  Every projection / layer form is measured at first use (ops._pick_form) instead of inheriting LLaVA-1.5-7B's crossovers: the
    decode step must then never get markedly SLOWER by dropping rows (what a wrong crossover looks like: 9.0 -> 10.5 ms from 15 to 24
    rows at 13B widths in round 4), and the tokens are the fp32 reference's."""
    from llava_align_amd import ops
    eng = _engine(dict(d=8192, n_heads=64, n_kv_heads=8, head_dim=128, ffn=28672, vocab=32000), n_layers=2, vit_layers=2)
    ids, imgs = _prompts(64, 1, 32000, seed=23)
    curve = []
    for nq in (1, 2, 4, 8, 16, 32, 64):                                   # x 2 branches = 2 ... 128 rows, one image per question (ungrouped)
        kw = dict(images=imgs[:nq], use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, max_new_tokens=24, seed=1)
        best = 1e9
        for rep in range(4):
            eng.call_log = []
            eng.generate(ids[:nq], **kw)
            t = eng.call_timing(eng.call_log[-1])
            if rep:
                best = min(best, t["decode_ms"] / t["decode_steps"])
        curve.append((2 * nq, best))
    eng.call_log = None
    for (r0, t0), (r1, t1) in zip(curve, curve[1:]):
        assert t1 >= 0.93 * t0, curve                                     # monotone up to timing noise
    assert curve[-1][1] <= 4.0 * curve[0][1], curve                        # and flat-ish: 64 x the rows cost a few x the time (weights dominate)
    dt = ops._MODEL_DT[eng.dtype]
    picked = {k: v for k, v in ops._form_choice.items() if k[2] in (8192, 10240, 57344, 32000) or (k[0] == "layer" and k[2] == 8192)}
    assert any(k[0] == "to_norm" and k[2] == 8192 and k[3] == 28672 for k in picked) and any(k[0] == "swiglu" for k in picked), sorted(picked)
    assert any(k[0] == "layer" and k[2] == 8192 for k in picked)           # the few-row layer form of THIS width was measured too
    ref = RefLlava(eng.w, device=DEV)
    out, checked = _compare(eng, ref, ids[:6], imgs[:6], dict(use_dd_unk=True), dict(temperature=0.5), n_new=4, questions=range(6), tol=0.5)
    assert out.stats["n_rows"] == 12 and checked >= 6
