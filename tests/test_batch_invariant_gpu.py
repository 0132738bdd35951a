"""Batch-invariant mode (ops.GEMM_BATCH_INVARIANT / `with ops.batch_invariant():`): one arithmetic form per op, so a row's results
are a function of the row alone - the property behind SURVEY 8(e) "top-k=1 runs are shard-invariant" (the reference decodes every
question at B = 1, llava_calibrate.py:130: its answers cannot depend on the other questions of a list).

What is compared is STRONGER than tokens: the post-warp score rows of every step (output_scores) bit for bit, between a batch and
its sub-batches down to ONE question - i.e. across what are, with the tuned forms, four different kernel regimes (GEMM tile
schedules, weight-streaming projections, grouped / one-launch attention, packed suffix prefill)."""
import pytest
import torch

from test_engine_shapes_gpu import _engine, _prompts

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W7B = dict(d=4096, n_heads=32, n_kv_heads=32, head_dim=128, ffn=11008, vocab=32000)


def _gen(eng, ids, imgs, sel, **kw):
    return eng.generate([ids[i] for i in sel], images=[imgs[i] for i in sel], cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, output_scores=True,
                        max_new_tokens=12, **kw)


def _same(a, b, qa, qb):
    """question qa of run a == question qb of run b: tokens and every step's score row, bit for bit."""
    if not torch.equal(a.tokens[qa], b.tokens[qb]):
        return False
    return all(torch.equal(sa[qa].view(torch.int16), sb[qb].view(torch.int16)) for sa, sb in zip(a.scores, b.scores))


@pytest.mark.parametrize("mode", [dict(use_dd_unk=True, temperature=0.2), dict(use_dd=True, use_dd_unk=True, temperature=1.0, top_p=0.9)],
                         ids=["dd_unk", "both_top_p"])
def test_a_question_decodes_the_same_in_every_batch(mode):
    """24 questions (4 images x 6: shared image prefixes) against the runs of its first 20, of questions 6..8 and of question 13 alone."""
    from llava_align_amd import ops
    eng = _engine(W7B, n_layers=3, vit_layers=2)
    ids, imgs = _prompts(4, 6, 32000, seed=71)
    with ops.batch_invariant():
        full = _gen(eng, ids, imgs, range(24), **mode)
        part = _gen(eng, ids, imgs, range(20), **mode)
        three = _gen(eng, ids, imgs, [6, 7, 8], **mode)
        one = _gen(eng, ids, imgs, [13], **mode)
        again = _gen(eng, ids, imgs, range(24), **mode)
    assert all(_same(full, again, q, q) for q in range(24))                       # (run to run, first)
    assert all(_same(full, part, q, q) for q in range(20))
    assert all(_same(full, three, 6 + j, j) for j in range(3))
    assert _same(full, one, 13, 0)
    assert len({tuple(full.tokens[q].tolist()) for q in range(24)}) > 12          # the questions really decode differently
    assert full.stats["n_groups"] == 0                                            # nothing grouped: the per-row attention form


def test_the_tuned_forms_are_not_batch_invariant_and_the_mode_is_scoped():
    """The same comparison WITHOUT the mode shows why it exists (the score rows of a question differ between the 24-question batch and
    the question alone - different kernels, different summation orders), and leaving the `with` block restores the tuned forms."""
    from llava_align_amd import ops
    eng = _engine(W7B, n_layers=3, vit_layers=2)
    ids, imgs = _prompts(4, 6, 32000, seed=71)
    assert not ops.GEMM_BATCH_INVARIANT
    with ops.batch_invariant():
        assert ops.GEMM_BATCH_INVARIANT and ops.skinny_rows(4096, 4096) == 0 and ops.fused_attention_rows() == 0
        inv = _gen(eng, ids, imgs, range(24), use_dd_unk=True, temperature=0.2)
    assert not ops.GEMM_BATCH_INVARIANT and ops.skinny_rows(4096, 4096) > 0
    full = _gen(eng, ids, imgs, range(24), use_dd_unk=True, temperature=0.2)
    one = _gen(eng, ids, imgs, [13], use_dd_unk=True, temperature=0.2)
    assert full.stats["n_groups"] > 0                                             # tuned: shared prefixes attended once per group
    differs = sum(not torch.equal(sa[13].view(torch.int16), sb[0].view(torch.int16)) for sa, sb in zip(full.scores, one.scores))
    assert differs > 0
    # ... while both forms compute the same thing to the storage type's precision: step-0 scores of the two modes agree closely
    a, b = inv.scores[0].float(), full.scores[0].float()
    fin = torch.isfinite(a) & torch.isfinite(b)
    assert (a[fin] - b[fin]).abs().max().item() <= 0.25 + 2.0 ** -6 * b[fin].abs().max().item()


def test_run_pope_selects_the_mode_for_deterministic_decodes_only():
    """Driver default (shard.resolve_batch_invariant): cd_greedy / top_k = 1 -> batch-invariant, so batch_questions does not change an
    answer; sampled runs keep the tuned forms."""
    from llava_align_amd.pope_driver import run_pope
    eng = _engine(W7B, n_layers=2, vit_layers=2)
    ids, imgs = _prompts(3, 6, 32000, seed=5, image=336)
    images = {f"im{i}.jpg": imgs[6 * i] for i in range(3)}
    by_text = {f"q{i}": ids[i].tolist() for i in range(18)}
    qs = [{"question_id": i, "image": f"im{i // 6}.jpg", "text": f"q{i}", "label": ("yes", "no")[i % 2]} for i in range(18)]
    enc = lambda text, with_image: by_text[text] if with_image else [t for t in by_text[text] if t != -200]
    dec = lambda t: " ".join(map(str, t))
    kw = dict(unk_token_id=0, pad_token_id=0, eos_token_id=None, max_new_tokens=8, stop_str=None, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1,
              temperature=0.5)
    a = run_pope(eng, qs, enc, dec, lambda n: images[n], batch_questions=18, cd_greedy=True, **kw)
    b = run_pope(eng, qs, enc, dec, lambda n: images[n], batch_questions=6, cd_greedy=True, **kw)
    c = run_pope(eng, qs, enc, dec, lambda n: images[n], batch_questions=1, cd_greedy=True, **kw)
    assert a["batch_invariant"] and b["batch_invariant"] and c["batch_invariant"]
    assert [x["text"] for x in a["answers"]] == [x["text"] for x in b["answers"]] == [x["text"] for x in c["answers"]]
    assert [x["naive"] for x in a["answers"]] == [x["naive"] for x in c["answers"]]          # the step-0 label dicts too: same floats
    s = run_pope(eng, qs, enc, dec, lambda n: images[n], batch_questions=18, seed=3, **kw)
    assert not s["batch_invariant"]
    assert run_pope(eng, qs, enc, dec, lambda n: images[n], batch_questions=18, seed=3, batch_invariant=True, **kw)["batch_invariant"]


def test_prefill_planning_choices_do_not_change_a_bit_in_the_mode():
    """What lets the mode keep the prefill's batch-dependent PLANNING (tools/invariance_probe.py): packs of four short suffixes per attention
    workgroup, two-level prefixes (system prompt prefilled once when >= 5 images share it), prompt-prefix sharing itself and the common-prefix
    slot of text-only prompts only change where a key is read from - every sequence's keys are still summed in 64-key tiles at the same global
    key indices.  Score rows of every step, bit for bit."""
    from llava_align_amd import ops
    eng = _engine(W7B, n_layers=3, vit_layers=2)
    ids, imgs = _prompts(6, 4, 32000, seed=71)
    kw = dict(images=imgs, cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, output_scores=True, max_new_tokens=5, use_dd=True, use_dd_unk=True, temperature=1.0, top_p=0.9)

    def run(**over):
        eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
        return eng.generate(ids, **dict(kw, **over))

    def same(a, b):
        return torch.equal(a.tokens, b.tokens) and all(torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in zip(a.scores, b.scores))
    with ops.batch_invariant():
        base = run()
        assert base.stats["prefill_tokens"] < base.stats["unshared_prefill_tokens"]
        old = ops.FLASH_PACKS_IN_INVARIANT_MODE
        try:
            ops.FLASH_PACKS_IN_INVARIANT_MODE = not old
            assert same(base, run())
        finally:
            ops.FLASH_PACKS_IN_INVARIANT_MODE = old
        try:
            eng.two_level_prefix = False
            one_level = run()
            assert one_level.stats["prefill_tokens"] > base.stats["prefill_tokens"] and same(base, one_level)
        finally:
            eng.two_level_prefix = True
        assert same(base, run(share_prefix=False))


def test_repeated_image_free_rows_share_their_prompt_without_changing_a_bit():
    """POPE repeats a few dozen question texts over hundreds of images: the `unk` / `none` rows (and text-only prompts) with identical ids share
    everything but their last position (engine._plan, share_repeated_rows).  Fewer prefill tokens, the same score rows at every step - the rows
    diverge after step 0 (each follows its own question's sampled tokens) in their own slots."""
    from llava_align_amd import ops
    eng = _engine(W7B, n_layers=3, vit_layers=2)
    ids, imgs = _prompts(8, 6, 32000, seed=71)                                   # 48 questions, 6 per image
    texts = [ids[k][ids[k].tolist().index(-200) + 1:] for k in range(4)]         # four question texts asked about every image
    ids = [torch.cat([r[: r.tolist().index(-200) + 1], texts[i % 4]]) for i, r in enumerate(ids)]
    kw = dict(images=imgs, cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, output_scores=True, max_new_tokens=6, temperature=1.0, top_p=0.9)

    def run(**over):
        eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
        return eng.generate(ids, **dict(kw, **over))
    same = lambda a, b: torch.equal(a.tokens, b.tokens) and all(torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in zip(a.scores, b.scores))
    with ops.batch_invariant():
        for mode in (dict(use_dd_unk=True), dict(use_dd=True, use_dd_unk=True)):
            shared = run(**mode)
            try:
                eng.share_repeated_rows = False
                plain = run(**mode)
            finally:
                eng.share_repeated_rows = True
            n_free = 2 if len(mode) == 2 else 1
            assert plain.stats["prefill_tokens"] - shared.stats["prefill_tokens"] >= n_free * 44 * 15 and same(shared, plain), mode
            assert len({tuple(t) for t in shared.tokens.tolist()}) > 8           # the questions decode differently although their image-free rows started equal
        # text-only prompts (the calibrate drivers' prior passes): 48 prompts, 4 distinct
        text_only = [torch.tensor([t for t in r.tolist() if t != -200]) for r in ids]
        a = eng.generate(text_only, images=None, max_new_tokens=3, temperature=0.5, cd_greedy=True, output_scores=True, n_top=10)
        try:
            eng.share_repeated_rows = False
            eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
            b = eng.generate(text_only, images=None, max_new_tokens=3, temperature=0.5, cd_greedy=True, output_scores=True, n_top=10)
        finally:
            eng.share_repeated_rows = True
        assert same(a, b) and torch.equal(a.top_prob, b.top_prob) and a.stats["prefill_tokens"] < b.stats["prefill_tokens"]
    # the tuned forms take the same plan (grouped attention over the shared rows too); their low-order bits depend on the plan, the first tokens agree
    t = run(use_dd_unk=True)
    try:
        eng.share_repeated_rows = False
        u = run(use_dd_unk=True)
    finally:
        eng.share_repeated_rows = True
    assert t.stats["n_groups"] > u.stats["n_groups"] >= 8 and (t.tokens[:, 0] == u.tokens[:, 0]).float().mean().item() >= 0.9
