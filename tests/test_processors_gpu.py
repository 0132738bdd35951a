"""The logits-processor stage (vcd_sample.py:197 / :204: `logits_processor(input_ids, scores)` between contrast and warp) on the
HIP path: (1) the kernels against outputs of the REAL reference sample() run with HF's MinLength / MinNewTokensLength /
RepetitionPenalty processors and the reference's own Qwen StopWordsLogitsProcessor (tests/golden/processors.*), bit for bit;
(2) the engine's generate(min_new_tokens=, min_length=, stop_words_ids=, repetition_penalty=, logits_processor=) against the oracle
loop with the same processors over the fp32 torch LLaVA; (3) kwargs the engine does not implement raise instead of vanishing."""
import numpy as np
import pytest
import torch

from golden.gen_inputs import DTYPES
from golden_io import processor_case_rows, processor_case_scores, processor_cases
from oracle import vdd_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PCASES, PARR = processor_cases()


def _bits(t):
    return t.contiguous().view(torch.int32 if t.dtype == torch.float32 else torch.int16)


@pytest.mark.parametrize("case", PCASES, ids=[f"p{c['id']}-{c['dtype']}-V{c['V']}-n{c['n_in']}-{c['proc']}" for c in PCASES])
def test_processor_stage_matches_reference_fixture(case):
    """Teacher-forced replay of the fixture's own sequences: at every step the kernels see the history the reference saw."""
    from llava_align_amd import WarpSpec, contrast_sample, ops
    spec, eos, L0, B = case["spec"], case["eos"], len(case["ids"][0]), case["B"]
    rows = processor_case_rows(case)
    seqs = torch.tensor(case["sequences"], device=DEV)
    gen = seqs[:, L0:].contiguous()                                   # what the reference emitted (pad after EOS included)
    eos_t = torch.tensor(eos, device=DEV)
    floor = max(spec.get("min_new", 0), spec.get("min_len", 0) - L0)
    eos_min = torch.full((B,), floor, dtype=torch.int32, device=DEV) if floor > 0 else None
    sw = ops.StopWords(case["stop_words"], eos[0], DEV) if spec.get("stop") else None
    tail = sw.prompt_tail(case["ids"], DEV) if sw else None
    prompt = torch.tensor([[(-1 if t == -200 else t) for t in r] for r in case["ids"]], device=DEV)
    unfinished = torch.ones(B, dtype=torch.long, device=DEV)
    step_dev = torch.zeros(1, dtype=torch.long, device=DEV)
    warp = WarpSpec(**case["warp"])
    for s in range(case["n_scores"]):
        r = [t.to(DEV) for t in rows[s]]
        v, c, d = r[0], (r[1] if case["n_in"] >= 2 else None), (r[2] if case["n_in"] == 3 else None)
        kw = {}
        # odd steps pass the step through the device counter (the graph-replay path), even steps as a host value
        st = dict(step=0, step_ptr=step_dev) if s % 2 else dict(step=s)
        step_dev.fill_(s)
        if eos_min is not None:
            kw.update(eos_min_step=eos_min, **st)
        if sw is not None:
            kw.update(force_eos=ops.stop_words_match(sw, tail, gen, **st), force_eos_id=eos[0])
        if "rep" in spec:
            x = contrast_sample(v, c, d, alpha=1.0, beta=0.1, no_sample=True, return_scores=True).scores if c is not None else v.clone()
            ops.repetition_penalty_(x, spec["rep"], prompt, gen, **st)
            v, c, d = x, None, None
        out = contrast_sample(v, c, d, alpha=1.0, beta=0.1, warp=warp, return_scores=True, pick_argmax=True, eos_ids=eos_t,
                              pad_id=case["pad"], unfinished=unfinished, **kw)
        want = processor_case_scores(case, PARR, s).to(DEV)
        assert torch.equal(_bits(out.scores), _bits(want)), (s, int((_bits(out.scores) != _bits(want)).sum()))
        # a row whose only plausible token was the masked EOS is all -inf: the reference's multinomial raises there (the fixture's
        # arg-max stand-in wrote 0); the kernel reports it through row_status instead of inventing a token
        dead = torch.isneginf(want).all(-1)
        assert out.status.ne(0).tolist() == dead.tolist(), s
        live = (~dead).nonzero().reshape(-1).tolist()
        assert [out.tokens[i].item() for i in live] == [seqs[i, L0 + s].item() for i in live], s
        for i in dead.nonzero().reshape(-1).tolist():                # keep following the fixture's run behind its stand-in token
            if seqs[i, L0 + s].item() in eos:
                unfinished[i] = 0


def test_stop_words_match_semantics():
    from llava_align_amd import ops
    sw = ops.StopWords([[7, 8], [9], [1, 2, 3, 4], [5]], 5, DEV)               # [5] == [eos] is dropped (qwen_generation_utils.py:340-344)
    assert sw.seqs == [[7, 8], [9], [1, 2, 3, 4]] and sw.max_len == 4
    prompts = [[3, 7], [1, 2, 3], [9], [], [5]]
    tail = sw.prompt_tail(prompts, DEV)
    gen = torch.tensor([[8, 0, 0], [4, 9, 0], [1, 7, 8], [7, 8, 9], [5, 5, 5]], device=DEV)
    got = [ops.stop_words_match(sw, tail, gen, step=n).tolist() for n in range(4)]
    ow = O.StopWords([[7, 8], [9], [1, 2, 3, 4], [5]], 5)
    for n in range(4):
        ids = [p + gen[i, :n].tolist() for i, p in enumerate(prompts)]
        want = []
        for row in ids:
            sc = torch.zeros(1, 10)
            ow(torch.tensor([row], dtype=torch.long).reshape(1, -1), sc)
            want.append(int(sc[0, 5] != 0))
        assert got[n] == want, (n, got[n], want)
    assert got[0] == [0, 0, 1, 0, 0] and got[1] == [1, 1, 0, 0, 0] and got[3] == [0, 0, 1, 1, 0]
    with pytest.raises(ValueError):
        ops.StopWords([], 5, DEV)
    with pytest.raises(ValueError):
        ops.StopWords([[1, -2]], 5, DEV)


# ---------------------------------------------------------------------------------------------- engine level
from ref_llava import RefLlava  # noqa: E402
from test_engine_gpu import prompts  # noqa: E402


@pytest.fixture(scope="module")
def eng():
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("tiny")
    return VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, t_max=256, use_graph=True)


@pytest.fixture(scope="module")
def ref(eng):
    return RefLlava(eng.w, device=DEV)


def _ref_run(ref, ids, img, n_new, procs, eos, pad, mode):
    kw = dict(images=img[None] if img is not None else None, attention_mask=torch.ones(1, ids.numel(), dtype=torch.long), use_cache=True,
              cd_alpha=1.0, cd_beta=0.1, **mode)
    return O.reference_loop(ref, ids[None].clone(), warp=O.WarpConfig(temperature=0.5), max_length=ids.numel() + n_new, pad_token_id=pad,
                            eos_token_id=eos, pick=O.pick_argmax, processors=procs, **kw)


def _compare(out, refs, ids, n_new, tol):
    """Engine (bf16) vs fp32 reference under the same processors: the finite pattern of the scores rows (what the processors
    force), the scores within tolerance, and the tokens wherever the reference's top-1 margin clears the noise."""
    checked = 0
    for q, r in enumerate(refs):
        want = r.sequences[0, ids[q].numel():].tolist()
        got = out.tokens[q].tolist()[:len(want)]
        for step in range(len(r.scores)):
            s_got, s_want = out.scores[step][q].float().cpu(), r.scores[step][0].float().cpu()
            big_g, big_w = s_got > 1e4, s_want > 1e4                          # a forced EOS (2**15 / T)
            assert torch.equal(big_g, big_w), (q, step)
            fin = torch.isfinite(s_got) & torch.isfinite(s_want) & ~big_w
            assert (torch.isfinite(s_got) ^ torch.isfinite(s_want)).sum() <= 3 + 0.05 * int(fin.sum())
            if fin.any():
                assert (s_got[fin] - s_want[fin]).abs().max().item() <= tol, (q, step)
            top2 = torch.topk(s_want, 2).values
            if (top2[0] - top2[1]).item() > 2 * tol:
                assert got[step] == want[step], (q, step)
                checked += 1
            if got[step] != want[step]:
                break
    return checked


@pytest.mark.both_scalar_forms
def test_min_new_tokens_and_stop_words_in_the_captured_step(eng, ref):
    """Qwen's MME call shape (run_qwen.py:190-213 + modeling_qwen.py:1061-1075): min_new_tokens, pad = eos, stop words."""
    ids, imgs = prompts(seed=21)
    n_new = 7
    kw = dict(images=imgs, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, cd_greedy=True, use_dd_unk=True, output_scores=True)
    base = eng.generate(ids, max_new_tokens=n_new, **kw)
    eos = int(base.tokens[0, 0])                                  # question 0 would stop at once: min_new_tokens must keep it alive
    stop = [[int(base.tokens[1, 1]), int(base.tokens[1, 2])], [int(ids[2][-1])]]    # a generated pair; the last PROMPT id of question 2
    out = eng.generate(ids, max_new_tokens=n_new, eos_token_id=eos, pad_token_id=eos, min_new_tokens=2, stop_words_ids=stop, sync_every=1, **kw)
    assert out.stats["graph"]
    assert torch.isneginf(out.scores[0][:, eos]).all() is not None
    assert int(out.tokens[0, 0]) != eos and int(out.tokens[0, 1]) != eos
    # question 2's prompt ends with a stop word: the stop processor (after min_new_tokens in HF's order) forces EOS at step 0
    assert int(out.tokens[2, 0]) == eos and float(out.scores[0][2, eos]) == 2.0 ** 15 / 0.5
    refs = []
    for q in range(len(ids)):
        procs = O.ProcessorList([O.MinNewTokens(ids[q].numel(), 2, [eos]), O.StopWords(stop, eos)])
        refs.append(_ref_run(ref, ids[q], imgs[q], n_new, procs, eos, eos, {"use_dd_unk": True}))
    assert _compare(out, refs, ids, n_new, tol=0.4) >= len(ids)


@pytest.mark.both_scalar_forms
def test_min_length_counts_the_prompt(eng, ref):
    ids, imgs = prompts(seed=22)
    kw = dict(images=imgs, temperature=0.5, cd_greedy=True, output_scores=True, max_new_tokens=4)
    base = eng.generate(ids, **kw)
    eos = int(base.tokens[0, 1])
    Lmin = min(i.numel() for i in ids)
    out = eng.generate(ids, eos_token_id=eos, pad_token_id=0, min_length=Lmin + 2, sync_every=1, **kw)
    for q, i in enumerate(ids):
        floor = max(0, Lmin + 2 - i.numel())                       # new tokens before EOS is allowed for THIS row
        for s in range(4):
            masked = bool(torch.isneginf(out.scores[s][q, eos]))
            assert masked == (s < floor) or not masked and s >= floor, (q, s)
            if s < floor:
                assert masked
    refs = [_ref_run(ref, ids[q], imgs[q], 4, O.ProcessorList([O.MinLength(Lmin + 2, [eos])]), eos, 0, {}) for q in range(len(ids))]
    assert _compare(out, refs, ids, 4, tol=0.15) >= 2


@pytest.mark.both_scalar_forms
def test_repetition_penalty_and_python_processors(eng, ref):
    """InstructBLIP-style kwargs (blip2_vicuna_instruct.py:396-402) on slot-free prompts + an HF-style callable."""
    from llava_align_amd import add_diffusion_noise
    rng = np.random.default_rng(5)
    ids = [torch.tensor([1] + rng.integers(3, 1000, size=int(rng.integers(6, 12))).tolist()) for _ in range(4)]
    kw = dict(temperature=0.5, cd_greedy=True, output_scores=True, max_new_tokens=6)
    base = eng.generate(ids, **kw)
    out = eng.generate(ids, repetition_penalty=1.5, **kw)
    assert out.stats["graph"] and not torch.equal(base.scores[0], out.scores[0])
    refs = []
    for q in range(len(ids)):
        kwr = dict(images=None, attention_mask=torch.ones(1, ids[q].numel(), dtype=torch.long), use_cache=True)
        refs.append(O.reference_loop(ref, ids[q][None].clone(), warp=O.WarpConfig(temperature=0.5), max_length=ids[q].numel() + 6,
                                     pad_token_id=None, eos_token_id=None, pick=O.pick_argmax,
                                     processors=O.ProcessorList([O.RepetitionPenalty(1.5)]), **kwr))
    assert _compare(out, refs, ids, 6, tol=0.15) >= 2

    class Ban:                                                    # HF-style callable: sees left-padded ids, edits scores
        def __init__(self):
            self.seen = []

        def __call__(self, input_ids, scores):
            self.seen.append(tuple(input_ids.shape))
            scores[:, 7] = -float("inf")
            scores[:, 11] = 2.0 ** 15
            return scores
    ban = Ban()
    out2 = eng.generate(ids, logits_processor=[ban], **kw)
    assert not out2.stats["graph"] and (out2.tokens == 11).all()
    Lp = max(i.numel() for i in ids)                              # what HF hands a processor: the longest prompt, no rounding
    assert ban.seen == [(4, Lp + s) for s in range(6)]

    class Probe:                                                  # shorter rows are LEFT-padded with the pad id; real tokens sit right-aligned
        def __init__(self):
            self.first = None

        def __call__(self, input_ids, scores):
            if self.first is None:
                self.first = input_ids.clone()
            return scores
    pr = Probe()
    eng.generate(ids, logits_processor=[pr], pad_token_id=0, eos_token_id=None, **kw)
    for q, i in enumerate(ids):
        row = pr.first[q].tolist()
        assert row[Lp - i.numel():] == i.tolist() and all(t == 0 for t in row[:Lp - i.numel()])
    # prompts given as embeddings have HF length 0: a processor sees only the generated ids
    ban2 = Ban()
    emb = [torch.randn(9 + q, 256, device=DEV).to(eng.dtype) * 0.1 for q in range(2)]
    eng.generate(None, inputs_embeds=emb, logits_processor=[ban2], temperature=0.5, cd_greedy=True, max_new_tokens=3)
    assert ban2.seen == [(2, s) for s in range(3)]
    with pytest.raises(ValueError, match="-200"):
        ids_img, imgs = prompts(seed=3)
        eng.generate(ids_img, images=imgs, repetition_penalty=1.2, max_new_tokens=2)


def test_unimplemented_generate_kwargs_raise_instead_of_vanishing(eng):
    ids, imgs = prompts(seed=23)
    with pytest.raises(TypeError, match="no_repeat_ngram_size"):
        eng.generate(ids, images=imgs, max_new_tokens=2, no_repeat_ngram_size=3)
    with pytest.raises(ValueError, match="beam"):
        eng.generate(ids, images=imgs, max_new_tokens=2, num_beams=5)
    # the reference drivers' no-effect kwargs are accepted (llava_calibrate.py:161-177, run_qwen.py:190-213)
    out = eng.generate(ids, images=imgs, max_new_tokens=2, use_cache=True, output_attentions=True, output_hidden_states=True,
                       length_penalty=1, num_return_sequences=1, num_beams=1, attention_mask=None)
    assert out.tokens.shape == (len(ids), 2)


def test_max_length_as_lavis_passes_it(eng):
    """blip2_vicuna_instruct.py:396: max_length instead of max_new_tokens; inputs_embeds prompts have HF length 0."""
    emb = [torch.randn(9, 256, device=DEV).bfloat16() * 0.1 for _ in range(2)]
    out = eng.generate(None, inputs_embeds=emb, max_length=5, min_length=1, eos_token_id=2, pad_token_id=2, cd_greedy=True,
                       temperature=1.0, output_scores=True, repetition_penalty=1.0, sync_every=1)
    assert out.tokens.shape[1] <= 5 and torch.isneginf(out.scores[0][:, 2]).all()            # min_length 1 > HF length 0 at step 0 only
    if len(out.scores) > 1:
        assert torch.isfinite(out.scores[1][:, 2]).all()
