"""pope_driver.run_pope (batched main / none / unk passes over the engine) against the reference's per-question procedure
(llava_calibrate.py:130-219) restated with the oracle loop + the fp32 reference LLaVA: same answer tokens, same step-0 label
probabilities, the reference's JSONL schema, and the plain / calibrated scorers run on the result."""
import json

import numpy as np
import pytest
import torch

from oracle import vdd_oracle as O
from ref_llava import RefLlava

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SYS = [1, 17, 23, 99, 140, 7, 311, 12]


def decode_token(t):                       # a tokenizer with collisions: several ids spell yes / no in different cases
    return {0: "yes", 1: " Yes", 2: "no", 3: "No "}.get(t % 11, f"w{t}")


def decode(ids):
    return " ".join(decode_token(t).strip() for t in ids)


def encode(text, with_image):
    n = int(text[1:])
    body = [(n * 37 + 11 * k) % 997 + 3 for k in range(5 + n % 4)]
    return SYS + ([-200] if with_image else []) + body


def test_run_pope_matches_the_per_question_reference_procedure(tmp_path):
    from llava_align_amd import calibrate as C
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    from llava_align_amd.pope_driver import run_pope
    cfg = preset("tiny")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=False)
    ref = RefLlava(eng.w, device=DEV)
    images = {f"img{i}.jpg": torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(40 + i)) for i in range(3)}
    questions = [{"question_id": 100 + i, "image": f"img{i % 3}.jpg", "text": f"q{i}", "label": ("yes", "no")[i % 2]} for i in range(9)]
    path = tmp_path / "answers.jsonl"
    import threading
    loads = []

    def load_image(name):                              # host work of batch k + 1 runs on a worker thread while the GPU decodes batch k
        loads.append((name, threading.current_thread() is threading.main_thread()))
        return images[name]
    res = run_pope(eng, questions, encode, decode, load_image, answers_path=str(path), model_id="tiny", batch_questions=6,
                   unk_token_id=0, max_new_tokens=4, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, cd_greedy=True)
    assert sorted(n for n, _ in loads) == sorted(images) and not any(on_main for _, on_main in loads)      # each file once, none on the main thread
    lines = [json.loads(l) for l in open(path)]
    assert [l["question_id"] for l in lines] == [q["question_id"] for q in questions]
    assert all(tuple(l.keys()) == C.AnswerWriter.FIELDS for l in lines)
    assert lines == json.loads(json.dumps(res["answers"]))
    assert set(res["scores"]) == {"string_match", "naive", "none", "unk", "none_unk"}

    def step0_dict(ids, img, **kw):
        kw = dict(images=img[None] if img is not None else None, attention_mask=torch.ones(1, len(ids), dtype=torch.long), use_cache=True,
                  cd_alpha=1.0, cd_beta=0.1, **kw)
        r = O.reference_loop(ref, torch.tensor([ids]), warp=O.WarpConfig(temperature=0.5), max_length=len(ids) + 4, pad_token_id=None,
                             eos_token_id=None, pick=O.pick_argmax, **kw)
        probs = torch.softmax(r.scores[0][0].float(), -1)             # metrics.py:103
        tp, tt = torch.topk(probs, 10)
        return C.label_dict_from_top(tt.tolist(), tp.tolist(), decode_token), r.sequences[0, len(ids):].tolist(), probs
    checked = 0
    for q, a in zip(questions, res["answers"]):
        ids = encode(q["text"], True)
        d_main, toks, probs = step0_dict(ids, images[q["image"]], use_dd_unk=True)
        d_none, _, _ = step0_dict(encode(q["text"], False), None)
        d_unk, _, _ = step0_dict([0 if t == -200 else t for t in ids], None)
        for got, want in ((a["naive"], d_main), (a["none"], d_none), (a["unk"], d_unk)):
            # bf16 engine vs fp32 reference: label probabilities agree to a few percent; the 10th entry of a top-10 may swap
            pg, pw = np.array(C.get_prob_from_logits(got)), np.array(C.get_prob_from_logits(want))
            assert np.abs(pg - pw).max() <= 0.05 + 0.15 * pw.max(), (q["question_id"], pg, pw)
            assert len(set(got) & set(want)) >= min(len(got), len(want)) - 2      # collisions merge entries; the 10th may swap
        top2 = torch.topk(probs, 2).values
        if (top2[0] / top2[1]).item() > 1.5:                           # clear first token: the generated text starts with it
            assert a["text"].split(" ")[0] == decode_token(toks[0]).strip()
            checked += 1
        assert a["logits_score"] == C.get_prob_from_logits(a["naive"])
    assert checked >= 3


def test_image_priors_swap_the_image_and_keep_the_prompt(tmp_path):
    """test_samples_llava.py:134-158 / llava_calibrate.py:188-190: content-free passes with the image prompt and a noised / zero / all-ones
    image, plain sampling, step-0 label dict - 'zeros' and 'ones' against the oracle loop over the fp32 reference model."""
    from llava_align_amd import calibrate as C
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    from llava_align_amd.pope_driver import run_pope
    cfg = preset("tiny")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=False)
    ref = RefLlava(eng.w, device=DEV)
    images = {f"img{i}.jpg": torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(40 + i)) for i in range(2)}
    questions = [{"question_id": i, "image": f"img{i % 2}.jpg", "text": f"q{i}", "label": ("yes", "no")[i % 2]} for i in range(6)]
    path = tmp_path / "answers.jsonl"
    torch.manual_seed(5)
    res = run_pope(eng, questions, encode, decode, lambda name: images[name], answers_path=str(path), batch_questions=4, unk_token_id=0,
                   max_new_tokens=3, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, cd_greedy=True,
                   image_priors=("noise", "zeros", "ones"))
    lines = [json.loads(l) for l in open(path)]
    assert list(lines[0]) == list(C.AnswerWriter.FIELDS[:-1]) + ["noise", "zeros", "ones", "metadata"]
    assert set(res["scores"]) == {"string_match", "naive", "none", "unk", "none_unk", "noise", "zeros", "ones"}
    for q, a in zip(questions, lines):
        ids = encode(q["text"], True)
        for name, img in (("zeros", torch.zeros(3, 56, 56)), ("ones", torch.ones(3, 56, 56))):
            r = O.reference_loop(ref, torch.tensor([ids]), warp=O.WarpConfig(temperature=0.5), max_length=len(ids) + 1, pad_token_id=None,
                                 eos_token_id=None, pick=O.pick_argmax, images=img[None], attention_mask=torch.ones(1, len(ids), dtype=torch.long),
                                 use_cache=True, cd_alpha=1.0, cd_beta=0.1)
            tp, tt = torch.topk(torch.softmax(r.scores[0][0].float(), -1), 10)
            want = C.label_dict_from_top(tt.tolist(), tp.tolist(), decode_token)
            pg, pw = np.array(C.get_prob_from_logits(a[name])), np.array(C.get_prob_from_logits(want))
            assert np.abs(pg - pw).max() <= 0.05 + 0.15 * pw.max(), (q["question_id"], name, pg, pw)
        assert a["noise"] != a["zeros"] and a["zeros"] != a["naive"]
    with pytest.raises(ValueError, match="image_priors"):
        run_pope(eng, questions, encode, decode, lambda name: images[name], image_priors=("white",))


def test_unk_prior_is_read_off_the_unk_branch_of_the_main_pass():
    """With use_dd_unk the `unk` prior prompt IS the main pass's `unk` branch (llava_calibrate.py:59-60 / vcd_sample.py:154-155 build the same
    ids): run_pope reads its step-0 label dict off that branch instead of prefilling the prompt a second time.  In batch-invariant mode
    (cd_greedy) a row's logits do not depend on its batch, so the dicts equal those of the separate pass bit for bit; without use_dd_unk,
    with another <unk> id or with the VCD branch in the unk branch's place the separate pass runs."""
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    from llava_align_amd.pope_driver import run_pope
    cfg = preset("tiny")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=False)
    images = {f"img{i}.jpg": torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(40 + i)) for i in range(3)}
    questions = [{"question_id": i, "image": f"img{i % 3}.jpg", "text": f"q{i}", "label": ("yes", "no")[i % 2]} for i in range(9)]
    calls = []
    real = eng.generate
    eng.generate = lambda *a, **k: calls.append((len(a[0]), k.get("branch_priors", False))) or real(*a, **k)
    kw = dict(batch_questions=6, unk_token_id=0, max_new_tokens=3, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, cd_greedy=True)
    a = run_pope(eng, questions, encode, decode, lambda n: images[n], use_dd_unk=True, **kw)
    assert calls == [(6, True), (6, False), (3, True), (3, False)]                    # main + the `none` prompts only
    calls.clear()
    b = run_pope(eng, questions, encode, decode, lambda n: images[n], use_dd_unk=True, reuse_unk_branch=False, **kw)
    assert calls == [(6, False), (12, False), (3, False), (6, False)]                 # main + none and unk prompts
    assert [x["unk"] for x in a["answers"]] == [x["unk"] for x in b["answers"]] and [x["none"] for x in a["answers"]] == [x["none"] for x in b["answers"]]
    assert [x["text"] for x in a["answers"]] == [x["text"] for x in b["answers"]]
    assert json.dumps(a["scores"], sort_keys=True) == json.dumps(b["scores"], sort_keys=True)       # (NaN confidences compare as text)
    for extra in (dict(use_dd=True), dict(use_dd_unk=True, unk_token_id=5), dict(use_dd_unk=True, noise_step=500)):
        calls.clear()
        run_pope(eng, questions[:3], encode, decode, lambda n: images[n], **{**kw, **extra})
        assert calls[1] == (6, False) and not calls[0][1], extra


def test_repeated_question_texts_run_their_prior_prompt_once():
    """POPE asks the same questions about many images: a text-only prior prompt's step-0 label dict depends on its ids alone, so run_pope runs
    every distinct prompt once and hands the result to all its questions - the dicts a per-question pass gives (batch-invariant mode: bit for bit)."""
    from llava_align_amd import calibrate as C
    from llava_align_amd import ops
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    from llava_align_amd.pope_driver import run_pope
    cfg = preset("tiny")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, DEV, seed=3, std=0.06), device=DEV, use_graph=False)
    images = {f"img{i}.jpg": torch.randn(3, 56, 56, generator=torch.Generator().manual_seed(40 + i)) for i in range(4)}
    questions = [{"question_id": i, "image": f"img{i % 4}.jpg", "text": f"q{i % 3}", "label": ("yes", "no")[i % 2]} for i in range(12)]     # 3 distinct texts
    kw = dict(unk_token_id=0, max_new_tokens=2, cd_alpha=1.0, cd_beta=0.1, temperature=0.5, cd_greedy=True)
    a = run_pope(eng, questions, encode, decode, lambda n: images[n], batch_questions=12, use_dd_unk=True, **kw)
    assert (a["prior_prompts"], a["prior_prompts_run"]) == (12, 3)
    b = run_pope(eng, questions, encode, decode, lambda n: images[n], batch_questions=12, use_dd=True, **kw)                                  # none + unk prompts
    assert (b["prior_prompts"], b["prior_prompts_run"]) == (24, 6)
    with ops.batch_invariant():
        o = eng.generate([torch.tensor(encode(q["text"], False)) for q in questions], images=None, max_new_tokens=1, n_top=10, temperature=0.5)
    want = [C.label_dict_from_top(t, p_, decode_token) for t, p_ in zip(o.top_tok.tolist(), o.top_prob.tolist())]
    assert [x["none"] for x in a["answers"]] == want == [x["none"] for x in b["answers"]]
    assert [x["unk"] for x in a["answers"]] == [x["unk"] for x in b["answers"]]
    assert len({json.dumps(x["none"], sort_keys=True) for x in a["answers"]}) == 3
