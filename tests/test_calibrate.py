"""Product-side calibration / scoring (llava_align_amd.calibrate) against what the reference's own functions and
scorer scripts produced (fixtures from tests/golden/make_golden.py).  Pure host arithmetic: no GPU."""
import json
import os

import numpy as np
import pytest
import torch

from golden.gen_inputs import DTYPES, from_bits
from golden_io import load_json
from llava_align_amd import calibrate as C


class FakeTok:
    TABLE = {0: "<unk>", 1: "Yes", 2: " yes", 3: "No", 4: "no ", 5: "YES", 6: "maybe"}

    def decode(self, i):
        return self.TABLE.get(int(i), f"t{int(i)}")


def test_label_dict_and_label_probs_match_reference():
    g = load_json("calibration.json")
    for e in g["label_dict"]:
        dt = DTYPES[e["dtype"]]
        row = from_bits(np.array(e["row_bits"], dtype=np.int32 if dt == torch.float32 else np.int16), dt)
        probs = torch.softmax(row, dim=-1).float()            # what the fused kernel emits as top_prob/top_tok
        p, t = torch.topk(probs, 10)
        d = C.label_dict_from_top(t[0].tolist(), p[0].tolist(), FakeTok().decode)
        assert d == e["dict"]
        assert C.get_prob_from_logits(d) == e["p"]


def test_affine_modes_match_reference_eval_accuracy():
    g = load_json("calibration.json")
    for e in g["affine"]:
        for p, want in zip(e["probs"], e["calibrated"]):
            if e["p_cf"] is None:
                q, _ = C.affine_calibrate(p, None)
            else:     # metrics.eval_accuracy uses p_cf as is (no normalisation / eps): reproduce through calibrate_weight
                W, b = C.calibrate_weight(np.array(e["p_cf"]), e["mode"])
                pp = np.array(p) / np.sum(p)
                q = np.matmul(W, pp[:, None]) + b
                q = (q / np.sum(q)).reshape(-1)
            assert np.array_equal(q, np.array(want))


def test_pope_scorers_match_reference_scripts():
    g = load_json("scorers.json")
    s = C.pope_scores(g["gt"], g["gen"])
    for k_ref, k in (("precision", "precision"), ("recall", "recall"), ("f1", "f1"), ("accuracy", "accuracy"), ("yes", "yes"), ("unknow", "unknown")):
        assert s[k] == pytest.approx(g["eval_pope"][k_ref], abs=1e-12)
    for name, want in g["eval_pope_calibrate"].items():
        got = C.pope_scores_calibrated(g["gt"], g["gen"], name)
        assert got["n"] == want["n"]
        for k in ("f1", "accuracy", "precision", "recall", "yes"):
            assert float(f"{got[k] * 100:.4}") == want[k], (name, k)        # the script prints percentages with 4 significant digits
        assert got["confidence"] == pytest.approx(want["confidence"], rel=1e-12)


def test_every_calibration_setting_of_the_reference_script():
    """The script's full setting list (its line 82: the priors qwen_calibrate.py writes - noise, none, zero, unk - and their sums)."""
    g = load_json("scorers_all.json")
    assert list(g["eval_pope_calibrate"]) == ["naive", "noise", "none", "zero", "unk", "none_noise", "none_unk", "none_unk_noise", "all"]
    for name, want in g["eval_pope_calibrate"].items():
        got = C.pope_scores_calibrated(g["gt"], g["gen"], name)
        assert got["n"] == want["n"]
        for k in ("f1", "accuracy", "precision", "recall", "yes"):
            assert float(f"{got[k] * 100:.4}") == want[k], (name, k)
        assert got["confidence"] == pytest.approx(want["confidence"], rel=1e-12)
    assert C.calibrate_sources("naive") == ("naive",) and C.calibrate_sources("zero") == ("naive", "zero")
    assert C.calibrate_sources("all") == ("naive", "noise", "none", "zero", "unk")


def test_vectorised_calibration_equals_the_per_question_arithmetic_bit_for_bit():
    """calibrate._calibrated_rows (all questions at once) against calibrate._calibrated_row (the script's statements on one question): every
    bit of q, NaN patterns included - rows without a label among the top-10 of the answer or of the prior, zero priors, tiny and huge
    probabilities, both calibration modes."""
    rng = np.random.default_rng(17)
    n = 4000
    P = rng.random((n, 2)) ** rng.integers(1, 8, size=(n, 1))
    CF = rng.random((n, 2)) ** rng.integers(1, 8, size=(n, 1)) * rng.choice([1.0, 2.0, 4.0], size=(n, 1))     # sums of up to four priors
    for rows, col in ((slice(0, 40), None), (slice(40, 80), 0), (slice(80, 120), 1)):
        if col is None:
            P[rows] = 0.0                                     # neither label in the answer's top-10
        else:
            P[rows, col] = 0.0
    CF[100:160] = 0.0                                         # neither label in the prior's top-10
    CF[160:200, 0] = 0.0
    P[200:210] = 1e-300; CF[210:220] = 1e-300; P[220:230, 0] = 1e300
    for mode in ("diagonal_W", "identity_W"):
        for cf in (None, CF):
            got = C._calibrated_rows(P, cf, mode)
            want = np.stack([C._calibrated_row(P[i], None if cf is None else cf[i], mode).reshape(-1) for i in range(n)])
            assert np.array_equal(got.view(np.int64), want.view(np.int64)), (mode, cf is None)
            assert np.array_equal(np.argmax(got, 1), np.array([int(np.argmax(w)) for w in want]))
    assert np.isnan(C._calibrated_rows(P, CF, "diagonal_W")).any(1).sum() >= 100


def test_answer_writer_schema(tmp_path):
    p = tmp_path / "a.jsonl"
    with C.AnswerWriter(str(p)) as w:
        w.write(7, "Is there a dog?", "Yes", "llava-1.5-7b", "x.jpg", [0.9, 0.1], {"yes": 0.9}, {"yes": 0.5}, {"no": 0.6})
        line = json.loads(open(p).read().splitlines()[0])      # flushed before close
    assert tuple(line.keys()) == C.AnswerWriter.FIELDS and line["question_id"] == 7 and line["metadata"] == {}


def test_mme_scorer_matches_reference_on_its_own_answer_set(golden_dir):
    """The one answer set the reference ships (experiments/eval_tool/answers/llava-v1.5-7b, copied as DATA) scored by the
    reference's own experiments/eval/MME/eval_tool/calculation.py gives Perception 648.33 / Cognition 363.21."""
    s = C.mme_scores(os.path.join(golden_dir, "mme_answers_llava15_7b"))
    assert s["Perception"]["total"] == pytest.approx(648.3333333333333, abs=1e-9)
    assert s["Cognition"]["total"] == pytest.approx(363.2142857142857, abs=1e-9)
    want = {"existence": 190.0, "count": 155.0, "position": 133.33333333333334, "color": 170.0, "commonsense_reasoning": 110.71428571428571,
            "numerical_calculation": 70.0, "text_translation": 107.5, "code_reasoning": 75.0}
    got = {**s["Perception"]["tasks"], **s["Cognition"]["tasks"]}
    for k, v in want.items():
        assert got[k] == pytest.approx(v, abs=1e-9), k
    assert C.mme_parse_pred("yes") == "yes" and C.mme_parse_pred("no, it") == "no" and C.mme_parse_pred("maybe") == "other"
    assert C.mme_parse_pred("the yes") == "other"          # only the first 4 characters are searched


def test_mme_convert_matches_the_reference_converter_script():
    """tests/golden/mme_convert.json: what experiments/eval/MME/convert_answer_to_mme_calibrate.py (run unmodified, make_golden.py
    gen_mme_convert) wrote for a synthetic benchmark tree + answers file: 'naive' keeps the text, none / unk / none_unk answer by the
    individually calibrated arg-max; the ground-truth key quirks (single-word suffix dropped, double-space variant) included."""
    import json
    import os
    from llava_align_amd import calibrate as C
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mme_convert.json")))
    gt = {(c, f, q): a for c, f, q, a in g["gt"]}
    got = C.mme_convert(g["answers"], gt)
    assert set(got) == set(g["results"]) == {"naive", "none", "unk", "none_unk"}
    for name, cats in g["results"].items():
        assert got[name] == cats, name
    # the calibrated variants really differ from each other and from the generated text somewhere
    flat = lambda n: [l.split("\t")[3] for c in sorted(got[n]) for l in got[n][c]]
    assert flat("none") != flat("unk") and flat("none_unk") != flat("none") and set(flat("none")) <= {"Yes", "No"}
    assert any("  Please answer yes or no." in l for c in got["naive"].values() for l in c)


def test_mme_results_files_feed_the_scorer(tmp_path):
    import json
    import os
    from llava_align_amd import calibrate as C
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mme_convert.json")))
    gt = {(c, f, q): a for c, f, q, a in g["gt"]}
    dirs = C.write_mme_results(C.mme_convert(g["answers"], gt), str(tmp_path), "exp")
    for name, d in dirs.items():
        assert sorted(os.listdir(d)) == sorted(f"{c}.txt" for c in g["results"][name])
        assert open(os.path.join(d, "color.txt")).read().splitlines() == g["results"][name]["color"]
        s = C.mme_scores(d)
        assert set(s) == {"Perception", "Cognition"} and 0 <= s["Perception"]["total"] <= 800
    # mme_load_gt reads the benchmark tree back (both layouts: <cat>/*.txt and <cat>/questions_answers_YN/*.txt next to images/)
    bench = tmp_path / "bench"
    for c, f, q, a in g["gt"]:
        d = bench / c / "questions_answers_YN" if c in ("existence", "position") else bench / c
        d.mkdir(parents=True, exist_ok=True)
        if c in ("existence", "position"):
            (bench / c / "images").mkdir(exist_ok=True)
        with open(d / f, "a") as fp:
            fp.write(q + "\t" + a + "\n")
    assert C.mme_load_gt(str(bench)) == gt
