"""The CPU oracle must reproduce, bit for bit, what the REAL reference produced
(fixtures made by tests/golden/make_golden.py from /root/reference)."""
import hashlib

import numpy as np
import pytest
import torch

from golden.gen_inputs import DTYPES, from_bits, to_bits
from golden_io import case_inputs, check_scores, kernel_cases, load_json
from oracle import vdd_oracle as O
from toy_lm import BankModel, ToyVLM


def _ids(meta_cases):
    return [f"c{c['id']}-{c['dtype']}-V{c['V']}-n{c['n_in']}-{c['kind']}" for c in meta_cases]


META, ARR = kernel_cases()
SMALL = [c for c in META["cases"] if c["V"] <= 1003]
LARGE = [c for c in META["cases"] if c["V"] > 1003]


@pytest.mark.parametrize("case", SMALL + LARGE, ids=_ids(SMALL + LARGE))
def test_step_scores_match_reference(case):
    rows = case_inputs(case)
    warp = O.WarpConfig(**case["warp"])
    for s, step_rows in enumerate(rows):
        v = step_rows[0]
        c = step_rows[1] if case["n_in"] >= 2 else None
        d = step_rows[2] if case["n_in"] == 3 else None
        got = O.step_scores(v, c, d, case["alpha"], case["beta"], warp)
        ok, bad = check_scores(case, ARR, s, got)
        assert ok, f"{bad} mismatching elements at step {s}"
        tok = O.pick_argmax(torch.softmax(got, -1))
        assert tok.tolist() == [r[s] for r in case["tokens"]]


def test_all_masked_row_raises_in_reference():
    assert META["all_masked_row_raises"] == "RuntimeError"


TRACES = load_json("loop_traces.json")


@pytest.mark.parametrize("tr", TRACES, ids=[f"{t['dtype']}-{t['mode']}-q{t['q']}" for t in TRACES])
def test_loop_trace_matches_reference(tr):
    ids = torch.tensor(tr["ids"])
    img, img_cd = torch.tensor(tr["img"]), torch.tensor(tr["img_cd"])
    kw = dict(images=img, attention_mask=torch.ones_like(ids), use_cache=True, cd_alpha=1.0, cd_beta=0.1)
    kw.update({"plain": {}, "cd": {"images_cd": img_cd}, "dd": {"use_dd": True}, "dd_unk": {"use_dd_unk": True},
               "both": {"use_dd": True, "use_dd_unk": True}}[tr["mode"]])
    model = ToyVLM(logit_dtype=DTYPES[tr["dtype"]])
    r = O.reference_loop(model, ids.clone(), warp=O.WarpConfig(top_k=1), max_length=ids.shape[1] + 8,
                         pad_token_id=None, eos_token_id=None, pick=O.pick_argmax, **kw)
    assert r.sequences[:, ids.shape[1]:].tolist() == tr["tokens"]
    assert [[list(x) if isinstance(x, tuple) else x for x in c] for c in model.calls] == tr["schedule"]
    assert [hashlib.sha256(to_bits(s).tobytes()).hexdigest() for s in r.scores] == tr["score_sha256"]


def test_eos_pad_matches_reference():
    g = load_json("eos_pad.json")
    for case in g["cases"]:
        plan = torch.tensor(case["plan"])
        B, S = plan.shape
        bank = []
        for s in range(S):
            for _ in range(2):
                row = torch.zeros(B, case["V"], dtype=torch.float16)
                row[torch.arange(B), plan[:, s]] = 9.0
                bank.append(row)
        ids = torch.ones(B, 4, dtype=torch.long)
        r = O.reference_loop(BankModel(bank), ids, warp=O.WarpConfig(top_k=1), max_length=4 + S, pad_token_id=case["pad"],
                             eos_token_id=case["eos"], pick=O.pick_argmax, attention_mask=torch.ones_like(ids),
                             use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1)
        assert r.sequences.tolist() == case["sequences"]
        assert len(r.scores) == case["n_scores"]
    with pytest.raises(ValueError) as e:
        O.reference_loop(BankModel(bank, pad=None), ids, warp=O.WarpConfig(top_k=1), max_length=6, pad_token_id=None, eos_token_id=2,
                         pick=O.pick_argmax, attention_mask=torch.ones_like(ids), use_dd_unk=True)
    assert str(e.value) == g["eos_without_pad_error"]


def test_noise_matches_reference(golden_dir):
    z = np.load(f"{golden_dir}/noise.npz")
    for t in (0, 1, 500, 999):
        x = torch.from_numpy(z[f"x_{t}"])
        torch.manual_seed(200 + t)
        eps = torch.randn_like(x)
        y = O.add_diffusion_noise(x, t, noise=eps)
        # bit-exact on the host that made the fixture; other CPUs' vectorised sigmoid/cumprod may
        # differ in the last bit of the schedule constants
        assert torch.allclose(y, torch.from_numpy(z[f"y_{t}"]), rtol=3e-7, atol=1e-7)


class FakeTok:
    TABLE = {0: "<unk>", 1: "Yes", 2: " yes", 3: "No", 4: "no ", 5: "YES", 6: "maybe"}

    def decode(self, i):
        return self.TABLE.get(int(i), f"t{int(i)}")


def test_calibration_matches_reference():
    g = load_json("calibration.json")
    for e in g["label_dict"]:
        dt = DTYPES[e["dtype"]]
        row = from_bits(np.array(e["row_bits"], dtype=np.int32 if dt == torch.float32 else np.int16), dt)
        d = O.top_token_probs(row, FakeTok().decode)
        assert d == e["dict"]
        assert O.label_probs(d) == e["p"]
    for e in g["affine"]:
        hits = []
        for p, lab, want in zip(e["probs"], e["labels"], e["calibrated"]):
            q, ans = O.affine_calibrate(p, e["p_cf"], e["mode"] or "diagonal_W")
            assert np.array_equal(q.reshape(-1), np.array(want))
            hits.append(int(ans == lab))
        assert float(np.mean(hits)) == e["acc"]


# ---- logits processors between contrast and warp (vcd_sample.py:197 / :204) ---------------------------------------------------
from golden_io import processor_case_rows, processor_case_scores, processor_cases  # noqa: E402

PCASES, PARR = processor_cases()


def oracle_processors(case):
    """The oracle's restatements in HF's order (repetition penalty, min_length, min_new_tokens, caller's stop words)."""
    spec, eos, L0 = case["spec"], case["eos"], len(case["ids"][0])
    lst = O.ProcessorList()
    if "rep" in spec:
        lst.append(O.RepetitionPenalty(spec["rep"]))
    if "min_len" in spec:
        lst.append(O.MinLength(spec["min_len"], eos))
    if "min_new" in spec:
        lst.append(O.MinNewTokens(L0, spec["min_new"], eos))
    if spec.get("stop"):
        lst.append(O.StopWords(case["stop_words"], eos[0]))
    return lst


@pytest.mark.parametrize("case", PCASES, ids=[f"p{c['id']}-{c['dtype']}-V{c['V']}-n{c['n_in']}-{c['proc']}" for c in PCASES])
def test_loop_with_logits_processors_matches_reference(case):
    rows = processor_case_rows(case)
    bank = [r for step in rows for r in step]
    ids = torch.tensor(case["ids"])
    mode = {1: {}, 2: {"use_dd_unk": True}, 3: {"use_dd": True, "use_dd_unk": True}}[case["n_in"]]
    r = O.reference_loop(BankModel(bank), ids.clone(), warp=O.WarpConfig(**case["warp"]), max_length=ids.shape[1] + case["steps"],
                         pad_token_id=case["pad"], eos_token_id=case["eos"], pick=O.pick_argmax, processors=oracle_processors(case),
                         attention_mask=torch.ones_like(ids), cd_alpha=1.0, cd_beta=0.1, **mode)
    assert r.sequences.tolist() == case["sequences"]
    assert len(r.scores) == case["n_scores"]
    for s, got in enumerate(r.scores):
        want = processor_case_scores(case, PARR, s)
        assert torch.equal(torch.from_numpy(to_bits(got)), torch.from_numpy(to_bits(want))), (s, int((got != want).sum()))


def test_processor_fixture_bites():
    """The processors really change the reference's output in the fixture: some EOS that the free run emits is masked, some stop
    sequence forces EOS (a 2**15 score appears, also from the prompt tail at step 0), the penalty changes scores."""
    forced = masked_first = 0
    for c in PCASES:
        s0 = processor_case_scores(c, PARR, 0)
        if c["spec"].get("stop"):
            forced += int((s0[1] == (2.0 ** 15) / c["warp"].get("temperature", 1.0)).any())      # row 1's prompt ends with 6 = a stop word
        if "min_new" in c["spec"] or "min_len" in c["spec"]:
            masked_first += int(c["sequences"][0][len(c["ids"][0])] != c["free_tokens"][0][0])
    assert forced >= 20 and masked_first >= 20


# ---- second golden set: torch-GPU scalar arithmetic ------------------------------------------------------------------------
from golden_io import gpu_scalar_cases  # noqa: E402

META2, ARR2 = gpu_scalar_cases()


@pytest.mark.gpu_scalar
@pytest.mark.parametrize("case", META2["cases"], ids=_ids(META2["cases"]))
def test_step_scores_gpu_scalar_form_matches_second_golden_set(case):
    assert O.GPU_SCALAR is True
    rows = case_inputs(case)
    warp = O.WarpConfig(**case["warp"])
    for s, step_rows in enumerate(rows):
        got = O.step_scores(step_rows[0], step_rows[1] if case["n_in"] >= 2 else None, step_rows[2] if case["n_in"] == 3 else None,
                            case["alpha"], case["beta"], warp)
        ok, bad = check_scores(case, ARR2, s, got)
        assert ok, f"{bad} mismatching elements at step {s}"


def test_the_two_golden_sets_differ_only_by_an_ulp_somewhere():
    cpu = {c["id"]: c for c in META["cases"]}
    differing = [c["id"] for c in META2["cases"] if c["sha256"] != cpu[c["id"]]["sha256"]]
    assert 1 <= len(differing) <= len(META2["cases"]) // 4
