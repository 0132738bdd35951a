"""The drop-in sample() loop (HIP tail on the GPU) reproduces the reference's token
traces, forward-call schedule and scores for the 5 decoding modes (fixtures made from
the real reference).  The toy LM is hosted on the CPU so its logits are bit-identical to
the fixture run; everything after `outputs.logits` happens on the GPU."""
import hashlib

import pytest
import torch
import transformers
from transformers.generation.logits_process import LogitsProcessorList, TopKLogitsWarper

from golden.gen_inputs import DTYPES, to_bits
from golden_io import load_json
from toy_lm import BankModel, ToyVLM

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TRACES = load_json("loop_traces.json")


class Hosted(ToyVLM):
    def __call__(self, input_ids=None, **kw):
        o = super().__call__(input_ids=input_ids.cpu(), **{k: (t.cpu() if torch.is_tensor(t) else t) for k, t in kw.items()})
        o.logits = o.logits.to(DEV)
        return o


class HostedBank(BankModel):
    def __call__(self, **kw):
        o = super().__call__(**kw)
        o.logits = o.logits.to(DEV)
        return o


def _crit(n):
    return transformers.StoppingCriteriaList([transformers.MaxLengthCriteria(max_length=n)])


@pytest.mark.parametrize("tr", TRACES, ids=[f"{t['dtype']}-{t['mode']}-q{t['q']}" for t in TRACES])
def test_loop_trace(tr):
    from llava_align_amd import sample
    ids = torch.tensor(tr["ids"])
    img, img_cd = torch.tensor(tr["img"]), torch.tensor(tr["img_cd"])
    kw = dict(images=img, attention_mask=torch.ones_like(ids).to(DEV), use_cache=True, cd_alpha=1.0, cd_beta=0.1)
    det = dict(cd_greedy=True)      # fp16 ties at the max: the oracle takes the lowest index, so must we
    kw.update({"plain": {}, "cd": {"images_cd": img_cd}, "dd": {"use_dd": True}, "dd_unk": {"use_dd_unk": True},
               "both": {"use_dd": True, "use_dd_unk": True}}[tr["mode"]])
    model = Hosted(logit_dtype=DTYPES[tr["dtype"]])
    out = sample(model, ids.to(DEV), logits_warper=LogitsProcessorList([TopKLogitsWarper(1)]),
                 stopping_criteria=_crit(ids.shape[1] + 8), output_scores=True, return_dict_in_generate=True, **kw, **det)
    # the forward-call schedule does not depend on the logits: must equal the reference's
    assert [[list(x) if isinstance(x, tuple) else x for x in c] for c in model.calls] == tr["schedule"]
    # tokens and scores: bit-exact against the oracle loop run on THIS host.  (The toy LM's fp32
    # CPU math differs in the last bit between host CPUs, so the fixture's tokens/score hashes
    # are only guaranteed on the build container, where tests/test_oracle_golden.py pins the
    # oracle to them.)
    from oracle import vdd_oracle as O
    kw_cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in kw.items()}
    r = O.reference_loop(ToyVLM(logit_dtype=DTYPES[tr["dtype"]]), ids.clone(), warp=O.WarpConfig(top_k=1),
                         max_length=ids.shape[1] + 8, pad_token_id=None, eos_token_id=None, pick=O.pick_argmax, **kw_cpu)
    assert out["sequences"].cpu().tolist() == r.sequences.tolist()
    assert all(torch.equal(to_bits(a.cpu()) if False else a.cpu().view(torch.int16 if a.dtype != torch.float32 else torch.int32),
                           b.view(torch.int16 if b.dtype != torch.float32 else torch.int32)) for a, b in zip(out["scores"], r.scores))


def test_eos_pad_and_early_stop():
    from llava_align_amd import sample
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
        ids = torch.ones(B, 4, dtype=torch.long, device=DEV)
        out = sample(HostedBank(bank), ids, logits_warper=LogitsProcessorList([TopKLogitsWarper(1)]),
                     stopping_criteria=_crit(4 + S), pad_token_id=case["pad"], eos_token_id=case["eos"],
                     output_scores=True, return_dict_in_generate=True,
                     attention_mask=torch.ones_like(ids), use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, cd_greedy=True)
        assert out["sequences"].cpu().tolist() == case["sequences"]
        assert len(out["scores"]) == case["n_scores"]


def test_python_processor_between_contrast_and_warp():
    """Qwen-style in-place processor (qwen_generation_utils.py:352-359 sets scores[i, eos] = 2**15)."""
    from llava_align_amd import sample
    from oracle import vdd_oracle as O

    class ForceTok:
        def __call__(self, input_ids, scores):
            scores[:, 3] = 2.0 ** 15
            return scores
    bank = [torch.randn(1, 50).half() * 3 for _ in range(4)]
    ids = torch.ones(1, 4, dtype=torch.long, device=DEV)
    out = sample(HostedBank([b.clone() for b in bank]), ids, logits_processor=LogitsProcessorList([ForceTok()]),
                 logits_warper=LogitsProcessorList([TopKLogitsWarper(1)]), stopping_criteria=_crit(6),
                 attention_mask=torch.ones_like(ids), use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1)
    assert out[0, 4:].cpu().tolist() == [3, 3]
    r = O.reference_loop(BankModel([b.clone() for b in bank]), ids.cpu(), warp=O.WarpConfig(top_k=1), max_length=6,
                         pad_token_id=None, eos_token_id=None, pick=O.pick_argmax,
                         processors=lambda ids_, x: ForceTok()(ids_, x), attention_mask=torch.ones_like(ids).cpu(),
                         use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1)
    assert out.cpu().tolist() == r.sequences.tolist()


def test_use_dd_rejects_batches_like_the_reference_cannot_handle():
    from llava_align_amd import sample
    ids = torch.ones(2, 4, dtype=torch.long, device=DEV)
    ids[:, 1] = -200
    with pytest.raises(ValueError, match="batch-1"):
        sample(HostedBank([torch.zeros(2, 20).half()] * 4), ids, stopping_criteria=_crit(5),
               attention_mask=torch.ones_like(ids), use_dd=True)


def test_v5_entry_greedy_disables_contrast_with_warning():
    from llava_align_amd.vcd_sample import _sample_v5
    gc = transformers.GenerationConfig(do_sample=False, max_length=6)
    bank = [torch.randn(1, 30).half() for _ in range(4)]
    m = HostedBank(bank)
    ids = torch.ones(1, 4, dtype=torch.long, device=DEV)
    with pytest.warns(UserWarning, match="WITHOUT contrastive"):
        out = _sample_v5(m, ids, LogitsProcessorList(), _crit(6), gc, attention_mask=torch.ones_like(ids), use_dd_unk=True)
    assert m.i == 2                                   # one forward per step: no contrast branch ran
    assert out[0, 4:].cpu().tolist() == [int(bank[0].argmax()), int(bank[1].argmax())]


def test_streamer_gets_every_step_and_one_end():
    """vcd_sample.py:264-265 `streamer.put(next_tokens.cpu())` after every step (pad tokens of finished rows included, as in the
    reference) and :299-300 `streamer.end()` once."""
    from llava_align_amd import sample

    class Rec:
        def __init__(self):
            self.items, self.ended = [], 0
        def put(self, v):
            assert v.device.type == "cpu"
            self.items.append(v.clone())
        def end(self):
            self.ended += 1
    g = load_json("eos_pad.json")
    case = g["cases"][0]
    plan = torch.tensor(case["plan"])
    B, S = plan.shape
    bank = []
    for s in range(S):
        for _ in range(2):
            row = torch.zeros(B, case["V"], dtype=torch.float16)
            row[torch.arange(B), plan[:, s]] = 9.0
            bank.append(row)
    ids = torch.ones(B, 4, dtype=torch.long, device=DEV)
    rec = Rec()
    out = sample(HostedBank(bank), ids, logits_warper=LogitsProcessorList([TopKLogitsWarper(1)]), stopping_criteria=_crit(4 + S),
                 pad_token_id=case["pad"], eos_token_id=case["eos"], return_dict_in_generate=True, streamer=rec,
                 attention_mask=torch.ones_like(ids), use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, cd_greedy=True)
    seq = out["sequences"].cpu()
    assert rec.ended == 1 and len(rec.items) == seq.shape[1] - 4
    assert torch.equal(torch.stack(rec.items, 1), seq[:, 4:])
