"""CPU side of the HF adapter (llava-align_amd/hf_adapter.py): the architecture read from a live LlavaLlamaForCausalLM-shaped module,
the parameter mapping (checked by running tests/ref_llava.py on the mapped weights against the HF module itself), and the way
generate()'s keywords are resolved against model.generation_config.  The GPU run is tests/test_hf_adapter_gpu.py."""
import types

import pytest
import torch

import hf_llava
from llava_align_amd import hf_adapter as A
from llava_align_amd.engine import LlavaWeights, preset
from ref_llava import RefLlava


@pytest.fixture(scope="module")
def model():
    return hf_llava.build("cpu", torch.float32)


def test_config_is_read_from_the_live_modules(model):
    c, t = A.config_from_hf(model), preset("tiny")
    assert c.lm == t.lm and c.vision == t.vision
    model.get_vision_tower().select_feature = "cls_patch"
    with pytest.raises(ValueError, match="select_feature"):
        A.config_from_hf(model)
    model.get_vision_tower().select_feature = "patch"
    with pytest.raises(ValueError, match="fp16 or bf16"):            # the engine serves the model where the reference puts it: GPU, half precision
        A.weights_from_hf(model)


def test_mapped_weights_reproduce_the_hf_module(model):
    """HfLlava (installed transformers' Llama + CLIP, spliced as llava_arch.py:122-163) == ref_llava on LlavaWeights.from_state_dict
    of ITS state dict (tower under `model.vision_tower.vision_tower.`, as in a LLaVA checkpoint), for the main / <unk> / image-free ids."""
    cfg = A.config_from_hf(model)
    w = LlavaWeights.from_state_dict(cfg, model.state_dict(), "cpu", dtype=torch.float32)
    ref = RefLlava(w, device="cpu", logit_dtype=torch.float32, dtype=torch.float32, store=torch.float32)
    ids = torch.tensor([[1, 17, 250, 33, -200, 400, 401, 77, 12]])
    img = torch.randn(1, 3, cfg.vision.image, cfg.vision.image)
    unk = ids.clone(); unk[unk == -200] = 0
    with torch.no_grad():
        for i, im in ((ids, img), (unk, None), (ids[ids != -200][None], None)):
            a, b = model(input_ids=i, images=im).logits, ref(input_ids=i, images=im).logits
            assert a.shape == b.shape and torch.allclose(a, b, rtol=2e-3, atol=2e-3), (a - b).abs().max()


class _StubEngine:
    def __init__(self, model=None):
        self.calls = []
        # what the adapter's stale-weights check reads: the engine's lm_head is the model's own storage after attach_engine
        self.w = types.SimpleNamespace(t={"lm_head": model.lm_head.weight if model is not None else torch.zeros(1)})

    def generate(self, input_ids, **kw):
        self.calls.append(kw)
        Q, n = input_ids.shape[0], kw.get("max_new_tokens") or (kw["max_length"] - input_ids.shape[1])
        gen = torch.full((Q, n), 5, dtype=torch.long)
        return types.SimpleNamespace(sequences=[torch.cat([r, g]) for r, g in zip(input_ids, gen)], tokens=gen,
                                     scores=[torch.zeros(Q, 7)] * n if kw.get("output_scores") else None, top_prob=None, top_tok=None, stats={})


def test_generate_keywords_are_resolved_like_hf(model):
    stub = _StubEngine(model)
    model._vdd_engine = stub
    A._set_guard(model)
    ids = torch.tensor([[1, 9, -200, 4]])
    gc = model.generation_config
    gc.temperature, gc.top_p, gc.top_k, gc.do_sample = 0.9, 0.6, 50, False
    try:
        # the reference's call (llava_calibrate.py:161-177): explicit None for top_p / top_k overrides the checkpoint's defaults
        out = A._native_generate(model, ids, images=torch.zeros(1), cd_alpha=1.0, cd_beta=0.1, use_dd=False, use_dd_unk=True, do_sample=True,
                                 temperature=0.2, top_p=None, top_k=None, max_new_tokens=64, use_cache=True, output_attentions=True,
                                 output_scores=True, return_dict_in_generate=True)
        kw = stub.calls[-1]
        assert kw["do_sample"] is True and kw["temperature"] == 0.2 and kw["top_p"] is None and kw["top_k"] is None
        assert kw["max_new_tokens"] == 64 and "max_length" not in kw and kw["use_dd_unk"] is True and kw["cd_alpha"] == 1.0
        assert "use_cache" not in kw and kw["output_attentions"] is True and "return_dict_in_generate" not in kw
        assert out["sequences"].shape == (1, 4 + 64) and out.sequences is out["sequences"] and len(out["scores"]) == 64
        with pytest.raises(KeyError, match="llava_calibrate.py:180-182"):     # (the stub engine produced no map)
            out["attentions"]
        # nothing given: the model's generation_config decides (greedy, its warper defaults, max_length 20)
        seq = A._native_generate(model, ids, images=torch.zeros(1))
        kw = stub.calls[-1]
        assert kw["do_sample"] is False and kw["temperature"] == 0.9 and kw["top_p"] == 0.6 and kw["top_k"] == 50
        assert kw["max_length"] == (gc.max_length or 20) and "max_new_tokens" not in kw and torch.is_tensor(seq)
        with pytest.warns(UserWarning, match="pad_token_id"):
            A._native_generate(model, ids, eos_token_id=[2, 3], pad_token_id=None, max_new_tokens=1)
        assert stub.calls[-1]["pad_token_id"] == 2 and stub.calls[-1]["eos_token_id"] == [2, 3]
        with pytest.raises(ValueError, match="attention_mask"):
            A._native_generate(model, ids, attention_mask=torch.tensor([[0, 1, 1, 1]]), max_new_tokens=1)
        with pytest.raises(ValueError, match=r"\[batch, length\]"):
            A._native_generate(model, [ids[0]], max_new_tokens=1)
        # parameters re-allocated behind the engine's back (model.to() / .half()): refuse instead of decoding with the old weights
        old_w = model.lm_head.weight
        model.lm_head.weight = torch.nn.Parameter(old_w.detach().clone())
        try:
            with pytest.raises(RuntimeError, match="attach_engine"):
                A._native_generate(model, ids, max_new_tokens=1)
        finally:
            model.lm_head.weight = old_w
    finally:
        del model._vdd_engine
        model.__dict__.pop("_vdd_guard", None)


def test_stale_weights_guard_follows_the_models_own_parameter():
    """the guard compares the MODEL's lm_head with what it was at attach time (address, in-place version, shape, dtype) - not
    with the engine's tensor, which is a copy when lm_head is tied to the embedding or not contiguous (every generate() used to raise)."""
    import torch
    from llava_align_amd import hf_adapter as A

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embed = torch.nn.Embedding(16, 8)
            self.lm_head = torch.nn.Linear(8, 16, bias=False)
            self.lm_head.weight = self.embed.weight                      # tied
    m = M()
    A._set_guard(m)
    engine_copy = m.lm_head.weight.detach().clone()                      # what a .contiguous() of a tied / strided head gives the engine
    assert engine_copy.data_ptr() != m.lm_head.weight.data_ptr()
    A._check_guard(m, "moved")                                           # unchanged model: fine, although the engine holds a copy
    with torch.no_grad():
        m.lm_head.weight.add_(1.0)                                       # a LoRA merge / in-place update
    with pytest.raises(RuntimeError, match="moved"):
        A._check_guard(m, "moved")
    A._set_guard(m)
    m.half()                                                             # re-allocation
    with pytest.raises(RuntimeError, match="moved"):
        A._check_guard(m, "moved")
