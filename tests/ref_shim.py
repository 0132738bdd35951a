"""Build-container-only loader for the REAL reference sample() (read-only import from
/root/reference; nothing is copied).  Recipe: SURVEY.md Appendix B.

Used by tests/golden/make_golden.py (fixture generation) only; /root/reference does not exist
on the GPU box, and no test imports this module.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("VDD_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "vcd_utils", "vcd_sample.py"))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_cache = {}


def load_reference():
    """Returns a namespace with .sample (vcd_sample.py:25), .add_diffusion_noise
    (vcd_add_noise.py:3) and .metrics (experiments/utils/metrics.py)."""
    if "ref" in _cache:
        return _cache["ref"]
    sys.dont_write_bytecode = True
    import transformers.generation as G
    import transformers.generation.utils as GU
    for m in (G, GU):  # names removed after the 4.3x era
        for n, r in (("SampleOutput", GU.GenerateDecoderOnlyOutput),
                     ("SampleDecoderOnlyOutput", GU.GenerateDecoderOnlyOutput),
                     ("SampleEncoderDecoderOutput", GU.GenerateEncoderDecoderOutput)):
            if not hasattr(m, n):
                setattr(m, n, r)
    for name in ("experiments", "experiments.llava"):  # bypass experiments/llava/__init__ (AutoConfig collision)
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = []
            sys.modules[name] = pkg
    sys.modules["experiments.llava.constants"] = _load(
        "experiments.llava.constants", os.path.join(REF_ROOT, "experiments/llava/constants.py"))
    ns = types.SimpleNamespace()
    ns.vcd_sample = _load("ref_vcd_sample", os.path.join(REF_ROOT, "vcd_utils/vcd_sample.py"))
    ns.sample = ns.vcd_sample.sample
    ns.add_diffusion_noise = _load("ref_vcd_noise", os.path.join(REF_ROOT, "vcd_utils/vcd_add_noise.py")).add_diffusion_noise
    ns.metrics = _load("ref_metrics", os.path.join(REF_ROOT, "experiments/utils/metrics.py"))
    _cache["ref"] = ns
    return ns


def hf_warpers(temperature=None, top_k=None, top_p=None, min_keep=1):
    """The LogitsProcessorList HF 4.31 `_get_logits_warper` builds for these args."""
    from transformers.generation.logits_process import (LogitsProcessorList, TemperatureLogitsWarper,
                                                        TopKLogitsWarper, TopPLogitsWarper)
    lst = LogitsProcessorList()
    if temperature is not None and temperature != 1.0:
        lst.append(TemperatureLogitsWarper(float(temperature)))
    if top_k is not None and top_k != 0:
        lst.append(TopKLogitsWarper(top_k=int(top_k), min_tokens_to_keep=min_keep))
    if top_p is not None and top_p < 1.0:
        lst.append(TopPLogitsWarper(top_p=float(top_p), min_tokens_to_keep=min_keep))
    return lst


def load_qwen_stop_words():
    """The reference's own StopWordsLogitsProcessor class (experiments/Qwen_VL/qwen_generation_utils.py:305-385), loaded by path."""
    if "qgu" not in _cache:
        sys.dont_write_bytecode = True
        _cache["qgu"] = _load("ref_qwen_generation_utils", os.path.join(REF_ROOT, "experiments/Qwen_VL/qwen_generation_utils.py"))
    return _cache["qgu"].StopWordsLogitsProcessor


def run_reference(model, input_ids, *, max_length, warp=None, pad=None, eos=None, output_scores=True,
                  multinomial=None, logits_processor=None, **model_kwargs):
    """Drive the reference sample() once.  `multinomial`, if given, temporarily replaces
    torch.multinomial (the reference's only RNG consumer on this path, vcd_sample.py:202)."""
    import torch
    ref = load_reference()

    class MaxLen431(list):
        """4.31-era MaxLengthCriteria semantics: ONE python bool for the whole batch
        (5.x returns a per-row tensor, which `if stopping_criteria(...)` at
        vcd_sample.py:295 cannot take at batch > 1)."""

        def __call__(self, ids, scores, **kw):
            return ids.shape[-1] >= max_length

    warp = warp or {}
    saved = torch.multinomial
    if multinomial is not None:
        torch.multinomial = multinomial
    try:
        out = ref.sample(model, input_ids, logits_processor=logits_processor,
                         logits_warper=hf_warpers(**warp),
                         stopping_criteria=MaxLen431(),
                         pad_token_id=pad, eos_token_id=eos, output_scores=output_scores,
                         return_dict_in_generate=True, **model_kwargs)
    finally:
        torch.multinomial = saved
    return out
