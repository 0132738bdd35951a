"""Readers for the committed golden fixtures (tests/golden/*)."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from golden.gen_inputs import DTYPES, from_bits, logit_rows

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def kernel_cases():
    if "k" not in _cache:
        with open(os.path.join(GOLD, "kernel_vectors.json")) as f:
            meta = json.load(f)
        _cache["k"] = (meta, np.load(os.path.join(GOLD, "kernel_vectors.npz")))
    return _cache["k"]


def case_inputs(case):
    return logit_rows(case["seed"], case["B"], case["V"], DTYPES[case["dtype"]], case["n_in"], case["kind"], case["steps"])


def check_scores(case, arrays, step, got: torch.Tensor):
    """Bit-compare a [B, V] scores tensor with the stored reference output of `step`."""
    import hashlib
    dt = DTYPES[case["dtype"]]
    ci = case["id"]
    if case["dense"]:
        want = from_bits(arrays[f"c{ci}_s{step}_scores"], dt)
        same = (got.view(torch.int16 if dt != torch.float32 else torch.int32)
                == want.view(torch.int16 if dt != torch.float32 else torch.int32))
        return bool(same.all()), int((~same).sum())
    flat = got.reshape(-1)
    idx = torch.from_numpy(arrays[f"c{ci}_s{step}_idx"])
    val = from_bits(arrays[f"c{ci}_s{step}_val"], dt)
    n_neg, n_nan, n_pos = case["counts"][str(step)]
    bits = torch.int16 if dt != torch.float32 else torch.int32
    bad = int((flat[idx].contiguous().view(bits) != val.view(bits)).sum())
    bad += abs(int((flat == -float("inf")).sum()) - n_neg) + abs(int(torch.isnan(flat).sum()) - n_nan)
    sha = hashlib.sha256(got.contiguous().view(bits).numpy().tobytes()).hexdigest()
    if sha != case["sha256"][str(step)]:
        bad = max(bad, 1)
    return bad == 0, bad


def load_json(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def processor_cases():
    """tests/golden/processors.{json,npz}: the real reference sample() run with HF's MinLength / MinNewTokensLength /
    RepetitionPenalty processors and the reference's own Qwen StopWordsLogitsProcessor between contrast and warp."""
    if "p" not in _cache:
        _cache["p"] = (load_json("processors.json"), np.load(os.path.join(GOLD, "processors.npz")))
    return _cache["p"]


def processor_case_rows(case):
    """-> per step the [v, c, d][:n_in] rows (each [B, V]) of a processors.json case."""
    return logit_rows(case["seed"], case["B"], case["V"], DTYPES[case["dtype"]], case["n_in"], "normal", case["steps"])


def processor_case_scores(case, arrays, step):
    return from_bits(arrays[f"p{case['id']}_s{step}"], DTYPES[case["dtype"]])


def gpu_scalar_cases():
    """tests/golden/kernel_vectors_gpu_scalar.*: the kernel-vector cases (fp16 / bf16, V <= 32000) from the reference run with
    torch-GPU's scalar arithmetic emulated at the two places the backends differ (make_golden.py::_GpuScalarEmulation)."""
    if "k2" not in _cache:
        with open(os.path.join(GOLD, "kernel_vectors_gpu_scalar.json")) as f:
            meta = json.load(f)
        _cache["k2"] = (meta, np.load(os.path.join(GOLD, "kernel_vectors_gpu_scalar.npz")))
    return _cache["k2"]
