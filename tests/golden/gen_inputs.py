"""Deterministic input generators shared by make_golden.py (build container) and the
tests (any box).  Large rows are regenerated from a seed instead of being committed."""
from __future__ import annotations

import numpy as np
import torch

DTYPES = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}
BITS = {torch.float32: torch.int32, torch.float16: torch.int16, torch.bfloat16: torch.int16}


def to_bits(t: torch.Tensor) -> np.ndarray:
    return t.contiguous().view(BITS[t.dtype]).numpy().copy()


def from_bits(a: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).view(dtype)


def logit_rows(seed: int, B: int, V: int, dtype: torch.dtype, n_in: int, kind: str = "normal", steps: int = 1):
    """Returns a list (per step) of n_in tensors [B, V].

    kind:
      normal     v ~ N(0, 4^2) with a planted unique row max; c (, d) = v + N(0, 1.5^2)
      flat       v ~ N(0, 0.5^2): many survivors under the beta mask
      vc_equal   c == v bit-for-bit (VCD-after-step-0 / Qwen degeneracy, SURVEY A.3 #1,#4)
      big        |v| up to ~6e4 in fp16 range: (1+a)v overflows to +-inf in fp16
      one_left   one token far above the rest: everything else masked
      max_tie    the row max appears twice
    """
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(steps):
        v = rng.standard_normal((B, V), dtype=np.float32) * (0.5 if kind == "flat" else 4.0)
        if kind == "big":
            v *= 2500.0
        pos = rng.integers(0, V, size=B)
        if kind in ("normal", "vc_equal", "one_left", "max_tie"):
            v[np.arange(B), pos] = np.abs(v).max(axis=1) + (30.0 if kind == "one_left" else 1.37)
        if kind == "max_tie":
            pos2 = (pos + 1 + rng.integers(0, V - 1, size=B)) % V
            v[np.arange(B), pos2] = v[np.arange(B), pos]
        rows = [torch.from_numpy(v).to(dtype)]
        for _j in range(n_in - 1):
            if kind == "vc_equal":
                rows.append(rows[0].clone())
            else:
                c = v + rng.standard_normal((B, V), dtype=np.float32) * 1.5
                rows.append(torch.from_numpy(c).to(dtype))
        out.append(rows)
    return out
