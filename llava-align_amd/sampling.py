"""Tensor-level entry to the fused HIP sampling tail (C ABI `vdd_contrast_sample`).

Replaces, in one launch and without host synchronisation, the reference's per-step
tail: vcd_utils/vcd_sample.py:185-207 (average, contrast, plausibility mask, warpers,
softmax, multinomial), :257-260 (pad after EOS), :285-288 (unfinished update).
torch is used only for device memory and the stream handle.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch

from . import _lib

_DT = {torch.float32: _lib.VDD_F32, torch.float16: _lib.VDD_F16, torch.bfloat16: _lib.VDD_BF16}


@dataclass
class WarpSpec:
    """What HF's warper list amounts to (temperature -> top-k -> top-p).  Same activation
    rules as the reference era's `_get_logits_warper`."""
    temperature: Optional[float] = None
    top_k: Optional[int] = None
    top_p: Optional[float] = None
    min_keep: int = 1

    @property
    def t(self) -> float:
        return float(self.temperature) if (self.temperature is not None and self.temperature != 1.0) else 0.0

    @property
    def k(self) -> int:
        return int(self.top_k) if (self.top_k is not None and self.top_k != 0) else 0

    @property
    def p(self) -> float:
        return float(self.top_p) if (self.top_p is not None and self.top_p < 1.0) else 2.0


@dataclass
class SampleOutput:
    tokens: Optional[torch.Tensor]        # [B] int64 (a view into the caller's buffer when `out_tokens` was given)
    scores: Optional[torch.Tensor]        # [B, V] model dtype, post-warp (what output_scores returns)
    top_prob: Optional[torch.Tensor]      # [B, n_top] fp32
    top_tok: Optional[torch.Tensor]       # [B, n_top] int64
    status: torch.Tensor                  # [B] int32, 0 ok / 1 empty-or-NaN distribution

    def raise_if_invalid(self):
        """Host sync.  The reference surfaces this as torch.multinomial's RuntimeError."""
        if bool((self.status != 0).any()):
            raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0 "
                               "(vdd_contrast_sample: no finite score survived)")


def fresh_offset() -> int:
    """A Philox counter offset drawn from torch's default generator: every call advances the generator (as the reference's
    torch.multinomial does, vcd_sample.py:202), so repeated generations differ, and torch.manual_seed() reproduces a run."""
    return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())



def log_beta_f32(beta: float) -> float:
    """vcd_sample.py:191 computes torch.log(torch.tensor(cd_beta)) in fp32; reuse torch's own
    logf so the cutoff is bit-identical."""
    return float(torch.log(torch.tensor(float(beta), dtype=torch.float32)).item())


def _row_view(t: torch.Tensor, name: str):
    if t.dim() != 2:
        raise ValueError(f"{name}: expected [B, V], got {tuple(t.shape)}")
    if t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t


# True (the default since round 3): every contrast_sample call that does not say otherwise (the drop-in sample() and the engine
# included) uses torch-GPU's scalar arithmetic for the plausibility cutoff and the temperature - what the reference computes when
# its tensors live on a GPU, as in every one of its drivers (`.cuda()` before generate, llava_calibrate.py:163): log(beta) enters
# the cutoff add as an fp32 scalar, `scores / T` is a multiplication by fl32(1 / T).  False: torch-CPU's arithmetic (log(beta)
# demoted to the model dtype first, a true division).  Both forms are pinned bit for bit: tests/golden/kernel_vectors.* (the
# reference run on a CPU) and tests/golden/kernel_vectors_gpu_scalar.* (the same run with the two GPU scalar paths emulated);
# they differ by at most 1 ulp of the model dtype, on a handful of elements.
GPU_SCALAR_SEMANTICS = True


def contrast_sample(logits_v: torch.Tensor, logits_cd: Optional[torch.Tensor] = None,
                    logits_dd: Optional[torch.Tensor] = None, *, alpha: float = 0.5, beta: float = 0.1,
                    warp: Optional[WarpSpec] = None, seed: Optional[int] = None, offset: Optional[int] = None,
                    uniforms: Optional[torch.Tensor] = None, eos_ids: Optional[torch.Tensor] = None,
                    pad_id: Optional[int] = None, unfinished: Optional[torch.Tensor] = None,
                    out_tokens: Optional[torch.Tensor] = None, return_scores: bool = False,
                    out_scores: Optional[torch.Tensor] = None, n_top: int = 0, pick_argmax: bool = False,
                    no_sample: bool = False, cutoff_f32_scalar: Optional[bool] = None, temp_reciprocal: Optional[bool] = None,
                    topp_fp32_mass: bool = False,
                    workspace: Optional[torch.Tensor] = None, stream: Optional[int] = None,
                    offset_ptr: Optional[torch.Tensor] = None, status_out: Optional[torch.Tensor] = None,
                    eos_min_step: Optional[torch.Tensor] = None, step: int = 0, step_ptr: Optional[torch.Tensor] = None,
                    force_eos: Optional[torch.Tensor] = None, force_eos_id: Optional[int] = None,
                    force_eos_value: float = float(2 ** 15)) -> SampleOutput:
    """Fused contrastive sampling tail on [B, V] last-position logits (any row stride).

    logits_cd=None is the reference's plain path (:204-207); logits_dd selects the
    both-branches average (:185).  Asynchronous on the current stream.

    Logits-processor stage (where vcd_sample.py:197 / :204 call `logits_processor`, before the warpers): eos_min_step int32 [B] -
    every id of eos_ids scores -inf while step (+ the int64 device scalar step_ptr) < eos_min_step[row] (HF MinNewTokensLength /
    MinLength processors; needs eos_ids); force_eos int32 [B] flags + force_eos_id - scores[row, force_eos_id] = force_eos_value
    where set (Qwen StopWordsLogitsProcessor, qwen_generation_utils.py:352-359; flags from ops.stop_words_match).

    cutoff_f32_scalar / temp_reciprocal select torch-GPU's scalar arithmetic (log(beta) added in fp32 before the rounding, the
    temperature division as a multiply by the reciprocal) instead of torch-CPU's, which the golden vectors were made with; None =
    the module default GPU_SCALAR_SEMANTICS (True: torch-GPU's).  Both forms are bit-exact against their own torch backend
    (tests/test_kernel_gpu.py::test_torch_gpu_eager_agrees_within_reference_tolerance); they differ by at most 1 ulp.
    """
    if cutoff_f32_scalar is None:
        cutoff_f32_scalar = GPU_SCALAR_SEMANTICS
    if temp_reciprocal is None:
        temp_reciprocal = GPU_SCALAR_SEMANTICS
    if not logits_v.is_cuda:
        raise _lib.VddLibraryError("contrast_sample needs device tensors: this package has no CPU path")
    lib = _lib.load_lib()
    warp = warp or WarpSpec()
    v = _row_view(logits_v, "logits_v")
    B, V = v.shape
    if v.dtype not in _DT:
        raise ValueError(f"unsupported logits dtype {v.dtype}")
    dev = v.device
    prm = _lib.VddSampleParams()
    prm.abi_version = _lib.ABI_VERSION
    prm.flags = ((_lib.PICK_ARGMAX if pick_argmax else 0) | (_lib.CUTOFF_F32_SCALAR if cutoff_f32_scalar else 0)
                 | (_lib.TEMP_RECIPROCAL if temp_reciprocal else 0) | (_lib.NO_SAMPLE if no_sample else 0)
                 | (_lib.TOPP_FP32_MASS if topp_fp32_mass else 0))
    prm.logit_v, prm.stride_v = v.data_ptr(), v.stride(0)
    keep = [v]
    if logits_cd is not None:
        c = _row_view(logits_cd, "logits_cd")
        if c.shape != v.shape or c.dtype != v.dtype:
            raise ValueError("logits_cd must match logits_v in shape and dtype")
        prm.logit_cd, prm.stride_cd = c.data_ptr(), c.stride(0)
        keep.append(c)
    if logits_dd is not None:
        if logits_cd is None:
            raise ValueError("logits_dd given without logits_cd")
        d = _row_view(logits_dd, "logits_dd")
        if d.shape != v.shape or d.dtype != v.dtype:
            raise ValueError("logits_dd must match logits_v in shape and dtype")
        prm.logit_dd, prm.stride_dd = d.data_ptr(), d.stride(0)
        keep.append(d)
    prm.B, prm.V, prm.dtype, prm.min_keep = B, V, _DT[v.dtype], int(warp.min_keep)
    prm.alpha = float(alpha)
    prm.log_beta = log_beta_f32(beta) if logits_cd is not None else 0.0
    prm.temperature, prm.top_p, prm.top_k = warp.t, warp.p, warp.k
    prm.philox_seed = (torch.initial_seed() if seed is None else int(seed)) & 0xFFFFFFFFFFFFFFFF
    prm.philox_offset = fresh_offset() if offset is None else int(offset)
    if offset_ptr is not None:                     # int64 device scalar added to the offset at run time (graph replay)
        prm.philox_offset_ptr = offset_ptr.data_ptr()
    if uniforms is not None:
        if uniforms.dtype != torch.float32 or uniforms.numel() != B or not uniforms.is_contiguous():
            raise ValueError("uniforms must be contiguous fp32 [B]")
        prm.uniforms = uniforms.data_ptr()
    if eos_ids is not None and eos_ids.numel() > 0 and not (unfinished is None and eos_min_step is not None):
        if unfinished is None:
            raise ValueError("eos_ids given without an `unfinished` state tensor")
        if pad_id is None:
            raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")  # :258-259
        prm.eos_ids, prm.n_eos, prm.pad_id = eos_ids.data_ptr(), eos_ids.numel(), int(pad_id)
        prm.unfinished = unfinished.data_ptr()
    if eos_min_step is not None:
        if eos_ids is None or eos_ids.numel() == 0:
            raise ValueError("eos_min_step given without eos_ids")
        if eos_min_step.dtype != torch.int32 or eos_min_step.numel() != B or not eos_min_step.is_contiguous():
            raise ValueError("eos_min_step must be contiguous int32 [B]")
        if unfinished is None:                     # the processor stage only needs the id list, not the pad / unfinished bookkeeping
            prm.eos_ids, prm.n_eos = eos_ids.data_ptr(), eos_ids.numel()
        prm.eos_min_step, prm.step = eos_min_step.data_ptr(), int(step)
        if step_ptr is not None:
            prm.step_ptr = step_ptr.data_ptr()
        keep.append(eos_min_step)
    if force_eos is not None:
        if force_eos.dtype != torch.int32 or force_eos.numel() != B or not force_eos.is_contiguous() or force_eos_id is None:
            raise ValueError("force_eos must be contiguous int32 [B] and comes with force_eos_id")
        prm.force_eos, prm.force_eos_id, prm.force_eos_value = force_eos.data_ptr(), int(force_eos_id), float(force_eos_value)
        keep.append(force_eos)
    tokens = None
    if not no_sample:
        tokens = out_tokens if out_tokens is not None else torch.empty(B, dtype=torch.long, device=dev)
        if tokens.dtype != torch.long or tokens.numel() != B:
            raise ValueError("out_tokens must be int64 with B elements")
        prm.next_tokens = tokens.data_ptr()
        prm.stride_tokens = tokens.stride(0) if tokens.dim() >= 1 and B > 1 else 1
    scores = None
    if return_scores or out_scores is not None:
        scores = out_scores if out_scores is not None else torch.empty(B, V, dtype=v.dtype, device=dev)
        prm.scores_out, prm.stride_scores = scores.data_ptr(), scores.stride(0)
    elif V > lib.vdd_lds_row_capacity(_DT[v.dtype]):
        workspace = workspace if workspace is not None else torch.empty(B, V, dtype=v.dtype, device=dev)
        prm.workspace, prm.stride_workspace = workspace.data_ptr(), workspace.stride(0)
        keep.append(workspace)
    top_prob = top_tok = None
    if n_top > 0:
        top_prob = torch.empty(B, n_top, dtype=torch.float32, device=dev)
        top_tok = torch.empty(B, n_top, dtype=torch.long, device=dev)
        prm.top_prob, prm.top_tok, prm.n_top = top_prob.data_ptr(), top_tok.data_ptr(), n_top
    status = status_out if status_out is not None else torch.empty(B, dtype=torch.int32, device=dev)
    prm.row_status = status.data_ptr()
    st = torch.cuda.current_stream(dev).cuda_stream if stream is None else stream
    with torch.cuda.device(dev):
        _lib.check(lib.vdd_contrast_sample(C.byref(prm), C.c_void_p(st)))
    return SampleOutput(tokens, scores, top_prob, top_tok, status)


KERNEL_THREADS = 512      # threads per row of vdd_contrast_sample_kernel (csrc/vdd_contrast_sample.hip: BLOCK)


def thread_major_order(V: int, dtype: torch.dtype) -> "list[int]":
    """Element enumeration order of the kernel's inverse-CDF draw (chunk ch -> thread
    ch % KERNEL_THREADS; a thread walks its chunks in increasing ch).  Any fixed order gives an
    exact categorical sample; this is exposed so tests can recompute the drawn token."""
    epc = 4 if dtype == torch.float32 else 8
    nch = (V + epc - 1) // epc
    order = []
    for t in range(min(KERNEL_THREADS, nch)):
        for ch in range(t, nch, KERNEL_THREADS):
            order.extend(i for i in range(ch * epc, min(V, ch * epc + epc)))
    return order
