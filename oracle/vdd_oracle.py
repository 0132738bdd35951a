"""CPU ORACLE — test infrastructure, NOT product code.

This file restates, on plain torch-CPU eager ops, the algorithm of the reference
contrastive-decoding loop so that the HIP path can be checked against it on any
box (the reference Python itself never travels to the GPU box).

Who may import this: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
The product package (llava-align_amd/) must never import it and fails loudly when
the HIP library is missing.

Parity status: PINNED against the reference itself.  tests/golden/make_golden.py
drives the unmodified reference `sample()` (vcd_utils/vcd_sample.py:25-323, loaded
through the two runtime shims of SURVEY.md Appendix B) in the build container and
commits its inputs/outputs under tests/golden/*.npz; tests/test_oracle_golden.py
checks this restatement against those vectors bit-for-bit (per-step rows in both torch-CPU and torch-GPU
scalar arithmetic, loop traces, EOS/pad, logits-processor runs, noise, calibration).  Version caveat: the
fixtures were produced with torch 2.10 / transformers 5.15 warpers, not the
reference's (unpinned, ~4.31-era) versions.

Every function cites the reference file:line it follows (paths relative to the
reference repo root).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import torch

IMAGE_TOKEN_INDEX = -200  # experiments/llava/constants.py:8
NEG_INF = -float("inf")


# --------------------------------------------------------------------------------------
# Per-step row arithmetic (vcd_utils/vcd_sample.py:185-202)
# --------------------------------------------------------------------------------------
def average_branches(logits_unk: torch.Tensor, logits_none: torch.Tensor) -> torch.Tensor:
    """vcd_sample.py:185 — `(cd + dd) / 2`, two roundings in the model dtype."""
    return (logits_unk + logits_none) / 2


def contrast_and_mask(v: torch.Tensor, c: torch.Tensor, alpha: float, beta: float) -> torch.Tensor:
    """vcd_sample.py:188-194.

    cutoff = log(beta) + max_j v_j        (:191; log(beta) is a 0-dim fp32 tensor, the add
                                           result takes v's dtype)
    diffs  = (1+alpha)*v - alpha*c        (:193; three roundings in v's dtype)
    out    = where(v < cutoff, -inf, diffs)  (:194; the mask tests the ORIGINAL v, strict <)
    """
    if GPU_SCALAR:
        # torch on a GPU: the 0-dim CPU tensor log(beta) enters the add as an fp32 SCALAR (not demoted to v's dtype first):
        # fl_dtype(float(max) + fl32(log beta)), one rounding
        cutoff = (v.max(dim=-1, keepdim=True).values.float() + torch.log(torch.tensor(beta))).to(v.dtype) if v.dtype != torch.float32 \
            else torch.log(torch.tensor(beta)) + v.max(dim=-1, keepdim=True).values
    else:
        cutoff = torch.log(torch.tensor(beta)) + v.max(dim=-1, keepdim=True).values
    diffs = (1 + alpha) * v - alpha * c
    return diffs.masked_fill(v < cutoff, NEG_INF)


# False: torch-CPU scalar arithmetic (what tests/golden/kernel_vectors.* were made with: the reference run on the build container's
# CPU).  True: torch-GPU's (what the reference computes when its tensors live on a GPU, as in every driver): the plausibility cutoff
# adds log(beta) as an fp32 scalar, and a division by a Python scalar is a multiplication by its fp32 reciprocal.  The two differ
# by at most 1 ulp of the model dtype; tests/golden/kernel_vectors_gpu_scalar.* pins this form.
GPU_SCALAR = False


def warp_temperature(x: torch.Tensor, temperature: float) -> torch.Tensor:
    """HF TemperatureLogitsWarper (called at vcd_sample.py:198): scores / T."""
    if GPU_SCALAR:
        inv = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(float(temperature), dtype=torch.float32)     # fl32(1 / T)
        return (x.float() * inv).to(x.dtype)
    return x / temperature


def warp_top_k(x: torch.Tensor, top_k: int, min_keep: int = 1) -> torch.Tensor:
    """HF TopKLogitsWarper (vcd_sample.py:198): scores < kth-largest -> -inf (ties kept)."""
    k = min(max(top_k, min_keep), x.size(-1))
    kth = torch.topk(x, k)[0][..., -1, None]
    return x.masked_fill(x < kth, NEG_INF)


def warp_top_p(x: torch.Tensor, top_p: float, min_keep: int = 1) -> torch.Tensor:
    """HF TopPLogitsWarper (vcd_sample.py:198): ascending sort, softmax, cumsum,
    drop while cum <= 1-p, always keep the last `min_keep`."""
    srt, idx = torch.sort(x, descending=False)
    cum = srt.softmax(dim=-1).cumsum(dim=-1)
    drop = cum <= (1 - top_p)
    drop[..., -min_keep:] = 0
    drop = drop.scatter(1, idx, drop)
    return x.masked_fill(drop, NEG_INF)


@dataclass
class WarpConfig:
    """The warper list HF's generate() builds for `do_sample=True` (4.31-era rules,
    SURVEY.md A.1): temperature iff not in (None, 1.0); top-k iff not in (None, 0);
    top-p iff not None and < 1.0.  Order: temperature -> top-k -> top-p."""
    temperature: Optional[float] = None
    top_k: Optional[int] = None
    top_p: Optional[float] = None
    min_keep: int = 1

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.temperature is not None and self.temperature != 1.0:
            x = warp_temperature(x, float(self.temperature))
        if self.top_k is not None and self.top_k != 0:
            x = warp_top_k(x, int(self.top_k), self.min_keep)
        if self.top_p is not None and self.top_p < 1.0:
            x = warp_top_p(x, float(self.top_p), self.min_keep)
        return x


# --------------------------------------------------------------------------------------
# Logits processors the reference's drivers put between contrast and warp (vcd_sample.py:197 / :204):
# `logits_processor(input_ids, scores)`.  HF builds the list in this order [ext, transformers
# `_get_logits_processor`]: repetition penalty, min_length, min_new_tokens, ..., then the caller's
# own (Qwen appends its stop-words processor, modeling_qwen.py:1061-1075).
# --------------------------------------------------------------------------------------
class RepetitionPenalty:
    """HF RepetitionPenaltyLogitsProcessor [ext] (InstructBLIP passes repetition_penalty, blip2_vicuna_instruct.py:400):
    gather the scores of every id in input_ids, `score < 0 ? score * p : score / p`, scatter back."""

    def __init__(self, penalty: float):
        self.penalty = float(penalty)

    def __call__(self, input_ids, scores):
        if input_ids.shape[1] == 0:
            return scores
        score = torch.gather(scores, 1, input_ids)
        if GPU_SCALAR and score.dtype != torch.float32:
            # torch-GPU divides a tensor by a python scalar as a multiplication by fl32(1 / p) in fp32 opmath, rounded once to the
            # tensor's dtype (ATen BinaryDivTrueKernel.cu: `inv_b = 1 / b`); torch-CPU (the form the goldens pin) does a true division
            inv = torch.tensor(1.0, dtype=torch.float32) / torch.tensor(self.penalty, dtype=torch.float32)
            quot = (score.float() * inv).to(score.dtype)
        elif GPU_SCALAR:
            quot = score * (torch.tensor(1.0, dtype=torch.float32) / torch.tensor(self.penalty, dtype=torch.float32))
        else:
            quot = score / self.penalty
        score = torch.where(score < 0, score * self.penalty, quot)
        return scores.scatter(1, input_ids, score)


class MinLength:
    """HF MinLengthLogitsProcessor [ext] (min_length, blip2_vicuna_instruct.py:397): eos ids score -inf while the row is shorter
    than min_length (prompt included)."""

    def __init__(self, min_length: int, eos_token_id):
        self.min_length, self.eos = int(min_length), [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id)

    def __call__(self, input_ids, scores):
        if input_ids.shape[-1] < self.min_length:
            scores = scores.clone()
            scores[:, self.eos] = NEG_INF
        return scores


class MinNewTokens:
    """HF MinNewTokensLengthLogitsProcessor [ext] (min_new_tokens=1, MME/run_qwen.py:194): eos ids score -inf until
    min_new_tokens ids have been generated behind the prompt."""

    def __init__(self, prompt_length: int, min_new_tokens: int, eos_token_id):
        self.prompt_length, self.min_new = int(prompt_length), int(min_new_tokens)
        self.eos = [eos_token_id] if isinstance(eos_token_id, int) else list(eos_token_id)

    def __call__(self, input_ids, scores):
        if input_ids.shape[-1] - self.prompt_length < self.min_new:
            scores = scores.clone()
            scores[:, self.eos] = NEG_INF
        return scores


class StopWords:
    """experiments/Qwen_VL/qwen_generation_utils.py:305-385: a row whose ids END with one of the stop sequences gets
    scores[row, eos] = 2**15 (:352-359); sequences equal to [eos] are dropped at construction (:340-344)."""

    def __init__(self, stop_words_ids, eos_token_id: int):
        self.stop = [list(w) for w in stop_words_ids if list(w) != [eos_token_id]]
        self.eos = int(eos_token_id)

    def __call__(self, input_ids, scores):
        for i, row in enumerate(input_ids.tolist()):
            if any(len(w) <= len(row) and row[len(row) - len(w):] == w for w in self.stop):      # _tokens_match, :361-372
                scores[i, self.eos] = float(2 ** 15)
        return scores


class ProcessorList(list):
    def __call__(self, input_ids, scores):
        for p in self:
            scores = p(input_ids, scores)
        return scores


def step_scores(v: torch.Tensor, c: Optional[torch.Tensor], d: Optional[torch.Tensor],
                alpha: float, beta: float, warp: WarpConfig,
                processors: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
                input_ids: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One decode step's post-warp `next_token_scores` row(s) — what `output_scores`
    returns (vcd_sample.py:200,240).  `c is None` is the plain path (:204-205).
    `processors(input_ids, scores)` is the reference's `logits_processor` (:197 / :204)."""
    if c is not None:
        if d is not None:
            c = average_branches(c, d)
        x = contrast_and_mask(v, c, alpha, beta)
    else:
        x = v
    if processors is not None:
        x = processors(input_ids, x)
    return warp(x)


def pad_finished(tokens: torch.Tensor, unfinished: torch.Tensor, pad_id: int) -> torch.Tensor:
    """vcd_sample.py:260."""
    return tokens * unfinished + pad_id * (1 - unfinished)


def update_unfinished(unfinished: torch.Tensor, tokens: torch.Tensor, eos_ids: Sequence[int]) -> torch.Tensor:
    """vcd_sample.py:286-288 — a row finishes once it emits any EOS id."""
    eos = torch.tensor(list(eos_ids))
    hit_none = tokens.tile(eos.shape[0], 1).ne(eos.unsqueeze(1)).prod(dim=0)
    return unfinished.mul(hit_none)


# --------------------------------------------------------------------------------------
# The loop (vcd_utils/vcd_sample.py:25-323), restated around a minimal model protocol
# --------------------------------------------------------------------------------------
@dataclass
class LoopResult:
    sequences: torch.Tensor
    scores: List[torch.Tensor] = field(default_factory=list)
    schedule: List[tuple] = field(default_factory=list)   # (step, branch, ids_len, mask_len, has_image, past_len)


def _strip_image_slot(ids: torch.Tensor, mask: torch.Tensor):
    """vcd_sample.py:157-160 (and :173-176).  `torch.where(ids != -200)[0]` on a 2-D
    tensor yields ROW indices; indexing mask columns with them re-reads column 0
    len-1 times.  Only meaningful at batch 1 — reproduced as is."""
    keep_rows = torch.where(ids != IMAGE_TOKEN_INDEX)[0]
    return ids[ids != IMAGE_TOKEN_INDEX].unsqueeze(0), mask[:, keep_rows]


def reference_loop(model, input_ids: torch.Tensor, *, warp: WarpConfig, max_length: int,
                   pad_token_id: Optional[int], eos_token_id, pick: Callable[[torch.Tensor], torch.Tensor],
                   processors=None, record_schedule: bool = False, **model_kwargs) -> LoopResult:
    """Restatement of the patched sample() loop.

    `model` follows the protocol sample() needs:
    prepare_inputs_for_generation[_cd], __call__ -> .logits/.past_key_values,
    _update_model_kwargs_for_generation.  `pick(probs)` stands in for
    torch.multinomial (vcd_sample.py:202,207) so tests can inject determinism.
    """
    if isinstance(eos_token_id, int):
        eos_token_id = [eos_token_id]                                   # :57-58
    unfinished = torch.ones(input_ids.shape[0], dtype=torch.long)       # :88
    kw_unk = kw_none = None                                             # :91
    out = LoopResult(sequences=input_ids)
    step = 0
    while True:
        def fwd(tag, inputs):
            if record_schedule:
                ids = inputs.get("input_ids")
                emb = inputs.get("inputs_embeds")
                past = inputs.get("past_key_values")
                am = inputs.get("attention_mask")
                out.schedule.append((step, tag,
                                     int(ids.shape[1]) if ids is not None else int(emb.shape[1]),
                                     int(am.shape[1]) if am is not None else -1,
                                     inputs.get("images") is not None,
                                     int(past[0][0].shape[-2]) if past else 0))
            return model(**inputs, return_dict=True, output_attentions=None, output_hidden_states=None)

        main = fwd("main", model.prepare_inputs_for_generation(input_ids, **model_kwargs))   # :106-114
        v = main.logits[:, -1, :]                                                           # :119
        use_cd = model_kwargs.get("images_cd") is not None                                  # :122
        use_dd = model_kwargs.get("use_dd")                                                 # :123
        use_dd_unk = model_kwargs.get("use_dd_unk")                                         # :124

        c = d = None
        o_cd = o_dd = None
        if use_cd or use_dd or use_dd_unk:                                                  # :147
            if use_cd:
                kw_unk = model_kwargs.copy()                                                # :149 (quirk A.3 #1)
                cd_inputs = model.prepare_inputs_for_generation_cd(input_ids, **kw_unk)     # :150
            else:
                kw_unk = model_kwargs.copy() if kw_unk is None else kw_unk                  # :152
                if use_dd_unk:
                    ids2 = input_ids.clone()
                    ids2[ids2 == IMAGE_TOKEN_INDEX] = 0                                     # :154-155  <unk>
                elif use_dd:
                    ids2, kw_unk["attention_mask"] = _strip_image_slot(input_ids, model_kwargs["attention_mask"])
                cd_inputs = model.prepare_inputs_for_generation_cd(ids2, **kw_unk)          # :161
            o_cd = fwd("cd", cd_inputs)                                                     # :163-168
            c = o_cd.logits[:, -1, :]                                                       # :169
            if use_dd_unk and use_dd:                                                       # :171
                kw_none = model_kwargs.copy() if kw_none is None else kw_none               # :172
                ids3, kw_none["attention_mask"] = _strip_image_slot(input_ids, model_kwargs["attention_mask"])
                o_dd = fwd("dd", model.prepare_inputs_for_generation_cd(ids3, **kw_none))   # :177-183
                d = o_dd.logits[:, -1, :]                                                   # :184
            alpha = model_kwargs.get("cd_alpha") if model_kwargs.get("cd_alpha") is not None else 0.5   # :188
            beta = model_kwargs.get("cd_beta") if model_kwargs.get("cd_beta") is not None else 0.1      # :189
            scores = step_scores(v, c, d, alpha, beta, warp, processors, input_ids)                  # :185-198
        else:
            scores = step_scores(v, None, None, 0.0, 0.0, warp, processors, input_ids)               # :204-205
        probs = torch.nn.functional.softmax(scores, dim=-1)                                 # :201 / :206
        tokens = pick(probs)                                                                # :202 / :207
        out.scores.append(scores)                                                           # :240

        if eos_token_id is not None:
            if pad_token_id is None:
                raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")  # :258-259
            tokens = pad_finished(tokens, unfinished, pad_token_id)                         # :260
        input_ids = torch.cat([input_ids, tokens[:, None]], dim=-1)                         # :263
        model_kwargs = model._update_model_kwargs_for_generation(main, model_kwargs, is_encoder_decoder=False)  # :266
        if o_cd is not None:
            kw_unk = model._update_model_kwargs_for_generation(o_cd, kw_unk, is_encoder_decoder=False)  # :271
        if o_dd is not None:
            kw_none = model._update_model_kwargs_for_generation(o_dd, kw_none, is_encoder_decoder=False)  # :275
        done = False
        if eos_token_id is not None:
            unfinished = update_unfinished(unfinished, tokens, eos_token_id)                # :286-288
            done = bool(unfinished.max() == 0)                                              # :291
        if input_ids.shape[-1] >= max_length:                                               # :295 MaxLengthCriteria
            done = True
        step += 1
        if done:
            break
    out.sequences = input_ids
    return out


def pick_argmax(probs: torch.Tensor) -> torch.Tensor:
    """Deterministic stand-in for multinomial: valid whenever the distribution has a
    single survivor (TopK(1) without ties), which is how 'greedy' is driven through
    sample()."""
    return probs.argmax(dim=-1)


def pick_multinomial(probs: torch.Tensor) -> torch.Tensor:
    return torch.multinomial(probs, num_samples=1).squeeze(1)


# --------------------------------------------------------------------------------------
# VCD noise branch input (vcd_utils/vcd_add_noise.py:3-28)
# --------------------------------------------------------------------------------------
def diffusion_schedule(num_steps: int = 1000):
    """vcd_add_noise.py:7-16 — sigmoid beta schedule; returns (sqrt(abar), sqrt(1-abar))."""
    betas = torch.sigmoid(torch.linspace(-6, 6, num_steps)) * (0.5e-2 - 1e-5) + 1e-5
    abar = torch.cumprod(1 - betas, dim=0)
    return torch.sqrt(abar), torch.sqrt(1 - abar)


def add_diffusion_noise(image: torch.Tensor, noise_step: int, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """vcd_add_noise.py:18-28 — x_t = sqrt(abar_t) x_0 + sqrt(1-abar_t) eps."""
    a, b = diffusion_schedule()
    eps = torch.randn_like(image) if noise is None else noise
    return a[noise_step] * image + b[noise_step] * eps


# --------------------------------------------------------------------------------------
# Post-hoc calibration (experiments/utils/metrics.py:8-41,102-125;
# experiments/eval/eval_pope_calibrate.py:65-74)
# --------------------------------------------------------------------------------------
def top_token_probs(scores_row: torch.Tensor, decode: Callable[[int], str], top_k: int = 10) -> dict:
    """metrics.py:102-113 — softmax in the scores dtype, .float(), top-k, decode,
    lower/strip, FIRST (highest-prob) occurrence of each string wins."""
    probs = torch.softmax(scores_row, dim=-1).float().cpu()
    p, t = torch.topk(probs, k=top_k)
    out = {}
    for prob, tok in zip(p[0], t[0]):
        s = decode(int(tok.item())).lower().strip()
        if s not in out:
            out[s] = prob.item()
    return out


def label_probs(token_probs: dict, label_dict=None) -> list:
    """metrics.py:115-125 — missing label -> 0."""
    label_dict = label_dict or {0: ["yes"], 1: ["no"]}
    return [sum(token_probs.get(a.lower(), 0) for a in answers) for _, answers in label_dict.items()]


def affine_calibrate(p, p_cf=None, mode: str = "diagonal_W"):
    """metrics.py:8-41 / eval_pope_calibrate.py:65-74 — q = W p + b, renormalised; float64 numpy."""
    import numpy as np
    p = np.asarray(p, dtype=np.float64)
    n = p.shape[0]
    if p_cf is None:
        W, b = np.identity(n), np.zeros([n, 1])
    elif mode == "diagonal_W":
        W, b = np.linalg.inv(np.identity(n) * np.asarray(p_cf, dtype=np.float64)), np.zeros([n, 1])
    elif mode == "identity_W":
        W, b = np.identity(n), -1 * np.expand_dims(np.asarray(p_cf, dtype=np.float64), axis=-1)
    else:
        raise AssertionError(mode)
    p = p / np.sum(p)
    q = np.matmul(W, np.expand_dims(p, axis=-1)) + b
    q /= np.sum(q)
    return q, int(np.argmax(q))
