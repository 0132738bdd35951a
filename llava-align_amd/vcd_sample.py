"""Drop-in for the reference's `vcd_utils/vcd_sample.py`: `sample()` + `evolve_vcd_sampling()`.

Same call surface (vcd_sample.py:25-41) and the same extra generate() kwargs read out
of model_kwargs — images_cd, use_dd, use_dd_unk, cd_alpha, cd_beta (:122-124,188-189) —
but the per-step tail (average, contrast, plausibility mask, warpers, softmax, draw,
pad/EOS bookkeeping; :185-207,257-260,285-288) is ONE HIP kernel launch
(`vdd_contrast_sample`, include/vdd_hip.h) instead of ~15 eager kernels, the id buffer
is preallocated instead of torch.cat-grown (:263), the per-step deep copy of input_ids
(:154,160,176) is replaced by appending the new token to per-branch id buffers, and
the host synchronises at most once per step (reference: twice, :291,:295).

This generic loop drives ANY model exposing the reference's protocol
(prepare_inputs_for_generation[_cd], forward, _update_model_kwargs_for_generation), one
forward per branch like the reference.  The branch-batched native engine lives in
engine.py; both share the kernel.

Reference-compat behaviours kept on purpose, each covered by a test:
  #1 VCD re-copies the main kwargs every step, so from step 1 on the cd branch runs on
     the main KV cache (c == v);  #2 use_dd drops the -200 slot with batch-1 semantics;
  #5 do_sample=False never reaches this function in the reference era — under
     transformers>=4.39 `_sample` receives it and we decode greedily WITHOUT contrast.
"""
from __future__ import annotations

import copy
import inspect
import warnings
from typing import List, Optional, Union

import torch

from .sampling import WarpSpec, contrast_sample

IMAGE_TOKEN_INDEX = -200   # experiments/llava/constants.py:8 (imported at vcd_sample.py:23)
_NEG_INF = -float("inf")


# ------------------------------------------------------------------ warper / criteria parsing
def _split_warpers(processors, warpers):
    """Returns (python_processors, WarpSpec, fused).  HF's Temperature/TopK/TopP warpers with
    the default -inf filter are absorbed into the kernel; anything else runs in Python
    between a contrast-only launch and a plain-path launch (order kept: :197 then :198)."""
    from transformers.generation.logits_process import (TemperatureLogitsWarper, TopKLogitsWarper,
                                                        TopPLogitsWarper)
    procs = list(processors) if processors is not None else []
    warps = list(warpers) if warpers is not None else []
    # transformers >= 4.39 hands `_sample` one merged list: peel the trailing warpers off it
    if not warps:
        while procs and isinstance(procs[-1], (TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)):
            warps.insert(0, procs.pop())
    spec = WarpSpec()
    stage = 0
    for w in warps:
        if isinstance(w, TemperatureLogitsWarper) and stage == 0:
            spec.temperature, stage = float(w.temperature), 1
        elif isinstance(w, TopKLogitsWarper) and stage <= 1 and w.filter_value == _NEG_INF:
            spec.top_k, stage = int(w.top_k), 2
            spec.min_keep = max(spec.min_keep, int(getattr(w, "min_tokens_to_keep", 1)))
        elif isinstance(w, TopPLogitsWarper) and stage <= 2 and w.filter_value == _NEG_INF:
            spec.top_p, stage = float(w.top_p), 3
            spec.min_keep = max(spec.min_keep, int(w.min_tokens_to_keep))
        else:
            return procs + warps, WarpSpec(), False      # unknown / re-ordered warper: all of it in Python
    return procs, spec, len(procs) == 0


def _parse_criteria(stopping_criteria, max_length):
    """Pulls MaxLengthCriteria / EosTokenCriteria out of the list; the rest is evaluated in Python."""
    rest, eos_from_criteria = [], None
    for c in (stopping_criteria or []):
        name = type(c).__name__
        if name == "MaxLengthCriteria":
            max_length = c.max_length if max_length is None else min(max_length, c.max_length)
        elif name == "EosTokenCriteria":
            eos_from_criteria = c.eos_token_id
        else:
            rest.append(c)
    return rest, max_length, eos_from_criteria


def _strip_image_slot(ids: torch.Tensor, mask: torch.Tensor):
    """vcd_sample.py:157-160 / :173-176 with their batch-1 semantics:
    the mask is re-indexed with ROW indices, i.e. column 0 repeated."""
    keep_rows = torch.where(ids != IMAGE_TOKEN_INDEX)[0]
    return ids[ids != IMAGE_TOKEN_INDEX].unsqueeze(0), mask[:, keep_rows]


def _is_mutable_cache(obj) -> bool:
    """transformers >= 4.36 caches are objects updated IN PLACE by forward(); the reference era used immutable tuples."""
    return obj is not None and not isinstance(obj, (tuple, list)) and hasattr(obj, "get_seq_length")


def _branch_kwargs(model_kwargs: dict) -> dict:
    """`model_kwargs.copy()` (vcd_sample.py:149,152,172) plus a private copy of a mutable cache object: with the
    reference's shallow copy two branches would append to the SAME cache under transformers 5.x."""
    kw = model_kwargs.copy()
    for k, v in model_kwargs.items():
        if _is_mutable_cache(v):
            kw[k] = copy.deepcopy(v)
    return kw


_PREP_EXTRAS = {}


def _accepts(fn, name):
    params = inspect.signature(fn).parameters
    return name in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())


def _prepare(model, fn, ids, kwargs, step):
    """Calls a prepare_inputs_for_generation[_cd] of either era.  Implementations of transformers >= 4.5x slice the ids
    by `next_sequence_length` (the reference-era ones by `past_key_values`); the era is read off the model's main
    prepare_inputs_for_generation, and the extras are handed to any variant that can take them (explicitly or via **kw)."""
    key = (type(model), getattr(fn, "__func__", fn))
    if key not in _PREP_EXTRAS:
        main = inspect.signature(model.prepare_inputs_for_generation).parameters
        _PREP_EXTRAS[key] = ("next_sequence_length" in main and _accepts(fn, "next_sequence_length"),
                             "is_first_iteration" in main and _accepts(fn, "is_first_iteration"))
    has_nsl, has_first = _PREP_EXTRAS[key]
    extra = {}
    if has_nsl:
        extra["next_sequence_length"] = None if step == 0 else (1 if kwargs.get("use_cache", True) is not False else None)
    if has_first:
        extra["is_first_iteration"] = step == 0
    return fn(ids, **extra, **kwargs)


class _Branch:
    """Per-branch generation state: its own model_kwargs (KV cache, mask) and id buffer."""

    def __init__(self, kwargs, ids):
        self.kwargs, self.ids = kwargs, ids

    def drop_image_slot(self, ids: torch.Tensor, main_mask: torch.Tensor):
        """The image-token-dropped branch's inputs for this step.  The reference re-filters the
        whole id tensor every step (:157-160); the filter result only ever grows by the new
        token, so it is computed once (one boolean-index sync) and appended to afterwards.
        Its mask is `main_mask[:, rows]` = column 0 repeated (quirk #2), rebuilt as a view."""
        if self.ids is None:
            if ids.shape[0] != 1:
                raise ValueError("use_dd is batch-1 only: the reference flattens the batch at "
                                 "vcd_sample.py:160 (SURVEY.md A.3 #2); use use_dd_unk for batches")
            self.ids, _ = _strip_image_slot(ids, main_mask)
        else:
            self.ids = torch.cat([self.ids, ids[:, -1:]], dim=-1)
        self.kwargs["attention_mask"] = main_mask[:, :1].expand(main_mask.shape[0], self.ids.shape[1])


# ------------------------------------------------------------------ the loop
@torch.no_grad()
def sample(self, input_ids: torch.LongTensor, logits_processor=None, stopping_criteria=None, logits_warper=None,
           max_length: Optional[int] = None, pad_token_id: Optional[int] = None,
           eos_token_id: Optional[Union[int, List[int]]] = None, output_attentions: Optional[bool] = None,
           output_hidden_states: Optional[bool] = None, output_scores: Optional[bool] = None,
           return_dict_in_generate: Optional[bool] = None, synced_gpus: bool = False, streamer=None,
           **model_kwargs):
    gc = self.generation_config
    if max_length is not None:
        warnings.warn("`max_length` is deprecated in this function, use"
                      " `stopping_criteria=StoppingCriteriaList(MaxLengthCriteria(max_length=max_length))` instead.",
                      UserWarning)                                                               # :45-50
    other_criteria, max_length, eos_crit = _parse_criteria(stopping_criteria, max_length)
    pad_token_id = pad_token_id if pad_token_id is not None else gc.pad_token_id                 # :53
    eos_token_id = eos_token_id if eos_token_id is not None else gc.eos_token_id                 # :54
    if eos_token_id is None and eos_crit is not None:
        eos_token_id = eos_crit
    if isinstance(eos_token_id, int):
        eos_token_id = [eos_token_id]                                                            # :57-58
    if torch.is_tensor(eos_token_id):
        eos_token_id = eos_token_id.reshape(-1).tolist()
    if torch.is_tensor(pad_token_id):
        pad_token_id = int(pad_token_id.reshape(-1)[0].item())
    output_scores = output_scores if output_scores is not None else gc.output_scores              # :60
    output_attentions = output_attentions if output_attentions is not None else gc.output_attentions
    output_hidden_states = output_hidden_states if output_hidden_states is not None else gc.output_hidden_states
    return_dict_in_generate = (return_dict_in_generate if return_dict_in_generate is not None
                               else gc.return_dict_in_generate)
    if eos_token_id is not None and pad_token_id is None:
        raise ValueError("If `eos_token_id` is defined, make sure that `pad_token_id` is defined.")  # :258-259

    dev = input_ids.device
    B, L0 = input_ids.shape
    py_procs, spec, fused = _split_warpers(logits_processor, logits_warper)
    want_scores = bool(return_dict_in_generate and output_scores)
    scores = [] if want_scores else None
    attns = [] if (return_dict_in_generate and output_attentions) else None
    hiddens = [] if (return_dict_in_generate and output_hidden_states) else None

    use_cd = model_kwargs.get("images_cd") is not None                                           # :122
    use_dd = bool(model_kwargs.get("use_dd"))                                                    # :123
    use_dd_unk = bool(model_kwargs.get("use_dd_unk"))                                            # :124
    contrast = use_cd or use_dd or use_dd_unk
    alpha = model_kwargs.get("cd_alpha") if model_kwargs.get("cd_alpha") is not None else 0.5    # :188
    beta = model_kwargs.get("cd_beta") if model_kwargs.get("cd_beta") is not None else 0.1       # :189
    # extension (not in the reference): deterministic draw = arg-max of the final distribution,
    # ties to the lowest index.  With top_k=1 this is the reference's sample() whenever its
    # multinomial has a single survivor, and stays deterministic when fp16 ties leave several.
    greedy = bool(model_kwargs.pop("cd_greedy", False))

    # id buffer: preallocated when the length bound is known (replaces torch.cat at :263)
    cap = max_length if max_length is not None else L0 + 64
    ids_buf = torch.empty(B, max(cap, L0 + 1), dtype=torch.long, device=dev)
    ids_buf[:, :L0] = input_ids
    cur = L0
    unfinished = torch.ones(B, dtype=torch.long, device=dev)                                     # :88
    eos_t = torch.tensor(eos_token_id, dtype=torch.long, device=dev) if eos_token_id is not None else None
    statuses = []
    fwd = dict(return_dict=True, output_attentions=output_attentions, output_hidden_states=output_hidden_states)

    unk = none = None
    this_peer_finished = False
    while True:
        if synced_gpus:                                                                          # :94-102
            import torch.distributed as dist
            flag = torch.tensor(0.0 if this_peer_finished else 1.0, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.SUM)
            if flag.item() == 0.0:
                break
        ids = ids_buf[:, :cur]
        step = cur - L0
        if step == 0 and contrast:
            # pristine per-branch kwargs, taken BEFORE the main forward: a mutable cache object is filled in place by it
            # (in the reference era model_kwargs still held no cache at this point, :149/:152/:172)
            fresh = [_branch_kwargs(model_kwargs) for _ in range(2)]
        out_main = self(**_prepare(self, self.prepare_inputs_for_generation, ids, model_kwargs, step), **fwd)   # :106-114
        if synced_gpus and this_peer_finished:
            continue
        v = out_main.logits[:, -1, :]                                                            # :119
        c = d = None
        out_cd = out_dd = None
        if contrast:
            vcd_shortcut = use_cd and step > 0 and _is_mutable_cache(model_kwargs.get("past_key_values"))
            if vcd_shortcut:
                # quirk #1 with a mutable cache object: re-running the cd branch on the main cache (what :149 amounts
                # to) would append to it twice; its logits equal the main branch's anyway, so take c = v.
                cd_inputs = None
            elif use_cd:
                unk = _Branch(model_kwargs.copy() if step > 0 else fresh[0], ids)                # :149 (quirk #1)
                cd_inputs = _prepare(self, self.prepare_inputs_for_generation_cd, ids, unk.kwargs, step)      # :150
            else:
                if unk is None:                                                                  # :152
                    if use_dd_unk:
                        b_ids = ids.clone()
                        b_ids[b_ids == IMAGE_TOKEN_INDEX] = 0                                    # :154-155 <unk>
                        unk = _Branch(fresh[0], b_ids)
                    else:
                        unk = _Branch(fresh[0], None)
                elif use_dd_unk:
                    unk.ids = torch.cat([unk.ids, ids[:, -1:]], dim=-1)                          # new token only
                if not use_dd_unk:                                                               # use_dd alone
                    unk.drop_image_slot(ids, model_kwargs["attention_mask"])                     # :157-160
                cd_inputs = _prepare(self, self.prepare_inputs_for_generation_cd, unk.ids, unk.kwargs, step)   # :161
            if cd_inputs is not None:
                out_cd = self(**cd_inputs, **fwd)                                                # :163-168
                c = out_cd.logits[:, -1, :]                                                      # :169
            else:
                c = v
            if use_dd and use_dd_unk:                                                            # :171
                if none is None:
                    none = _Branch(fresh[1], None)                                               # :172
                none.drop_image_slot(ids, model_kwargs["attention_mask"])                        # :173-176
                out_dd = self(**_prepare(self, self.prepare_inputs_for_generation_cd, none.ids, none.kwargs, step), **fwd)   # :177-183
                d = out_dd.logits[:, -1, :]                                                      # :184

        if cur >= ids_buf.shape[1]:                                                              # unbounded run: grow
            ids_buf = torch.cat([ids_buf, torch.empty(B, ids_buf.shape[1], dtype=torch.long, device=dev)], dim=1)
        tok_col = ids_buf[:, cur]
        eos_kw = dict(eos_ids=eos_t, pad_id=pad_token_id, unfinished=unfinished) if eos_t is not None else {}
        if fused:
            r = contrast_sample(v, c, d, alpha=alpha, beta=beta, warp=spec, out_tokens=tok_col,
                                return_scores=want_scores, pick_argmax=greedy, **eos_kw)
        else:
            # a Python logits_processor sits between contrast and warp (e.g. Qwen's
            # StopWordsLogitsProcessor, qwen_generation_utils.py:352-359): contrast-only launch,
            # processors in torch, then a plain-path launch that warps and draws.
            x = contrast_sample(v, c, d, alpha=alpha, beta=beta, no_sample=True, return_scores=True).scores if contrast else v
            for proc in py_procs:
                x = proc(ids, x)
            r = contrast_sample(x, None, None, warp=spec, out_tokens=tok_col, return_scores=want_scores,
                                pick_argmax=greedy, **eos_kw)
        statuses.append(r.status)
        if want_scores:
            scores.append(r.scores)                                                              # :240
        if attns is not None:
            attns.append(out_main.attentions)
        if hiddens is not None:
            hiddens.append(out_main.hidden_states)
        cur += 1
        if streamer is not None:
            streamer.put(tok_col.cpu())                                                          # :265
        model_kwargs = self._update_model_kwargs_for_generation(out_main, model_kwargs,
                                                                is_encoder_decoder=self.config.is_encoder_decoder)  # :266
        if out_cd is not None:
            unk.kwargs = self._update_model_kwargs_for_generation(out_cd, unk.kwargs,
                                                                  is_encoder_decoder=self.config.is_encoder_decoder)  # :271
        if out_dd is not None:
            none.kwargs = self._update_model_kwargs_for_generation(out_dd, none.kwargs,
                                                                   is_encoder_decoder=self.config.is_encoder_decoder)  # :275
        if eos_t is not None and bool((unfinished.max() == 0).item()):                           # :291 (the one sync)
            this_peer_finished = True
        if max_length is not None and cur >= max_length:                                         # :295 MaxLengthCriteria
            this_peer_finished = True
        for crit in other_criteria:
            res = crit(ids_buf[:, :cur], tuple(scores) if scores is not None else None)
            if bool(res.all().item()) if torch.is_tensor(res) else bool(res):
                this_peer_finished = True
        if this_peer_finished and not synced_gpus:
            break

    if streamer is not None:
        streamer.end()
    bad = torch.stack(statuses).ne(0).any() if statuses else None
    if bad is not None and bool(bad.item()):
        raise RuntimeError("probability tensor contains either `inf`, `nan` or element < 0")     # multinomial, :202
    sequences = ids_buf[:, :cur]
    if return_dict_in_generate:
        from transformers.generation.utils import GenerateDecoderOnlyOutput
        return GenerateDecoderOnlyOutput(sequences=sequences, scores=tuple(scores) if scores is not None else None,
                                         attentions=tuple(attns) if attns is not None else None,
                                         hidden_states=tuple(hiddens) if hiddens is not None else None)   # :304-321
    return sequences


def _sample_v5(self, input_ids, logits_processor, stopping_criteria, generation_config, synced_gpus=False,
               streamer=None, **model_kwargs):
    """transformers >= 4.39 entry (`GenerationMixin._sample`): warpers arrive merged into
    `logits_processor`, EOS as a stopping criterion, flags in `generation_config`."""
    if not generation_config.do_sample:
        if any(model_kwargs.get(k) for k in ("use_dd", "use_dd_unk")) or model_kwargs.get("images_cd") is not None:
            warnings.warn("do_sample=False: the reference patches only sample(), so greedy decoding runs "
                          "WITHOUT contrastive decoding (SURVEY.md A.3 #5). Use do_sample=True with top_k=1.")
        for k in ("images_cd", "use_dd", "use_dd_unk"):
            model_kwargs.pop(k, None)
        from transformers.generation.logits_process import TopKLogitsWarper
        logits_processor = list(logits_processor) + [TopKLogitsWarper(top_k=1)]
        model_kwargs["cd_greedy"] = True            # HF greedy takes the FIRST index of the maximum: arg-max, not a draw among ties
    pad = getattr(generation_config, "_pad_token_tensor", None)
    eos = getattr(generation_config, "_eos_token_tensor", None)
    if eos is None:
        eos = generation_config.eos_token_id         # the per-call config wins over the model's (generate(eos_token_id=X))
    return sample(self, input_ids, logits_processor=logits_processor, stopping_criteria=stopping_criteria,
                  pad_token_id=pad if pad is not None else generation_config.pad_token_id, eos_token_id=eos,
                  output_attentions=generation_config.output_attentions,
                  output_hidden_states=generation_config.output_hidden_states,
                  output_scores=generation_config.output_scores,
                  return_dict_in_generate=generation_config.return_dict_in_generate,
                  synced_gpus=synced_gpus, streamer=streamer, **model_kwargs)


def evolve_vcd_sampling():
    """vcd_sample.py:325-326 — installs the loop process-wide.  Idempotent.  Sets the
    4.31-era `GenerationMixin.sample` and, where it exists, the >=4.39 `_sample`."""
    import transformers
    mixin = transformers.generation.utils.GenerationMixin
    mixin.sample = sample
    if hasattr(mixin, "_sample"):
        mixin._sample = _sample_v5
