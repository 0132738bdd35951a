"""InstructBLIP POPE driver over the native front-end + engine (BASELINE config #5): the batched replacement of the reference's
per-question loop experiments/eval/calibrate/blip_calibrate.py:57-113 and of the generate() it calls
(experiments/lavis/models/blip2_models/blip2_vicuna_instruct.py:240-418).

Per question the reference runs THREE `model.generate({"image", "prompt"}, use_nucleus_sampling=True, num_beams=1, top_p=...,
repetition_penalty=1, ...)` calls at B = 1 (blip_calibrate.py:83-98):
  main    the image, with `images_cd = add_diffusion_noise(image, noise_step)` when --use_cd (VCD; cd_alpha is NOT forwarded, so the
          sampler's default 0.5 applies), cd_beta                     -> `text`, `naive` (step-0 top-10 label dict)
  noise   image = add_diffusion_noise(image, 999), plain sampling (images_cd=None)    -> `noise`
  zeros   image = zeros_like(image), plain sampling                                   -> `zeros`
Each generate() builds `inputs_embeds = [Q-Former(image, instruction) -> llm_proj | LLM token embeddings of the prompt]`
(blip2_vicuna_instruct.py:333-388; here: blip_frontend.InstructBlipFrontEnd.build) and calls the LLM's patched sample() with
max_length=256, min_length=1, temperature=1, repetition_penalty, top_p; afterwards token id 0 is mapped to 2 (:414) before decoding.
Only the step-0 scores of the two prior passes are used, so they decode ONE token here.

Tokenisers stay with the caller: `tokenize_llm(prompt) -> ids` (the Vicuna tokenizer, BOS included) and
`tokenize_qformer(prompt) -> ids` (the BERT tokenizer the Q-Former reads the instruction with, truncated to max_txt_len).
"""
from __future__ import annotations

import json
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import calibrate as C
from .blip_frontend import InstructBlipFrontEnd
from .engine import VddLlavaEngine
from .vcd_add_noise import add_diffusion_noise

QUESTION_SUFFIX = " Please answer this question with one word."      # blip_calibrate.py:42,74


def map_pad_to_eos(tokens: torch.Tensor) -> torch.Tensor:
    """blip2_vicuna_instruct.py:414: `outputs[outputs == 0] = 2  # convert output id 0 to 2 (eos_token_id)`."""
    out = tokens.clone()
    out[out == 0] = 2
    return out


def run_blip_pope(engine: VddLlavaEngine, front: InstructBlipFrontEnd, questions: Sequence[dict],
                  tokenize_llm: Callable[[str], List[int]], tokenize_qformer: Callable[[str], List[int]],
                  decode: Callable[[List[int]], str], load_image: Callable[[str], torch.Tensor], answers_path: Optional[str] = None,
                  batch_questions: int = 128, use_cd: bool = False, noise_step: int = 500, cd_beta: Optional[float] = 0.1,
                  cd_alpha: Optional[float] = None, top_p: float = 1.0, top_k: Optional[int] = 50, temperature: float = 1.0,
                  repetition_penalty: float = 1.0,
                  max_length: int = 256, min_length: int = 1, eos_token_id=2, pad_token_id: Optional[int] = 2,
                  model_id: str = "instruct_blip", rank: Optional[int] = None, world: Optional[int] = None,
                  batch_invariant: Optional[bool] = None, **generate_kw) -> dict:
    """questions: POPE json lines (question_id, image, text[, label]).  Defaults are the reference driver's: top_p 1, temperature 1,
    top_k 50 (LAVIS never passes top_k, so HF's GenerationConfig default warps every sampled step: blip2_vicuna_instruct.py:390-410),
    repetition_penalty 1, max_length 256, min_length 1, cd_alpha left at the sampler's default (None -> 0.5), noise_step 500.
    generate_kw: seed, cd_greedy, sync_every ...  rank / world (default: the initialised torch.distributed group): every rank decodes
    its chunk of whole images (shard.ShardPlan; BASELINE config #5 runs on 4 GPUs), ONE collective gathers the results, rank 0 writes
    the file, every rank returns the full result.  Returns {"answers": [...], "scores": {...}}; the JSONL has the reference's
    fields (question_id, prompt, text, model_id, image, naive, noise, zeros, metadata; blip_calibrate.py:100-109)."""
    import contextlib
    from . import ops
    from .pope_driver import ResultRows, cut_at_eos
    from .shard import ShardPlan, resolve_batch_invariant
    order = sorted(range(len(questions)), key=lambda i: (questions[i]["image"], i))
    plan = ShardPlan([questions[i]["image"] for i in order], rank, world)
    mine = [order[p_] for p_ in plan.mine]
    if generate_kw.get("seed") is not None:
        generate_kw = dict(generate_kw, seed=int(generate_kw["seed"]) + plan.rank)
    decode_token = lambda t: decode([t])
    embed = engine.w.t["embed"]
    eos_set = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
    base_kw = dict(do_sample=True, top_p=top_p, top_k=top_k, temperature=temperature, num_beams=1, repetition_penalty=repetition_penalty,
                   min_length=min_length, eos_token_id=eos_token_id, pad_token_id=pad_token_id, n_top=10, cd_beta=cd_beta,
                   cd_alpha=cd_alpha, **generate_kw)
    rows = ResultRows(engine.device, max_length, pad_token_id if pad_token_id is not None else 0, n_sets=3)
    invariant = resolve_batch_invariant(batch_invariant, plan.world, generate_kw)    # (the LM side; the EVA-ViT / Q-Former front-end is per image / per question)
    with (ops.batch_invariant() if invariant else contextlib.nullcontext()):
        for b0 in range(0, len(mine), batch_questions):
            idx = mine[b0:b0 + batch_questions]
            qs = [questions[i] for i in idx]
            prompts = [q["text"] + QUESTION_SUFFIX for q in qs]
            llm_ids = [tokenize_llm(p) for p in prompts]
            qf_ids = [tokenize_qformer(p) for p in prompts]
            cache: Dict[str, torch.Tensor] = {}
            for q in qs:
                if q["image"] not in cache:
                    cache[q["image"]] = load_image(q["image"]).to(engine.device)
            # EVA-ViT once per DISTINCT clean image and once for the `zeros` image; the noised copies are drawn per question (fresh noise inside
            # the reference's loop, blip_calibrate.py:80-82, :96) and so are their ViT passes; the Q-Former reads the instruction: per question
            names = list(cache)
            where = [names.index(q["image"]) for q in qs]
            uniq = torch.stack([cache[n] for n in names])
            ie_u = front.image_embeds(uniq)
            ie_main = ie_u[where]
            imgs = uniq[where]
            emb = front.assemble(front.embeds_to_llm(ie_main, qf_ids), llm_ids, embed)
            emb_cd = None
            if use_cd:
                imgs_cd = torch.stack([add_diffusion_noise(im, noise_step) for im in imgs])
                emb_cd = front.assemble(front.embeds_to_llm(front.image_embeds(imgs_cd), qf_ids), llm_ids, embed)
            main = engine.generate(None, inputs_embeds=emb, images_cd=emb_cd, max_length=max_length, **base_kw)
            noise999 = torch.stack([add_diffusion_noise(im, 999) for im in imgs])
            emb_n = front.assemble(front.embeds_to_llm(front.image_embeds(noise999), qf_ids), llm_ids, embed)
            ie_zero = front.image_embeds(torch.zeros_like(uniq[:1])).expand(len(qs), -1, -1).contiguous()
            emb_z = front.assemble(front.embeds_to_llm(ie_zero, qf_ids), llm_ids, embed)
            prior_kw = {k: v for k, v in base_kw.items() if k not in ("cd_beta", "cd_alpha")}
            noise = engine.generate(None, inputs_embeds=emb_n, max_length=1, **prior_kw)
            zeros = engine.generate(None, inputs_embeds=emb_z, max_length=1, **prior_kw)
            rows.add(idx, map_pad_to_eos(main.tokens), [(o.top_tok, o.top_prob) for o in (main, noise, zeros)])
    got = rows.gather(plan, len(questions))                    # ONE collective; every rank holds every question's results behind it
    dicts = [[C.label_dict_from_top(t, p_, decode_token) for t, p_ in got["tops"][s_]] for s_ in range(3)]
    ordered = [{"question_id": q["question_id"], "prompt": q["text"] + QUESTION_SUFFIX, "text": decode(cut_at_eos(got["tokens"][i], eos_set)).strip(),
                "model_id": model_id, "image": q["image"], "naive": dicts[0][i], "noise": dicts[1][i], "zeros": dicts[2][i], "metadata": {}}
               for i, q in enumerate(questions)]
    if plan.rank != 0:
        answers_path = None                                    # rank 0 owns the file
    if answers_path is not None:
        with open(answers_path, "w") as f:
            for a in ordered:
                f.write(json.dumps(a) + "\n")
    scores = {}
    if all("label" in q for q in questions):
        gt = [{"question_id": q["question_id"], "label": q["label"]} for q in questions]
        for name, fn, args in (("string_match", C.pope_scores, ()), ("naive", C.pope_scores_calibrated, ("naive",)),
                               ("noise", C.pope_scores_calibrated, ("noise",)), ("zeros", C.pope_scores_calibrated, ("zeros",))):
            try:
                scores[name] = fn(gt, ordered, *args)
            except ZeroDivisionError:
                scores[name] = None
    return {"answers": ordered, "scores": scores, "batch_invariant": invariant}
