"""Qwen-VL POPE / open-ended drivers over the native engine: the batched replacement of the reference's per-question loops
experiments/eval/calibrate/qwen_calibrate.py:90-168 (POPE answers + four content-free priors) and
experiments/eval/sampling/qwenvl_sampling.py:48-115 (answers only).

Per question qwen_calibrate.py runs FIVE `model.generate(input_ids, attention_mask, images=..., do_sample=True, max_new_tokens=20,
min_new_tokens=1, pad = eos = tokenizer.eod_id, temperature, top_p, top_k, ...)` calls at B = 1:
  main    '<img>{path}</img>{q} Answer:' with the image tensor, images_cd = add_diffusion_noise(image, noise_step) under --use_cd,
          use_dd / use_dd_unk, cd_alpha, cd_beta (:113-136)                                  -> `text`, `naive` (step-0 top-10 label dict)
  none    '{q} Answer:', plain sampling (:36-37, :142)                                           -> `none`
  unk     'None {q} Answer:', plain (:38-39, :143)                                               -> `unk`
  noise   the main prompt with images = add_diffusion_noise(image, 999), plain (:145-146)        -> `noise`
  zero    the main prompt with images = zeros_like(image), plain (:148-149)                      -> `zero`
Only the step-0 scores of the four prior passes are used (:66-68), so they decode ONE token here.  Each pass is one engine call
over a whole batch of questions.

The Qwen ViT + resampler are outside the north-star path: `embed_prompt(text, image) -> [T, d]` is the
caller's front-end - token embeddings with the 256 slots between <img> and </img> filled from `image` (a [3, S, S] tensor: what the
reference passes as `images=`, modeling_qwen.py:565-566) or, for prompts without an image span, image = None.  It may return
`(embeddings, n_shared)`, n_shared = the leading rows every prompt about the same image tensor starts with ('<img>' + the slots).
"""
from __future__ import annotations

import json
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import calibrate as C
from .engine import VddLlavaEngine

POPE_PROMPT = "<img>{}</img>{} Answer:"                       # qwen_calibrate.py:41, :97
SAMPLING_PROMPT = "Question: <img>{}</img> {} Answer:"        # qwenvl_sampling.py:54
PRIORS = ("none", "unk", "noise", "zero")                     # the order of qwen_calibrate.py:142-149 (file keys :160-164)
CALIBRATE_NAMES = ("naive", "none", "unk", "none_unk", "noise", "zero", "none_noise", "none_unk_noise", "all")


def _unwrap(e):
    return (e[0], int(e[1])) if isinstance(e, tuple) else (e, None)


def _embeds(embed_prompt, texts, images, keys):
    """-> (list of [T, d], embeds_prefix list or None).  keys[i]: what identifies images[i] (rows with one key share the image span's K/V)."""
    out, pre = [], []
    for t, im, k in zip(texts, images, keys):
        e, n = _unwrap(embed_prompt(t, im))
        out.append(e)
        pre.append((k, n) if (n is not None and k is not None) else None)
    return out, (pre if all(p_ is not None for p_ in pre) else None)


def run_qwen_pope(engine: VddLlavaEngine, questions: Sequence[dict], embed_prompt: Callable[[str, Optional[torch.Tensor]], torch.Tensor],
                  decode: Callable[[List[int]], str], load_image: Callable[[str], torch.Tensor], image_path: Callable[[str], str] = lambda f: f,
                  answers_path: Optional[str] = None, model_id: str = "qwen-vl", batch_questions: int = 128, eos_token_id=151643,
                  pad_token_id: Optional[int] = 151643, max_new_tokens: int = 20, min_new_tokens: Optional[int] = 1, use_cd: bool = False,
                  noise_step: int = 500, priors: Sequence[str] = PRIORS, prompt_format: str = POPE_PROMPT, rank: Optional[int] = None,
                  world: Optional[int] = None, batch_invariant: Optional[bool] = None, sweep: Optional[Sequence[dict]] = None, **generate_kw) -> dict:
    """questions: POPE json lines (question_id, image, text[, label]).  load_image(name) -> the [3, S, S] tensor the model's own
    `visual.image_transform` makes of the file (qwen_calibrate.py:100-101).  generate_kw: temperature, top_p, top_k, use_dd, use_dd_unk,
    cd_alpha, cd_beta, seed, cd_greedy ... - the reference's generate kwargs (:113-136; defaults there: temperature 0.2, cd_alpha 1,
    cd_beta 0.1).  priors: which content-free passes to run (all four as the reference; () = answers only, the call shape of
    qwenvl_sampling.py with prompt_format = SAMPLING_PROMPT).
    rank / world (default: the initialised torch.distributed group): every rank decodes its chunk of whole images (shard.ShardPlan), ONE
    collective gathers the results, rank 0 writes the file; every rank returns the full result.  batch_invariant: as in
    pope_driver.run_pope.  The JSONL carries the reference's fields (:155-166): question_id, prompt, text, naive, noise, none, zero, unk,
    model_id, image, metadata.  Returns {"answers": [...], "scores": {...}}.
    sweep (answers-only runs, priors = (): qwenvl_sampling.py:147-185 walks the question file once per sampling setting): a list of
    {"tag", "temperature", "top_p", "top_k"[, "answers_path"]} decodes every setting from ONE prefill per batch (engine.generate(reuse_prefill=True);
    not with --use_cd, whose noise is drawn per call, and not for open-ended lists, which go through generate_list per setting).
    Returns {"runs": {tag: result}}."""
    import contextlib
    from . import ops
    from .pope_driver import ResultRows, cut_at_eos
    from .shard import ShardPlan, resolve_batch_invariant
    from .vcd_add_noise import add_diffusion_noise
    priors = tuple(priors)
    if any(p_ not in PRIORS for p_ in priors):
        raise ValueError(f"priors must be among {PRIORS}")
    order = sorted(range(len(questions)), key=lambda i: (questions[i]["image"], i))
    plan = ShardPlan([questions[i]["image"] for i in order], rank, world)
    mine = [order[p_] for p_ in plan.mine]
    if generate_kw.get("seed") is not None:
        generate_kw = dict(generate_kw, seed=int(generate_kw["seed"]) + plan.rank)
    decode_token = lambda t: decode([t])
    eos_set = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
    rows = ResultRows(engine.device, max_new_tokens, pad_token_id if pad_token_id is not None else 0, n_sets=1 + len(priors))
    plain_kw = {k: v for k, v in generate_kw.items() if k in ("temperature", "top_p", "top_k", "seed", "cd_alpha", "cd_beta")}
    prior_kw = dict(max_new_tokens=1, n_top=10, min_new_tokens=min_new_tokens, eos_token_id=eos_token_id, pad_token_id=pad_token_id, **plain_kw)
    invariant = resolve_batch_invariant(batch_invariant, plan.world, generate_kw)
    prompts = {}
    if sweep is not None and priors:
        raise ValueError("sweep: answers-only runs (priors=())")
    settings = [dict(tag=None)] if sweep is None else [dict(s_) for s_ in sweep]
    over = lambda s_: {k: s_[k] for k in ("temperature", "top_p", "top_k") if k in s_}
    extra_rows = [ResultRows(engine.device, max_new_tokens, pad_token_id if pad_token_id is not None else 0, n_sets=1) for _ in settings[1:]]
    # answers only, no EOS floor (qwenvl_sampling.py:89-103: open-ended answers, max_new_tokens 1024): the whole shard as ONE list, batch_questions
    # in flight, waiting questions admitted into the slots of finished ones (engine.generate_list)
    list_kw = ("temperature", "top_p", "top_k", "use_dd", "use_dd_unk", "cd_alpha", "cd_beta", "seed", "cd_greedy", "sync_every", "admit_min")
    as_list = (not priors and not min_new_tokens and eos_token_id is not None and pad_token_id is not None and bool(mine)
               and engine.cfg.lm.head_dim == 128 and all(k in list_kw for k in generate_kw))
    if sweep is not None and as_list:
        raise ValueError("sweep: open-ended lists (no EOS floor) decode one setting per call")
    with (ops.batch_invariant() if invariant else contextlib.nullcontext()):
        if as_list:
            cache: Dict[str, torch.Tensor] = {}
            for i in mine:
                if questions[i]["image"] not in cache:
                    cache[questions[i]["image"]] = load_image(questions[i]["image"]).to(engine.device)
            imgs = [cache[questions[i]["image"]] for i in mine]
            texts = [prompt_format.format(image_path(questions[i]["image"]), questions[i]["text"]) for i in mine]
            prompts.update(zip(mine, texts))
            emb, pre = _embeds(embed_prompt, texts, imgs, [("clean", questions[i]["image"]) for i in mine])
            kw = dict(generate_kw)
            if use_cd:
                kw["images_cd"] = _embeds(embed_prompt, texts, [add_diffusion_noise(im, noise_step) for im in imgs], [None] * len(mine))[0]
            out = engine.generate_list(None, inputs_embeds=emb, embeds_prefix=pre, in_flight=batch_questions, max_new_tokens=max_new_tokens, n_top=10,
                                       eos_token_id=eos_token_id, pad_token_id=pad_token_id, **kw)
            rows.add(mine, out.tokens, [(out.top_tok, out.top_prob)])
            list_stats = out.stats
        for b0 in range(0, 0 if as_list else len(mine), batch_questions):
            idx = mine[b0:b0 + batch_questions]
            qs = [questions[i] for i in idx]
            cache: Dict[str, torch.Tensor] = {}
            for q in qs:
                if q["image"] not in cache:
                    cache[q["image"]] = load_image(q["image"]).to(engine.device)
            imgs = [cache[q["image"]] for q in qs]
            texts = [prompt_format.format(image_path(q["image"]), q["text"]) for q in qs]
            for i, t in zip(idx, texts):
                prompts[i] = t

            built = {}

            def gen(texts_, images_, keys_, **kw):
                k_ = (id(texts_), id(images_))                  # the settings of a sweep pass the same lists: the same embedding tensors
                if k_ not in built:
                    built[k_] = (texts_, images_, _embeds(embed_prompt, texts_, images_, keys_))
                emb, pre = built[k_][2]
                if pre is not None:
                    kw["embeds_prefix"] = pre
                return engine.generate(None, inputs_embeds=emb, **kw)
            kw = dict(generate_kw)
            if use_cd:            # fresh noise per QUESTION, as the reference draws it inside its loop (:103-106); no sharing between them
                cd, _ = _embeds(embed_prompt, texts, [add_diffusion_noise(im, noise_step) for im in imgs], [None] * len(qs))
                kw["images_cd"] = cd
            main = gen(texts, imgs, [("clean", q["image"]) for q in qs], max_new_tokens=max_new_tokens, min_new_tokens=min_new_tokens, n_top=10,
                       eos_token_id=eos_token_id, pad_token_id=pad_token_id, reuse_prefill=len(settings) > 1 and not use_cd, **dict(kw, **over(settings[0])))
            tops = [(main.top_tok, main.top_prob)]
            for s_, r_ in zip(settings[1:], extra_rows):       # the other settings of a sweep: the same prompts, decoded again from the kept prefill
                o_ = gen(texts, imgs, [("clean", q["image"]) for q in qs], max_new_tokens=max_new_tokens, min_new_tokens=min_new_tokens, n_top=10,
                         eos_token_id=eos_token_id, pad_token_id=pad_token_id, reuse_prefill=not use_cd, **dict(kw, **over(s_)))
                r_.add(idx, o_.tokens, [(o_.top_tok, o_.top_prob)])
            def text_only(fmt):
                """POPE repeats its question texts over the images, and a text-only prompt's step-0 label dict depends on the text alone: every
                distinct prompt runs once"""
                ts = [fmt.format(q["text"]) for q in qs]
                first = {}
                where = [first.setdefault(t, len(first)) for t in ts]
                o_ = gen(list(first), [None] * len(first), [None] * len(first), **prior_kw)
                back = torch.tensor(where, dtype=torch.long).to(o_.top_tok.device, non_blocking=True)
                return type("Top", (), {"top_tok": o_.top_tok[back], "top_prob": o_.top_prob[back]})
            for name in priors:
                if name == "none":
                    o = text_only("{} Answer:")
                elif name == "unk":
                    o = text_only("None {} Answer:")
                elif name == "noise":
                    o = gen(texts, [add_diffusion_noise(im, 999) for im in imgs], [None] * len(qs), **prior_kw)
                else:             # one zero image per tensor shape: the span's rows replace the path bytes, so EVERY question shares them
                    zero = {}
                    zs = [zero.setdefault(tuple(im.shape), torch.zeros_like(im)) for im in imgs]
                    o = gen(texts, zs, [("zero", tuple(im.shape)) for im in imgs], **prior_kw)
                tops.append((o.top_tok, o.top_prob))
            rows.add(idx, main.tokens, tops)
            engine.clear_image_cache()
    def finish(rows, answers_path):
        got = rows.gather(plan, len(questions))                    # ONE collective; every rank holds every question's results behind it
        dicts = [[C.label_dict_from_top(t, p_, decode_token) for t, p_ in got["tops"][s_]] for s_ in range(1 + len(priors))]
        answers = []
        for i, q in enumerate(questions):
            a = {"question_id": q["question_id"], "prompt": prompts.get(i, prompt_format.format(image_path(q["image"]), q["text"])),
                 "text": decode(cut_at_eos(got["tokens"][i], eos_set)).strip()}
            if priors:
                a["naive"] = dicts[0][i]
                for name in ("noise", "none", "zero", "unk"):      # the file's key order (:160-164)
                    if name in priors:
                        a[name] = dicts[1 + priors.index(name)][i]
            a.update(model_id=model_id, image=q["image"], metadata={})
            answers.append(a)
        if answers_path is not None and plan.rank == 0:
            import os
            os.makedirs(os.path.dirname(os.path.abspath(answers_path)), exist_ok=True)
            with open(answers_path, "w") as f:
                for a in answers:
                    f.write(json.dumps(a) + "\n")
        scores = {}
        if all("label" in q for q in questions):
            gt = [{"question_id": q["question_id"], "label": q["label"]} for q in questions]
            for name in ("string_match",) + (CALIBRATE_NAMES if priors else ()):
                if name != "string_match" and not set(C.calibrate_sources(name)) <= set(("naive",) + priors):
                    continue
                try:
                    scores[name] = C.pope_scores(gt, answers) if name == "string_match" else C.pope_scores_calibrated(gt, answers, name)
                except ZeroDivisionError:
                    scores[name] = None
        return {"answers": answers, "scores": scores, "rank": plan.rank, "world": plan.world, "batch_invariant": invariant,
                "stats": list_stats if as_list else {}}
    if sweep is None:
        return finish(rows, answers_path)
    return {"runs": {s_["tag"]: finish(r_, s_.get("answers_path")) for s_, r_ in zip(settings, [rows] + extra_rows)}, "batch_invariant": invariant}


def main(argv=None):
    """python -m llava_align_amd.qwen_driver --model-path DIR --question-file Q.json --image-folder IMGS --answers-file OUT.jsonl
    [--use_dd --use_dd_unk --use_cd --noise_step 500 --cd_alpha 1 --cd_beta 0.1 --temperature 0.2 --top_p P --top_k K --seed 42]
    [--sampling [--no-sweep]]: the arguments of experiments/eval/calibrate/qwen_calibrate.py:170-195; with --sampling the answers-only runs of
    experiments/eval/sampling/qwenvl_sampling.py:117-185 ('setting' in the answers file name replaced by default / temp_T / top_p_P / top_k_K;
    max_new_tokens 20 for POPE question files, 1024 otherwise).  The Qwen-VL directory is loaded through transformers (trust_remote_code: the
    caller's environment must provide what modeling_qwen.py imports); its ViT + resampler fill the image slots, the language model runs
    natively.  Under torchrun: one rank per GPU, whole images per rank, one gather."""
    import argparse
    import os
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", default="Qwen/Qwen-VL-Chat")
    ap.add_argument("--model-base", default=None)
    ap.add_argument("--image-folder", default="")
    ap.add_argument("--question-file", required=True)
    ap.add_argument("--answers-file", required=True)
    ap.add_argument("--conv-mode", default="llava_v1")
    ap.add_argument("--num-chunks", type=int, default=1)
    ap.add_argument("--chunk-idx", type=int, default=0)
    ap.add_argument("--temperature", type=float, default=None, help="default 0.2 (qwen_calibrate.py:182); --sampling runs at 1.0 as qwenvl_sampling.py:147 sets it")
    ap.add_argument("--top_p", type=float, default=None)
    ap.add_argument("--top_k", type=int, default=None)
    ap.add_argument("--noise_step", type=int, default=500)
    ap.add_argument("--use_cd", action="store_true")
    ap.add_argument("--use_dd", action="store_true")
    ap.add_argument("--use_dd_unk", action="store_true")
    ap.add_argument("--cd_alpha", type=float, default=1.0)
    ap.add_argument("--cd_beta", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--sampling", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--dtype", choices=("float16", "bfloat16"), default="bfloat16")
    a = ap.parse_args(argv)
    from . import checkpoint as K
    from .shard import get_chunk, init_from_env
    rank, world, device = init_from_env()
    eng, tok, model, embed_prompt = K.load_qwen(a.model_path, device, dtype=getattr(torch, a.dtype))
    questions = [json.loads(q) for q in open(os.path.expanduser(a.question_file))]
    if a.num_chunks > 1:
        questions = [questions[i] for i in get_chunk(len(questions), a.num_chunks, a.chunk_idx, group=1)]

    def load_image(name):                                      # qwen_calibrate.py:100-101
        from PIL import Image
        return model.transformer.visual.image_transform(Image.open(os.path.join(a.image_folder, name)).convert("RGB"))
    runs = [(None, a.temperature if a.temperature is not None else 0.2, a.top_p, a.top_k)]
    if a.sampling:                                             # qwenvl_sampling.py:147-185: temperature 1.0, no top-p / top-k whatever the CLI says
        runs = [("default", 1.0, None, None)]
        if not (a.no_sweep or a.use_cd):
            runs += [(f"temp_{t}", float(t), None, None) for t in np.round(np.arange(0.05, 1.05, 0.05), 2)]
            runs += [(f"top_p_{p_}", 1.0, float(p_), None) for p_ in np.round(np.arange(0, 1.05, 0.05), 2)]
            runs += [(f"top_k_{k}", 1.0, None, k) for k in (1, 2, 5, 10, 20, 50, 100, 200, 500)]
    pope = "POPE" in a.question_file
    common = dict(image_path=lambda f: os.path.join(a.image_folder, f), model_id="qwen-vl" if "Chat" not in a.model_path else "qwen-vl-chat",
                  batch_questions=a.batch, eos_token_id=tok.eod_id, pad_token_id=tok.eod_id, use_cd=a.use_cd, noise_step=a.noise_step, rank=rank, world=world,
                  use_dd=a.use_dd, use_dd_unk=a.use_dd_unk, cd_alpha=a.cd_alpha, cd_beta=a.cd_beta, seed=a.seed)
    if a.sampling and pope and len(runs) > 1:
        # the POPE sampling sweep (answers of <= 20 tokens): every setting from ONE pass over the question file, a batch's prefill shared by its settings
        sweep = [dict(tag=tag, temperature=temp, top_p=top_p, top_k=top_k, answers_path=os.path.expanduser(a.answers_file).replace("setting", tag))
                 for tag, temp, top_p, top_k in runs]
        res = run_qwen_pope(eng, questions, embed_prompt, lambda ids: tok.decode(ids, skip_special_tokens=True), load_image, priors=(), prompt_format=SAMPLING_PROMPT,
                            max_new_tokens=20, min_new_tokens=1, sweep=sweep, **common)
        if rank == 0:
            for s_ in sweep:
                print(json.dumps({"run": s_["tag"], "answers_file": s_["answers_path"], "n_answers": len(res["runs"][s_["tag"]]["answers"]),
                                  "scores": res["runs"][s_["tag"]]["scores"]}), flush=True)
        runs = []
    for tag, temp, top_p, top_k in runs:
        path = os.path.expanduser(a.answers_file)
        path = path.replace("setting", tag) if tag else path
        kw = dict(max_new_tokens=20, min_new_tokens=1) if (pope or not a.sampling) else dict(max_new_tokens=1024, min_new_tokens=None)
        res = run_qwen_pope(eng, questions, embed_prompt, lambda ids: tok.decode(ids, skip_special_tokens=True), load_image,
                            image_path=lambda f: os.path.join(a.image_folder, f), answers_path=path,
                            model_id="qwen-vl" if "Chat" not in a.model_path else "qwen-vl-chat", batch_questions=a.batch, eos_token_id=tok.eod_id,
                            pad_token_id=tok.eod_id, use_cd=a.use_cd, noise_step=a.noise_step, priors=() if a.sampling else PRIORS,
                            prompt_format=SAMPLING_PROMPT if a.sampling else POPE_PROMPT, rank=rank, world=world, use_dd=a.use_dd,
                            use_dd_unk=a.use_dd_unk, cd_alpha=a.cd_alpha, cd_beta=a.cd_beta, temperature=temp, top_p=top_p, top_k=top_k, seed=a.seed, **kw)
        if rank == 0:
            nan = {k: v["nan_rows"] for k, v in res["scores"].items() if isinstance(v, dict) and v.get("nan_rows")}
            print(json.dumps({"run": tag or "calibrate", "answers_file": path, "n_answers": len(res["answers"]), "scores": res["scores"],
                              "rows_whose_calibrated_vector_is_nan": nan}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
