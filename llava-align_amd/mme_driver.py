"""MME evaluation driver over the native engine (BASELINE config #4): the batched replacement of the reference's per-question loops
experiments/eval/MME/run_qwen.py:143-238 (Qwen-VL) and experiments/eval/MME/run_llava.py:130-237 (LLaVA), followed by the
calibrate converter (convert_answer_to_mme_calibrate.py -> calibrate.mme_convert) and the MME scorer (eval_tool/calculation.py ->
calibrate.mme_scores).

Per question the reference runs THREE generate() calls at B = 1:
  main   image + question with the VDD / VCD kwargs                                   -> `text`, `naive` (step-0 top-10 label dict)
  none   the question as TEXT only, plain sampling                                    -> `none`   (run_qwen.py:100-137, run_llava.py:95-128)
  unk    the image replaced by a placeholder ('None ' text / the <unk> token), plain  -> `unk`
Only the step-0 scores of `none` / `unk` are ever used (:135-137), so here they decode ONE token.  Each of the three is one engine
call over a whole batch of questions.

Model-specific prompt construction stays with the caller through `build_inputs(line, kind)`, kind in {"main", "none", "unk"}, which
returns either {"input_ids": 1-D ids with one -200 image slot, "image": [3,S,S] tensor or None} (LLaVA: llava_mme_inputs) or
{"inputs_embeds": [T, d] embeddings} (Qwen-VL, whose ViT + resampler are outside the north-star path and fill the 256 image slots
upstream: qwen_mme_inputs).
"""
from __future__ import annotations

import uuid
from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import calibrate as C
from .engine import IMAGE_TOKEN_INDEX, VddLlavaEngine

MME_SUBSETS = ("existence", "count", "position", "color", "commonsense_reasoning", "numerical_calculation", "text_translation",
               "code_reasoning")                                                   # run_qwen.py:146, run_llava.py:133
ONE_WORD = " Please answer this question with one word."                          # run_llava.py:106


def vicuna_v1_prompt(user_text: str) -> str:
    """conv_templates['vicuna_v1'] with one user turn (experiments/llava/conversation.py:252-262, SeparatorStyle.TWO)."""
    system = ("A chat between a curious user and an artificial intelligence assistant. "
              "The assistant gives helpful, detailed, and polite answers to the user's questions.")
    return f"{system} USER: {user_text} ASSISTANT:"


def llava_mme_inputs(encode: Callable[[str], List[int]], load_image: Callable[[str], torch.Tensor], unk_token_id: int = 0):
    """run_llava.py: main = '<image>\\n' + question in the vicuna_v1 template (:52-62, no one-word suffix); none = question + suffix
    without an image token (:101-109 with images=None); unk = '<image>\\n' + question + suffix with the slot replaced by the
    tokenizer's <unk> (:102-115).  `encode(prompt)` tokenises with -200 where '<image>' stands (tokenizer_image_token)."""
    cache: Dict[str, torch.Tensor] = {}         # MME asks two questions per image, adjacent after the driver's sort: decode each image once

    def image(name):
        if name not in cache:
            if len(cache) >= 64:
                cache.pop(next(iter(cache)))
            cache[name] = load_image(name)
        return cache[name]

    def build(line, kind):
        q = line["text"]
        if kind == "main":
            return {"input_ids": torch.tensor(encode(vicuna_v1_prompt("<image>\n" + q))), "image": image(line["image"])}
        if kind == "none":
            return {"input_ids": torch.tensor(encode(vicuna_v1_prompt(q + ONE_WORD))), "image": None}
        ids = encode(vicuna_v1_prompt("<image>\n" + q + ONE_WORD))
        return {"input_ids": torch.tensor([unk_token_id if t == IMAGE_TOKEN_INDEX else t for t in ids]), "image": None}
    return build


def qwen_mme_inputs(embed_prompt: Callable[[str, Optional[str]], torch.Tensor], image_path: Callable[[str], str] = lambda f: f):
    """run_qwen.py: main = '<img>{path}</img>{q} Answer:' (:176-177), none = '{q} Answer:' (:101-102), unk = 'None {q} Answer:'
    (:103-104).  `embed_prompt(text, image_path_or_None) -> [T, d]` is the caller's Qwen front-end: token embeddings with the 256
    image slots between <img> and </img> filled by its ViT + resampler; it may return `(embeddings, n_shared)` instead, n_shared =
    the leading rows every prompt about that image starts with ('<img>' + the image slots): the engine then prefills them once per
    image (`embeds_prefix`)."""
    def build(line, kind):
        q = line["text"]
        if kind == "main":
            p = image_path(line["image"])
            e = embed_prompt("<img>{}</img>{} Answer:".format(p, q), p)
            if isinstance(e, tuple):
                return {"inputs_embeds": e[0], "embeds_prefix": (p, int(e[1]))}
            return {"inputs_embeds": e}
        if kind == "none":
            return {"inputs_embeds": embed_prompt("{} Answer:".format(q), None)}
        return {"inputs_embeds": embed_prompt("{} {} Answer:".format("None", q), None)}
    return build


def _generate(engine, batch, **kw):
    if "inputs_embeds" in batch[0]:
        unwrap = lambda e: e[0] if isinstance(e, tuple) else e
        if all("embeds_prefix" in b for b in batch):
            kw = dict(kw, embeds_prefix=[b["embeds_prefix"] for b in batch])
        return engine.generate(None, inputs_embeds=[unwrap(b["inputs_embeds"]) for b in batch], **kw)
    imgs = [b["image"] for b in batch]
    has_img = imgs[0] is not None
    return engine.generate([b["input_ids"] for b in batch], images=imgs if has_img else None, **kw)


def run_mme(engine: VddLlavaEngine, questions: Sequence[dict], build_inputs: Callable[[dict, str], dict],
            decode: Callable[[List[int]], str], answers_path: Optional[str] = None, model_id: str = "llava-align_amd",
            batch_questions: int = 256, eos_token_id=None, pad_token_id: Optional[int] = None, stop_str: Optional[str] = None,
            max_new_tokens: int = 20, min_new_tokens: Optional[int] = None, noise_step: Optional[int] = None,
            gt: Optional[Dict[tuple, str]] = None, results_root: Optional[str] = None, experiment: str = "exp",
            subsets: Optional[Sequence[str]] = MME_SUBSETS, chunk: Optional[tuple] = None, rank: Optional[int] = None,
            world: Optional[int] = None, batch_invariant: Optional[bool] = None, sweep: Optional[Sequence[dict]] = None, **generate_kw) -> dict:
    """questions: the llava_mme.jsonl lines (question_id 'category/image.ext', image, text, category); subsets filters them as the
    reference does; chunk=(n, k) takes the reference's k-th of n contiguous ceil-chunks (get_chunk, run_llava.py:32-40).
    generate_kw: cd_alpha, cd_beta, use_dd, use_dd_unk, temperature, top_p, top_k, seed - the reference's generate kwargs; the Qwen
    call shape adds min_new_tokens=1 and pad = eos = eod id (run_qwen.py:190-213); noise_step adds the VCD branch for LLaVA inputs.
    gt + results_root: also convert ('naive', 'none', 'unk', 'none_unk') and score.
    rank / world (default: the initialised torch.distributed group): the data-parallel form of the reference's --num-chunks /
    --chunk-idx processes (run_llava.py:261-262) - every rank calls this with the same list, decodes its chunk of whole images
    (shard.ShardPlan), ONE collective gathers the per-question results, rank 0 writes the answers / result files; every rank returns
    the full result.  batch_invariant: as in pope_driver.run_pope (default: on for cd_greedy / top_k = 1 / do_sample = False decodes).
    sweep: the reference's scripts run the question file once per sampling setting (51 of them, run_llava.py:281-318).  A list of
    {"tag", "temperature", "top_p", "top_k"[, "answers_path", "experiment"]} runs them ALL in one pass over the batches: the settings of a batch
    share its vision-tower pass and its prefills (engine.generate(reuse_prefill=True): only the warpers differ, and they do not enter the
    prefill), each decodes from that state with its own warpers.  Returns {"runs": {tag: the per-setting result}}; not with the VCD branch
    (its noise is drawn per call), which decodes every setting from scratch.
    Returns {"answers": [...], "results": {name: dir}, "scores": {name: mme_scores}}."""
    import contextlib
    from . import ops
    from .shard import get_chunk, resolve_batch_invariant
    qs_all = [q for q in questions if subsets is None or q.get("category") in subsets]
    if chunk is not None:
        qs_all = [qs_all[i] for i in get_chunk(len(qs_all), chunk[0], chunk[1], group=1)]
    from .pope_driver import ResultRows, cut_at_eos
    from .shard import ShardPlan
    order = sorted(range(len(qs_all)), key=lambda i: (qs_all[i]["image"], i))          # an image's two questions adjacent: shared features
    plan = ShardPlan([qs_all[i]["image"] for i in order], rank, world)
    mine = [order[p_] for p_ in plan.mine]
    decode_token = lambda t: decode([t])
    eos_set = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
    if generate_kw.get("seed") is not None:
        generate_kw = dict(generate_kw, seed=int(generate_kw["seed"]) + plan.rank)
    settings = [dict(tag=None)] if sweep is None else [dict(s_) for s_ in sweep]
    if not settings:
        raise ValueError("sweep: at least one setting")
    over = lambda s_: {k: s_[k] for k in ("temperature", "top_p", "top_k") if k in s_}
    kws = [dict(generate_kw, **over(s_)) for s_ in settings]
    keep = len(settings) > 1 and noise_step is None            # the settings of a batch decode from ONE prefill
    all_rows = [ResultRows(engine.device, max_new_tokens, pad_token_id if pad_token_id is not None else 0, n_sets=3) for _ in settings]
    invariant = resolve_batch_invariant(batch_invariant, plan.world, generate_kw)
    with (ops.batch_invariant() if invariant else contextlib.nullcontext()):
        for b0 in range(0, len(mine), batch_questions):
            idx = mine[b0:b0 + batch_questions]
            lines = [qs_all[i] for i in idx]
            img_cache: Dict[str, dict] = {}

            def main_inputs(line):
                # one tensor object per distinct image, so that the engine shares its features and prompt-prefix KV
                b = build_inputs(line, "main")
                if b.get("image") is not None:
                    b["image"] = img_cache.setdefault(line["image"], b)["image"]
                return b
            mains = [main_inputs(l) for l in lines]
            nones, unks = [build_inputs(l, "none") for l in lines], [build_inputs(l, "unk") for l in lines]
            outs = [[None, None, None] for _ in settings]
            # pass by pass, every setting behind the other: the kept prefill lives in the engine's pools, which the next pass type overwrites
            for j, kw_s in enumerate(kws):
                kw = dict(kw_s)
                if noise_step is not None and "image" in mains[0]:
                    from .vcd_add_noise import add_diffusion_noise
                    kw["images_cd"] = [add_diffusion_noise(img_cache[l["image"]]["image"].to(engine.device), noise_step) for l in lines]   # run_llava.py:187-190
                outs[j][0] = _generate(engine, mains, max_new_tokens=max_new_tokens, n_top=10, eos_token_id=eos_token_id, pad_token_id=pad_token_id,
                                       min_new_tokens=min_new_tokens, reuse_prefill=keep, **kw)
            # content-free priors: plain sampling, only the step-0 distribution is used; min_new_tokens = 1 keeps EOS out of it as in
            # the reference's calibration calls (run_qwen.py:111-131)
            for which, inputs in ((1, nones), (2, unks)):
                for j, kw_s in enumerate(kws):
                    prior_kw = dict(max_new_tokens=1, n_top=10, **{k: v for k, v in kw_s.items() if k in ("temperature", "top_p", "top_k", "seed", "cd_alpha", "cd_beta")})
                    if min_new_tokens:
                        prior_kw.update(min_new_tokens=min_new_tokens, eos_token_id=eos_token_id, pad_token_id=pad_token_id)
                    outs[j][which] = _generate(engine, inputs, reuse_prefill=keep, **prior_kw)
            for j in range(len(settings)):
                all_rows[j].add(idx, outs[j][0].tokens, [(o.top_tok, o.top_prob) for o in outs[j]])
            engine.clear_image_cache()

    def finish(rows, answers_path, experiment):
        got = rows.gather(plan, len(qs_all))                   # ONE collective per setting; every rank holds every question's results behind it
        dicts = [[C.label_dict_from_top(t, p_, decode_token) for t, p_ in got["tops"][s_]] for s_ in range(3)]
        answers = []
        for i, line in enumerate(qs_all):
            text = decode(cut_at_eos(got["tokens"][i], eos_set)).strip()
            if stop_str and text.endswith(stop_str):
                text = text[:-len(stop_str)]
            answers.append({"question_id": line["question_id"], "prompt": line["text"], "text": text.strip(), "naive": dicts[0][i],
                            "none": dicts[1][i], "unk": dicts[2][i], "answer_id": uuid.uuid4().hex[:22], "model_id": model_id, "metadata": {}})
        root = results_root
        if plan.rank != 0:
            answers_path = root = None                         # rank 0 owns the files
        if answers_path is not None:
            import json
            import os
            os.makedirs(os.path.dirname(os.path.abspath(answers_path)), exist_ok=True)
            with open(answers_path, "w") as f:
                for a in answers:
                    f.write(json.dumps(a) + "\n")
        out = {"answers": answers, "results": {}, "scores": {}, "batch_invariant": invariant}
        if gt is not None:
            conv = C.mme_convert(answers, gt)
            out["converted"] = conv
            if root is not None:
                out["results"] = C.write_mme_results(conv, root, experiment)
                for name, d in out["results"].items():
                    try:
                        out["scores"][name] = C.mme_scores(d)
                    except (FileNotFoundError, AssertionError):    # a chunk / subset without all 8 task files or with odd line counts
                        out["scores"][name] = None
        return out
    if sweep is None:
        return finish(all_rows[0], answers_path, experiment)
    return {"runs": {s_["tag"]: finish(r_, s_.get("answers_path"), s_.get("experiment", f"{experiment}-{s_['tag']}")) for s_, r_ in zip(settings, all_rows)},
            "batch_invariant": invariant}


def _sweep_settings(args) -> list:
    """The runs of one invocation of the reference's scripts (run_llava.py:281-318, run_qwen.py:268-304): 'default' first (run_llava: at
    temperature 1.0; run_qwen: at --temperature), then - only with --use_dd / --use_dd_unk - temperature 0.05 ... 1.0, top_p 0 ... 1.0 and
    top_k in 1 ... 500, each into `answers-file` with 'setting' replaced.  -> [(tag, temperature, top_p, top_k)]."""
    import numpy as np
    base_t = 1.0 if args.arch == "llava" else args.temperature
    runs = [("default", base_t, args.top_p, args.top_k)]
    if args.no_sweep or not (args.use_dd or args.use_dd_unk):
        return runs
    runs += [(f"temp_{t}", float(t), args.top_p, args.top_k) for t in np.round(np.arange(0.05, 1.05, 0.05), 2)]
    runs += [(f"top_p_{p_}", args.temperature, float(p_), args.top_k) for p_ in np.round(np.arange(0, 1.05, 0.05), 2)]
    runs += [(f"top_k_{k}", args.temperature, args.top_p, k) for k in (1, 2, 5, 10, 20, 50, 100, 200, 500)]
    return runs


def main(argv=None):
    """python -m llava_align_amd.mme_driver --arch llava|qwen --model-path DIR --image-folder MME_Benchmark --question-file llava_mme.jsonl
    --answers-file OUT-setting.jsonl [--use_dd --use_dd_unk --use_cd --noise_step 500 --cd_alpha 1 --cd_beta 0.1 --temperature T --top_p P
    --top_k K --max_new_tokens N --num-chunks n --chunk-idx k --gt-root MME_Benchmark --no-sweep]: the arguments of the reference's
    experiments/eval/MME/run_llava.py:253-318 and run_qwen.py:240-304 (and their sweep over temperature / top_p / top_k when a VDD flag
    is on) over the native engine; with --gt-root also the calibrate converter + MME scorer that the reference runs as separate scripts.
    --arch llava: a LLaVA-1.5 checkpoint directory (checkpoint.load_llava).  --arch qwen: a Qwen-VL directory loaded through transformers
    (trust_remote_code, the caller's environment must provide what modeling_qwen.py imports) - its ViT + resampler fill the image slots,
    the language model runs natively (hf_adapter.lm_weights_from_hf).  Under torchrun: one rank per GPU, whole images per rank, one gather."""
    import argparse
    import json
    import os
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", choices=("llava", "qwen"), default="llava")
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--model-base", default=None)
    ap.add_argument("--image-folder", required=True)
    ap.add_argument("--question-file", required=True)
    ap.add_argument("--answers-file", required=True, help="'setting' in the name is replaced per run (default / temp_T / top_p_P / top_k_K)")
    ap.add_argument("--conv-mode", default="vicuna_v1")
    ap.add_argument("--num-chunks", type=int, default=1)
    ap.add_argument("--chunk-idx", type=int, default=0)
    ap.add_argument("--temperature", type=float, default=None, help="default 0.2 (llava) / 1.0 (qwen), as the reference's scripts")
    ap.add_argument("--top_p", type=float, default=None)
    ap.add_argument("--top_k", type=int, default=None)
    ap.add_argument("--num_beams", type=int, default=1)
    ap.add_argument("--max_new_tokens", type=int, default=None, help="default: 128 (llava, run_llava.py:267) / 20 (qwen: the generate call's own value, run_qwen.py:195)")
    ap.add_argument("--noise_step", type=int, default=500)
    ap.add_argument("--use_cd", action="store_true")
    ap.add_argument("--use_dd", action="store_true")
    ap.add_argument("--use_dd_unk", action="store_true")
    ap.add_argument("--cd_alpha", type=float, default=1.0)
    ap.add_argument("--cd_beta", type=float, default=0.1)
    ap.add_argument("--no-sweep", action="store_true", help="only the 'default' run")
    ap.add_argument("--gt-root", default=None, help="MME benchmark tree: also convert (naive / none / unk / none_unk) and score")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--preset", default="llava-1.5-7b")
    ap.add_argument("--vision-tower", default=None)
    ap.add_argument("--dtype", choices=("float16", "bfloat16"), default="float16")
    a = ap.parse_args(argv)
    if a.num_beams != 1:
        raise SystemExit("num_beams > 1: the reference patches sample() only (vcd_sample.py:325-326)")
    if a.temperature is None:
        a.temperature = 0.2 if a.arch == "llava" else 1.0
    from . import checkpoint as K
    from .shard import init_from_env
    rank, world, device = init_from_env()
    dtype = getattr(torch, a.dtype)
    questions = [json.loads(q) for q in open(os.path.expanduser(a.question_file))]
    gen_kw = dict(use_dd=a.use_dd, use_dd_unk=a.use_dd_unk, cd_alpha=a.cd_alpha, cd_beta=a.cd_beta)
    if a.arch == "llava":
        eng, tok, proc = K.load_llava(a.model_path, device, dtype=dtype, vision_tower=a.vision_tower, fallback_preset=a.preset)
        build = llava_mme_inputs(lambda prompt: K.tokenizer_image_token(tok, prompt), lambda name: K.clip_preprocess(proc, os.path.join(a.image_folder, name)),
                                 unk_token_id=tok.unk_token_id if tok.unk_token_id is not None else 0)
        decode = lambda ids: tok.decode(ids, skip_special_tokens=True)
        run_kw = dict(eos_token_id=tok.eos_token_id, pad_token_id=tok.pad_token_id or 0, stop_str="</s>", max_new_tokens=a.max_new_tokens or 128,
                      noise_step=a.noise_step if a.use_cd else None)
    else:
        eng, tok, _model, embed_prompt = K.load_qwen(a.model_path, device, dtype=dtype)      # run_qwen.py:176-186: the ViT reads the path in the prompt
        build = qwen_mme_inputs(embed_prompt, lambda f: os.path.join(a.image_folder, f))
        decode = lambda ids: tok.decode(ids, skip_special_tokens=True)
        run_kw = dict(eos_token_id=tok.eod_id, pad_token_id=tok.eod_id, max_new_tokens=a.max_new_tokens or 20, min_new_tokens=1)      # run_qwen.py:194-197
    gt = calibrate_gt = None
    if a.gt_root:
        gt = C.mme_load_gt(a.gt_root)
    out_scores = {}
    extra = dict(seed=a.seed) if a.seed is not None else {}
    sweep = []
    for tag, temp, top_p, top_k in _sweep_settings(a):
        path = os.path.expanduser(a.answers_file).replace("setting", tag)
        sweep.append(dict(tag=tag, temperature=temp, top_p=top_p, top_k=top_k, answers_path=path, experiment=os.path.splitext(os.path.basename(path))[0]))
    # ONE pass over the question file for all settings: a batch's vision-tower pass and prefills are shared by its settings (run_mme, `sweep`)
    res = run_mme(eng, questions, build, decode, model_id=os.path.basename(a.model_path.rstrip("/")), batch_questions=a.batch, gt=gt,
                  results_root=os.path.join(os.path.dirname(os.path.abspath(sweep[0]["answers_path"])), "eval_tool_answers") if gt else None,
                  chunk=(a.num_chunks, a.chunk_idx) if a.num_chunks > 1 else None, rank=rank, world=world, sweep=sweep, **gen_kw, **run_kw, **extra)
    for s_ in sweep:
        r_ = res["runs"][s_["tag"]]
        out_scores[s_["tag"]] = {k: (v and {"Perception": v["Perception"]["total"], "Cognition": v["Cognition"]["total"]}) for k, v in r_["scores"].items()}
        if rank == 0:
            print(json.dumps({"run": s_["tag"], "answers_file": s_["answers_path"], "n_answers": len(r_["answers"]), "scores": out_scores[s_["tag"]]}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
