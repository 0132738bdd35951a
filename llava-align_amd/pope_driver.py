"""POPE / yes-no evaluation driver over the native engine: the batched replacement of the reference's per-question loop
(experiments/eval/calibrate/llava_calibrate.py:130-219), same per-question inputs and outputs.

Per question the reference runs THREE generate() calls, each B = 1:
  main   image + question, VDD / VCD kwargs, 64 new tokens            -> `text`, `naive` (step-0 top-10 label dict), `logits_score`
  none   the question WITHOUT the image token, no image (llava_calibrate.py:46-61 with images=None)      -> `none`
  unk    the image token replaced by <unk>, no image (:59-61)                                            -> `unk`
The `none` / `unk` calls only ever use their step-0 scores (:80-85), so here they decode ONE token instead of up to 1024.
This driver runs each of the three as one engine call over a whole batch of questions (questions of one image adjacent, so
they share its ViT features and prompt-prefix KV), takes the step-0 top-10 probabilities from the fused sampling kernel,
builds the label dicts / label probabilities with the reference's first-wins rule (calibrate.py), writes the reference's JSONL
schema and scores it with the plain and the calibrated POPE scorers.

Tokenisation stays outside (the reference's conv template + tokenizer_image_token are tokenizer-specific): the caller passes
`encode(text, with_image) -> list[int]` (with -200 where the image goes) and `decode(ids) -> str`.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch

from . import calibrate as C
from .engine import IMAGE_TOKEN_INDEX, VddLlavaEngine

QUESTION_SUFFIX = " Please answer this question with one word."      # llava_calibrate.py:52,146


def llava_v1_prompt(question: str, with_image: bool) -> str:
    """conv_templates['llava_v1'] with one user turn (experiments/llava/conversation.py:335-345, get_prompt for SeparatorStyle.TWO)."""
    system = ("A chat between a curious human and an artificial intelligence assistant. "
              "The assistant gives helpful, detailed, and polite answers to the human's questions.")
    qs = ("<image>\n" if with_image else "") + question + QUESTION_SUFFIX
    return f"{system} USER: {qs} ASSISTANT:"


def _top_dicts(out, decode_token: Callable[[int], str]) -> List[Dict[str, float]]:
    tt, tp = out.top_tok.cpu().tolist(), out.top_prob.cpu().tolist()
    return [C.label_dict_from_top(t, p, decode_token) for t, p in zip(tt, tp)]


def cut_at_eos(toks: List[int], eos_set) -> List[int]:
    for k, t in enumerate(toks):                       # cut at the first EOS (the rest is padding)
        if t in eos_set:
            return toks[:k + 1]
    return toks


class ResultRows:
    """Per-question results of a rank's chunk as device tensors - generated ids (padded to `width` columns) and the step-0 top-10
    (token, probability) lists of the main pass and of the prior passes side by side - in the layout of the one result gather
    (shard.gather_results)."""

    def __init__(self, device, width: int, pad: int, n_sets: int, k: int = 10):
        self.device, self.width, self.pad, self.n_sets, self.k = device, int(width), int(pad), n_sets, k
        self.qids: List[int] = []
        self.tokens, self.n_tok, self.top_tok, self.top_prob = [], [], [], []

    def add(self, qids: Sequence[int], main_tokens: torch.Tensor, tops: Sequence):
        """tops: one (top_tok [n, k], top_prob [n, k]) pair per set, main pass first."""
        n, T = main_tokens.shape
        assert len(tops) == self.n_sets and T <= self.width and n == len(qids)
        self.qids += list(qids)
        self.tokens.append(torch.nn.functional.pad(main_tokens, (0, self.width - T), value=self.pad))
        self.n_tok.append(torch.full((n,), T, dtype=torch.long, device=self.device))
        self.top_tok.append(torch.cat([t for t, _ in tops], 1))
        self.top_prob.append(torch.cat([p_ for _, p_ in tops], 1))

    def gather(self, plan, n_total: int) -> dict:
        """-> host lists for ALL questions on every rank: tokens[i] (valid columns only), tops[s][i] = (tok list, prob list)."""
        from .shard import gather_results
        dev, k = self.device, self.n_sets * self.k
        z = lambda *shape, dt=torch.long: torch.zeros(*shape, dtype=dt, device=dev)
        res = gather_results(torch.tensor(self.qids, dtype=torch.long, device=dev),
                             torch.cat(self.tokens) if self.tokens else z(0, self.width), torch.cat(self.n_tok) if self.n_tok else z(0),
                             torch.cat(self.top_tok) if self.top_tok else z(0, k), torch.cat(self.top_prob) if self.top_prob else z(0, k, dt=torch.float32),
                             n_total, pad=self.pad, capacity=plan.capacity, width=self.width, world=plan.world)
        if not bool((res["count"] == 1).all().item()):
            raise RuntimeError(f"result gather: {int((res['count'] != 1).sum())} of {n_total} questions were not delivered exactly once")
        n_tok, toks = res["n_tokens"].cpu().tolist(), res["tokens"].cpu().tolist()
        tt, tp = res["top_tok"].cpu().tolist(), res["top_prob"].cpu().tolist()
        K = self.k
        return {"tokens": [r[:n] for r, n in zip(toks, n_tok)],
                "tops": [[(tt[i][s_ * K:(s_ + 1) * K], tp[i][s_ * K:(s_ + 1) * K]) for i in range(n_total)] for s_ in range(self.n_sets)]}


def run_pope(engine: VddLlavaEngine, questions: Sequence[dict], encode: Callable[[str, bool], List[int]],
             decode: Callable[[List[int]], str], load_image: Callable[[str], torch.Tensor], answers_path: Optional[str] = None,
             model_id: str = "llava-align_amd", batch_questions: int = 384, unk_token_id: int = 0, eos_token_id=None,
             pad_token_id: Optional[int] = None, stop_str: Optional[str] = "</s>", max_new_tokens: int = 64, noise_step: Optional[int] = None,
             rank: Optional[int] = None, world: Optional[int] = None, batch_invariant: Optional[bool] = None,
             image_priors: Sequence[str] = (), reuse_unk_branch: bool = True, image_workers: int = 8, **generate_kw) -> dict:
    """questions: dicts with question_id, image, text, label (the POPE json lines).  generate_kw: cd_alpha, cd_beta, use_dd,
    use_dd_unk, temperature, top_p, top_k, seed ... exactly the reference's model.generate kwargs (llava_calibrate.py:161-177);
    noise_step adds the VCD branch (images_cd = add_diffusion_noise(image, noise_step), :152-155).

    Data-parallel (the reference: one process per chunk, `--num-chunks / --chunk-idx`, scripts/pope/run_dataset.sh:14-33):
    with `rank` / `world` (default: the initialised torch.distributed group, else one rank) every rank calls this with the SAME
    question list, decodes its contiguous chunk of whole images (shard.ShardPlan), the per-question results are gathered in ONE
    collective, and rank 0 writes the answers file; every rank returns the full result.  An explicit `seed` is offset by the rank
    (sampled runs then differ from a 1-rank run).  Deterministic decodes (cd_greedy / top_k = 1 / do_sample = False) run in
    batch-invariant mode unless `batch_invariant=False` (shard.resolve_batch_invariant, ops.GEMM_BATCH_INVARIANT): their answers are
    then token for token the same on 1 and on N ranks and for every batch_questions; with the tuned kernel forms (the default for
    sampled runs) a row's low-order bits depend on who shares its batch.
    image_priors: further content-free passes that keep the image prompt and swap the IMAGE - 'noise' = add_diffusion_noise(image, 999),
    'zeros', 'ones' (llava_calibrate.py:188-190 prepares them, experiments/eval/calibrate/test_samples_llava.py:134-145 runs them: plain
    sampling, step-0 label dict) - written under those keys and scored like the text priors.
    image_workers: threads that run `load_image` for the NEXT batch's files while the GPU runs the current batch.
    reuse_unk_branch: with use_dd_unk the `unk` prior pass would feed the ids the main pass's `unk` branch already ran (image slot -> <unk>:
    llava_calibrate.py:59-60 and vcd_sample.py:154-155 build the same row): its step-0 label dict is read off that branch
    (generate(branch_priors=True)) and only the `none` prompts are prefilled again.  False: run it as a pass of its own, as the reference does.
    Returns {"answers": [...], "scores": {"string_match": ..., "naive": ..., "none": ..., "unk": ..., "none_unk": ...}}."""
    import contextlib
    image_priors = tuple(image_priors)
    if any(n_ not in ("noise", "zeros", "ones") for n_ in image_priors):
        raise ValueError("image_priors must be among 'noise', 'zeros', 'ones'")
    from . import ops
    from .shard import ShardPlan, resolve_batch_invariant
    order = sorted(range(len(questions)), key=lambda i: (questions[i]["image"], i))      # one image's questions adjacent
    plan = ShardPlan([questions[i]["image"] for i in order], rank, world)
    mine = [order[p_] for p_ in plan.mine]
    decode_token = lambda t: decode([t])
    if generate_kw.get("seed") is not None:
        generate_kw = dict(generate_kw, seed=int(generate_kw["seed"]) + plan.rank)
    rows = ResultRows(engine.device, max_new_tokens, pad_token_id if pad_token_id is not None else 0, n_sets=3 + len(image_priors))
    img_cache: Dict[str, torch.Tensor] = {}
    prior_prompts = prior_distinct = 0
    invariant = resolve_batch_invariant(batch_invariant, plan.world, generate_kw)
    def host_inputs(b0):
        """What a batch needs from the HOST - image files decoded / preprocessed (`load_image`), prompts tokenised (`encode`) - and nothing of the
        device: prepared for batch k + 1 on a worker thread while the GPU runs batch k (the main thread sits in a stream wait then; with real
        files the CLIP preprocessing of 128 images is of the order of the batch's GPU time)."""
        idx = mine[b0:b0 + batch_questions]
        qs = [questions[i] for i in idx]
        names = list(dict.fromkeys(q["image"] for q in qs if q["image"] not in img_cache))
        host_imgs: Dict[str, torch.Tensor] = dict(zip(names, loaders.map(load_image, names)))        # (PIL decoding / resizing releases the GIL)
        ids_main = [torch.tensor(encode(q["text"], True)) for q in qs]
        ids_none = [torch.tensor(encode(q["text"], False)) for q in qs]
        ids_unk = [torch.tensor([unk_token_id if t == IMAGE_TOKEN_INDEX else t for t in r.tolist()]) for r in ids_main]    # :59-60
        return idx, qs, host_imgs, ids_main, ids_none, ids_unk

    from concurrent.futures import ThreadPoolExecutor
    starts = list(range(0, len(mine), batch_questions))
    with (ops.batch_invariant() if invariant else contextlib.nullcontext()), ThreadPoolExecutor(max_workers=1) as pool, \
            ThreadPoolExecutor(max_workers=image_workers) as loaders:
        ahead_inputs = pool.submit(host_inputs, starts[0]) if starts else None
        for k, b0 in enumerate(starts):
            idx, qs, host_imgs, ids_main, ids_none, ids_unk = ahead_inputs.result()
            for name, im in host_imgs.items():
                if name not in img_cache:
                    img_cache[name] = im.to(engine.device)
            if k + 1 < len(starts):
                ahead_inputs = pool.submit(host_inputs, starts[k + 1])
            imgs = [img_cache[q["image"]] for q in qs]
            kw = dict(generate_kw)
            if noise_step is not None:
                from .vcd_add_noise import add_diffusion_noise
                # fresh noise per QUESTION, as the reference draws it inside its per-question loop (llava_calibrate.py:152-155)
                kw["images_cd"] = [add_diffusion_noise(img_cache[q["image"]], noise_step) for q in qs]
            # the engine's `unk` branch replaces the slot by token 0 (the reference's literal, :154-155); it exists unless the VCD branch took its place
            from_branch = bool(reuse_unk_branch and generate_kw.get("use_dd_unk") and unk_token_id == 0 and noise_step is None
                               and generate_kw.get("do_sample", True) is not False)
            main = engine.generate(ids_main, images=imgs, max_new_tokens=max_new_tokens, n_top=10, eos_token_id=eos_token_id,
                                   pad_token_id=pad_token_id, branch_priors=from_branch, **kw)
            # content-free priors: plain sampling (no image -> no contrast branch), step-0 distribution only
            plain_kw = {k: v for k, v in generate_kw.items() if k in ("temperature", "top_p", "top_k", "seed", "cd_alpha", "cd_beta")}
            n = len(qs)
            branch_top = getattr(main, "branch_top", None) or {}
            reuse = from_branch and "unk" in branch_top
            # (one call for both priors: text-only prompts that share the conversation template's system prompt as a prefix slot.)  POPE asks the
            # same few dozen questions ("Is there a <object> in the image?") about hundreds of images, and a text-only prompt's step-0 label dict
            # is a function of its ids alone (the top-n of the warped distribution: nothing is drawn): every DISTINCT prompt is run once
            want = ids_none if reuse else ids_none + ids_unk
            first: Dict[tuple, int] = {}
            where = [first.setdefault(tuple(r.tolist()), len(first)) for r in want]
            uniq = [None] * len(first)
            for r, j in zip(want, where):
                if uniq[j] is None:
                    uniq[j] = r
            prior = engine.generate(uniq, images=None, max_new_tokens=1, n_top=10, **plain_kw)
            back = torch.tensor(where, dtype=torch.long).to(prior.top_tok.device, non_blocking=True)
            p_tok, p_prob = prior.top_tok[back], prior.top_prob[back]
            unk_top = branch_top["unk"] if reuse else (p_tok[n:], p_prob[n:])
            tops = [(main.top_tok, main.top_prob), (p_tok[:n], p_prob[:n]), unk_top]
            prior_prompts = prior_prompts + len(want)
            prior_distinct = prior_distinct + len(uniq)
            for name in image_priors:
                if name == "noise":                                  # fresh noise per question, as the reference draws it inside its loop
                    from .vcd_add_noise import add_diffusion_noise
                    swapped = [add_diffusion_noise(im, 999) for im in imgs]
                    o = engine.generate(ids_main, images=swapped, max_new_tokens=1, n_top=10, **plain_kw)
                    tops.append((o.top_tok, o.top_prob))
                else:                                                # ONE tensor object: every question shares its features and prompt prefix,
                    const = (torch.zeros_like if name == "zeros" else torch.ones_like)(imgs[0])          # and every distinct question text runs once
                    first_i: Dict[tuple, int] = {}
                    where_i = [first_i.setdefault(tuple(r.tolist()), len(first_i)) for r in ids_main]
                    uniq_i = [None] * len(first_i)
                    for r, j in zip(ids_main, where_i):
                        if uniq_i[j] is None:
                            uniq_i[j] = r
                    o = engine.generate(uniq_i, images=[const] * len(uniq_i), max_new_tokens=1, n_top=10, **plain_kw)
                    back_i = torch.tensor(where_i, dtype=torch.long).to(o.top_tok.device, non_blocking=True)
                    tops.append((o.top_tok[back_i], o.top_prob[back_i]))
            rows.add(idx, main.tokens, tops)
            ahead = {questions[i]["image"] for i in mine[b0 + batch_questions:b0 + 2 * batch_questions]}
            for gone in [name for name in img_cache if name not in ahead]:
                img_cache.pop(gone)                                # images are revisited only within a sorted neighbourhood (the worker's batch = `ahead`)
            engine.clear_image_cache()
    got = rows.gather(plan, len(questions))                    # ONE collective; every rank holds every question's results behind it
    eos_set = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
    dicts = [[C.label_dict_from_top(t, p_, decode_token) for t, p_ in got["tops"][s_]] for s_ in range(3 + len(image_priors))]
    ordered = []
    for i, q in enumerate(questions):
        text = decode(cut_at_eos(got["tokens"][i], eos_set)).strip()
        if stop_str and text.endswith(stop_str):
            text = text[:-len(stop_str)]
        ordered.append({"question_id": q["question_id"], "prompt": q["text"], "text": text.strip(), "model_id": model_id,
                        "image": q["image"], "logits_score": C.get_prob_from_logits(dicts[0][i]), "naive": dicts[0][i],
                        "unk": dicts[2][i], "none": dicts[1][i], **{n_: dicts[3 + k][i] for k, n_ in enumerate(image_priors)}, "metadata": {}})
    if answers_path is not None and plan.rank == 0:
        with C.AnswerWriter(answers_path) as w:
            for a in ordered:
                w.write(a["question_id"], a["prompt"], a["text"], a["model_id"], a["image"], a["logits_score"], a["naive"], a["unk"], a["none"],
                        extra={n_: a[n_] for n_ in image_priors})
    scores = {}
    if all("label" in q for q in questions):
        gt = [{"question_id": q["question_id"], "label": q["label"]} for q in questions]
        scores["string_match"] = _try(C.pope_scores, gt, ordered)
        for name in ("naive", "none", "unk", "none_unk") + image_priors:
            scores[name] = _try(C.pope_scores_calibrated, gt, ordered, name)
    return {"answers": ordered, "scores": scores, "rank": plan.rank, "world": plan.world, "batch_invariant": invariant,
            "prior_prompts": prior_prompts, "prior_prompts_run": prior_distinct}


def _try(f, *a):
    try:
        return f(*a)
    except ZeroDivisionError:            # the reference scorers divide by tp + fp etc.: undefined on degenerate (tiny / random) runs
        return None


def main(argv=None):
    """python -m llava_align_amd.pope_driver --model-path DIR --question-file Q.json --image-folder IMGS --answers-file OUT.jsonl
    [--use_dd --use_dd_unk --cd_alpha 1 --cd_beta 0.1 --temperature 0.2 --noise_step N]: the reference CLI's arguments
    (llava_calibrate.py:222-246) over the native engine.  Needs a LLaVA-1.5 checkpoint directory (HF safetensors + tokenizer).
    On a node: `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m llava_align_amd.pope_driver ...`
    - one rank per GPU, the question list sharded by images, one RCCL gather, rank 0 writes OUT.jsonl and prints the scores (this
    replaces the reference's --num-chunks / --chunk-idx processes + `cat` of their files, scripts/pope/run_dataset.sh:14-33)."""
    import argparse
    import json
    import os
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--question-file", required=True)
    ap.add_argument("--image-folder", required=True)
    ap.add_argument("--answers-file", required=True)
    ap.add_argument("--preset", default="llava-1.5-7b", help="shapes for whatever the checkpoint's config.json does not say (its CLIP tower is named on the hub)")
    ap.add_argument("--vision-tower", default=None, help="local directory of the CLIP tower `mm_vision_tower` names (weights + preprocessor_config.json)")
    ap.add_argument("--max_new_tokens", type=int, default=64)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--cd_greedy", action="store_true", help="arg-max of the contrasted distribution (deterministic: 1-rank and N-rank runs agree token for token)")
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--top_p", type=float, default=None)
    ap.add_argument("--top_k", type=int, default=None)
    ap.add_argument("--noise_step", type=int, default=None)
    ap.add_argument("--use_dd", action="store_true")
    ap.add_argument("--use_dd_unk", action="store_true")
    ap.add_argument("--cd_alpha", type=float, default=1.0)
    ap.add_argument("--cd_beta", type=float, default=0.1)
    ap.add_argument("--batch", type=int, default=384)
    ap.add_argument("--image-priors", nargs="*", default=[], choices=("noise", "zeros", "ones"),
                    help="further content-free passes with the image swapped (test_samples_llava.py:134-145)")
    ap.add_argument("--dtype", choices=("float16", "bfloat16"), default="float16", help="model dtype (the reference loads fp16, builder.py:40)")
    a = ap.parse_args(argv)
    from .checkpoint import clip_preprocess, load_llava, tokenizer_image_token
    from .shard import init_from_env
    rank, world, device = init_from_env()
    eng, tok, proc = load_llava(a.model_path, device, dtype=getattr(torch, a.dtype), vision_tower=a.vision_tower, fallback_preset=a.preset)
    encode = lambda text, with_image: tokenizer_image_token(tok, llava_v1_prompt(text, with_image))

    questions = [json.loads(q) for q in open(os.path.expanduser(a.question_file))]
    os.makedirs(os.path.dirname(os.path.abspath(a.answers_file)), exist_ok=True)
    extra = {k: v for k, v in (("seed", a.seed), ("cd_greedy", a.cd_greedy or None)) if v is not None}
    res = run_pope(eng, questions, encode, lambda ids: tok.decode(ids, skip_special_tokens=True),
                   lambda name: clip_preprocess(proc, os.path.join(a.image_folder, name)),
                   answers_path=a.answers_file, model_id=os.path.basename(a.model_path.rstrip("/")), batch_questions=a.batch,
                   unk_token_id=tok.unk_token_id if tok.unk_token_id is not None else 0, eos_token_id=tok.eos_token_id,
                   pad_token_id=tok.pad_token_id or 0, max_new_tokens=a.max_new_tokens, noise_step=a.noise_step, use_dd=a.use_dd,
                   use_dd_unk=a.use_dd_unk, cd_alpha=a.cd_alpha, cd_beta=a.cd_beta, temperature=a.temperature, top_p=a.top_p, top_k=a.top_k,
                   rank=rank, world=world, image_priors=tuple(a.image_priors), **extra)
    if rank == 0:
        nan = {k: v["nan_rows"] for k, v in res["scores"].items() if isinstance(v, dict) and v.get("nan_rows")}
        print(json.dumps({"scores": res["scores"], "batch_invariant": res["batch_invariant"], "world": world,
                          "rows_whose_calibrated_vector_is_nan": nan}, indent=1))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
