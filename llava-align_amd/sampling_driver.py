"""Open-ended generation driver over the native engine (BASELINE config #3: LLaVA-Bench / POPE answers with sampling): the batched
replacement of the reference's per-question loop experiments/eval/sampling/llava_sampling.py:57-126.

Per question the reference runs ONE `model.generate(input_ids, images=..., images_cd=..., cd_alpha, cd_beta, use_dd, use_dd_unk,
do_sample, temperature, top_p, top_k, max_new_tokens=1024)` at B = 1 (:96-109) and writes `{question_id, prompt, text, model_id, image,
metadata}` (:119-124).  Here the whole list goes to the engine: `generate_list` keeps `in_flight` questions decoding and refills the
slots of finished ones (answers are 20 - 1,000 tokens: a fixed batch would decode to its slowest member), also with the VCD branch
(`noise_step`: images_cd = add_diffusion_noise(image, noise_step), :88-91).
The prompt is the conv template's with '<image>\\n' in front of the question, + ' Please answer this question with one word.' when the
question file is a POPE file (:71-76).  Tokenisation stays outside: `encode(prompt) -> ids` with -200 at '<image>', `decode(ids)`.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch

from .engine import VddLlavaEngine
from .pope_driver import QUESTION_SUFFIX, cut_at_eos


def llava_v1_user_prompt(user_text: str) -> str:
    """conv_templates['llava_v1'] with one user turn (experiments/llava/conversation.py:335-345)."""
    system = ("A chat between a curious human and an artificial intelligence assistant. "
              "The assistant gives helpful, detailed, and polite answers to the human's questions.")
    return f"{system} USER: {user_text} ASSISTANT:"


def run_sampling(engine: VddLlavaEngine, questions: Sequence[dict], encode: Callable[[str], List[int]], decode: Callable[[List[int]], str],
                 load_image: Callable[[str], torch.Tensor], answers_path: Optional[str] = None, model_id: str = "llava-align_amd",
                 pope_suffix: bool = False, in_flight: int = 90, max_new_tokens: int = 1024, eos_token_id=2, pad_token_id: Optional[int] = 0,
                 stop_str: Optional[str] = "</s>", noise_step: Optional[int] = None, rank: Optional[int] = None, world: Optional[int] = None,
                 batch_invariant: Optional[bool] = None, **generate_kw) -> dict:
    """questions: dicts with question_id, image, text.  generate_kw: cd_alpha, cd_beta, use_dd, use_dd_unk, temperature, top_p, top_k, seed,
    cd_greedy - the reference's generate kwargs (llava_sampling.py:96-109; temperature 0 there means greedy: pass do_sample=False).
    rank / world (default: the initialised torch.distributed group): every rank decodes its contiguous chunk of whole images
    (shard.ShardPlan), ONE collective gathers the answers, rank 0 writes the file; every rank returns the full result.  Deterministic decodes
    run in batch-invariant mode unless told otherwise (shard.resolve_batch_invariant): same answers on 1 and N ranks, at any in_flight.
    Returns {"answers": [...], "stats": the engine's stats of this rank's chunk}."""
    import contextlib
    from . import ops
    from .shard import ShardPlan, gather_results, resolve_batch_invariant
    order = sorted(range(len(questions)), key=lambda i: (questions[i]["image"], i))
    plan = ShardPlan([questions[i]["image"] for i in order], rank, world)
    mine = [order[p_] for p_ in plan.mine]
    if generate_kw.get("seed") is not None:
        generate_kw = dict(generate_kw, seed=int(generate_kw["seed"]) + plan.rank)
    dev = engine.device
    cache = {}
    imgs, ids = [], []
    for i in mine:
        q = questions[i]
        if q["image"] not in cache:
            cache[q["image"]] = load_image(q["image"]).to(dev)
        imgs.append(cache[q["image"]])                                   # the SAME tensor for questions about one image: shared features / prefix
        text = "<image>\n" + q["text"] + (QUESTION_SUFFIX if pope_suffix else "")
        ids.append(torch.tensor(encode(llava_v1_user_prompt(text))))
    invariant = resolve_batch_invariant(batch_invariant, plan.world, generate_kw)
    stats = {}
    with (ops.batch_invariant() if invariant else contextlib.nullcontext()):
        if not mine:
            toks = torch.zeros(0, 1, dtype=torch.long, device=dev)
        elif generate_kw.get("do_sample", True) is not False:
            kw = {k: v for k, v in generate_kw.items() if k != "do_sample"}
            if noise_step is not None:
                from .vcd_add_noise import add_diffusion_noise
                kw["images_cd"] = [add_diffusion_noise(im, noise_step) for im in imgs]                 # fresh noise per question (:88-91)
            out = engine.generate_list(ids, imgs, in_flight=in_flight, max_new_tokens=max_new_tokens, eos_token_id=eos_token_id, pad_token_id=pad_token_id, **kw)
            toks, stats = out.tokens, out.stats
        else:                       # do_sample=False: greedy_search is not patched - plain arg-max decoding, batch after batch
            parts = []
            for b0 in range(0, len(mine), in_flight):
                sl = slice(b0, b0 + in_flight)
                o = engine.generate(ids[sl], images=imgs[sl], max_new_tokens=max_new_tokens, eos_token_id=eos_token_id, pad_token_id=pad_token_id, **generate_kw)
                parts.append(o.tokens)
                stats = o.stats
            T = max(p_.shape[1] for p_ in parts)
            toks = torch.cat([torch.nn.functional.pad(p_, (0, T - p_.shape[1]), value=pad_token_id if pad_token_id is not None else 0) for p_ in parts])
    width = max_new_tokens
    pad = pad_token_id if pad_token_id is not None else 0
    n_tok = torch.full((toks.shape[0],), toks.shape[1], dtype=torch.long, device=dev)
    res = gather_results(torch.tensor(mine, dtype=torch.long, device=dev), torch.nn.functional.pad(toks, (0, width - toks.shape[1]), value=pad), n_tok, None, None,
                         len(questions), pad=pad, capacity=plan.capacity, width=width, world=plan.world)
    if not bool((res["count"] == 1).all().item()):
        raise RuntimeError(f"result gather: {int((res['count'] != 1).sum())} of {len(questions)} questions were not delivered exactly once")
    eos_set = set([eos_token_id] if isinstance(eos_token_id, int) else (eos_token_id or []))
    rows, lens = res["tokens"].cpu().tolist(), res["n_tokens"].cpu().tolist()
    answers = []
    for i, q in enumerate(questions):
        text = decode(cut_at_eos(rows[i][:lens[i]], eos_set)).strip()
        if stop_str and text.endswith(stop_str):
            text = text[:-len(stop_str)]
        answers.append({"question_id": q["question_id"], "prompt": q["text"], "text": text.strip(), "model_id": model_id, "image": q["image"], "metadata": {}})
    if answers_path is not None and plan.rank == 0:
        import json
        import os
        os.makedirs(os.path.dirname(os.path.abspath(answers_path)), exist_ok=True)
        with open(answers_path, "w") as f:
            for a in answers:
                f.write(json.dumps(a) + "\n")
    return {"answers": answers, "stats": stats, "rank": plan.rank, "world": plan.world, "batch_invariant": invariant}


def main(argv=None):
    """python -m llava_align_amd.sampling_driver --model-path DIR --question-file Q.jsonl --image-folder IMGS --answers-file OUT-setting.jsonl
    [--use_dd --use_dd_unk --use_cd --noise_step 500 --cd_alpha 1 --cd_beta 0.1 --seed 42 --no-sweep]: the arguments of
    experiments/eval/sampling/llava_sampling.py:128-195 and its runs - 'default' (temperature 1, no top-p / top-k), then, unless --use_cd,
    temperature 0.05 ... 1.0, top_p 0 ... 1.0, top_k 1 ... 500, each into `answers-file` with 'setting' replaced.  Under torchrun: one rank
    per GPU, whole images per rank, one gather, rank 0 writes the files."""
    import argparse
    import json
    import os
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--model-base", default=None)
    ap.add_argument("--image-folder", default="")
    ap.add_argument("--question-file", required=True)
    ap.add_argument("--answers-file", required=True)
    ap.add_argument("--conv-mode", default="llava_v1")
    ap.add_argument("--num-chunks", type=int, default=1)
    ap.add_argument("--chunk-idx", type=int, default=0)
    ap.add_argument("--noise_step", type=int, default=500)
    ap.add_argument("--use_cd", action="store_true")
    ap.add_argument("--cd_alpha", type=float, default=1.0)
    ap.add_argument("--cd_beta", type=float, default=0.1)
    ap.add_argument("--use_dd", action="store_true")
    ap.add_argument("--use_dd_unk", action="store_true")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--max_new_tokens", type=int, default=1024)
    ap.add_argument("--in-flight", type=int, default=90)
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--naive", action="store_true", help="experiments/eval/llava_naive.py:39-98: ONE run at --temperature / --top_p / --top_k (greedy at "
                    "temperature 0), always with the one-word suffix, into --answers-file as given")
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--top_p", type=float, default=None)
    ap.add_argument("--top_k", type=int, default=None)
    ap.add_argument("--preset", default="llava-1.5-7b")
    ap.add_argument("--vision-tower", default=None)
    ap.add_argument("--dtype", choices=("float16", "bfloat16"), default="float16")
    a = ap.parse_args(argv)
    from . import checkpoint as K
    from .shard import get_chunk, init_from_env
    rank, world, device = init_from_env()
    eng, tok, proc = K.load_llava(a.model_path, device, dtype=getattr(torch, a.dtype), vision_tower=a.vision_tower, fallback_preset=a.preset)
    questions = [json.loads(q) for q in open(os.path.expanduser(a.question_file))]
    if a.num_chunks > 1:
        questions = [questions[i] for i in get_chunk(len(questions), a.num_chunks, a.chunk_idx, group=1)]
    runs = [("default", 1.0, None, None)]
    if a.naive:
        runs = [(None, a.temperature, a.top_p, a.top_k)]
    elif not (a.no_sweep or a.use_cd):                                    # llava_sampling.py:162-195
        runs += [(f"temp_{t}", float(t), None, None) for t in np.round(np.arange(0.05, 1.05, 0.05), 2)]
        runs += [(f"top_p_{p_}", 1.0, float(p_), None) for p_ in np.arange(0, 1.05, 0.05)]
        runs += [(f"top_k_{k}", 1.0, None, k) for k in (1, 2, 5, 10, 20, 50, 100, 200, 500)]
    for tag, temp, top_p, top_k in runs:
        path = os.path.expanduser(a.answers_file)
        path = path.replace("setting", tag) if tag else path
        greedy = dict(do_sample=False) if (a.naive and temp <= 0) else {}
        res = run_sampling(eng, questions, lambda p: K.tokenizer_image_token(tok, p), lambda ids: tok.decode(ids, skip_special_tokens=True),
                           lambda name: K.clip_preprocess(proc, os.path.join(a.image_folder, name)), answers_path=path,
                           model_id=os.path.basename(a.model_path.rstrip("/")), pope_suffix=a.naive or "POPE" in a.question_file, in_flight=a.in_flight,
                           max_new_tokens=a.max_new_tokens, eos_token_id=tok.eos_token_id, pad_token_id=tok.pad_token_id or 0,
                           noise_step=a.noise_step if a.use_cd else None, rank=rank, world=world, use_dd=a.use_dd, use_dd_unk=a.use_dd_unk,
                           cd_alpha=a.cd_alpha, cd_beta=a.cd_beta, temperature=temp if temp > 0 else 1.0, top_p=top_p, top_k=top_k, seed=a.seed, **greedy)
        if rank == 0:
            print(json.dumps({"run": tag or "naive", "answers_file": path, "n_answers": len(res["answers"]), "stats": {k: v for k, v in res["stats"].items() if not hasattr(v, "__len__") or isinstance(v, str)}}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
