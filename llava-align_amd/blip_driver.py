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
            # the zero image's prior depends on the question text alone (POPE repeats its texts over the images): Q-Former + LLM once per distinct prompt
            first_z: Dict[str, int] = {}
            where_z = [first_z.setdefault(p, len(first_z)) for p in prompts]
            pick_z = [prompts.index(p) for p in first_z]
            ie_zero = front.image_embeds(torch.zeros_like(uniq[:1])).expand(len(pick_z), -1, -1).contiguous()
            emb_z = front.assemble(front.embeds_to_llm(ie_zero, [qf_ids[j] for j in pick_z]), [llm_ids[j] for j in pick_z], embed)
            prior_kw = {k: v for k, v in base_kw.items() if k not in ("cd_beta", "cd_alpha")}
            noise = engine.generate(None, inputs_embeds=emb_n, max_length=1, **prior_kw)
            zeros = engine.generate(None, inputs_embeds=emb_z, max_length=1, **prior_kw)
            back_z = torch.tensor(where_z, dtype=torch.long).to(zeros.top_tok.device, non_blocking=True)
            rows.add(idx, map_pad_to_eos(main.tokens), [(main.top_tok, main.top_prob), (noise.top_tok, noise.top_prob), (zeros.top_tok[back_z], zeros.top_prob[back_z])])
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


LAVIS_MEAN, LAVIS_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)


def blip_image_eval(path: str, size: int = 224) -> torch.Tensor:
    """vis_processors['eval'] of load_model_and_preprocess('blip2_vicuna_instruct') (lavis/processors/blip_processors.py BlipImageEvalProcessor):
    Resize((size, size), bicubic) -> ToTensor -> Normalize(mean, std) of an RGB image file -> [3, size, size] fp32."""
    import numpy as np
    from PIL import Image
    im = Image.open(path).convert("RGB").resize((size, size), Image.BICUBIC)
    x = torch.from_numpy(np.asarray(im, dtype=np.float32) / 255.0).permute(2, 0, 1)
    return (x - torch.tensor(LAVIS_MEAN)[:, None, None]) / torch.tensor(LAVIS_STD)[:, None, None]


def main(argv=None):
    """python -m llava_align_amd.blip_driver --blip-checkpoint instruct_blip_vicuna7b.pth --llm-path vicuna-7b-v1.1 --bert-tokenizer bert-base-uncased
    --image-folder IMGS --question-file coco_pope_adversarial.json --answers-file OUT.jsonl [--use_cd --noise_step 500 --cd_beta 0.1 --top_p 1
    --seed 42 --num-chunks n --chunk-idx k]: the arguments of experiments/eval/calibrate/blip_calibrate.py:113-135 over the native front-end +
    engine.  The model the reference gets from `load_model_and_preprocess(name="blip2_vicuna_instruct", model_type="vicuna7b")` (:66) is given
    as its parts: the InstructBLIP state dict (visual_encoder.*, ln_vision.*, Qformer.*, query_tokens, llm_proj.* - LAVIS' released .pth, or
    safetensors), the Vicuna directory (HF weights + tokenizer) and the BERT tokenizer the Q-Former reads the instruction with.
    --cd_alpha is accepted and NOT forwarded, as in the reference (the sampler's 0.5 applies); --temperature / --top_k / --conv-mode
    are parsed and unused there too.  Under torchrun: one rank per GPU (BASELINE config #5 runs on 4), whole images per rank, one gather."""
    import argparse
    import os
    ap = argparse.ArgumentParser()
    ap.add_argument("--blip-checkpoint", required=True)
    ap.add_argument("--llm-path", required=True)
    ap.add_argument("--bert-tokenizer", required=True)
    ap.add_argument("--model-base", default=None)
    ap.add_argument("--image-folder", default="")
    ap.add_argument("--question-file", required=True)
    ap.add_argument("--answers-file", required=True)
    ap.add_argument("--conv-mode", default="llava_v1")
    ap.add_argument("--num-chunks", type=int, default=1)
    ap.add_argument("--chunk-idx", type=int, default=0)
    ap.add_argument("--temperature", type=float, default=1.0)
    ap.add_argument("--top_p", type=float, default=1.0)
    ap.add_argument("--top_k", type=int, default=None)
    ap.add_argument("--noise_step", type=int, default=500)
    ap.add_argument("--use_cd", action="store_true")
    ap.add_argument("--cd_alpha", type=float, default=1.0)
    ap.add_argument("--cd_beta", type=float, default=0.1)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--max_txt_len", type=int, default=128)
    ap.add_argument("--dtype", choices=("float16", "bfloat16"), default="float16")
    a = ap.parse_args(argv)
    from . import checkpoint as K
    from .blip_frontend import BlipWeights
    from .engine import LlavaWeights
    from .hf_adapter import blip_config_from_model
    from .shard import get_chunk, init_from_env
    rank, world, device = init_from_env()
    dtype = getattr(torch, a.dtype)
    sd = K.load_state_dict(a.blip_checkpoint)
    bcfg = blip_config_from_model(None, sd)
    front = InstructBlipFrontEnd(BlipWeights.from_state_dict(bcfg, sd, device, dtype=dtype))
    cfg = K.config_from_dir(a.llm_path, fallback="llava-1.5-7b")           # the LM side only: the preset's CLIP tower is unused on this path
    llm_sd = K.load_state_dict(a.llm_path)
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.lm_from_state_dict(cfg, llm_sd, device, dtype=dtype), device=device)
    llm_tok = K.load_tokenizer(a.llm_path)
    from transformers import AutoTokenizer
    bert = AutoTokenizer.from_pretrained(a.bert_tokenizer)
    questions = [json.loads(q) for q in open(os.path.expanduser(a.question_file))]
    if a.num_chunks > 1:
        questions = [questions[i] for i in get_chunk(len(questions), a.num_chunks, a.chunk_idx, group=1)]
    os.makedirs(os.path.dirname(os.path.abspath(os.path.expanduser(a.answers_file))), exist_ok=True)
    size = bcfg.vit.image
    res = run_blip_pope(eng, front, questions, lambda p: llm_tok(p).input_ids,
                        lambda p: bert(p, truncation=True, max_length=a.max_txt_len).input_ids,
                        lambda ids: llm_tok.decode(ids, skip_special_tokens=True), lambda name: blip_image_eval(os.path.join(a.image_folder, name), size),
                        answers_path=os.path.expanduser(a.answers_file), batch_questions=a.batch, use_cd=a.use_cd, noise_step=a.noise_step, cd_beta=a.cd_beta,
                        top_p=a.top_p, rank=rank, world=world, seed=a.seed)
    if rank == 0:
        nan = {k: v["nan_rows"] for k, v in res["scores"].items() if isinstance(v, dict) and v.get("nan_rows")}
        print(json.dumps({"scores": res["scores"], "n_answers": len(res["answers"]), "rows_whose_calibrated_vector_is_nan": nan}, indent=1))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
