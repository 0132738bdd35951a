"""BASELINE.json configs #4 and #5 on ONE MI355X through the drivers (synthetic weights / inputs, SURVEY 8d shapes):
  #4  Qwen-VL-7B LM shape (32 layers, d 4096, qkv bias, V = 151,936), MME-like: 504 items = 252 images x 2 questions, prompt = 256 image
      slots + ~40 text tokens as embeddings, use_dd_unk dual pass (the image-free branch re-runs the same inputs), 20 new
      tokens, min_new_tokens 1, pad = eos = eod, step-0 top-10, + the two text-only prior passes, converter and scorer (mme_driver.run_mme)
  #5  InstructBLIP-Vicuna-7B: EVA-ViT-g (39 layers, 1408 wide) + Q-Former (12 layers) + Vicuna-7B, POPE-like: 3 x 128 questions (6 per image),
      VCD branch from add_diffusion_noise(image, 500), alpha 0.5, beta 0.1, top-p 1, noise / zeros priors (blip_driver.run_blip_pope)
Prints one JSON line per config.  python tools/config45_probe.py [4|5]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

dev = "cuda:0"
which = sys.argv[1:] or ["4", "5"]
decode = lambda ids: " ".join(("yes", "no", "w")[t % 3] + str(t % 7) for t in ids)


def config4():
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    from llava_align_amd.mme_driver import MME_SUBSETS, qwen_mme_inputs, run_mme
    cfg = preset("qwen-vl-7b-lm")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, dev, seed=0, lm_head_gain=4.0), device=dev, use_graph=True)
    rng = np.random.default_rng(4)
    qs, gt = [], {}
    for i in range(252):
        cat = MME_SUBSETS[i % 8]
        for k in range(2):
            text = f"Is item {i} {k} shown in the picture?"
            qs.append({"question_id": f"{cat}/{i:04d}.png", "image": f"{cat}/{i:04d}.png", "category": cat, "text": text + " Please answer yes or no."})
            gt[(cat, f"{i:04d}.txt", text + " Please answer yes or no.")] = ("Yes", "No")[(i + k) % 2]
    table = eng.w.t["embed"]
    g = torch.Generator(device=dev).manual_seed(2)
    feats = {}

    lead = table[torch.tensor([151857, 151857], device=dev)]

    def embed_prompt(text, path):                       # 256 resampler slots + ~40 text tokens (the Qwen ViT / resampler are upstream of this path)
        n = 40 + (len(text) % 9)
        e = table[torch.from_numpy(rng.integers(3, 151000, size=n)).to(dev)]
        if path is not None:
            if path not in feats:
                feats[path] = (torch.randn(256, cfg.lm.d, device=dev, generator=g) * 0.02).to(torch.bfloat16)
            e = torch.cat([lead, feats[path], e[2:]], 0)
            return e, 258                                # '<img>' tokens + the 256 slots: the same for both questions about this image
        return e
    kw = dict(batch_questions=504, max_new_tokens=20, min_new_tokens=1, eos_token_id=151643, pad_token_id=151643, gt=gt,
              results_root="/tmp/mme_res", experiment="qwen", use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, seed=1)
    build = qwen_mme_inputs(embed_prompt)
    run_mme(eng, qs, build, decode, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_mme(eng, qs, build, decode, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_tok = sum(len(a["text"].split()) for a in res["answers"])
    print(json.dumps({"config": 4, "model": "qwen-vl-7b LM shape (V 151,936, qkv bias), synthetic weights", "items": len(qs), "seconds": round(dt, 2),
                      "items_per_s": round(len(qs) / dt, 1), "main_pass_new_tokens": 20, "generated_tokens_per_s_main_pass_equiv": round(len(qs) * 20 / dt, 1),
                      "answer_words": n_tok, "passes": "main (dual pass, 20 tokens) + none + unk (1 token each) + convert + score",
                      "hbm_peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1), "scores_naive_perception": res["scores"]["naive"]["Perception"]["total"]}), flush=True)
    del eng
    torch.cuda.empty_cache()


def config5():
    from llava_align_amd.blip_driver import run_blip_pope
    from llava_align_amd.blip_frontend import BlipConfig, BlipWeights, InstructBlipFrontEnd
    from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
    cfg = preset("llava-1.5-7b")                         # Vicuna-7B LM (the CLIP tower of the preset is unused on this path)
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, dev, seed=0, lm_head_gain=4.0), device=dev, use_graph=True)
    front = InstructBlipFrontEnd(BlipWeights.random(BlipConfig(), dev, seed=1))
    n_q = 384
    images = {f"im{i}.jpg": torch.randn(3, 224, 224, generator=torch.Generator().manual_seed(i)) for i in range(n_q // 6)}
    qs = [{"question_id": i, "image": f"im{i // 6}.jpg", "text": f"Is there a thing number {i} in the image?", "label": ("yes", "no")[i % 2]} for i in range(n_q)]
    tok_llm = lambda p: [1] + [(sum(map(ord, w)) * 31 + 7) % 31990 + 3 for w in p.split()]
    tok_qf = lambda p: [101] + [(sum(map(ord, w)) * 17) % 30000 + 200 for w in p.split()][:30] + [102]
    kw = dict(batch_questions=128, use_cd=True, noise_step=500, cd_beta=0.1, max_length=20, seed=1)
    run_blip_pope(eng, front, qs[:128], tok_llm, tok_qf, decode, lambda n: images[n], **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_blip_pope(eng, front, qs, tok_llm, tok_qf, decode, lambda n: images[n], **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": 5, "model": "InstructBLIP-Vicuna-7B shape (EVA-ViT-g 39 x 1408, Q-Former 12 x 768, Vicuna-7B), synthetic weights",
                      "items": n_q, "seconds": round(dt, 2), "items_per_s": round(n_q / dt, 1), "max_length": 20,
                      "passes": "EVA-ViT + Q-Former for image, noised image (t=500), noise (t=999), zeros; main VCD generate + 2 one-token priors",
                      "hbm_peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1), "n_answers": len(res["answers"])}), flush=True)


def config2():
    """POPE proper through pope_driver.run_pope: 768 questions (128 images x 6), use_dd_unk, answers of <= 2 tokens, + the none / unk prior passes,
    answers file fields and scorers (llava_calibrate.py:130-219)."""
    from bench import pope_prompts
    from llava_align_amd.engine import VddLlavaEngine
    from llava_align_amd.pope_driver import run_pope
    eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=True, lm_head_gain=4.0)
    ids, imgs = pope_prompts(128, seed=1234)
    images = {f"im{i}.jpg": imgs[6 * i] for i in range(128)}
    by_text = {f"q{i}": ids[i].tolist() for i in range(768)}
    qs = [{"question_id": i, "image": f"im{i // 6}.jpg", "text": f"q{i}", "label": ("yes", "no")[i % 2]} for i in range(768)]
    enc = lambda text, with_image: by_text[text] if with_image else [t for t in by_text[text] if t != -200]
    kw = dict(batch_questions=768, unk_token_id=0, eos_token_id=2, pad_token_id=0, max_new_tokens=2, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1,
              temperature=0.2, seed=1)
    for label, patch in (("shared system prompt in the prior passes", False), ("nothing shared in the prior passes", True)):
        if patch:
            eng._common_split = lambda rows: 0
        run_pope(eng, qs, enc, decode, lambda n: images[n], **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run_pope(eng, qs, enc, decode, lambda n: images[n], **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"config": 2, "what": "run_pope: main (use_dd_unk, 2 new tokens) + none + unk passes + label dicts + scorers, " + label,
                          "items": 768, "seconds": round(dt, 2), "items_per_s": round(768 / dt, 1), "n_answers": len(res["answers"])}), flush=True)


if "2" in which:
    config2()
if "4" in which:
    config4()
if "5" in which:
    config5()
