"""Tokens/s of generate() (ViT + both-branch prefill + 64 new tokens, use_dd_unk) against the number of questions in flight."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-7b", device=dev, use_graph=True)
for n_img, per in ((1, 1), (1, 3), (1, 6), (2, 6), (4, 6), (8, 6), (16, 6), (32, 6), (64, 6), (128, 6)):
    ids, imgs = pope_prompts(n_img, per_img=per, seed=5)
    imgs = [im.to(dev) for im in {id(i): i for i in imgs}.values()] if False else imgs
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=64, seed=3)
    for _ in range(2): eng.generate(ids, **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); eng.generate(ids, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = sorted(ts)[1]
    print(json.dumps({"questions": n_img * per, "rows": 2 * n_img * per, "tokens_per_s": round(n_img * per * 64 / t, 1), "seconds": round(t, 3)}), flush=True)
