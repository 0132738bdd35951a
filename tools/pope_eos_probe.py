"""POPE-proper timing of the bench (768 questions, answers of 1-2 tokens then EOS): repeated, with the 2-token no-EOS call beside it."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import pope_prompts
from llava_align_amd.engine import VddLlavaEngine
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=True, lm_head_gain=4.0)
ids, host_imgs = pope_prompts(128, seed=1234, vocab=eng.cfg.lm.vocab, image=eng.cfg.vision.image)
on_dev = {}
imgs = [on_dev.setdefault(id(im), im.to(dev).to(torch.bfloat16)) for im in host_imgs]
kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=1, n_top=10)
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    out = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); out.append(round(time.perf_counter() - t0, 4))
    return out
o2 = eng.generate(ids, max_new_tokens=2, **kw)
eos = sorted(set(o2.tokens[:, 1].tolist()))
print(json.dumps({"two_tokens_no_eos": t(lambda: eng.generate(ids, max_new_tokens=2, **kw))}), flush=True)
print(json.dumps({"eos_64": t(lambda: eng.generate(ids, max_new_tokens=64, eos_token_id=eos, pad_token_id=0, sync_every=2, **kw))}), flush=True)
print(json.dumps({"eos_64_one_id": t(lambda: eng.generate(ids, max_new_tokens=64, eos_token_id=eos[:1], pad_token_id=0, sync_every=2, **kw))}), flush=True)
print(json.dumps({"one_token": t(lambda: eng.generate(ids, max_new_tokens=1, **kw))}), flush=True)
