"""Run-to-run determinism of the shipped engine on the bench batch: N fresh configurations (captured steps rebuilt) x 4 generate() calls with one
seed; prints how many calls produced exactly the first call's tokens.  (Round 5: the RoPE-inside-attention experiment failed this probe
3 - 4 times per 10 configurations while the shipped path stayed clean - profiles/r05_rope_fusion_determinism.txt.)
  python tools/determinism_soak.py [n_configs] [bf16|fp16]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
dev = torch.device("cuda:0")
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else torch.bfloat16
eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=True, lm_head_gain=4.0, dtype=dt)
few = int(os.environ.get("FEW_QUESTIONS", "0"))          # > 0: that many questions, one image each, three branches (the one-launch attention band)
ids, imgs = pope_prompts(few, per_img=1, seed=1234) if few else pope_prompts(128, seed=1234)
on_dev = {}
imgs = [on_dev.setdefault(id(im), im.to(dev).to(dt)) for im in imgs]
kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=1, n_top=10, max_new_tokens=64)
if few:
    kw.update(use_dd=True, max_new_tokens=256)
if os.environ.get("FUSE_ROPE"):
    eng.lm.fuse_rope = True          # (experiment builds only: RoPE + KV write inside the grouped attention)
first, clean, total = None, 0, 0
for cfg in range(n_cfg):
    eng._graphs = {}
    for _ in range(4):
        t = eng.generate(ids, **kw).tokens
        first = t.clone() if first is None else first
        clean += int(torch.equal(t, first)); total += 1
print(json.dumps({"dtype": str(dt), "questions": len(ids), "rows": len(ids) * (3 if few else 2), "configurations": n_cfg, "calls": total, "calls_equal_to_the_first": clean}), flush=True)
