"""Tokens/s of generate() for a few questions in flight (1 - 8 images x 6 questions = 12 - 96 decode rows), with the per-projection
switch between the weight-streaming kernels and the MFMA GEMM (ops.skinny_rows) and with the old fixed switch at 8 rows."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd.engine import VddLlavaEngine
from llava_align_amd import ops
from bench import pope_prompts
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-7b", device=dev, use_graph=True)
new_rule = ops.skinny_rows
for n_img, per in ((1, 1), (1, 3), (1, 6), (2, 6), (4, 6), (8, 6)):
    ids, imgs = pope_prompts(n_img, per_img=per, seed=5)
    rec = {"questions": n_img * per, "rows": 2 * n_img * per}
    for name, rule in (("fixed8", lambda N, K: 8), ("per_projection", new_rule)):
        ops.skinny_rows = rule
        e = VddLlavaEngine(eng.cfg, weights=eng.w, device=dev, use_graph=True)         # fresh graphs: the rule is baked in at capture
        kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=64, seed=3)
        for _ in range(2): e.generate(ids, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); e.generate(ids, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        rec[name] = round(n_img * per * 64 / sorted(ts)[1], 1)
        del e
    print(json.dumps(rec), flush=True)
