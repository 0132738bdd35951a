"""BASELINE.json config #3 shape on one GPU: LLaVA-1.5-13B, LLaVA-Bench-like open generation (90 questions, one image
each, text 80+-30 tokens), use_dd + use_dd_unk (3 branches), top-p 0.9, T=1, 256 new tokens."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from llava_align_amd.engine import VddLlavaEngine

dev = "cuda:0"
rng = np.random.default_rng(5)
sys_tok = [1] + rng.integers(3, 32000, size=34).tolist()
ids, imgs = [], []
g = torch.Generator().manual_seed(3)
for q in range(90):
    n = int(np.clip(rng.normal(80, 30), 10, 170))
    ids.append(torch.tensor(sys_tok + [-200] + rng.integers(3, 32000, size=n).tolist()))
    imgs.append(torch.randn(3, 336, 336, generator=g))
eng = VddLlavaEngine("llava-1.5-13b", device=dev, use_graph=True)
if os.environ.get("VDD_NO_GROUP"):
    eng.group_attention = False
kw = dict(images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=256, seed=1)
eng.generate(ids, **kw); torch.cuda.synchronize()
t0 = time.perf_counter(); out = eng.generate(ids, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"model": "llava-1.5-13b", "weights_GB": round(eng.w.nbytes() / 1e9, 1), "questions": 90, "branches": 3, "new_tokens": 256,
                  "total_s": round(dt, 2), "tokens_per_s": round(90 * 256 / dt, 1), "stats": out.stats,
                  "distinct_tokens_q0": int(out.tokens[0].unique().numel())}))
