"""BASELINE.json config #3 shape on one GPU: LLaVA-1.5-13B, LLaVA-Bench-like open generation (90 questions, one image
each, text 80+-30 tokens), use_dd + use_dd_unk (3 branches), top-p 0.9, T=1, 256 new tokens.
python tools/config3_probe.py [--rank-of 8]: only the questions ShardPlan gives rank 0 of N (the per-rank batch of the 8-GPU run: 12 questions =
36 rows), with ops.UNEVEN_BLOCKS_TO_SLABS / ops.SKINNY_ROWS_MEASURED off and on."""
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
NQ = 90
if "--rank-of" in sys.argv:
    world = int(sys.argv[sys.argv.index("--rank-of") + 1])
    NQ = -(-90 // world)                       # ceil-chunks of whole images (shard.ShardPlan / get_chunk, MME/run_llava.py:32-40): rank 0's share
    ids, imgs = ids[:NQ], imgs[:NQ]
eng = VddLlavaEngine("llava-1.5-13b", device=dev, use_graph=True)
if os.environ.get("VDD_NO_GROUP"):
    eng.group_attention = False
kw = dict(images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=256, seed=1)
from llava_align_amd import ops
table = dict(ops.SKINNY_ROWS_MEASURED)
for policy in (("round-4 dispatch", False), ("13B dispatch", True)) if NQ < 90 else (("13B dispatch", True),):
    ops.UNEVEN_BLOCKS_TO_SLABS = policy[1]
    ops.SKINNY_ROWS_MEASURED.clear()
    if policy[1]:
        ops.SKINNY_ROWS_MEASURED.update(table)
    e = VddLlavaEngine(eng.cfg, weights=eng.w, device=dev, use_graph=True)
    e.group_attention = eng.group_attention
    e.generate(ids, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = e.generate(ids, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"model": "llava-1.5-13b", "dispatch": policy[0], "weights_GB": round(eng.w.nbytes() / 1e9, 1), "questions": NQ, "branches": 3, "new_tokens": 256,
                      "total_s": round(dt, 2), "tokens_per_s": round(NQ * 256 / dt, 1), "stats": out.stats,
                      "distinct_tokens_q0": int(out.tokens[0].unique().numel())}), flush=True)
    del e
