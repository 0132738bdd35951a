"""generate_list against batch-after-batch generate() on a 360-question LLaVA-Bench-shaped list (bench.py `llava_bench_eos.list_of_360`), with the
admission phases' share.  Record: profiles/r06_list_of_360.jsonl"""
import sys, os, json, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from llava_align_amd.engine import VddLlavaEngine
dev = torch.device("cuda:0")
eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=True, lm_head_gain=4.0)
n_q = 90
ids4, imgs4 = bench.pope_prompts(4 * n_q, per_img=1, seed=778)
imgs4 = [im.to(dev).to(eng.dtype) for im in imgs4]
eos = sorted(set(np.random.default_rng(5).integers(3, 32000, size=250).tolist()))
kw = dict(use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=512, eos_token_id=eos, pad_token_id=0, seed=11, sync_every=8)
for am in (None, 4, 24):
    eng.generate_list(ids4, imgs4, in_flight=n_q, admit_min=am, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = eng.generate_list(ids4, imgs4, in_flight=n_q, admit_min=am, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"admit_min": am, "seconds": round(dt, 2), "tokens_per_s": round(o.stats["answer_tokens"] / dt, 1), **{k: v for k, v in o.stats.items()}}), flush=True)
