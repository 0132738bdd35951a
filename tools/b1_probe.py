import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-7b", device=dev, use_graph=True)
ids, imgs = pope_prompts(1, per_img=1, seed=99)
kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=64, seed=3)
for _ in range(3):
    eng.generate(ids, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter(); eng.generate(ids, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
t1 = time.perf_counter(); eng.generate(ids, **{**kw, "max_new_tokens": 2}); torch.cuda.synchronize(); d2 = time.perf_counter() - t1
print(json.dumps({"total_s": round(dt, 4), "gen2_s": round(d2, 4), "ms_per_decode_step": round((dt - d2) / 62 * 1e3, 3), "tok_per_s": round(64 / dt, 1)}))
