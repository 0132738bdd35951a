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
def timed(n_new):
    k2 = {**kw, "max_new_tokens": n_new}
    for _ in range(3):
        eng.generate(ids, **k2)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); eng.generate(ids, **k2); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[2]
if len(sys.argv) > 1 and sys.argv[1] == "--unfused":
    eng.lm.fuse_norms = False
t64, t32, t1 = timed(64), timed(32), timed(1)
print(json.dumps({"total_s_64": round(t64, 4), "prefill_plus_1_token_s": round(t1, 4), "ms_per_decode_step": round((t64 - t32) / 32 * 1e3, 3),
                  "tok_per_s": round(64 / t64, 1)}))
