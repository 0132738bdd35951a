import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from bench import pope_prompts
from llava_align_amd.engine import VddLlavaEngine
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=True, lm_head_gain=4.0)
ids, imgs = pope_prompts(128, seed=1234)
none = [i[i != -200] for i in ids]
for share in (True, False):
    for _ in range(2):
        eng.generate(none, max_new_tokens=1, n_top=10, temperature=0.2, share_prefix=share)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = eng.generate(none, max_new_tokens=1, n_top=10, temperature=0.2, share_prefix=share)
    torch.cuda.synchronize(); print(json.dumps({"share": share, "s": round(time.perf_counter() - t0, 3), "prefill_tokens": o.stats["prefill_tokens"]}), flush=True)
