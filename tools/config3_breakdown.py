"""Where a decode step of BASELINE config #3 (LLaVA-1.5-13B, 90 questions x 3 branches = 270 rows, one image per question: no shared image
prefixes) goes: kernel totals of a few eager decode steps through torch.profiler (rocprofv3 segfaults on this program)."""
import os, sys, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from torch.profiler import profile, ProfilerActivity
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
eng = VddLlavaEngine("llava-1.5-13b", device=dev, use_graph=False)
kw = dict(images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, seed=1)
n_a, n_b = 8, 24
eng.generate(ids, max_new_tokens=n_b, **kw); torch.cuda.synchronize()
tot = {}
for n in (n_a, n_b):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        eng.generate(ids, max_new_tokens=n, **kw); torch.cuda.synchronize()
    acc = collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            acc[e.name.replace("(anonymous namespace)::", "").replace("void ", "")[:60]] += e.device_time
    tot[n] = acc
diff = {k: (tot[n_b][k] - tot[n_a].get(k, 0)) / (n_b - n_a) for k in tot[n_b]}
top = sorted(diff.items(), key=lambda kv: -kv[1])[:14]
print(json.dumps({"us_per_decode_step": round(sum(diff.values()), 1), "top": [(k, round(v, 1)) for k, v in top]}, indent=0))
