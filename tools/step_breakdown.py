"""Where a decode step goes, per kernel (torch.profiler over eager steps; the difference of two generation lengths isolates the decode steps).
python tools/step_breakdown.py [7b|13b] <questions> [branches 2|3] [per_image]   e.g. 7b 32 2 6 -> the 64-row step of the step curve"""
import os, sys, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
model, nq = sys.argv[1], int(sys.argv[2])
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
per = int(sys.argv[4]) if len(sys.argv) > 4 else 6
eng = VddLlavaEngine(f"llava-1.5-{model}", device="cuda:0", use_graph=False)
ids, imgs = pope_prompts((nq + per - 1) // per, per_img=per, seed=5)
ids, imgs = ids[:nq], imgs[:nq]
kw = dict(images=imgs, use_dd_unk=True, use_dd=nb == 3, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=3)
n_a, n_b = 8, 24
eng.generate(ids, max_new_tokens=n_b, **kw); torch.cuda.synchronize()
tot = {}
for n in (n_a, n_b):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        eng.generate(ids, max_new_tokens=n, **kw); torch.cuda.synchronize()
    acc, cnt = collections.Counter(), collections.Counter()
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            k = e.name.replace("(anonymous namespace)::", "").replace("void ", "").replace("vdd_bf16::", "")[:48]
            acc[k] += e.device_time; cnt[k] += 1
    tot[n] = (acc, cnt)
L = eng.cfg.lm.n_layers
rows = []
for k in tot[n_b][0]:
    us = (tot[n_b][0][k] - tot[n_a][0].get(k, 0)) / (n_b - n_a)
    n = (tot[n_b][1][k] - tot[n_a][1].get(k, 0)) / (n_b - n_a)
    rows.append((us, n, k))
rows.sort(reverse=True)
print(json.dumps({"model": model, "rows": nb * nq, "kernel_us_per_step": round(sum(r[0] for r in rows), 1), "per_layer_us": round(sum(r[0] for r in rows) / L, 1)}))
for us, n, k in rows[:14]:
    print(f"{us:9.1f} us/step  {n:6.1f} launches/step  {us / max(n, 1):7.1f} us each  {k}")
