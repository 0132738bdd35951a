"""Where the host time of a ONE-question prefill goes (cProfile over generate(max_new_tokens=1))."""
import os, sys, cProfile, pstats, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
eng = VddLlavaEngine("llava-1.5-7b", device="cuda:0", use_graph=True)
ids, imgs = pope_prompts(1, per_img=1, seed=99)
kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=1, seed=3)
for _ in range(3):
    eng.generate(ids, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    eng.generate(ids, **kw)
torch.cuda.synchronize()
print("prefill+1 token wall ms:", round((time.perf_counter() - t0) / 5 * 1e3, 2))
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    eng.generate(ids, **kw)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:4500])
