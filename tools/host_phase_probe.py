"""Host-side time of each phase of a 384-question generate() (the GPU runs asynchronously behind the host)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import engine as E
from bench import pope_prompts
eng = E.VddLlavaEngine("llava-1.5-7b", device="cuda:0", use_graph=True)
ids, imgs = pope_prompts(int(sys.argv[1]) if len(sys.argv) > 1 else 128, seed=1234)
_dev = {}
imgs = [_dev.setdefault(id(im), im.to("cuda:0")) for im in imgs]          # the six questions of an image keep sharing ONE tensor
kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=64, seed=1)
T = collections.defaultdict(list)


def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); T[label].append((time.perf_counter() - t0) * 1e3); return r
    setattr(obj, name, g)


wrap(eng, "image_features", "image_features (ViT, 4 batches)")
wrap(eng, "_plan", "_plan"); wrap(eng, "_pack", "_pack"); wrap(eng.lm, "prefill", "lm.prefill (launch)"); wrap(eng.lm, "logits", "lm.logits")
wrap(E, "group_rows_by_prefix", "group_rows_by_prefix"); wrap(E.ops, "prefix_work_items", "prefix_work_items"); wrap(E, "h2d_int32", "h2d_int32")
wrap(torch, "stack", "torch.stack")
for i in range(5):
    if i == 2:
        T.clear()
    t0 = time.perf_counter(); eng.generate(ids, **kw); torch.cuda.synchronize(); print("generate", round((time.perf_counter() - t0) * 1e3, 1), "ms", flush=True)
for k, v in T.items():
    print(f"{k:34s} calls/generate {len(v) / 3:5.1f}  ms each {sum(v) / len(v):8.2f}  ms per generate {sum(v) / 3:8.1f}  max {max(v):7.1f}")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); eng.generate(ids, **kw); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
