import sys, time, torch, collections
sys.path.insert(0, "/root/repo")
from llava_align_amd import ops
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
eng = VddLlavaEngine("llava-1.5-7b", device="cuda:0", use_graph=True)
ids, imgs = pope_prompts(1, per_img=1, seed=99)
kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=1, seed=3)
for _ in range(3): eng.generate(ids, **kw)
torch.cuda.synchronize()
lib = ops._lib_ready()
T = collections.defaultdict(list)
def wrap(name):
    fn = getattr(lib, name)
    def w(*a):
        t0 = time.perf_counter(); r = fn(*a); T[name].append(time.perf_counter() - t0); return r
    return w
class Proxy:
    def __getattr__(self, n):
        return wrap(n) if n.startswith("vdd_") else getattr(lib, n)
px = Proxy()
ops._lib_ready = lambda: px
t0 = time.perf_counter(); eng.generate(ids, **kw); torch.cuda.synchronize(); print("wall ms", (time.perf_counter() - t0) * 1e3)
for n, v in sorted(T.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v); print(f"{n:34s} n={len(v):4d} total {sum(v)*1e3:7.2f} ms  median {v2[len(v2)//2]*1e6:7.1f} us  max {v2[-1]*1e6:8.1f} us")
