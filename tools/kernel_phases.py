"""Where the fused sampling kernel's time goes: the same B=4096 launch with parts switched off by flags."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import llava_align_amd as L
dev = torch.device("cuda:0")
B, V, dtype = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, 32000, torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
v = (torch.randn(B, V, device=dev, generator=g) * 4).to(dtype)
v[torch.arange(B, device=dev), torch.randint(0, V, (B,), device=dev, generator=g)] = 25.0
c = (v.float() + torch.randn(B, V, device=dev, generator=g) * 1.5).to(dtype)
sc = torch.empty(B, V, dtype=dtype, device=dev)
toks = torch.empty(B, dtype=torch.long, device=dev)
W = L.WarpSpec(temperature=0.2)


def t(fn, iters=100):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)


kw = dict(alpha=1.0, beta=0.1, warp=W, seed=0)
print(json.dumps({
    "full (scores + sample)": t(lambda i: L.contrast_sample(v, c, out_tokens=toks, out_scores=sc, offset=i, **kw)),
    "scores, no sample": t(lambda i: L.contrast_sample(v, c, out_scores=sc, no_sample=True, offset=i, **kw)),
    "sample, no scores": t(lambda i: L.contrast_sample(v, c, out_tokens=toks, offset=i, **kw)),
    "passes A+B + top-1 only": t(lambda i: L.contrast_sample(v, c, no_sample=True, n_top=1, offset=i, **kw)),
    "argmax pick + scores": t(lambda i: L.contrast_sample(v, c, out_tokens=toks, out_scores=sc, pick_argmax=True, offset=i, **kw)),
    "plain (no c) + scores + sample": t(lambda i: L.contrast_sample(v, out_tokens=toks, out_scores=sc, offset=i, **kw)),
    "copy v->scores (torch)": t(lambda i: sc.copy_(v)),
}))
