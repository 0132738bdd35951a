"""Where the beta -> 0 launch of the fused kernel spends its extra time (85 mask survivors per row instead of 1): the same B = 4096 launch
with the sampling tail off / the scores row off, at beta = 0.1 and 1e-6, and with the contrast row aliased to v (its chunks then L2-hot)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import llava_align_amd as L
dev = torch.device("cuda:0")
B, V, dtype = 4096, 32000, torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
v = (torch.randn(B, V, device=dev, generator=g) * 4).to(dtype)
v[torch.arange(B, device=dev), torch.randint(0, V, (B,), device=dev, generator=g)] = 25.0
c = (v.float() + torch.randn(B, V, device=dev, generator=g) * 1.5).to(dtype)
sc = torch.empty(B, V, dtype=dtype, device=dev)
toks = torch.empty(B, dtype=torch.long, device=dev)
W = L.WarpSpec(temperature=0.2)


def t(fn, iters=60):
    for i in range(8):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)


for beta in (0.1, 1e-6):
    kw = dict(alpha=1.0, beta=beta, warp=W, seed=0)
    out = L.contrast_sample(v, c, return_scores=True, **kw)
    nfin = torch.isfinite(out.scores.float()).sum(1)
    print(json.dumps({"beta": beta, "survivors_mean": round(float(nfin.float().mean()), 1), "survivors_max": int(nfin.max()),
                      "rows_65_to_128": int(((nfin > 64) & (nfin <= 128)).sum()), "rows_above_128": int((nfin > 128).sum()),
                      "full_us": t(lambda i: L.contrast_sample(v, c, out_tokens=toks, out_scores=sc, offset=i, **kw)),
                      "scores_no_sample_us": t(lambda i: L.contrast_sample(v, c, out_scores=sc, no_sample=True, offset=i, **kw)),
                      "sample_no_scores_us": t(lambda i: L.contrast_sample(v, c, out_tokens=toks, offset=i, **kw)),
                      "full_c_aliased_to_v_us": t(lambda i: L.contrast_sample(v, v, out_tokens=toks, out_scores=sc, offset=i, **kw))}), flush=True)
