import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
def t(M, N, K, cfg, NC=6, iters=36, skinny=False):
    W = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(NC)]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = (lambda w: ops.skinny_gemm(x, w, out=out)) if skinny else (lambda w: ops.gemm(x, w, config=cfg, out=out))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(NC): f(W[i])
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(iters): f(W[i % NC])
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(3): g.replay()
        e1.record(s); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / iters / 3, 1)
Ks = (256, 1024, 2048, 4096, 8192)
r = {"kind": "skinny 16 rows", "N": 4096}
for K in Ks: r[f"K{K}"] = t(16, 4096, K, 0, skinny=True)
print(json.dumps(r), flush=True)
for (M, N) in ((128, 4096), (128, 12288), (192, 12288), (1536, 4096)):
    for cfg in (8, 10, 7, 1):
        for sch in (1, 2):
            if cfg == 8 and M > 256: continue
            r = {"M": M, "N": N, "tile": cfg, "sched": sch}
            for K in Ks:
                r[f"K{K}"] = t(M, N, K, cfg + 16 * sch)
            print(json.dumps(r), flush=True)
