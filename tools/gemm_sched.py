"""Schedules of the persistent GEMM at the decode shapes: hybrid (0), data-parallel only (+16), stream-K only (+32)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_probe3 as G
from gemm_probe3 import gemm, timeit, dev
assert G.check()
g = torch.Generator(device=dev).manual_seed(1)
lm = [("qkv", 12288, 4096), ("wo", 4096, 4096), ("wgu", 22016, 4096), ("wd", 4096, 11008), ("lm_head", 32000, 4096)]
for M in ([int(a) for a in sys.argv[1:]] or (768, 1536, 384, 39140)):
    for n, N, K in lm:
        n_rot = max(2, min(8, int(600e6 // (N * K * 2)) + 1))
        ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(n_rot)]
        xs = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(2)]
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        rec = dict(tag=f"decode{M}.{n}")
        rec["blaslt"] = round(timeit(lambda i: torch.matmul(xs[i & 1], ws[i].t(), out=y), n_rot), 1)
        for cfg in (1, 2, 3, 4, 5, 6, 7):
            for sched in (0, 1, 2):
                rec[f"c{cfg}s{sched}"] = round(timeit(lambda i: gemm(xs[i & 1], ws[i], "none", cfg=cfg + 16 * sched, out=y), n_rot), 1)
        print(json.dumps(rec), flush=True)
        del ws, xs, y
