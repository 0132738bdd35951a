"""Macro tile x schedule at the prefill sizes (the tuner stops at 4,096 rows; above it the engine uses 256 x 256, hybrid)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gemm_probe3 import gemm, timeit, dev
g = torch.Generator(device=dev).manual_seed(1)
for tag, M, N, K in [("suffix.qkv", 36864, 12288, 4096), ("suffix.wo", 36864, 4096, 4096), ("suffix.wd", 36864, 4096, 11008), ("prefix.qkv", 78208, 12288, 4096)]:
    n_rot = 3
    ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(n_rot)]
    x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    rec = dict(tag=tag)
    for cfg in (1, 4, 5):
        for sched in (0, 1):
            us = timeit(lambda i: gemm(x, ws[i], "none", cfg=cfg + 16 * sched, out=y), n_rot, iters=6, warm=1)
            rec[f"c{cfg}s{sched}"] = round(2.0 * M * N * K / us / 1e6)
    print(json.dumps(rec), flush=True)
    del ws, x, y
