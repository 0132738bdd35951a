"""Timing experiments on the 256x256 GEMM (config 11: no LDS-DMA in the K loop, 12: no waits/barriers; both compute garbage)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gemm_probe3 import gemm, timeit, dev
g = torch.Generator(device=dev).manual_seed(1)
for tag, M, N, K in [("prefill.qkv", 39140, 12288, 4096), ("prefill.wo", 39140, 4096, 4096), ("sq8k", 8192, 8192, 8192), ("decode768.qkv", 768, 12288, 4096)]:
    ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(3)]
    xs = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(2)]
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    rec = dict(tag=tag)
    t = timeit(lambda i: torch.matmul(xs[i & 1], ws[i].t(), out=y), 3); rec["blaslt_TF"] = round(2.0 * M * N * K / t / 1e6)
    for cfg in (1, 15, 11, 12, 13, 14):
        t = timeit(lambda i: gemm(xs[i & 1], ws[i], "none", cfg=cfg, out=y), 3)
        rec[f"cfg{cfg}_TF"] = round(2.0 * M * N * K / t / 1e6)
    print(json.dumps(rec), flush=True)
