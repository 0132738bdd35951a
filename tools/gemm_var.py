"""A/B of schedule variants of the 256x256 GEMM (config 9: s_setprio around MFMA runs, 10: LDS-DMA spread over two k-steps, 11: both)
at prefill / decode shapes, several interleaved rounds."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_probe3 as G
from gemm_probe3 import gemm, timeit, dev
g = torch.Generator(device=dev).manual_seed(1)
for tag, M, N, K in [("prefill.qkv", 39140, 12288, 4096), ("prefill.wd", 39140, 4096, 11008), ("sq8k", 8192, 8192, 8192), ("decode768.wgu", 768, 22016, 4096), ("decode768.qkv", 768, 12288, 4096)]:
    n_rot = 3
    ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(n_rot)]
    xs = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(2)]
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    r0 = gemm(xs[0], ws[0], cfg=1).clone()
    rec = dict(tag=tag)
    for cfg in (9, 10, 11):
        assert torch.equal(gemm(xs[0], ws[0], cfg=cfg), r0), cfg
    res = {c: [] for c in ("blaslt", 1, 9, 10, 11)}
    for rnd in range(3):
        res["blaslt"].append(timeit(lambda i: torch.matmul(xs[i & 1], ws[i].t(), out=y), n_rot, iters=10))
        for cfg in (1, 9, 10, 11):
            res[cfg].append(timeit(lambda i: gemm(xs[i & 1], ws[i], "none", cfg=cfg, out=y), n_rot, iters=10))
    for k, v in res.items():
        rec[str(k)] = round(2.0 * M * N * K / min(v) / 1e6)
    print(json.dumps(rec), flush=True)
