"""The same GEMM launch on random and on all-zero weights, bf16 and fp16 (identical instruction stream, different switching activity):
what the power-managed clock costs the MFMA loop on real data.  python tools/gemm_data_dependence_probe.py -> profiles/r04_gemm_data_dependence.jsonl"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
def t(M, N, K, cfg, dt, epi=0, scale=0.02, NC=4, iters=12):
    W = [(torch.randn(N * (2 if epi == 4 else 1), K, device=dev) * scale).to(dt) for _ in range(NC)]
    x = torch.randn(M, K, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    for i in range(NC): ops.gemm(x, W[i], config=cfg, out=out, epi=epi)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): ops.gemm(x, W[i % NC], config=cfg, out=out, epi=epi)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    return [round(us, 1), round(2.0 * M * N * (2 if epi == 4 else 1) * K / us / 1e9, 3)]
for rep in range(2):
    for (M, N, K, epi) in ((39140, 12288, 4096, 0), (39140, 11008, 4096, 4), (39140, 4096, 11008, 0), (1536, 12288, 4096, 0)):
        r = {"M": M, "N": N, "K": K, "epi": epi}
        for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
            r[name] = t(M, N, K, 1 if M > 4096 else 4 + 32, dt, epi)
        r["fp16_zero_w"] = t(M, N, K, 1 if M > 4096 else 4 + 32, torch.float16, epi, scale=0.0)
        r["bf16_zero_w"] = t(M, N, K, 1 if M > 4096 else 4 + 32, torch.bfloat16, epi, scale=0.0)
        print(json.dumps(r), flush=True)
