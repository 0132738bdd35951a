"""rmsnorm with a split-K slab delta (the consumer of ops.gemm_slabs) against rmsnorm with a bf16 delta, isolated, per row count and S."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from llava_align_amd import ops
dev = "cuda"
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 2)
for d in (4096, 5120):
    w = torch.ones(d, device=dev, dtype=torch.bfloat16)
    for M in (18, 36, 64, 128, 256):
        x = torch.randn(M, d, device=dev).bfloat16(); dl = torch.randn(M, d, device=dev).bfloat16(); out = torch.empty_like(x)
        rec = {"d": d, "rows": M, "bf16_delta_us": t(lambda: ops.rmsnorm(x, w, 1e-5, delta=dl, resid_out=out))}
        for S in (4, 8, 16, 27):
            sl = torch.randn(S, M, d, device=dev)
            rec[f"slabs_{S}_us"] = t(lambda: ops.rmsnorm(x, w, 1e-5, delta=sl, resid_out=out))
        print(json.dumps(rec), flush=True)
