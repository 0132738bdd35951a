"""Where the weight-streaming kernels stop winning against the row-batched MFMA GEMM (ops.SKINNY_MAX_M): per-layer projection time
of LLaVA-1.5-7B at M rows, weights rotated through > 600 MB."""
import sys, json, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llava_align_amd import ops
dev = "cuda"
def t(fn, n=24):
    for i in range(4): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("wgu", 22016, 4096), ("down", 4096, 11008)]
W = {n: [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(max(2, int(6.5e8 / (N * K * 2))))] for n, N, K in shapes}
for M in (8, 16, 17, 24, 32, 40, 48, 64, 96, 128):
    rec = {"M": M}
    tot_s, tot_g = 0.0, 0.0
    for n, N, K in shapes:
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        ws = W[n]
        if n == "wgu":
            g = t(lambda i: ops.gemm(x, ws[i % len(ws)], epi=ops.EPI_SWIGLU))
            if M <= 64:
                s = t(lambda i: ops.swiglu_linear(x, ws[i % len(ws)]))              # the fused weight-streaming kernels (<= 16 rows: 8 features per block, 17 - 64: 16)
            else: s = float("nan")
        else:
            g = t(lambda i: ops.gemm(x, ws[i % len(ws)]))
            s = t(lambda i: ops.skinny_gemm(x, ws[i % len(ws)])) if M <= 64 else float("nan")
        rec[n] = [round(s, 1), round(g, 1)]
        tot_s += s; tot_g += g
    rec["layer_skinny_us"], rec["layer_gemm_us"] = round(tot_s, 1), round(tot_g, 1)
    # d-wide projection + the RMSNorm that consumes it: finished product vs fp32 split-K slabs added by the norm
    for n, N, K in (("o", 4096, 4096), ("down", 4096, 11008)):
        x = torch.randn(M, K, device=dev).to(torch.bfloat16)
        res, lnw = torch.randn(M, N, device=dev).to(torch.bfloat16), torch.ones(N, device=dev, dtype=torch.bfloat16)
        ws = W[n]
        S = ops.slab_splits(M, N, K)
        a = t(lambda i: ops.rmsnorm(res, lnw, 1e-5, delta=ops.linear(x, ws[i % len(ws)]), resid_out=res))
        b = t(lambda i: ops.rmsnorm(res, lnw, 1e-5, delta=ops.gemm_slabs(x, ws[i % len(ws)], S), resid_out=res)) if S else float("nan")
        rec[n + "+norm"] = [round(a, 1), round(b, 1), S]
    print(json.dumps(rec), flush=True)
