"""The d-wide projections (attention output, MLP down) of LLaVA-1.5-13B (d = 5120: 320 column blocks of 16 on 256 CUs) at 3 - 32 rows:
the weight-streaming kernel, the MFMA GEMM, its split-K slabs (what rmsnorm adds) and the fused residual + sums-of-squares form.
Record: profiles/r05_13b_d_wide_projections.jsonl."""
import sys, json, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llava_align_amd import ops
dev = "cuda"
def t(fn, n=40):
    for i in range(5): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 1)
d, F = 5120, 13824
Wo = [(torch.randn(d, d, device=dev) * 0.02).to(torch.bfloat16) for _ in range(16)]
Wd = [(torch.randn(d, F, device=dev) * 0.02).to(torch.bfloat16) for _ in range(6)]
for M in (3, 8, 12, 16, 17, 24, 32):
    x, xf, res = torch.randn(M, d, device=dev).to(torch.bfloat16), torch.randn(M, F, device=dev).to(torch.bfloat16), torch.randn(M, d, device=dev).to(torch.bfloat16)
    rec = {"d": d, "M": M, "o_skinny": t(lambda i: ops.skinny_gemm(x, Wo[i % 16])), "o_gemm": t(lambda i: ops.gemm(x, Wo[i % 16])),
           "down_skinny": t(lambda i: ops.skinny_gemm(xf, Wd[i % 6])), "down_gemm": t(lambda i: ops.gemm(xf, Wd[i % 6]))}
    S = ops.slab_splits(M, d, F)
    if S: rec["down_slabs"] = [t(lambda i: ops.gemm_slabs(xf, Wd[i % 6], S)), S]
    So = ops.slab_splits(M, d, d)
    if So: rec["o_slabs"] = [t(lambda i: ops.gemm_slabs(x, Wo[i % 16], So)), So]
    if M <= 16:
        rec["o_resid_ss"] = t(lambda i: ops.linear_resid_ss(x, Wo[i % 16], res)); rec["down_resid_ss"] = t(lambda i: ops.linear_resid_ss(xf, Wd[i % 6], res))
    print(json.dumps(rec), flush=True)
