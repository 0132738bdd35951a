"""Does a weight matrix that was just read sit in the Infinity Cache for the next kernel?  One-question decode (2 rows) is a chain of
weight-streaming GEMVs with HBM-idle phases between them (attention, RMSNorm, launch gaps): if a re-read is much faster than a
first read, a prefetcher on a second stream could fill those phases."""
import sys, torch
sys.path.insert(0, "/root/repo")
from llava_align_amd import ops
dev = "cuda"
x = torch.randn(2, 4096, device=dev).to(torch.bfloat16)
x2 = torch.randn(2, 11008, device=dev).to(torch.bfloat16)
def t(fn, n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(3): fn(i)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for name, N, K, xx in (("qkv", 12288, 4096, x), ("o", 4096, 4096, x), ("down", 4096, 11008, x2), ("gate/up + SwiGLU", 22016, 4096, x)):
    ws = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(max(2, int(1.2e9 / (N * K * 2))))]
    f = ops.swiglu_linear if N == 22016 else ops.linear
    cold = t(lambda i: f(xx, ws[i % len(ws)]), 64)
    hot = t(lambda i: f(xx, ws[0]), 64)
    print(f"{name}: {N*K*2/1e6:.0f} MB  cold {cold:.1f} us ({N*K*2/cold/1e6:.2f} TB/s)  hot {hot:.1f} us ({N*K*2/hot/1e6:.2f} TB/s)", flush=True)
