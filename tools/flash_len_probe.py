import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
H, D = 32, 128
for G, T, causal in ((64, 611, True), (64, 611, False), (32, 1280, True), (16, 2560, True), (16, 2560, False), (8, 5120, False)):
    tm = (T + 63) // 64 * 64
    kp, vp = bf(G, H, tm, D), bf(G, H, tm, D)
    q = bf(G * T, H * D)
    seqs = torch.tensor([[g * T, T, 0, g, 0, 0] for g in range(G)], dtype=torch.int32, device=dev)
    us = timeit(lambda: ops.flash_attention(q, kp, vp, seqs, G, T, H, H, D, causal=causal))
    fl = G * H * 4 * T * T * D / (2 if causal else 1)
    print(json.dumps(dict(G=G, T=T, causal=causal, us=round(us, 1), TFs=round(fl / us / 1e6, 1))), flush=True)
