"""Prefill flash attention at the bench shapes: (a) 64 image prefixes of 611 tokens, (b) 768 suffixes of ~25 tokens behind a
shared prefix, (c) the ViT (64 x 577 tokens, 16 heads of 64, non-causal)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


H, D = 32, 128
res = {}
# (a) prefixes
G, T = 64, 611
kp, vp = bf(G + 1, H, 640, D), bf(G + 1, H, 640, D)
q = bf(G * T, H * D)
seqs = torch.tensor([[g * T, T, 0, g, 0, 0] for g in range(G)], dtype=torch.int32, device=dev)
us = timeit(lambda: ops.flash_attention(q, kp, vp, seqs, G, T, H, H, D, causal=True))
fl = G * H * 4 * T * T * D / 2
res["prefix 64x611 causal"] = dict(us=round(us, 1), TFs=round(fl / us / 1e6, 1))
# (b) suffixes: 384 main (prefix 611 shared by 6) + 384 image-free (prefix 36 shared by all)
ko, vo = bf(768, H, 128, D), bf(768, H, 128, D)
S = []
r = 0
for i in range(384):
    S.append([r, 25, 611, i, i // 6, 611]); r += 25
for i in range(384):
    S.append([r, 25, 36, 384 + i, 64, 36]); r += 25
q2 = bf(r, H * D)
seqs2 = torch.tensor(S, dtype=torch.int32, device=dev)
us = timeit(lambda: ops.flash_attention(q2, ko, vo, seqs2, len(S), 25, H, H, D, causal=True, k_prefix=kp, v_prefix=vp))
fl = sum(H * 4 * s[1] * (s[2] + s[1] / 2) * D for s in S)
res["suffix 768x25 behind prefix"] = dict(us=round(us, 1), TFs=round(fl / us / 1e6, 1))
packs = ops.flash_packs(S)
pk = torch.tensor(packs, dtype=torch.int32, device=dev)
us = timeit(lambda: ops.flash_attention_packed(q2, ko, vo, seqs2, pk, len(packs), H, H, D, k_prefix=kp, v_prefix=vp))
res["suffix packed (4 per block)"] = dict(us=round(us, 1), TFs=round(fl / us / 1e6, 1), packs=len(packs))
# (c) ViT
Hv, Dv, Tv, N = 16, 64, 577, 64
kc, vc = bf(N, Hv, 584, Dv), bf(N, Hv, 584, Dv)
q3 = bf(N * Tv, Hv * Dv)
seqs3 = torch.tensor([[n * Tv, Tv, 0, n, 0, 0] for n in range(N)], dtype=torch.int32, device=dev)
us = timeit(lambda: ops.flash_attention(q3, kc, vc, seqs3, N, Tv, Hv, Hv, Dv, causal=False))
res["vit 64x577 full"] = dict(us=round(us, 1), TFs=round(N * Hv * 4 * Tv * Tv * Dv / us / 1e6, 1))
print(json.dumps(res))
