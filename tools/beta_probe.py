"""Fused sampling kernel vs the number of plausibility-mask survivors (beta), with the contrast row (a) a separate tensor and (b) aliased
to v (its chunks are then TLB- and L2-hot): separates the cost of the scattered contrast reads from the cost of the tail."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import llava_align_amd as L
dev = torch.device("cuda:0")
B, V = 4096, 32000
g = torch.Generator(device=dev).manual_seed(0)
v = (torch.randn(B, V, device=dev, generator=g) * 4).to(torch.bfloat16)
v[torch.arange(B, device=dev), torch.randint(0, V, (B,), device=dev, generator=g)] = 25.0
c = (v.float() + torch.randn(B, V, device=dev, generator=g) * 1.5).to(torch.bfloat16)
sc = torch.empty_like(v); toks = torch.empty(B, dtype=torch.long, device=dev)
W = L.WarpSpec(temperature=0.2)


def t(cc, beta, iters=50):
    run = lambda i: L.contrast_sample(v, cc, None, alpha=1.0, beta=beta, warp=W, out_tokens=toks, out_scores=sc, seed=0, offset=i)
    for i in range(5): run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters): run(i)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)


for beta in (0.1, 1e-4, 1e-5, 3e-6, 1e-6, 1e-7, 1e-9):
    print(json.dumps({"beta": beta, "separate_c_us": t(c, beta), "c_aliased_to_v_us": t(v, beta)}), flush=True)
