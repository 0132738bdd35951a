"""Per-launch times of the four weight-streaming projections of a few-row decoder layer (LLaVA-1.5-7B shapes by default), each
against ROTATING copies of its weight (the copies together exceed the 256-MiB Infinity Cache, so every launch streams from HBM), in
a captured HIP graph of the layer chain (qkv -> o -> gate/up -> down) so that launch gaps are the graph's, not Python's.
    python tools/skinny_probe.py [M] [d] [ffn]
Prints one JSON line: us per launch, GB/s per launch, and the chain's us per layer."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from llava_align_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
ffn = int(sys.argv[3]) if len(sys.argv) > 3 else 11008
dev = "cuda:0"
NCOPY = int(os.environ.get("NCOPY", "6"))          # 1: the weight stays in the Infinity Cache (what a launch costs when its W was prefetched)
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.02).bfloat16()
wqkv = [mk(3 * d, d) for _ in range(NCOPY)]
wo = [mk(d, d) for _ in range(NCOPY)]
wgu = [mk(2 * ffn, d) for _ in range(NCOPY)]
wd = [mk(d, ffn) for _ in range(NCOPY)]
ln = (torch.ones(d, device=dev) + 0.1 * torch.randn(d, device=dev, generator=g)).bfloat16()
h0 = mk(M, d) * 50
x_attn = mk(M, d) * 50
eps = 1e-5


def timed(fn, n=60):
    for i in range(NCOPY):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(5):
        s.record()
        for i in range(n):
            fn(i % NCOPY)
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3 / n)
    return sorted(ts)[2]


res = {"M": M, "d": d, "ffn": ffn}
h1, ss1 = ops.linear_resid_ss(x_attn, wo[0], h0)
act = ops.swiglu_linear_normed(h1, ss1, ln, eps, wgu[0])
cases = {
    "qkv_normed": (lambda i: ops.linear_normed(h1, ss1, ln, eps, wqkv[i]), 3 * d * d * 2),
    "o_resid_ss": (lambda i: ops.linear_resid_ss(x_attn, wo[i], h0), d * d * 2),
    "gate_up_normed": (lambda i: ops.swiglu_linear_normed(h1, ss1, ln, eps, wgu[i]), 2 * ffn * d * 2),
    "down_resid_ss": (lambda i: ops.linear_resid_ss(act, wd[i], h1), d * ffn * 2),
}
# eager loops carry Python launch overhead for the small ones: time each launch inside a graph of NCOPY * 4 launches instead
for name, (fn, nbytes) in cases.items():
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(NCOPY):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side):
        for rep in range(4):
            for i in range(NCOPY):
                fn(i)
    us = timed(lambda i: gr.replay(), n=10) / (4 * NCOPY)
    res[name] = {"us": round(us, 2), "GBps": round(nbytes / us / 1e3, 0)}
# the chain of one layer (no attention), graph of NCOPY layers
gr = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
torch.cuda.synchronize()
with torch.cuda.graph(gr, stream=side):
    h, ss = h1, ss1
    for i in range(NCOPY):
        q = ops.linear_normed(h, ss, ln, eps, wqkv[i])
        h, ss = ops.linear_resid_ss(q[:, :d], wo[i], h)
        a = ops.swiglu_linear_normed(h, ss, ln, eps, wgu[i])
        h, ss = ops.linear_resid_ss(a, wd[i], h)
res["chain_us_per_layer"] = round(timed(lambda i: gr.replay(), n=10) / NCOPY, 2)
res["weights_MB_per_layer"] = round((3 * d * d + d * d + 3 * ffn * d) * 2 / 1e6, 1)
res["floor_us_at_8TBps"] = round((3 * d * d + d * d + 3 * ffn * d) * 2 / 8e6, 1)
print(json.dumps(res))
