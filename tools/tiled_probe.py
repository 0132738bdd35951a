"""Row-major vs pre-tiled weight layout for the skinny (M <= 8) weight-streaming kernels."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"


def timeit(fn, iters=200, warm=20):
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


M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
# rotate over several weight copies so that nothing is served from the 256 MB MALL
for name, N, K, ns in (("qkv", 12288, 4096, 1), ("wo", 4096, 4096, 4), ("down", 4096, 11008, 2), ("lm_head", 32000, 4096, 1)):
    ws = [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(6)]
    tw = [ops.TiledWeight(w) for w in ws]
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    a = ops.skinny_gemm(x, ws[0], n_split=ns, slabs=ns > 1)
    b = ops.skinny_gemm(x, tw[0], n_split=ns, slabs=ns > 1)
    assert torch.equal(a, b), name
    i = [0]
    def f(wl):
        i[0] = (i[0] + 1) % 6
        ops.skinny_gemm(x, wl[i[0]], n_split=ns, slabs=ns > 1)
    t0, t1 = timeit(lambda: f(ws)), timeit(lambda: f(tw))
    print(json.dumps(dict(name=name, M=M, N=N, K=K, split=ns, rowmajor_us=round(t0, 1), tiled_us=round(t1, 1),
                          rowmajor_TBs=round(N * K * 2 / t0 / 1e6, 2), tiled_TBs=round(N * K * 2 / t1 / 1e6, 2))), flush=True)
F, K = 11008, 4096
ws = [torch.randn(2 * F, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(6)]
tw = [ops.TiledWeight(w, swiglu_pairs=True) for w in ws]
x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
assert torch.equal(ops.swiglu_linear(x, ws[0]), ops.swiglu_linear(x, tw[0]))
i = [0]
def f(wl):
    i[0] = (i[0] + 1) % 6
    ops.swiglu_linear(x, wl[i[0]])
t0, t1 = timeit(lambda: f(ws)), timeit(lambda: f(tw))
print(json.dumps(dict(name="gate_up+swiglu", M=M, rowmajor_us=round(t0, 1), tiled_us=round(t1, 1), rowmajor_TBs=round(2 * F * K * 2 / t0 / 1e6, 2),
                      tiled_TBs=round(2 * F * K * 2 / t1 / 1e6, 2))))
