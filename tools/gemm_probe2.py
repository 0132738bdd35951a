import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = {"qkv": (12288, 4096, 1), "o": (4096, 4096, 4), "gu": (22016, 4096, 1), "down": (4096, 11008, 4), "head": (32000, 4096, 1)}
# rotate through 8 copies of each weight so that nothing stays in the 256 MiB Infinity Cache (as in a real decode step)
Ws = {k: [(torch.randn(n, kk, device=dev) * 0.02).bfloat16() for _ in range(8)] for k, (n, kk, _) in shapes.items()}
for M in (12, 24, 48, 96, 128, 192, 256):
    line, tot_t, tot_m = [], 0, 0
    for name, (N, K, ns) in shapes.items():
        x = torch.randn(M, K, device=dev).bfloat16()
        it = [0]
        def f_t():
            it[0] += 1; torch.matmul(x, Ws[name][it[0] % 8].t())
        def f_m():
            it[0] += 1
            if ns > 1: ops.mid_gemm(x, Ws[name][it[0] % 8], n_split=ns, slabs=True)
            else: ops.mid_gemm(x, Ws[name][it[0] % 8])
        a, b = timeit(f_t), timeit(f_m)
        mult = 1 if name == "head" else 32
        tot_t += a * mult; tot_m += b * mult
        line.append(f"{name}: {a:.0f}/{b:.0f}")
    print(f"M={M}: per-step us  hipBLASLt={tot_t:.0f}  mid_gemm={tot_m:.0f}   " + "  ".join(line), flush=True)
