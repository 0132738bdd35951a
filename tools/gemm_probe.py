import torch, time, sys
dev = "cuda:0"
torch.manual_seed(0)
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gu": (22016, 4096), "down": (4096, 11008), "head": (32000, 4096)}
Ws = {k: (torch.randn(n, kk, device=dev) * 0.02).bfloat16() for k, (n, kk) in shapes.items()}
for M in (16, 48, 96, 192, 256, 384, 768):
    tot = {"xWt": 0, "WxT": 0, "pad256": 0}
    line = []
    for name, (N, K) in shapes.items():
        W = Ws[name]
        x = torch.randn(M, K, device=dev).bfloat16()
        xp = torch.randn((M + 255) // 256 * 256, K, device=dev).bfloat16()
        xt = x.t().contiguous()
        a = timeit(lambda: torch.matmul(x, W.t()))
        b = timeit(lambda: torch.matmul(W, x.t()))
        c = timeit(lambda: torch.matmul(xp, W.t()))
        mult = 1 if name == "head" else 32
        tot["xWt"] += a * mult; tot["WxT"] += b * mult; tot["pad256"] += c * mult
        line.append(f"{name}: {a:.0f}/{b:.0f}/{c:.0f}")
    floor = sum(n * k * 2 * (1 if nm == "head" else 32) for nm, (n, k) in shapes.items()) / 5.5e12 * 1e6
    print(f"M={M}: per-step us  x@Wt={tot['xWt']:.0f}  W@xT={tot['WxT']:.0f}  padM256={tot['pad256']:.0f}  (stream floor {floor:.0f})   " + "  ".join(line), flush=True)
