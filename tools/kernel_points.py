"""The bench's fused-kernel roofline points on their own (B, V, beta, scores): python tools/kernel_points.py [hot]"""
import sys, json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, bench
dev = torch.device("cuda:0")
pts = [dict(B=4096, V=32000)] * 3 if sys.argv[1:] == ["hot"] else [dict(B=4096, V=32000), dict(B=1024, V=151936, iters=50), dict(B=1024, V=151936, scores=False, iters=50),
                                                                  dict(B=1024, V=151936, beta=1e-6, iters=50), dict(B=4096, V=32000, beta=1e-6, iters=50)]
for kw in pts:
    r = bench.kernel_point(dev, **kw)
    print(json.dumps({k: r[k] for k in ("launch_us", "frac", "frac_traffic", "survivors_per_row")} | {"shape": r["shape"]}))
