import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_probe3 as G
dev = G.dev
g = torch.Generator(device=dev).manual_seed(0)
for cfgx in (1 + 32, 1, 2, 3):
    for (M, N, K) in [(77, 136, 128), (300, 520, 128), (300, 520, 256), (768, 1024, 512), (768, 4096, 4096)]:
        x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        y = G.gemm(x, w, "none", cfg=cfgx)
        r = G.ref(x, w, "none", None, None)
        d = (y.float() - r).abs()
        tol = 0.02 * r.abs().max().item() + 1e-3
        badrows = (d.max(1).values > tol).nonzero().reshape(-1)
        badcols = (d.max(0).values > tol).nonzero().reshape(-1)
        print("cfg", cfgx, (M, N, K), "maxerr %.3g" % d.max().item(), "bad rows", badrows.numel(), badrows[:4].tolist(), badrows[-2:].tolist(), "bad cols", badcols.numel(), badcols[:4].tolist(), badcols[-2:].tolist(), flush=True)
