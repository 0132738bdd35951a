"""Selected tile configs of the persistent GEMM vs hipBLASLt at LLaVA-1.5-7B shapes (weights rotated through > 256 MiB).
  python tools/gemm_cfg_probe.py "1,9" 1536 39140        -> one JSON line per (M, projection): us per launch and PF/s"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gemm_probe3 as G
from gemm_probe3 import gemm, timeit, dev, ref
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 9]
g = torch.Generator(device=dev).manual_seed(1)
# correctness of the configs under test first (all epilogues at an odd shape)
for cfg in cfgs:
    for epi in ("none", "bias", "bias_resid", "swiglu"):
        M, N, K = 777, 1280, 512
        x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(2 * N if epi == "swiglu" else N, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        b, r = torch.randn(N, device=dev, generator=g).to(torch.bfloat16), torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        for sched in (0, 1, 2):
            try:
                y = gemm(x, w, epi, bias=b, resid=r, cfg=cfg + 16 * sched)
            except AssertionError as e:
                print(json.dumps({"cfg": cfg, "epi": epi, "unsupported": str(e)[:60]})); break
            want = ref(x, w, epi, b, r)
            err = (y.float() - want).abs().max().item() / (want.abs().max().item() + 1e-6)
            assert err < 2e-2, (cfg, epi, sched, err)
print(json.dumps({"correct": cfgs}), flush=True)
lm = [("qkv", 12288, 4096, "none"), ("wo", 4096, 4096, "none"), ("wgu", 11008, 4096, "swiglu"), ("wd", 4096, 11008, "none")]
for M in ([int(a) for a in sys.argv[2:]] or (1536, 39140)):
    for n, N, K, epi in lm:
        Nw = 2 * N if epi == "swiglu" else N
        n_rot = max(2, min(8, int(600e6 // (Nw * K * 2)) + 1))
        ws = [(torch.randn(Nw, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(n_rot)]
        xs = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(2)]
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        flop = 2.0 * M * Nw * K
        rec = dict(tag=f"M{M}.{n}")
        if epi == "none":
            t = timeit(lambda i: torch.matmul(xs[i & 1], ws[i].t(), out=y), n_rot)
            rec["blaslt"] = [round(t, 1), round(flop / t / 1e9, 3)]
        for cfg in cfgs:
            for sched in ((0,) if M > 4096 else (0, 1, 2)):
                t = timeit(lambda i: gemm(xs[i & 1], ws[i], epi, cfg=cfg + 16 * sched, out=y), n_rot)
                rec[f"c{cfg}s{sched}"] = [round(t, 1), round(flop / t / 1e9, 3)]
        print(json.dumps(rec), flush=True)
        del ws, xs, y
