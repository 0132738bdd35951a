"""Per-projection time of the tuned MFMA GEMM against the row count (LLaVA-1.5-7B decoder shapes, rotating cache-cold weights):
which (tile, schedule) the tuner settles on per 64-row bucket and what a layer's four projections cost at that M."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
NC = 5
SHAPES = [("qkv", 12288, 4096, ops.EPI_NONE), ("o", 4096, 4096, ops.EPI_NONE), ("gate_up", 22016, 4096, ops.EPI_SWIGLU), ("down", 4096, 11008, ops.EPI_NONE)]
W = {n: [torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(NC)] for n, N, K, _ in SHAPES}
ops._load_persisted(torch.device(dev)); ops._gemm_choice.clear(); ops._persist["path"] = None     # time the tuner afresh at every M
Ms = [int(a) for a in sys.argv[1:]] or [66, 96, 128, 160, 192, 224, 256, 320, 384, 512, 768]
for M in Ms:
    rec, tot = {"M": M}, 0.0
    for n, N, K, epi in SHAPES:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        ops.gemm(x, W[n][0], epi=epi)                                     # tunes this bucket at this M
        key = ops._gemm_key(M, N // 2 if epi == ops.EPI_SWIGLU else N, K, epi)
        cfg = ops._gemm_choice.get(key)
        for i in range(NC):
            ops.gemm(x, W[n][i], epi=epi)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(40):
            ops.gemm(x, W[n][i % NC], epi=epi)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40
        tot += us
        rec[n] = {"us": round(us, 1), "tile": cfg % 16 if cfg is not None else None, "sched": cfg // 16 if cfg is not None else None}
    rec["layer_us"] = round(tot, 1)
    rec["ns_per_row"] = round(tot * 1e3 / M, 1)
    print(json.dumps(rec), flush=True)
    ops._gemm_choice.clear()
