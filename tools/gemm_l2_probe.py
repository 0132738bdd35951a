"""Decode-shape GEMMs under rocprofv3 --pmc FETCH_SIZE / TCC_HIT_sum TCC_MISS_sum: L2 behaviour per (tile config, schedule).
Launch order = the order printed by the host side (argv: M)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
for N, K in ((12288, 4096), (4096, 11008)):
    x, w = bf(M, K), bf(N, K) * 0.02
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for cfg in (1, 4):
        for sch in (0, 1, 2):
            ops.gemm(x, w, out=y, config=cfg + 16 * sch)
            print(f"N={N} K={K} cfg={cfg} sched={sch}", flush=True)
    torch.cuda.synchronize()
