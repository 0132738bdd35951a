"""One library-GEMM shape launched a few times (for rocprofv3 --pmc passes): python tools/gemm_counters.py M N K [cfg]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gemm_probe3 import gemm, dev
M, N, K = (int(a) for a in sys.argv[1:4])
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else 1
g = torch.Generator(device=dev).manual_seed(1)
ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(4)]
x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for i in range(8):
    gemm(x, ws[i % 4], "none", cfg=cfg, out=y)
torch.cuda.synchronize()
