"""Workload for `rocprofv3 --pmc MfmaUtil`: the hand-written GEMM (ops.gemm, tuned config) at the decode (768 rows) and prefill (39,140 rows) sizes, the
MFMA prefix pass of the grouped decode attention, the skinny weight-streaming GEMVs and the prefill flash attention."""
import os, sys, runpy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
W = [bf(n, k) * 0.02 for n, k in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008))]
for M in (768, 39140):
    X = [bf(M, w.shape[1]) for w in W]
    for _ in range(3):
        for x, w in zip(X, W):
            ops.gemm(x, w)
    ops.swiglu_linear(X[0], W[2])
    del X
for M in (2,):
    for _ in range(3):
        ops.skinny_gemm(bf(M, 4096), W[0]); ops.swiglu_linear(bf(M, 4096), W[2]); ops.skinny_gemm(bf(M, 11008), W[3], n_split=2, slabs=True)
torch.cuda.synchronize()
sys.argv = ["attn_probe.py"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_probe.py"), run_name="__main__")
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "flash_probe.py"), run_name="__main__")
