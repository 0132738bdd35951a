"""What the GEMM tuner picks at the decode shapes (incl. column-split plans) and what it costs."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
for M in ([int(a) for a in sys.argv[1:]] or (1536, 768)):
    for name, N, K in (("qkv", 12288, 4096), ("wo", 4096, 4096), ("wd", 4096, 11008), ("lm_head", 32000, 4096)):
        ws = [bf(N, K) * 0.02 for _ in range(3)]
        x = bf(M, K); y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(x, ws[0], out=y)                   # tunes
        choice = ops._gemm_choice[(M, N, K, ops.EPI_NONE, ops.GEMM_BATCH_INVARIANT)]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): ops.gemm(x, ws[i % 3], out=y)
        e1.record(); torch.cuda.synchronize()
        print(json.dumps(dict(M=M, name=name, choice=choice, us=round(e0.elapsed_time(e1) / 30 * 1e3, 1))), flush=True)
