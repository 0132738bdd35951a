"""Host time of one vdd_flash_attention launch on an idle GPU vs behind a queue of GEMM launches (one-question prefill shapes)."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from llava_align_amd import ops
dev = "cuda"
H, D, T = 32, 128, 640
q = torch.randn(611, H * D, device=dev).to(torch.bfloat16)
kc = torch.randn(2, H, T, D, device=dev).to(torch.bfloat16); vc = torch.randn_like(kc)
seqs = torch.tensor([[0, 611, 0, 0, 0, 0]], dtype=torch.int32, device=dev)
x = torch.randn(611, 4096, device=dev).to(torch.bfloat16); w = torch.randn(12288, 4096, device=dev).to(torch.bfloat16)
for _ in range(3): ops.flash_attention(q, kc, vc, seqs, 1, 611, H, H, D); ops.gemm(x, w)
torch.cuda.synchronize()
def host(fn, n=50):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ts.sort(); return ts[len(ts) // 2] * 1e6, ts[-1] * 1e6
print("flash alone: median %.1f us max %.1f us" % host(lambda: ops.flash_attention(q, kc, vc, seqs, 1, 611, H, H, D)))
print("gemm alone:  median %.1f us max %.1f us" % host(lambda: ops.gemm(x, w)))
def mix():
    for _ in range(4): ops.gemm(x, w)
    t0 = time.perf_counter(); ops.flash_attention(q, kc, vc, seqs, 1, 611, H, H, D); return time.perf_counter() - t0
ts = sorted(mix() for _ in range(30)); torch.cuda.synchronize()
print("flash after 4 gemms: median %.1f us max %.1f us" % (ts[15] * 1e6, ts[-1] * 1e6))
