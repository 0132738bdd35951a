"""Where the host-image (PCIe-inclusive) path spends its time: stacking 16 pageable fp32 images into the pinned staging buffer, the
asynchronous upload, the ViT call."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
ims = [torch.randn(3, 336, 336) for _ in range(128)]
pin = torch.empty(16, 3, 336, 336, pin_memory=True)
dev = torch.empty(16, 3, 336, 336, device="cuda")
def t(fn, n=8):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("stack 16 -> pinned      %.2f ms" % t(lambda i=0: torch.stack(ims[16 * (i % 8):16 * (i % 8) + 16], out=pin)))
print("stack 16 -> pageable    %.2f ms" % t(lambda i=0: torch.stack(ims[16 * (i % 8):16 * (i % 8) + 16])))
print("16 x copy_ -> pinned    %.2f ms" % t(lambda i=0: [pin[j].copy_(ims[16 * (i % 8) + j]) for j in range(16)]))
print("H2D pinned 21.7 MB      %.2f ms" % t(lambda i=0: dev.copy_(pin, non_blocking=True)))
pg = torch.stack(ims[:16])
print("H2D pageable 21.7 MB    %.2f ms" % t(lambda i=0: dev.copy_(pg, non_blocking=True)))
torch.set_num_threads(8)
print("stack 16 -> pinned, 8 threads %.2f ms" % t(lambda i=0: torch.stack(ims[16 * (i % 8):16 * (i % 8) + 16], out=pin)))
