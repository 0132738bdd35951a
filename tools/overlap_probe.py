"""Can an MFMA-bound library GEMM phase and the HBM-bound grouped decode attention of ANOTHER half-batch overlap on two
streams?  Times a layer's GEMMs (M rows) and a layer's attention (M rows) serially on one stream and concurrently on two."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
H = Hkv = 32; D = 128
G = int(sys.argv[1]) if len(sys.argv) > 1 else 64          # image groups of 6 -> M = 12 G rows
PER, PL, UPL, OWN = 6, 611, 36, 57
Q = G * PER
M = 2 * Q
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
ko, vo = bf(2 * Q, Hkv, 128, D), bf(2 * Q, Hkv, 128, D)
kp, vp = bf(G + 1, Hkv, 640, D), bf(G + 1, Hkv, 640, D)
pf = torch.empty((vp.shape[0], vp.shape[1], 2 * vp.shape[2], vp.shape[3]), dtype=vp.dtype, device=dev)
ops.prefix_fragments(kp, vp, pf, torch.tensor([PL] * G + [UPL], dtype=torch.int32, device=dev))
rows, groups, members = [], [], []
for g in range(G):
    groups.append([len(members), PER, g, PL])
    for i in range(PER):
        members.append(len(rows)); rows.append([len(rows), PL + OWN, g, PL])
groups.append([len(members), Q, G, UPL])
for i in range(Q):
    members.append(len(rows)); rows.append([len(rows), UPL + OWN, G, UPL])
q = bf(M, H * D)
rt = torch.tensor(rows, dtype=torch.int32, device=dev); gt = torch.tensor(groups, dtype=torch.int32, device=dev)
mt = torch.tensor(members, dtype=torch.int32, device=dev)
cpi = ops.prefix_chunks_per_item(groups, H)
it = ops.prefix_work_items(groups, cpi); itt = torch.tensor(it, dtype=torch.int32, device=dev)
ws = ops.attention_workspace(M, H, D, 640 + 128, dev)
out = torch.empty_like(q)
W = [bf(n, k) * 0.02 for n, k in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008))]
X = [bf(M, 4096), bf(M, 4096), bf(M, 4096), bf(M, 11008)]
Y = [torch.empty(M, w.shape[0], device=dev, dtype=torch.bfloat16) for w in W]


def attn():
    ops.decode_attention_grouped(q, ko, vo, kp, vp, rt, gt, mt, itt, len(it), H, Hkv, D, PL, OWN, out=out, workspace=ws, prefix_frag=pf,
                                 chunks_per_item=cpi)


def gemms():
    for x, w, y in zip(X, W, Y):
        torch.matmul(x, w.t(), out=y)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 40


def run(mode):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if mode == "serial":
        for _ in range(N):
            gemms(); attn()
    elif mode == "gemm":
        for _ in range(N):
            gemms()
    elif mode == "attn":
        for _ in range(N):
            attn()
    else:
        s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s1):
            for _ in range(N):
                gemms()
        with torch.cuda.stream(s2):
            for _ in range(N):
                attn()
        torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / N * 1e3, 1)


for m in ("gemm", "attn", "serial", "overlap"):
    run(m)
print(json.dumps({"rows": M, **{m: run(m) for m in ("gemm", "attn", "serial", "overlap", "serial", "overlap")}}))
