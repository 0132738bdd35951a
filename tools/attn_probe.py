"""Grouped decode attention at the bench shape (64 image groups x 6 rows + one 384-row image-free group) for several
prefix chunks-per-item settings; total time of prefix pass + own split-KV pass + combine per call."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
H = Hkv = 32; D = 128
G, PER, PL, UPL, OWN = 64, 6, 611, 36, int(sys.argv[1]) if len(sys.argv) > 1 else 57
Q = G * PER
T_OWN, T_PRE = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 640)
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
ko, vo = bf(2 * Q, Hkv, T_OWN, D), bf(2 * Q, Hkv, T_OWN, D)
kp, vp = bf(G + 1, Hkv, T_PRE, D), bf(G + 1, Hkv, T_PRE, D)
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
M = len(rows)
q = bf(M, H * D)
rt = torch.tensor(rows, dtype=torch.int32, device=dev)
gt = torch.tensor(groups, dtype=torch.int32, device=dev)
mt = torch.tensor(members, dtype=torch.int32, device=dev)
ws = ops.attention_workspace(M, H, D, T_PRE + T_OWN, dev)


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


import os as _os
SCALE = float(_os.environ["ATTN_SCALE"]) if "ATTN_SCALE" in _os.environ else None
ref = None
for rep in range(2):
    for cpi in (2, 4, 10):
        it = ops.prefix_work_items(groups, cpi)
        itt = torch.tensor(it, dtype=torch.int32, device=dev)
        f = lambda: ops.decode_attention_grouped(q, ko, vo, kp, vp, rt, gt, mt, itt, len(it), H, Hkv, D, PL, OWN, workspace=ws,
                                                 prefix_frag=pf, chunks_per_item=cpi, scale=SCALE)
        out = f()
        if ref is None:
            ref = out
        err = (out.float() - ref.float()).abs().max().item()
        print(json.dumps(dict(cpi=cpi, items=len(it), us=round(timeit(f), 1), max_diff_vs_cpi1=round(err, 4))), flush=True)
