"""Grouped decode attention at the bench shape (128 image groups x 6 rows + one 768-row image-free group, 57 own keys): rope_kv + the
two attention launches against the form with RoPE + the KV write inside them (ops.decode_attention_grouped(rope=...)).  us per call;
run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (the <128, true> / <128, false> instances)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
H = Hkv = 32; D = 128
G, PER, PL, UPL, OWN = 128, 6, 611, 36, 57
Q = G * PER
T_OWN, T_PRE = 96, 640
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
qkv = bf(M, 3 * H * D)
i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)
rt, gt, mt = i32(rows), i32(groups), i32(members)
pos, cpos, slot = i32([r[1] - 1 for r in rows]), i32([r[1] - r[3] - 1 for r in rows]), i32([r[0] for r in rows])
t = torch.arange(2048, dtype=torch.float32)[:, None] * (10000.0 ** (-torch.arange(0, D, 2, dtype=torch.float32) / D))[None]
cs = torch.stack([t.cos(), t.sin()], -1).contiguous().to(dev)
ws = ops.attention_workspace(M, H, D, T_PRE + T_OWN, dev)
cpi = ops.prefix_chunks_per_item(groups, H)
it = i32(ops.prefix_work_items(groups, cpi))
q_buf, out = torch.empty(M, H * D, dtype=torch.bfloat16, device=dev), torch.empty(M, H * D, dtype=torch.bfloat16, device=dev)


def timeit(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / iters * 1e3, 1)


def separate():
    q = ops.rope_kv_write(qkv, pos, slot, cs, ko, vo, H, Hkv, D, q_out=q_buf, cpos=cpos)
    return ops.decode_attention_grouped(q, ko, vo, kp, vp, rt, gt, mt, it, it.shape[0], H, Hkv, D, PL, T_OWN, workspace=ws, prefix_frag=pf,
                                        chunks_per_item=cpi, out=out)


def fused():
    return ops.decode_attention_grouped(qkv, ko, vo, kp, vp, rt, gt, mt, it, it.shape[0], H, Hkv, D, PL, T_OWN, workspace=ws, prefix_frag=pf,
                                        chunks_per_item=cpi, out=out, rope=(pos, cpos, slot, cs))


a = separate().clone(); b = fused().clone()
for rep in range(2):
    print(json.dumps({"rows": M, "cpi": cpi, "items": int(it.shape[0]), "rope_kv_plus_grouped_us": timeit(separate), "grouped_rope_us": timeit(fused),
                      "rope_kv_alone_us": timeit(lambda: ops.rope_kv_write(qkv, pos, slot, cs, ko, vo, H, Hkv, D, q_out=q_buf, cpos=cpos)),
                      "bit_equal": bool(torch.equal(a, b))}), flush=True)
