"""Ungrouped decode attention (ops.decode_attention: split-KV waves + combine) at the shape of BASELINE config #3 - every question its own
image, so no prefix is shared: rows = questions x 3 branches, the image branch with a 611-key prefix + own keys, the other two with a
36-key prefix.  us per call and bytes of K / V read per second."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd import ops
dev = "cuda:0"
H = Hkv = int(os.environ.get("HEADS", 32)); D = 128
PL, UPL, OWN, T_OWN, T_PRE = 611, 36, 80, 128, 640
bf = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
for Q in (11, 45, 64, 90):
    M = 3 * Q
    ko, vo = bf(M, Hkv, T_OWN, D), bf(M, Hkv, T_OWN, D)
    kp, vp = bf(Q + 1, Hkv, T_PRE, D), bf(Q + 1, Hkv, T_PRE, D)
    rows = []
    for qi in range(Q):
        rows += [[3 * qi, PL + OWN, qi, PL], [3 * qi + 1, UPL + OWN, Q, UPL], [3 * qi + 2, UPL + OWN, Q, UPL]]
    rt = torch.tensor(rows, dtype=torch.int32, device=dev)
    q = bf(M, H * D)
    ws = ops.attention_workspace(M, H, D, T_PRE + T_OWN, dev)
    f = lambda: ops.decode_attention(q, ko, vo, rt, H, Hkv, D, k_prefix=kp, v_prefix=vp, max_len=T_PRE + T_OWN, workspace=ws)
    for _ in range(5): out = f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    kv_bytes = sum(r[1] for r in rows) * Hkv * D * 2 * 2
    # fp32 reference of row 0 (image branch) and row 1
    errs = []
    for r in (0, 1, M - 1):
        slot, ln, ps, pl = rows[r]
        K = torch.cat([kp[ps, :, :pl], ko[slot, :, :ln - pl]], 1).float(); V = torch.cat([vp[ps, :, :pl], vo[slot, :, :ln - pl]], 1).float()
        qq = q[r].view(H, D).float()
        p = torch.softmax(torch.einsum("hd,htd->ht", qq, K) / D ** 0.5, -1)
        ref = torch.einsum("ht,htd->hd", p, V).reshape(-1)
        errs.append((out[r].float() - ref).abs().max().item())
    print(json.dumps({"questions": Q, "rows": M, "us": round(us, 1), "kv_GB": round(kv_bytes / 1e9, 3), "TB_per_s": round(kv_bytes / us / 1e6, 2), "max_err": round(max(errs), 4)}), flush=True)
