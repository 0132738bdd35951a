"""Every (tile, schedule) candidate of the MFMA GEMM with guard bands around the output and the workspace: result against an fp32
matmul, sentinels intact, arrival counters back at zero.  Shape sets: decode (the 7B decoder projections at the given row counts),
vit (CLIP-L layer + projector at 16 x 577 rows, with their epilogues), 13b (LLaVA-13B projections), prefill (ragged 39,140 rows).
    python tools/gemm_guard_probe.py decode 1536 768 | vit | 13b 270 | prefill"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from llava_align_amd import ops
dev = "cuda:0"
E = ops
which = sys.argv[1] if len(sys.argv) > 1 else "decode"
Ms = [int(a) for a in sys.argv[2:]]
if which == "decode":
    Ms = Ms or [1536]
    SH = [(M, n, N, K, e) for M in Ms for n, N, K, e in (("qkv", 12288, 4096, E.EPI_NONE), ("o", 4096, 4096, E.EPI_BIAS_RESID), ("gate_up", 22016, 4096, E.EPI_SWIGLU),
                                                         ("down", 4096, 11008, E.EPI_NONE), ("lm_head", 32000, 4096, E.EPI_NONE))]
elif which == "vit":
    Ms = Ms or [16 * 577, 577]
    SH = [(M, n, N, K, e) for M in Ms for n, N, K, e in (("patch", 1024, 640, E.EPI_NONE), ("qkv", 3072, 1024, E.EPI_BIAS), ("wo", 1024, 1024, E.EPI_BIAS_RESID),
                                                         ("fc1", 4096, 1024, E.EPI_BIAS_QUICK_GELU), ("fc2", 1024, 4096, E.EPI_BIAS_RESID),
                                                         ("mm1", 4096, 1024, E.EPI_BIAS_GELU), ("mm2", 4096, 4096, E.EPI_BIAS))]
elif which == "13b":
    Ms = Ms or [270]
    SH = [(M, n, N, K, e) for M in Ms for n, N, K, e in (("qkv", 15360, 5120, E.EPI_NONE), ("o", 5120, 5120, E.EPI_NONE), ("gate_up", 27648, 5120, E.EPI_SWIGLU),
                                                         ("down", 5120, 13824, E.EPI_NONE))]
else:
    Ms = Ms or [39140]
    SH = [(M, n, N, K, e) for M in Ms for n, N, K, e in (("qkv", 12288, 4096, E.EPI_NONE), ("gate_up", 22016, 4096, E.EPI_SWIGLU), ("down", 4096, 11008, E.EPI_NONE))]
G = 1 << 20
bad, checked = [], 0
for M, name, N, K, epi in SH:
    g = torch.Generator(device=dev).manual_seed(1)
    x = (torch.randn(M, K, device=dev, generator=g) * 0.5).bfloat16()
    w = (torch.randn(N, K, device=dev, generator=g) * 0.02).bfloat16()
    No = N // 2 if epi == E.EPI_SWIGLU else N
    bias = (torch.randn(No, device=dev, generator=g) * 0.1).bfloat16() if epi in (E.EPI_BIAS, E.EPI_BIAS_QUICK_GELU, E.EPI_BIAS_GELU, E.EPI_BIAS_RESID) else None
    resid = (torch.randn(M, No, device=dev, generator=g) * 0.5).bfloat16() if epi == E.EPI_BIAS_RESID else None
    acc = x.float() @ w.float().t()
    if epi == E.EPI_SWIGLU:
        want = F.silu(acc[:, :No].bfloat16().float()).bfloat16().float() * acc[:, No:].bfloat16().float()
    else:
        want = acc
        if bias is not None: want = want + bias.float()
        if epi == E.EPI_BIAS_QUICK_GELU: want = want * torch.sigmoid(1.702 * want)
        if epi == E.EPI_BIAS_GELU: want = F.gelu(want)
        if resid is not None: want = want.bfloat16().float() + resid.float()
    need = ops._gemm_workspace(x.device, M, No).numel()
    for c, sch in ops.GEMM_CANDIDATES:
        if (epi == E.EPI_SWIGLU and c in (5, 6, 7, 9)) or (c == 8 and M > 256):
            continue
        cfg = c + 16 * sch
        wsbuf = torch.full((need + 2 * G,), 0x5A, dtype=torch.uint8, device=dev)
        ws = wsbuf[G:G + need]; ws.zero_()
        obuf = torch.full((M * No + 2 * G,), -7.0, dtype=torch.bfloat16, device=dev)
        out = obuf[G:G + M * No].view(M, No)
        for rep in range(3):
            ops._gemm_call(x, w, out, bias, resid, M, No, K, epi, cfg, ws)
        torch.cuda.synchronize()
        ok_guard = bool((wsbuf[:G] == 0x5A).all() and (wsbuf[G + need:] == 0x5A).all() and (obuf[:G] == -7.0).all() and (obuf[G + M * No:] == -7.0).all())
        ok_cnt = bool((ws[: 4 << 20] == 0).all())
        err = (out.float() - want).abs().max().item() / max(want.abs().max().item(), 1e-6)
        checked += 1
        if not (ok_guard and ok_cnt and err < 0.03):
            bad.append({"M": M, "shape": name, "tile": c, "sched": sch, "guards": ok_guard, "counters_zero": ok_cnt, "rel_err": round(err, 4)})
            print(json.dumps(bad[-1]), flush=True)
print(json.dumps({"set": which, "rows": Ms, "launch_configs_checked": checked, "bad": len(bad)}), flush=True)
