"""17 - 64 rows: the slab projections (vdd_skinny_slab.hip, K cut over workgroups, X staged once per workgroup) against what the
decoder layer ran before them - the 16 / 32-column weight-streaming kernels and the MFMA GEMM, with the RMSNorm launches they need -
per projection of LLaVA-1.5-7B (or 13B: argv[1] = 13b), weights rotated through > 600 MB.  One JSON line per row count."""
import sys, json, os, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))                       # lost_ops.py, the lab tests
import lost_ops as ops                   # lab entries + the product's ops
dev = "cuda"
d, F = (5120, 13824) if len(sys.argv) > 1 and sys.argv[1] == "13b" else (4096, 11008)
def t(fn, n=24):
    for i in range(4): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 1)
def rot(N, K): return [(torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16) for _ in range(max(2, int(6.5e8 / (N * K * 2))))]
Wq, Wo, Wg, Wd = rot(3 * d, d), rot(d, d), rot(2 * F, d), rot(d, F)
ln = torch.ones(d, device=dev, dtype=torch.bfloat16)
for M in (17, 18, 24, 32, 34, 48, 64):
    x, xf = torch.randn(M, d, device=dev).to(torch.bfloat16), torch.randn(M, F, device=dev).to(torch.bfloat16)
    res = torch.randn(M, d, device=dev).to(torch.bfloat16)
    _, ss = ops.slab_linear(x, Wo[0], resid=res, want_ss=True)
    rec = {"M": M, "d": d, "F": F}
    rec["qkv"] = {"old_norm+linear": t(lambda i: ops.linear(ops.rmsnorm(res, ln, 1e-5), Wq[i % len(Wq)])),
                  "old_linear": t(lambda i: ops.linear(x, Wq[i % len(Wq)])),
                  "slab": t(lambda i: ops.slab_linear(x, Wq[i % len(Wq)])),
                  "slab_normed": t(lambda i: ops.slab_linear(res, Wq[i % len(Wq)], ss=ss, ln_w=ln, eps=1e-5))}
    rec["o"] = {"old_linear_to_norm+norm": t(lambda i: ops.rmsnorm(res, ln, 1e-5, delta=ops.linear_to_norm(x, Wo[i % len(Wo)]), resid_out=res)),
                "slab_resid_ss": t(lambda i: ops.slab_linear(x, Wo[i % len(Wo)], resid=res, want_ss=True))}
    rec["gate_up"] = {"old_swiglu": t(lambda i: ops.swiglu_linear(x, Wg[i % len(Wg)])),
                      "slab": t(lambda i: ops.slab_linear(x, Wg[i % len(Wg)], swiglu=True)),
                      "slab_normed": t(lambda i: ops.slab_linear(res, Wg[i % len(Wg)], ss=ss, ln_w=ln, eps=1e-5, swiglu=True))}
    rec["down"] = {"old_linear_to_norm+norm": t(lambda i: ops.rmsnorm(res, ln, 1e-5, delta=ops.linear_to_norm(xf, Wd[i % len(Wd)]), resid_out=res)),
                   "slab_resid_ss": t(lambda i: ops.slab_linear(xf, Wd[i % len(Wd)], resid=res, want_ss=True))}
    rec["layer_old_us"] = round(rec["qkv"]["old_norm+linear"] + rec["o"]["old_linear_to_norm+norm"] + rec["gate_up"]["old_swiglu"] + rec["down"]["old_linear_to_norm+norm"], 1)
    rec["layer_slab_us"] = round(rec["qkv"]["slab_normed"] + rec["o"]["slab_resid_ss"] + rec["gate_up"]["slab_normed"] + rec["down"]["slab_resid_ss"], 1)
    print(json.dumps(rec), flush=True)
