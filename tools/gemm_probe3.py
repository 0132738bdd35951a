"""vdd_gemm (csrc/vdd_gemm.hip) vs hipBLASLt (torch.matmul) on MI355X: correctness of every tile config / epilogue, then
timings at the decode, prefill and ViT shapes of LLaVA-1.5-7B.  Weights rotate through several copies so that the 256 MiB
Infinity Cache does not hold them between launches (a decode step streams 13 GB of weights).
  python tools/gemm_probe3.py [--quick] [--out gpurun_out/gemm_probe3.jsonl]"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import llava_align_amd as L  # noqa: E402
from llava_align_amd import _lib  # noqa: E402

_P, _I, _L = C.c_void_p, C.c_int, C.c_int64
lib = _lib.load_lib()
lib.vdd_gemm.argtypes = [_P, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _L, _I, _I, _P, _L, _I, _P]      # ABI 3: dtype in front of the stream
lib.vdd_gemm.restype = C.c_int
lib.vdd_gemm_workspace_bytes.argtypes = [_I, _I]
lib.vdd_gemm_workspace_bytes.restype = C.c_int64
EPI = dict(none=0, bias=1, bias_quick_gelu=2, bias_gelu=3, swiglu=4, bias_resid=5)
dev = torch.device("cuda:0")
_ws = {}


def workspace(M, N):
    need = lib.vdd_gemm_workspace_bytes(M, N)
    w = _ws.get("w")
    if w is None or w.numel() < need:
        w = _ws["w"] = torch.zeros(need, dtype=torch.uint8, device=dev)
    return w


def gemm(x, w, epi="none", bias=None, resid=None, cfg=1, out=None):
    M, K = x.shape
    N = w.shape[0] // 2 if epi == "swiglu" else w.shape[0]
    st = torch.cuda.current_stream().cuda_stream
    ws = workspace(M, N)
    out = torch.empty(M, N, dtype=x.dtype, device=dev) if out is None else out
    rc = lib.vdd_gemm(x.data_ptr(), w.data_ptr(), out.data_ptr(), bias.data_ptr() if bias is not None else None,
                      resid.data_ptr() if resid is not None else None, M, N, K, x.stride(0), w.stride(0), out.stride(0),
                      resid.stride(0) if resid is not None else 0, EPI[epi], cfg, ws.data_ptr(), ws.numel(),
                      _lib.VDD_F16 if x.dtype == torch.float16 else _lib.VDD_BF16, st)
    assert rc == 0, (rc, lib.vdd_last_error())
    return out


def ref(x, w, epi, bias, resid):
    acc = x.float() @ w.float().t()
    bf = lambda t: t.to(torch.bfloat16).float()
    if epi == "none":
        return bf(acc)
    if epi == "swiglu":
        F = w.shape[0] // 2
        g, u = bf(acc[:, :F]), bf(acc[:, F:])
        return bf(bf(g / (1 + torch.exp(-g))) * u)
    y = bf(acc + bias.float())
    if epi == "bias":
        return y
    if epi == "bias_quick_gelu":
        return bf(y / (1 + torch.exp(-1.702 * y)))
    if epi == "bias_gelu":
        return bf(torch.nn.functional.gelu(y))
    if epi == "bias_resid":
        return bf(y + resid.float())


def check():
    g = torch.Generator(device=dev).manual_seed(0)
    bad = 0
    for (M, N, K) in [(300, 520, 256), (768, 1024, 512), (77, 136, 128), (1000, 2304, 1024), (768, 4096, 4096), (2000, 12288, 1024)]:
        for cfg in (1, 2, 3, 4, 5):
            for epi in EPI:
                if epi == "swiglu":
                    Nw = 2 * ((N + 127) // 128 * 128)
                else:
                    Nw = N
                x = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
                w = (torch.randn(Nw, K, device=dev, generator=g) * 0.05).to(torch.bfloat16)
                # asymmetric structure: a transposed or shifted tile shows up as O(1) error
                w[:, 0] += torch.arange(Nw, device=dev).to(torch.bfloat16) * 0.001
                bias = torch.randn(Nw, device=dev, generator=g).to(torch.bfloat16)
                resid = torch.randn(M, Nw, device=dev, generator=g).to(torch.bfloat16)
                if epi == "swiglu" and cfg == 5:
                    continue
                y = gemm(x, w, epi, bias, resid, cfg)
                r = ref(x, w, epi, bias, resid)
                err = (y.float() - r).abs().max().item()
                tol = 0.02 * r.abs().max().item() + 1e-3
                ok = err <= tol and torch.isfinite(y.float()).all().item()
                bad += not ok
                if not ok or (cfg == 1 and epi in ("none", "swiglu")):
                    print(f"check M={M} N={N} K={K} cfg={cfg} {epi:16s} max_err={err:.4g} tol={tol:.3g} {'ok' if ok else 'FAIL'}", flush=True)
    print("CHECK", "PASS" if bad == 0 else f"FAIL ({bad})", flush=True)
    return bad == 0


def timeit(fn, n_rot, iters=20, warm=3):
    for i in range(warm):
        fn(i % n_rot)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % n_rot)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def bench(shapes, out_path, cfgs=(1, 2, 3, 4, 5)):
    g = torch.Generator(device=dev).manual_seed(1)
    rows = []
    for tag, M, N, K in shapes:
        n_rot = max(2, min(8, int(600e6 // (N * K * 2)) + 1))
        ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(n_rot)]
        xs = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(2)]
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        flops = 2.0 * M * N * K
        rec = dict(tag=tag, M=M, N=N, K=K)
        t = timeit(lambda i: torch.matmul(xs[i & 1], ws[i].t(), out=y), n_rot)
        rec["hipblaslt_us"], rec["hipblaslt_TF"] = round(t, 1), round(flops / t / 1e6, 1)
        for cfg in cfgs:
            t = timeit(lambda i: gemm(xs[i & 1], ws[i], "none", cfg=cfg, out=y), n_rot)
            rec[f"cfg{cfg}_us"], rec[f"cfg{cfg}_TF"] = round(t, 1), round(flops / t / 1e6, 1)
        rows.append(rec)
        print(json.dumps(rec), flush=True)
        del ws, xs, y
        torch.cuda.empty_cache()
    if out_path:
        os.makedirs(os.path.dirname(out_path), exist_ok=True)
        with open(out_path, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--out", default="gpurun_out/gemm_probe3.jsonl")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    if not a.no_check:
        check()
    lm = [("qkv", 12288, 4096), ("wo", 4096, 4096), ("wgu", 22016, 4096), ("wd", 4096, 11008), ("lm_head", 32000, 4096)]
    shapes = []
    for M in ((768,) if a.quick else (384, 768, 1536)):
        shapes += [(f"decode{M}.{n}", M, N, K) for n, N, K in lm]
    shapes += [(f"prefill.{n}", 39140, N, K) for n, N, K in lm[:4]]
    if not a.quick:
        shapes += [(f"prefill18k.{n}", 18114, N, K) for n, N, K in lm[:4]]
        shapes += [("vit.qkv", 9232, 3072, 1024), ("vit.wo", 9232, 1024, 1024), ("vit.fc1", 9232, 4096, 1024), ("vit.fc2", 9232, 1024, 4096),
                   ("vit.patch", 9216, 1024, 640), ("mm.w1", 9216, 4096, 1024), ("mm.w2", 9216, 4096, 4096)]
    bench(shapes, a.out)


if __name__ == "__main__":
    main()
