"""One question in flight (2 rows): does pulling the NEXT projections' weights towards the chip while the attention kernel runs - the
one phase of the five-launch layer in which HBM idles - shorten the layer?  The captured chain (engine.LanguageModel.
_decode_step_few_rows) against the same chain with a forked branch per layer that reads W_o (and optionally the head of W_gate/up)
on a second stream beside the attention launch.  The stand-in prefetch is a plain read (torch sum over an int32 view): it fills the
XCD L2s / the Infinity Cache exactly as a dedicated prefetch kernel would.  python tools/probes/lost_kernels/prefetch_overlap_probe.py [--mb 33]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))                       # lost_ops.py, the lab tests
import torch
import test_persistent_layers_gpu as T
import lost_ops as ops                   # lab entries + the product's ops

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--iters", type=int, default=30)
a = ap.parse_args()
T.DT = torch.bfloat16
d, H, F, M, eps, D = 4096, 32, 11008, 2, 1e-5, 128
ctx = 650
L, rows, pos, cpos, slot, x, cs = T._setup(ops, M, d, H, F, a.layers, [ctx, 75], [ctx - 39, 36], seed=1)
side2 = torch.cuda.Stream()
sinks = [torch.zeros((), dtype=torch.int32, device=T.DEV) for _ in range(a.layers)]
sinks2 = [torch.zeros((), dtype=torch.int32, device=T.DEV) for _ in range(a.layers)]


def chain(prefetch_mb):
    resid, ss = x, None
    for i, l in enumerate(L):
        if ss is None:
            qkv = ops.linear(ops.rmsnorm(resid, l["ln1"], eps), l["wqkv"], bias=l["bqkv"])
        else:
            qkv = ops.linear_normed(resid, ss, l["ln1"], eps, l["wqkv"], bias=l["bqkv"])
        cur = torch.cuda.current_stream()
        if prefetch_mb > 0:
            side2.wait_stream(cur)                                          # fork behind the qkv projection
            with torch.cuda.stream(side2):
                n = min(prefetch_mb, 33) * (1 << 20) // 4
                sinks[i].copy_(l["wo"].view(torch.int32).view(-1)[:n].max())
                if prefetch_mb > 33:
                    n2 = (prefetch_mb - 33) * (1 << 20) // 4
                    sinks2[i].copy_(l["wgu"].view(torch.int32).view(-1)[:n2].max())
        att = ops.decode_attention_fused(qkv, pos, cpos, slot, cs, l["k_own"], l["v_own"], rows, H, H, D, k_prefix=l["k_pre"], v_prefix=l["v_pre"])
        if prefetch_mb > 0:
            cur.wait_stream(side2)                                          # join in front of the attention-output projection
        resid, ss = ops.linear_resid_ss(att, l["wo"], resid)
        act = ops.swiglu_linear_normed(resid, ss, l["ln2"], eps, l["wgu"])
        resid, ss = ops.linear_resid_ss(act, l["wd"], resid)
    return resid, ss


out = {"layers": a.layers, "rows": M}
for mb in (0, 8, 33, 60, 0):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            r = chain(mb)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        r = chain(mb)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.iters
    out.setdefault("us_per_layer_by_prefetch_MB", []).append([mb, round(dt / a.layers * 1e6, 2)])
print(json.dumps(out))
