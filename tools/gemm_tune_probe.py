"""Library-GEMM probe: default hipBLASLt heuristic vs torch TunableOp selection for the decode/prefill shapes.
Usage: python tools/gemm_tune_probe.py [--ms 768,...] [--out gpurun_out/gemm_tune]   (writes tunableop_results.csv + gemm_tune.jsonl)"""
import argparse, json, os, sys, time
import torch
import torch.cuda.tunable as tn

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/gemm_tune")
ap.add_argument("--ms", default="768")
ap.add_argument("--sweep", default="384,512,640,704,768,832,896,1024,1152,1536")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
dev = "cuda:0"
SHAPES = [("qkv", 12288, 4096), ("wo", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)]


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


W = {n: torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02 for n, N, K in SHAPES}
res = []
for M in [int(x) for x in a.sweep.split(",") if x]:
    tot = 0.0
    for n, N, K in SHAPES[:4]:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: torch.matmul(x, W[n].t()))
        tot += us
        res.append(dict(mode="default", M=M, name=n, N=N, K=K, us=us, tflops=2 * M * N * K / us / 1e6))
    print(f"default M={M:5d} layer GEMMs {tot:8.1f} us  {tot / M * 1e3:7.1f} ns/row", flush=True)
tn.set_filename(os.path.join(a.out, "tunableop_results.csv"))
tn.set_max_tuning_duration(60)
tn.set_max_tuning_iterations(30)
for M in [int(x) for x in a.ms.split(",")]:
    for n, N, K in SHAPES:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.matmul(x, W[n].t())
        tn.enable(False)
        d0 = timeit(f)
        tn.enable(True); tn.tuning_enable(True)
        t0 = time.time(); f(); torch.cuda.synchronize(); tt = time.time() - t0
        tn.tuning_enable(False)
        t1 = timeit(f)
        tn.enable(False)
        d1 = timeit(f)
        tn.enable(True)
        t2 = timeit(f)
        tn.enable(False)
        r = dict(mode="ab", M=M, name=n, N=N, K=K, default_us=[round(d0, 1), round(d1, 1)], tuned_us=[round(t1, 1), round(t2, 1)], tune_s=round(tt, 1))
        res.append(r)
        print(json.dumps(r), flush=True)
print(tn.get_results())
with open(os.path.join(a.out, "gemm_tune.jsonl"), "w") as f:
    for r in res:
        f.write(json.dumps(r) + "\n")
