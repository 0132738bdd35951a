"""Library-GEMM probe: default hipBLASLt heuristic vs torch TunableOp selection for the decode/prefill shapes.
Usage: python tools/gemm_tune_probe.py [--tune] [--out gpurun_out/gemm_tune]"""
import argparse, json, os, sys, time
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--tune", action="store_true")
ap.add_argument("--out", default="gpurun_out/gemm_tune")
ap.add_argument("--ms", default="384,512,640,704,768,832,896,1024,1152,1536")
ap.add_argument("--tune-ms", default="768")
a = ap.parse_args()
os.makedirs(a.out, exist_ok=True)
dev = "cuda:0"
SHAPES = [("qkv", 12288, 4096), ("wo", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)]


def timeit(fn, iters=30, warm=5):
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
for M in [int(x) for x in a.ms.split(",")]:
    tot = 0.0
    for n, N, K in SHAPES[:4]:
        x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        us = timeit(lambda: torch.matmul(x, W[n].t()))
        tot += us
        res.append(dict(mode="default", M=M, name=n, N=N, K=K, us=us, tflops=2 * M * N * K / us / 1e6))
    print(f"default M={M:5d} layer GEMMs {tot:8.1f} us  {tot / M * 1e3:7.1f} ns/row", flush=True)
if a.tune:
    import torch.cuda.tunable as tn
    tn.enable(True)
    tn.tuning_enable(True)
    tn.set_max_tuning_duration(40)
    tn.set_max_tuning_iterations(20)
    tn.set_filename(os.path.join(a.out, "tunableop_results.csv"))
    for M in [int(x) for x in a.tune_ms.split(",")]:
        tot = 0.0
        for n, N, K in SHAPES:
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            t0 = time.time()
            torch.matmul(x, W[n].t())
            torch.cuda.synchronize()
            tt = time.time() - t0
            us = timeit(lambda: torch.matmul(x, W[n].t()))
            if n != "lm_head":
                tot += us
            res.append(dict(mode="tuned", M=M, name=n, N=N, K=K, us=us, tflops=2 * M * N * K / us / 1e6, tune_s=tt))
            print(f"tuned   M={M:5d} {n:8s} {us:8.1f} us {2 * M * N * K / us / 1e6:7.1f} TF/s (tuning took {tt:.1f}s)", flush=True)
        print(f"tuned   M={M:5d} layer GEMMs {tot:8.1f} us", flush=True)
    tn.write_file()
    print(tn.get_results())
with open(os.path.join(a.out, "gemm_tune.jsonl"), "w") as f:
    for r in res:
        f.write(json.dumps(r) + "\n")
for r in res:
    if r["M"] in (768,):
        print(r)
