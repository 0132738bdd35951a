"""bench.py — contract: python bench.py --gpus N --steps K --warmup W  -> ONE JSON line on rank 0.

Legs
  kernel   the fused contrastive-sampling kernel at a bandwidth-meaningful batch
           (SURVEY.md §8d microbench: B rows x V=32000, bf16, use_dd_unk, scores on),
           timed with HIP events on the launch stream -> `roofline`
  cpu      the reference CPU path (oracle restatement of sample()'s tail, torch eager,
           stubbed forward) on a bounded sample -> `cpu_baseline`
  e2e      (engine) LLaVA-1.5-7B-shaped VDD generation -> `value`   [added with engine.py]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def bench_kernel(dev, B=4096, V=32000, dtype=torch.bfloat16, n_in=2, scores=True, iters=200, warmup=20):
    import llava_align_amd as L
    g = torch.Generator(device=dev).manual_seed(0)
    v = (torch.randn(B, V, device=dev, generator=g) * 4).to(dtype)
    v[torch.arange(B, device=dev), torch.randint(0, V, (B,), device=dev, generator=g)] = 25.0   # planted row max
    c = (v.float() + torch.randn(B, V, device=dev, generator=g) * 1.5).to(dtype)
    d = (v.float() + torch.randn(B, V, device=dev, generator=g) * 1.5).to(dtype) if n_in == 3 else None
    out_scores = torch.empty(B, V, dtype=dtype, device=dev) if scores else None
    toks = torch.empty(B, dtype=torch.long, device=dev)
    spec = L.WarpSpec(temperature=0.2)
    run = lambda i: L.contrast_sample(v, c, d, alpha=1.0, beta=0.1, warp=spec, out_tokens=toks, out_scores=out_scores,
                                      seed=0, offset=i)
    for i in range(warmup):
        run(i)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # HIP events on the launch stream
    e0.record()
    for i in range(iters):
        run(i)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    es = torch.finfo(dtype).bits // 8
    alg_bytes = B * ((n_in + (1 if scores else 0)) * V * es + 8)          # (n_in+n_out)*V*e + 8 per row
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": None, "kernel": "vdd_contrast_sample_kernel", "launch_us": round(ms * 1e3, 2),
            "algorithmic_bytes_per_launch": alg_bytes,
            "shape": {"B": B, "V": V, "dtype": str(dtype).split(".")[-1], "n_in": n_in, "scores_out": scores}}


def bench_cpu_tail(seconds=12.0, V=32000, dtype=torch.bfloat16):
    """Reference CPU path: the oracle's restatement of sample()'s per-step tail (use_dd_unk,
    T=0.2, softmax, multinomial) with the forward stubbed out, B=1 as every reference driver runs."""
    from oracle import vdd_oracle as O
    torch.manual_seed(0)
    bank = [(torch.randn(1, V) * 4).to(dtype) for _ in range(64)]
    warp = O.WarpConfig(temperature=0.2)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        v, c = bank[n % 64], bank[(n + 1) % 64]
        s = O.step_scores(v, c, None, 1.0, 0.1, warp)
        O.pick_multinomial(torch.softmax(s, -1))
        n += 1
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 1), "unit": "sampling-tail steps/s (B=1 rows/s)", "cores": torch.get_num_threads(),
            "kind": "port", "host_cpus": os.cpu_count(),
            "sample": f"{n} steps of the per-step tail at V={V} {str(dtype).split('.')[-1]} in {dt:.1f}s, forward stubbed"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rows", type=int, default=4096)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    rank, local, world = dist_env()
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    roof = bench_kernel(dev, B=a.rows, iters=a.steps, warmup=a.warmup)
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    rows_per_s = a.rows / (roof["launch_us"] * 1e-6)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([roof["launch_us"]], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rows_per_s = world * a.rows / (t.item() * 1e-6)
        dist.barrier()
    if rank == 0:
        cpu = None if a.no_cpu else bench_cpu_tail()
        line = {"metric": "fused contrastive sampling tail rows/s (interim line: e2e decode tokens/s lands with engine.py)",
                "value": round(rows_per_s, 1), "unit": "rows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": round(roof["launch_us"] / 1e3, 4), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "vdd_contrast_sample B=%d V=32000 bf16 use_dd_unk T=0.2 scores_out" % a.rows},
                "roofline": roof, "cpu_baseline": cpu, "wall_s": round(wall, 2)}
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
