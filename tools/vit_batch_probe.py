"""CLIP ViT-L/14-336 + projector: time per image against the images per forward (the engine cuts a call's distinct images into chunks
of VddLlavaEngine.VIT_CHUNK).  At 16 images the tower's GEMMs are 9,232-row products with K = 1,024 / 4,096: 148 - 592 tiles on 256 CUs,
0.5 PF/s in the bench trace (profiles/r05_bench_kernel_medians.txt: gemm_kernel<256, 256, 2, 4, {1, 2, 5}>); larger chunks fill the chip."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd.engine import VddLlavaEngine
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-7b", device=dev, use_graph=True)
vit = eng.vit
g = torch.Generator(device=dev).manual_seed(1)
for n in (16, 32, 64, 128):
    x = torch.randn(n, 3, 336, 336, device=dev, generator=g).to(torch.bfloat16)
    for graph in (False, True):
        if graph and n not in vit.GRAPH_SIZES:
            vit.GRAPH_SIZES = tuple(vit.GRAPH_SIZES) + (n,)
        f = (lambda: vit(x)) if graph else (lambda: vit._forward(x))
        for _ in range(3):
            y = f()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); y = f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        t = sorted(ts)[2]
        flops = n * 577 * 23 * 2 * (4 * 1024 * 1024 + 2 * 1024 * 4096)
        print(json.dumps({"images": n, "graph": graph, "ms": round(t * 1e3, 2), "ms_per_image": round(t * 1e3 / n, 3),
                          "tower_gemm_PFs_if_all_gemm": round(flops / t / 1e15, 3), "hbm_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}), flush=True)
