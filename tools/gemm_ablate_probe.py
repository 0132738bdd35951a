"""Where does the persistent GEMM's time go?  Timing-only ablations of its K loop (csrc/vdd_gemm.hip built with -DVDD_GEMM_ABLATE:
config bits 8-12 switch off the X / W LDS-DMA, the W / X fragment reads, the per-tile barrier - results are WRONG, what is measured
is the time of what remains).  Build the probe library in the build container, run on the GPU box:

    python tools/gemm_ablate_probe.py --build                       # -> tools/probes/libvdd_ablate.so
    VDD_HIP_LIB=tools/probes/libvdd_ablate.so python tools/gemm_ablate_probe.py [M ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "probes", "libvdd_ablate.so")


def build(extra_defs=("-DVDD_GEMM_ABLATE",), lib=None):
    lib = lib or LIB
    from importlib import import_module
    B = import_module("llava_align_amd._build")
    obj_dir = os.path.join(ROOT, "build", "obj_ablate")
    os.makedirs(obj_dir, exist_ok=True)
    objs, procs = [], []
    for src, obj, extra in B.units():
        o = os.path.join(obj_dir, os.path.basename(obj))
        objs.append(o)
        cmd = [B.hipcc(), *B.CFLAGS, *extra, *extra_defs, "-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-c", src, "-o", o]
        procs.append(subprocess.Popen(cmd))
    assert all(p.wait() == 0 for p in procs)
    subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib], check=True)
    print(lib)


def main():
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from gemm_probe3 import dev, gemm, timeit
    assert os.environ.get("VDD_HIP_LIB", "").endswith("libvdd_ablate.so"), "run with VDD_HIP_LIB=tools/probes/libvdd_ablate.so"
    g = torch.Generator(device=dev).manual_seed(1)
    variants = [("full", 0), ("no_X_dma", 1), ("no_W_dma", 2), ("no_dma", 3), ("no_W_reads", 4), ("no_X_reads", 8), ("no_reads", 12),
                ("no_dma_no_reads", 15), ("no_barrier", 16), ("mfma_only", 31)]
    shapes = [("qkv", 12288, 4096), ("wd", 4096, 11008)]
    for M in ([int(a) for a in sys.argv[1:]] or (1536, 39140)):
        for name, N, K in shapes:
            n_rot = max(2, min(8, int(600e6 // (N * K * 2)) + 1))
            ws = [(torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(n_rot)]
            xs = [torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16) for _ in range(2)]
            y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            flop = 2.0 * M * N * K
            rec = dict(tag=f"M{M}.{name}")
            for cfg in ((1, 4) if M <= 4096 else (1,)):
                for vn, bits in variants:
                    t = timeit(lambda i: gemm(xs[i & 1], ws[i], "none", cfg=cfg + (bits << 16), out=y), n_rot)
                    rec[f"c{cfg}.{vn}"] = [round(t, 1), round(flop / t / 1e9, 3)]
            print(json.dumps(rec), flush=True)
            del ws, xs, y


if __name__ == "__main__":
    if "--build-variant" in sys.argv:          # e.g. --build-variant -DVDD_GEMM_SPLIT_STAGE=0 tools/probes/libvdd_nosplit.so: an A/B library for gemm_cfg_probe.py
        i = sys.argv.index("--build-variant")
        build((sys.argv[i + 1],), os.path.join(ROOT, sys.argv[i + 2]))
    elif "--build" in sys.argv:
        build()
    else:
        main()
