"""A/B of two builds of the library on the projections' GEMMs at prefill size (and the decode batch): alternating subprocesses, each loading
one build through VDD_HIP_LIB, weights rotated through > 256 MiB, median of the launch times.  Prints PF/s per (shape, build) and checks that
both builds produce the same bits.
  python tools/gemm_lib_ab.py llava-align_amd/lib/libvdd_hip.so llava-align_amd/lib/libvdd_hip_l2pf.so"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("qkv", 39168, 12288, 4096, 0), ("o", 39168, 4096, 4096, 0), ("gate_up_swiglu", 39168, 11008, 4096, 4), ("down", 39168, 4096, 11008, 0),
          ("qkv_74k", 73728, 12288, 4096, 0), ("gate_up_74k", 73728, 11008, 4096, 4), ("qkv_decode_1536", 1536, 12288, 4096, 0)]

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from llava_align_amd import ops
    dev = "cuda:0"
    out = {}
    for name, M, N, K, epi in SHAPES:
        g = torch.Generator(device=dev).manual_seed(1)
        x = (torch.randn(M, K, device=dev, generator=g) * 0.5).to(torch.bfloat16)
        rows = 2 * N if epi == 4 else N
        n_w = max(2, -(-320 * 2 ** 20 // (rows * K * 2)))
        ws = [(torch.randn(rows, K, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(n_w)]
        y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        cfg = (4 if M <= 4096 else 1)
        f = lambda i: ops.gemm(x, ws[i % n_w], epi=epi, out=y, config=cfg)
        for i in range(3):
            f(i)
        torch.cuda.synchronize()
        ts = []
        for i in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(i); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        t = sorted(ts)[len(ts) // 2] * 1e-3
        f(0)
        out[name] = {"PFs": round(2.0 * M * N * K * (2 if epi == 4 else 1) / t / 1e15, 4), "us": round(t * 1e6, 1),
                     "checksum": float(y.float().abs().sum())}
    print(json.dumps(out))
    sys.exit(0)

libs = sys.argv[1:3]
res = {l: [] for l in libs}
for rep in range(3):
    for l in libs:
        env = dict(os.environ, VDD_HIP_LIB=os.path.join(ROOT, l) if not os.path.isabs(l) else l)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
        line = [x for x in r.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(r.stderr[-2000:], file=sys.stderr); sys.exit(1)
        res[l].append(json.loads(line[-1]))
for name, *_ in SHAPES:
    rec = {"shape": name}
    for l in libs:
        rec[os.path.basename(l)] = [r[name]["PFs"] for r in res[l]]
    rec["same_bits"] = len({r[name]["checksum"] for l in libs for r in res[l]}) == 1
    print(json.dumps(rec), flush=True)
