"""Where the norm-fused five-launch decoder layer stops paying: decode step (2 branch rows per question) with the layer's row limits
swept - ops.UNEVEN_FUSED_MAX_M for LLaVA-1.5-13B (d = 5120), ops.NORM_FUSED_MAX_M for 7B widths.  python tools/fused_band_probe.py
llava-1.5-13b 1 2 3 4   Record: profiles/r05_fused_band_sweep.jsonl."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llava_align_amd.engine import VddLlavaEngine
from llava_align_amd import ops
dev = "cuda:0"
model = sys.argv[1]
eng = VddLlavaEngine(model, device=dev, use_graph=True)
rng = np.random.default_rng(5)
sys_tok = [1] + rng.integers(3, 32000, size=34).tolist()
g = torch.Generator().manual_seed(3)
for nq in [int(a) for a in sys.argv[2:]]:
    ids = [torch.tensor(sys_tok + [-200] + rng.integers(3, 32000, size=int(np.clip(rng.normal(60, 20), 10, 120))).tolist()) for _ in range(nq)]
    imgs = [torch.randn(3, 336, 336, generator=g) for _ in range(nq)]
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, seed=1)
    rec = {"model": model, "questions": nq, "rows": 2 * nq}
    variants = ((("fused<=8", 8, 16), ("fused<=4", 4, 16), ("fused<=2", 2, 16), ("fused<=0", 0, 16)) if "13b" in model else
                (("fused<=16", 8, 16), ("fused<=8", 8, 8), ("fused<=16_again", 8, 16), ("fused<=8_again", 8, 8)))
    for name, uf, nf in variants:
        ops.UNEVEN_FUSED_MAX_M, ops.NORM_FUSED_MAX_M = uf, nf
        e = VddLlavaEngine(eng.cfg, weights=eng.w, device=dev, use_graph=True)
        def timed(n_new):
            for _ in range(2): e.generate(ids, max_new_tokens=n_new, **kw)
            torch.cuda.synchronize(); ts = []
            for _ in range(4):
                t0 = time.perf_counter(); e.generate(ids, max_new_tokens=n_new, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return min(ts)
        rec[name] = round((timed(48) - timed(16)) / 32 * 1e3, 3)
        del e
    print(json.dumps(rec), flush=True)
