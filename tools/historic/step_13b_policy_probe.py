"""Decode step of LLaVA-1.5-13B (3 branch rows per question, one image each) with ops.UNEVEN_BLOCKS_TO_SLABS off / on: above 8 rows the
d-wide projections of d = 5120 leave the weight-streaming kernels for the GEMM's split-K slabs.  Record: profiles/r05_13b_step_policy.jsonl."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from llava_align_amd.engine import VddLlavaEngine, LanguageModel
from llava_align_amd import ops
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-13b", device=dev, use_graph=True)
rng = np.random.default_rng(5)
sys_tok = [1] + rng.integers(3, 32000, size=34).tolist()
g = torch.Generator().manual_seed(3)
for nq in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5, 8, 11]:
    ids = [torch.tensor(sys_tok + [-200] + rng.integers(3, 32000, size=int(np.clip(rng.normal(60, 20), 10, 120))).tolist()) for _ in range(nq)]
    imgs = [torch.randn(3, 336, 336, generator=g) for _ in range(nq)]
    rec = {"model": "13b", "questions": nq, "rows": 3 * nq}
    table = dict(ops.SKINNY_ROWS_MEASURED)
    for name, slabs, measured in (("old_policy", False, False), ("slabs", True, False), ("slabs+measured_crossovers", True, True), ("old_policy_again", False, False),
                                  ("slabs+measured_crossovers_again", True, True)):
        ops.UNEVEN_BLOCKS_TO_SLABS = slabs
        ops.SKINNY_ROWS_MEASURED.clear()
        if measured:
            ops.SKINNY_ROWS_MEASURED.update(table)
        e = VddLlavaEngine(eng.cfg, weights=eng.w, device=dev, use_graph=True)
        kw = dict(images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, seed=1)
        def timed(n_new):
            for _ in range(2): e.generate(ids, max_new_tokens=n_new, **kw)
            torch.cuda.synchronize(); ts = []
            for _ in range(4):
                t0 = time.perf_counter(); e.generate(ids, max_new_tokens=n_new, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return min(ts)
        rec[name] = round((timed(48) - timed(16)) / 32 * 1e3, 3)
        del e
    ops.SKINNY_ROWS_MEASURED.update(table)
    print(json.dumps(rec), flush=True)
