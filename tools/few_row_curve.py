"""Decode-step time at 17 - 64 rows whose prefixes are NOT shared (one image per question, use_dd + use_dd_unk = 3 branch rows per
question: the per-rank shape of BASELINE config #3 on 8 GPUs, config #5 on 4): the one-launch RoPE + KV write + attention kernel
(ops.FUSED_ATTN_UNGROUPED_MAX_M = 64) against rope_kv + split-KV attention + combine (= 0).  (t(48) - t(16)) / 32 new tokens per
point, graph-captured steps."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from llava_align_amd.engine import VddLlavaEngine
from llava_align_amd import ops
dev = "cuda:0"
model = os.environ.get("VDD_MODEL", "llava-1.5-7b")
eng = VddLlavaEngine(model, device=dev, use_graph=True)
rng = np.random.default_rng(5)
sys_tok = [1] + rng.integers(3, 32000, size=34).tolist()
g = torch.Generator().manual_seed(3)
points = [int(a) for a in sys.argv[1:]] or [6, 8, 11, 16, 21]
for nq in points:
    ids = [torch.tensor(sys_tok + [-200] + rng.integers(3, 32000, size=int(np.clip(rng.normal(60, 20), 10, 120))).tolist()) for _ in range(nq)]
    imgs = [torch.randn(3, 336, 336, generator=g) for _ in range(nq)]
    rec = {"model": model, "questions": nq, "rows": 3 * nq}
    for name, fmax in (("split_kv_3_launches", 0), ("fused_1_launch", 64)):
        ops.FUSED_ATTN_UNGROUPED_MAX_M = fmax
        e = VddLlavaEngine(eng.cfg, weights=eng.w, device=dev, use_graph=True)
        kw = dict(images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, seed=1)
        def timed(n_new):
            for _ in range(2):
                e.generate(ids, max_new_tokens=n_new, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                t0 = time.perf_counter(); out = e.generate(ids, max_new_tokens=n_new, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return min(ts), out
        (t48, out), (t16, _) = timed(48), timed(16)
        rec[name + "_ms_per_step"] = round((t48 - t16) / 32 * 1e3, 3)
        rec[name + "_n_groups"] = out.stats.get("n_groups")
        del e
    print(json.dumps(rec), flush=True)
