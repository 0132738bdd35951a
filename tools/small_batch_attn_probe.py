"""Few rows in flight: grouped prefix attention (3 launches per layer) vs the fused RoPE + KV write + attention kernel (1 launch,
the shared prefix re-read per row).  Measured on MI355X (tokens/s, 64 new tokens): 4 rows 411 vs 448, 6 rows 537 vs 579, 12 rows
1,241 vs 1,301, 16 rows 1,554 vs 1,618, 24 rows 2,092 vs 2,020 (grouped wins from here)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd.engine import VddLlavaEngine
from llava_align_amd import ops
from bench import pope_prompts
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-7b", device=dev, use_graph=True)
for n_img, per in ((1, 2), (1, 3), (1, 6), (1, 8), (2, 6)):
    ids, imgs = pope_prompts(n_img, per_img=per, seed=5)
    rec = {"questions": n_img * per, "rows": 2 * n_img * per}
    for name, fmax in (("grouped", 0), ("fused", 64)):
        ops.FUSED_ATTN_MAX_M = fmax                   # 0: the engine groups whenever grouping pays; 64: never below 64 rows
        e = VddLlavaEngine(eng.cfg, weights=eng.w, device=dev, use_graph=True)
        kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, max_new_tokens=64, seed=3)
        for _ in range(2): e.generate(ids, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); e.generate(ids, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        rec[name] = round(n_img * per * 64 / sorted(ts)[1], 1)
        del e
    print(json.dumps(rec), flush=True)
