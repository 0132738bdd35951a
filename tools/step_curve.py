"""Decode-step time against the number of rows in flight (LLaVA-1.5-7B, use_dd_unk: rows = 2 x questions; 6 questions per image):
(t(64 new tokens) - t(32 new tokens)) / 32 per point, graph-captured steps.  Shows where the step leaves the weight-stream regime
(13.2 GB per step) and the switch points between the few-row layer (<= 16 rows), the weight-streaming projections (ops.skinny_rows)
and the MFMA GEMM."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
dev = "cuda:0"
eng = VddLlavaEngine("llava-1.5-7b", device=dev, use_graph=True)
points = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 6, 8, 9, 12, 16, 17, 24, 32, 33, 48, 64, 96, 128, 192, 256, 384]
for nq in points:
    n_img, per = (nq + 5) // 6, 6
    ids, imgs = pope_prompts(n_img, per_img=per, seed=5)
    ids, imgs = ids[:nq], imgs[:nq]
    kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=3)
    def timed(n_new):
        for _ in range(2):
            eng.generate(ids, max_new_tokens=n_new, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); out = eng.generate(ids, max_new_tokens=n_new, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return min(ts), out.stats      # host pauses only ever add time (a median of three still moved single points by 1 ms)
    (t64, st), (t32, _) = timed(64), timed(32)
    step = (t64 - t32) / 32
    print(json.dumps({"questions": nq, "rows": 2 * nq, "ms_per_step": round(step * 1e3, 3), "us_per_row": round(step * 1e6 / (2 * nq), 1),
                      "decode_tokens_per_s": round(nq / step, 1), "captured_step": bool(st.get("graph")), "n_groups": st.get("n_groups")}), flush=True)
