"""Is the MEASURED layer form (ops.norm_fused_pays: five-launch norm-fused layer vs seven-launch layer, timed on one layer's projections + norms)
the one that makes the captured decode STEP faster?  Per model shape and row count: the step forced to each form against the measured pick.
python tools/layer_form_probe.py [7b|13b] ...   Record: profiles/r06_layer_form_probe.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from llava_align_amd import ops
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
dev = "cuda:0"
for model in ([a for a in sys.argv[1:] if a in ("7b", "13b")] or ["7b", "13b"]):
    eng = VddLlavaEngine(f"llava-1.5-{model}", device=dev, use_graph=True)
    for nq, both in ((1, False), (1, True), (2, True), (3, False), (4, False), (5, False), (6, False), (7, False), (4, True), (5, True)):
        ids, imgs = pope_prompts(nq, per_img=1, seed=5)
        kw = dict(images=imgs, use_dd=both, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=3, max_new_tokens=48)
        rows = nq * (3 if both else 2)
        rec = {"model": model, "rows": rows}
        for form in ("fused", "plain", None):
            ops.FORCE_LAYER_FORM = form
            eng._graphs.clear()
            best = 1e9
            for rep in range(4):
                eng.call_log = []
                eng.generate(ids, **kw)
                t = eng.call_timing(eng.call_log[-1])
                if rep:
                    best = min(best, t["decode_ms"] / t["decode_steps"])
            rec[form or "measured_pick"] = round(best, 3)
        ops.FORCE_LAYER_FORM = None
        rec["pick"] = ops._form_choice.get(("layer", rows, eng.cfg.lm.d, 0, 2))
        print(json.dumps(rec), flush=True)
    del eng
    torch.cuda.empty_cache()
