"""Round 6: the decode-step curves with MEASURED projection / layer forms (in-tree defaults form_choices_mi355x.json) against the round-5 curves made
with the crossover literals (profiles/r05_step_curve_final.jsonl: 7B, 2 branches, 6 questions per image; r05_13b_step_policy.jsonl: 13B, 3 branches).
python tools/step_curve_check.py [7b] [13b]   Record: profiles/r06_step_curve_forms.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
dev = "cuda:0"
which = [a for a in sys.argv[1:] if a in ("7b", "13b")] or ["7b", "13b"]
r05 = {}
for l in open(os.path.join(ROOT, "profiles", "r05_step_curve_final.jsonl")):
    try:
        d = json.loads(l)
        r05[("7b", d["rows"])] = d["ms_per_step"]
    except (ValueError, KeyError):
        pass


def step_ms(eng, ids, kw):
    best = 1e9
    for rep in range(4):
        eng.call_log = []
        eng.generate(ids, **kw)
        t = eng.call_timing(eng.call_log[-1])
        if rep:
            best = min(best, t["decode_ms"] / t["decode_steps"])
    return best


for model in which:
    eng = VddLlavaEngine(f"llava-1.5-{model}", device=dev, use_graph=True)
    if model == "7b":
        for nq in (1, 2, 3, 4, 6, 8, 9, 12, 16, 17, 24, 32, 33, 48, 64, 96, 128):
            n_img = (nq + 5) // 6
            ids, imgs = pope_prompts(n_img, per_img=6, seed=5)
            ids, imgs = ids[:nq], imgs[:nq]
            ms = step_ms(eng, ids, dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=3, max_new_tokens=48))
            old = r05.get(("7b", 2 * nq))
            print(json.dumps({"model": model, "rows": 2 * nq, "ms_per_step": round(ms, 3), "r05_ms_per_step": old, "ratio": round(ms / old, 3) if old else None}), flush=True)
    else:
        for nq in (1, 2, 3, 5, 8, 11, 12):                     # 3, 6, 9, 15, 24, 33, 36 rows: r05 7.17 ms at 9 rows, 7.9 at 15, 8.5 at 24, 9.25 at 33, one question 6.2 ms
            ids, imgs = pope_prompts(nq, per_img=1, seed=5)
            ms = step_ms(eng, ids, dict(images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, seed=3, max_new_tokens=48))
            print(json.dumps({"model": model, "rows": 3 * nq, "ms_per_step": round(ms, 3)}), flush=True)
    del eng
    torch.cuda.empty_cache()
