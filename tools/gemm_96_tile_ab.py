"""A/B of the 96-row GEMM tiles (configs 16 - 18, round 6) on decode steps of a few hundred rows: BASELINE config #3's 270 rows (13B and 7B shapes,
3 branches, one image per question) and neighbours; the tuner with and without them, fresh tuning in both runs (VDD_GEMM_DEFAULTS=off, no cache file).
python tools/gemm_96_tile_ab.py   -> one JSON line per (variant, model, rows).  Record: profiles/r06_gemm_96_tile_ab.jsonl"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, ROOT)
    import torch
    from llava_align_amd import ops
    from llava_align_amd.engine import VddLlavaEngine
    from bench import pope_prompts
    for model in ("13b", "7b"):
        eng = VddLlavaEngine(f"llava-1.5-{model}", device="cuda:0", use_graph=True)
        for nq in (30, 60, 90, 120):
            ids, imgs = pope_prompts(nq, per_img=1, seed=5)
            imgs = [im.to("cuda:0").to(torch.bfloat16) for im in imgs]
            kw = dict(images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, seed=3, max_new_tokens=32)
            best = 1e9
            for rep in range(3):
                eng.call_log = []
                eng.generate(ids, **kw)
                t = eng.call_timing(eng.call_log[-1])
                if rep:
                    best = min(best, t["decode_ms"] / t["decode_steps"])
            b = -(-3 * nq // 64)
            picks = {k.rsplit(",", 2)[0]: (v & 15) | (((v >> 6) & 3) << 4) for k, v in ops.gemm_choices_export().items() if not k.startswith("form") and k.split(",")[0] == str(b) and k.endswith(",2")}
            print(json.dumps({"variant": sys.argv[2], "model": model, "rows": 3 * nq, "ms_per_step": round(best, 3), "tiles": picks}), flush=True)
        del eng
        torch.cuda.empty_cache()
else:
    for variant, excl in (("with_96_row_tiles", ""), ("without", "16,17,18")):
        env = dict(os.environ, VDD_GEMM_DEFAULTS="off", VDD_GEMM_CHOICES="off", VDD_GEMM_EXCLUDE=excl)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", variant], env=env, check=False)
