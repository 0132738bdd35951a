"""A/B of the 64 x 128 GEMM tiles (configs 12 / 13, round 6) on the 17 - 128-row decode step: the tuner with and without them, fresh tuning in both runs
(VDD_GEMM_DEFAULTS=off, no cache file), 7B, use_dd_unk, 6 questions per image.  python tools/gemm_small_tile_ab.py  -> one JSON line per (variant, rows)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    sys.path.insert(0, ROOT)
    import torch
    from llava_align_amd import ops
    from llava_align_amd.engine import VddLlavaEngine
    from bench import pope_prompts
    eng = VddLlavaEngine("llava-1.5-7b", device="cuda:0", use_graph=True)
    for nq in ((17, 20, 24, 33, 40, 48, 64) if os.environ.get("VDD_GEMM_SMALL_TILE_ROWS", "64,32") != "64,32" or "again" in sys.argv[2] or sys.argv[2].startswith("tiles_to") else (1, 2, 4, 6, 9, 12, 16, 24, 32, 48, 64)):
        ids, imgs = pope_prompts((nq + 5) // 6, per_img=6, seed=5)
        ids, imgs = ids[:nq], imgs[:nq]
        kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=3, max_new_tokens=48)
        best = 1e9
        for rep in range(4):
            eng.call_log = []
            eng.generate(ids, **kw)
            t = eng.call_timing(eng.call_log[-1])
            if rep:
                best = min(best, t["decode_ms"] / t["decode_steps"])
        picks = {k: v & 15 for k, v in ops.gemm_choices_export().items() if not k.startswith("form") and k.split(",")[0] in ("-1", "1", "2")}
        print(json.dumps({"variant": sys.argv[2], "rows": 2 * nq, "ms_per_step": round(best, 3), "tile_of_small_buckets": picks}), flush=True)
else:
    variants = (("all_tiles", "", "64,32"), ("no_32x128", "14,15", "64,32"), ("no_small_tiles", "12,13,14,15", "64,32"))
    if "--wide" in sys.argv:      # may the 64- / 32-row tiles also serve 65 - 128 / 33 - 64 rows (two row tiles per column panel)?
        variants = (("tiles_to_64_32_rows", "", "64,32"), ("tiles_to_128_64_rows", "", "128,64"), ("tiles_to_64_32_rows_again", "", "64,32"))
    for variant, excl, lim in variants:
        env = dict(os.environ, VDD_GEMM_DEFAULTS="off", VDD_GEMM_CHOICES="off", VDD_GEMM_EXCLUDE=excl, VDD_GEMM_SMALL_TILE_ROWS=lim)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", variant], env=env, check=False)
