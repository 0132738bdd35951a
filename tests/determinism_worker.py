"""Worker of tests/test_determinism_gpu.py: one fresh process = one engine (LLaVA-1.5-7B widths, 2 decoder layers, seeded weights), one
SAMPLED generation (seeded) at 96 decode rows and one at 1,536 rows; writes the generated ids (and the GEMM choices in effect) as JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def prompts(n_img, seed, vocab=32000):
    rng = np.random.default_rng(seed)
    sys_tok = [1] + rng.integers(3, vocab, size=34).tolist()
    ids, imgs = [], []
    for i in range(n_img):
        im = torch.randn(3, 336, 336, generator=torch.Generator().manual_seed(700 + i))
        for _ in range(6):
            ids.append(torch.tensor(sys_tok + [-200] + rng.integers(3, vocab, size=int(rng.integers(19, 29))).tolist()))
            imgs.append(im)
    return ids, imgs


def main(out_path, dtype):
    from llava_align_amd import ops
    from llava_align_amd.engine import LlavaConfig, LlavaWeights, LMConfig, VddLlavaEngine, VisionConfig
    dev = "cuda:0"
    dt = getattr(torch, dtype)
    cfg = LlavaConfig(LMConfig(n_layers=2, max_pos=1024), VisionConfig(layers=3), "determinism")
    eng = VddLlavaEngine(cfg, weights=LlavaWeights.random(cfg, dev, seed=11, std=0.02, lm_head_gain=2.0, dtype=dt), device=dev, use_graph=True)
    res = {}
    for n_img in (8, 128):                                    # 48 questions = 96 rows; 768 questions = 1,536 rows
        ids, imgs = prompts(n_img, seed=40 + n_img)
        out = eng.generate(ids, images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=6, seed=123)
        res[f"rows{2 * len(ids)}"] = out.tokens.cpu().tolist()
    res["choices"] = ops.gemm_choices_export()
    with open(out_path, "w") as f:
        json.dump(res, f)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "bfloat16")
