"""Engine (bf16, hand-written kernels) vs the fp32 torch LLaVA at FULL depth: step-0 logit error against depth (1 / 8 / 16 / 32 decoder
layers of the same 7B-shaped weights), and token agreement over a decode run.  Prints JSON lines; tests/test_full_depth_gpu.py
asserts bounds derived from these numbers.  Usage: python tools/depth_probe.py [--rows-big]"""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vdd_oracle as O  # noqa: E402
from ref_llava import RefLlava  # noqa: E402
from llava_align_amd.engine import LlavaConfig, LlavaWeights, LMConfig, VddLlavaEngine, VisionConfig  # noqa: E402

DEV = "cuda:0"


def prompts(n_img, per_img, vocab, seed):
    rng = np.random.default_rng(seed)
    sys_tok = [1] + rng.integers(3, vocab, size=34).tolist()
    ids, imgs = [], []
    for i in range(n_img):
        im = torch.randn(3, 336, 336, generator=torch.Generator().manual_seed(900 + i))
        for _ in range(per_img):
            ids.append(torch.tensor(sys_tok + [-200] + rng.integers(3, vocab, size=int(rng.integers(19, 29))).tolist()))
            imgs.append(im)
    return ids, imgs


def main():
    full = LlavaConfig(LMConfig(n_layers=32, max_pos=1024), VisionConfig(layers=3), "depth-probe")
    w = LlavaWeights.random(full, DEV, seed=5, std=0.02, lm_head_gain=2.0)
    ref = RefLlava(w, device=DEV)
    ids, imgs = prompts(1, 6, 32000, seed=31)
    for L in (1, 8, 16, 32):
        cfg = LlavaConfig(LMConfig(n_layers=L, max_pos=1024), VisionConfig(layers=3), f"depth-{L}")
        wl = copy.copy(w); wl.cfg = cfg
        eng = VddLlavaEngine(cfg, weights=wl, device=DEV, use_graph=False)
        eng.generate(ids, images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, max_new_tokens=1, cd_greedy=True)
        got = eng.debug_logits0.float().cpu()
        ref.cfg = cfg
        errs, sig = [], []
        for q in range(len(ids)):
            unk = ids[q].clone(); unk[unk == -200] = 0
            for b, (i_, im) in enumerate(((ids[q], imgs[q][None]), (unk, None))):
                want = ref(input_ids=i_[None], images=im).logits[0, -1].float()
                errs.append((got[b * len(ids) + q] - want).abs().max().item()); sig.append(want.std().item())
        print(json.dumps({"layers": L, "logit_max_err": max(errs), "logit_mean_max_err": float(np.mean(errs)), "logit_sigma": float(np.mean(sig))}), flush=True)
        del eng
        torch.cuda.empty_cache()
    ref.cfg = full


if __name__ == "__main__":
    main()
