"""Bisect of the rocprofv3 abort on tools/config3_probe.py (profiles/README.md): run one STAGE of that probe per rocprofv3 invocation.
  rocprofv3 --kernel-trace -d /tmp/x -- python tools/rocprof_abort_bisect.py <stage>
stages: weights (torch only: the 13B random weights), vit (engine + one ViT batch), gen1 (one new token, 90 questions x 3 branches),
gen8 (8 tokens: decode graph capture + replays), gen8_eager (the same without graphs), gen8_7b (7B widths), gen<N>[x2][_eager]: N new tokens (twice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
stage = sys.argv[1]
dev = "cuda:0"
from llava_align_amd.engine import LlavaWeights, VddLlavaEngine, preset
name = "llava-1.5-7b" if stage.endswith("7b") else "llava-1.5-13b"
if stage == "weights":
    w = LlavaWeights.random(preset(name), dev)
    torch.cuda.synchronize(); print("stage weights ok", flush=True); sys.exit(0)
rng = np.random.default_rng(5)
sys_tok = [1] + rng.integers(3, 32000, size=34).tolist()
ids, imgs = [], []
g = torch.Generator().manual_seed(3)
for q in range(90):
    n = int(np.clip(rng.normal(80, 30), 10, 170))
    ids.append(torch.tensor(sys_tok + [-200] + rng.integers(3, 32000, size=n).tolist()))
    imgs.append(torch.randn(3, 336, 336, generator=g))
eng = VddLlavaEngine(name, device=dev, use_graph="eager" not in stage)
print("engine built", flush=True)
if stage == "vit":
    eng.vit(torch.stack(imgs[:16])); torch.cuda.synchronize(); print("stage vit ok", flush=True); sys.exit(0)
n_new = 1 if stage == "gen1" else (int(stage[3:].split("_")[0].split("x")[0]) if stage[3:4].isdigit() else 8)
for rep in range(2 if "x2" in stage else 1):
  out = eng.generate(ids, images=imgs, use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=n_new, seed=1)
torch.cuda.synchronize()
print("stage", stage, "ok", out.stats, flush=True)
