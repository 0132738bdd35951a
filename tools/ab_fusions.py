"""A/B of the round-5 launch fusions on the bench workload (768 questions, 1,536 decode rows): RoPE + KV write inside the grouped
attention (LanguageModel.fuse_rope) and the residual add in the o / down projections' epilogue (fuse_resid).  Per setting: prefill +
first token (median of 3), decode step (64 - 2 new tokens), and how many of the 768 x 64 tokens equal the all-off run's (fuse_rope is
bit-exact by construction; fuse_resid changes the tile / schedule the tuner picks for the two projections, so low-order bits may move)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llava_align_amd.engine import VddLlavaEngine
from bench import pope_prompts
dev = torch.device("cuda:0")
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 768
eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=True, lm_head_gain=4.0)
ids, imgs = pope_prompts(max(1, nq // 6), seed=1234)
on_dev = {}
imgs = [on_dev.setdefault(id(im), im.to(dev).to(torch.bfloat16)) for im in imgs]
kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2, seed=1, n_top=10)
base = None
for rope, resid in ((False, False), (True, False), (False, True), (True, True), (False, False), (True, True)):
    eng.lm.fuse_rope, eng.lm.fuse_resid = rope, resid
    eng._graphs = {}
    def timed(n_new, reps):
        eng.generate(ids, max_new_tokens=n_new, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); out = eng.generate(ids, max_new_tokens=n_new, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], out
    t2, _ = timed(2, 3)
    t64, out = timed(64, 3)
    if base is None:
        base = out.tokens.clone()
    print(json.dumps({"fuse_rope": rope, "fuse_resid": resid, "questions": len(ids), "prefill_plus_first_token_s": round(t2, 4),
                      "decode_step_ms": round((t64 - t2) / 62 * 1e3, 3), "tokens_per_s": round(len(ids) * 64 / t64, 1),
                      "tokens_equal_to_all_off": round(float((out.tokens == base).float().mean()), 4)}), flush=True)
