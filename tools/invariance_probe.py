"""Batch-invariant mode: which of the prefill's batch-dependent PLANNING choices change a row's bits?  Under ops.batch_invariant() the
same 24 questions (6 images x 4) are decoded with (a) packs of four short suffixes per attention workgroup on / off, (b) two-level prefixes
(system prompt prefilled once) on / off, (c) prompt-prefix sharing on / off; every step's score rows compared bit for bit.  A form that
comes out identical may stay on in the mode.  Record: profiles/r06_invariance_probe.jsonl"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from llava_align_amd import ops
from test_engine_shapes_gpu import _engine, _prompts

W7B = dict(d=4096, n_heads=32, n_kv_heads=32, head_dim=128, ffn=11008, vocab=32000)
eng = _engine(W7B, n_layers=3, vit_layers=2)
ids, imgs = _prompts(6, 4, 32000, seed=71)
kw = dict(images=imgs, cd_alpha=1.0, cd_beta=0.1, cd_greedy=True, output_scores=True, max_new_tokens=6, use_dd=True, use_dd_unk=True, temperature=1.0, top_p=0.9)


def run(**over):
    eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
    return eng.generate(ids, **dict(kw, **over))


def same(a, b):
    bad = sum(not torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in zip(a.scores, b.scores))
    return {"steps_differing": bad, "step0_identical": bool(torch.equal(a.scores[0].view(torch.int16), b.scores[0].view(torch.int16))),
            "tokens_identical": bool(torch.equal(a.tokens, b.tokens))}


with ops.batch_invariant():
    base = run()
    print(json.dumps({"what": "run to run", **same(base, run())}))
    ops.FLASH_PACKS_IN_INVARIANT_MODE = True
    print(json.dumps({"what": "packs of four suffixes vs one sequence per workgroup", **same(base, run())}))
    ops.FLASH_PACKS_IN_INVARIANT_MODE = False
    eng.two_level_prefix = False
    print(json.dumps({"what": "one-level vs two-level prefixes", "two_level_prefill_tokens": base.stats["prefill_tokens"], **same(base, run())}))
    eng.two_level_prefix = True
    print(json.dumps({"what": "share_prefix False vs True", **same(base, run(share_prefix=False))}))
    # text-only prompts: the common system prompt as a prefix slot (engine._common_split) vs nothing shared
    tids = [torch.tensor([t for t in r.tolist() if t != -200]) for r in ids]
    tk = dict(cd_greedy=True, output_scores=True, max_new_tokens=6, temperature=1.0)
    eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
    a = eng.generate(tids, **tk)
    eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
    b = eng.generate(tids, share_prefix=False, **tk)
    print(json.dumps({"what": "text-only prompts: common-prefix slot vs nothing shared", "prefill_tokens": [a.stats["prefill_tokens"], b.stats["prefill_tokens"]], **same(a, b)}))
