"""generate_list on the 360-question LLaVA-Bench-shaped list of bench.py (`llava_bench_eos.list_of_360`) at several `in_flight` values: the
own-KV pools grow with it (3 x in_flight slots of suffix + max_new_tokens), the tokens per decode step too.  Record: profiles/r06_list_in_flight.jsonl"""
import sys, os, json, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import bench
from llava_align_amd.engine import VddLlavaEngine
dev = torch.device("cuda:0")
eng = VddLlavaEngine("llava-1.5-7b", device=dev, seed=0, use_graph=True, lm_head_gain=4.0)
ids4, imgs4 = bench.pope_prompts(360, per_img=1, seed=778)
imgs4 = [im.to(dev).to(eng.dtype) for im in imgs4]
eos = sorted(set(np.random.default_rng(5).integers(3, 32000, size=250).tolist()))
kw = dict(use_dd=True, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=1.0, top_p=0.9, max_new_tokens=512, eos_token_id=eos, pad_token_id=0, seed=11, sync_every=8)
for n_q in [int(a) for a in sys.argv[1:]] or [90, 135, 160]:
    eng._kvs.clear(); eng._graphs.clear(); eng._kv = None
    torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    eng.generate_list(ids4, imgs4, in_flight=n_q, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = eng.generate_list(ids4, imgs4, in_flight=n_q, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps({"in_flight": n_q, "seconds": round(dt, 2), "tokens_per_s": round(o.stats["answer_tokens"] / dt, 1), "hbm_peak_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1),
                      **{k: v for k, v in o.stats.items() if k in ("admissions", "steps", "tail_shrinks", "admit_gpu_s", "mean_live_rows", "answer_tokens")}}), flush=True)
