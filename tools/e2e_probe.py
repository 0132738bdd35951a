"""Timing probe of the engine at LLaVA-1.5-7B shapes (synthetic weights / prompts)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from llava_align_amd.engine import VddLlavaEngine, preset
from llava_align_amd import ops

dev = "cuda:0"


def pope_prompts(n_img, per_img=6, seed=1234, vocab=32000, n_sys=35, txt=(19, 29)):
    rng = np.random.default_rng(seed)
    sys_tok = [1] + rng.integers(3, vocab, size=n_sys - 1).tolist()
    ids, imgs = [], []
    g = torch.Generator().manual_seed(7)
    for i in range(n_img):
        im = torch.randn(3, 336, 336, generator=g)
        for _ in range(per_img):
            t = rng.integers(3, vocab, size=int(rng.integers(*txt))).tolist()
            ids.append(torch.tensor(sys_tok + [-200] + t))
            imgs.append(im)
    return ids, imgs


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    t0 = time.perf_counter()
    eng = VddLlavaEngine("llava-1.5-7b", device=dev, t_max=768, use_graph=(os.environ.get("VDD_GRAPH", "1") == "1"))
    torch.cuda.synchronize()
    print("weights", round(eng.w.nbytes() / 1e9, 2), "GB; init", round(time.perf_counter() - t0, 1), "s", flush=True)
    if what in ("all", "gemm"):
        w = eng.w.t["l0.wgu"]
        for M in (1, 2, 4, 8, 16, 32, 64, 128, 256):
            x = torch.randn(M, 4096, device=dev).bfloat16()
            tm = timeit(lambda: torch.matmul(x, w.t()), 20)
            ts = timeit(lambda: ops.skinny_gemm(x, w), 20) if M <= 64 else float("nan")
            print(f"gate_up M={M}: torch {tm*1e6:.1f} us ({w.numel()*2/tm/1e9:.0f} GB/s)  skinny {ts*1e6:.1f} us ({w.numel()*2/ts/1e9:.0f} GB/s)", flush=True)
    if what in ("all", "e2e"):
        cfgs = ((16, 64), (32, 64), (64, 64), (128, 64)) if len(sys.argv) < 3 else ((int(sys.argv[2]), 64),)
        for n_img, new in cfgs:
            ids, imgs = pope_prompts(n_img)
            kw = dict(images=imgs, use_dd_unk=True, cd_alpha=1.0, cd_beta=0.1, temperature=0.2)
            for _ in range(2):                       # warm both graph keys
                eng.generate(ids, max_new_tokens=new, **kw); eng.generate(ids, max_new_tokens=2, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); out = eng.generate(ids, max_new_tokens=new, **kw); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            t1 = time.perf_counter(); eng.generate(ids, max_new_tokens=2, **kw); torch.cuda.synchronize(); dt2 = time.perf_counter() - t1
            Q = len(ids)
            step = (dt - dt2) / (new - 2)
            print(json.dumps({"Q": Q, "new": new, "total_s": round(dt, 3), "gen2_s": round(dt2, 3), "decode_ms_per_step": round(step * 1e3, 2),
                              "tok_per_s": round(Q * new / dt, 1), "decode_tok_per_s": round(Q / step, 1), "stats": out.stats}), flush=True)
